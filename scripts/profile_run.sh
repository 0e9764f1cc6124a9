set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${ROUND:-r03}a
mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
echo "--- kernel trace"
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt --output-format csv -- python3 bench.py --steps 5 --warmup 2 --serial-towers --no-cpu-baseline --no-extras > $O/kt_bench.json 2> $O/kt.err
ls $O/ktrace | head
echo "--- pmc mfma"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_mfma -o m --output-format csv -- python3 bench.py --steps 1 --warmup 1 --serial-towers --no-cpu-baseline --no-extras --no-profile-pass > $O/pmc_mfma.json 2> $O/pmc_mfma.err || echo "pmc mfma failed"
echo "--- pmc fetch"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --serial-towers --no-cpu-baseline --no-extras --no-profile-pass > $O/pmc_f.json 2> $O/pmc_f.err || echo "pmc f failed"
echo "--- pmc write"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --serial-towers --no-cpu-baseline --no-extras --no-profile-pass > $O/pmc_w.json 2> $O/pmc_w.err || echo "pmc w failed"
echo "--- calibration"
rocprofv3 --pmc FETCH_SIZE -d $O/cal_f -o f --output-format csv -- python3 scripts/pmc_calibrate.py > $O/cal_f.log 2>&1 || echo "cal f failed"
rocprofv3 --pmc WRITE_SIZE -d $O/cal_w -o w --output-format csv -- python3 scripts/pmc_calibrate.py > $O/cal_w.log 2>&1 || echo "cal w failed"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/cal_r -o r --output-format csv -- python3 scripts/pmc_calibrate.py > $O/cal_r.log 2>&1 || echo "cal r failed"
rocprofv3 -L > $O/counters.txt 2>&1 || true
find $O -name "*.csv" | head -30
echo "--- SD reference generator: kernel trace of one batched generation (12 images, 20 steps, 64 x 64 latents + VAE)"
rocprofv3 --kernel-trace --stats -d $O/sd_ktrace -o sd --output-format csv -- python3 scripts/sd_profile.py 20 > $O/sd_profile.json 2> $O/sd_profile.err || echo "sd trace failed"
tail -1 $O/sd_profile.json
