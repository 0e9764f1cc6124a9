"""Post-process the output of scripts/profile_run.sh (gpurun_out/<round>a) into profiles/<round>_*: copies the bench lines and
the kernel-trace stats, summarises the PMC passes (pmc_traffic.py, pmc_summary.py + derived fractions) and prints the
numbers the docs quote.  usage: python scripts/profiles_postprocess.py [r03]"""
import csv, json, os, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
O = f"gpurun_out/{R}a"
shutil.copy(f"{O}/bench.json", f"profiles/{R}_bench.json")
shutil.copy(f"{O}/kt_bench.json", f"profiles/{R}_bench_under_rocprof.json")
shutil.copy(f"{O}/ktrace/kt_kernel_stats.csv", f"profiles/{R}_kernel_stats.csv")
subprocess.run([sys.executable, "scripts/pmc_traffic.py", f"{O}/pmc_f/f_counter_collection.csv", f"{O}/pmc_w/w_counter_collection.csv",
                f"profiles/{R}_hbm_traffic_pmc.json"], check=True, stdout=subprocess.DEVNULL)
subprocess.run([sys.executable, "scripts/pmc_summary.py", f"{O}/pmc_mfma/m_counter_collection.csv", f"profiles/{R}_mfma_pmc.json"],
               check=True, stdout=subprocess.DEVNULL)
d = json.load(open(f"profiles/{R}_mfma_pmc.json"))
for k, v in d.items():
    if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_WAVE_CYCLES"):
        cyc = v["GRBM_GUI_ACTIVE"] / 8          # the counter sums the 8 XCDs
        v["derived"] = {"gpu_cycles_per_launch": round(cyc),
                        "mfma_pipe_busy_frac_of_1024_simds": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),
                        "wave_wait_any_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4),
                        "wave_wait_inst_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4),
                        "wave_active_inst_frac": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4)}
json.dump(d, open(f"profiles/{R}_mfma_pmc.json", "w"), indent=1)
def cal(path, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == name and "gemm_ring" in r["Kernel_Name"]]
    return sum(vals) / len(vals)
c = json.load(open("profiles/r02_pmc_calibration.json"))       # known byte counts of the calibration launch (scripts/pmc_calibrate.py)
f, w = cal(f"{O}/cal_f/f_counter_collection.csv", "FETCH_SIZE"), cal(f"{O}/cal_w/w_counter_collection.csv", "WRITE_SIZE")
c["FETCH_SIZE_raw_KB_per_launch"] = round(f, 1); c["WRITE_SIZE_KB_per_launch"] = round(w, 1)
c["fetch_raw_bytes_over_known"] = round(f * 1024 / c["known_read_bytes"], 4)
c["fetch_x2_bytes_over_known"] = round(2 * f * 1024 / c["known_read_bytes"], 4)
c["write_bytes_over_known"] = round(w * 1024 / c["known_write_bytes"], 4)
json.dump(c, open(f"profiles/{R}_pmc_calibration.json", "w"), indent=1)
b = json.load(open(f"profiles/{R}_bench.json")); u = json.load(open(f"profiles/{R}_bench_under_rocprof.json"))
print("bench:", b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"], b["kernel_ms_per_step"])
print("  api:", b["through_api"]["pipeline_detect_qps"], b["through_api"]["defense_batch_detect_qps"], "dense", b["dense_text_qps"], "host", b["host_inputs_qps"],
      "pgd", b["pgd_inner_loop"]["image_steps_per_s"], "cpu", b["cpu_baseline"]["value"], b["speedup_vs_cpu"], "bank", b["bank_stage"])
print("under rocprof:", u["value"], u["kernel_ms_per_step"], u["roofline"]["avg_launch_ms"])
rows = list(csv.DictReader(open(f"profiles/{R}_kernel_stats.csv")))
g = [r for r in rows if "gemm" in r["Name"]]
n = sum(int(r["Calls"]) for r in g); t = sum(float(r["TotalDurationNs"]) for r in g)
print("trace: gemm launches", n, "total ms", round(t / 1e6, 1), "avg ms", round(t / n / 1e6, 4))
for r in rows[:8]:
    print("  ", r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg")
for k in list(d)[:4]:
    print(k, d[k].get("derived"))
t = json.load(open(f"profiles/{R}_hbm_traffic_pmc.json"))
for k in list(t)[:4]:
    print(k, t[k])
print("calibration:", c["fetch_x2_bytes_over_known"], c["write_bytes_over_known"])

if os.path.exists(f"{O}/sd_ktrace/sd_kernel_stats.csv"):
    shutil.copy(f"{O}/sd_ktrace/sd_kernel_stats.csv", f"profiles/{R}_sd_kernel_stats.csv")
    shutil.copy(f"{O}/sd_profile.json", f"profiles/{R}_sd_profile.json")
    rows = list(csv.DictReader(open(f"profiles/{R}_sd_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("SD generation trace (two generations: warm-up + measured), total kernel ms", round(tot / 1e6, 1))
    for r in rows[:10]:
        print("  ", r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg", round(100 * float(r["TotalDurationNs"]) / tot, 1), "%")
    print(open(f"profiles/{R}_sd_profile.json").read().strip().splitlines()[-1])
