#!/bin/bash
# per-shape table of the GEMM launches of one batched SD generation (12 images x 20 steps): scripts/gemm_shape_table.py
cd $GRAFT_REPO_ROOT
rm -f /tmp/sd_gemm_dump.txt
TVC_PROF_DUMP=/tmp/sd_gemm_dump.txt python scripts/sd_profile.py 20 > /dev/null 2>&1
python scripts/gemm_shape_table.py /tmp/sd_gemm_dump.txt
