#!/bin/bash
# A/B of attention builds on one box: every gpurun_abl/libtvc_att_*.so (scripts/build_variant.sh style) through scripts/attn_bench.py
for f in "" gpurun_abl/libtvc_base.so gpurun_abl/libtvc_att_*.so; do
  echo "== ${f:-product}"
  TVC_LIB_PATH=$f python scripts/attn_bench.py 2>&1 | grep "T=257"
done
