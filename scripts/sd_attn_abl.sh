#!/bin/bash
# A/B of sd_flash_attention builds on one box: the product library and every gpurun_abl/libtvc_sda_*.so through scripts/sd_attn_bench.py
for f in "" gpurun_abl/libtvc_sda_*.so; do
  echo "== ${f:-product}"
  TVC_LIB_PATH=$f python scripts/sd_attn_bench.py 2>&1 | grep "dh="
done
