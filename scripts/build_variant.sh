#!/bin/bash
# Experiment builds of the library (never the product): scripts/build_variant.sh <name> <extra hipcc flags...>
# -> gpurun_abl/libtvc_<name>.so, selected at run time with TVC_LIB_PATH (see _lib.py).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/multimodal-detection-consistency_amd/csrc
out=$root/gpurun_abl; mkdir -p $out/obj_$name
for f in gemm elementwise attention bank consistency backward attention_bwd precise split sd_ops sd_attention; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" -c $src/$f.hip -o $out/obj_$name/$f.o &
done
for f in tvc_abi tvc_precise tvc_split tvc_sd; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip "$@" -c $src/$f.cpp -o $out/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libtvc_$name.so $out/obj_$name/*.o
echo built $out/libtvc_$name.so
