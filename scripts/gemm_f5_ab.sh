#!/bin/bash
# A/B of the ring-GEMM experiment builds (gpurun_abl/libtvc_<name>.so, scripts/build_variant.sh) against the product on ONE
# box: bit-identity (checksums of gemm_form_check.py), the four tower shapes, the square shapes of the HIP guide.
cd $GRAFT_REPO_ROOT
for lib in product f5 f5u; do
  if [ $lib = product ]; then unset TVC_LIB_PATH; else export TVC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_abl/libtvc_$lib.so; fi
  echo "=== $lib"
  python scripts/gemm_form_check.py | grep -E "checksum|FORM_OK" | md5sum
  python scripts/gemm_shapes.py
  python scripts/gemm_shapes.py
  python scripts/gemm_square.py 2>&1 | grep -E "random|I=1024|I=4096 J=32768|I=3072"
done
