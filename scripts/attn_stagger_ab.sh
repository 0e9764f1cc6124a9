#!/bin/bash
# A/B of a start stagger for the second resident workgroup of a compute unit in the ViT attention kernel (experiment builds
# -DTVC_ATT_STAGGER=<n>: n x 6 400 clocks) against the product on one box: scripts/attn_bench.py twice per build.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in product stg1 stg2 stg3 stg4; do   # built by: scripts/build_variant.sh stgN -DTVC_ATT_STAGGER=N on the experiment commit (git log: "attention stagger experiment")
  if [ $lib = product ]; then unset TVC_LIB_PATH; else export TVC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_abl/libtvc_$lib.so; fi
  echo -n "$lib: "; python scripts/attn_bench.py 2>&1 | grep "T=257"
done
done
