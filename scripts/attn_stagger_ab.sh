#!/bin/bash
# A/B of a start stagger for the second resident workgroup of a compute unit in the ViT attention kernel against the product
# on one box: scripts/attn_bench.py twice per build.  The experiment builds (scripts/build_variant.sh stgN -DTVC_ATT_STAGGER=N)
# carried this block at the top of attention_kernel (never in the product; N x 6 400 clocks):
#     #ifdef TVC_ATT_STAGGER
#     if (EXACT && blockIdx.x < 512 && (__builtin_amdgcn_s_getreg(6148) & 1))       // HW_ID wave slot 1 = the CU's 2nd workgroup
#         for (int i = 0; i < TVC_ATT_STAGGER; ++i) __builtin_amdgcn_s_sleep(100);
#     #endif
# Result (profiles/r04_attention_stagger_ab.log): 292-293 us -> 283-286 us for N = 1 .. 4: -2.5 %, not kept.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in product stg1 stg2 stg3 stg4; do
  if [ $lib = product ]; then unset TVC_LIB_PATH; else export TVC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_abl/libtvc_$lib.so; fi
  echo -n "$lib: "; python scripts/attn_bench.py 2>&1 | grep "T=257"
done
done
