#!/bin/bash
# A/B of the persistent LDS-DMA ViT attention (attention_vit_stream_kernel) against the one-item kernel (TVC_ATT_STREAM=0), one box,
# three passes each.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo -n "one-item kernel: "; TVC_ATT_STREAM=0 python scripts/attn_bench.py 2>&1 | grep "T=257"
  echo -n "stream (product): "; python scripts/attn_bench.py 2>&1 | grep "T=257"
done
