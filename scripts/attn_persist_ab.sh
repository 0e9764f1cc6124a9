#!/bin/bash
# A/B of the persistent ViT attention experiment (TVC_ATT_PERSIST=1, KPIN 0 / 2 / 4 pinned K tiles) against the product on one
# box: bit-identity of the outputs, then scripts/attn_bench.py twice per build.
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, subprocess, sys, torch
sys.path.insert(0, ".")
code = r'''
import sys, torch
sys.path.insert(0, ".")
import tvc_amd as pkg
eng = pkg.TVCEngine(); torch.manual_seed(0)
qkv = torch.randn((96 * 257, 3 * 16 * 64), device="cuda:0").to(torch.bfloat16)
o = eng.attention(qkv, 96, 257, 16, False)
print(int(o.view(torch.int16).to(torch.int64).sum().item()), float(o.float().abs().max()))
'''
for lib in ("", "attp0", "attp2", "attp4"):
    env = dict(os.environ)
    if lib: env["TVC_LIB_PATH"] = os.path.abspath(f"gpurun_abl/libtvc_{lib}.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    print(lib or "product", "checksum", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
PY
for rep in 1 2; do
for lib in product attp0 attp2 attp4; do
  if [ $lib = product ]; then unset TVC_LIB_PATH; else export TVC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_abl/libtvc_$lib.so; fi
  echo -n "$lib: "; python scripts/attn_bench.py 2>&1 | grep "T=257"
done
done
