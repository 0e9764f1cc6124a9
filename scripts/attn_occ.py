"""Occupancy probe of the ViT-L/14 attention kernel: 256 / 512 / 768 / 1024 (sequence, head) items = 1 / 2 / 3 / 4 workgroups per CU."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import tvc_amd as pkg
eng = pkg.TVCEngine(); dev = "cuda:0"; torch.manual_seed(0)
for n_seq in (16, 32, 48, 64, 128, 512):
    qkv = torch.randn((n_seq * 257, 3 * 16 * 64), device=dev).to(torch.bfloat16)
    for _ in range(3): eng.attention(qkv, n_seq, 257, 16, False)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): eng.attention(qkv, n_seq, 257, 16, False)
    t1.record(); torch.cuda.synchronize()
    print(f"items={n_seq*16}: {t0.elapsed_time(t1)/20*1e3:.1f} us", flush=True)
