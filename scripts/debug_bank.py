"""GPU debug: large-scale bank search vs torch matmul (not part of the product)."""
import sys, time
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg

def run(R, M, D, k, seed=0):
    eng = pkg.TVCEngine()
    bank = pkg.synth.make_bank(R, D, seed=7, device="cuda:0", dtype=torch.bfloat16)
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    q = torch.randn((M, D), generator=g, device="cuda:0")
    q = q / q.norm(dim=-1, keepdim=True)
    eng.set_bank(bank)
    idx, sim, mom = eng.bank_search(q, k, 0.1)
    try:
        eng.bank_status(); st = "ok"
    except Exception as e:
        st = str(e)[:60]
    # reference: chunked fp32 matmul on bf16-exact bank (queries fp32 -> tf32-free fp32 matmul)
    best_v = torch.full((M, k), -2.0, device="cuda:0"); best_i = torch.zeros((M, k), dtype=torch.long, device="cuda:0")
    cnt = torch.zeros(M, device="cuda:0")
    for r0 in range(0, R, 1 << 17):
        b = bank[r0:r0 + (1 << 17)].float()
        s = q @ b.t()
        cnt += (s >= 0.1).sum(1)
        v, i = s.topk(min(k, s.shape[1]), dim=1)
        allv = torch.cat([best_v, v], 1); alli = torch.cat([best_i, i + r0], 1)
        o = allv.argsort(1, descending=True)[:, :k]
        best_v = allv.gather(1, o); best_i = alli.gather(1, o)
    dv = (sim - best_v).abs().max().item()
    di = (idx.long() != best_i).float().mean().item()
    dc = (mom[:, 3] - cnt).abs().max().item()
    print(f"R={R} M={M} D={D} k={k}: status={st} max|dsim|={dv:.2e} idx mismatch frac={di:.4f} count diff={dc}", flush=True)
    eng.close()

if __name__ == "__main__":
    run(100_000, 5120, 768, 5)
    run(1_000_000, 256, 768, 5)
    run(1_000_000, 5120, 768, 5)
    run(1_000_000, 5120, 768, 20)
