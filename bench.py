#!/usr/bin/env python3
"""bench.py -- defended queries/sec of the TVC hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic queries with
inputs resident in HBM: encode B images + B*(N+1) token rows with the HIP CLIP
towers, exact top-k search of all B*(N+2) embedding rows against the bank,
per-query consistency scores, records copied to the host.  Default workload =
BASELINE.json configs[2]: ViT-L/14 bf16, B=512, N=8, 1M-row bank, one GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: queries are data-parallel (encoders replicated, bank replicated per
GPU, no collective on the data path) -> "scaling": "weak".  Rank 0 prints ONE
JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_DENSE_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="ViT-L/14")
    p.add_argument("--batch", type=int, default=512, help="queries per GPU per step")
    p.add_argument("--variants", type=int, default=8)
    p.add_argument("--bank-rows", type=int, default=1_000_000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--no-profile-pass", action="store_true")
    p.add_argument("--shard-bank", action="store_true",
                   help="BASELINE configs[3]: --bank-rows is the GLOBAL bank, row-sharded over the ranks; "
                        "partial top-k lists are exchanged with RCCL (all-gather + all-to-all) and merged on the GPU")
    p.add_argument("--rehearse-one-gpu", action="store_true",
                   help="multi-rank rehearsal on a ONE-GPU box: every rank uses cuda:0 and the process group is "
                        "gloo (RCCL refuses two ranks on one device); checks the launch contract, not speed")
    p.add_argument("--serial-towers", action="store_true",
                   help="encode text then images on ONE stream (default: two streams, the towers overlap)")
    p.add_argument("--chunk-images", type=int, default=0, help="images per tower pass (0 = library default)")
    p.add_argument("--no-prefix-sharing", action="store_true",
                   help="encode every variant in full (default: variants share the rows of the token prefix they have "
                        "in common with their original - bit-identical, see TVC_OPT_TEXT_GROUP)")
    p.add_argument("--dense-text", action="store_true",
                   help="run the text tower on all 77 positions (disable EOT packing)")
    return p.parse_args()


def host_cpus() -> int:
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(pkg, arch, weights, images, tokens, bank_cpu, budget_s):
    """The oracle (CPU restatement of the reference path, PyTorch-CPU fp32 towers +
    numpy scores) timed on this box's host cores on a bounded sample of the same
    workload, de-duplicated schedule (image encoded once per query)."""
    import numpy as np
    import torch
    from oracle import clip_oracle, tvc_oracle
    vw, tw = weights
    torch.set_num_threads(host_cpus())      # more threads than the cgroup quota only thrash

    def one(i):
        with torch.no_grad():
            fi = clip_oracle.vision_forward(vw, images[i:i + 1], arch.vision.heads, arch.patch)
            ft = clip_oracle.text_forward(tw, tokens[i], arch.text.heads)
        return tvc_oracle.detect_batch(fi.numpy(), ft.numpy()[None], bank_cpu)

    one(0)                      # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    done = 0
    while done < images.shape[0]:
        one(done)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, dt


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("multimodal-detection-consistency_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.rehearse_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or a.shard_bank:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if a.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    arch = pkg.get_arch(a.model)
    B, N, R, D = a.batch, a.variants, a.bank_rows, arch.embed_dim
    weights = pkg.synth.make_clip_weights(arch, seed=0)
    eng = pkg.TVCEngine(arch, weights[0], weights[1], device=str(dev))
    images = pkg.synth.make_images(B, arch.image_size, seed=1 + rank).to(dev)
    tokens = pkg.synth.make_tokens(B, N, arch.ctx, seed=2 + 1000 * rank).to(dev)
    if a.shard_bank:
        lo, hi = pkg.sharding.shard_bounds(R, world, rank)
        bank = pkg.synth.make_bank(hi - lo, D, seed=7 + rank, device=str(dev), dtype=torch.bfloat16)
        sharded = pkg.sharding.ShardedBankSearch(pkg.sharding.HipShardOps(eng, lo, 0.3),
                                                 rows_per_shard=(R + world - 1) // world)
    else:
        bank = pkg.synth.make_bank(R, D, seed=7, device=str(dev), dtype=torch.bfloat16)
    eng.set_bank(bank)
    if a.dense_text:
        eng.set_option(pkg._lib.TVC_OPT_TEXT_PACKING, 0)
    if a.chunk_images:
        eng.set_option(pkg._lib.TVC_OPT_MAX_CHUNK_IMAGES, a.chunk_images)
    cfg = pkg.ConsistencyConfig()
    k = max(cfg.search_k, cfg.reference_count)

    s_img, s_txt = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def step(serial=False):
        if a.serial_towers or serial:
            ft = eng.encode_text(tokens.view(B * (N + 1), arch.ctx), group=0 if a.no_prefix_sharing else N + 1)   # first: its one row-count read-back
            fi = eng.encode_image(images)                               # happens while the GPU is still idle
        else:
            # The towers are independent until the bank search: run them on two streams.  The HBM-bound
            # row kernels / attention of one tower then overlap the MFMA-bound GEMMs of the other, and
            # a GEMM's last partial round of tiles no longer leaves CUs idle.
            main = torch.cuda.current_stream()
            s_txt.wait_stream(main); s_img.wait_stream(main)
            with torch.cuda.stream(s_txt):
                ft = eng.encode_text(tokens.view(B * (N + 1), arch.ctx), group=0 if a.no_prefix_sharing else N + 1)
            with torch.cuda.stream(s_img):
                fi = eng.encode_image(images)
            main.wait_stream(s_txt); main.wait_stream(s_img)
        rows = torch.cat([fi, ft])                                  # M = B*(N+2) query-side rows
        if a.shard_bank:
            # every rank searches ALL ranks' rows on its shard; partials go back to the rows' owners
            idx, sim, feat = sharded.search(rows, k, cfg.reference_count)
            rec = eng.consistency(fi, ft.view(B, N + 1, D), cfg, idx[B:].contiguous(), sim[B:].contiguous(),
                                  feat[B:].contiguous())
            return rec.cpu(), idx[:B].cpu()
        idx, sim, _ = eng.bank_search(rows, k, cfg.similarity_threshold, want_moments=False)
        tidx, tsim = idx[B:], sim[B:]
        feat = eng.bank_gather(tidx[:, :cfg.reference_count].contiguous())
        rec = eng.consistency(fi, ft.view(B, N + 1, D), cfg, tidx.contiguous(), tsim.contiguous(), feat)
        return rec.cpu(), idx[:B].cpu()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    eng.bank_status()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rec, _ = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if a.rehearse_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.bank_status()
    assert torch.isfinite(rec[:, :11]).all()      # words >= 12+N hold int32 bit patterns (-1 = NaN bits)

    roof = None
    prof = None
    if not a.no_profile_pass:
        # separate pass with HIP events around every launch (not part of the timed region)
        eng.profile_begin()
        step(serial=True)       # one stream: kernel durations not inflated by the other tower's kernels
        prof = eng.profile_end()
        g = prof["gemm"]
        achieved = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        roof = {"bound": "mfma", "kernel": "gemm_ring_kernel<EPI> (all tvc GEMM launches of a step: ring + the few small gemm_bf16_kernel ones)", "achieved": round(achieved, 2),
                "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4),
                "traffic": None, "launches_per_step": g["launches"],
                "avg_launch_ms": round(g["ms"] / max(g["launches"], 1), 4)}

    if roof is not None and rank == 0:
        # HBM bytes per launch of the dominant kernel family from the committed PMC summary of this
        # command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; bench.py cannot collect counters)
        try:
            import glob
            import json as _json
            here = os.path.dirname(os.path.abspath(__file__))
            files = sorted(glob.glob(os.path.join(here, "profiles", "*hbm_traffic_pmc.json")))
            if files and a.model == "ViT-L/14" and B == 512 and N == 8 and R == 1_000_000:
                pm = _json.load(open(files[-1]))
                ring = [(v["launches"], v["hbm_MB_per_launch_corrected"]) for k, v in pm.items()
                        if k.startswith("gemm_ring_kernel")]
                if ring:
                    roof["traffic"] = round(sum(n * mb for n, mb in ring) / sum(n for n, _ in ring) / 1e3, 3)
                    roof["traffic_unit"] = "GB per launch (PMC, launch-weighted mean over the ring GEMMs)"
                    roof["traffic_source"] = "profiles/" + os.path.basename(files[-1])
        except Exception:       # a missing / unreadable summary leaves traffic null
            pass

    if rank == 0:
        qps = world * B * a.steps / dt
        flops_q = arch.flops_image() + (N + 1) * arch.flops_text() + 2.0 * (N + 2) * R * D
        # executed MFMA work of one step (the text tower runs on the packed rows only; the bank GEMM
        # multiplies two bf16 planes per query row)
        exec_flops = None
        if prof:
            exec_flops = prof["gemm"]["work"] + prof["attention"]["work"] + prof["bank"]["work"]
        out = {
            "metric": "defended queries/sec", "value": round(qps, 2), "unit": "queries/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{a.model} bf16, batch={B}/GPU, N={N} variants, {R}-row bf16 bank, "
                                   f"encode + exact top-{k} bank search + consistency (BASELINE configs[2])",
                       "global_batch": world * B, "parallelism": (f"dp{world} queries x bank rows sharded {world}-way (RCCL all-gather + all-to-all of partial top-k)"
                                       if a.shard_bank else f"dp{world}"),
                       "text_packing": "dense-77" if a.dense_text else ("eot-packed" + ("" if a.no_prefix_sharing else " + variant prefix sharing") + " (bit-identical, see DESIGN.md)"),
                       "algorithmic_gflop_per_query_dense": round(flops_q / 1e9, 2),
                       "executed_gflop_per_query": round(exec_flops / B / 1e9, 2) if exec_flops else None,
                       "executed_path_tflops": round(qps * exec_flops / B / 1e12, 1) if exec_flops else None},
            "roofline": roof,
        }
        if prof:
            out["kernel_ms_per_step"] = {c: round(v["ms"], 2) for c, v in prof.items()}
            bk = prof["bank"]
            if bk["ms"] > 0:
                out["bank_stage"] = {"mfma_tflops": round(bk["work"] / (bk["ms"] * 1e-3) / 1e12, 1),
                                     "bank_stream_GBps": round(R * D * 2 / (bk["ms"] * 1e-3) / 1e9, 1)}
        if not a.no_cpu_baseline and world == 1:
            import numpy as np
            nq = min(B, 64)
            bank_cpu = bank.float().cpu().numpy()
            v, done, secs = cpu_baseline(pkg, arch, weights, images[:nq].cpu(), tokens[:nq].cpu().long(), bank_cpu,
                                         a.cpu_seconds)
            out["cpu_baseline"] = {"value": round(v, 3), "unit": "queries/s", "cores": torch.get_num_threads(),
                                   "kind": "port",
                                   "sample": f"{done} queries of the same workload in {secs:.1f}s, oracle "
                                             f"(PyTorch-CPU fp32 towers + numpy scores), de-duplicated schedule"}
            out["speedup_vs_cpu"] = round(qps / v, 1) if v > 0 else None
        print(json.dumps(out), flush=True)
    if world > 1 or a.shard_bank:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
