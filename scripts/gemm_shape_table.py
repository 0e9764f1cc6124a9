"""Per-shape table of the GEMM launches of one profiled pass (TVC_PROF_DUMP=<file> + eng.profile_begin/_end): launches,
ms, useful TFLOP/s, tiles, tile efficiency (useful rows / multiplied rows) and rounds on 256 workgroups -- where a
model's GEMM time goes.  python scripts/gemm_shape_table.py <dump file>"""
import collections, sys
rows = collections.defaultdict(lambda: [0, 0.0])
for ln in open(sys.argv[1]):
    I, J, K, P, S, ms = ln.split()
    key = (int(I), int(J), int(K), int(P), int(S))
    rows[key][0] += 1; rows[key][1] += float(ms)
tot = sum(v[1] for v in rows.values())
print(f"{'I':>6} {'J':>7} {'K':>6} {'pl':>2} {'S':>2} {'n':>5} {'ms':>8} {'%':>5} {'TF/s':>6} {'tiles':>6} {'rounds':>6} {'tile eff':>8}")
for (I, J, K, P, S), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    fl = 2.0 * I * J * K * P * n
    ti, tj = (I + 255) // 256, (J + 255) // 256
    print(f"{I:6d} {J:7d} {K:6d} {P:2d} {S:2d} {n:5d} {ms:8.2f} {100 * ms / tot:5.1f} {fl / ms / 1e9:6.0f} {ti * tj:6d} {ti * tj * max(S, 1) / 256:6.2f} "
          f"{I * J / (ti * tj * 65536):8.2f}")
print(f"total {tot:.1f} ms")
