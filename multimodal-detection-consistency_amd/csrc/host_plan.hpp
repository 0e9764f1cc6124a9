// Host-side planning arithmetic shared by the launchers (bank.hip, sd_ops.hip) and the C-ABI translation units: pure
// functions of sizes, no HIP.  Kept in a header so that the host-only sanitizer build of the C-ABI (tests/host_san:
// -fsanitize=address,undefined with the kernels stubbed out) runs the SAME arithmetic as the product.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define BANK_CAP 128
#define BANK_KEFF 16       // tau = max(k, 16)-th largest group maximum: a looser but far less noisy bound
#define HOST_PLAN_GEMM_BM 256
#define HOST_PLAN_GEMM_BN 256

inline void bank_plan(int64_t R, int M, int k, int* n_sample, int* sample_stride, int* S, int* cap) {
    const int64_t nbt = (R + HOST_PLAN_GEMM_BM - 1) / HOST_PLAN_GEMM_BM;
    const int nqt = (M + HOST_PLAN_GEMM_BN - 1) / HOST_PLAN_GEMM_BN;
    int64_t s = (1280 + nqt - 1) / nqt;
    if (s > nbt) s = nbt;
    if (s < 1) s = 1;
    const int64_t tpc = (nbt + s - 1) / s;
    s = (nbt + tpc - 1) / tpc;
    // expected survivors per query ~ keff * R / n_sample (relative spread ~ keff^-1/2):
    // aim at 8 per chunk list (cap 128) and at most ~2048 per query (select pool 6144)
    const int keff = k > BANK_KEFF ? k : BANK_KEFF;
    int64_t ns = (int64_t)keff * R / (8 * s);
    // the select pass re-scores every survivor of a query in ONE workgroup (~2 us per 16 rows): ~2 000 survivors cost 0.3-0.5 ms
    // there whatever M is -- more than streaming a 1 M-row bank -- while the sample's GEMM and the tau selection over it grow
    // with M: aim at ~256 survivors per query up to M = 2 048 (round 4; measured at R = 1 M, k = 10, M = 128 / 256 / 1 024 / 2 048:
    // 1.21 / 1.12 / 2.73 / 4.15 -> 0.57 / 0.62 / 1.73 / 3.26 ms; at M = 4 608 the two costs cancel: 6.9 -> 7.3 ms at k = 5)
    const int64_t ns2 = (int64_t)keff * R / (M <= 2048 ? 256 : 2048);
    if (ns < ns2) ns = ns2;
    if (ns < 4096) ns = 4096;
    // cap: the pre-pass similarities [M, ns] fp32 stay under 8 GiB (ns = 262144 at M = 5120 keeps a
    // 10 M-row bank at ~600 survivors per query; the old 65536 cap left ~2800 +- 25 % and overflowed
    // the 128-entry lists / 6144-entry pool for some of 5120 queries)
    int64_t ns_cap = ((int64_t)8 << 30) / ((int64_t)(M > 0 ? M : 1) * 4);
    if (ns_cap > 262144) ns_cap = 262144;
    if (ns_cap < 65536) ns_cap = 65536;
    if (ns > ns_cap) ns = ns_cap;
    ns = (ns + 255) / 256 * 256;
    if (ns > R) ns = R;
    if (ns < 1) ns = 1;
    // when the cap bites, keep the per-(chunk, query) lists short (<= ~24 expected, cap 128) by
    // cutting the bank into more chunks instead
    {
        int64_t s_min = ((int64_t)keff * R + 24 * ns - 1) / (24 * ns);
        if (s_min > nbt) s_min = nbt;
        if (s < s_min) {
            const int64_t tpc2 = (nbt + s_min - 1) / s_min;
            s = (nbt + tpc2 - 1) / tpc2;
        }
    }
    *n_sample = (int)ns;
    *sample_stride = (int)(R / ns > 0 ? R / ns : 1);
    *S = (int)s;
    *cap = BANK_CAP;
}

// GroupNorm statistics slabs: small slabs = many workgroups (the pass is latency-bound on few), at most 1024 slabs per image
inline int gn_slab_tokens(int HW) { int s = 64; while ((HW + s - 1) / s > 1024) s *= 2; return s; }
// ws: >= n * nslab * groups * 2 + n * groups * 2 floats
inline size_t sd_groupnorm_ws_floats(int n, int HW, int groups) {
    const int slab = gn_slab_tokens(HW);
    const int nslab = (HW + slab - 1) / slab;
    return (size_t)n * nslab * groups * 2 + (size_t)n * groups * 2;
}
