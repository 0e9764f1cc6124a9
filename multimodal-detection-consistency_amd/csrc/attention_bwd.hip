// Attention backward (dQ, dK, dV from dO) for the vision tower's input gradient -- what a PGD / Hubness
// attack needs (src/attacks/pgd_attack.py:456-486).  Non-causal, fixed-length
// sequences of T <= 288 tokens, head_dim 64: everything of one (sequence, head) fits in LDS, so the scores are
// recomputed (never stored) and the softmax is exact, as in the forward kernel (attention.hip), whose "swapped"
// MFMA idiom (the reduction-free index on the lane) both passes reuse:
//
//   pass A  (lane = query)  S^T = K Q^T,  P,  dP^T = V dO^T,  delta = rowsum(P * dP),  dS = P * (dP - delta) / 8,
//                           dQ^T = K^T dS^T;   writes dQ and the per-(row, head) softmax statistics
//   pass B  (lane = key)    S = Q K^T,  P (from the stored statistics),  dP = dO V^T,  dS,
//                           dV^T += dO^T P,  dK^T += Q^T dS   over all query tiles;  writes dK, dV
//
// Operands read along rows come from LDS by ds_read_b128, operands read along columns by ds_read_b64_tr_b16;
// ONE image per tensor with 160-byte rows serves both (conflict-free for both read kinds: rows 8 apart differ
// by 32 dwords, the 16 lanes a ds_read_b128 services together hit 16 distinct 16-byte slots).
#include "common.hpp"
#include "kernels.hpp"
#include <mutex>

#define ATT_DH 64
#define ATT_ROW 160
#ifndef DKV_NW
#define DKV_NW 8         // waves per workgroup of the 257-token dK / dV kernel (4: 741 us per call, 8: ~325)
#endif

namespace {

// Fill two [rows_pad][160 B] images with `T` rows of 64 bf16 each, taken from qkv-like rows (zero beyond T).  ALL loads of
// both images are issued before the first LDS write and none sits behind a condition (row index clamped, zeros selected at
// the write), so a workgroup pays one memory latency for its operands: the first cut -- a load, a wait and an LDS write
// per 16-byte piece and loop trip -- paid eighteen in series (attention.hip's fill, round 3).
// SWZ_B: image b has 128-byte rows with the 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (row reads only:
// conflict-free ds_read_b128, as attention.hip's K image) instead of the padded 160-byte rows both read kinds need.
template <int MAXROWS, bool SWZ_B = false, int NTHR = 256>
__device__ __forceinline__ void fill_two_images(char* lds_a, const uint16_t* __restrict__ src_a, int64_t ld_a, int col_a, int rows_a,
                                                char* lds_b, const uint16_t* __restrict__ src_b, int64_t ld_b, int col_b, int rows_b,
                                                int64_t row0, int T, int tid) {
    constexpr int NIT = (MAXROWS * 8 + NTHR - 1) / NTHR;
    u32x4_t va[NIT], vb[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
        const int rc = r < T ? r : T - 1;
        va[i] = __builtin_nontemporal_load((const u32x4_t*)(src_a + (row0 + rc) * ld_a + col_a + c * 8));
        vb[i] = __builtin_nontemporal_load((const u32x4_t*)(src_b + (row0 + rc) * ld_b + col_b + c * 8));
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
        const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
        if (r < rows_a) *(u32x4_t*)(lds_a + r * ATT_ROW + (c << 4)) = r < T ? va[i] : z;
        if (r < rows_b) {
            if (SWZ_B) *(u32x4_t*)(lds_b + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = r < T ? vb[i] : z;
            else *(u32x4_t*)(lds_b + r * ATT_ROW + (c << 4)) = r < T ? vb[i] : z;
        }
    }
}

__device__ __forceinline__ bf16x8_t row_frag(const char* img, int row, int ks, int g) {
    return *(const bf16x8_t*)(img + row * ATT_ROW + ((ks * 4 + g) << 4));
}

// A operand [16 x 32] = columns (dh md*16 ..) of the rows of two 16-row tiles t0, t1 (k-slots j < 4: tile t0 rows
// 4g + j, j >= 4: tile t1) -- the transposed read of attention.hip
__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int tr_off, int md, int t0, int t1) {
    const char* vb = img + tr_off + md * 32;
    const bf16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + t0 * 16 * ATT_ROW));
    const bf16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + t1 * 16 * ATT_ROW));
    bf16x8_t a;
    a[0] = v0[0]; a[1] = v0[1]; a[2] = v0[2]; a[3] = v0[3];
    a[4] = v1[0]; a[5] = v1[1]; a[6] = v1[2]; a[7] = v1[3];
    return a;
}

}  // namespace

// ---------------------------------------------------------------------------
// pass A: dQ and the softmax statistics.  stats fp32 [rows, heads, 4] = {max * c, 1 / sum, delta, 0}
// ---------------------------------------------------------------------------
template <int MAXT>
__global__ __launch_bounds__(256, 2) void attention_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dao,
                                                                  uint16_t* __restrict__ dqkv, float* __restrict__ stats,
                                                                  int T, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int width = heads * ATT_DH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous item order, as the forward kernel: the 16 heads of a sequence -- adjacent 128-byte pieces of every
    // packed row -- run on neighbouring CUs of one XCD at the same time (attention.hip)
    const int item = xcd_contiguous(blockIdx.x, gridDim.x);
    const int seq = item / heads, h = item - seq * heads;
    const int64_t row0 = (int64_t)seq * T, ld = 3 * (int64_t)width;
    const int NT = (T + 15) >> 4, NP = (NT + 1) >> 1;
    // K [NT*16][160]: row reads (S) and transposed reads (dQ); V [NT*16][128, swizzled]: row reads (dP).  78.3 KB at 257
    // tokens: two workgroups per CU (the first cut held 89.6 KB and one).
    char* ldsK = smem;
    char* ldsV = smem + NT * 16 * ATT_ROW;
    const int g = lane >> 4, r16 = lane & 15;
    const int NQ = NT;
    // the Q / dO fragments of a block come straight from HBM: fetched one block ahead (the first block's before the fill)
    bf16x8_t nq0, nq1, nd0, nd1;
    auto fetch = [&](int qb) __attribute__((always_inline)) {
        int qrow = qb * 16 + r16;
        qrow = qrow < T ? qrow : T - 1;
        const uint16_t* qp = qkv + (row0 + qrow) * ld + h * ATT_DH + 8 * g;
        const uint16_t* dp_ = dao + (row0 + qrow) * (int64_t)width + h * ATT_DH + 8 * g;
        nq0 = __builtin_nontemporal_load((const bf16x8_t*)qp); nq1 = __builtin_nontemporal_load((const bf16x8_t*)(qp + 32));
        nd0 = __builtin_nontemporal_load((const bf16x8_t*)dp_); nd1 = __builtin_nontemporal_load((const bf16x8_t*)(dp_ + 32));
    };
    fetch(wave < NQ ? wave : NQ - 1);
    fill_two_images<MAXT * 16, true>(ldsK, qkv, ld, width + h * ATT_DH, NT * 16, ldsV, qkv, ld, 2 * width + h * ATT_DH, NT * 16, row0, T, tid);
    __syncthreads();

    const int tr_off = (4 * g + (r16 >> 2)) * ATT_ROW + ((r16 & 3) << 3);
    const int vsw0 = ((0 + g) ^ ((lane >> 1) & 7)) << 4, vsw1 = ((4 + g) ^ ((lane >> 1) & 7)) << 4;     // swizzled chunk of the V image
    const float scale_log2 = 0.125f * 1.4426950408889634f;
    f32x4_t pen_tail;
#pragma unroll
    for (int r = 0; r < 4; ++r) pen_tail[r] = ((NT - 1) * 16 + 4 * g + r >= T) ? -INFINITY : 0.f;
    for (int qb = wave; qb < NQ; qb += 4) {
        const int qr = qb * 16 + r16;
        const bf16x8_t bq0 = nq0, bq1 = nq1, bd0 = nd0, bd1 = nd1;
        fetch(qb + 4 < NQ ? qb + 4 : qb);          // unconditional: the waits stay counted (past the end: this block again)

        // dP^T = V dO^T is multiplied TWICE (once for delta = rowsum(P * dP), once for dS) instead of being kept: 72 fewer
        // live registers per lane, which is what lets two workgroups share a CU; the matrix pipe has the time.
        auto dp_tile = [&](int t) __attribute__((always_inline)) {
            const char* vr = ldsV + (t * 16 + r16) * 128;
            const f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(vr + vsw0), bd0, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(vr + vsw1), bd1, d0, 0, 0, 0);
        };
        f32x4_t s[MAXT];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (t < NT) {
                f32x4_t c0 = (t == NT - 1) ? pen_tail : f32x4_t{0.f, 0.f, 0.f, 0.f};
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsK, t * 16 + r16, 0, g), bq0, c0, 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsK, t * 16 + r16, 1, g), bq1, c0, 0, 0, 0);
                mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale_log2;
        float lsum = 0.f, dl = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < NT) {
                const f32x4_t d = dp_tile(t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], scale_log2, -mxs));
                    s[t][r] = p;
                    lsum += p;
                    dl = fmaf(p, d[r], dl);
                }
            }
        }
        lsum += __shfl_xor(lsum, 16, 64); lsum += __shfl_xor(lsum, 32, 64);
        dl += __shfl_xor(dl, 16, 64); dl += __shfl_xor(dl, 32, 64);
        const float inv = 1.0f / lsum;
        const float delta = dl * inv;
        // dS^T = P (dP - delta) / 8, packed for the dQ product
        f32x4_t o[4];
#pragma unroll
        for (int md = 0; md < 4; ++md) o[md] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const float k8 = 0.125f * inv;
#pragma unroll
        for (int u = 0; u < (MAXT + 1) / 2; ++u) {
            const int t0 = 2 * u, t1 = 2 * u + 1;
            if (t0 < NT) {
                f32x4_t e0, e1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const f32x4_t d0 = dp_tile(t0);
#pragma unroll
                for (int r = 0; r < 4; ++r) e0[r] = s[t0][r] * k8 * (d0[r] - delta);
                if (t1 < MAXT && t1 < NT) {
                    const f32x4_t d1 = dp_tile(t1 < MAXT ? t1 : 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) e1[r] = s[t1 < MAXT ? t1 : 0][r] * k8 * (d1[r] - delta);
                }
                u32x4_t pk;
                pk[0] = pack_bf16x2(e0[0], e0[1]); pk[1] = pack_bf16x2(e0[2], e0[3]);
                pk[2] = pack_bf16x2(e1[0], e1[1]); pk[3] = pack_bf16x2(e1[2], e1[3]);
                const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pk);
#pragma unroll
                for (int md = 0; md < 4; ++md)      // a tile without a partner multiplies its own rows by zeros
                    o[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(ldsK, tr_off, md, t0, t1 < NT ? t1 : t0), pb, o[md], 0, 0, 0);
            }
        }
        asm volatile("" : "+v"(nq0), "+v"(nq1), "+v"(nd0), "+v"(nd1));      // the next block's fragments are waited for here, not behind the stores
        u32x4_t ow[2];
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {           // 16-byte stores: v_permlane16_swap between the tiles md, md + 1 (attention.hip)
            const auto r0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(o[2 * mp][0], o[2 * mp][1]), pack_bf16x2(o[2 * mp + 1][0], o[2 * mp + 1][1]), false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(o[2 * mp][2], o[2 * mp][3]), pack_bf16x2(o[2 * mp + 1][2], o[2 * mp + 1][3]), false, false);
            ow[mp][0] = r0[0]; ow[mp][1] = r1[0]; ow[mp][2] = r0[1]; ow[mp][3] = r1[1];
        }
        if (qr < T) {
            uint16_t* op = dqkv + (row0 + qr) * ld + h * ATT_DH + (g & 1) * 16 + (g >> 1) * 8;
            *(u32x4_t*)op = ow[0];
            *(u32x4_t*)(op + 32) = ow[1];
            if (g == 0) *(f32x4_t*)(stats + ((row0 + qr) * heads + h) * 4) = f32x4_t{mxs, inv, delta, 0.f};
        }
    }
}

// ---------------------------------------------------------------------------
// pass B: dK, dV (lane = key; loops over all query tiles)
// ---------------------------------------------------------------------------
// NW waves per workgroup (one workgroup per CU: its two images need 87 KB); XNP > 0: the number of query-tile pairs is a
// compile-time constant (9 at 257 tokens) and the pair loop is straight-line code, so the products of one pair overlap the
// softmax arithmetic of its neighbours -- with one wave per SIMD nothing else hides them.
template <int MAXT, int NW, int XNP>
__global__ __launch_bounds__(NW * 64, NW / 4) void attention_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dao,
                                                                   uint16_t* __restrict__ dqkv, const float* __restrict__ stats,
                                                                   int T, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int width = heads * ATT_DH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous item order, as the forward kernel: the 16 heads of a sequence -- adjacent 128-byte pieces of every
    // packed row -- run on neighbouring CUs of one XCD at the same time (attention.hip)
    const int item = xcd_contiguous(blockIdx.x, gridDim.x);
    const int seq = item / heads, h = item - seq * heads;
    const int64_t row0 = (int64_t)seq * T, ld = 3 * (int64_t)width;
    const int NT = (T + 15) >> 4, NP = XNP > 0 ? XNP : (NT + 1) >> 1;
    char* ldsQ = smem;                                   // [NP*32][160] queries: row reads (S) + transposed reads (dK)
    char* ldsO = smem + NP * 32 * ATT_ROW;               // [NP*32][160] dO: row reads (dP) + transposed reads (dV)
    float* ldsS = (float*)(smem + 2 * NP * 32 * ATT_ROW);    // [NP*32][4] statistics per query
    const int g = lane >> 4, r16 = lane & 15;
    // this lane's key: its K / V fragments come straight from HBM, fetched one key block ahead (the first before the fill)
    bf16x8_t nk0, nk1, nv0, nv1;
    auto fetch = [&](int kb) __attribute__((always_inline)) {
        int krow = kb * 16 + r16;
        krow = krow < T ? krow : T - 1;
        const uint16_t* kp = qkv + (row0 + krow) * ld + width + h * ATT_DH + 8 * g;
        const uint16_t* vp = kp + width;
        nk0 = __builtin_nontemporal_load((const bf16x8_t*)kp); nk1 = __builtin_nontemporal_load((const bf16x8_t*)(kp + 32));
        nv0 = __builtin_nontemporal_load((const bf16x8_t*)vp); nv1 = __builtin_nontemporal_load((const bf16x8_t*)(vp + 32));
    };
    fetch(wave < NT ? wave : NT - 1);
    fill_two_images<((MAXT + 1) / 2) * 32, false, NW * 64>(ldsQ, qkv, ld, h * ATT_DH, NP * 32, ldsO, dao, width, h * ATT_DH, NP * 32, row0, T, tid);
    for (int q = tid; q < NP * 32; q += NW * 64) {
        const f32x4_t st = *(const f32x4_t*)(stats + ((row0 + (q < T ? q : T - 1)) * heads + h) * 4);
        *(f32x4_t*)(ldsS + q * 4) = (q < T) ? st : f32x4_t{0.f, 0.f, 0.f, 0.f};         // inv = 0: padded queries give P = 0
    }
    __syncthreads();

    const int tr_off = (4 * g + (r16 >> 2)) * ATT_ROW + ((r16 & 3) << 3);
    const float scale_log2 = 0.125f * 1.4426950408889634f;
    for (int kb = wave; kb < NT; kb += NW) {
        const int kc = kb * 16 + r16;                    // this lane's key
        const bool kvalid = kc < T;
        const bf16x8_t bk0 = nk0, bk1 = nk1, bv0 = nv0, bv1 = nv1;
        fetch(kb + NW < NT ? kb + NW : kb);              // unconditional: counted waits (past the end: this block again)
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int md = 0; md < 4; ++md) { dk[md] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[md] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        auto pair = [&](int u) __attribute__((always_inline)) {
            f32x4_t pv[2], dsv[2];
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
                const int tq = 2 * u + w2;               // query tile; rows beyond T are zero images with inv = 0
                f32x4_t sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsQ, tq * 16 + r16, 0, g), bk0,
                                                                   f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsQ, tq * 16 + r16, 1, g), bk1, sc, 0, 0, 0);
                f32x4_t dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsO, tq * 16 + r16, 0, g), bv0,
                                                                   f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsO, tq * 16 + r16, 1, g), bv1, dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4_t st = *(const f32x4_t*)(ldsS + (tq * 16 + 4 * g + r) * 4);      // {m2, inv, delta}
                    float p = __builtin_amdgcn_exp2f(fmaf(sc[r], scale_log2, -st[0])) * st[1];
                    p = kvalid ? p : 0.f;
                    pv[w2][r] = p;
                    dsv[w2][r] = p * 0.125f * (dp[r] - st[2]);
                }
            }
            u32x4_t pk, sk;
            pk[0] = pack_bf16x2(pv[0][0], pv[0][1]); pk[1] = pack_bf16x2(pv[0][2], pv[0][3]);
            pk[2] = pack_bf16x2(pv[1][0], pv[1][1]); pk[3] = pack_bf16x2(pv[1][2], pv[1][3]);
            sk[0] = pack_bf16x2(dsv[0][0], dsv[0][1]); sk[1] = pack_bf16x2(dsv[0][2], dsv[0][3]);
            sk[2] = pack_bf16x2(dsv[1][0], dsv[1][1]); sk[3] = pack_bf16x2(dsv[1][2], dsv[1][3]);
            const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pk), sb = __builtin_bit_cast(bf16x8_t, sk);
#pragma unroll
            for (int md = 0; md < 4; ++md) {
                dv[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(ldsO, tr_off, md, 2 * u, 2 * u + 1), pb, dv[md], 0, 0, 0);
                dk[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(ldsQ, tr_off, md, 2 * u, 2 * u + 1), sb, dk[md], 0, 0, 0);
            }
        };
        if (XNP > 0) {
#pragma unroll
            for (int u = 0; u < XNP; ++u) pair(u);
        } else {
#pragma unroll 1
            for (int u = 0; u < NP; ++u) pair(u);
        }
        asm volatile("" : "+v"(nk0), "+v"(nk1), "+v"(nv0), "+v"(nv1));      // the next block's fragments are waited for here, not behind the stores
        u32x4_t wk[2], wv[2];
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {           // 16-byte stores: v_permlane16_swap between the tiles md, md + 1 (attention.hip)
            auto r0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(dk[2 * mp][0], dk[2 * mp][1]), pack_bf16x2(dk[2 * mp + 1][0], dk[2 * mp + 1][1]), false, false);
            auto r1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(dk[2 * mp][2], dk[2 * mp][3]), pack_bf16x2(dk[2 * mp + 1][2], dk[2 * mp + 1][3]), false, false);
            wk[mp][0] = r0[0]; wk[mp][1] = r1[0]; wk[mp][2] = r0[1]; wk[mp][3] = r1[1];
            r0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(dv[2 * mp][0], dv[2 * mp][1]), pack_bf16x2(dv[2 * mp + 1][0], dv[2 * mp + 1][1]), false, false);
            r1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(dv[2 * mp][2], dv[2 * mp][3]), pack_bf16x2(dv[2 * mp + 1][2], dv[2 * mp + 1][3]), false, false);
            wv[mp][0] = r0[0]; wv[mp][1] = r1[0]; wv[mp][2] = r0[1]; wv[mp][3] = r1[1];
        }
        if (kvalid) {
            uint16_t* ok = dqkv + (row0 + kc) * ld + width + h * ATT_DH + (g & 1) * 16 + (g >> 1) * 8;
            uint16_t* ov = ok + width;
            *(u32x4_t*)ok = wk[0]; *(u32x4_t*)(ok + 32) = wk[1];
            *(u32x4_t*)ov = wv[0]; *(u32x4_t*)(ov + 32) = wv[1];
        }
    }
}

// qkv bf16 [n_seq*T, 3*width], dao bf16 [n_seq*T, width] (gradient w.r.t. the attention output) ->
// dqkv bf16 [n_seq*T, 3*width]; stats_ws fp32 [n_seq*T, heads, 4] scratch.
hipError_t launch_attention_bwd(const uint16_t* qkv, const uint16_t* dao, uint16_t* dqkv, float* stats_ws, int n_seq, int T,
                                int heads, hipStream_t stream) {
    if (n_seq <= 0) return hipSuccess;
    if (T < 1 || T > 288 || heads < 1) return hipErrorInvalidValue;
    const int NT = (T + 15) / 16, NP = (NT + 1) / 2;
    const size_t lds_a = (size_t)NT * 16 * ATT_ROW + (size_t)NT * 16 * 128;
    const size_t lds_b = (size_t)2 * NP * 32 * ATT_ROW + (size_t)NP * 32 * 16;
    static std::once_flag once;
    static hipError_t st = hipSuccess;
    std::call_once(once, [] {
        st = hipFuncSetAttribute((const void*)attention_bwd_dq_kernel<18>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (st == hipSuccess)
            st = hipFuncSetAttribute((const void*)attention_bwd_dq_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (st == hipSuccess)
            st = hipFuncSetAttribute((const void*)attention_bwd_dkv_kernel<18, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (st == hipSuccess)
            st = hipFuncSetAttribute((const void*)attention_bwd_dkv_kernel<18, DKV_NW, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    if (st != hipSuccess) return st;
    const dim3 grid(n_seq * heads), block(256);
    if (NT <= 4)
        hipLaunchKernelGGL(attention_bwd_dq_kernel<4>, grid, block, lds_a, stream, qkv, dao, dqkv, stats_ws, T, heads);
    else
        hipLaunchKernelGGL(attention_bwd_dq_kernel<18>, grid, block, lds_a, stream, qkv, dao, dqkv, stats_ws, T, heads);
    if (NP == 9)        // 257 tokens (ViT-L/14): straight-line pair loop
        hipLaunchKernelGGL((attention_bwd_dkv_kernel<18, DKV_NW, 9>), grid, dim3(DKV_NW * 64), lds_b, stream, qkv, dao, dqkv, stats_ws, T, heads);
    else
        hipLaunchKernelGGL((attention_bwd_dkv_kernel<18, 4, 0>), grid, block, lds_b, stream, qkv, dao, dqkv, stats_ws, T, heads);
    return hipGetLastError();
}
