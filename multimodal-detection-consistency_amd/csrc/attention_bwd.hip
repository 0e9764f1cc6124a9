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

namespace {

// fill a [rows_pad][160 B] image with `T` rows of 64 bf16 taken from qkv-like rows (zero beyond T)
__device__ __forceinline__ void fill_image(char* lds, const uint16_t* __restrict__ src, int64_t row0, int64_t ld, int col0,
                                           int T, int rows_pad, int tid, int nthreads) {
    for (int idx = tid; idx < rows_pad * 8; idx += nthreads) {
        const int r = idx >> 3, c = idx & 7;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (r < T) v = *(const u32x4_t*)(src + (row0 + r) * ld + col0 + c * 8);
        *(u32x4_t*)(lds + r * ATT_ROW + (c << 4)) = v;
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
__global__ __launch_bounds__(256, 1) void attention_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dao,
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
    char* ldsK = smem;                                   // [NP*32][160]: row reads (S) and transposed reads (dQ)
    char* ldsV = smem + NP * 32 * ATT_ROW;               // [NT*16][160]: row reads (dP)
    fill_image(ldsK, qkv, row0, ld, width + h * ATT_DH, T, NP * 32, tid, 256);
    fill_image(ldsV, qkv, row0, ld, 2 * width + h * ATT_DH, T, NT * 16, tid, 256);
    __syncthreads();

    const int g = lane >> 4, r16 = lane & 15;
    const int tr_off = (4 * g + (r16 >> 2)) * ATT_ROW + ((r16 & 3) << 3);
    const float scale_log2 = 0.125f * 1.4426950408889634f;
    f32x4_t pen_tail;
#pragma unroll
    for (int r = 0; r < 4; ++r) pen_tail[r] = ((NT - 1) * 16 + 4 * g + r >= T) ? -INFINITY : 0.f;
    const int NQ = NT;
    for (int qb = wave; qb < NQ; qb += 4) {
        const int qr = qb * 16 + r16;
        const int qrow = qr < T ? qr : T - 1;
        const uint16_t* qp = qkv + (row0 + qrow) * ld + h * ATT_DH + 8 * g;
        const uint16_t* dp_ = dao + (row0 + qrow) * (int64_t)width + h * ATT_DH + 8 * g;
        const bf16x8_t bq0 = *(const bf16x8_t*)qp, bq1 = *(const bf16x8_t*)(qp + 32);
        const bf16x8_t bd0 = *(const bf16x8_t*)dp_, bd1 = *(const bf16x8_t*)(dp_ + 32);

        f32x4_t s[MAXT], dpv[MAXT];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dpv[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (t < NT) {
                f32x4_t c0 = (t == NT - 1) ? pen_tail : f32x4_t{0.f, 0.f, 0.f, 0.f};
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsK, t * 16 + r16, 0, g), bq0, c0, 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsK, t * 16 + r16, 1, g), bq1, c0, 0, 0, 0);
                mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
                f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsV, t * 16 + r16, 0, g), bd0,
                                                                   f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dpv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(ldsV, t * 16 + r16, 1, g), bd1, d0, 0, 0, 0);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale_log2;
        float lsum = 0.f, dl = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], scale_log2, -mxs));
                    s[t][r] = p;
                    lsum += p;
                    dl = fmaf(p, dpv[t][r], dl);
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
#pragma unroll
                for (int r = 0; r < 4; ++r) e0[r] = s[t0][r] * k8 * (dpv[t0][r] - delta);
                if (t1 < MAXT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) e1[r] = s[t1 < MAXT ? t1 : 0][r] * k8 * (dpv[t1 < MAXT ? t1 : 0][r] - delta);
                }
                u32x4_t pk;
                pk[0] = pack_bf16x2(e0[0], e0[1]); pk[1] = pack_bf16x2(e0[2], e0[3]);
                pk[2] = pack_bf16x2(e1[0], e1[1]); pk[3] = pack_bf16x2(e1[2], e1[3]);
                const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pk);
#pragma unroll
                for (int md = 0; md < 4; ++md)
                    o[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(ldsK, tr_off, md, t0, t1), pb, o[md], 0, 0, 0);
            }
        }
        if (qr < T) {
            uint16_t* op = dqkv + (row0 + qr) * ld + h * ATT_DH + 4 * g;
#pragma unroll
            for (int md = 0; md < 4; ++md)
                *(u32x2_t*)(op + md * 16) = u32x2_t{pack_bf16x2(o[md][0], o[md][1]), pack_bf16x2(o[md][2], o[md][3])};
            if (g == 0) *(f32x4_t*)(stats + ((row0 + qr) * heads + h) * 4) = f32x4_t{mxs, inv, delta, 0.f};
        }
    }
}

// ---------------------------------------------------------------------------
// pass B: dK, dV (lane = key; loops over all query tiles)
// ---------------------------------------------------------------------------
template <int MAXT>
__global__ __launch_bounds__(256, 1) void attention_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dao,
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
    const int NT = (T + 15) >> 4, NP = (NT + 1) >> 1;
    char* ldsQ = smem;                                   // [NP*32][160] queries: row reads (S) + transposed reads (dK)
    char* ldsO = smem + NP * 32 * ATT_ROW;               // [NP*32][160] dO: row reads (dP) + transposed reads (dV)
    float* ldsS = (float*)(smem + 2 * NP * 32 * ATT_ROW);    // [NP*32][4] statistics per query
    fill_image(ldsQ, qkv, row0, ld, h * ATT_DH, T, NP * 32, tid, 256);
    fill_image(ldsO, dao, row0, width, h * ATT_DH, T, NP * 32, tid, 256);
    for (int q = tid; q < NP * 32; q += 256)
        *(f32x4_t*)(ldsS + q * 4) = (q < T) ? *(const f32x4_t*)(stats + ((row0 + q) * heads + h) * 4)
                                            : f32x4_t{0.f, 0.f, 0.f, 0.f};         // inv = 0: padded queries give P = 0
    __syncthreads();

    const int g = lane >> 4, r16 = lane & 15;
    const int tr_off = (4 * g + (r16 >> 2)) * ATT_ROW + ((r16 & 3) << 3);
    const float scale_log2 = 0.125f * 1.4426950408889634f;
    for (int kb = wave; kb < NT; kb += 4) {
        const int kc = kb * 16 + r16;                    // this lane's key
        const bool kvalid = kc < T;
        const int krow = kvalid ? kc : T - 1;
        const uint16_t* kp = qkv + (row0 + krow) * ld + width + h * ATT_DH + 8 * g;
        const uint16_t* vp = kp + width;
        const bf16x8_t bk0 = *(const bf16x8_t*)kp, bk1 = *(const bf16x8_t*)(kp + 32);
        const bf16x8_t bv0 = *(const bf16x8_t*)vp, bv1 = *(const bf16x8_t*)(vp + 32);
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int md = 0; md < 4; ++md) { dk[md] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[md] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
        for (int u = 0; u < NP; ++u) {
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
        }
        if (kvalid) {
            uint16_t* ok = dqkv + (row0 + kc) * ld + width + h * ATT_DH + 4 * g;
            uint16_t* ov = ok + width;
#pragma unroll
            for (int md = 0; md < 4; ++md) {
                *(u32x2_t*)(ok + md * 16) = u32x2_t{pack_bf16x2(dk[md][0], dk[md][1]), pack_bf16x2(dk[md][2], dk[md][3])};
                *(u32x2_t*)(ov + md * 16) = u32x2_t{pack_bf16x2(dv[md][0], dv[md][1]), pack_bf16x2(dv[md][2], dv[md][3])};
            }
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
    const size_t lds_a = (size_t)NP * 32 * ATT_ROW + (size_t)NT * 16 * ATT_ROW;
    const size_t lds_b = (size_t)2 * NP * 32 * ATT_ROW + (size_t)NP * 32 * 16;
    static std::once_flag once;
    static hipError_t st = hipSuccess;
    std::call_once(once, [] {
        st = hipFuncSetAttribute((const void*)attention_bwd_dq_kernel<18>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (st == hipSuccess)
            st = hipFuncSetAttribute((const void*)attention_bwd_dq_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (st == hipSuccess)
            st = hipFuncSetAttribute((const void*)attention_bwd_dkv_kernel<18>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    if (st != hipSuccess) return st;
    const dim3 grid(n_seq * heads), block(256);
    if (NT <= 4)
        hipLaunchKernelGGL(attention_bwd_dq_kernel<4>, grid, block, lds_a, stream, qkv, dao, dqkv, stats_ws, T, heads);
    else
        hipLaunchKernelGGL(attention_bwd_dq_kernel<18>, grid, block, lds_a, stream, qkv, dao, dqkv, stats_ws, T, heads);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel<18>, grid, block, lds_b, stream, qkv, dao, dqkv, stats_ws, T, heads);
    return hipGetLastError();
}
