// Row kernels and attention of the split-bf16 tower mode (TVC_OPT_TOWER_PRECISION = 2; include/tvc.h): every fp32
// activation x travels to the matrix cores as TWO bf16 planes, hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo to
// ~2^-17 relative), and every product a * b is formed as a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_16x16x32_bf16
// with fp32 accumulation (the a_lo b_lo term is ~2^-18 of the result and is dropped): fp32-grade results at a third of
// the bf16 matrix rate instead of the 1/16 of the exact-f32 instruction (precise.hip).
//
// Plane layout of a GEMM operand: [rows, 2 * K] bf16, hi plane in columns 0 .. K-1, lo plane in K .. 2K-1 -- the
// GEMM's "planes" mechanism (gemm_core.hpp) addresses them as column offsets, so the tower GEMMs run on the same
// ring kernel as the bf16 mode with planes = 3.
#include "common.hpp"
#include "kernels.hpp"
#include <mutex>

namespace {

__device__ __forceinline__ void split2(float x, float& hi, float& lo) {
    hi = bf16_bits_to_f32(f32_to_bf16_bits(x));
    lo = x - hi;                                 // exact in fp32; rounded to bf16 when packed
}

// 8 consecutive fp32 -> one 16-byte hi piece + one 16-byte lo piece
__device__ __forceinline__ void split8(const f32x4_t a, const f32x4_t b, u32x4_t& hi, u32x4_t& lo) {
    float h[8], l[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { split2(a[e], h[e], l[e]); split2(b[e], h[4 + e], l[4 + e]); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = pack_bf16x2(h[2 * e], h[2 * e + 1]); lo[e] = pack_bf16x2(l[2 * e], l[2 * e + 1]); }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm of the fp32 residual stream (+ up to two fp32 deltas: the previous store-only projections, folded in here
// as the bf16 tower's layernorm_kernel folds its bf16 ones) -> hi | lo planes [rows, 2d] and / or fp32 rows.
// One wave per row, d % 4 == 0, d <= 1024.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_split_kernel(float* __restrict__ x, int64_t x_row_stride, const int32_t* __restrict__ row_idx,
                                                       const float* __restrict__ d1, const float* __restrict__ d2, int write_x,
                                                       const float* __restrict__ g, const float* __restrict__ b,
                                                       uint16_t* __restrict__ planes, float* __restrict__ y32, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t src = row_idx ? (int64_t)row_idx[row] : (int64_t)row;
    f32x4_t* xr = (f32x4_t*)(x + src * x_row_stride);
    const f32x4_t* p1 = d1 ? (const f32x4_t*)(d1 + src * x_row_stride) : nullptr;
    const f32x4_t* p2 = d2 ? (const f32x4_t*)(d2 + src * x_row_stride) : nullptr;
    const int nv = d >> 2;
    f32x4_t v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[i] = xr[c];
            if (p1) v[i] += p1[c];
            if (p2) v[i] += p2[c];
            if (write_x && (p1 || p2)) xr[c] = v[i];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { const float dl = v[i][t] - mean; q += dl * dl; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4_t gg = ((const f32x4_t*)g)[c], bb = ((const f32x4_t*)b)[c];
            f32x4_t o;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = (v[i][t] - mean) * rstd * gg[t] + bb[t];
            if (y32) ((f32x4_t*)(y32 + (int64_t)row * d))[c] = o;
            if (planes) {
                float h[4], l[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) split2(o[t], h[t], l[t]);
                uint16_t* pr = planes + (int64_t)row * 2 * d + c * 4;
                *(u32x2_t*)pr = u32x2_t{pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
                *(u32x2_t*)(pr + d) = u32x2_t{pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
            }
        }
    }
}

// fp32 rows [rows, ld_in] (first K columns used) -> planes [rows, 2 * Kp], zero padded from K to Kp; gelu 1 applies
// QuickGELU x * sigmoid(1.702 x) first (exact expf and division, as the fp32 CPU path), gelu 2 the erf GELU.
// K % 4 == 0, Kp % 4 == 0.
__global__ __launch_bounds__(256) void rows_split_kernel(const float* __restrict__ x, int64_t ld_in, uint16_t* __restrict__ out,
                                                         int64_t rows, int K, int Kp, int gelu) {
    const int nv = Kp >> 2;
    const int64_t total = rows * nv;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / nv;
        const int c = (int)(t - r * nv);
        f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (c * 4 < K) v = __builtin_nontemporal_load((const f32x4_t*)(x + r * ld_in) + c);
        float h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = v[e];
            if (gelu == 1) u = u / (1.0f + expf(-1.702f * u));
            else if (gelu == 2) u = 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
            split2(u, h[e], l[e]);
        }
        uint16_t* o = out + r * (int64_t)(2 * Kp) + c * 4;
        *(u32x2_t*)o = u32x2_t{pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
        *(u32x2_t*)(o + Kp) = u32x2_t{pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention on split operands: qkv fp32 [rows, 3 * width] (q | k | v) -> out planes [rows, 2 * width].
// Structure of attention.hip's kernel (one workgroup of 4 waves per (sequence, head); K and V of the head in LDS,
// "swapped" products S^T = K Q^T and O^T = V^T P^T so that softmax statistics are per-lane scalars; exact single-pass
// softmax), with K, V, Q and the probabilities each held as hi | lo bf16 and three MFMAs per product:
//   S = K_hi Q_lo + K_lo Q_hi + K_hi Q_hi,   O = V_hi P_lo + V_lo P_hi + V_hi P_hi   (small terms first)
// The probabilities are exp2 of fp32 scores, their row sum is taken in fp32 over the un-rounded values.
// Ragged / prefix-sharing sequences as attention.hip (starts, pfx).  LDS: 2 x KT x 128 B + 2 x KT x 160 B
// (T = 257: 156 672 B, one workgroup per CU).
// ---------------------------------------------------------------------------------------------------------------
#define SA_KROW 128
#define SA_VROW 160

template <int MAXT, bool CAUSAL>
__global__ __launch_bounds__(256) void attention_split_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out,
                                                              const int32_t* __restrict__ starts, int T_fixed, int heads,
                                                              int n_items, int kt_alloc, const int32_t* __restrict__ pfx, int n_seq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int width = heads * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = xcd_contiguous(blockIdx.x, gridDim.x);
    if (item >= n_items) return;
    const int seq = item / heads, h = item - seq * heads;
    int64_t row0, prow0 = 0;
    int T, P = 0;
    if (starts) {
        const int s0 = __builtin_amdgcn_readfirstlane(starts[seq]);
        const int s1 = __builtin_amdgcn_readfirstlane(starts[seq + 1]);
        row0 = s0; T = s1 - s0;
        if (CAUSAL && pfx) {
            P = __builtin_amdgcn_readfirstlane(pfx[seq]);
            prow0 = __builtin_amdgcn_readfirstlane(pfx[n_seq + seq]);
            T += P;
        }
    } else { row0 = (int64_t)seq * T_fixed; T = T_fixed; }
    auto key_row = [&](int t) -> int64_t { return t < P ? prow0 + t : row0 + (t - P); };
    if (T > MAXT * 16) T = MAXT * 16;
    const int NT = (T + 15) >> 4, KT = NT * 16;
    char* ldsKh = smem;
    char* ldsKl = ldsKh + kt_alloc * SA_KROW;
    char* ldsVh = ldsKl + kt_alloc * SA_KROW;
    char* ldsVl = ldsVh + kt_alloc * SA_VROW;
    const int64_t ld = 3 * (int64_t)width;
    const int g = lane >> 4, r16 = lane & 15;
    const int sw0 = ((0 + g) ^ ((lane >> 1) & 7)) << 4;
    const int sw1 = ((4 + g) ^ ((lane >> 1) & 7)) << 4;
    const int tr_off = (4 * g + (r16 >> 2)) * SA_VROW + ((r16 & 3) << 3);
    const float scale_log2 = 0.125f * 1.4426950408889634f;
    const int own = T - P;
    const int NQ = (own + 15) >> 4;

    // ---- fill: 8 fp32 (two 16-byte loads) -> one hi piece + one lo piece; zero rows beyond T
    for (int idx = tid; idx < KT * 8; idx += 256) {
        const int key = idx >> 3, c = idx & 7;
        const float* src = qkv + key_row(key < T ? key : T - 1) * ld + h * 64 + c * 8;
        const f32x4_t k0 = *(const f32x4_t*)(src + width), k1 = *(const f32x4_t*)(src + width + 4);
        const f32x4_t v0 = *(const f32x4_t*)(src + 2 * width), v1 = *(const f32x4_t*)(src + 2 * width + 4);
        u32x4_t kh, kl, vh, vl;
        split8(k0, k1, kh, kl);
        split8(v0, v1, vh, vl);
        const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
        const bool ok = key < T;
        const int ko = key * SA_KROW + ((c ^ ((key >> 1) & 7)) << 4), vo = key * SA_VROW + (c << 4);
        *(u32x4_t*)(ldsKh + ko) = ok ? kh : z;
        *(u32x4_t*)(ldsKl + ko) = ok ? kl : z;
        *(u32x4_t*)(ldsVh + vo) = ok ? vh : z;
        *(u32x4_t*)(ldsVl + vo) = ok ? vl : z;
    }
    __syncthreads();

    f32x4_t pen_tail;
#pragma unroll
    for (int r = 0; r < 4; ++r) pen_tail[r] = ((NT - 1) * 16 + 4 * g + r >= T) ? -INFINITY : 0.f;

    for (int qb = wave; qb < NQ; qb += 4) {
        const int qr = qb * 16 + r16;
        const int qrow = qr < own ? qr : own - 1;
        const float* qp = qkv + (row0 + qrow) * ld + h * 64 + 8 * g;
        u32x4_t q0h, q0l, q1h, q1l;
        split8(*(const f32x4_t*)qp, *(const f32x4_t*)(qp + 4), q0h, q0l);
        split8(*(const f32x4_t*)(qp + 32), *(const f32x4_t*)(qp + 36), q1h, q1l);
        const bf16x8_t bq0h = __builtin_bit_cast(bf16x8_t, q0h), bq0l = __builtin_bit_cast(bf16x8_t, q0l);
        const bf16x8_t bq1h = __builtin_bit_cast(bf16x8_t, q1h), bq1l = __builtin_bit_cast(bf16x8_t, q1l);
        const int qmin = P + qb * 16, qmax = qmin + 15, qpos = P + qr;
        const int nt_c = (qmax >> 4) + 1;
        const int nt_q = CAUSAL ? (nt_c < NT ? nt_c : NT) : NT;

        f32x4_t s[MAXT];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (t < nt_q) {
                f32x4_t c0 = (t == NT - 1) ? pen_tail : f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (CAUSAL && t * 16 + 15 > qmin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * 16 + 4 * g + r > qpos) c0[r] = -INFINITY;
                }
                const int ro = (t * 16 + r16) * SA_KROW;
                const bf16x8_t a0h = *(const bf16x8_t*)(ldsKh + ro + sw0), a1h = *(const bf16x8_t*)(ldsKh + ro + sw1);
                const bf16x8_t a0l = *(const bf16x8_t*)(ldsKl + ro + sw0), a1l = *(const bf16x8_t*)(ldsKl + ro + sw1);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bq0l, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bq1l, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, bq0h, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, bq1h, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bq0h, c0, 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bq1h, c0, 0, 0, 0);
                mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale_log2;
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < nt_q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t][r] = exp2f(fmaf(s[t][r], scale_log2, -mxs));
                    lsum += s[t][r];
                }
            }
        }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);

        f32x4_t o[4];
#pragma unroll
        for (int md = 0; md < 4; ++md) o[md] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < (MAXT + 1) / 2; ++u) {
            const int t0 = 2 * u, t1 = 2 * u + 1;
            if (t0 < nt_q) {
                // k-slots j < 4: keys 16 t0 + 4g + j; j >= 4: keys 16 t1 + 4g + (j - 4).  An unpaired last tile multiplies
                // its own V rows by zero probabilities (no zero partner rows in LDS).
                const bool pair = t1 < nt_q;
                const f32x4_t p0 = s[t0];
                const f32x4_t p1 = (t1 < MAXT && pair) ? s[t1 < MAXT ? t1 : 0] : f32x4_t{0.f, 0.f, 0.f, 0.f};
                float ph[8], pl[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { split2(p0[e], ph[e], pl[e]); split2(p1[e], ph[4 + e], pl[4 + e]); }
                u32x4_t pkh, pkl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pkh[e] = pack_bf16x2(ph[2 * e], ph[2 * e + 1]); pkl[e] = pack_bf16x2(pl[2 * e], pl[2 * e + 1]); }
                const bf16x8_t pbh = __builtin_bit_cast(bf16x8_t, pkh), pbl = __builtin_bit_cast(bf16x8_t, pkl);
                const int tt1 = pair ? t1 : t0;
#pragma unroll
                for (int md = 0; md < 4; ++md) {
                    const int o0 = tr_off + md * 32 + t0 * 16 * SA_VROW, o1 = tr_off + md * 32 + tt1 * 16 * SA_VROW;
                    const bf16x4_t h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(ldsVh + o0));
                    const bf16x4_t h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(ldsVh + o1));
                    const bf16x4_t l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(ldsVl + o0));
                    const bf16x4_t l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(ldsVl + o1));
                    bf16x8_t ah, al;
                    ah[0] = h0[0]; ah[1] = h0[1]; ah[2] = h0[2]; ah[3] = h0[3]; ah[4] = h1[0]; ah[5] = h1[1]; ah[6] = h1[2]; ah[7] = h1[3];
                    al[0] = l0[0]; al[1] = l0[1]; al[2] = l0[2]; al[3] = l0[3]; al[4] = l1[0]; al[5] = l1[1]; al[6] = l1[2]; al[7] = l1[3];
                    o[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pbl, o[md], 0, 0, 0);
                    o[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, pbh, o[md], 0, 0, 0);
                    o[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pbh, o[md], 0, 0, 0);
                }
            }
        }
        const float inv = 1.0f / lsum;
        if (qr < own) {
            uint16_t* op = out + (row0 + qr) * (int64_t)(2 * width) + h * 64 + 4 * g;
#pragma unroll
            for (int md = 0; md < 4; ++md) {
                float hh[4], ll[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(o[md][e] * inv, hh[e], ll[e]);
                *(u32x2_t*)(op + md * 16) = u32x2_t{pack_bf16x2(hh[0], hh[1]), pack_bf16x2(hh[2], hh[3])};
                *(u32x2_t*)(op + md * 16 + width) = u32x2_t{pack_bf16x2(ll[0], ll[1]), pack_bf16x2(ll[2], ll[3])};
            }
        }
    }
}

template <int MAXT, bool CAUSAL>
hipError_t launch_split_one(const float* qkv, uint16_t* out, const int32_t* starts, int n_seq, int T, int heads,
                            hipStream_t stream, const int32_t* pfx) {
    const int kt = (T + 15) / 16 * 16;
    const size_t lds = (size_t)kt * (2 * SA_KROW + 2 * SA_VROW);
    static std::once_flag once;
    static hipError_t attr_st = hipSuccess;
    std::call_once(once, [] {
        attr_st = hipFuncSetAttribute((const void*)attention_split_kernel<MAXT, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
    });
    if (attr_st != hipSuccess) return attr_st;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int n_items = n_seq * heads;
    hipLaunchKernelGGL((attention_split_kernel<MAXT, CAUSAL>), dim3(n_items), dim3(256), lds, stream, qkv, out, starts, T, heads,
                       n_items, kt, pfx, n_seq);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_ln_split(float* x, int64_t x_row_stride, const int32_t* row_idx, const float* d1, const float* d2, int write_x,
                           const float* g, const float* b, uint16_t* planes, float* y32, int rows, int d, hipStream_t stream) {
    if (d % 4 != 0 || d > 1024 || rows < 0 || (!planes && !y32)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(ln_split_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, x_row_stride, row_idx, d1, d2, write_x, g, b,
                       planes, y32, rows, d);
    return hipGetLastError();
}

hipError_t launch_rows_split(const float* x, int64_t ld_in, uint16_t* out, int64_t rows, int K, int Kp, int gelu, hipStream_t stream) {
    if (K % 4 != 0 || Kp % 4 != 0 || Kp < K || rows < 0) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const int64_t total = rows * (Kp >> 2);
    int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    hipLaunchKernelGGL(rows_split_kernel, dim3(grid), dim3(256), 0, stream, x, ld_in, out, rows, K, Kp, gelu);
    return hipGetLastError();
}

// starts == nullptr: n_seq sequences of seq_len rows; else packed rows (+ pfx: shared prefixes), seq_len = the maximum
hipError_t launch_attention_split(const float* qkv, uint16_t* out, const int32_t* starts, int n_seq, int seq_len, int heads,
                                  int causal, hipStream_t stream, const int32_t* pfx) {
    if (n_seq <= 0) return hipSuccess;
    // K / V of a head as hi | lo images: 576 bytes per key row, 160 KB of LDS -> 272 keys (17 tiles: the 257 tokens of ViT-L/14)
    if (seq_len < 1 || seq_len > 272 || heads < 1 || (pfx && (!starts || !causal))) return hipErrorInvalidValue;
    const int NT = (seq_len + 15) / 16;
    if (causal) {
        if (NT <= 2) return launch_split_one<2, true>(qkv, out, starts, n_seq, seq_len, heads, stream, pfx);
        if (NT <= 6) return launch_split_one<6, true>(qkv, out, starts, n_seq, seq_len, heads, stream, pfx);
        return launch_split_one<18, true>(qkv, out, starts, n_seq, seq_len, heads, stream, pfx);
    }
    if (NT <= 2) return launch_split_one<2, false>(qkv, out, starts, n_seq, seq_len, heads, stream, nullptr);
    if (NT <= 6) return launch_split_one<6, false>(qkv, out, starts, n_seq, seq_len, heads, stream, nullptr);
    return launch_split_one<18, false>(qkv, out, starts, n_seq, seq_len, heads, stream, nullptr);
}
