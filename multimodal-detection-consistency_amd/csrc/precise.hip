// fp32-grade tower kernels (TVC_OPT_TOWER_PRECISION = 1): the validation / attack-generation mode in which the CLIP
// towers reproduce the reference's fp32 CPU path (src/detector.py:461-485; configs/attacks/pgd.yaml:80 asks for fp32)
// to ~1e-6 per embedding component, so that BASELINE.json's "scores within 1e-4 of the CPU path" holds END TO END
// and not only on identical embeddings.  Everything stays fp32: weights as given (no bf16 rounding), activations,
// the residual stream, attention probabilities.  The GEMM runs on the exact-f32 matrix instruction
// v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain per output element, 1/16 of the bf16 MFMA rate: this mode is ~10x
// slower than the bf16 towers and is not the benchmarked path).
#include "common.hpp"
#include "kernels.hpp"
#include <mutex>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

constexpr int PG_BM = 128;      // tokens per workgroup tile
constexpr int PG_BN = 128;      // out-features per workgroup tile
constexpr int PG_BK = 32;       // k per LDS stage (128 B of a row)
constexpr int PG_LD = 36;       // LDS row stride in floats: 16-byte aligned rows, conflict-free b128 fragment reads

__device__ __forceinline__ float quick_gelu_f32(float v) { return v / (1.0f + expf(-1.702f * v)); }

// out[j, i] (op)= sum_k X[j, k] * W[i, k] + bias[i]
//   epi 0: store      1: QuickGELU, store      2: out += (residual add in place)
// MFMA rows = tokens, MFMA columns (the lane index) = out-features, so a store instruction writes two 128-byte runs.
// The k order inside a group of 8 is (s, 4 + s) for s = 0..3 -- both operands use the same map, so every product is
// formed exactly once; only the (immaterial) order of the fp32 additions differs from a sequential loop.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ W, int64_t ldw,
                                                       const float* __restrict__ X, int64_t ldx,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int64_t ldo, int I, int J, int K, int epi) {
    __shared__ __attribute__((aligned(16))) float Xs[PG_BM * PG_LD];
    __shared__ __attribute__((aligned(16))) float Ws[PG_BN * PG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.x * PG_BN;
    const int64_t j0 = (int64_t)blockIdx.y * PG_BM;
    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4_t xr[4], wr[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + t * 256;
            const int row = idx >> 3, c4 = idx & 7;
            const int k = k0 + c4 * 4;
            xr[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            wr[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                if (j0 + row < J) xr[t] = *(const f32x4_t*)(X + (j0 + row) * ldx + k);
                if (i0 + row < I) wr[t] = *(const f32x4_t*)(W + (int64_t)(i0 + row) * ldw + k);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + t * 256;
            const int row = idx >> 3, c4 = idx & 7;
            *(f32x4_t*)(Xs + row * PG_LD + c4 * 4) = xr[t];
            *(f32x4_t*)(Ws + row * PG_LD + c4 * 4) = wr[t];
        }
    };

    load_tile(0);
    for (int k0 = 0; k0 < K; k0 += PG_BK) {
        __syncthreads();            // the previous stage's fragment reads are done
        store_tile();
        __syncthreads();
        if (k0 + PG_BK < K) load_tile(k0 + PG_BK);      // in flight while this stage multiplies
        const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int k8 = 0; k8 < PG_BK / 8; ++k8) {
            f32x4_t a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *(const f32x4_t*)(Xs + (wm * 64 + t * 32 + fr) * PG_LD + k8 * 8 + fh * 4);
                b[t] = *(const f32x4_t*)(Ws + (wn * 64 + t * 32 + fr) * PG_LD + k8 * 8 + fh * 4);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
        }
    }
    // epilogue: lane owns feature (lane & 31), tokens (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int f = i0 + wn * 64 + tn * 32 + (lane & 31);
        if (f >= I) continue;
        const float bv = bias ? bias[f] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t j = j0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (j >= J) continue;
                float v = acc[tm][tn][r] + bv;
                float* o = out + j * ldo + f;
                if (epi == 1) v = quick_gelu_f32(v);
                if (epi == 2) v += *o;
                *o = v;
            }
        }
    }
}

// Exact-softmax attention in fp32, head_dim 64: one workgroup per (sequence, head); K and V of the head live in LDS
// as fp32 rows (every lane reads the same key row: broadcast reads), one query per thread, online softmax.
// qkv fp32 [n_seq * T, 3 * width] (q | k | v), out fp32 [n_seq * T, width].
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int T, int heads, int causal) {
    extern __shared__ __attribute__((aligned(16))) char att32_smem[];
    float* Ks = (float*)att32_smem;           // [T][64]
    float* Vs = Ks + (size_t)T * 64;          // [T][64]
    const int seq = blockIdx.x / heads, head = blockIdx.x - seq * heads;
    const int width = heads * 64;
    const int64_t ld = 3 * (int64_t)width;
    const float* base = qkv + (int64_t)seq * T * ld + head * 64;
    for (int i = threadIdx.x; i < T * 16; i += 256) {
        const int row = i >> 4, c4 = i & 15;
        ((f32x4_t*)Ks)[i] = *(const f32x4_t*)(base + row * ld + width + c4 * 4);
        ((f32x4_t*)Vs)[i] = *(const f32x4_t*)(base + row * ld + 2 * width + c4 * 4);
    }
    __syncthreads();
    const float scale = 0.125f;               // 64 ** -0.5
    for (int qi = threadIdx.x; qi < T; qi += 256) {
        f32x4_t q[16], o[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            q[c] = *(const f32x4_t*)(base + qi * ld + c * 4);
            o[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        float m = -INFINITY, l = 0.f;
        const int nk = causal ? qi + 1 : T;
        for (int j = 0; j < nk; ++j) {
            const f32x4_t* kr = (const f32x4_t*)(Ks + (size_t)j * 64);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4_t kv = kr[c];
                s0 = fmaf(q[c][0], kv[0], s0); s1 = fmaf(q[c][1], kv[1], s1);
                s2 = fmaf(q[c][2], kv[2], s2); s3 = fmaf(q[c][3], kv[3], s3);
            }
            const float s = ((s0 + s1) + (s2 + s3)) * scale;
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn);         // exp(-inf) = 0 on the first key
            const float p = expf(s - mn);
            l = l * alpha + p;
            m = mn;
            const f32x4_t* vr = (const f32x4_t*)(Vs + (size_t)j * 64);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4_t vv = vr[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[c][e] = fmaf(p, vv[e], o[c][e] * alpha);
            }
        }
        const float inv = 1.0f / l;
        float* orow = out + ((int64_t)seq * T + qi) * width + head * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            f32x4_t r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = o[c][e] * inv;
            *(f32x4_t*)(orow + c * 4) = r;
        }
    }
}

// pix fp32 [B, 3, S, S] -> fp32 [B * P, 3 * patch * patch], column order (c, ky, kx) = the conv weight's flatten order
__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ pix, float* __restrict__ out, int B,
                                                         int S, int patch) {
    const int g = S / patch, P = g * g, K = 3 * patch * patch;
    const int64_t total = (int64_t)B * P * K;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(t % K);
        const int64_t row = t / K;
        const int p = (int)(row % P);
        const int64_t bimg = row / P;
        const int py = p / g, px = p - py * g;
        const int c = col / (patch * patch), rem = col - c * patch * patch;
        const int ky = rem / patch, kx = rem - ky * patch;
        out[t] = pix[((bimg * 3 + c) * S + (py * patch + ky)) * S + (px * patch + kx)];
    }
}

// rows of x gathered by index: out[n, :] = x[idx[n], :]
__global__ __launch_bounds__(256) void gather_f32_rows_kernel(const float* __restrict__ x, int64_t ld,
                                                              const int32_t* __restrict__ idx, int64_t idx_mul,
                                                              float* __restrict__ out, int n, int d) {
    const int64_t total = (int64_t)n * (d >> 2);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % (d >> 2));
        const int64_t r = t / (d >> 2);
        const int64_t src = idx ? (int64_t)idx[r] : r * idx_mul;
        ((f32x4_t*)(out + r * d))[c] = ((const f32x4_t*)(x + src * ld))[c];
    }
}

}  // namespace

hipError_t launch_gemm_f32(const float* W, int64_t ldw, const float* X, int64_t ldx, const float* bias, float* out,
                           int64_t ldo, int I, int J, int K, int epi, hipStream_t stream) {
    if (I <= 0 || J <= 0) return hipSuccess;
    if (K <= 0 || K % 4 != 0 || ldw % 4 != 0 || ldx % 4 != 0 || epi < 0 || epi > 2) return hipErrorInvalidValue;
    const int64_t gy = ((int64_t)J + PG_BM - 1) / PG_BM;
    if (gy > 65535) return hipErrorInvalidValue;
    dim3 grid((I + PG_BN - 1) / PG_BN, (unsigned)gy);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, W, ldw, X, ldx, bias, out, ldo, I, J, K, epi);
    return hipGetLastError();
}

hipError_t launch_attention_f32(const float* qkv, float* out, int n_seq, int T, int heads, int causal,
                                hipStream_t stream) {
    if (n_seq <= 0) return hipSuccess;
    if (T <= 0 || T > 288 || heads <= 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)T * 64 * 4 * 2;
    static std::once_flag attr_once;          // thread-safe: two engines may first-launch from two threads
    static hipError_t attr_st = hipSuccess;
    std::call_once(attr_once, [] {
        attr_st = hipFuncSetAttribute((const void*)attention_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    });
    if (attr_st != hipSuccess) return attr_st;
    hipLaunchKernelGGL(attention_f32_kernel, dim3(n_seq * heads), dim3(256), lds, stream, qkv, out, T, heads, causal);
    return hipGetLastError();
}

hipError_t launch_im2col_f32(const float* pix, float* out, int B, int image, int patch, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const int g = image / patch;
    const int64_t total = (int64_t)B * g * g * 3 * patch * patch;
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(im2col_f32_kernel, dim3(grid), dim3(256), 0, stream, pix, out, B, image, patch);
    return hipGetLastError();
}

hipError_t launch_gather_f32_rows(const float* x, int64_t ld, const int32_t* idx, int64_t idx_mul, float* out, int n,
                                  int d, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (d % 4 != 0) return hipErrorInvalidValue;
    const int64_t total = (int64_t)n * (d >> 2);
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gather_f32_rows_kernel, dim3(grid), dim3(256), 0, stream, x, ld, idx, idx_mul, out, n, d);
    return hipGetLastError();
}
