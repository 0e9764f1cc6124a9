// Dense bf16 GEMM with fused epilogues for the CLIP towers (K1/K2) and the
// bank-search pre-pass.  See gemm_core.hpp for the tiling.
#include "gemm_core.hpp"
#include "kernels.hpp"

struct GemmEpilogue {
    const float* bias;     // [I] or nullptr
    void* out;             // [J, ldo]
    int64_t ldo;
};

__device__ __forceinline__ float quick_gelu(float x) {
    // x * sigmoid(1.702 x)
    return x / (1.0f + __expf(-1.702f * x));
}

// out[j, i..i+3] for one lane: i = 4 consecutive out-features.  The vector
// path needs all four in range and a 4-element-aligned leading dimension;
// ragged edges (bank samples, cosine matrices) take the scalar path.
template <int EPI>
__device__ __forceinline__ void gemm_store4(const GemmEpilogue& e, int I, int i, int j, f32x4_t v) {
    const bool vec = (i + 3 < I) && ((e.ldo & 3) == 0);
    if (vec) {
        if (e.bias) v += *(const f32x4_t*)(e.bias + i);
        if (EPI == TVC_EPI_F32) {
            *(f32x4_t*)((float*)e.out + (int64_t)j * e.ldo + i) = v;
        } else if (EPI == TVC_EPI_RESID_F32) {
            float* p = (float*)e.out + (int64_t)j * e.ldo + i;
            const f32x4_t r = *(const f32x4_t*)p;
            *(f32x4_t*)p = r + v;
        } else {
            if (EPI == TVC_EPI_GELU_BF16) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = quick_gelu(v[t]);
            }
            u32x2_t o;
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
            *(u32x2_t*)((uint16_t*)e.out + (int64_t)j * e.ldo + i) = o;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (i + t >= I) break;
        float x = v[t] + (e.bias ? e.bias[i + t] : 0.f);
        const int64_t off = (int64_t)j * e.ldo + i + t;
        if (EPI == TVC_EPI_F32) ((float*)e.out)[off] = x;
        else if (EPI == TVC_EPI_RESID_F32) ((float*)e.out)[off] += x;
        else {
            if (EPI == TVC_EPI_GELU_BF16) x = quick_gelu(x);
            ((uint16_t*)e.out)[off] = f32_to_bf16_bits(x);
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_bf16_kernel(GemmOperands g, GemmEpilogue e,
                                                                  int nIt, int nJt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lin = xcd_contiguous(blockIdx.x, nIt * nJt);
    const int jt = lin / nIt, it = lin - jt * nIt;   // out-feature tile fastest: an XCD
    const int i0 = it * GEMM_BM, j0 = jt * GEMM_BN;  // re-uses one token panel from L2

    gemm_acc_t acc;
    gemm_zero_acc(acc);
    gemm_mainloop(acc, g, i0, j0, smem);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int j = j0 + wn * 64 + n * 16 + (lane & 15);
        if (j >= g.J) continue;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int i = i0 + wm * 128 + m * 16 + (lane >> 4) * 4;
            if (i < g.I) gemm_store4<EPI>(e, g.I, i, j, acc[m][n]);
        }
    }
}

static hipError_t set_lds_attr_once() {
    static bool done = false;
    static hipError_t st = hipSuccess;
    if (done) return st;
    done = true;
#define SET_ATTR(K)                                                                              \
    if (st == hipSuccess)                                                                        \
        st = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                 GEMM_LDS_BYTES);
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_GELU_BF16>)
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_RESID_F32>)
#undef SET_ATTR
    return st;
}

hipError_t launch_gemm_bf16(const GemmLaunch& L, hipStream_t stream) {
    hipError_t st = set_lds_attr_once();
    if (st != hipSuccess) return st;
    GemmOperands g;
    g.A = L.A; g.B = L.B; g.lda = L.lda; g.ldb = L.ldb; g.I = L.I; g.J = L.J;
    g.ksteps_per_plane = L.K / GEMM_BK;
    g.planes = L.planes;
    for (int p = 0; p < 4; ++p) { g.a_plane_off[p] = L.a_plane_off[p]; g.b_plane_off[p] = L.b_plane_off[p]; }
    GemmEpilogue e;
    e.bias = L.bias; e.out = L.out; e.ldo = L.ldo;
    const int nIt = (L.I + GEMM_BM - 1) / GEMM_BM, nJt = (L.J + GEMM_BN - 1) / GEMM_BN;
    const dim3 grid(nIt * nJt), block(GEMM_THREADS);
    switch (L.epilogue) {
        case TVC_EPI_F32:
            hipLaunchKernelGGL(gemm_bf16_kernel<TVC_EPI_F32>, grid, block, GEMM_LDS_BYTES, stream, g, e, nIt, nJt);
            break;
        case TVC_EPI_BF16:
            hipLaunchKernelGGL(gemm_bf16_kernel<TVC_EPI_BF16>, grid, block, GEMM_LDS_BYTES, stream, g, e, nIt, nJt);
            break;
        case TVC_EPI_GELU_BF16:
            hipLaunchKernelGGL(gemm_bf16_kernel<TVC_EPI_GELU_BF16>, grid, block, GEMM_LDS_BYTES, stream, g, e, nIt, nJt);
            break;
        case TVC_EPI_RESID_F32:
            hipLaunchKernelGGL(gemm_bf16_kernel<TVC_EPI_RESID_F32>, grid, block, GEMM_LDS_BYTES, stream, g, e, nIt, nJt);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
