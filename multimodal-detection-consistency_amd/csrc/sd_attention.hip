// Streaming (flash) attention for the latent-diffusion UNet: self-attention over T = 4096 / 1024 / 256 / 64 latent
// positions and cross-attention onto the 77 text states, 8 heads of head_dim 40 / 80 / 160 (head_dim % 8 == 0, <= 160).
//
// One workgroup of 4 waves per (64 queries, head, sample); keys / values stream through LDS in tiles of 64 with an
// online softmax.  The products are "swapped" as in attention.hip so that the query sits on the lane (column) index:
//   S^T[key, q] = K . Q^T     MFMA 16x16x32: A = K rows from LDS (ds_read_b128), B = Q fragments held in registers
//   O^T[d, q]  += V^T . P^T   A = V^T by ds_read_b64_tr_b16 (hardware transpose of the row-major V tile), B = P^T packed
//                             from the S^T accumulators with no lane movement
// so the running max / sum are per-lane scalars (+ two xor-shuffles per tile for the max) and a lane's output is 4
// consecutive head dims of its query (8-byte stores).  head_dim is zero-padded to 32 * KS for Q.K^T and covered by
// DV tiles of 16 for P.V.
#include "common.hpp"
#include "kernels.hpp"
#include <mutex>

namespace {

template <int KS, int DV>
struct FaCfg {
    static constexpr int DHP = 32 * KS;                 // padded head dim of the K tile
    static constexpr int KROW = DHP * 2 + 16;           // bytes per K row in LDS (+16: spreads the b128 reads over banks)
    static constexpr int VROW = DV * 32 + 32;           // bytes per V row in LDS
    static constexpr int KCH = DHP / 8;                 // 16-byte chunks per K row
    static constexpr int VCH = DV * 2;
    static constexpr int KBYTES = 64 * KROW;
    static constexpr int VBYTES = 64 * VROW;
    static constexpr int KIT = (64 * KCH + 255) / 256;  // staging pieces per thread
    static constexpr int VIT = (64 * VCH + 255) / 256;
};

template <int KS, int DV>
__global__ __launch_bounds__(256) void sd_flash_attention_kernel(const uint16_t* __restrict__ Q, int64_t ldq,
                                                                 const uint16_t* __restrict__ K, int64_t ldk,
                                                                 const uint16_t* __restrict__ V, int64_t ldv,
                                                                 uint16_t* __restrict__ O, int64_t ldo, int Tq, int Tk,
                                                                 int dh, float scale_log2, int heads, int nqb) {
    using C = FaCfg<KS, DV>;
    __shared__ __attribute__((aligned(16))) char smem[C::KBYTES + C::VBYTES];
    char* ldsK = smem;
    char* ldsV = smem + C::KBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    // linear grid, XCD-contiguous: the 8 XCDs take workgroups round-robin, so consecutive blockIdx.x land on different
    // XCDs; remapped, the query blocks of one (sample, head) -- which all stream the SAME keys / values -- run on one XCD
    // and find them in its L2 after the first block's pass (speed only)
    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int bh = lin / nqb, qb = lin - bh * nqb;
    const int b = bh / heads, h = bh - b * heads;
    const int q0 = qb * 64 + wave * 16;
    const int dchunks = dh >> 3;                        // valid 16-byte chunks of a head row

    // ---- Q fragments (registers, whole kernel): lane holds Q[q][32 s + 8 g .. + 7]
    int qrow = q0 + r16;
    qrow = qrow < Tq ? qrow : Tq - 1;
    const uint16_t* qp = Q + ((int64_t)b * Tq + qrow) * ldq + h * dh;
    bf16x8_t bq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 4 * s + g;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (c < dchunks) v = *(const u32x4_t*)(qp + c * 8);
        bq[s] = __builtin_bit_cast(bf16x8_t, v);
    }

    const uint16_t* kbase = K + (int64_t)b * Tk * ldk + h * dh;
    const uint16_t* vbase = V + (int64_t)b * Tk * ldv + h * dh;
    // TWO register sets: tile t + 2 is requested while tile t is multiplied, so a key / value tile has two tile times
    // (not one) to arrive from L2 -- one was measured latency-bound at head_dim 40 (a tile's arithmetic is ~0.15 us)
    u32x4_t kregA[C::KIT], vregA[C::VIT], kregB[C::KIT], vregB[C::VIT];
    auto prefetch = [&](int key0, u32x4_t (&kreg)[C::KIT], u32x4_t (&vreg)[C::VIT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::KIT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::KCH, c = idx - row * C::KCH;
            kreg[i] = u32x4_t{0u, 0u, 0u, 0u};
            if (idx < 64 * C::KCH && key0 + row < Tk && c < dchunks)
                kreg[i] = *(const u32x4_t*)(kbase + (int64_t)(key0 + row) * ldk + c * 8);
        }
#pragma unroll
        for (int i = 0; i < C::VIT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::VCH, c = idx - row * C::VCH;
            vreg[i] = u32x4_t{0u, 0u, 0u, 0u};
            if (idx < 64 * C::VCH && key0 + row < Tk && c < dchunks)
                vreg[i] = *(const u32x4_t*)(vbase + (int64_t)(key0 + row) * ldv + c * 8);
        }
    };
    auto commit = [&](const u32x4_t (&kreg)[C::KIT], const u32x4_t (&vreg)[C::VIT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::KIT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::KCH, c = idx - row * C::KCH;
            if (idx < 64 * C::KCH) *(u32x4_t*)(ldsK + row * C::KROW + c * 16) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < C::VIT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::VCH, c = idx - row * C::VCH;
            if (idx < 64 * C::VCH) *(u32x4_t*)(ldsV + row * C::VROW + c * 16) = vreg[i];
        }
    };

    // transposed-read lane address (attention.hip): lane i of a 16-lane group supplies key row 4 g + (i >> 2),
    // head dims 4 (i & 3) .. + 3 of a [16 keys][16 dims] block
    const int tr_off = (4 * g + (r16 >> 2)) * C::VROW + ((r16 & 3) << 3);

    f32x4_t o[DV];
#pragma unroll
    for (int md = 0; md < DV; ++md) o[md] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (Tk + 63) >> 6;
    prefetch(0, kregA, vregA);
    if (nkt > 1) prefetch(64, kregB, vregB);
    auto tile = [&](int kt, u32x4_t (&kreg)[C::KIT], u32x4_t (&vreg)[C::VIT]) __attribute__((always_inline)) {
        __syncthreads();                       // every wave is done reading the previous tile
        commit(kreg, vreg);
        __syncthreads();
        if (kt + 2 < nkt) prefetch((kt + 2) * 64, kreg, vreg);      // in flight behind two tiles' arithmetic
        const int key0 = kt * 64;
        // ---- S^T = K . Q^T for the 4 key sub-tiles of 16; keys >= Tk start at -inf
        f32x4_t s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] = (key0 + t * 16 + 4 * g + r >= Tk) ? -INFINITY : 0.f;
            const char* kr = ldsK + (t * 16 + r16) * C::KROW + g * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t a = *(const bf16x8_t*)(kr + ks * 64);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[ks], s[t], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);             // finite: every tile holds at least one key < Tk
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);
        const float mns = m_new * scale_log2;
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], scale_log2, -mns));
                psum += s[t][r];
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int md = 0; md < DV; ++md)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[md][r] *= alpha;
        // ---- O^T += V^T . P^T, two k-steps of 32 keys
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4_t p0 = s[2 * u], p1 = s[2 * u + 1];
            const u32x4_t pk = u32x4_t{pack_bf16x2(p0[0], p0[1]), pack_bf16x2(p0[2], p0[3]), pack_bf16x2(p1[0], p1[1]),
                                       pack_bf16x2(p1[2], p1[3])};
            const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pk);
#pragma unroll
            for (int md = 0; md < DV; ++md) {
                const char* vb = ldsV + tr_off + md * 32;
                const bf16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4_t*)(vb + (2 * u) * 16 * C::VROW));
                const bf16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4_t*)(vb + (2 * u + 1) * 16 * C::VROW));
                bf16x8_t a;
                a[0] = v0[0]; a[1] = v0[1]; a[2] = v0[2]; a[3] = v0[3];
                a[4] = v1[0]; a[5] = v1[1]; a[6] = v1[2]; a[7] = v1[3];
                o[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb, o[md], 0, 0, 0);
            }
        }
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, kregA, vregA);
        if (kt + 1 < nkt) tile(kt + 1, kregB, vregB);
    }
    // ---- the lane groups hold disjoint keys of the same query: combine the row sums, normalise, store
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const int q = q0 + r16;
    if (q < Tq) {
        const float inv = 1.0f / l_run;
        uint16_t* op = O + ((int64_t)b * Tq + q) * ldo + h * dh;
#pragma unroll
        for (int md = 0; md < DV; ++md) {
            const int d = md * 16 + 4 * g;
            if (d < dh) {
                u32x2_t w;
                w[0] = pack_bf16x2(o[md][0] * inv, o[md][1] * inv);
                w[1] = pack_bf16x2(o[md][2] * inv, o[md][3] * inv);
                *(u32x2_t*)(op + d) = w;
            }
        }
    }
}

template <int KS, int DV>
hipError_t launch_fa(const uint16_t* Q, int64_t ldq, const uint16_t* K, int64_t ldk, const uint16_t* V, int64_t ldv,
                     uint16_t* O, int64_t ldo, int n, int heads, int Tq, int Tk, int dh, hipStream_t st) {
    const float scale_log2 = 1.4426950408889634f / sqrtf((float)dh);
    const int nqb = (Tq + 63) / 64;
    dim3 grid((unsigned)((int64_t)nqb * heads * n));
    hipLaunchKernelGGL((sd_flash_attention_kernel<KS, DV>), grid, dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, Tq, Tk, dh,
                       scale_log2, heads, nqb);
    return hipGetLastError();
}

}  // namespace

// Q [n * Tq, ldq], K / V [n * Tk, ldk / ldv], O [n * Tq, ldo]; head h occupies columns [h * dh, (h + 1) * dh)
hipError_t sd_flash_attention(const uint16_t* Q, int64_t ldq, const uint16_t* K, int64_t ldk, const uint16_t* V, int64_t ldv,
                              uint16_t* O, int64_t ldo, int n, int heads, int Tq, int Tk, int dh, hipStream_t st) {
    if (n <= 0 || Tq <= 0) return hipSuccess;
    if (Tk <= 0 || dh % 8 != 0 || dh < 8 || dh > 160 || heads > 65535 || n > 65535 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4)
        return hipErrorInvalidValue;
    if (dh <= 32) return launch_fa<1, 2>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, dh, st);
    if (dh <= 48) return launch_fa<2, 3>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, dh, st);
    if (dh <= 64) return launch_fa<2, 4>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, dh, st);
    if (dh <= 80) return launch_fa<3, 5>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, dh, st);
    if (dh <= 96) return launch_fa<3, 6>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, dh, st);
    if (dh <= 128) return launch_fa<4, 8>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, dh, st);
    return launch_fa<5, 10>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, dh, st);
}
