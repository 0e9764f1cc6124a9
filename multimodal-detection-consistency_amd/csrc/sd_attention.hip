// Streaming (flash) attention for the latent-diffusion UNet: self-attention over T = 4096 / 1024 / 256 / 64 latent
// positions and cross-attention onto the 77 text states, 8 heads of head_dim 40 / 80 / 160 (supported: 8, 16, 24, 32,
// 40, 48, 56, 64, 80, 96, 128, 160).
//
// One workgroup of NW (4 or 8) waves per (NW * 16 * QB queries, head, sample); keys / values stream through LDS in tiles of 64 with an
// online softmax.  The products are "swapped" as in attention.hip so that the query sits on the lane (column) index:
//   S^T[key, q] = K . Q^T     MFMA 16x16x32: A = K rows from LDS (ds_read_b128), B = Q fragments held in registers
//   O^T[d, q]  += V^T . P^T   A = V^T by ds_read_b64_tr_b16 (hardware transpose of the row-major V tile), B = P^T packed
//                             from the S^T accumulators with no lane movement
// so the running max / sum are per-lane scalars (+ two xor-shuffles per tile for the max) and a lane's output is 4
// consecutive head dims of its query (8-byte stores).  head_dim is zero-padded to 32 * KS for Q.K^T and covered by
// DV tiles of 16 for P.V; the pad columns of the LDS tiles are zeroed ONCE (the per-tile staging never touches them).
// A wave owns QB blocks of 16 queries: every K / V fragment read from LDS feeds QB MFMAs, and the tile staging and its
// two barriers are shared by 64 * QB queries.
//
// The tile loop is written for a short instruction stream (the first cut spent ~600 instructions per tile at head_dim 40,
// most of them exec-masked load guards and 64-bit address arithmetic; `git log` has it): head_dim is a template
// parameter (piece -> (row, chunk) is a compile-time division), full tiles are loaded unguarded through per-thread
// pointers, the ragged last tile clamps its row index instead of masking, and the "key >= Tk" initialisation of the
// scores exists only in that last tile.
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>

namespace {

__device__ __forceinline__ float mx3(float a, float b) { return __builtin_elementwise_maximum(a, b); }

template <int DH, int QB, int NW>
struct FaCfg {
    static constexpr int KS = (DH + 31) / 32;           // 32-wide k-steps of Q.K^T
    static constexpr int DV = (DH + 15) / 16;           // 16-wide output tiles of P.V
    static constexpr int DCH = DH / 8;                  // valid 16-byte chunks of a head row
    static constexpr int KCH = KS * 4;                  // chunks of a K row in LDS (zero padded)
    static constexpr int VCH = DV * 2;
    // Row pitches chosen for conflict-free fragment reads (enumerated over the instructions' lane groups,
    // MI355X_MICROARCH.md "LDS"): ds_read_b128 of (row = lane & 15, chunk = lane >> 4) is conflict-free when the pitch is
    // 16 * m bytes with m = 2 mod 4; ds_read_b64_tr_b16 of (row = 4 g + (i >> 2), 8-byte column i & 3) when the pitch is
    // 32 mod 64 bytes (a pitch of 128 B serialises it 4-way: measured 1.7 -> ... ms on the 64 x 64 self-attention).
    static constexpr int KROW = KCH * 16 + 32;
    static constexpr int VROW = (VCH * 16 - 32 + 63) / 64 * 64 + 32;
    static constexpr int KBYTES = 64 * KROW;
    static constexpr int VBYTES = 64 * VROW;
    static constexpr int PIECES = 64 * DCH;             // 16-byte pieces of one operand tile
    static constexpr int NT = NW * 64;                  // threads per workgroup
    static constexpr int NIT = (PIECES + NT - 1) / NT;  // staging pieces per thread and operand
    // TWO eight-wave workgroups per CU (128 registers per lane) where the kernel fits them: head dims <= 48 with ONE staging
    // register set (the other workgroup's arithmetic hides a tile's load latency instead of a second set).  Two resident
    // workgroups run out of phase, so one's softmax overlaps the other's products: head_dim 40 at T = 4096 1 099 -> 928 us
    // (head_dim 32: 1 052 -> 876).  Head dims 56 / 64 fit with one spilled register and run the same as before; 80 needs 174.
    static constexpr bool OCC2 = NW == 8 && DH <= 48;
    static constexpr int NSETS = OCC2 ? 1 : 2;
};

template <int DH, int QB, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 && DH <= 48) ? 4 : 1)    /* = FaCfg::OCC2 */ void sd_flash_attention_kernel(const uint16_t* __restrict__ Q, int64_t ldq,
                                                                 const uint16_t* __restrict__ K, int64_t ldk,
                                                                 const uint16_t* __restrict__ V, int64_t ldv,
                                                                 uint16_t* __restrict__ O, int64_t ldo, int Tq, int Tk,
                                                                 float scale_log2, int heads, int nqb) {
    using C = FaCfg<DH, QB, NW>;
    constexpr int KS = C::KS, DV = C::DV, NT = C::NT;
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
    const int bh = lin / nqb, qblk = lin - bh * nqb;
    const int b = bh / heads, h = bh - b * heads;
    const int q0 = qblk * (NW * 16 * QB) + wave * (16 * QB);

    // ---- zero the pad columns of both tiles once
    if (C::KCH > C::DCH)
        for (int i = tid; i < 64 * (C::KCH - C::DCH); i += NT) {
            const int row = i / (C::KCH - C::DCH), c = C::DCH + i % (C::KCH - C::DCH);
            *(u32x4_t*)(ldsK + row * C::KROW + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }
    if (C::VCH > C::DCH)
        for (int i = tid; i < 64 * (C::VCH - C::DCH); i += NT) {
            const int row = i / (C::VCH - C::DCH), c = C::DCH + i % (C::VCH - C::DCH);
            *(u32x4_t*)(ldsV + row * C::VROW + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }

    // ---- Q fragments (registers, whole kernel): lane holds Q[q][32 s + 8 g .. + 7] of its QB query blocks
    bf16x8_t bq[QB][KS];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        int qrow = q0 + u * 16 + r16;
        qrow = qrow < Tq ? qrow : Tq - 1;
        const uint16_t* qp = Q + ((int64_t)b * Tq + qrow) * ldq + h * DH;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 4 * s + g;
            u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
            if (c < C::DCH) v = *(const u32x4_t*)(qp + c * 8);
            bq[u][s] = __builtin_bit_cast(bf16x8_t, v);
        }
    }

    // ---- staging: piece idx = tid + i * 256 -> (row, chunk) by a compile-time division; per-thread source pointers
    const uint16_t* kbase = K + (int64_t)b * Tk * ldk + h * DH;
    const uint16_t* vbase = V + (int64_t)b * Tk * ldv + h * DH;
    int prow[C::NIT], pcol[C::NIT];
    const uint16_t* kp[C::NIT];
    const uint16_t* vp[C::NIT];
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
        int idx = tid + i * NT;
        if (idx >= C::PIECES) idx = C::NIT > 1 ? tid : tid % C::PIECES;      // no last piece: re-read a valid one (never written to LDS)
        prow[i] = idx / C::DCH;
        pcol[i] = idx - prow[i] * C::DCH;
        kp[i] = kbase + (int64_t)prow[i] * ldk + pcol[i] * 8;
        vp[i] = vbase + (int64_t)prow[i] * ldv + pcol[i] * 8;
    }
    const bool last_piece_valid = (tid + (C::NIT - 1) * NT) < C::PIECES;      // only the last piece of a thread can be out of range
    // TWO register sets: tile t + 2 is requested while tile t is multiplied.
    // Every load below is UNCONDITIONAL (a thread without a last piece re-reads piece 0 of its column; the tile index is
    // clamped by the caller; only the LDS write is masked): vmcnt is a counter, and with the loads behind per-thread or
    // per-tile conditionals hipcc could not count the other register set's loads -- the wait before a tile's LDS write
    // was `vmcnt(0)`, which drained the set requested one tile ago as well and made the prefetch one tile deep.
    u32x4_t kregA[C::NIT], vregA[C::NIT], kregB[C::NIT], vregB[C::NIT];
    auto prefetch = [&](int key0, u32x4_t (&kreg)[C::NIT], u32x4_t (&vreg)[C::NIT]) __attribute__((always_inline)) {
        if (key0 + 64 <= Tk) {                 // full tile (uniform): unguarded loads through the per-thread pointers
#pragma unroll
            for (int i = 0; i < C::NIT; ++i) {
                kreg[i] = *(const u32x4_t*)(kp[i] + (int64_t)key0 * ldk);
                vreg[i] = *(const u32x4_t*)(vp[i] + (int64_t)key0 * ldv);
            }
        } else {                               // ragged last tile: rows clamped to the last key (their scores start at -inf)
#pragma unroll
            for (int i = 0; i < C::NIT; ++i) {
                int r = key0 + prow[i];
                r = r < Tk ? r : Tk - 1;
                kreg[i] = *(const u32x4_t*)(kbase + (int64_t)r * ldk + pcol[i] * 8);
                vreg[i] = *(const u32x4_t*)(vbase + (int64_t)r * ldv + pcol[i] * 8);
            }
        }
    };
    auto commit = [&](const u32x4_t (&kreg)[C::NIT], const u32x4_t (&vreg)[C::NIT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::NIT; ++i) {
            if (i < C::NIT - 1 || last_piece_valid) {
                *(u32x4_t*)(ldsK + prow[i] * C::KROW + pcol[i] * 16) = kreg[i];
                *(u32x4_t*)(ldsV + prow[i] * C::VROW + pcol[i] * 16) = vreg[i];
            }
        }
    };

    // transposed-read lane address (attention.hip): lane i of a 16-lane group supplies key row 4 g + (i >> 2),
    // head dims 4 (i & 3) .. + 3 of a [16 keys][16 dims] block
    const int tr_off = (4 * g + (r16 >> 2)) * C::VROW + ((r16 & 3) << 3);

    // Softmax bookkeeping per 16-query block (second session of round 4; the tile loop is bound by vector issue at head_dim
    // 40: 0.47 clocks per score and SIMD against 0.22 of MFMA):
    //  * the row sums come out of the matrix pipe -- one more "V^T tile" of ones per 32 keys gives sum_k P[k, q] in every
    //    accumulator row (as attention.hip does), over the SAME bf16-rounded probabilities the numerator uses and over all
    //    four lane groups' keys at once: 16 adds per block and tile and the two closing shuffles less;
    //  * the reference maximum `m_run` is LAZY: it moves (and the running output / sum are rescaled) only when the true
    //    maximum has grown by more than 2^8 in the exponent's units -- the probabilities are then at most 2^8 instead of
    //    at most 1, which neither bf16 nor the fp32 accumulators notice, and the final o / l does not depend on the
    //    reference.  After the first tiles of a row no rescale happens: the exp of alpha and DV * 4 + 1 multiplies per
    //    block and tile are skipped by a wave-uniform branch.
    f32x4_t o[QB][DV], osum[QB];
    float m_run[QB];
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        m_run[u] = -INFINITY;
        osum[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int md = 0; md < DV; ++md) o[u][md] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    const int nkt = (Tk + 63) >> 6;
    prefetch(0, kregA, vregA);
    if (C::NSETS == 2) prefetch(nkt > 1 ? 64 : 0, kregB, vregB);
    auto tile = [&](int kt, u32x4_t (&kreg)[C::NIT], u32x4_t (&vreg)[C::NIT]) __attribute__((always_inline)) {
        __syncthreads();                       // every wave is done reading the previous tile (and the pad columns are zero)
        commit(kreg, vreg);
        __syncthreads();
        prefetch((kt + C::NSETS < nkt ? kt + C::NSETS : nkt - 1) * 64, kreg, vreg);      // in flight behind NSETS tiles' arithmetic (past the end: the last tile again, never used)
        const int key0 = kt * 64;
        const bool ragged = key0 + 64 > Tk;   // uniform; true for the last tile only
        // ---- S^T = K . Q^T for the 4 key sub-tiles of 16; the K fragment of a (sub-tile, k-step) feeds all QB blocks
        f32x4_t s[QB][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4_t init = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ragged) {
#pragma unroll
                for (int r = 0; r < 4; ++r) init[r] = (key0 + t * 16 + 4 * g + r >= Tk) ? -INFINITY : 0.f;
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) s[u][t] = init;
            const char* kr = ldsK + (t * 16 + r16) * C::KROW + g * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t a = *(const bf16x8_t*)(kr + ks * 64);
#pragma unroll
                for (int u = 0; u < QB; ++u) s[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[u][ks], s[u][t], 0, 0, 0);
            }
        }
        bf16x8_t pb[QB][2];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            // IEEE-754-2019 maximum (v_maximum3_f32, NaN-propagating: no canonicalising copies of the MFMA outputs)
            float mx = mx3(mx3(s[u][0][0], s[u][0][1]), mx3(s[u][0][2], s[u][0][3]));
#pragma unroll
            for (int t = 1; t < 4; ++t) mx = mx3(mx, mx3(mx3(s[u][t][0], s[u][t][1]), mx3(s[u][t][2], s[u][t][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[u], mx);          // finite: every tile holds at least one key < Tk
            // a row's first tile: m_run = -inf, so the difference is +inf and the reference is set (alpha = 0 on zeros)
            const bool move = (m_new - m_run[u]) * scale_log2 > 8.0f;
            if (__builtin_amdgcn_ballot_w64(move) != 0) {                       // wave-uniform; rare after a row's first tiles
                const float m_use = move ? m_new : m_run[u];
                const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_use) * scale_log2);      // 1 for the lanes that stay
                m_run[u] = m_use;
                osum[u] *= alpha;
#pragma unroll
                for (int md = 0; md < DV; ++md) o[u][md] *= alpha;
            }
            const float mns = m_run[u] * scale_log2;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[u][t][r] = __builtin_amdgcn_exp2f(fmaf(s[u][t][r], scale_log2, -mns));
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4_t p0 = s[u][2 * kk], p1 = s[u][2 * kk + 1];
                pb[u][kk] = __builtin_bit_cast(bf16x8_t, u32x4_t{pack_bf16x2(p0[0], p0[1]), pack_bf16x2(p0[2], p0[3]),
                                                                  pack_bf16x2(p1[0], p1[1]), pack_bf16x2(p1[2], p1[3])});
            }
        }
        // ---- O^T += V^T . P^T, two k-steps of 32 keys; a V^T fragment feeds all QB blocks
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int md = 0; md < DV; ++md) {
                const char* vb = ldsV + tr_off + md * 32;
                const bf16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4_t*)(vb + (2 * kk) * 16 * C::VROW));
                const bf16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4_t*)(vb + (2 * kk + 1) * 16 * C::VROW));
                bf16x8_t a;
                a[0] = v0[0]; a[1] = v0[1]; a[2] = v0[2]; a[3] = v0[3];
                a[4] = v1[0]; a[5] = v1[1]; a[6] = v1[2]; a[7] = v1[3];
#pragma unroll
                for (int u = 0; u < QB; ++u) o[u][md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[u][kk], o[u][md], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) osum[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[u][kk], osum[u], 0, 0, 0);
        }
    };
    if (C::NSETS == 1) {
        for (int kt = 0; kt < nkt; ++kt) tile(kt, kregA, vregA);
    } else {
        for (int kt = 0; kt < nkt; kt += 2) {
            tile(kt, kregA, vregA);
            if (kt + 1 < nkt) tile(kt + 1, kregB, vregB);
        }
    }
    // ---- every row of the ones tile holds the query's sum over ALL keys: normalise, store
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const float l = osum[u][0];
        const int q = q0 + u * 16 + r16;
        if (q < Tq) {
            const float inv = 1.0f / l;
            uint16_t* op = O + ((int64_t)b * Tq + q) * ldo + h * DH;
#pragma unroll
            for (int md = 0; md < DV; ++md) {
                const int d = md * 16 + 4 * g;
                if (d < DH) {
                    u32x2_t w;
                    w[0] = pack_bf16x2(o[u][md][0] * inv, o[u][md][1] * inv);
                    w[1] = pack_bf16x2(o[u][md][2] * inv, o[u][md][3] * inv);
                    *(u32x2_t*)(op + d) = w;
                }
            }
        }
    }
}

template <int DH, int QB, int NW>
hipError_t launch_fa(const uint16_t* Q, int64_t ldq, const uint16_t* K, int64_t ldk, const uint16_t* V, int64_t ldv,
                     uint16_t* O, int64_t ldo, int n, int heads, int Tq, int Tk, hipStream_t st) {
    const float scale_log2 = 1.4426950408889634f / sqrtf((float)DH);
    const int nqb = (Tq + NW * 16 * QB - 1) / (NW * 16 * QB);
    dim3 grid((unsigned)((int64_t)nqb * heads * n));
    hipLaunchKernelGGL((sd_flash_attention_kernel<DH, QB, NW>), grid, dim3(NW * 64), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, Tq, Tk,
                       scale_log2, heads, nqb);
    return hipGetLastError();
}

}  // namespace

// Q [n * Tq, ldq], K / V [n * Tk, ldk / ldv], O [n * Tq, ldo]; head h occupies columns [h * dh, (h + 1) * dh)
hipError_t sd_flash_attention(const uint16_t* Q, int64_t ldq, const uint16_t* K, int64_t ldk, const uint16_t* V, int64_t ldv,
                              uint16_t* O, int64_t ldo, int n, int heads, int Tq, int Tk, int dh, hipStream_t st) {
    if (n <= 0 || Tq <= 0) return hipSuccess;
    if (Tk <= 0 || heads < 1 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || (int64_t)n * heads * ((Tq + 63) / 64) > 0x7fffffffLL)
        return hipErrorInvalidValue;
    // Workgroup shape.  Every workgroup streams ALL keys / values of its (sample, head), so the bytes pulled through the
    // cache hierarchy scale with 1 / (queries per workgroup): at T = 4096 and 64 queries per workgroup a layer re-read
    // its 126 MB of K / V 64 times (8 GB, measured memory-bound at 1.5 - 2.1 ms); 8 waves x 2 blocks = 256 queries per
    // workgroup cut that to 2 GB.  Short sequences keep 4 waves x 1 block so that the grid still fills the chip.
    static const int shape_env = [] { const char* v = getenv("TVC_SD_ATTN_SHAPE"); return v ? atoi(v) : 0; }();     // experiments: 1 / 2 / 3
    const int64_t items = (int64_t)n * heads;
    int shape = 1;                                           // 1: 4 waves x 1 block (64 queries)
    if (items * ((Tq + 127) / 128) >= 512) shape = 2;        // 2: 4 waves x 2 blocks (128 queries)
    if (items * ((Tq + 255) / 256) >= 512) shape = 3;        // 3: 8 waves x 2 blocks (256 queries)
    if (shape_env) shape = shape_env;
#define FA_CASE(DH_, MAXSHAPE_)                                                                                          \
    case DH_: {                                                                                                         \
        const int sh = shape < MAXSHAPE_ ? shape : MAXSHAPE_;                                                           \
        if (sh == 3) return launch_fa<DH_, (MAXSHAPE_ >= 3 ? 2 : 1), (MAXSHAPE_ >= 3 ? 8 : 4)>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, st); \
        if (sh == 2) return launch_fa<DH_, (MAXSHAPE_ >= 2 ? 2 : 1), 4>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, st);      \
        return launch_fa<DH_, 1, 4>(Q, ldq, K, ldk, V, ldv, O, ldo, n, heads, Tq, Tk, st);                             \
    }
    switch (dh) {
        FA_CASE(8, 3) FA_CASE(16, 3) FA_CASE(24, 3) FA_CASE(32, 3) FA_CASE(40, 3) FA_CASE(48, 3) FA_CASE(56, 3) FA_CASE(64, 3)
        FA_CASE(80, 3) FA_CASE(96, 1) FA_CASE(128, 1) FA_CASE(160, 1)
        default: return hipErrorInvalidValue;       // head_dim not instantiated
    }
#undef FA_CASE
}
