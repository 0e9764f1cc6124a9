// Dense bf16 GEMM with fused epilogues for the CLIP towers (K1/K2) and the
// bank-search pre-pass.  See gemm_core.hpp for the tiling.
#include "gemm_epilogue.hpp"
#include "gemm_ring4.hpp"
#include <cstdlib>
#include <mutex>
#include <type_traits>

// s_setprio levels of the MFMA phases of the two wave groups of the ring kernel (see gemm_ring_kernel)
#define TVC_PRIO_G0 1
#define TVC_PRIO_G1 2

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
    gemm_tile_epilogue<EPI>(acc, g, e, i0, j0, wave >> 2, wave & 3, lane);
}


// ---------------------------------------------------------------------------
// Persistent ring-pipelined variant (gemm_ring.hpp): used for the big tower GEMMs.
// ---------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_ring_kernel(GemmOperands g, GemmEpilogue e, int nIt, int nJt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kpp = g.ksteps_per_plane * (GEMM_BK / RING_BK);     // 32-deep stages per plane
    const int nk = g.planes * kpp;                                 // stages per tile

    RingSchedule sch;
    sch.init(nIt * nJt);
    const int my_tiles = sch.count();
    const int total = my_tiles * nk;                               // stages in this workgroup's stream
    if (total == 0) return;

    const uint32_t smem_lds = lds_addr(smem);
    // ---- issue side: scalar tile bases + per-lane 32-bit offsets (recomputed per tile only)
    int is_tile = 0, is_p = 0, is_kk = 0, is_n = 0;
    const char* is_abase; const char* is_bbase;
    uint32_t va[2], vb[2];
    auto issue_tile = [&](int lin) {
        const int jt = lin / nIt;
        const int i0 = (lin - jt * nIt) * GEMM_BM, j0 = jt * GEMM_BN;
        is_abase = (const char*)(g.A + (int64_t)i0 * g.lda);
        is_bbase = (const char*)(g.B + (int64_t)j0 * g.ldb);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 32 + i * 16 + (lane >> 2);
            const int c = (lane & 3) ^ (3 * ((r >> 3) & 1));
            int ra = r, rb = r;
            if (i0 + ra >= g.I) ra = g.I - 1 - i0;
            if (j0 + rb >= g.J) rb = g.J - 1 - j0;
            va[i] = (uint32_t)ra * (uint32_t)(g.lda * 2) + c * 16;
            vb[i] = (uint32_t)rb * (uint32_t)(g.ldb * 2) + c * 16;
        }
    };
    issue_tile(sch.tile(0));
    auto issue = [&]() {
        const uint32_t slot = smem_lds + (is_n & (RING_SLOTS - 1)) * RING_SLOT_BYTES + wave * (32 * 64);
        const char* ab = is_abase + (int64_t)(g.a_plane_off[is_p] + is_kk * RING_BK) * 2;
        const char* bb = is_bbase + (int64_t)(g.b_plane_off[is_p] + is_kk * RING_BK) * 2;
        glds16x4_asm(ab, va[0], va[1], slot, bb, vb[0], vb[1], slot + RING_HALF_BYTES);   // 16 rows = 0x400 B apart
        ++is_n;
        if (++is_kk == kpp) {
            is_kk = 0;
            if (++is_p == g.planes) {
                is_p = 0;
                if (++is_tile < my_tiles) issue_tile(sch.tile(is_tile));
            }
        }
    };

    // lane-constant fragment read offsets inside a stage
    const int pos = ((lane >> 4) ^ (3 * ((lane >> 3) & 1))) * 16;
    const int a_off = (wm * 128 + (lane & 15)) * 64 + pos;
    const int b_off = RING_HALF_BYTES + (wn * 64 + (lane & 15)) * 64 + pos;

    // ---- rotated ping-pong schedule, ONE barrier per stage ----------------------------------
    // Waves w and w+4 share a SIMD.  Per barrier interval S every wave issues its pieces of
    // stage S+3, reads stage S's fragments and multiplies one stage, but the two groups run the
    // phases in opposite order:
    //     group 0 (waves 0-3):  L(S) then C(S)          load fragments, then 32 MFMAs
    //     group 1 (waves 4-7):  C(S-1) then L(S)        32 MFMAs on last interval's fragments, then load
    // so while one wave of a SIMD feeds the matrix pipe its partner issues LDS-DMA / ds_reads,
    // and the barrier + scalar bookkeeping is paid once per 64 MFMAs of a SIMD.
    //   RAW  stage S is read in interval S; every wave retired its pieces of stage S (counted
    //        vmcnt) before the barrier that ended interval S-1.
    //   WAR  stage S+3 goes to slot (S-1)%4, whose reads (interval S-1, both groups, lgkmcnt(0)
    //        before the barrier) are complete.
    //   Flight time of a stage: issued in interval S-3+..., retired at the end of interval S-1.
    const int gid = wave >> 2;
    for (int s = 0; s < 3 && s < total; ++s) issue();
    if (total > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // stage 0 (own pieces)
    else if (total == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    gemm_acc_t acc;
    gemm_zero_acc(acc);
    bf16x8_t a[8], b[4];
    int ct = 0, cks = 0;         // tile / k-stage of the stage the MFMAs of this wave work on

    auto tile_origin = [&](int t, int& i0, int& j0) {
        const int lin = sch.tile(t);
        const int jt = lin / nIt;
        i0 = (lin - jt * nIt) * GEMM_BM; j0 = jt * GEMM_BN;
    };
    auto load_frags = [&](int S) {
        const char* slot = smem + (S & (RING_SLOTS - 1)) * RING_SLOT_BYTES;
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = *(const bf16x8_t*)(slot + a_off + m * 1024);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *(const bf16x8_t*)(slot + b_off + n * 1024);
    };
    auto mfma_stage = [&](auto prio) {
        __builtin_amdgcn_s_setprio(decltype(prio)::value);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // after the MFMAs of stage (ct, cks): tile epilogue when it was the tile's last stage
    auto finish_stage = [&](int) {
        if (++cks == nk) {
            int i0, j0;
            tile_origin(ct, i0, j0);
            gemm_tile_epilogue<EPI, true>(acc, g, e, i0, j0, wm, wn, lane, smem + RING_LDS_BYTES + (ct & 1) * 1024);
            gemm_zero_acc(acc);
            cks = 0; ++ct;
        }
    };
    auto retire_and_barrier = [&](int S) {
        const int n_out = (S + 3 < total ? S + 3 : total - 1) - S;           // stages in flight beyond S
        // (Rounds 1-3 credited a fast epilogue's 16 / 32 stores here -- vmcnt(24) / vmcnt(40) -- so that the waits behind a
        // tile end did not drain them.  That assumed the counter retires in issue order; it does not between loads and stores
        // (round 4: a vmcnt(2) meant to leave two stores in flight let OLDER loads arrive late in an attention experiment):
        // stores that retire early would have let the wait pass with the stage's loads still in flight.  It never failed
        // -- the loads were a whole epilogue old -- but this kernel is the any-shape fallback: correct by construction first.)
        if (n_out >= 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (n_out == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (gid == 0) {
        int it = 0, iks = 0;     // tile / k-stage of the stage being loaded (== multiplied) this interval
        for (int S = 0; S < total; ++S) {
            if (S + 3 < total) issue();
            if (iks == 0 && wave == 0 && e.bias) {
                // the tile's 256 bias values -> one of TWO alternating LDS slots (group 1 may still be
                // in the previous tile's epilogue, reading the other slot); retired by this wave's
                // counted waits long before the epilogue (>= 8 stages follow)
                int i0, j0;
                tile_origin(it, i0, j0);
                if (i0 + GEMM_BM <= g.I) glds16_asm(e.bias + i0, lane * 16, smem_lds + RING_LDS_BYTES + (it & 1) * 1024);
            }
            if (++iks == nk) { iks = 0; ++it; }
            load_frags(S);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mfma_stage(std::integral_constant<int, TVC_PRIO_G0>{});
            finish_stage(3);
            retire_and_barrier(S);
        }
    } else {
        for (int S = 0; S < total; ++S) {
            if (S > 0) { mfma_stage(std::integral_constant<int, TVC_PRIO_G1>{}); finish_stage(2); }
            if (S + 3 < total) issue();
            load_frags(S);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            retire_and_barrier(S);
        }
        mfma_stage(std::integral_constant<int, TVC_PRIO_G1>{});
        finish_stage(0);
    }
}


// ---------------------------------------------------------------------------
// Ring main loop, fifth form (round 4; "ring4" in the names is the kernel family): barrier-staggered ping-pong in phases of
// 16 MFMAs after the "256^2 8-phase" recipe of /opt/skills/guides/cdna_hip_programming.md section 5, rebuilt for a
// persistent tile stream.
//
//   * A K-tile (64 deep) is multiplied in FOUR phases, one quadrant (64 out-features x 32 tokens x K 64 = 16 MFMAs)
//     of the wave's 128 x 64 sub-tile each: (Aq0,Bq0) (Aq0,Bq1) (Aq1,Bq1) (Aq1,Bq0).
//   * A phase is  {LDS reads + ONE unit of LDS-DMA}  barrier  {16 MFMAs}  barrier.  The wave group wm = 1 runs one
//     barrier behind wm = 0, so on every SIMD one wave is in its MFMA segment while its partner is in its load segment:
//     the matrix pipe never waits for a ds_read, and the LDS-DMA stream is spread evenly (16 KiB per phase).
//   * LDS: 2 K-tile buffers x (A [256][64] + B [256][64]).  The STAGING unit is the 128 rows one quadrant reads:
//     Aq = rows {wm*128 + q*64 ..+63}, Bq = rows {wn*64 + q*32 ..+31} over all waves (16 KiB = 16 pieces, two per wave).
//     A unit is read in exactly ONE phase (p0: Aq0 + Bq0, p1: Bq1, p2: Aq1; the fragments then stay in registers).
//   * Schedule of K-tile t:  p0 reads Aq0, Bq0 (t), stages Aq1 (t+1);  p1 reads Bq1, stages Aq0 (t+2);  p2 reads Aq1, stages
//     Bq0 (t+2);  p3 stages Bq1 (t+2) and holds the K-tile's ONE counted wait, s_waitcnt vmcnt(6): every unit of K-tile t+1
//     has landed, Aq0 / Bq0 / Bq1 of t+2 stay in flight (3..7 units = 48..112 KiB in flight over a K-tile).
//   * write-after-read: a B unit is restaged two phases after its only read; an A unit ONE phase after it -- its rows are
//     staged and read by the SAME wave group (rA[] below: wave >> 2 = wm), whose waves have all passed the barrier that ends
//     the reading phase's MFMA segment before any of them issues the next load segment.  read-after-write: the wait sits
//     before p3's first barrier, the reads start in the next phase (one barrier more than the wait, because the two groups
//     are a barrier apart).  The previous tile's epilogue stores are older than the first K-tile's Aq1 and drain with it.
//   * Two K-tiles per loop iteration when a tile has an even number of them (every tower shape): the LDS buffer of a K-tile
//     is then a compile-time constant -- fewer instructions in every load segment, which is the critical path of a slot.
//   * Tile ends.  Group 1 runs its epilogue before the tile's last barrier and group 0 after it, so the two overlap; the
//     epilogue reads the lane's bias vectors once (gemm_tile_epilogue<.., BIAS_REGS>).
//   The fp32 sums are taken in the same order as in form 1 and in the round-2/3 form 4 (three counted waits per K-tile, units
//   a phase later: git show 1744251:multimodal-detection-consistency_amd/csrc/gemm.hip): results are bit-identical; same-box
//   A/B against form 4: +2.5-3.3 % on the four tower shapes, 1 355 against 1 250 TFLOP/s at 4096^3
//   (profiles/r04_gemm_form5_ab.log).
// ---------------------------------------------------------------------------
//   SPLIT (the latent-diffusion model's fixed K split, tvc_sd.cpp Run::fixed_split): the stream walks VIRTUAL tiles
//   (tile, slice s of S) -- K-tiles [s * nkt, (s + 1) * nkt) of the tile, nkt = K-tiles / S -- and a virtual tile ends in a
//   store of its fp32 accumulators to ws[tile * S + s] in lane order (what gemm_splitk_partial_kernel writes, bit for bit:
//   the same products in the same order), summed by gemm_splitk_finish_kernel.  A slice of a few-tile launch then runs at
//   the ring's rate instead of the one-tile kernel's (about half of it).
template <int EPI, bool SPLIT>
__device__ __forceinline__ void gemm_ring4_body(const GemmOperands& g, const GemmEpilogue& e, int nIt, int nJt, int S, float* ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kpp = g.ksteps_per_plane;
    const int nkt = SPLIT ? g.planes * kpp / S : g.planes * kpp;

    RingSchedule sch;
    sch.init(SPLIT ? nIt * nJt * S : nIt * nJt);
    const int my_tiles = sch.count();
    const int T = my_tiles * nkt;                                  // K-tiles in this workgroup's stream
    if (T == 0) return;

    const uint32_t smem_lds = lds_addr(smem);
    struct Cursor { int tile, p, kk; const char* abase; const char* bbase; const char* ap; const char* bp; int left; };
    auto opaque_lane = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto cur_tile = [&](Cursor& c, int lin) __attribute__((always_inline)) {
        int slice = 0;
        if (SPLIT) { const int vt = lin; lin = vt / S; slice = vt - lin * S; }
        const int jt = lin / nIt;
        const int i0 = (lin - jt * nIt) * GEMM_BM, j0 = jt * GEMM_BN;
        c.abase = (const char*)(g.A + (int64_t)i0 * g.lda);
        c.bbase = (const char*)(g.B + (int64_t)j0 * g.ldb);
        if (SPLIT) {                        // stand on K-tile slice * nkt of the tile
            const int b = slice * nkt;
            c.p = b / kpp; c.kk = b - c.p * kpp; c.left = nkt;
            c.ap = c.abase + ((int64_t)g.a_plane_off[c.p] + (int64_t)c.kk * GEMM_BK) * 2;
            c.bp = c.bbase + ((int64_t)g.b_plane_off[c.p] + (int64_t)c.kk * GEMM_BK) * 2;
        } else {
            c.ap = c.abase + (int64_t)g.a_plane_off[0] * 2;
            c.bp = c.bbase + (int64_t)g.b_plane_off[0] * 2;
        }
    };
    auto cur_advance = [&](Cursor& c) __attribute__((always_inline)) {
        c.ap += GEMM_BK * 2; c.bp += GEMM_BK * 2;
        if (SPLIT) {
            if (--c.left == 0) {
                cur_tile(c, sch.tile(++c.tile < my_tiles ? c.tile : my_tiles - 1));
            } else if (++c.kk == kpp) {
                c.kk = 0; ++c.p;
                c.ap = c.abase + (int64_t)g.a_plane_off[c.p] * 2;
                c.bp = c.bbase + (int64_t)g.b_plane_off[c.p] * 2;
            }
            return;
        }
        if (++c.kk == kpp) {
            c.kk = 0;
            if (++c.p == g.planes) {
                c.p = 0;
                // past the last tile the stream re-stages the LAST tile's rows (nobody reads them; the stages and
                // waits of the loop then need no end-of-stream cases: fewer branches in every load segment)
                cur_tile(c, sch.tile(++c.tile < my_tiles ? c.tile : my_tiles - 1));
            } else {
                c.ap = c.abase + (int64_t)g.a_plane_off[c.p] * 2;
                c.bp = c.bbase + (int64_t)g.b_plane_off[c.p] * 2;
            }
        }
    };
    Cursor is{0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    cur_tile(is, sch.tile(0));
    // Staging.  Units in stream order per K-tile: Aq0, Bq0, Bq1, Aq1 (issued at phases p2, p3 of K-tile t-2 and p0, p1
    // of K-tile t-1).  The kind a phase issues is a compile-time constant; the cursor `is` stands on the K-tile being
    // issued and moves on after its Aq1.  Lane offsets and row offsets are loop constants (4 VGPRs, 4 SGPRs): a load
    // segment must stay shorter than the partner's 16 MFMAs.
    const uint32_t pitchA = (uint32_t)(g.lda * 2), pitchB = (uint32_t)(g.ldb * 2);              // bytes per row
    const int rl = lane >> 3;
    const uint32_t swz = (uint32_t)(((lane & 7) ^ ((rl >> 1) & 7)) * 16);
    const uint32_t vA0 = (uint32_t)rl * pitchA + swz, vA1 = (vA0 ^ 64u) + 8u * pitchA;
    const uint32_t vB0 = (uint32_t)rl * pitchB + swz, vB1 = (vB0 ^ 64u) + 8u * pitchB;
    // this wave's 16 rows of unit Aq / Bq: 64-row block of wave group (wave >> 2), 32-row block of wave column (wave >> 1)
    const int rA[2] = {(wave >> 2) * 128 + (wave & 3) * 16, (wave >> 2) * 128 + 64 + (wave & 3) * 16};
    const int rB[2] = {(wave >> 1) * 64 + (wave & 1) * 16, (wave >> 1) * 64 + 32 + (wave & 1) * 16};
    uint32_t buf_issue = smem_lds;   // LDS buffer of the K-tile the cursor stands on
    auto issue_unit = [&](auto kind_c) __attribute__((always_inline)) {
        constexpr int kind = decltype(kind_c)::value;                  // 0 Aq0, 1 Bq0, 2 Bq1, 3 Aq1
        constexpr bool isA = (kind == 0 || kind == 3);
        constexpr int q = (kind >= 2) ? 1 : 0;
        const int row = isA ? rA[q] : rB[q];
        const char* base = (isA ? is.ap : is.bp) + (uint32_t)row * (isA ? pitchA : pitchB);
        const uint32_t dst = buf_issue + (isA ? 0 : GEMM_TILE_BYTES) + row * 128;
        if (isA) glds16_rows2_asm(base, vA0, vA1, dst); else glds16_rows2_asm(base, vB0, vB1, dst);
        if (kind == 3) { buf_issue = (buf_issue == smem_lds) ? smem_lds + R3_SLOT_BYTES : smem_lds; cur_advance(is); }
    };
    using U_A0 = std::integral_constant<int, 0>; using U_B0 = std::integral_constant<int, 1>;
    using U_B1 = std::integral_constant<int, 2>; using U_A1 = std::integral_constant<int, 3>;

    gemm_acc_t acc;
    gemm_zero_acc(acc);
    bf16x8_t A0f[4][2], A1f[4][2], B0f[2][2], B1f[2][2];
    int ct = 0;

    auto tile_origin = [&](int t, int& i0, int& j0) __attribute__((always_inline)) {
        const int lin = SPLIT ? sch.tile(t) / S : sch.tile(t);
        const int jt = lin / nIt;
        i0 = (lin - jt * nIt) * GEMM_BM; j0 = jt * GEMM_BN;
    };
    typedef const __attribute__((address_space(3))) bf16x8_t* lds_frag_p;
    // Fragment read addresses are loop constants (4 VGPRs; this form has ~50 to spare): a load segment is the
    // critical path of a slot, so it carries no address arithmetic beyond one add of the buffer offset per K-tile.
    const uint32_t sw_rd = (uint32_t)((((lane >> 4) ^ ((lane >> 1) & 7)) * 16));
    const uint32_t a_rd0 = smem_lds + (wm * 128 + (lane & 15)) * 128 + sw_rd, a_rd1 = a_rd0 ^ 64u;                    // k-sub-step 0 / 1
    const uint32_t b_rd0 = smem_lds + GEMM_TILE_BYTES + (wn * 64 + (lane & 15)) * 128 + sw_rd, b_rd1 = b_rd0 ^ 64u;
    auto load_A = [&](int t, int q, bf16x8_t (&a)[4][2]) __attribute__((always_inline)) {
        const uint32_t po = (uint32_t)(t & 1) * R3_SLOT_BYTES + q * 8192;
        const uint32_t r0 = a_rd0 + po, r1 = a_rd1 + po;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            a[m][0] = *(lds_frag_p)(uintptr_t)(r0 + m * 2048);
            a[m][1] = *(lds_frag_p)(uintptr_t)(r1 + m * 2048);
        }
    };
    auto load_B = [&](int t, int q, bf16x8_t (&b)[2][2]) __attribute__((always_inline)) {
        const uint32_t po = (uint32_t)(t & 1) * R3_SLOT_BYTES + q * 4096;
        const uint32_t r0 = b_rd0 + po, r1 = b_rd1 + po;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            b[n][0] = *(lds_frag_p)(uintptr_t)(r0 + n * 2048);
            b[n][1] = *(lds_frag_p)(uintptr_t)(r1 + n * 2048);
        }
    };
#define RING4_MFMA(A_, B_, QA_, QB_)                                                                              \
    {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                          \
            _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                         \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                     \
                    acc[(QA_) * 4 + m][(QB_) * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                  \
                        A_[m][ks], B_[n][ks], acc[(QA_) * 4 + m][(QB_) * 2 + n], 0, 0, 0);                        \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    }
#define RING4_BARRIER() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

    // ---- prologue: K-tile 0 whole, Aq0 / Bq0 / Bq1 of K-tile 1; K-tile 0 landed and published
    issue_unit(U_A0{}); issue_unit(U_B0{}); issue_unit(U_B1{}); issue_unit(U_A1{});
    issue_unit(U_A0{}); issue_unit(U_B0{}); issue_unit(U_B1{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    RING4_BARRIER()
    if (wm == 1) RING4_BARRIER()            // group 1 runs one barrier behind group 0 from here on

    // One K-tile; t selects the LDS buffer (t & 1), a literal in the two-K-tile loop below.
    auto ktile = [&](int t) __attribute__((always_inline)) {
        // ===== p0: read Aq0, Bq0 of K-tile t; stage Aq1 of K-tile t+1
        issue_unit(U_A1{});
        load_B(t, 0, B0f);
        load_A(t, 0, A0f);
        RING4_BARRIER()
        RING4_MFMA(A0f, B0f, 0, 0)
        RING4_BARRIER()
        // ===== p1: read Bq1; stage Aq0 of K-tile t+2
        issue_unit(U_A0{});
        load_B(t, 1, B1f);
        RING4_BARRIER()
        RING4_MFMA(A0f, B1f, 0, 1)
        RING4_BARRIER()
        // ===== p2: read Aq1; stage Bq0 of K-tile t+2
        issue_unit(U_B0{});
        load_A(t, 1, A1f);
        RING4_BARRIER()
        RING4_MFMA(A1f, B1f, 1, 1)
        RING4_BARRIER()
        // ===== p3: stage Bq1 of K-tile t+2; the ONE wait: all of K-tile t+1 landed
        issue_unit(U_B1{});
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        RING4_BARRIER()
        RING4_MFMA(A1f, B0f, 1, 0)
        // (the phase's second barrier is the caller's: at a tile end the two groups place their epilogues differently)
    };
    auto stage_bias = [&](int tile) __attribute__((always_inline)) {
        if (SPLIT) return;                  // bias and epilogue belong to gemm_splitk_finish_kernel
        if (wave == 0 && e.bias) {
            int i0, j0;
            tile_origin(tile, i0, j0);
            if (i0 + GEMM_BM <= g.I) glds16_asm(e.bias + i0, lane * 16, smem_lds + R3_LDS_BYTES + (tile & 1) * 1024);
        }
    };
    stage_bias(0);
    int t = 0;
#pragma unroll 1
    for (ct = 0; ct < my_tiles; ++ct) {
        if ((nkt & 1) == 0) {
            // two K-tiles per loop iteration: a tile starts on LDS buffer 0
            ktile(0); RING4_BARRIER() ktile(1);
#pragma unroll 1
            for (int k = 2; k < nkt; k += 2) {
                RING4_BARRIER()
                ktile(0);
                RING4_BARRIER()
                ktile(1);
            }
            t += nkt;
        } else {                        // odd K-tile counts (e.g. nine planes of K = 320): one K-tile per iteration
            ktile(t);
            ++t;
#pragma unroll 1
            for (int k = 1; k < nkt; ++k, ++t) {
                RING4_BARRIER()
                ktile(t);
            }
        }
        // the NEXT tile's bias slice goes into the queue ahead of this tile's stores
        if (ct + 1 < my_tiles) stage_bias(ct + 1);
        int i0, j0;
        tile_origin(ct, i0, j0);
        // Tile end.  Group 1 runs its epilogue BEFORE the last phase's second barrier, group 0 after it: group 0 reaches
        // that barrier a slot earlier, so the two epilogues run side by side (one slot of bias / convert / store latency
        // per tile instead of two in a row).
        auto tile_end = [&]() __attribute__((always_inline)) {
            if (SPLIT) {
                f32x4_t* o = (f32x4_t*)(ws + (int64_t)sch.tile(ct) * (GEMM_BM * GEMM_BN)) + threadIdx.x;
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) o[(m * 4 + n) * GEMM_THREADS] = acc[m][n];
                return;
            }
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            gemm_tile_epilogue<EPI, true, 4, true>(acc, g, e, i0, j0, wm, wn, lane_e, smem + R3_LDS_BYTES + (ct & 1) * 1024);
        };
        if (wm == 1) { tile_end(); }
        RING4_BARRIER()
        if (wm == 0) { tile_end(); }
        gemm_zero_acc(acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stream's overrun stages must have landed before the LDS is given back
    if (wm == 0) RING4_BARRIER()            // pairs with group 1's last barrier
#undef RING4_MFMA
#undef RING4_BARRIER
}

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_ring4_kernel(GemmOperands g, GemmEpilogue e, int nIt, int nJt) {
    gemm_ring4_body<EPI, false>(g, e, nIt, nJt, 1, nullptr);
}
__global__ __launch_bounds__(GEMM_THREADS) void gemm_ring4_split_kernel(GemmOperands g, float* ws, int nIt, int nJt, int S) {
    GemmEpilogue e;
    e.bias = nullptr; e.out = nullptr; e.ldo = 0;
    gemm_ring4_body<TVC_EPI_F32, true>(g, e, nIt, nJt, S, ws);
}


// ---------------------------------------------------------------------------
// Split-K tail.  A persistent launch over T tiles on 256 workgroups costs ceil(T / 256) rounds; at
// B = 512 images the ViT-L token count is 514 tile columns, so out-proj / FC2 (4 tile rows) pay a
// ninth round for 8 tiles (10.8 % of the launch), QKV / FC1 a 25th / 33rd.  The launcher gives the
// whole rounds to the ring kernel and the few left-over tile columns to these two kernels: every
// left-over tile is multiplied by S workgroups over 1/S of K each (fp32 partial tiles, stored in the
// accumulator's lane order: coalesced), then summed and passed through the usual epilogue.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(GEMM_THREADS) void gemm_splitk_partial_kernel(GemmOperands g, float* __restrict__ ws,
                                                                           int nIt, int jt0, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x / S, split = blockIdx.x - tile * S;
    const int jt = tile / nIt, it = tile - jt * nIt;
    const int nk64 = g.planes * g.ksteps_per_plane;
    const int b = (int)((int64_t)nk64 * split / S), e = (int)((int64_t)nk64 * (split + 1) / S);
    gemm_acc_t acc;
    gemm_zero_acc(acc);
    if (e > b) gemm_mainloop(acc, g, it * GEMM_BM, (jt0 + jt) * GEMM_BN, smem, b, e);
    f32x4_t* o = (f32x4_t*)(ws + (int64_t)blockIdx.x * (GEMM_BM * GEMM_BN)) + threadIdx.x;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) o[(m * 4 + n) * GEMM_THREADS] = acc[m][n];
}

// One workgroup per (left-over tile, 16 x 16 sub-tile pair index m*4+n): 32 workgroups per tile, every
// thread sums the S partial values of its 4 out-features of one token and stores them.
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_splitk_finish_kernel(GemmOperands g, GemmEpilogue e,
                                                                          const float* __restrict__ ws, int nIt,
                                                                          int jt0, int S) {
    const int tile = blockIdx.x >> 5, mn = blockIdx.x & 31;
    const int jt = tile / nIt, it = tile - jt * nIt;
    const int m = mn >> 2, n = mn & 3;
    f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const f32x4_t* p = (const f32x4_t*)(ws + (int64_t)tile * S * (GEMM_BM * GEMM_BN)) + mn * GEMM_THREADS + threadIdx.x;
    for (int s2 = 0; s2 < S; ++s2) v += p[(int64_t)s2 * (GEMM_BM * GEMM_BN / 4)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = it * GEMM_BM + (wave >> 2) * 128 + m * 16 + (lane >> 4) * 4;
    const int j = (jt0 + jt) * GEMM_BN + (wave & 3) * 64 + n * 16 + (lane & 15);
    if (i < g.I && j < g.J) gemm_store4<EPI>(e, g.I, i, j, v);
}

static hipError_t set_lds_attr_impl();
static hipError_t set_lds_attr_once() {
    // thread-safe: the Python lock is per engine, two engines may first-launch from two threads
    static std::once_flag once;
    static hipError_t st = hipSuccess;
    std::call_once(once, [] { st = set_lds_attr_impl(); });
    return st;
}
static hipError_t set_lds_attr_impl() {
    hipError_t st = hipSuccess;
#define SET_ATTR(K)                                                                              \
    if (st == hipSuccess)                                                                        \
        st = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                 GEMM_LDS_BYTES);
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_GELU_BF16>)
    SET_ATTR(gemm_bf16_kernel<TVC_EPI_RESID_F32>)
    SET_ATTR(gemm_splitk_partial_kernel)
#undef SET_ATTR
#define SET_ATTR(K)                                                                              \
    if (st == hipSuccess)                                                                        \
        st = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                 RING_LDS_BYTES + 4096);
    SET_ATTR(gemm_ring_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_ring_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_ring_kernel<TVC_EPI_GELU_BF16>)
    SET_ATTR(gemm_ring_kernel<TVC_EPI_RESID_F32>)
    SET_ATTR(gemm_ring4_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_ring4_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_ring4_kernel<TVC_EPI_GELU_BF16>)
    SET_ATTR(gemm_ring4_split_kernel)
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
    if (L.planes < 1 || L.planes > GEMM_MAX_PLANES) return hipErrorInvalidValue;
    for (int p = 0; p < GEMM_MAX_PLANES; ++p) { g.a_plane_off[p] = L.a_plane_off[p]; g.b_plane_off[p] = L.b_plane_off[p]; }
    GemmEpilogue e;
    e.bias = L.bias; e.out = L.out; e.ldo = L.ldo;
    const int nIt = (L.I + GEMM_BM - 1) / GEMM_BM, nJt = (L.J + GEMM_BN - 1) / GEMM_BN;
    const dim3 block(GEMM_THREADS);
    // variant: 0 = one tile per workgroup (gemm_core.hpp), 1 = persistent ring (gemm_ring.hpp / gemm_ring4.hpp).
    // The ring needs enough tiles to keep 256 persistent workgroups busy.
    static const int forced = [] { const char* v = getenv("TVC_GEMM_VARIANT"); return v ? atoi(v) : -1; }();
    const int ntiles = nIt * nJt;
    const bool deep = (int64_t)L.K * L.planes >= 256;      // >= 8 ring stages per tile
    // (the ring's deep LDS-DMA pipeline also beats the one-tile kernel's wait-per-K-tile loop on launches of fewer
    // tiles than CUs, one tile per workgroup: TVC_GEMM_RING_MIN_TILES, default 8 -- 64 until the latent-diffusion model's
    // guidance halves went to two streams: a half's 16 x 16-level launches are 30-60 tiles, and 8 / 16 / 32 / 64 gave
    // 17.2 / 17.1 / 17.15 / 16.85 images/s with the same bits, profiles/r04_sd_ring_min_tiles.log)
    static const int ring_min_env = [] { const char* v = getenv("TVC_GEMM_RING_MIN_TILES"); return v ? atoi(v) : 8; }();
    // launches that may split K over idle CUs (`splitk_small`: the input-gradient path at small batches) keep the old bound:
    // below 64 tiles they split
    const int ring_min = (L.splitk_small && ring_min_env < 64) ? 64 : ring_min_env;
    // Launches that opted into split-K (`splitk_small`: the latent-diffusion model, whose results carry no batch-position
    // invariance to protect) and have 64..128 tiles of a DEEP K (a 3 x 3 convolution at 16 x 16 latents: 120 tiles x 180
    // K-tiles) also take the split-K kernels below instead of one tile per workgroup on half the chip.
    const int nk64_all = (int)((int64_t)L.K * L.planes / GEMM_BK);
    const bool auto_split = L.splitk_fixed == 0;       // splitk_fixed: the caller fixed the K split (kernels.hpp)
    if (L.splitk_fixed >= 2) {
        int S = L.splitk_fixed;
        if (S > nk64_all) S = nk64_all;
        if (!L.splitk_ws || (size_t)ntiles * S * GEMM_BM * GEMM_BN * 4 > L.splitk_ws_bytes) return hipErrorInvalidValue;
        // slices of equal depth on whole-row operands run in the ring kernel (virtual tiles); anything else in the one-tile loop
        static const bool ring_split = [] { const char* v = getenv("TVC_GEMM_RING_SPLIT"); return !v || atoi(v) != 0; }();
        const int vt = ntiles * S;
        if (ring_split && nk64_all % S == 0 && nk64_all / S >= 4 && (L.I % GEMM_BM == 0 || L.a_rows_padded) &&
            (L.J % GEMM_BN == 0 || L.b_rows_padded) && L.lda % 64 == 0 && L.ldb % 64 == 0) {
            const dim3 rgrid(vt >= 256 ? 256 : (vt + 7) / 8 * 8);
            hipLaunchKernelGGL(gemm_ring4_split_kernel, rgrid, block, R3_LDS_BYTES + 4096, stream, g, L.splitk_ws, nIt, nJt, S);
        } else {
            hipLaunchKernelGGL(gemm_splitk_partial_kernel, dim3(vt), block, GEMM_LDS_BYTES, stream, g, L.splitk_ws, nIt, 0, S);
        }
        switch (L.epilogue) {
            case TVC_EPI_F32:
                hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_F32>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                break;
            case TVC_EPI_BF16:
                hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_BF16>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                break;
            case TVC_EPI_GELU_BF16:
                hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_GELU_BF16>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                break;
            case TVC_EPI_RESID_F32:
                hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_RESID_F32>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                break;
            default:
                return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const bool mid_split = auto_split && L.splitk_small && L.splitk_ws && forced < 0 && ntiles >= ring_min && ntiles <= 128 && nk64_all >= 32 &&
                           (size_t)ntiles * (256 / ntiles) * GEMM_BM * GEMM_BN * 4 <= L.splitk_ws_bytes;
    const bool ring = deep && !mid_split && (forced >= 0 ? (forced >= 1 && ntiles >= 8) : (ntiles >= ring_min));
    if (ring) {
        // ---- split-K tail: whole rounds to the ring kernel, the left-over tile columns split over K
        // Opt-in (TVC_GEMM_SPLITK_TAIL=1): it shortens the GEMM launches themselves by 1.4 % (89.7 vs 91.0 ms
        // per step) but the step does not get faster when the two towers run on two streams - the other
        // tower's kernels already fill the idle CUs of a last round - and it adds two launches per GEMM.
        static const bool tail_on = [] { const char* v = getenv("TVC_GEMM_SPLITK_TAIL"); return v && atoi(v) != 0; }();
        // (`splitk_small` launches take it too, with whole tile COLUMNS for the ring kernel even when the tile rows do not
        // divide 256: 96 token columns x 3 feature rows = 288 tiles -> 85 columns = 255 tiles in one round + 33 tiles split 7-way)
        const int full_tiles = ntiles / 256 * 256;
        const int jt_full = full_tiles / nIt;                 // tile columns the ring kernel keeps
        const int left = ntiles - jt_full * nIt;              // tiles of the left-over columns
        const int nk64 = (int)((int64_t)L.K * L.planes / GEMM_BK);
        int S = left > 0 ? 256 / left : 0;
        if (S > nk64 / 4) S = nk64 / 4;
        if (S > 16) S = 16;
        const bool whole_rounds = (jt_full * nIt) % 256 == 0;
        const bool tail = auto_split && forced < 0 && L.splitk_ws && jt_full >= 1 && left >= 1 && S >= 2 &&
                          ((tail_on && whole_rounds && left <= 64) || (L.splitk_small && left <= 128)) &&
                          (size_t)left * S * GEMM_BM * GEMM_BN * 4 <= L.splitk_ws_bytes;
        if (tail) {
            GemmLaunch M2 = L;
            M2.J = jt_full * GEMM_BN;
            M2.splitk_ws = nullptr;
            hipError_t st2 = launch_gemm_bf16(M2, stream);
            if (st2 != hipSuccess) return st2;
            hipLaunchKernelGGL(gemm_splitk_partial_kernel, dim3(left * S), block, GEMM_LDS_BYTES, stream, g,
                               L.splitk_ws, nIt, jt_full, S);
            switch (L.epilogue) {
                case TVC_EPI_F32:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_F32>, dim3(left * 32), block, 0, stream, g, e, L.splitk_ws, nIt, jt_full, S);
                    break;
                case TVC_EPI_BF16:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_BF16>, dim3(left * 32), block, 0, stream, g, e, L.splitk_ws, nIt, jt_full, S);
                    break;
                case TVC_EPI_GELU_BF16:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_GELU_BF16>, dim3(left * 32), block, 0, stream, g, e, L.splitk_ws, nIt, jt_full, S);
                    break;
                case TVC_EPI_RESID_F32:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_RESID_F32>, dim3(left * 32), block, 0, stream, g, e, L.splitk_ws, nIt, jt_full, S);
                    break;
                default:
                    return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        const dim3 rgrid(ntiles >= 256 ? 256 : (ntiles + 7) / 8 * 8);   // a workgroup without a tile returns at once
        // ring form: 4 (barrier-staggered ping-pong in 16-MFMA phases over 64-deep whole-line K-tiles) where its
        // preconditions hold, else 1 (32-deep stages, clamped rows: any shape); TVC_GEMM_RING_FORM=1 forces form 1.
        // (Forms 2 and 3 and the four-wave gemm_solo kernel of rounds 1-2 measured no faster than these two and were
        // removed in round 3; DESIGN.md 4.1 keeps their numbers.)
        static const int ring_form = [] { const char* v = getenv("TVC_GEMM_RING_FORM"); return v ? atoi(v) : 4; }();
        // form 4 reads whole rows without clamping: out-feature rows must fill whole tiles, B must have readable
        // rows up to the next multiple of 256 (J % 256 == 0, or a padded workspace: GemmLaunch::b_rows_padded), and
        // the row pitches must be multiples of 128 bytes (its source swizzle flips address bit 6)
        if (ring_form == 4 && L.epilogue != TVC_EPI_RESID_F32 && (L.I % GEMM_BM == 0 || L.a_rows_padded) &&
            (L.J % GEMM_BN == 0 || L.b_rows_padded) && L.lda % 64 == 0 && L.ldb % 64 == 0) {
            switch (L.epilogue) {
                case TVC_EPI_F32:
                    hipLaunchKernelGGL(gemm_ring4_kernel<TVC_EPI_F32>, rgrid, block, R3_LDS_BYTES + 4096, stream, g, e, nIt, nJt);
                    break;
                case TVC_EPI_BF16:
                    hipLaunchKernelGGL(gemm_ring4_kernel<TVC_EPI_BF16>, rgrid, block, R3_LDS_BYTES + 4096, stream, g, e, nIt, nJt);
                    break;
                case TVC_EPI_GELU_BF16:
                    hipLaunchKernelGGL(gemm_ring4_kernel<TVC_EPI_GELU_BF16>, rgrid, block, R3_LDS_BYTES + 4096, stream, g, e, nIt, nJt);
                    break;
                default:
                    return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        switch (L.epilogue) {
            case TVC_EPI_F32:
                hipLaunchKernelGGL(gemm_ring_kernel<TVC_EPI_F32>, rgrid, block, RING_LDS_BYTES + 2048, stream, g, e, nIt, nJt);
                break;
            case TVC_EPI_BF16:
                hipLaunchKernelGGL(gemm_ring_kernel<TVC_EPI_BF16>, rgrid, block, RING_LDS_BYTES + 2048, stream, g, e, nIt, nJt);
                break;
            case TVC_EPI_GELU_BF16:
                hipLaunchKernelGGL(gemm_ring_kernel<TVC_EPI_GELU_BF16>, rgrid, block, RING_LDS_BYTES + 2048, stream, g, e, nIt, nJt);
                break;
            case TVC_EPI_RESID_F32:
                hipLaunchKernelGGL(gemm_ring_kernel<TVC_EPI_RESID_F32>, rgrid, block, RING_LDS_BYTES + 2048, stream, g, e, nIt, nJt);
                break;
            default:
                return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    // ---- few tiles, deep K (small batches): one tile per workgroup would leave most of the 256 CUs idle
    // and run K serially (FC2 at one image: 8 workgroups x 64 K-steps).  Split K over S workgroups per
    // tile (fp32 partial tiles + the finish kernel of the split-K tail above).
    {
        // Opt-in (TVC_GEMM_SPLITK_SMALL=1, a latency mode: one query 7.5 -> 5.3 ms): the fp32 sums are taken
        // in a different order than in the one-pass kernels, so a query's embedding would depend (in the
        // last bits) on the size of the batch it arrives in; by default it does not
        // (tests/test_gpu_configs.py::test_config1_scale_properties, batch-split invariance).
        static const bool small_on = [] { const char* v = getenv("TVC_GEMM_SPLITK_SMALL"); return v && atoi(v) != 0; }();
        const int nk64 = (int)((int64_t)L.K * L.planes / GEMM_BK);
        int S = ntiles > 0 ? 256 / ntiles : 0;
        if (S > nk64 / 2) S = nk64 / 2;
        if (S > 16) S = 16;
        if (auto_split && (small_on || L.splitk_small) && forced < 0 && L.splitk_ws && S >= 2 &&
            (size_t)ntiles * S * GEMM_BM * GEMM_BN * 4 <= L.splitk_ws_bytes) {
            hipLaunchKernelGGL(gemm_splitk_partial_kernel, dim3(ntiles * S), block, GEMM_LDS_BYTES, stream, g,
                               L.splitk_ws, nIt, 0, S);
            switch (L.epilogue) {
                case TVC_EPI_F32:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_F32>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                    break;
                case TVC_EPI_BF16:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_BF16>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                    break;
                case TVC_EPI_GELU_BF16:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_GELU_BF16>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                    break;
                case TVC_EPI_RESID_F32:
                    hipLaunchKernelGGL(gemm_splitk_finish_kernel<TVC_EPI_RESID_F32>, dim3(ntiles * 32), block, 0, stream, g, e, L.splitk_ws, nIt, 0, S);
                    break;
                default:
                    return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
    }
    const dim3 grid(nIt * nJt);
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
