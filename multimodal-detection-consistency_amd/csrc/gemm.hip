// Dense bf16 GEMM with fused epilogues for the CLIP towers (K1/K2) and the
// bank-search pre-pass.  See gemm_core.hpp for the tiling.
#include "gemm_epilogue.hpp"
#include "gemm_ring4.hpp"
#include <cstdlib>
#include <mutex>
#include <type_traits>

// s_setprio levels of the MFMA phases of the two wave groups of the ring kernel (see gemm_ring_kernel)
#ifndef TVC_PRIO_G0
#define TVC_PRIO_G0 1
#endif
#ifndef TVC_PRIO_G1
#define TVC_PRIO_G1 2
#endif

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
#ifdef TVC_RING_STAMPS
// diagnostic build only (scripts/ring_stamps.py): per-wave shader-clock totals of the loop phases
__device__ unsigned long long ring_stamps[256 * 8 * 8];
extern "C" int tvc_debug_ring_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ring_stamps), sizeof(ring_stamps));
}
__device__ unsigned long long ring_trace[4 * 512];
// form 4: per wave {first K-tile of a tile, other K-tiles, epilogue issue, barrier after / before the epilogue}
__device__ unsigned long long ring4_tile_stamps[256 * 8 * 4];
extern "C" int tvc_debug_ring4_tile_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ring4_tile_stamps), sizeof(ring4_tile_stamps));
}
extern "C" int tvc_debug_ring_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ring_trace), sizeof(ring_trace));
}
#define STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; \
                   if ((i) == 5 && S < 512 && lane == 0 && (blockIdx.x == 8 || blockIdx.x == 100) && (wave & 3) == 0) \
                       ring_trace[((blockIdx.x == 100) * 2 + (wave >> 2)) * 512 + S] = t_; }
#else
#define STAMP(i)
#endif
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

#ifdef TVC_RING_STAGGER
    // XCD x starts x/8 of a tile period late: the tile epilogues (a 4 MiB dirty burst per XCD) of
    // different XCDs no longer hit HBM at the same time
    for (int i = (blockIdx.x & 7) * nk * TVC_RING_STAGGER; i > 0; --i) __builtin_amdgcn_s_sleep(1);
#endif
    const uint32_t smem_lds = lds_addr(smem);
    // ---- issue side: scalar tile bases + per-lane 32-bit offsets (recomputed per tile only)
    int is_tile = 0, is_p = 0, is_kk = 0, is_n = 0;
    const char* is_abase; const char* is_bbase;
    uint32_t va[2], vb[2];
    auto issue_tile = [&](int lin) {
#ifdef TVC_RING_ALIAS
        lin %= TVC_RING_ALIAS;       // diagnostic build only: every workgroup re-reads the first few tiles' operands (L2-resident)
#endif
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
    int credit = 0;              // upcoming waits that still see a fast epilogue's stores in the queue
    const bool st16 = (EPI == TVC_EPI_BF16 || EPI == TVC_EPI_GELU_BF16) && (e.ldo & 7) == 0;
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
    auto finish_stage = [&](int credit_after) {
        if (++cks == nk) {
            int i0, j0;
            tile_origin(ct, i0, j0);
            gemm_tile_epilogue<EPI, true>(acc, g, e, i0, j0, wm, wn, lane, smem + RING_LDS_BYTES + (ct & 1) * 1024);
            gemm_zero_acc(acc);
            const bool fast = (i0 + GEMM_BM <= g.I) && (j0 + GEMM_BN <= g.J) && ((e.ldo & 3) == 0);
            credit = fast ? credit_after : 0;
            cks = 0; ++ct;
        }
    };
    auto retire_and_barrier = [&](int S) {
        const int n_out = (S + 3 < total ? S + 3 : total - 1) - S;           // stages in flight beyond S
        // a fast epilogue left 32 (16 for the 16-byte bf16 form) stores in the queue behind the loads
        if (n_out >= 3 && credit > 0 && st16) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
        else if (n_out >= 3 && credit > 0) asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
        else if (n_out >= 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (n_out == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (credit > 0) --credit;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

#ifdef TVC_RING_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    if (gid == 0) {
        int it = 0, iks = 0;     // tile / k-stage of the stage being loaded (== multiplied) this interval
        for (int S = 0; S < total; ++S) {
            STAMP(0)
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
            STAMP(1)
            load_frags(S);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(2)
            mfma_stage(std::integral_constant<int, TVC_PRIO_G0>{});
            STAMP(3)
            finish_stage(3);
            STAMP(4)
            retire_and_barrier(S);
            STAMP(5)
        }
    } else {
        for (int S = 0; S < total; ++S) {
            STAMP(0)
            if (S > 0) { mfma_stage(std::integral_constant<int, TVC_PRIO_G1>{}); STAMP(3) finish_stage(2); STAMP(4) }
            if (S + 3 < total) issue();
            STAMP(1)
            load_frags(S);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(2)
            retire_and_barrier(S);
            STAMP(5)
        }
        mfma_stage(std::integral_constant<int, TVC_PRIO_G1>{});
        finish_stage(0);
    }
#ifdef TVC_RING_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) ring_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_acc[i];
#endif
}


// ---------------------------------------------------------------------------
// Ring main loop, second form: fragments PREFETCHED one stage ahead (two named register sets).
//
// In-kernel stamps of gemm_ring_kernel (scripts/ring_stamps.py) put one stage of one wave at
//   loop bookkeeping 196 + LDS-DMA issue 352 + ds_read + wait 231 + MFMA 665 + barrier wait 355 = 1873 clk (FC2)
// for 2 x 512 clk of matrix-pipe work per SIMD: the interval is the SERIAL sum of a wave's own phases (each wave
// runs load -> wait -> multiply in order), not the matrix pipe.  Here the fragments of stage S+1 are read while
// stage S multiplies (the kernel had 53 free VGPRs: a second a[8] / b[4] set costs 48), so the ds_read latency
// leaves the serial chain, the LDS-DMA of stage S+4 is issued first and the bookkeeping is incremental.
//   ring protocol (4 slots, stage S in slot S % 4):
//     interval S:  issue stage S+4 into slot S % 4 (stage S's fragments were read in interval S-1, every wave's
//                  lgkmcnt(0) precedes the barrier that ended it);  read the fragments of stage S+1;  multiply
//                  stage S;  retire: stage S+2 must have landed before interval S+1 reads it, i.e. at most the
//                  pieces of stages S+3, S+4 (+ a tile epilogue's stores) stay in flight: vmcnt(8 / 24 / 40).
//   The two waves of a SIMD still run the phases in opposite order (group 0: issue, read, multiply; group 1:
//   multiply, issue, read) so that one feeds the matrix pipe while the other sits in LDS-DMA issue.
// ---------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_ring2_kernel(GemmOperands g, GemmEpilogue e, int nIt, int nJt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kpp = g.ksteps_per_plane * (GEMM_BK / RING_BK);     // 32-deep stages per plane
    const int nk = g.planes * kpp;                                 // stages per tile (even: K % 64 == 0)

    RingSchedule sch;
    sch.init(nIt * nJt);
    const int my_tiles = sch.count();
    const int total = my_tiles * nk;                               // stages in this workgroup's stream (even)
    if (total == 0) return;

    const uint32_t smem_lds = lds_addr(smem);
    // ---- issue side: running scalar pointers of the next stage, per-lane 32-bit offsets per tile
    int is_tile = 0, is_p = 0, is_kk = 0, is_n = 0;
    const char* is_abase; const char* is_bbase;       // tile bases
    const char* is_ap; const char* is_bp;             // next stage of the current plane
    uint32_t va[2], vb[2];
    auto issue_tile = [&](int lin) __attribute__((always_inline)) {
#ifdef TVC_RING_ALIAS
        lin %= TVC_RING_ALIAS;
#endif
        const int jt = lin / nIt;
        const int i0 = (lin - jt * nIt) * GEMM_BM, j0 = jt * GEMM_BN;
        is_abase = (const char*)(g.A + (int64_t)i0 * g.lda);
        is_bbase = (const char*)(g.B + (int64_t)j0 * g.ldb);
        is_ap = is_abase + (int64_t)g.a_plane_off[0] * 2;
        is_bp = is_bbase + (int64_t)g.b_plane_off[0] * 2;
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
    auto issue = [&]() __attribute__((always_inline)) {
        const uint32_t slot = smem_lds + (is_n & (RING_SLOTS - 1)) * RING_SLOT_BYTES + wave * (32 * 64);
        glds16x4_asm(is_ap, va[0], va[1], slot, is_bp, vb[0], vb[1], slot + RING_HALF_BYTES);
        ++is_n;
        is_ap += RING_BK * 2; is_bp += RING_BK * 2;
        if (++is_kk == kpp) {
            is_kk = 0;
            if (++is_p == g.planes) {
                is_p = 0;
                if (++is_tile < my_tiles) issue_tile(sch.tile(is_tile));
            } else {
                is_ap = is_abase + (int64_t)g.a_plane_off[is_p] * 2;
                is_bp = is_bbase + (int64_t)g.b_plane_off[is_p] * 2;
            }
        }
    };

    const int pos = ((lane >> 4) ^ (3 * ((lane >> 3) & 1))) * 16;
    const int a_off = (wm * 128 + (lane & 15)) * 64 + pos;
    const int b_off = RING_HALF_BYTES + (wn * 64 + (lane & 15)) * 64 + pos;
    const int gid = wave >> 2;

    gemm_acc_t acc;
    gemm_zero_acc(acc);
    bf16x8_t a0[8], b0[4], a1[8], b1[4];
    int credit = 0;
    const bool st16 = (EPI == TVC_EPI_BF16 || EPI == TVC_EPI_GELU_BF16) && (e.ldo & 7) == 0;
    int ct = 0, cks = 0;

    auto tile_origin = [&](int t, int& i0, int& j0) __attribute__((always_inline)) {
        const int lin = sch.tile(t);
        const int jt = lin / nIt;
        i0 = (lin - jt * nIt) * GEMM_BM; j0 = jt * GEMM_BN;
    };
    auto load_frags = [&](int S, bf16x8_t (&a)[8], bf16x8_t (&b)[4]) __attribute__((always_inline)) {
        const char* slot = smem + (S & (RING_SLOTS - 1)) * RING_SLOT_BYTES;
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = *(const bf16x8_t*)(slot + a_off + m * 1024);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *(const bf16x8_t*)(slot + b_off + n * 1024);
    };
    // MFMA phase priorities: the trailing group's phase runs at the higher priority (see gemm_ring_kernel)
#define RING2_MFMA(A_, B_)                                                                            \
    {                                                                                                 \
        if (gid == 0) __builtin_amdgcn_s_setprio(TVC_PRIO_G0); else __builtin_amdgcn_s_setprio(TVC_PRIO_G1); \
        _Pragma("unroll") for (int m = 0; m < 8; ++m)                                                 \
            _Pragma("unroll") for (int n = 0; n < 4; ++n)                                             \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[m], B_[n], acc[m][n], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                \
    }
    // stage S+2 must have landed (it is read in interval S+1); younger stages issued so far: up to S+4
    auto retire_and_barrier = [&](int S) __attribute__((always_inline)) {
        const int last = (S + 4 < total ? S + 4 : total - 1);
        const int n_out = last - (S + 2);                                     // stages that may stay in flight
        // lgkmcnt(0) through the BUILTIN: hipcc's waitcnt pass then knows that the fragment set read in this
        // interval is complete and does not guard the next interval's MFMAs with waits for the (younger)
        // prefetch reads; the LDS-DMA pieces are invisible to it, their counted vmcnt stays inline asm
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (n_out >= 2 && credit > 0 && st16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (n_out >= 2 && credit > 0) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (n_out >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n_out == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (credit > 0) --credit;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: stages 0..3 in flight, stage 0's fragments in set 0, stage 1 landed
    for (int s = 0; s < 4 && s < total; ++s) issue();
    if (total >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // stage 0 (own pieces)
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                       // total == 2
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(0, a0, b0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (total >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // stage 1
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int it = 0, iks = 0;         // tile / k-stage of the stage multiplied in the current interval (bias staging)
    auto stage_bias = [&]() __attribute__((always_inline)) {
        if (iks == 0 && wave == 0 && e.bias) {
            // the tile's 256 bias values -> one of TWO alternating LDS slots (the partner group may still be in
            // the previous tile's epilogue, reading the other slot); retired by this wave's counted waits long
            // before the epilogue (>= 8 stages follow)
            int i0, j0;
            tile_origin(it, i0, j0);
            if (i0 + GEMM_BM <= g.I) glds16_asm(e.bias + i0, lane * 16, smem_lds + RING_LDS_BYTES + (it & 1) * 1024);
        }
        if (++iks == nk) { iks = 0; ++it; }
    };
    // Two intervals per iteration (static register sets).  A tile has an even number of stages, so only the
    // SECOND interval of a pair can end a tile: one epilogue site, shared by both wave groups.
    for (int S = 0; S < total; S += 2) {
        // ---- interval S: multiply set 0, prefetch stage S+1 into set 1
        if (gid == 0) {
            if (S + 4 < total) issue();
            stage_bias();
            load_frags(S + 1, a1, b1);
        }
        RING2_MFMA(a0, b0)
        ++cks;
        if (gid != 0) {
            if (S + 4 < total) issue();
            load_frags(S + 1, a1, b1);
        }
        retire_and_barrier(S);
        // ---- interval S+1: multiply set 1, prefetch stage S+2 into set 0
        if (gid == 0) {
            if (S + 5 < total) issue();
            stage_bias();
            if (S + 2 < total) load_frags(S + 2, a0, b0);
        }
        RING2_MFMA(a1, b1)
        if (++cks == nk) {
            int i0, j0;
            tile_origin(ct, i0, j0);
            gemm_tile_epilogue<EPI, true>(acc, g, e, i0, j0, wm, wn, lane, smem + RING_LDS_BYTES + (ct & 1) * 1024);
            gemm_zero_acc(acc);
            const bool fast = (i0 + GEMM_BM <= g.I) && (j0 + GEMM_BN <= g.J) && ((e.ldo & 3) == 0);
            credit = fast ? (gid == 0 ? 3 : 2) : 0;
            cks = 0; ++ct;
        }
        if (gid != 0) {
            if (S + 5 < total) issue();
            if (S + 2 < total) load_frags(S + 2, a0, b0);
        }
        retire_and_barrier(S + 1);
    }
#undef RING2_MFMA
}


// ---------------------------------------------------------------------------
// Ring main loop, third form: 64-deep K-tiles, so that every LDS-DMA piece reads WHOLE 128-byte lines.
//
// scripts/ubench/dma_rate.hip (LDS-DMA alone, L2-resident source, 8 waves x 4 pieces per stage): pieces of
// 16 rows x 64 B (what a 32-deep stage reads) fill LDS at 29-32 B/clk/CU, pieces of 8 rows x 128 B at 58-60:
// the line is fetched whole either way and half of it is thrown away.  A 32-deep stage of the forms above
// therefore costs the load path twice its bytes, and that -- not the matrix pipe, which PMC counters show 45 %
// busy -- sets their stage interval (prefetching the fragments, form 2, changed nothing).
//   LDS: 2 slots x (A [256][64] 32 KiB + B [256][64] 32 KiB); images as gemm_core.hpp (128-byte rows, 16-byte
//   chunk c of row r at chunk c ^ ((r >> 1) & 7): conflict-free ds_read_b128, swizzle on the SOURCE address).
//   A K-tile is multiplied in two 32-deep steps; fragments are prefetched one step ahead (two register sets).
//     interval (t, 0): read fragments (t, 1);  multiply (t, 0);  retire K-tile t+1 (vmcnt: only a tile
//                      epilogue's stores may stay in flight);  barrier  -> slot t % 2 is free, t+1 is visible
//     interval (t, 1): issue K-tile t+2 into slot t % 2;  read fragments (t+1, 0);  multiply (t, 1);  tile
//                      epilogue after the last K-tile of a tile;  no wait, no barrier needed for the ring
//   K-tile t+2 is in flight during (t, 1) and (t+1, 0): two intervals for 64 KiB.
// ---------------------------------------------------------------------------

// four 1-KiB pieces (8 rows x 128 B each) of one operand: one M0 save / restore
__device__ __forceinline__ void glds16_rows4_asm(const void* base, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3,
                                                 uint32_t lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "s"(lds)
                 : "memory", "scc");
}

#ifndef TVC_R3_ODD_BARRIER
#define TVC_R3_ODD_BARRIER 1
#endif
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_ring3_kernel(GemmOperands g, GemmEpilogue e, int nIt, int nJt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int gid = wave >> 2;
    const int kpp = g.ksteps_per_plane;                            // 64-deep K-tiles per plane
    const int nkt = g.planes * kpp;                                // K-tiles per output tile

    RingSchedule sch;
    sch.init(nIt * nJt);
    const int my_tiles = sch.count();
    const int T = my_tiles * nkt;                                  // K-tiles in this workgroup's stream
    if (T == 0) return;

    const uint32_t smem_lds = lds_addr(smem);
    // ---- issue side.  No row clamping here: the launcher guarantees I % 256 == 0 and that B has readable rows up
    // to the next multiple of 256 (rows >= J are garbage that only reaches output columns that are never stored).
    // Piece i of a wave covers rows wave*32 + 8 i + (lane >> 3), 16-byte chunk (lane & 7) ^ ((row >> 1) & 7):
    // the swizzle term alternates between two values with the parity of i, the row term is a scalar step, so two
    // lane offsets per operand stay live across the loop and the four addresses are formed when issuing.
    struct Cursor { int tile, p, kk, n; const char* abase; const char* bbase; const char* ap; const char* bp; };
    const uint32_t a_rs = (uint32_t)(g.lda * 16), b_rs = (uint32_t)(g.ldb * 16);     // 8 rows, bytes
    // Lane-constant offsets are RECOMPUTED where they are used (from an opaque copy of the lane index, so that
    // hipcc neither hoists nor keeps them): the loop runs at the 256-register limit, and ONE spilled value makes
    // the compiler guard the loop with s_waitcnt vmcnt(0) for its scratch reloads -- draining the LDS-DMA ring.
    auto opaque_lane = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto cur_tile = [&](Cursor& c, int lin) __attribute__((always_inline)) {
#ifdef TVC_RING_ALIAS
        lin %= TVC_RING_ALIAS;
#endif
        const int jt = lin / nIt;
        const int i0 = (lin - jt * nIt) * GEMM_BM, j0 = jt * GEMM_BN;
        c.abase = (const char*)(g.A + (int64_t)i0 * g.lda);
        c.bbase = (const char*)(g.B + (int64_t)j0 * g.ldb);
        c.ap = c.abase + (int64_t)g.a_plane_off[0] * 2;
        c.bp = c.bbase + (int64_t)g.b_plane_off[0] * 2;
    };
    auto cur_advance = [&](Cursor& c) __attribute__((always_inline)) {
        ++c.n;
        c.ap += GEMM_BK * 2; c.bp += GEMM_BK * 2;
        if (++c.kk == kpp) {
            c.kk = 0;
            if (++c.p == g.planes) {
                c.p = 0;
                if (++c.tile < my_tiles) cur_tile(c, sch.tile(c.tile));
            } else {
                c.ap = c.abase + (int64_t)g.a_plane_off[c.p] * 2;
                c.bp = c.bbase + (int64_t)g.b_plane_off[c.p] * 2;
            }
        }
    };
    Cursor is{0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr};
    cur_tile(is, sch.tile(0));
    auto issue = [&]() __attribute__((always_inline)) {
        const uint32_t slot = smem_lds + (is.n & 1) * R3_SLOT_BYTES + wave * (32 * 128);
        const int l = opaque_lane();
        const int r0 = wave * 32 + (l >> 3);
        const int c0 = (l & 7) ^ ((r0 >> 1) & 7);
        const uint32_t va_e = (uint32_t)r0 * (uint32_t)(g.lda * 2) + c0 * 16;
        const uint32_t vb_e = (uint32_t)r0 * (uint32_t)(g.ldb * 2) + c0 * 16;
#ifndef TVC_R3_NO_DMA            // (ablation builds only)
        glds16_rows4_asm(is.ap, va_e, (va_e ^ 64) + a_rs, va_e + 2 * a_rs, (va_e ^ 64) + 3 * a_rs, slot);
        glds16_rows4_asm(is.bp, vb_e, (vb_e ^ 64) + b_rs, vb_e + 2 * b_rs, (vb_e ^ 64) + 3 * b_rs, slot + GEMM_TILE_BYTES);
#endif
        cur_advance(is);
    };

    // k-sub-step 1 = chunk + 4 = the same address with bit 6 flipped (the swizzle XORs the chunk's low bits only)

    gemm_acc_t acc;
    gemm_zero_acc(acc);
    bf16x8_t a0[8], b0[4], a1[8], b1[4];
    bool credit = false;         // a fast epilogue's stores are the youngest entries of the queue at the next retire
    const bool st16 = (EPI == TVC_EPI_BF16 || EPI == TVC_EPI_GELU_BF16) && (e.ldo & 7) == 0;
    int ct = 0, ckt = 0;         // output tile / K-tile inside it of the K-tile being multiplied

    auto tile_origin = [&](int t, int& i0, int& j0) __attribute__((always_inline)) {
        const int lin = sch.tile(t);
        const int jt = lin / nIt;
        i0 = (lin - jt * nIt) * GEMM_BM; j0 = jt * GEMM_BN;
    };
    auto load_frags = [&](int t, int ks, bf16x8_t (&a)[8], bf16x8_t (&b)[4]) __attribute__((always_inline)) {
        // ONE address register per operand, the sub-tiles are immediate offsets
        const int l = opaque_lane();
        const int sw0 = ((0 + (l >> 4)) ^ ((l >> 1) & 7)) * 16;
        const uint32_t a_rd = smem_lds + (wm * 128 + (l & 15)) * 128 + sw0;
        const uint32_t b_rd = smem_lds + GEMM_TILE_BYTES + (wn * 64 + (l & 15)) * 128 + sw0;
        const uint32_t so = (t & 1) * R3_SLOT_BYTES;
        const __attribute__((address_space(3))) char* pa = (const __attribute__((address_space(3))) char*)(uintptr_t)((a_rd ^ (ks * 64)) + so);
        const __attribute__((address_space(3))) char* pb = (const __attribute__((address_space(3))) char*)(uintptr_t)((b_rd ^ (ks * 64)) + so);
#ifdef TVC_R3_NO_DSREAD          // (ablation builds only: fragments keep whatever they hold, kept alive)
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" : "+v"(a[m]));
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(b[n]));
        return;
#endif
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = *(const __attribute__((address_space(3))) bf16x8_t*)(pa + m * 2048);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *(const __attribute__((address_space(3))) bf16x8_t*)(pb + n * 2048);
    };
#ifdef TVC_R3_NO_MFMA
#define RING3_MFMA(A_, B_) { _Pragma("unroll") for (int m = 0; m < 8; ++m) asm volatile("" :: "v"(A_[m])); _Pragma("unroll") for (int n = 0; n < 4; ++n) asm volatile("" :: "v"(B_[n])); }
#else
#define RING3_MFMA(A_, B_)                                                                            \
    {                                                                                                 \
        if (gid == 0) __builtin_amdgcn_s_setprio(TVC_PRIO_G0); else __builtin_amdgcn_s_setprio(TVC_PRIO_G1); \
        _Pragma("unroll") for (int m = 0; m < 8; ++m)                                                 \
            _Pragma("unroll") for (int n = 0; n < 4; ++n)                                             \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[m], B_[n], acc[m][n], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                \
    }
#endif

    // ---- prologue: K-tiles 0 and 1 in flight, fragments (0, 0) in set 0
    issue();
    if (T > 1) issue();
    if (T > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(0, 0, a0, b0);
    __builtin_amdgcn_s_waitcnt(0xC07F);

    for (int t = 0; t < T; ++t) {
        // ================= interval (t, 0): multiply set 0, prefetch (t, 1) into set 1
        if (gid == 0) {
            if (ckt == 0 && wave == 0 && e.bias) {
                // the tile's 256 bias values -> one of two alternating 1-KiB LDS slots (the other group may still
                // be in the previous tile's epilogue); landed long before this tile's epilogue
                int i0, j0;
                tile_origin(ct, i0, j0);
                if (i0 + GEMM_BM <= g.I) glds16_asm(e.bias + i0, lane * 16, smem_lds + R3_LDS_BYTES + (ct & 1) * 1024);
            }
            load_frags(t, 1, a1, b1);
        }
        RING3_MFMA(a0, b0)
        if (gid != 0) load_frags(t, 1, a1, b1);
        // retire: K-tile t+1 (issued in interval (t-1, 1)) must have landed; only a tile epilogue's stores and this
        // wave's bias piece are younger
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (credit && st16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (credit) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        credit = false;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ================= interval (t, 1): multiply set 1, issue K-tile t+2, prefetch (t+1, 0) into set 0
        // (at a tile's last K-tile the prefetch waits until the epilogue is through: its temporaries and a
        // second live fragment set do not fit the 256-register budget together)
        const bool tile_end = (ckt + 1 == nkt);
        if (gid == 0) {
            if (t + 2 < T) issue();
            if (!tile_end && t + 1 < T) load_frags(t + 1, 0, a0, b0);
        }
        RING3_MFMA(a1, b1)
        if (gid != 0) {
            if (t + 2 < T) issue();
            if (!tile_end && t + 1 < T) load_frags(t + 1, 0, a0, b0);
        }
        if (tile_end) {
            int i0, j0;
            tile_origin(ct, i0, j0);
            // an opaque copy of the lane index: hipcc would otherwise hoist the epilogue's lane-dependent 64-bit
            // address parts out of the K loop, keep ~10 registers live across it and SPILL them -- and a scratch
            // reload anywhere in the loop makes its waitcnt pass guard the loop header with vmcnt(0), which
            // drains the LDS-DMA ring every interval
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            gemm_tile_epilogue<EPI, true>(acc, g, e, i0, j0, wm, wn, lane_e, smem + R3_LDS_BYTES + (ct & 1) * 1024);
            gemm_zero_acc(acc);
            credit = (i0 + GEMM_BM <= g.I) && (j0 + GEMM_BN <= g.J) && ((e.ldo & 3) == 0);
            ckt = 0; ++ct;
            if (t + 1 < T) load_frags(t + 1, 0, a0, b0);
        } else {
            ++ckt;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
#if TVC_R3_ODD_BARRIER
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#endif
    }
#undef RING3_MFMA
}


// ---------------------------------------------------------------------------
// Ring main loop, fourth form: barrier-staggered ping-pong in phases of 16 MFMAs (after the "256^2 8-phase"
// recipe of /opt/skills/guides/cdna_hip_programming.md section 5, rebuilt here for a persistent tile stream).
//
//   * A K-tile (64 deep) is multiplied in FOUR phases, one quadrant (64 out-features x 32 tokens x K 64 = 16 MFMAs)
//     of the wave's 128 x 64 sub-tile each: (Aq0,Bq0) (Aq0,Bq1) (Aq1,Bq1) (Aq1,Bq0).
//   * A phase is  {LDS reads + ONE half-tile of LDS-DMA}  barrier  {16 MFMAs}  barrier.  The wave group wm = 1 runs one
//     barrier behind wm = 0, so on every SIMD one wave is in its MFMA segment while its partner is in its load segment:
//     the matrix pipe never waits for a ds_read, and the LDS-DMA stream is spread evenly (16 KiB per phase).
//   * LDS: 2 K-tile buffers x (A [256][64] + B [256][64]), the image of form 3.  The STAGING unit is not a contiguous
//     half but the 128 rows one quadrant reads: Aq = rows {wm*128 + q*64 ..+63}, Bq = rows {wn*64 + q*32 ..+31} over all
//     waves (16 KiB = 16 pieces, two per wave).  A unit is read in exactly ONE phase (p0: Aq0 + Bq0, p1: Bq1, p2: Aq1;
//     the fragments then stay in registers, 96 VGPRs), so it is free again two phases later: unit X of K-tile t+2 is
//     issued 2-3 phases after unit X of K-tile t was read, and FOUR phases (two quadrants' worth of both groups'
//     MFMAs) before the counted wait that precedes its first read -- s_waitcnt vmcnt(8) in p3 (Aq0, Bq0 of t+1), p0
//     (Bq1 of t) and p1 (Aq1 of t): four units = 64 KiB always in flight, issued 16 KiB per phase.
//   * write-after-read: restaged >= 2 phases after the only read.  read-after-write: the counted wait sits before a
//     phase's first barrier, the read is in the next phase (one barrier more than the wait, because the two groups
//     are a barrier apart).
//   * Tile ends.  The K-tile body exists twice: a tile's FIRST K-tile is a copy whose counted waits let the previous
//     epilogue's stores pass (compile-time vmcnt(8 + stores)); the steady-state copy carries no end-of-tile, bias or
//     credit test at all.  Group 1 runs its epilogue before the tile's last barrier and group 0 after it, so the two
//     overlap; the epilogue reads the lane's bias vectors once (gemm_tile_epilogue<.., BIAS_REGS>).
//   The fp32 sums are taken in the same order as in the other forms: results are bit-identical to them.
// ---------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_ring4_kernel(GemmOperands g, GemmEpilogue e, int nIt, int nJt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kpp = g.ksteps_per_plane;
    const int nkt = g.planes * kpp;

    RingSchedule sch;
    sch.init(nIt * nJt);
    const int my_tiles = sch.count();
    const int T = my_tiles * nkt;                                  // K-tiles in this workgroup's stream
    if (T == 0) return;

    const uint32_t smem_lds = lds_addr(smem);
    struct Cursor { int tile, p, kk; const char* abase; const char* bbase; const char* ap; const char* bp; };
    auto opaque_lane = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto cur_tile = [&](Cursor& c, int lin) __attribute__((always_inline)) {
        const int jt = lin / nIt;
        const int i0 = (lin - jt * nIt) * GEMM_BM, j0 = jt * GEMM_BN;
        c.abase = (const char*)(g.A + (int64_t)i0 * g.lda);
        c.bbase = (const char*)(g.B + (int64_t)j0 * g.ldb);
        c.ap = c.abase + (int64_t)g.a_plane_off[0] * 2;
        c.bp = c.bbase + (int64_t)g.b_plane_off[0] * 2;
    };
    auto cur_advance = [&](Cursor& c) __attribute__((always_inline)) {
        c.ap += GEMM_BK * 2; c.bp += GEMM_BK * 2;
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
    Cursor is{0, 0, 0, nullptr, nullptr, nullptr, nullptr};
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
#ifndef TVC_R4_NO_DMA            // (ablation builds only)
        if (isA) glds16_rows2_asm(base, vA0, vA1, dst); else glds16_rows2_asm(base, vB0, vB1, dst);
#endif
        if (kind == 3) { buf_issue = (buf_issue == smem_lds) ? smem_lds + R3_SLOT_BYTES : smem_lds; cur_advance(is); }
    };
    using U_A0 = std::integral_constant<int, 0>; using U_B0 = std::integral_constant<int, 1>;
    using U_B1 = std::integral_constant<int, 2>; using U_A1 = std::integral_constant<int, 3>;

    gemm_acc_t acc;
    gemm_zero_acc(acc);
    bf16x8_t A0f[4][2], A1f[4][2], B0f[2][2], B1f[2][2];
    int ct = 0;

    auto tile_origin = [&](int t, int& i0, int& j0) __attribute__((always_inline)) {
        const int lin = sch.tile(t);
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
#ifdef TVC_R4_NO_MFMA            // (ablation builds only: the fragments are consumed, nothing is multiplied)
#define RING4_MFMA(A_, B_, QA_, QB_) { _Pragma("unroll") for (int m = 0; m < 4; ++m) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) asm volatile("" :: "v"(A_[m][ks])); _Pragma("unroll") for (int n = 0; n < 2; ++n) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) asm volatile("" :: "v"(B_[n][ks])); }
#else
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
#endif
#define RING4_BARRIER() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

#define RING4_WAIT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // ---- prologue: K-tile 0 whole, Aq0 / Bq0 of K-tile 1; Aq0 / Bq0 of K-tile 0 landed and published
    issue_unit(U_A0{}); issue_unit(U_B0{}); issue_unit(U_B1{}); issue_unit(U_A1{});
    issue_unit(U_A0{}); issue_unit(U_B0{});
    RING4_WAIT8()
    RING4_BARRIER()
    if (wm == 1) RING4_BARRIER()            // group 1 runs one barrier behind group 0 from here on

    // One K-tile.  CREDIT: the number of store instructions the previous tile's epilogue left behind the ring's loads in
    // this wave's (in-order) vector memory queue: the three counted waits of a tile's first K-tile let them pass
    // (vmcnt(8 + CREDIT)) instead of draining them; by the next K-tile's first wait (>= 4 phases later) they are the
    // oldest entries.  It is compile-time: the steady-state body carries neither a compare nor a branch for it, nor for
    // the tile's bias slice (staged ahead of the previous tile's stores, see below).
    auto ktile = [&](int t, auto credit_c) __attribute__((always_inline)) {
        constexpr int CREDIT = decltype(credit_c)::value;
#define RING4_WAITC() { if (CREDIT == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
                        else if (CREDIT == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); \
                        else asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); }
        // ===== p0: read Aq0, Bq0 of K-tile t; stage Bq1 of K-tile t+1; Bq1 of K-tile t must have landed (read in p1)
        issue_unit(U_B1{});
        load_B(t, 0, B0f);
        load_A(t, 0, A0f);
        RING4_WAITC()
        RING4_BARRIER()
        RING4_MFMA(A0f, B0f, 0, 0)
        RING4_BARRIER()
        // ===== p1: read Bq1; stage Aq1 of K-tile t+1; Aq1 of K-tile t must have landed (read in p2)
        issue_unit(U_A1{});
        load_B(t, 1, B1f);
        RING4_WAITC()
        RING4_BARRIER()
        RING4_MFMA(A0f, B1f, 0, 1)
        RING4_BARRIER()
        // ===== p2: read Aq1; stage Aq0 of K-tile t+2
        issue_unit(U_A0{});
        load_A(t, 1, A1f);
        RING4_BARRIER()
        RING4_MFMA(A1f, B1f, 1, 1)
        RING4_BARRIER()
        // ===== p3: stage Bq0 of K-tile t+2; Aq0, Bq0 of K-tile t+1 must have landed (read in the next p0)
        issue_unit(U_B0{});
        RING4_WAITC()
        RING4_BARRIER()
        RING4_MFMA(A1f, B0f, 1, 0)
        // (the phase's second barrier is the caller's: at a tile end the two groups place their epilogues differently)
#undef RING4_WAITC
    };
    auto stage_bias = [&](int tile) __attribute__((always_inline)) {
        if (wave == 0 && e.bias) {
            int i0, j0;
            tile_origin(tile, i0, j0);
            if (i0 + GEMM_BM <= g.I) glds16_asm(e.bias + i0, lane * 16, smem_lds + R3_LDS_BYTES + (tile & 1) * 1024);
        }
    };
    stage_bias(0);
    using C0 = std::integral_constant<int, 0>;
    // stores of a fast epilogue per wave: 16 (16-byte bf16 pieces) or 32 (fp32, or bf16 rows not 16-byte aligned)
    constexpr int EPI_STORES = (EPI == TVC_EPI_BF16 || EPI == TVC_EPI_GELU_BF16) ? 16 : 32;
    using CE = std::integral_constant<int, EPI_STORES>;
    const bool credit_ok = (EPI_STORES == 32) ? ((e.ldo & 3) == 0) : ((e.ldo & 7) == 0);
    bool credit = false;
    int t = 0;
#ifdef TVC_RING_STAMPS
    unsigned long long ts_first = 0, ts_rest = 0, ts_epi = 0, ts_bar = 0, ts_last = __builtin_amdgcn_s_memtime();
#define R4T(acc_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - ts_last; ts_last = t_; }
#else
#define R4T(acc_)
#endif
#pragma unroll 1
    for (ct = 0; ct < my_tiles; ++ct) {
        if (credit) ktile(t, CE{}); else ktile(t, C0{});
        ++t;
        R4T(ts_first)
#pragma unroll 1
        for (int k = 1; k < nkt; ++k, ++t) {
            RING4_BARRIER()
            ktile(t, C0{});
        }
        R4T(ts_rest)
        // the NEXT tile's bias slice goes into the queue ahead of this tile's stores: with it there, wave 0's credited
        // waits ask for one OLDER load more, never for a store
        if (ct + 1 < my_tiles) stage_bias(ct + 1);
        int i0, j0;
        tile_origin(ct, i0, j0);
        // Tile end.  Group 1 runs its epilogue BEFORE the last phase's second barrier, group 0 after it: group 0 reaches
        // that barrier a slot earlier, so the two epilogues run side by side (one slot of bias / convert / store latency
        // per tile instead of two in a row).
        auto tile_end = [&]() __attribute__((always_inline)) {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
#if defined(TVC_R4_NO_EPI)            // (ablation builds only: the sums are consumed, nothing is stored)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" :: "v"(acc[m][n]));
#else
            gemm_tile_epilogue<EPI, true, 4, true>(acc, g, e, i0, j0, wm, wn, lane_e, smem + R3_LDS_BYTES + (ct & 1) * 1024);
#endif
        };
        if (wm == 1) { tile_end(); R4T(ts_epi) }
        RING4_BARRIER()
        R4T(ts_bar)
        if (wm == 0) { tile_end(); R4T(ts_epi) }
        gemm_zero_acc(acc);
        credit = credit_ok && (i0 + GEMM_BM <= g.I) && (j0 + GEMM_BN <= g.J);
    }
#ifdef TVC_RING_STAMPS
    if (lane == 0) {
        unsigned long long* o = ring4_tile_stamps + (blockIdx.x * 8 + wave) * 4;
        o[0] = ts_first; o[1] = ts_rest; o[2] = ts_epi; o[3] = ts_bar;
    }
#endif
#undef R4T
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stream's overrun stages must have landed before the LDS is given back
    if (wm == 0) RING4_BARRIER()            // pairs with group 1's last barrier
#undef RING4_MFMA
#undef RING4_BARRIER
#undef RING4_WAIT8
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
    SET_ATTR(gemm_ring3_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_ring3_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_ring3_kernel<TVC_EPI_GELU_BF16>)
    SET_ATTR(gemm_ring4_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_ring4_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_ring4_kernel<TVC_EPI_GELU_BF16>)
    SET_ATTR(gemm_ring2_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_ring2_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_ring2_kernel<TVC_EPI_GELU_BF16>)
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
    const dim3 block(GEMM_THREADS);
    // variant: 0 = one tile per workgroup (gemm_core.hpp), 1 = persistent ring (gemm_ring.hpp).
    // The ring needs enough tiles to keep 256 persistent workgroups busy.
    static const int forced = [] { const char* v = getenv("TVC_GEMM_VARIANT"); return v ? atoi(v) : -1; }();
    const int ntiles = nIt * nJt;
    const bool deep = (int64_t)L.K * L.planes >= 256;      // >= 8 ring stages per tile
    // (the ring's deep LDS-DMA pipeline also beats the one-tile kernel's wait-per-K-tile loop on launches of fewer
    // tiles than CUs, one tile per workgroup: TVC_GEMM_RING_MIN_TILES, default 64)
    static const int ring_min = [] { const char* v = getenv("TVC_GEMM_RING_MIN_TILES"); return v ? atoi(v) : 64; }();
    const bool ring = deep && (forced >= 0 ? (forced >= 1 && ntiles >= 8) : (ntiles >= ring_min));
    // the four-wave kernel takes whole tiles, bf16 outputs and an even stage count; a ragged
    // remainder of token rows is a second launch on the eight-wave kernels
    const bool solo_ok = (L.epilogue == TVC_EPI_BF16 || L.epilogue == TVC_EPI_GELU_BF16) && L.I % GEMM_BM == 0 &&
                         L.ldo % 8 == 0 && (((int64_t)L.K * L.planes / RING_BK) % 2 == 0) && !L.no_solo;
    const int Jf = L.J / GEMM_BN * GEMM_BN;
    const bool solo = deep && solo_ok && (forced >= 0 ? (forced == 2 && nIt * (Jf / GEMM_BN) >= 8)
                                                       : false);
    if (solo) {
        g.J = Jf;
        hipError_t st2 = launch_gemm_solo(g, e, L.epilogue, nIt, Jf / GEMM_BN, stream);
        if (st2 != hipSuccess || Jf == L.J) return st2;
        GemmLaunch R = L;
        R.B = L.B + (int64_t)Jf * L.ldb;
        R.out = (char*)L.out + (int64_t)Jf * L.ldo * 2;       // bf16 outputs
        R.J = L.J - Jf;
        R.no_solo = true;
        return launch_gemm_bf16(R, stream);
    }
    if (ring) {
        // ---- split-K tail: whole rounds to the ring kernel, the left-over tile columns split over K
        // Opt-in (TVC_GEMM_SPLITK_TAIL=1): it shortens the GEMM launches themselves by 1.4 % (89.7 vs 91.0 ms
        // per step) but the step does not get faster when the two towers run on two streams - the other
        // tower's kernels already fill the idle CUs of a last round - and it adds two launches per GEMM.
        static const bool tail_on = [] { const char* v = getenv("TVC_GEMM_SPLITK_TAIL"); return v && atoi(v) != 0; }();
        const int full_tiles = ntiles / 256 * 256;
        const int jt_full = full_tiles / nIt;                 // tile columns the ring kernel keeps
        const int left = ntiles - jt_full * nIt;              // tiles of the left-over columns
        const int nk64 = (int)((int64_t)L.K * L.planes / GEMM_BK);
        int S = left > 0 ? 256 / left : 0;
        if (S > nk64 / 4) S = nk64 / 4;
        if (S > 16) S = 16;
        const bool tail = tail_on && forced < 0 && L.splitk_ws && jt_full >= 1 && left >= 1 && left <= 64 && S >= 2 &&
                          (jt_full * nIt) % 256 + left > 0 && (jt_full * nIt) % 256 == 0 &&
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
        // preconditions hold (they are form 3's), else 1; TVC_GEMM_RING_FORM=1|2|3|4 forces one (experiments)
        static const int ring_form = [] { const char* v = getenv("TVC_GEMM_RING_FORM"); return v ? atoi(v) : 4; }();
        // form 3 reads whole rows without clamping: out-feature rows must fill whole tiles, B must have readable
        // rows up to the next multiple of 256 (J % 256 == 0, or a padded workspace: GemmLaunch::b_rows_padded), and
        // the row pitches must be multiples of 128 bytes (its source swizzle flips address bit 6)
        if (ring_form == 4 && L.epilogue != TVC_EPI_RESID_F32 && L.I % GEMM_BM == 0 &&
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
        if ((ring_form == 3 || ring_form == 4) && L.epilogue != TVC_EPI_RESID_F32 && L.I % GEMM_BM == 0 &&
            (L.J % GEMM_BN == 0 || L.b_rows_padded) && L.lda % 64 == 0 && L.ldb % 64 == 0) {
            switch (L.epilogue) {
                case TVC_EPI_F32:
                    hipLaunchKernelGGL(gemm_ring3_kernel<TVC_EPI_F32>, rgrid, block, R3_LDS_BYTES + 4096, stream, g, e, nIt, nJt);
                    break;
                case TVC_EPI_BF16:
                    hipLaunchKernelGGL(gemm_ring3_kernel<TVC_EPI_BF16>, rgrid, block, R3_LDS_BYTES + 4096, stream, g, e, nIt, nJt);
                    break;
                case TVC_EPI_GELU_BF16:
                    hipLaunchKernelGGL(gemm_ring3_kernel<TVC_EPI_GELU_BF16>, rgrid, block, R3_LDS_BYTES + 4096, stream, g, e, nIt, nJt);
                    break;
                default:
                    return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        if (ring_form == 2 && L.epilogue != TVC_EPI_RESID_F32) {       // (the read-modify-write epilogue spills in this form)
            switch (L.epilogue) {
                case TVC_EPI_F32:
                    hipLaunchKernelGGL(gemm_ring2_kernel<TVC_EPI_F32>, rgrid, block, RING_LDS_BYTES + 2048, stream, g, e, nIt, nJt);
                    break;
                case TVC_EPI_BF16:
                    hipLaunchKernelGGL(gemm_ring2_kernel<TVC_EPI_BF16>, rgrid, block, RING_LDS_BYTES + 2048, stream, g, e, nIt, nJt);
                    break;
                case TVC_EPI_GELU_BF16:
                    hipLaunchKernelGGL(gemm_ring2_kernel<TVC_EPI_GELU_BF16>, rgrid, block, RING_LDS_BYTES + 2048, stream, g, e, nIt, nJt);
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
        if (small_on && forced < 0 && L.splitk_ws && S >= 2 &&
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
