// GEMM main loop, fourth / fifth form (barrier-staggered ping-pong in phases of 16 MFMAs over 64-deep whole-line K-tiles), as a
// reusable stream: a workgroup multiplies a SEQUENCE of 256 x 256 output tiles, the LDS-DMA ring running across tile
// boundaries.  The schedule, its hazards and its measurements are described at gemm_ring4_kernel (gemm.hip), which keeps
// its own copy of this loop with the GEMM's tile-end extras (bias slices, store credit); this header serves the callers
// whose tile end is not a store of the tile (the bank search: bank.hip).
#pragma once
#include "gemm_core.hpp"
#include <type_traits>

#define R3_SLOT_BYTES (2 * GEMM_TILE_BYTES)        // 64 KiB: one K-tile of both operands
#define R3_LDS_BYTES (2 * R3_SLOT_BYTES)           // 128 KiB

// Two 1-KiB pieces (8 rows x 128 B each) of one operand, LDS destination in M0.
// M0 is NOT saved and restored here (two scalar instructions fewer in every load segment, +1-2 %): hipcc generates no
// M0 user of its own in the kernels that use this (gfx950 LDS instructions do not read M0; the only other M0 user, the
// GEMM's bias piece, goes through glds16_asm, which saves and restores).  hipcc ignores a clobber of the reserved register
// -- and says so -- hence the local pragma; tests/test_gpu_kernels.py::test_ring_forms_are_bit_identical and the bank
// search parity tests guard the result.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_rows2_asm(const void* base, uint32_t v0, uint32_t v1, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "v"(v0), "v"(v1), "s"(base), "s"(lds)
                 : "memory", "scc", "m0");
}
#pragma clang diagnostic pop

// ring4_stream(g, smem, ntiles, origin, tile_end)
//   origin(n, i0, j0): first A row / first B row of the stream's n-th tile (n < ntiles; uniform).
//   tile_end(n, acc):  consumes the tile's sums (acc[m][n'] = 16 x 16 sub-tile m of the wave's 128 A rows x sub-tile n' of
//                      its 64 B rows, MFMA accumulator layout); called by wave group 1 BEFORE and by group 0 AFTER the
//                      tile's last barrier, so the two groups' tile ends overlap.  It must not touch the ring's LDS
//                      (smem[0 .. R3_LDS_BYTES)), must contain no barrier, and must leave no vector-memory LOAD pending
//                      (stores may stay in flight: they only make the counted waits stricter).
//   Preconditions (the caller's to check on the host): every tile's 256 A rows and 256 B rows are readable without
//   clamping, row pitches are multiples of 128 bytes, K is a multiple of 64 per plane.
//   All 512 threads of the workgroup call it together; it ends with the LDS-DMA queue drained and a barrier.
template <class Origin, class TileEnd>
__device__ __forceinline__ void ring4_stream(const GemmOperands& g, char* smem, int ntiles, Origin origin, TileEnd tile_end) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kpp = g.ksteps_per_plane;
    const int nkt = g.planes * kpp;
    if (ntiles <= 0) return;

    const uint32_t smem_lds = lds_addr(smem);
    struct Cursor { int tile, p, kk; const char* abase; const char* bbase; const char* ap; const char* bp; };
    auto cur_tile = [&](Cursor& c, int n) __attribute__((always_inline)) {
        int i0, j0;
        origin(n, i0, j0);
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
                // past the last tile the stream re-stages the LAST tile's rows (nobody reads them; the loop then needs
                // no end-of-stream cases)
                cur_tile(c, ++c.tile < ntiles ? c.tile : ntiles - 1);
            } else {
                c.ap = c.abase + (int64_t)g.a_plane_off[c.p] * 2;
                c.bp = c.bbase + (int64_t)g.b_plane_off[c.p] * 2;
            }
        }
    };
    Cursor is{0, 0, 0, nullptr, nullptr, nullptr, nullptr};
    cur_tile(is, 0);
    const uint32_t pitchA = (uint32_t)(g.lda * 2), pitchB = (uint32_t)(g.ldb * 2);              // bytes per row
    const int rl = lane >> 3;
    const uint32_t swz = (uint32_t)(((lane & 7) ^ ((rl >> 1) & 7)) * 16);
    const uint32_t vA0 = (uint32_t)rl * pitchA + swz, vA1 = (vA0 ^ 64u) + 8u * pitchA;
    const uint32_t vB0 = (uint32_t)rl * pitchB + swz, vB1 = (vB0 ^ 64u) + 8u * pitchB;
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
    typedef const __attribute__((address_space(3))) bf16x8_t* lds_frag_p;
    const uint32_t sw_rd = (uint32_t)((((lane >> 4) ^ ((lane >> 1) & 7)) * 16));
    const uint32_t a_rd0 = smem_lds + (wm * 128 + (lane & 15)) * 128 + sw_rd, a_rd1 = a_rd0 ^ 64u;
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
#define RING4S_MFMA(A_, B_, QA_, QB_)                                                                             \
    {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                          \
            _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                         \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                     \
                    acc[(QA_) * 4 + m][(QB_) * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                  \
                        A_[m][ks], B_[n][ks], acc[(QA_) * 4 + m][(QB_) * 2 + n], 0, 0, 0);                        \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    }
#define RING4S_BARRIER() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#define RING4S_WAIT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // Form 5 (round 4; the schedule of gemm_ring4_kernel, gemm.hip): every unit is issued one phase earlier than in form 4
    // (A units one phase after their only read -- staged and read by the same wave group --, B units two) and ONE counted wait
    // per K-tile, in p3: vmcnt(6) retires all four units of K-tile t+1 and leaves Aq0 / Bq0 / Bq1 of K-tile t+2 in flight.
    // ---- prologue: K-tile 0 whole, Aq0 / Bq0 / Bq1 of K-tile 1; K-tile 0 landed and published
    issue_unit(U_A0{}); issue_unit(U_B0{}); issue_unit(U_B1{}); issue_unit(U_A1{});
    issue_unit(U_A0{}); issue_unit(U_B0{}); issue_unit(U_B1{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    RING4S_BARRIER()
    if (wm == 1) RING4S_BARRIER()            // group 1 runs one barrier behind group 0 from here on

    auto ktile = [&](int t) __attribute__((always_inline)) {
        // p0: read Aq0, Bq0 of K-tile t; stage Aq1 of K-tile t+1
        issue_unit(U_A1{});
        load_B(t, 0, B0f);
        load_A(t, 0, A0f);
        RING4S_BARRIER()
        RING4S_MFMA(A0f, B0f, 0, 0)
        RING4S_BARRIER()
        // p1: read Bq1; stage Aq0 of K-tile t+2
        issue_unit(U_A0{});
        load_B(t, 1, B1f);
        RING4S_BARRIER()
        RING4S_MFMA(A0f, B1f, 0, 1)
        RING4S_BARRIER()
        // p2: read Aq1; stage Bq0 of K-tile t+2
        issue_unit(U_B0{});
        load_A(t, 1, A1f);
        RING4S_BARRIER()
        RING4S_MFMA(A1f, B1f, 1, 1)
        RING4S_BARRIER()
        // p3: stage Bq1 of K-tile t+2; all of K-tile t+1 must have landed (read from the next p0 on)
        issue_unit(U_B1{});
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        RING4S_BARRIER()
        RING4S_MFMA(A1f, B0f, 1, 0)
        // (the phase's second barrier is placed by the tile loop)
    };
    int t = 0;
#pragma unroll 1
    for (int ct = 0; ct < ntiles; ++ct) {
        if ((nkt & 1) == 0) {
            // two K-tiles per loop iteration: the LDS buffer of a K-tile is a compile-time constant (a tile starts on buffer 0)
            ktile(0); RING4S_BARRIER() ktile(1);
#pragma unroll 1
            for (int k = 2; k < nkt; k += 2) {
                RING4S_BARRIER()
                ktile(0);
                RING4S_BARRIER()
                ktile(1);
            }
            t += nkt;
        } else {
        ktile(t);
        ++t;
#pragma unroll 1
        for (int k = 1; k < nkt; ++k, ++t) {
            RING4S_BARRIER()
            ktile(t);
        }
        }
        if (wm == 1) tile_end(ct, acc);
        RING4S_BARRIER()
        if (wm == 0) tile_end(ct, acc);
        gemm_zero_acc(acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stream's overrun stages must have landed before the LDS is reused
    if (wm == 0) RING4S_BARRIER()            // pairs with group 1's last barrier
    __syncthreads();
#undef RING4S_MFMA
#undef RING4S_BARRIER
#undef RING4S_WAIT8
}
