// Dense bf16 GEMM with fused epilogues for the CLIP towers (K1/K2) and the
// bank-search pre-pass.  See gemm_core.hpp for the tiling.
#include "gemm_epilogue.hpp"
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
                                 RING_LDS_BYTES + 2048);
    SET_ATTR(gemm_ring_kernel<TVC_EPI_F32>)
    SET_ATTR(gemm_ring_kernel<TVC_EPI_BF16>)
    SET_ATTR(gemm_ring_kernel<TVC_EPI_GELU_BF16>)
    SET_ATTR(gemm_ring_kernel<TVC_EPI_RESID_F32>)
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
    const bool ring = deep && (forced >= 0 ? (forced >= 1 && ntiles >= 8) : (ntiles >= 512));
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
        const dim3 rgrid(ntiles >= 256 ? 256 : (ntiles / 8) * 8);
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
