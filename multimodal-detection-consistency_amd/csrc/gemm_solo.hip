// Four-wave (one wave per SIMD) persistent bf16 GEMM: see the kernel comment.  Own translation
// unit: the 256-accumulator kernel takes a while to compile.
#include "gemm_epilogue.hpp"
#include <mutex>

// ---------------------------------------------------------------------------
// Four-wave variant: ONE wave per SIMD owns a 128 x 128 quarter of the 256 x 256 tile
// (8 x 8 MFMA tiles, 256 fp32 accumulators per lane, the 512-register budget).  Same LDS ring
// image as gemm_ring_kernel; one instruction stream per SIMD carries everything, so the
// LDS-DMA pieces of stage S+4 and the fragment reads of stage S+1 are interleaved by hand
// between the 64 MFMAs of stage S (one piece + two ds_read_b128 per 8 MFMAs): no arbitration
// between waves, 2/3 of the LDS read bytes of the eight-wave form, one barrier per 64 MFMAs.
//   RAW  fragments of stage S+1 are read in iteration S; every wave retired its pieces of stage
//        S+1 (counted vmcnt) before the barrier that ended iteration S-1.
//   WAR  stage S+4 goes to slot S%4, whose fragments were read in iteration S-1 (lgkmcnt(0)
//        before that barrier).
// ---------------------------------------------------------------------------
#define SOLO_THREADS 256
#define SOLO_LDS_BYTES (RING_LDS_BYTES + 2048 + 1024)   // ring + two bias slots + dump KiB
// MFMA with the accumulator pinned to AGPRs ("+a"): with 256 accumulators per lane hipcc otherwise
// keeps them in VGPRs it does not have and shuttles every tile through v_accvgpr moves.
__device__ __forceinline__ void mfma_agpr(f32x4_t& c, const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// pin a wave-uniform pointer to SGPRs (inline-asm "s" operands)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
typedef f32x4_t solo_acc_t[8][8];

// One accumulator tile out of the AGPRs, pinned in program order: left to itself hipcc copies ~150
// accumulators to VGPRs at the top of the epilogue and spills the next stage's fragments for them.
__device__ __forceinline__ f32x4_t acc_read(const f32x4_t& c) {
    f32x4_t v;
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(c[0]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(c[1]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x2) : "a"(c[2]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x3) : "a"(c[3]));
    v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3;
    return v;
}

// Epilogue of one 256 x 256 tile of the four-wave kernel: ONE branch-free code path (the launcher
// passes whole tiles only: I % 256 == 0, J % 256 == 0, ldo % 8 == 0; a ragged remainder of token
// rows goes to the eight-wave kernels), processed one 16-token sub-tile at a time so that only a
// few accumulators are out of the AGPRs at once.
template <int EPI>
__device__ __forceinline__ void solo_tile_epilogue(const f32x4_t (&acc)[8][8], const GemmEpilogue& e, int J,
                                                   int i0, int j0, int wm, int wn, int lane,
                                                   const char* bias_lds) {
    const int il = wm * 128 + (lane >> 4) * 4;          // tile-local first out-feature of this lane
    auto bias_of = [&](int m) -> f32x4_t {
        if (!e.bias) return f32x4_t{0.f, 0.f, 0.f, 0.f};
        return *(const __attribute__((address_space(3))) f32x4_t*)(
            (const __attribute__((address_space(3))) char*)bias_lds + (il + m * 16) * 4);
    };
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int j = j0 + wn * 128 + n * 16 + (lane & 15);
        {
            static_assert(EPI == TVC_EPI_BF16 || EPI == TVC_EPI_GELU_BF16, "bf16 epilogues only");
            {
                // 16-byte bf16 stores: see gemm_tile_epilogue (v_permlane16_swap pairs the sub-tiles m, m+1)
                const int gq = lane >> 4;
                uint16_t* p = (uint16_t*)e.out + (int64_t)j * e.ldo + i0 + wm * 128 + (gq & 1) * 16 + (gq >> 1) * 8;
#pragma unroll
                for (int mp = 0; mp < 4; ++mp) {
                    f32x4_t v0 = acc_read(acc[2 * mp][n]) + bias_of(2 * mp);
                    f32x4_t v1 = acc_read(acc[2 * mp + 1][n]) + bias_of(2 * mp + 1);
                    if (EPI == TVC_EPI_GELU_BF16) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) { v0[t] = quick_gelu(v0[t]); v1[t] = quick_gelu(v1[t]); }
                    }
                    const auto r0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
                    u32x4_t o;
                    o[0] = r0[0]; o[1] = r1[0]; o[2] = r0[1]; o[3] = r1[1];
                    // (a non-temporal store is 1-3 % faster for this kernel alone and 3.5 % slower for the
                    // layer chain: the next kernel reads these rows back out of L2 / Infinity Cache)
                    *(u32x4_t*)(p + mp * 32) = o;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int EPI>
__global__ __launch_bounds__(SOLO_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_solo_kernel(GemmOperands g, GemmEpilogue e, int nIt, int nJt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int kpp = g.ksteps_per_plane * (GEMM_BK / RING_BK);
    const int nk = g.planes * kpp;                                 // stages per tile (even)

    RingSchedule sch;
    sch.init(nIt * nJt);
    const int my_tiles = sch.count();
    const int total = my_tiles * nk;
    if (total == 0) return;

    const uint32_t smem_lds = lds_addr(smem);
    // ---- issue side: a wave stages rows wave*64 .. +63 of both operands (4 + 4 one-KiB pieces)
    int is_tile = 0, is_p = 0, is_kk = 0, is_n = 0;
    const char* is_abase; const char* is_bbase;
    const char* st_a = nullptr; const char* st_b = nullptr;        // bases of the stage being issued
    uint32_t st_slot = 0;
    uint32_t st_dump = 1;                       // 1: real stage (pieces 1 KiB apart), 0: dump (all pieces on one KiB)
    uint32_t va[4], vb[4];
    auto issue_tile = [&](int lin) __attribute__((always_inline)) {
        const int jt = lin / nIt;
        const int i0 = (lin - jt * nIt) * GEMM_BM, j0 = jt * GEMM_BN;
        is_abase = (const char*)(g.A + (int64_t)i0 * g.lda);
        is_bbase = (const char*)(g.B + (int64_t)j0 * g.ldb);
        const int c = (lane & 3) ^ (3 * ((lane >> 5) & 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 64 + i * 16 + (lane >> 2);
            int ra = r, rb = r;
            if (i0 + ra >= g.I) ra = g.I - 1 - i0;
            if (j0 + rb >= g.J) rb = g.J - 1 - j0;
            va[i] = (uint32_t)ra * (uint32_t)(g.lda * 2) + c * 16;
            vb[i] = (uint32_t)rb * (uint32_t)(g.ldb * 2) + c * 16;
        }
    };
    issue_tile(sch.tile(0));
    // begin_stage fixes the (scalar) addresses of the next stage to issue; its 8 pieces then go
    // out one by one.  A tile switch is deferred to the next begin_stage, so va / vb stay valid
    // for every piece of the stage they were computed for.
    bool pending_switch = false;
    int nxt_apo = g.a_plane_off[0], nxt_bpo = g.b_plane_off[0];
    auto begin_stage = [&]() __attribute__((always_inline)) {
        if (pending_switch) {
            pending_switch = false;
            if (is_tile < my_tiles) issue_tile(sch.tile(is_tile));
        }
        const bool live = is_n < total;
        // past the end of the stream the 8 pieces still go out (uniform vmcnt accounting, no
        // branches in the MFMA stream) but re-read the last stage into a dump area
        st_dump = live ? 1u : 0u;
        st_slot = live ? smem_lds + (is_n & (RING_SLOTS - 1)) * RING_SLOT_BYTES + wave * (64 * 64)
                       : smem_lds + RING_LDS_BYTES + 2048;
        ++is_n;
        if (live) {
            st_a = is_abase + (int64_t)(nxt_apo + is_kk * RING_BK) * 2;
            st_b = is_bbase + (int64_t)(nxt_bpo + is_kk * RING_BK) * 2;
            if (++is_kk == kpp) {
                is_kk = 0;
                if (++is_p == g.planes) { is_p = 0; ++is_tile; pending_switch = true; }
                // the plane offsets are re-read only when the plane changes: an indexed kernarg read
                // per stage would put a scalar-memory round trip + lgkmcnt(0) at the head of every
                // stage of the only wave on the SIMD
                nxt_apo = g.a_plane_off[is_p]; nxt_bpo = g.b_plane_off[is_p];
            }
        }
    };
    auto piece = [&](int gidx) __attribute__((always_inline)) {
        if (gidx < 4) glds16_asm(uniform_ptr(st_a), va[gidx], __builtin_amdgcn_readfirstlane(st_slot + st_dump * (gidx * 1024)));
        else glds16_asm(uniform_ptr(st_b), vb[gidx - 4],
                        __builtin_amdgcn_readfirstlane(st_slot + st_dump * (RING_HALF_BYTES + (gidx - 4) * 1024)));
    };

    const int pos = ((lane >> 4) ^ (3 * ((lane >> 3) & 1))) * 16;
    const int a_off = (wm * 128 + (lane & 15)) * 64 + pos;
    const int b_off = RING_HALF_BYTES + (wn * 128 + (lane & 15)) * 64 + pos;

    // ---- prologue: stages 0..3 in flight, fragments of stage 0 in registers
    for (int s2 = 0; s2 < 4; ++s2) {
        begin_stage();
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) piece(gi);
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // stages 0 and 1 (own pieces)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    solo_acc_t acc;
    bf16x8_t a0[8], b0[8], a1[8], b1[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) a0[m] = *(const bf16x8_t*)(smem + a_off + m * 1024);
#pragma unroll
    for (int n = 0; n < 8; ++n) b0[n] = *(const bf16x8_t*)(smem + b_off + n * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int credit = 0;
    int ct = 0, cks = 0;
    constexpr bool st16 = true;
    auto tile_origin = [&](int t, int& i0, int& j0) __attribute__((always_inline)) {
        const int lin = sch.tile(t);
        const int jt = lin / nIt;
        i0 = (lin - jt * nIt) * GEMM_BM; j0 = jt * GEMM_BN;
    };

    // one stage: MFMAs on (ac, bc); fragments of the next stage into (an, bn); pieces of stage S+4
    auto mma_stage = [&](int S, bf16x8_t (&ac)[8], bf16x8_t (&bc)[8], bf16x8_t (&an)[8], bf16x8_t (&bn)[8]) {
        begin_stage();
        if (cks == 0 && wave == 0 && e.bias) {
            int i0, j0;
            tile_origin(ct, i0, j0);
            if (i0 + GEMM_BM <= g.I)
                glds16_asm(uniform_ptr((const char*)(e.bias + i0)), lane * 16,
                           __builtin_amdgcn_readfirstlane(smem_lds + RING_LDS_BYTES + (ct & 1) * 1024));
        }
        const char* nslot = smem + ((S + 1) & (RING_SLOTS - 1)) * RING_SLOT_BYTES;
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            piece(gi);
            // (past the last stage this reads a stale slot: harmless, never multiplied)
            an[gi] = *(const bf16x8_t*)(nslot + a_off + gi * 1024);
            bn[gi] = *(const bf16x8_t*)(nslot + b_off + gi * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 8; ++n) mfma_agpr(acc[gi][n], ac[gi], bc[n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        ++cks;
    };
    // nk is even: a tile can only end after an odd stage, so only the second half carries the epilogue
    auto tile_end = [&]() __attribute__((always_inline)) {
        if (cks == nk) {
            int i0, j0;
            tile_origin(ct, i0, j0);
            // the asm MFMAs are invisible to hipcc's hazard recognizer: cover the matrix-pipe latency
            // before the first v_accvgpr_read of the epilogue by hand
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            solo_tile_epilogue<EPI>(acc, e, g.J, i0, j0, wm, wn, lane, smem + RING_LDS_BYTES + (ct & 1) * 1024);
            // 32 sixteen-byte stores sit behind the loads in the queue (whole token tiles only)
            credit = (st16 && j0 + GEMM_BN <= g.J) ? 3 : 0;
            cks = 0; ++ct;
        }
    };
    auto retire = [&](int S) __attribute__((always_inline)) {
        // stage S+2 (own pieces) landed; S+3, S+4 (real or dump) stay in flight
        if (credit > 0) asm volatile("s_waitcnt vmcnt(48) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        if (credit > 0) --credit;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // The accumulators are (re)defined at the top of the tile loop and only read by the epilogue:
    // with a conditional re-zeroing inside one flat stage loop hipcc no longer keeps the 256
    // accumulators in place in the AGPRs and spills.
    for (int t = 0; t < my_tiles; ++t) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nk; ks += 2) {
            const int S = t * nk + ks;
            mma_stage(S, a0, b0, a1, b1);
            retire(S);
            mma_stage(S + 1, a1, b1, a0, b0);
            tile_end();
            retire(S + 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // dump pieces land before the LDS is released
}


hipError_t launch_gemm_solo(const GemmOperands& g, const GemmEpilogue& e, int epilogue, int nIt, int nJt,
                            hipStream_t stream) {
    static std::once_flag attr_once;
    static hipError_t attr_st = hipSuccess;
    std::call_once(attr_once, [] {
        attr_st = hipFuncSetAttribute((const void*)gemm_solo_kernel<TVC_EPI_BF16>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLO_LDS_BYTES);
        if (attr_st == hipSuccess)
            attr_st = hipFuncSetAttribute((const void*)gemm_solo_kernel<TVC_EPI_GELU_BF16>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, SOLO_LDS_BYTES);
    });
    if (attr_st != hipSuccess) return attr_st;
    const int ntiles = nIt * nJt;
    const dim3 rgrid(ntiles >= 256 ? 256 : (ntiles / 8) * 8);
    const dim3 sblock(SOLO_THREADS);
    switch (epilogue) {
        case TVC_EPI_BF16:
            hipLaunchKernelGGL(gemm_solo_kernel<TVC_EPI_BF16>, rgrid, sblock, SOLO_LDS_BYTES, stream, g, e, nIt, nJt);
            break;
        case TVC_EPI_GELU_BF16:
            hipLaunchKernelGGL(gemm_solo_kernel<TVC_EPI_GELU_BF16>, rgrid, sblock, SOLO_LDS_BYTES, stream, g, e, nIt, nJt);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
