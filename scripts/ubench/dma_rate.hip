// Micro-benchmark (diagnostic, not product): LDS-DMA fill rate per CU for different piece shapes.
//   one 512-thread workgroup per CU, every wave issues PIECES 1-KiB global_load_lds_dwordx4 pieces per
//   "stage" into a 4-slot LDS ring with 3 stages in flight (counted vmcnt) and one barrier per stage --
//   the load side of the ring GEMM with nothing else in the loop.
//   shape 0: 16 rows x 64 B per piece (row pitch PITCH), the stage advances 64 B along the row   (ring GEMM, BK = 32)
//   shape 1:  8 rows x 128 B per piece, the stage advances 128 B                                   (BK = 64)
//   shape 2:  1 KiB contiguous per piece
// build: hipcc -O3 --offload-arch=gfx950 -o dma_rate dma_rate.hip ;  run: ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16_asm(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// GEMM-like sharing: per XCD (workgroups with equal blockIdx % 8) 32 workgroups = 4 A tiles x 8 B panels; a workgroup
// streams K over its A tile (256 rows) and its B panel (256 rows), 8 rows x 128 B pieces, 2 K-tiles in flight.
__global__ __launch_bounds__(512) void gemm_like_kernel(const char* __restrict__ src, int pitch, int ktiles, int rounds,
                                                        unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;          // j = 0..31 inside the XCD
    const size_t tile_bytes = (size_t)256 * pitch;
    uint32_t voff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { const int r = wave * 32 + p * 8 + (lane >> 3); voff[p] = (uint32_t)r * pitch + (lane & 7) * 16; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int n = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        // A tile (j % 4) is the same in every round (weights); B panel changes every round (token stream)
        const char* a = src + (size_t)(j & 3) * tile_bytes;
        const char* b = src + (size_t)(4 + ((rd * 8 + xcd) * 8 + (j >> 2))) * tile_bytes;
        for (int t = 0; t < ktiles; ++t, ++n) {
            const uint32_t dst = lds + (n & 1) * 65536 + wave * 4096;
#pragma unroll
            for (int p = 0; p < 4; ++p) glds16_asm(a + (size_t)t * 128, voff[p], dst + p * 1024);
#pragma unroll
            for (int p = 0; p < 4; ++p) glds16_asm(b + (size_t)t * 128, voff[p], dst + 32768 + p * 1024);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the previous K-tile has landed
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

static void run_gemm_like(const char* name, const char* src, int pitch, int rounds) {
    unsigned long long* clk;
    hipMalloc(&clk, 256 * 8);
    hipFuncSetAttribute((const void*)gemm_like_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int ktiles = pitch / 128;                                // K = pitch / 2 elements, 64-deep K-tiles
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(gemm_like_kernel, dim3(256), dim3(512), 131072, 0, src, pitch, ktiles, rounds, clk);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : h) mean += c; mean /= 256;
    const double bytes = (double)rounds * ktiles * 65536;
    printf("%-46s %8.3f ms  %6.1f GB/s/CU  %6.2f TB/s  %5.1f B/clk/CU  (%.0f clk/K-tile)\n", name, best, bytes / best / 1e6,
           bytes * 256 / best / 1e9, bytes / mean, mean / (rounds * ktiles));
    hipFree(clk);
}

template <int SHAPE, int PIECES>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, size_t bytes_per_wg, int pitch,
                                                   int stages, int alias, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int wg = alias ? (blockIdx.x % alias) : blockIdx.x;
    const char* base = src + (size_t)wg * bytes_per_wg;
    // per-lane offsets of this wave's PIECES pieces inside a stage
    uint32_t voff[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const int piece = wave * PIECES + p;                  // 0 .. 8*PIECES-1
        if (SHAPE == 0) { const int r = piece * 16 + (lane >> 2); voff[p] = (uint32_t)r * pitch + (lane & 3) * 16; }
        else if (SHAPE == 1) { const int r = piece * 8 + (lane >> 3); voff[p] = (uint32_t)r * pitch + (lane & 7) * 16; }
        else voff[p] = piece * 1024 + lane * 16;
    }
    const int adv = SHAPE == 0 ? 64 : SHAPE == 1 ? 128 : 8 * PIECES * 1024;     // bytes per stage along the stream
    const int slot_bytes = 8 * PIECES * 1024;
    auto issue = [&](int s) {
        const char* sb = base + (size_t)s * adv;
        const uint32_t dst = lds + (s & 3) * slot_bytes + wave * PIECES * 1024;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) glds16_asm(sb, voff[p], dst + p * 1024);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < 3 && s < stages; ++s) issue(s);
    for (int s = 0; s < stages; ++s) {
        if (s + 3 < stages) issue(s + 3);
        if (PIECES == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (PIECES == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int PIECES>
static void run(const char* name, const char* src, size_t bytes_per_wg, int pitch, int stages, int alias) {
    unsigned long long* clk;
    hipMalloc(&clk, 256 * 8);
    const size_t lds = 4 * 8 * PIECES * 1024;
    hipFuncSetAttribute((const void*)dma_kernel<SHAPE, PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((dma_kernel<SHAPE, PIECES>), dim3(256), dim3(512), lds, 0, src, bytes_per_wg, pitch, stages, alias, clk);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : h) mean += c; mean /= 256;
    const double bytes = (double)stages * 8 * PIECES * 1024;
    printf("%-46s %8.3f ms  %6.1f GB/s/CU  %6.2f TB/s  %5.1f B/clk/CU  (%.0f clk/stage)\n", name, best, bytes / best / 1e6,
           bytes * 256 / best / 1e9, bytes / mean, mean / stages);
    hipFree(clk);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gload16_asm(u32x4& dst, const void* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// Split delivery (candidate GEMM form 4): B (tokens) through a 4-slot x 32 KiB LDS ring by LDS-DMA, 3 K-tiles in flight;
// A (weights) straight into REGISTERS in MFMA operand layout (wave w owns rows w*32 .. w*32+31 of the 256-row tile: no row
// is needed by two waves): 4 x global_load_dwordx4 per wave and 64-deep K-tile (2 sub-tiles of 16 rows x the two 64-B
// halves of a line, issued back to back), 3 K-tiles in flight = 48 VGPRs.  One barrier per K-tile.  Nothing is computed.
template <int ADIRECT>
__global__ __launch_bounds__(512) void gemm_like2_kernel(const char* __restrict__ src, int pitch, int ktiles, int rounds,
                                                         unsigned long long* clk, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const size_t tile_bytes = (size_t)256 * pitch;
    uint32_t vb[4], va[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { const int r = wave * 32 + p * 8 + (lane >> 3); vb[p] = (uint32_t)r * pitch + (lane & 7) * 16; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {                 // p = m * 2 + ks
        const int r = wave * 32 + (p >> 1) * 16 + (lane & 15);
        va[p] = (uint32_t)r * pitch + (p & 1) * 64 + (lane >> 4) * 16;
    }
    const int T = rounds * ktiles;
    u32x4 A[4][4];
    unsigned acc = 0;
    auto src_of = [&](int n, const char*& a, const char*& b) {
        const int rd = n / ktiles, t = n - rd * ktiles;
        a = src + (size_t)(j & 3) * tile_bytes + (size_t)t * 128;
        b = src + (size_t)(4 + ((rd * 8 + xcd) * 8 + (j >> 2))) * tile_bytes + (size_t)t * 128;
    };
#define ISSUE(N_, SET_)                                                                                   \
    if ((N_) < T) {                                                                                       \
        const char *a_, *b_;                                                                              \
        src_of((N_), a_, b_);                                                                             \
        const uint32_t dst = lds + ((N_) & 3) * 32768 + wave * 4096;                                      \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) glds16_asm(b_, vb[p], dst + p * 1024);              \
        if (ADIRECT) { _Pragma("unroll") for (int p = 0; p < 4; ++p) gload16_asm(A[SET_][p], a_, va[p]); }  \
        else { _Pragma("unroll") for (int p = 0; p < 4; ++p) glds16_asm(a_, vb[p], lds + 131072 + wave * 4096 + p * 1024); } \
    }
#define STEP(I_)                                                                                          \
    {                                                                                                     \
        const int n_ = t + (I_);                                                                          \
        if (n_ < T) {                                                                                     \
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(A[I_][0]), "+v"(A[I_][1]), "+v"(A[I_][2]), "+v"(A[I_][3]) :: "memory"); \
            __builtin_amdgcn_s_barrier();                                                                 \
            ISSUE(n_ + 3, ((I_) + 3) & 3)                                                                 \
            acc += A[I_][0][0] ^ A[I_][1][1] ^ A[I_][2][2] ^ A[I_][3][3];                                 \
        }                                                                                                 \
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) A[i][p] = u32x4{0u, 0u, 0u, 0u};
    { const int t = 0; (void)t; ISSUE(0, 0) ISSUE(1, 1) ISSUE(2, 2) }
    for (int t = 0; t < T; t += 4) { STEP(0) STEP(1) STEP(2) STEP(3) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
#undef STEP
#undef ISSUE
}

template <int ADIRECT>
static void run_gemm_like2(const char* name, const char* src, int pitch, int rounds) {
    unsigned long long* clk; unsigned* sink;
    hipMalloc(&clk, 256 * 8); hipMalloc(&sink, 64);
    const int lds_bytes = ADIRECT ? 131072 : 131072 + 32768;
    hipFuncSetAttribute((const void*)gemm_like2_kernel<ADIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int ktiles = pitch / 128;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(gemm_like2_kernel<ADIRECT>, dim3(256), dim3(512), lds_bytes, 0, src, pitch, ktiles, rounds, clk, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : h) mean += c; mean /= 256;
    const double bytes = (double)rounds * ktiles * 65536;
    printf("%-46s %8.3f ms  %6.1f GB/s/CU  %6.2f TB/s  %5.1f B/clk/CU  (%.0f clk/K-tile)\n", name, best, bytes / best / 1e6,
           bytes * 256 / best / 1e9, bytes / mean, mean / (rounds * ktiles));
    hipFree(clk); hipFree(sink);
}

int main() {
    const size_t total = (size_t)2 << 30;                // 2 GiB source: 8 MiB per workgroup
    char* src; hipMalloc(&src, total + (1 << 20)); hipMemset(src, 1, total);
    const size_t per = total / 256;
    // ring-GEMM-like: 256 rows x pitch 2 KiB (K = 1024 bf16): a stage walks 64 B (shape 0) / 128 B (shape 1) along every row
    const int pitch = 2048;
    // stages so that a workgroup stays inside its 8 MiB: rows used = 8*PIECES*16 (shape0) / *8 (shape1); k-range = pitch
    run<0, 4>("16 rows x 64 B, 4 pieces/wave, streaming", src, per, pitch, pitch / 64, 0);
    run<1, 4>(" 8 rows x 128 B, 4 pieces/wave, streaming", src, per, pitch, pitch / 128, 0);
    run<2, 4>("1 KiB contiguous, 4 pieces/wave, streaming", src, per, 0, (int)(per / (32 * 1024)), 0);
    run<0, 4>("16 rows x 64 B, 4 pieces/wave, L2 (alias 8)", src, per, pitch, pitch / 64, 8);
    run<1, 4>(" 8 rows x 128 B, 4 pieces/wave, L2 (alias 8)", src, per, pitch, pitch / 128, 8);
    run<2, 4>("1 KiB contiguous, 4 pieces/wave, L2 (alias 8)", src, per, 0, (int)(per / (32 * 1024)), 8);
    // long rows (pitch 8 KiB = K 4096): more stages per launch
    run<0, 4>("16 rows x 64 B, pitch 8 KiB, streaming", src, per, 8192, 8192 / 64, 0);
    run<1, 4>(" 8 rows x 128 B, pitch 8 KiB, streaming", src, per, 8192, 8192 / 128, 0);
    run<0, 4>("16 rows x 64 B, pitch 8 KiB, L2 (alias 8)", src, per, 8192, 8192 / 64, 8);
    run<1, 4>(" 8 rows x 128 B, pitch 8 KiB, L2 (alias 8)", src, per, 8192, 8192 / 128, 8);
    run<0, 2>("16 rows x 64 B, 2 pieces/wave, pitch 8 KiB, L2", src, per, 8192, 8192 / 64, 8);
    run<1, 2>(" 8 rows x 128 B, 2 pieces/wave, pitch 8 KiB, L2", src, per, 8192, 8192 / 128, 8);
    // GEMM-like sharing pattern: FC2 (pitch 8 KiB, 8 rounds: 4 + 8*8*8 tiles of 2 MiB = 1 GiB), out-proj / QKV (pitch 2 KiB)
    run_gemm_like("gemm-like, pitch 8192 (fc2), 8 rounds", src, 8192, 8);
    run_gemm_like("gemm-like, pitch 8320 (fc2 padded)", src, 8320, 8);
    run_gemm_like("gemm-like, pitch 2048 (K=1024), 8 rounds", src, 2048, 8);
    run_gemm_like("gemm-like, pitch 2176 (padded), 8 rounds", src, 2176, 8);
    run_gemm_like("gemm-like, pitch 2048 (K=1024), 24 rounds", src, 2048, 24);
    // split delivery: B through a 4 x 32 KiB LDS ring (3 K-tiles in flight), A direct to registers (3 K-tiles in flight)
    run_gemm_like2<1>("split: B ring 4x32K + A->regs, pitch 8192", src, 8192, 8);
    run_gemm_like2<1>("split: B ring 4x32K + A->regs, pitch 2048", src, 2048, 8);
    run_gemm_like2<1>("split: B ring 4x32K + A->regs, pitch 2048 x24", src, 2048, 24);
    run_gemm_like2<0>("B ring 4x32K + A one LDS buffer (no ring)", src, 8192, 8);
    return 0;
}
