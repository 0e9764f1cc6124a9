// bf16 MFMA GEMM core for gfx950:  acc[i, j] = sum_k A[i, k] * B[j, k]
//
//   A  bf16 [I, lda]   "row operand": weights (out-features) or bank rows
//   B  bf16 [J, ldb]   "column operand": tokens or query rows
//
// Both operands are K-contiguous (the nn.Linear weight layout and the natural
// layout of activations / bank rows), so both are staged with the same 16-byte
// global_load_lds pieces.  Workgroup tile 256 (I) x 256 (J) x 64 (K), 8 waves
// as 2 (I) x 4 (J); each wave owns 128 x 64 = 8 x 4 MFMA 16x16x32 tiles
// (128 fp32 accumulators per lane).  MFMA rows = I so that a lane holds 4
// consecutive out-features of one token: the epilogue stores 8-byte (bf16) or
// 16-byte (fp32) pieces of the token-major output.
//
// LDS: 2 buffers x (A tile 32 KiB + B tile 32 KiB) = 128 KiB, tiles are
// [256 rows][64 k] bf16 with 128-byte rows.  global_load_lds writes a wave's
// 64 x 16 B linearly (8 rows), so the XOR swizzle that makes the ds_read_b128
// fragment reads conflict-free is applied to the per-lane SOURCE address and
// to the read address (16-byte chunk c of row r lives at chunk c ^ ((r>>1)&7)).
//
// K may consist of several "planes" (split-bf16 operands for fp32-grade
// cosines): plane p reads A columns a_plane_off[p] + k and B columns
// b_plane_off[p] + k, all accumulated into the same tile.
#pragma once
#include "common.hpp"

#define GEMM_BM 256          // I rows per workgroup
#define GEMM_BN 256          // J rows per workgroup
#define GEMM_BK 64
#define GEMM_THREADS 512
#define GEMM_MAX_PLANES 9
#define GEMM_TILE_BYTES (256 * 64 * 2)            // 32 KiB
#define GEMM_LDS_BYTES (4 * GEMM_TILE_BYTES)      // 128 KiB

struct GemmOperands {
    const uint16_t* A;
    const uint16_t* B;
    int64_t lda, ldb;          // elements
    int I, J;                  // valid rows of A / B (loads clamp to the last row)
    int ksteps_per_plane;      // K / 64
    int planes;                // 1..GEMM_MAX_PLANES (2-3: split-bf16 operands; 9: the taps of a 3x3 convolution)
    int a_plane_off[GEMM_MAX_PLANES];
    int b_plane_off[GEMM_MAX_PLANES];
};

typedef f32x4_t gemm_acc_t[8][4];

// Stage one [256][64] bf16 tile: wave w copies rows w*32 .. w*32+31 with four
// 1-KiB LDS-DMA pieces (8 rows each).
__device__ __forceinline__ void gemm_stage_tile(const uint16_t* __restrict__ base, int64_t ld,
                                                int row0, int nrows, int koff,
                                                char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        const uint16_t* src = base + (int64_t)gr * ld + koff + c * 8;
        glds16(src, lds_tile + (wave * 32 + i * 8) * 128);
    }
}

// acc must be zero-initialised by the caller (or carry a running sum).
// [step_begin, step_end): range of 64-deep K steps (over all planes) to multiply; the default is
// the whole K (split-K partial tiles pass a sub-range).
__device__ __forceinline__ void gemm_mainloop(gemm_acc_t& acc, const GemmOperands& g,
                                              int i0, int j0, char* smem, int step_begin = 0, int step_end = -1) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = (step_end < 0 ? g.planes * g.ksteps_per_plane : step_end) - step_begin;

    // lane-constant swizzled chunk offsets for k-sub-steps 0 and 1
    const int sw0 = ((0 + (lane >> 4)) ^ ((lane >> 1) & 7)) * 16;
    const int sw1 = ((4 + (lane >> 4)) ^ ((lane >> 1) & 7)) * 16;
    const int a_row_off = (wm * 128 + (lane & 15)) * 128;
    const int b_row_off = (wn * 64 + (lane & 15)) * 128;

    int p = step_begin / g.ksteps_per_plane, kk = step_begin - p * g.ksteps_per_plane;   // plane / k-step-in-plane of the NEXT tile to stage
    auto stage_next = [&](int buf) {
        const int aoff = g.a_plane_off[p] + kk * GEMM_BK;
        const int boff = g.b_plane_off[p] + kk * GEMM_BK;
        char* t = smem + buf * (2 * GEMM_TILE_BYTES);
        gemm_stage_tile(g.A, g.lda, i0, g.I, aoff, t, wave, lane);
        gemm_stage_tile(g.B, g.ldb, j0, g.J, boff, t + GEMM_TILE_BYTES, wave, lane);
        if (++kk == g.ksteps_per_plane) { kk = 0; ++p; }
    };

    stage_next(0);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed and every wave is done reading the other buffer.
        // The LDS-DMA pieces are only ordered by the issuing wave's vmcnt: wait
        // explicitly (hipcc does not always emit it before the barrier).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage_next((kt + 1) & 1);
        const char* ta = smem + (kt & 1) * (2 * GEMM_TILE_BYTES);
        const char* tb = ta + GEMM_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int sw = s ? sw1 : sw0;
            bf16x8_t a[8], b[4];
#pragma unroll
            for (int m = 0; m < 8; ++m)
                a[m] = *(const bf16x8_t*)(ta + a_row_off + m * 2048 + sw);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                b[n] = *(const bf16x8_t*)(tb + b_row_off + n * 2048 + sw);
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
    // all waves must be done with the last buffer before a caller restages
    __syncthreads();
}

__device__ __forceinline__ void gemm_zero_acc(gemm_acc_t& acc) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}
