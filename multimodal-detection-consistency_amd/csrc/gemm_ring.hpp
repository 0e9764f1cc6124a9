// Persistent, ring-pipelined bf16 MFMA GEMM main loop for gfx950 (helpers; the kernel and
// the full schedule description live in gemm.hip: gemm_ring_kernel).
//
// Same tile and wave decomposition as gemm_core.hpp (256 x 256 output tile, 8 waves as 2 x 4,
// 8 x 4 MFMA 16x16x32 tiles per wave) but:
//   * K advances in stages of 32 through a 4-slot LDS ring (4 x (A 16 KiB + B 16 KiB) = 128 KiB):
//     a stage is issued three barrier intervals before it is read;
//   * the waits are COUNTED (`s_waitcnt vmcnt(8)` leaves the two younger stages in flight across
//     the raw `s_barrier`) and the LDS-DMA is issued from inline asm, so hipcc neither adds its own
//     vmcnt(0) at the barrier nor drains the queue before a ds_read that "may alias" a pending DMA;
//   * workgroups are persistent: the stage stream runs across output tiles, so the next tile's
//     first stages are loading while the current tile's epilogue stores;
//   * rotated ping-pong: the two waves of a SIMD run {load, MFMA} and {MFMA, load} respectively,
//     one barrier per stage.
//
// LDS stage image: rows of 32 bf16 = 64 B, four 16-B chunks; chunk c of row r is stored at chunk
// c ^ (3 * ((r >> 3) & 1)).  With that swizzle the 16 lanes a ds_read_b128 services together
// (rows 0-3,12-15 at chunk g and rows 4-11 at chunk g^1, or the converse) hit 16 distinct 16-B
// slots of the 256-B bank row.  The DMA destination is lane-linear, so the swizzle is applied to
// the per-lane SOURCE address and to the fragment read address.
#pragma once
#include "gemm_core.hpp"

#define RING_BK 32
#define RING_SLOTS 4
#define RING_HALF_BYTES (256 * RING_BK * 2)        // one operand of one stage: 16 KiB
#define RING_SLOT_BYTES (2 * RING_HALF_BYTES)      // 32 KiB
#define RING_LDS_BYTES (RING_SLOTS * RING_SLOT_BYTES)

// wave w stages rows w*32 .. w*32+31 of a [256][32] operand tile: two 1-KiB pieces
__device__ __forceinline__ void ring_stage_half(const uint16_t* __restrict__ base, int64_t ld, int row0,
                                                int nrows, int koff, char* lds_half, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave * 32 + i * 16 + (lane >> 2);
        const int c = (lane & 3) ^ (3 * ((r >> 3) & 1));
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        glds16(base + (int64_t)gr * ld + koff + c * 8, lds_half + (wave * 32 + i * 16) * 64);
    }
}

struct RingSchedule {
    // persistent, XCD-contiguous tile walk: XCD x owns tiles [lo, hi); its workgroups
    // (wpx of them, local index j) take lo + j, lo + j + wpx, ...
    int lo, hi, wpx, j;
    __device__ __forceinline__ void init(int ntiles) {
        const int xcd = blockIdx.x & 7;
        const int q = ntiles >> 3, r = ntiles & 7;
        lo = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        hi = lo + q + (xcd < r ? 1 : 0);
        wpx = gridDim.x >> 3;
        j = blockIdx.x >> 3;
    }
    __device__ __forceinline__ int count() const { return (hi - lo - j + wpx - 1) / wpx > 0 ? (hi - lo - j + wpx - 1) / wpx : 0; }
    __device__ __forceinline__ int tile(int n) const { return lo + j + n * wpx; }
};
