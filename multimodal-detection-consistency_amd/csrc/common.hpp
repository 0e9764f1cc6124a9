// Shared device helpers for the TVC HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define TVC_WAVE 64

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
    return __uint_as_float(((uint32_t)h) << 16);
}
// round-to-nearest-even through the compiler's cast (v_cvt_pk_bf16_f32; keeps NaN a NaN)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// wave-level all-lane sum / max over 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte async global->LDS copy (global_load_lds_dwordx4).  `lds_wave_base`
// must be wave-uniform: the hardware writes lane L at lds_wave_base + 16*L.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same copy issued from inline asm: scalar 64-bit base + per-lane 32-bit byte offset,
// LDS destination byte address in M0 (saved / restored: M0 is compiler-reserved).
// hipcc does not count this load in its own vmcnt bookkeeping, so (unlike the builtin)
// it never drains the queue before a ds_read that "may alias" a pending LDS-DMA:
// ALL ordering of these loads is by the caller's explicit s_waitcnt vmcnt(N) + barrier.
// Compiler-counted waits for its own younger loads stay correct (they are only stricter).
__device__ __forceinline__ void glds16_asm(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// Four pieces of one ring stage in ONE asm statement: a single M0 save / restore, and the
// second destination of each operand is the first + `second` bytes.
__device__ __forceinline__ void glds16x4_asm(const void* abase, uint32_t va0, uint32_t va1, uint32_t a_lds,
                                             const void* bbase, uint32_t vb0, uint32_t vb1, uint32_t b_lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %4\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(va0), "v"(va1), "s"(abase), "s"(bbase), "s"(a_lds), "v"(vb0), "v"(vb1), "s"(b_lds)
                 : "memory", "scc");
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}

// Bijective XCD-contiguous remap of a 1-D grid (8 XCDs, round-robin dispatch):
// workgroups that land on one XCD get a contiguous range of `lin`, so tiles
// sharing an operand panel share that XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
    const int xcd = bid & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
