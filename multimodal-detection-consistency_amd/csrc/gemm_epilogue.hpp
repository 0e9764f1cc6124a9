// Fused epilogues of the bf16 GEMM kernels (bias, QuickGELU, bf16 / fp32 / residual stores).
#pragma once
#include "gemm_ring.hpp"
#include "kernels.hpp"

struct GemmEpilogue {
    const float* bias;     // [I] or nullptr
    void* out;             // [J, ldo]
    int64_t ldo;
};

__device__ __forceinline__ float quick_gelu(float x) {
    // x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)): v_exp_f32 + v_rcp_f32 (1 ulp each; the
    // result is rounded to bf16 anyway) instead of an IEEE division.  x -> -inf: 2^(+inf) = inf,
    // rcp(inf) = 0, x * 0 = -0.
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554670f * x));
}

// out[j, i..i+3] for one lane: i = 4 consecutive out-features.  The vector
// path needs all four in range and a 4-element-aligned leading dimension;
// ragged edges (bank samples, cosine matrices) take the scalar path.
template <int EPI>
__device__ __forceinline__ void gemm_store4(const GemmEpilogue& e, int I, int i, int j, f32x4_t v) {
    const bool vec = (i + 3 < I) && ((e.ldo & 3) == 0);
    if (vec) {
        if (e.bias) v += *(const f32x4_t*)(e.bias + i);
        if (EPI == TVC_EPI_F32) {
            *(f32x4_t*)((float*)e.out + (int64_t)j * e.ldo + i) = v;
        } else if (EPI == TVC_EPI_RESID_F32) {
            float* p = (float*)e.out + (int64_t)j * e.ldo + i;
            const f32x4_t r = *(const f32x4_t*)p;
            *(f32x4_t*)p = r + v;
        } else {
            if (EPI == TVC_EPI_GELU_BF16) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = quick_gelu(v[t]);
            }
            u32x2_t o;
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
            *(u32x2_t*)((uint16_t*)e.out + (int64_t)j * e.ldo + i) = o;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (i + t >= I) break;
        float x = v[t] + (e.bias ? e.bias[i + t] : 0.f);
        const int64_t off = (int64_t)j * e.ldo + i + t;
        if (EPI == TVC_EPI_F32) ((float*)e.out)[off] = x;
        else if (EPI == TVC_EPI_RESID_F32) ((float*)e.out)[off] += x;
        else {
            if (EPI == TVC_EPI_GELU_BF16) x = quick_gelu(x);
            ((uint16_t*)e.out)[off] = f32_to_bf16_bits(x);
        }
    }
}

// Epilogue of one 256 x 256 tile.  Fast path (interior tile, aligned rows): the
// bias vectors are loaded once up front and, for the residual form, the eight
// read-modify-write loads of a column block are issued before their stores, so
// no store waits behind a load's vmcnt.
// BIAS_LDS: the tile's 256 bias values were staged in LDS (at `bias_lds`) by the persistent
// kernel.  The fast path (tile fully inside the output, aligned rows) contains no
// exec-masked region and no global load that is not consumed before its end, so hipcc's
// waitcnt pass sees nothing pending when a persistent caller loops back.
// NT: 16-token sub-tiles per wave (4: eight-wave kernels, 8: the four-wave kernel).  BIAS_REGS: see below.
template <int EPI, bool BIAS_LDS = false, int NT = 4, bool BIAS_REGS = false>
__device__ __forceinline__ void gemm_tile_epilogue(const f32x4_t (&acc)[8][NT], const GemmOperands& g,
                                                   const GemmEpilogue& e, int i0, int j0, int wm, int wn, int lane,
                                                   const char* bias_lds = nullptr) {
    // the four-wave kernel (NT = 8) carries only the 16-byte bf16 form: fewer live registers
    constexpr bool BF16_OUT = (EPI == TVC_EPI_BF16 || EPI == TVC_EPI_GELU_BF16);
    const bool fast = (i0 + GEMM_BM <= g.I) && (j0 + GEMM_BN <= g.J) &&
                      ((e.ldo & ((NT > 4 && BF16_OUT) ? 7 : 3)) == 0);
    if (fast) {
        const int il = wm * 128 + (lane >> 4) * 4;          // tile-local first out-feature of this lane
        auto bias_read = [&](int m) -> f32x4_t {
            if (!e.bias) return f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (BIAS_LDS)
                return *(const __attribute__((address_space(3))) f32x4_t*)(
                    (const __attribute__((address_space(3))) char*)bias_lds + (il + m * 16) * 4);
            return *(const f32x4_t*)(e.bias + i0 + il + m * 16);
        };
        // BIAS_REGS (GEMM form 4, whose operand fragments are dead here): the lane's 8 bias vectors are read ONCE, back to
        // back, into registers.  Left to hipcc, every sub-tile re-reads its vector behind a branch on e.bias and waits
        // for it -- 32 exposed LDS round trips per wave and tile.  The other forms have no registers to spare.
        f32x4_t bias_v[BIAS_REGS ? 8 : 1];
        if (BIAS_REGS) {
            if (e.bias) {
#pragma unroll
                for (int m = 0; m < 8; ++m) bias_v[BIAS_REGS ? m : 0] = bias_read(m);
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) bias_v[BIAS_REGS ? m : 0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        auto bias_of = [&](int m) -> f32x4_t { return BIAS_REGS ? bias_v[BIAS_REGS ? m : 0] : bias_read(m); };
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int j = j0 + wn * (NT * 16) + n * 16 + (lane & 15);
            if (EPI == TVC_EPI_RESID_F32) {
                float* p = (float*)e.out + (int64_t)j * e.ldo + i0 + il;
#pragma unroll
                for (int h = 0; h < 2; ++h) {           // 4 + 4: read-modify-write loads before their stores
                    f32x4_t r[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) r[m] = *(const f32x4_t*)(p + (h * 4 + m) * 16);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        *(f32x4_t*)(p + (h * 4 + m) * 16) = r[m] + acc[h * 4 + m][n] + bias_of(h * 4 + m);
                }
            } else if (EPI == TVC_EPI_F32) {
                float* p = (float*)e.out + (int64_t)j * e.ldo + i0 + il;
#pragma unroll
                for (int m = 0; m < 8; ++m) *(f32x4_t*)(p + m * 16) = acc[m][n] + bias_of(m);
            } else if (NT > 4 || (e.ldo & 7) == 0) {
                // bf16 outputs: a lane's 4 features are 8 B and the 4 lanes of a token cover 32 B per
                // sub-tile.  Swapping 16-lane rows between the sub-tiles m and m+1
                // (v_permlane16_swap: odd rows of X <-> even rows of Y) leaves every lane with 8
                // consecutive features, so it stores 16 B and a token gets 64 contiguous bytes per
                // store instruction: half the store instructions, twice the segment size.
                const int gq = lane >> 4;
                uint16_t* p = (uint16_t*)e.out + (int64_t)j * e.ldo + i0 + wm * 128 + (gq & 1) * 16 + (gq >> 1) * 8;
#pragma unroll
                for (int mp = 0; mp < 4; ++mp) {
                    f32x4_t v0 = acc[2 * mp][n] + bias_of(2 * mp);
                    f32x4_t v1 = acc[2 * mp + 1][n] + bias_of(2 * mp + 1);
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
            } else {
                uint16_t* p = (uint16_t*)e.out + (int64_t)j * e.ldo + i0 + il;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    f32x4_t v = acc[m][n] + bias_of(m);
                    if (EPI == TVC_EPI_GELU_BF16) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = quick_gelu(v[t]);
                    }
                    u32x2_t o;
                    o[0] = pack_bf16x2(v[0], v[1]);
                    o[1] = pack_bf16x2(v[2], v[3]);
                    *(u32x2_t*)(p + m * 16) = o;
                }
            }
            if (NT > 4) __builtin_amdgcn_sched_barrier(0);      // one token sub-tile at a time: short live ranges
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int j = j0 + wn * (NT * 16) + n * 16 + (lane & 15);
        if (j >= g.J) continue;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int i = i0 + wm * 128 + m * 16 + (lane >> 4) * 4;
            if (i < g.I) gemm_store4<EPI>(e, g.I, i, j, acc[m][n]);
        }
        if (NT > 4) __builtin_amdgcn_sched_barrier(0);
    }
    // The loads above sit in exec-masked branches.  Tell hipcc's waitcnt pass that none is
    // pending when a persistent caller loops back (vmcnt(0), lgkmcnt/expcnt untouched): without
    // this it guards the loop body's first VGPR write with a vmcnt(0) that drains the LDS-DMA
    // ring on EVERY stage.
    __builtin_amdgcn_s_waitcnt(0x0F70);
}

