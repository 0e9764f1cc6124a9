// Input-gradient (dX only) row kernels of the vision tower: what a PGD / Hubness attack needs from
// `encode_image_tensor(x, requires_grad=True)` (reference: src/attacks/pgd_attack.py:456-486,
// src/attacks/hubness_attack.py:269-424).  No weight gradients.
// HBM-bound, one 64-lane wave per row, 16-byte accesses, shuffle reductions -- the layout of elementwise.hip.
#include "common.hpp"
#include "kernels.hpp"

#define LN_EPS 1e-5f
#define ROWS_PER_BLOCK 4

// ---------------------------------------------------------------------------
// LayerNorm backward with the residual path folded in:
//   x_eff = x (+ delta)            the forward's LN input (fp32 residual stream + the bf16 delta it folded)
//   g     = dy * gamma             dy: gradient w.r.t. the LN output (bf16, or fp32 when DY32)
//   dx    = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dres)
// dres / dx are fp32 [*, d] rows of the residual-stream gradient (may alias: in place); dx16 (optional) receives
// the bf16 copy that the next GEMM reads.  Row addressing: x rows at `x_row_stride` (elements) with optional
// row_idx; dy / delta / dres / dx compact [rows, d] unless *_strided says they share x's row layout.
// ---------------------------------------------------------------------------
template <bool DY32>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t x_row_stride,
                                                            const uint16_t* __restrict__ delta,
                                                            const void* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* dres, float* dx, uint16_t* __restrict__ dx16,
                                                            int rows, int d, int64_t out_row_stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4_t* xr = (const f32x4_t*)(x + (int64_t)row * x_row_stride);
    const u32x2_t* dr = delta ? (const u32x2_t*)(delta + (int64_t)row * x_row_stride) : nullptr;
    const int nv = d >> 2;
    f32x4_t v[4], g[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        g[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[i] = xr[c];
            if (dr) {
                const u32x2_t dd = dr[c];
                v[i][0] += __uint_as_float(dd[0] << 16); v[i][1] += __uint_as_float(dd[0] & 0xffff0000u);
                v[i][2] += __uint_as_float(dd[1] << 16); v[i][3] += __uint_as_float(dd[1] & 0xffff0000u);
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            const f32x4_t gm = ((const f32x4_t*)gamma)[c];
            if (DY32) {
                g[i] = ((const f32x4_t*)((const float*)dy + (int64_t)row * d))[c] * gm;
            } else {
                const u32x2_t yy = ((const u32x2_t*)((const uint16_t*)dy + (int64_t)row * d))[c];
                g[i][0] = __uint_as_float(yy[0] << 16) * gm[0]; g[i][1] = __uint_as_float(yy[0] & 0xffff0000u) * gm[1];
                g[i][2] = __uint_as_float(yy[1] << 16) * gm[2]; g[i][3] = __uint_as_float(yy[1] & 0xffff0000u) * gm[3];
            }
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[i][t] -= mean; q += v[i][t] * v[i][t]; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + LN_EPS);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[i][t] *= rstd; sg += g[i][t]; sgx = fmaf(g[i][t], v[i][t], sgx); }
        }
    }
    const float c1 = wave_sum(sg) / (float)d, c2 = wave_sum(sgx) / (float)d;
    const f32x4_t* rr = dres ? (const f32x4_t*)(dres + (int64_t)row * out_row_stride) : nullptr;
    f32x4_t* ox = (f32x4_t*)(dx + (int64_t)row * out_row_stride);
    u32x2_t* o16 = dx16 ? (u32x2_t*)(dx16 + (int64_t)row * out_row_stride) : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            f32x4_t o;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = rstd * (g[i][t] - c1 - v[i][t] * c2);
            if (rr) o += rr[c];
            ox[c] = o;
            if (o16) o16[c] = u32x2_t{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

hipError_t launch_layernorm_bwd(const float* x, int64_t x_row_stride, const uint16_t* delta, const void* dy, int dy_fp32,
                                const float* gamma, const float* dres, float* dx, uint16_t* dx16, int rows, int d,
                                int64_t out_row_stride, hipStream_t stream) {
    if (d % 4 != 0 || d > 1024 || rows < 0) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const int grid = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (dy_fp32)
        hipLaunchKernelGGL(layernorm_bwd_kernel<true>, dim3(grid), dim3(256), 0, stream, x, x_row_stride, delta, dy, gamma,
                           dres, dx, dx16, rows, d, out_row_stride);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<false>, dim3(grid), dim3(256), 0, stream, x, x_row_stride, delta, dy, gamma,
                           dres, dx, dx16, rows, d, out_row_stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// ln_pre backward: the forward's input row is (t == 0 ? cls : patch_out[b, t-1]) + pos[t] (assemble_lnpre_kernel);
// dy = fp32 gradient w.r.t. the residual stream after ln_pre; out = bf16 gradient w.r.t. patch_out rows
// [B * (T-1), d] (the class row's gradient goes to a parameter and is dropped).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lnpre_bwd_kernel(const float* __restrict__ patch_out, const float* __restrict__ pos,
                                                        const float* __restrict__ gamma, const float* __restrict__ dy,
                                                        uint16_t* __restrict__ dpatch, int B, int T, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t prow = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);     // row of patch_out
    if (prow >= (int64_t)B * (T - 1)) return;
    const int64_t b = prow / (T - 1);
    const int t = (int)(prow - b * (T - 1)) + 1;
    const f32x4_t* src = (const f32x4_t*)(patch_out + prow * d);
    const f32x4_t* pr = (const f32x4_t*)(pos + (int64_t)t * d);
    const f32x4_t* gy = (const f32x4_t*)(dy + (b * T + t) * d);
    const int nv = d >> 2;
    f32x4_t v[4], g[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; g[i] = v[i];
        if (c < nv) {
            v[i] = src[c] + pr[c];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            g[i] = gy[c] * ((const f32x4_t*)gamma)[c];
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] -= mean; q += v[i][e] * v[i][e]; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + LN_EPS);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] *= rstd; sg += g[i][e]; sgx = fmaf(g[i][e], v[i][e], sgx); }
        }
    }
    const float c1 = wave_sum(sg) / (float)d, c2 = wave_sum(sgx) / (float)d;
    u32x2_t* o = (u32x2_t*)(dpatch + prow * d);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            f32x4_t r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = rstd * (g[i][e] - c1 - v[i][e] * c2);
            o[c] = u32x2_t{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
        }
    }
}

hipError_t launch_lnpre_bwd(const float* patch_out, const float* pos, const float* gamma, const float* dy,
                            uint16_t* dpatch, int B, int T, int d, hipStream_t stream) {
    if (d % 4 != 0 || d > 1024) return hipErrorInvalidValue;
    const int64_t rows = (int64_t)B * (T - 1);
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(lnpre_bwd_kernel, dim3((int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, stream,
                       patch_out, pos, gamma, dy, dpatch, B, T, d);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// quick-GELU backward, in place on dm:  du = dm * (s + 1.702 u s (1 - s)),  s = sigmoid(1.702 u)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_bwd_kernel(uint16_t* __restrict__ dm, const uint16_t* __restrict__ u, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4_t a = ((const u32x4_t*)dm)[i], b = ((const u32x4_t*)u)[i];
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const float g = hh ? __uint_as_float(a[e] & 0xffff0000u) : __uint_as_float(a[e] << 16);
                const float x = hh ? __uint_as_float(b[e] & 0xffff0000u) : __uint_as_float(b[e] << 16);
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554670f * x));
                r[hh] = g * (sg + 1.702f * x * sg * (1.0f - sg));
            }
            o[e] = pack_bf16x2(r[0], r[1]);
        }
        ((u32x4_t*)dm)[i] = o;
    }
}

hipError_t launch_gelu_bwd(uint16_t* dm, const uint16_t* u, int64_t n, hipStream_t stream) {
    if (n % 8 != 0) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    int64_t grid = (n / 8 + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((int)grid), dim3(256), 0, stream, dm, u, n / 8);
    return hipGetLastError();
}

// quick-GELU forward on a bf16 buffer (the grad-mode forward stores the pre-activation and applies this)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const uint16_t* __restrict__ u, uint16_t* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4_t b = ((const u32x4_t*)u)[i];
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = __uint_as_float(b[e] << 16), x1 = __uint_as_float(b[e] & 0xffff0000u);
            o[e] = pack_bf16x2(x0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554670f * x0)),
                               x1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554670f * x1)));
        }
        ((u32x4_t*)out)[i] = o;
    }
}

hipError_t launch_gelu_fwd(const uint16_t* u, uint16_t* out, int64_t n, hipStream_t stream) {
    if (n % 8 != 0) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    int64_t grid = (n / 8 + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((int)grid), dim3(256), 0, stream, u, out, n / 8);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// y = x / |x| backward (rows of the projected embedding):  dx = (dy - y (y . dy)) / |x|;  out bf16 (GEMM operand)
// normalize == 0: plain cast.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         uint16_t* __restrict__ dx16, int rows, int d, int normalize) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * d;
    const float* gr = dy + (int64_t)row * d;
    uint16_t* o = dx16 + (int64_t)row * d;
    if (!normalize) {
        for (int c = lane; c < d; c += 64) o[c] = f32_to_bf16_bits(gr[c]);
        return;
    }
    float ss = 0.f, sd = 0.f;
    for (int c = lane; c < d; c += 64) { ss = fmaf(xr[c], xr[c], ss); sd = fmaf(xr[c], gr[c], sd); }
    ss = wave_sum(ss); sd = wave_sum(sd);
    const float inv = 1.0f / sqrtf(ss);
    const float k = sd * inv * inv;                     // (y . dy) / |x| = (x . dy) / |x|^2
    for (int c = lane; c < d; c += 64) o[c] = f32_to_bf16_bits((gr[c] - xr[c] * k) * inv);
}

hipError_t launch_l2norm_bwd(const float* x, const float* dy, uint16_t* dx16, int rows, int d, int normalize, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, stream, x, dy, dx16,
                       rows, d, normalize);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// col2im of the stride = patch conv: dcols fp32 [B*P, Kp] (column order (c, ky, kx)) -> dpix fp32 [B,3,S,S].
// Patches do not overlap: every pixel reads exactly one element.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dpix, int B, int S,
                                                     int patch, int Kp) {
    const int gside = S / patch, P = gside * gside;
    const int64_t total = (int64_t)B * 3 * S * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % S);
        const int yy = (int)((i / S) % S);
        const int c = (int)((i / ((int64_t)S * S)) % 3);
        const int64_t b = i / ((int64_t)3 * S * S);
        const int py = yy / patch, ky = yy - py * patch, px = xx / patch, kx = xx - px * patch;
        dpix[i] = dcols[(b * P + py * gside + px) * Kp + (c * patch + ky) * patch + kx];
    }
}

hipError_t launch_col2im(const float* dcols, float* dpix, int B, int S, int patch, int Kp, hipStream_t stream) {
    const int64_t total = (int64_t)B * 3 * S * S;
    if (total == 0) return hipSuccess;
    int64_t grid = (total + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(col2im_kernel, dim3((int)grid), dim3(256), 0, stream, dcols, dpix, B, S, patch, Kp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// bf16 [R, C] -> [C, R] (weights for the dX GEMMs: A operand = W^T, K-contiguous along the forward's out features)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C) {
    __shared__ uint16_t tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = in[(int64_t)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) out[(int64_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

hipError_t launch_transpose_bf16(const uint16_t* in, uint16_t* out, int R, int C, hipStream_t stream) {
    if (R <= 0 || C <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream, in, out, R, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// One PGD step on a batch (src/attacks/pgd_attack.py:500-521): momentum on the per-sample L1-normalised gradient,
// sign step, projection to the eps ball around the clean image, clamp.  One workgroup per image.
//   mom = mu * mom + grad / |grad|_1      (use_momentum; else the raw gradient)
//   adv = clamp(clean + clamp(adv + dir * alpha * sign(mom) - clean, -eps, eps), lo, hi)      dir = +1 untargeted, -1 targeted
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void pgd_step_kernel(float* __restrict__ adv, const float* __restrict__ clean,
                                                        const float* __restrict__ grad, float* __restrict__ mom, int64_t n,
                                                        float eps, float alpha, float mu, float lo, float hi, float dir) {
    __shared__ float red[16];
    const int64_t base = (int64_t)blockIdx.x * n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float inv = 1.0f;
    if (mom) {
        float s = 0.f;
        for (int64_t i = t; i < n; i += 1024) s += fabsf(grad[base + i]);
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += red[w];
        inv = 1.0f / tot;
    }
    for (int64_t i = t; i < n; i += 1024) {
        float g = grad[base + i];
        if (mom) { g = mu * mom[base + i] + g * inv; mom[base + i] = g; }
        const float sgn = (g > 0.f) ? 1.f : ((g < 0.f) ? -1.f : 0.f);      // torch.sign
        const float c = clean[base + i];
        float a = adv[base + i] + dir * alpha * sgn;
        float dl = a - c;
        dl = fminf(fmaxf(dl, -eps), eps);
        adv[base + i] = fminf(fmaxf(c + dl, lo), hi);
    }
}

// ---------------------------------------------------------------------------
// L2-constrained step (src/attacks/hubness_attack.py:378-386), one workgroup per image:
//   adv += dir * step * grad / (|grad|_2 + 1e-8);   d = adv - clean;   d *= min(|d|_2, eps) / (|d|_2 + 1e-8)
//   adv = clamp(clean + d, lo, hi)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void l2_step_kernel(float* __restrict__ adv, const float* __restrict__ clean,
                                                       const float* __restrict__ grad, int64_t n, float eps, float step,
                                                       float lo, float hi, float dir) {
    __shared__ float red[16];
    const int64_t base = (int64_t)blockIdx.x * n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += red[w];
        return tot;
    };
    float s = 0.f;
    for (int64_t i = t; i < n; i += 1024) { const float g = grad[base + i]; s += g * g; }
    const float gscale = dir * step / (sqrtf(block_sum(s)) + 1e-8f);
    float q = 0.f;
    for (int64_t i = t; i < n; i += 1024) {
        const float a = adv[base + i] + gscale * grad[base + i];
        adv[base + i] = a;                       // each element is read back by the thread that wrote it
        const float d = a - clean[base + i];
        q += d * d;
    }
    const float dn = sqrtf(block_sum(q));
    const float dscale = fminf(dn, eps) / (dn + 1e-8f);
    for (int64_t i = t; i < n; i += 1024) {
        const float c = clean[base + i];
        adv[base + i] = fminf(fmaxf(c + (adv[base + i] - c) * dscale, lo), hi);
    }
}

hipError_t launch_l2_step(float* adv, const float* clean, const float* grad, int B, int64_t n, float eps, float step, float lo,
                          float hi, int descent, hipStream_t stream) {
    if (B <= 0 || n <= 0) return hipSuccess;
    hipLaunchKernelGGL(l2_step_kernel, dim3(B), dim3(1024), 0, stream, adv, clean, grad, n, eps, step, lo, hi, descent ? -1.0f : 1.0f);
    return hipGetLastError();
}

hipError_t launch_pgd_step(float* adv, const float* clean, const float* grad, float* mom, int B, int64_t n, float eps,
                           float alpha, float mu, float lo, float hi, int targeted, hipStream_t stream) {
    if (B <= 0 || n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pgd_step_kernel, dim3(B), dim3(1024), 0, stream, adv, clean, grad, mom, n, eps, alpha, mu, lo, hi,
                       targeted ? -1.0f : 1.0f);
    return hipGetLastError();
}
