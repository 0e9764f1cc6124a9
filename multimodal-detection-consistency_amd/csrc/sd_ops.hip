// Row / elementwise kernels of the latent-diffusion reference generator (tvc_sd.cpp): everything between the GEMMs of
// the UNet and the VAE decoder.  Activations are bf16 NHWC -- [n * H * W tokens, C channels], channels contiguous -- so
// that every convolution is a GEMM over K-contiguous token rows (3x3: rows gathered by im2col3x3_kernel; 1x1: the rows
// themselves) and every normalisation is a streaming pass; statistics and arithmetic are fp32.
#include "common.hpp"
#include "kernels.hpp"

namespace {

__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
    return u32x4_t{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

// Token row of pixel t (= y * W + x) of image img.  Dense layout: img * HW + t.  PADDED layout (the operands and the
// results of the 3x3 convolutions run as 9-plane GEMMs): every image is an (H + 2) x (W + 2) grid whose border rows are
// zero in a convolution's input, so that tap (ky, kx) of a convolution is the same rows shifted by
// (ky - 1) * (W + 2) + (kx - 1): no masking, no im2col.
__device__ __forceinline__ int64_t tok_row(int64_t img, int t, int H, int W, int pad) {
    if (!pad) return img * ((int64_t)H * W) + t;
    const int y = t / W, x = t - y * W;
    return img * ((int64_t)(H + 2) * (W + 2)) + (int64_t)(y + 1) * (W + 2) + (x + 1);
}

// ---- 3x3 convolution rows: out[(n, y, x), (ky, kx, c)] = in[n, y*s + ky - 1, x*s + kx - 1, c] (zero outside)
// `up`: the source is the nearest-2x upsampling of `in` (Upsample2D + conv in one gather).  C % 8 == 0.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                        int n, int Hi, int Wi, int C, int Ho, int Wo, int stride, int up) {
    const int cv = C >> 3;
    const int64_t total = (int64_t)n * Ho * Wo * 9 * cv;
    const int Hs = up ? Hi * 2 : Hi, Ws = up ? Wi * 2 : Wi;       // source extent as the convolution sees it
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % cv);
        int64_t r = t / cv;
        const int tap = (int)(r % 9);
        r /= 9;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const int64_t img = r / Ho;
        int yy = y * stride + tap / 3 - 1, xx = x * stride + tap % 3 - 1;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) {
            if (up) { yy >>= 1; xx >>= 1; }
            v = *(const u32x4_t*)(in + ((img * Hi + yy) * Wi + xx) * C + c * 8);
        }
        ((u32x4_t*)out)[t] = v;
    }
}

// fp32 NCHW [n, Cin, H, W] (Cin * 9 <= Kp) -> bf16 rows [n * H * W, Kp], column tap * Cin + ci, zero padded;
// the input is multiplied by `scale` (the VAE's 1 / scaling_factor)
__global__ __launch_bounds__(256) void im2col_in_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int n,
                                                        int Cin, int H, int W, int Kp, float scale) {
    const int64_t total = (int64_t)n * H * W * Kp;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int k = (int)(t % Kp);
        int64_t r = t / Kp;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int64_t img = r / H;
        float v = 0.f;
        if (k < 9 * Cin) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[((img * Cin + ci) * H + yy) * W + xx] * scale;
        }
        out[t] = f32_to_bf16_bits(v);
    }
}

// ---- GroupNorm statistics.  Thread (token lane tl, 8-channel vector v) walks the tokens tl, tl + lanes, ... of its slab
// with 16-byte loads (a wave reads whole rows) and keeps per-channel sums in registers; the per-channel sums of all token
// lanes meet in LDS, thread g then adds its group's channels in a fixed order: part[n][slab][group] = {sum, sumsq}.
// x' = x + tadd[n, c] (the resnet's time projection, fp32 [n, ld_t]) when given.  C % 8 == 0, C <= 4096.
__global__ __launch_bounds__(256) void gn_partial_kernel(const uint16_t* __restrict__ x, const float* __restrict__ tadd,
                                                         int64_t ld_t, float* __restrict__ part, int H, int W, int C, int groups,
                                                         int slab_tokens, int nslab, int in_pad) {
    extern __shared__ __attribute__((aligned(16))) float gn_red[];      // [lanes][C][2]
    const int HW = H * W;
    const int img = blockIdx.x / nslab, slab = blockIdx.x - img * nslab;
    const int cv = C >> 3;
    const int Wv = cv < 256 ? cv : 256;          // vector columns handled side by side
    const int lanes = 256 / Wv;                  // token lanes
    const int tl = threadIdx.x / Wv, v0 = threadIdx.x - tl * Wv;
    const int t0 = slab * slab_tokens, t1 = min(HW, t0 + slab_tokens);
    if (tl < lanes) {
        for (int v = v0; v < cv; v += Wv) {
            float s[8], q[8], ta[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; ta[i] = tadd ? tadd[(int64_t)img * ld_t + v * 8 + i] : 0.f; }
            int t = t0 + tl;
            for (; t + lanes < t1; t += 2 * lanes) {           // two rows in flight
                const u32x4_t r0 = *(const u32x4_t*)(x + tok_row(img, t, H, W, in_pad) * C + v * 8);
                const u32x4_t r1 = *(const u32x4_t*)(x + tok_row(img, t + lanes, H, W, in_pad) * C + v * 8);
                float f[8], h8[8];
                unpack8(r0, f);
                unpack8(r1, h8);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a = f[i] + ta[i], b2 = h8[i] + ta[i];
                    s[i] += a + b2; q[i] += a * a + b2 * b2;
                }
            }
            for (; t < t1; t += lanes) {
                float f[8];
                unpack8(*(const u32x4_t*)(x + tok_row(img, t, H, W, in_pad) * C + v * 8), f);
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float a = f[i] + ta[i]; s[i] += a; q[i] += a * a; }
            }
            float* o = gn_red + ((size_t)tl * C + v * 8) * 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) { o[2 * i] = s[i]; o[2 * i + 1] = q[i]; }
        }
    }
    __syncthreads();
    const int g = threadIdx.x;
    if (g < groups) {
        const int cpg = C / groups;
        float ss = 0.f, qq = 0.f;
        for (int l = 0; l < lanes; ++l) {
            const float* r = gn_red + ((size_t)l * C + g * cpg) * 2;
            for (int c = 0; c < cpg; ++c) { ss += r[2 * c]; qq += r[2 * c + 1]; }
        }
        float* o = part + (((int64_t)img * nslab + slab) * groups + g) * 2;
        o[0] = ss; o[1] = qq;
    }
}

// y = ((x + tadd) - mean) * rstd * gamma + beta, optional SiLU.  One workgroup per OUTPUT row y of one image (with
// out_pad the two border rows and the border columns are written as zeros: a convolution reads them as padding); thread
// (pixel lane, 8-channel vector) keeps its vector's scale / shift -- rstd * gamma and beta + (tadd - mean) * rstd * gamma
// -- in registers and walks the row's pixels: one fma (+ SiLU) per element, no integer division in the loop.
// The image's {mean, rstd} per group come first: every workgroup sums the slabs' partials of ITS image in fp64 (thread = (group,
// one of 8 slab lanes), then the 8 lanes in a fixed order) -- a few KB out of L2 per workgroup instead of a third kernel
// between the statistics pass and this one.
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, const float* __restrict__ tadd,
                                                       int64_t ld_t, const float* __restrict__ part, int nslab, double count,
                                                       float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       uint16_t* __restrict__ y, int n, int H, int W, int C, int groups, int silu,
                                                       int in_pad, int out_pad) {
    __shared__ double gn_sum[8][32][2];
    __shared__ float stats[32 * 2];
    const int cv = C >> 3, cpg = C / groups;
    const int Ho = out_pad ? H + 2 : H, Wo = out_pad ? W + 2 : W;
    const int img = blockIdx.x / Ho, yo = blockIdx.x - img * Ho;
    const int yy = out_pad ? yo - 1 : yo;
    const int Wv = cv < 256 ? cv : 256;
    const int lanes = 256 / Wv;
    const int tl = threadIdx.x / Wv, v0 = threadIdx.x - tl * Wv;
    uint16_t* yrow = y + ((int64_t)img * Ho + yo) * Wo * C;
    if (yy < 0 || yy >= H) {                     // a border row of the padded layout (the whole workgroup leaves here)
        if (tl >= lanes) return;
        for (int v = v0; v < cv; v += Wv)
            for (int xo = tl; xo < Wo; xo += lanes) *(u32x4_t*)(yrow + (int64_t)xo * C + v * 8) = u32x4_t{0u, 0u, 0u, 0u};
        return;
    }
    {
        const int g = threadIdx.x & 31, sl0 = threadIdx.x >> 5;
        double s = 0.0, q = 0.0;
        if (g < groups)
            for (int sl = sl0; sl < nslab; sl += 8) {
                const float* p = part + (((int64_t)img * nslab + sl) * groups + g) * 2;
                s += p[0]; q += p[1];
            }
        gn_sum[sl0][g][0] = s; gn_sum[sl0][g][1] = q;
        __syncthreads();
        if (threadIdx.x < groups) {
            s = 0.0; q = 0.0;
#pragma unroll
            for (int l = 0; l < 8; ++l) { s += gn_sum[l][threadIdx.x][0]; q += gn_sum[l][threadIdx.x][1]; }
            const double mean = s / count;
            double var = q / count - mean * mean;
            if (var < 0.0) var = 0.0;
            stats[threadIdx.x * 2] = (float)mean;
            stats[threadIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
    }
    if (tl >= lanes) return;
    const uint16_t* xrow = x + tok_row(img, yy * W, H, W, in_pad) * C;       // pixel (yy, 0); a row's pixels are consecutive rows
    for (int v = v0; v < cv; v += Wv) {
        float sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = v * 8 + i, g = c / cpg;
            const float mean = stats[g * 2], rstd = stats[g * 2 + 1];
            sc[i] = rstd * gamma[c];
            sh[i] = beta[c] + ((tadd ? tadd[(int64_t)img * ld_t + c] : 0.f) - mean) * sc[i];
        }
        for (int xo = tl; xo < Wo; xo += lanes) {
            const int xx = out_pad ? xo - 1 : xo;
            u32x4_t o = u32x4_t{0u, 0u, 0u, 0u};
            if (xx >= 0 && xx < W) {
                float f[8];
                unpack8(*(const u32x4_t*)(xrow + (int64_t)xx * C + v * 8), f);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float val = fmaf(f[i], sc[i], sh[i]);
                    f[i] = silu ? silu_f(val) : val;
                }
                o = pack8(f);
            }
            *(u32x4_t*)(yrow + (int64_t)xo * C + v * 8) = o;
        }
    }
}

// ---- LayerNorm over channels, bf16 -> bf16, one wave per row, C % 8 == 0, C <= 1536.
// With `add` the row is first replaced by bf16(x + add), which is also written to `sum_out` (the residual stream a
// transformer block carries on): the residual add and the LayerNorm that follows it in ONE pass, bit-identical to
// add_bf16_kernel followed by this kernel (the sum is rounded to bf16 before it is normalised, as the stored one is).
__global__ __launch_bounds__(256) void ln_bf16_kernel(const uint16_t* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, uint16_t* __restrict__ y, int64_t rows,
                                                      int C, float eps, const uint16_t* __restrict__ add,
                                                      uint16_t* __restrict__ sum_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int cv = C >> 3;
    const u32x4_t* xr = (const u32x4_t*)(x + row * C);
    float f[3][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v = lane + i * 64;
        if (v < cv) {
            unpack8(xr[v], f[i]);
            if (add) {
                float a8[8];
                unpack8(((const u32x4_t*)(add + row * C))[v], a8);
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] += f[i][e];
                const u32x4_t pk = pack8(a8);
                ((u32x4_t*)(sum_out + row * C))[v] = pk;
                unpack8(pk, f[i]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v = lane + i * 64;
        if (v < cv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    u32x4_t* yr = (u32x4_t*)(y + row * C);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v = lane + i * 64;
        if (v < cv) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f[i][e] - mean) * rstd * g[v * 8 + e] + b[v * 8 + e];
            yr[v] = pack8(o);
        }
    }
}

// ---- GEGLU: in [rows, 2 * Ch] (value | gate) -> out [rows, Ch] = value * gelu(gate), exact (erf) GELU
__global__ __launch_bounds__(256) void geglu_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t rows,
                                                    int Ch) {
    const int cv = Ch >> 3;
    const int64_t total = rows * cv;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % cv);
        const int64_t r = t / cv;
        float a[8], g[8];
        unpack8(*(const u32x4_t*)(in + r * 2 * Ch + c * 8), a);
        unpack8(*(const u32x4_t*)(in + r * 2 * Ch + Ch + c * 8), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] *= 0.5f * g[i] * (1.0f + erff(g[i] * 0.70710678118654752f));
        *(u32x4_t*)(out + r * Ch + c * 8) = pack8(a);
    }
}

// ---- out = a + b (bf16, n8 pieces of 8)
__global__ __launch_bounds__(256) void add_bf16_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                       uint16_t* __restrict__ out, int64_t n8) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n8; t += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        unpack8(((const u32x4_t*)a)[t], x);
        unpack8(((const u32x4_t*)b)[t], y);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] += y[i];
        ((u32x4_t*)out)[t] = pack8(x);
    }
}

// ---- out[dense] = a[dense] + b[PADDED layout] (the residual add behind a resnet's second convolution)
__global__ __launch_bounds__(256) void add_padded_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                         uint16_t* __restrict__ out, int n, int H, int W, int C) {
    const int cv = C >> 3;
    const int64_t total = (int64_t)n * H * W * cv;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % cv);
        const int64_t tok = t / cv;
        const int img = (int)(tok / ((int64_t)H * W));
        const int p = (int)(tok - (int64_t)img * H * W);
        float x[8], y[8];
        unpack8(((const u32x4_t*)a)[t], x);
        unpack8(*(const u32x4_t*)(b + tok_row(img, p, H, W, 1) * C + c * 8), y);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] += y[i];
        ((u32x4_t*)out)[t] = pack8(x);
    }
}

// ---- copy between the dense and the padded layout (borders of a padded output are zeroed); up: the output is the
// nearest-2x upsampling of the input (H, W = the OUTPUT's extent; Upsample2D in front of a 9-plane convolution)
__global__ __launch_bounds__(256) void relayout_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int n, int H,
                                                       int W, int C, int in_pad, int out_pad, int up) {
    const int cv = C >> 3;
    const int Ho = out_pad ? H + 2 : H, Wo = out_pad ? W + 2 : W;
    const int Hi = up ? H >> 1 : H, Wi = up ? W >> 1 : W;
    const int64_t total = (int64_t)n * Ho * Wo * cv;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % cv);
        const int64_t row = t / cv;
        const int img = (int)(row / ((int64_t)Ho * Wo));
        const int r = (int)(row - (int64_t)img * Ho * Wo);
        int yy = r / Wo, xx = r - yy * Wo;
        if (out_pad) { --yy; --xx; }
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const int ys = up ? yy >> 1 : yy, xs = up ? xx >> 1 : xx;
            v = *(const u32x4_t*)(in + tok_row(img, ys * Wi + xs, Hi, Wi, in_pad) * C + c * 8);
        }
        ((u32x4_t*)out)[t] = v;
    }
}

// ---- channel concatenation of two token-major tensors: out[t] = a[t] | b[t]
__global__ __launch_bounds__(256) void concat_kernel(const uint16_t* __restrict__ a, int Ca, const uint16_t* __restrict__ b,
                                                     int Cb, uint16_t* __restrict__ out, int64_t tokens) {
    const int cv = (Ca + Cb) >> 3, ca8 = Ca >> 3;
    const int64_t total = tokens * cv;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % cv);
        const int64_t r = t / cv;
        ((u32x4_t*)out)[t] = c < ca8 ? *(const u32x4_t*)(a + r * Ca + c * 8) : *(const u32x4_t*)(b + r * Cb + (c - ca8) * 8);
    }
}

// ---- fp32 -> bf16 with optional SiLU (time-embedding MLP)
__global__ __launch_bounds__(256) void cast_silu_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int64_t n,
                                                        int silu) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const float v = in[t];
        out[t] = f32_to_bf16_bits(silu ? silu_f(v) : v);
    }
}

// ---- fp32 token-major [n * HW, ld] (first C columns) -> fp32 NCHW [n, C, H, W]; out = in * mul + add, optional clamp
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const float* __restrict__ in, int64_t ld, float* __restrict__ out,
                                                             int n, int C, int H, int W, float mul, float add, int clamp01,
                                                             int in_pad) {
    const int HW = H * W;
    const int64_t total = (int64_t)n * C * HW;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int p = (int)(t % HW);
        int64_t r = t / HW;
        const int c = (int)(r % C);
        const int64_t img = r / C;
        float v = in[tok_row(img, p, H, W, in_pad) * ld + c] * mul + add;
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        out[t] = v;
    }
}

// ---- out = bias[c] + sum_ci w[c, ci] * in[n, ci, p] (the VAE's 1x1 post_quant_conv on C <= 8 fp32 channels)
__global__ __launch_bounds__(256) void pointwise_small_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out, int n,
                                                              int C, int HW, float in_scale) {
    const int64_t total = (int64_t)n * C * HW;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int p = (int)(t % HW);
        int64_t r = t / HW;
        const int c = (int)(r % C);
        const int64_t img = r / C;
        float v = bias[c];
        for (int ci = 0; ci < C; ++ci) v += w[c * C + ci] * (in[(img * C + ci) * HW + p] * in_scale);
        out[t] = v;
    }
}

// ---- classifier-free guidance: eps = eu + g * (ec - eu); e = [uncond half | cond half], each n elements
__global__ __launch_bounds__(256) void cfg_kernel(const float* __restrict__ e, float* __restrict__ out, int64_t n, float g) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const float eu = e[t], ec = e[n + t];
        out[t] = eu + g * (ec - eu);
    }
}

// ---- out = cs * sample - ce * (c0 e0 + c1 e1 + c2 e2 + c3 e3)   (the PLMS update; unused e_i may be NULL with c_i = 0)
__global__ __launch_bounds__(256) void lincomb_kernel(float* __restrict__ out, const float* __restrict__ sample, float cs, float ce,
                                                      const float* __restrict__ e0, float c0, const float* __restrict__ e1, float c1,
                                                      const float* __restrict__ e2, float c2, const float* __restrict__ e3, float c3,
                                                      int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        float m = c0 * e0[t];
        if (e1) m += c1 * e1[t];
        if (e2) m += c2 * e2[t];
        if (e3) m += c3 * e3[t];
        out[t] = cs * sample[t] - ce * m;
    }
}

// ---- row softmax of fp32 scores (already scaled) -> bf16 probabilities; one workgroup per row, T <= 16384
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, uint16_t* __restrict__ p, int T, float scale) {
    __shared__ float red[4];
    const float* sr = s + (int64_t)blockIdx.x * T;
    uint16_t* pr = p + (int64_t)blockIdx.x * T;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < T; i += 256) mx = fmaxf(mx, sr[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < T; i += 256) sum += __expf(sr[i] * scale - mx);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x; i < T; i += 256) pr[i] = f32_to_bf16_bits(__expf(sr[i] * scale - mx) * inv);
}

// ---- layout converters of the block-level entry points: fp32 NCHW <-> bf16 token-major
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int n, int C,
                                                             int HW) {
    const int64_t total = (int64_t)n * HW * C;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % C);
        int64_t r = t / C;
        const int p = (int)(r % HW);
        const int64_t img = r / HW;
        out[t] = f32_to_bf16_bits(in[(img * C + c) * HW + p]);
    }
}
__global__ __launch_bounds__(256) void tokens_bf16_to_nchw_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int n,
                                                                  int C, int HW) {
    const int64_t total = (int64_t)n * C * HW;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int p = (int)(t % HW);
        int64_t r = t / HW;
        const int c = (int)(r % C);
        const int64_t img = r / C;
        out[t] = bf16_bits_to_f32(in[(img * HW + p) * C + c]);
    }
}

// ---- sinusoidal timestep embedding (flip_sin_to_cos, freq_shift 0): out bf16 [n, dim] = [cos | sin](t * 10000^(-i / half))
__global__ __launch_bounds__(256) void timestep_embed_kernel(uint16_t* __restrict__ out, int n, int dim, float t) {
    const int half = dim >> 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n * dim; i += gridDim.x * 256) {
        const int k = i % dim;
        const int j = k < half ? k : k - half;
        const float ang = t * expf(-9.210340371976184f * (float)j / (float)half);
        out[i] = f32_to_bf16_bits(k < half ? cosf(ang) : sinf(ang));
    }
}

// ---- antialiased resize + centre crop + mean / std normalisation of fp32 NCHW images (values in [0, 1]): the CLIP
// `preprocess` (bicubic, a = -0.5) or the reference's torchvision Resize (bilinear) applied to generated references
// without leaving the GPU.  PIL / torch(antialias=True) semantics: the filter support is scaled by the downscale factor,
// weights are normalised per output pixel, separable.  One thread per output element; the two 1-D weight sets
// are recomputed per thread (support <= 2 * 2 * scale + 2 taps: a few dozen).
template <int CUBIC>
__device__ __forceinline__ float resize_filter(float x) {
    x = fabsf(x);
    if (CUBIC) {
        const float a = -0.5f;
        if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
        if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
        return 0.f;
    }
    return x < 1.f ? 1.f - x : 0.f;
}

template <int CUBIC>
__global__ __launch_bounds__(256) void resize_norm_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int H, int W,
                                                          int Hr, int Wr, int oy, int ox, int S, float m0, float m1, float m2,
                                                          float s0, float s1, float s2) {
    const int64_t total = (int64_t)n * 3 * S * S;
    const float sy = (float)H / (float)Hr, sx = (float)W / (float)Wr;
    const float fy = sy > 1.f ? sy : 1.f, fx = sx > 1.f ? sx : 1.f;          // filter scale (antialias only when shrinking)
    const float supy = (CUBIC ? 2.f : 1.f) * fy, supx = (CUBIC ? 2.f : 1.f) * fx;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % S);
        int64_t r = t / S;
        const int y = (int)(r % S);
        r /= S;
        const int c = (int)(r % 3);
        const int64_t img = r / 3;
        const float cy = ((float)(y + oy) + 0.5f) * sy, cx = ((float)(x + ox) + 0.5f) * sx;
        int y0 = (int)(cy - supy + 0.5f), y1 = (int)(cy + supy + 0.5f);
        int x0 = (int)(cx - supx + 0.5f), x1 = (int)(cx + supx + 0.5f);
        y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
        y1 = y1 > H ? H : y1; x1 = x1 > W ? W : x1;
        float wys = 0.f, wxs = 0.f;
        for (int yy = y0; yy < y1; ++yy) wys += resize_filter<CUBIC>(((float)yy + 0.5f - cy) / fy);
        for (int xx = x0; xx < x1; ++xx) wxs += resize_filter<CUBIC>(((float)xx + 0.5f - cx) / fx);
        const float* p = in + (img * 3 + c) * (int64_t)H * W;
        float acc = 0.f;
        for (int yy = y0; yy < y1; ++yy) {
            const float wy = resize_filter<CUBIC>(((float)yy + 0.5f - cy) / fy);
            float row = 0.f;
            for (int xx = x0; xx < x1; ++xx) row += resize_filter<CUBIC>(((float)xx + 0.5f - cx) / fx) * p[(int64_t)yy * W + xx];
            acc += wy * row;
        }
        acc /= (wys * wxs);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        out[t] = (acc - mean) / sd;
    }
}

inline int grid_for(int64_t total) {
    const int64_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

hipError_t sd_im2col3x3(const uint16_t* in, uint16_t* out, int n, int Hi, int Wi, int C, int stride, int up, hipStream_t st) {
    if (C % 8 != 0 || (stride != 1 && stride != 2) || (up && stride != 1)) return hipErrorInvalidValue;
    const int Hs = up ? 2 * Hi : Hi, Ws = up ? 2 * Wi : Wi;
    const int Ho = (Hs - 1) / stride + 1, Wo = (Ws - 1) / stride + 1;       // padding 1, kernel 3
    const int64_t total = (int64_t)n * Ho * Wo * 9 * (C >> 3);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, out, n, Hi, Wi, C, Ho, Wo, stride, up);
    return hipGetLastError();
}

hipError_t sd_im2col_in(const float* in, uint16_t* out, int n, int Cin, int H, int W, int Kp, float scale, hipStream_t st) {
    if (9 * Cin > Kp) return hipErrorInvalidValue;
    hipLaunchKernelGGL(im2col_in_kernel, dim3(grid_for((int64_t)n * H * W * Kp)), dim3(256), 0, st, in, out, n, Cin, H, W, Kp, scale);
    return hipGetLastError();
}

// ws: >= n * nslab * groups * 2 + n * groups * 2 floats (sd_groupnorm_ws_floats)
// (sd_groupnorm_ws_floats, gn_slab_tokens: host_plan.hpp)

// in_pad / out_pad: the input / output is in the padded layout (tok_row); the output's border rows are zeroed
// tokens per statistics slab: small slabs = many workgroups (the pass is latency-bound on few), at most 1024 slabs per image


hipError_t sd_groupnorm(const uint16_t* x, const float* tadd, int64_t ld_t, const float* gamma, const float* beta, uint16_t* y,
                        int n, int H, int W, int C, int groups, float eps, int silu, int in_pad, int out_pad, float* ws,
                        hipStream_t st) {
    if (groups > 32 || C % groups != 0 || C % 8 != 0 || C > 4096) return hipErrorInvalidValue;
    const int HW = H * W, slab = gn_slab_tokens(HW);
    const int nslab = (HW + slab - 1) / slab;
    float* part = ws;
    const int cv = C >> 3, Wv = cv < 256 ? cv : 256, lanes = 256 / Wv;
    const size_t lds = (size_t)lanes * C * 2 * 4;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(n * nslab), dim3(256), lds, st, x, tadd, ld_t, part, H, W, C, groups, slab, nslab, in_pad);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(n * (out_pad ? H + 2 : H)), dim3(256), 0, st, x, tadd, ld_t, part, nslab,
                       (double)HW * (C / groups), eps, gamma, beta, y, n, H, W, C, groups, silu, in_pad, out_pad);
    return hipGetLastError();
}

// H, W: the OUTPUT's extent (up: the input is (H / 2) x (W / 2))
hipError_t sd_relayout(const uint16_t* in, uint16_t* out, int n, int H, int W, int C, int in_pad, int out_pad, int up, hipStream_t st) {
    if (C % 8 != 0 || (up && ((H | W) & 1))) return hipErrorInvalidValue;
    const int64_t rows = out_pad ? (int64_t)n * (H + 2) * (W + 2) : (int64_t)n * H * W;
    hipLaunchKernelGGL(relayout_kernel, dim3(grid_for(rows * (C >> 3))), dim3(256), 0, st, in, out, n, H, W, C, in_pad, out_pad, up);
    return hipGetLastError();
}

hipError_t sd_add_padded(const uint16_t* a, const uint16_t* b_padded, uint16_t* out, int n, int H, int W, int C, hipStream_t st) {
    if (C % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_padded_kernel, dim3(grid_for((int64_t)n * H * W * (C >> 3))), dim3(256), 0, st, a, b_padded, out, n, H, W, C);
    return hipGetLastError();
}

hipError_t sd_layernorm_bf16(const uint16_t* x, const float* g, const float* b, uint16_t* y, int64_t rows, int C, float eps,
                             hipStream_t st, const uint16_t* add, uint16_t* sum_out) {
    if (C % 8 != 0 || C > 1536 || ((add != nullptr) != (sum_out != nullptr))) return hipErrorInvalidValue;
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(ln_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, g, b, y, rows, C, eps, add, sum_out);
    return hipGetLastError();
}

hipError_t sd_geglu(const uint16_t* in, uint16_t* out, int64_t rows, int Ch, hipStream_t st) {
    if (Ch % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(geglu_kernel, dim3(grid_for(rows * (Ch >> 3))), dim3(256), 0, st, in, out, rows, Ch);
    return hipGetLastError();
}

hipError_t sd_add_bf16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, hipStream_t st) {
    if (n % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, st, a, b, out, n >> 3);
    return hipGetLastError();
}

hipError_t sd_concat(const uint16_t* a, int Ca, const uint16_t* b, int Cb, uint16_t* out, int64_t tokens, hipStream_t st) {
    if (Ca % 8 != 0 || Cb % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(concat_kernel, dim3(grid_for(tokens * ((Ca + Cb) >> 3))), dim3(256), 0, st, a, Ca, b, Cb, out, tokens);
    return hipGetLastError();
}

hipError_t sd_cast_silu(const float* in, uint16_t* out, int64_t n, int silu, hipStream_t st) {
    hipLaunchKernelGGL(cast_silu_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, n, silu);
    return hipGetLastError();
}

hipError_t sd_tokens_to_nchw(const float* in, int64_t ld, float* out, int n, int C, int H, int W, float mul, float add, int clamp01,
                             int in_pad, hipStream_t st) {
    hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3(grid_for((int64_t)n * C * H * W)), dim3(256), 0, st, in, ld, out, n, C, H, W, mul,
                       add, clamp01, in_pad);
    return hipGetLastError();
}

hipError_t sd_pointwise_small(const float* in, const float* w, const float* bias, float* out, int n, int C, int HW, float in_scale,
                              hipStream_t st) {
    if (C > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pointwise_small_kernel, dim3(grid_for((int64_t)n * C * HW)), dim3(256), 0, st, in, w, bias, out, n, C, HW,
                       in_scale);
    return hipGetLastError();
}

hipError_t sd_cfg(const float* e, float* out, int64_t n, float g, hipStream_t st) {
    hipLaunchKernelGGL(cfg_kernel, dim3(grid_for(n)), dim3(256), 0, st, e, out, n, g);
    return hipGetLastError();
}

hipError_t sd_lincomb(float* out, const float* sample, float cs, float ce, const float* e0, float c0, const float* e1, float c1,
                      const float* e2, float c2, const float* e3, float c3, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, sample, cs, ce, e0, c0, e1, c1, e2, c2, e3, c3, n);
    return hipGetLastError();
}

hipError_t sd_softmax_rows(const float* s, uint16_t* p, int64_t rows, int T, float scale, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, p, T, scale);
    return hipGetLastError();
}

hipError_t sd_nchw_to_tokens(const float* in, uint16_t* out, int n, int C, int HW, hipStream_t st) {
    hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3(grid_for((int64_t)n * C * HW)), dim3(256), 0, st, in, out, n, C, HW);
    return hipGetLastError();
}

hipError_t sd_tokens_bf16_to_nchw(const uint16_t* in, float* out, int n, int C, int HW, hipStream_t st) {
    hipLaunchKernelGGL(tokens_bf16_to_nchw_kernel, dim3(grid_for((int64_t)n * C * HW)), dim3(256), 0, st, in, out, n, C, HW);
    return hipGetLastError();
}

hipError_t sd_timestep_embed(uint16_t* out, int n, int dim, float t, hipStream_t st) {
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(grid_for((int64_t)n * dim)), dim3(256), 0, st, out, n, dim, t);
    return hipGetLastError();
}

// in fp32 [n, 3, H, W] -> out fp32 [n, 3, S, S]: resize to (Hr, Wr), crop at (oy, ox), (v - mean) / std; cubic: bicubic else bilinear
hipError_t sd_resize_norm(const float* in, float* out, int n, int H, int W, int Hr, int Wr, int oy, int ox, int S, int cubic,
                          const float* mean, const float* sd, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (H < 1 || W < 1 || Hr < S || Wr < S || oy < 0 || ox < 0 || oy + S > Hr || ox + S > Wr) return hipErrorInvalidValue;
    const int64_t total = (int64_t)n * 3 * S * S;
    if (cubic)
        hipLaunchKernelGGL(resize_norm_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, in, out, n, H, W, Hr, Wr, oy, ox, S, mean[0],
                           mean[1], mean[2], sd[0], sd[1], sd[2]);
    else
        hipLaunchKernelGGL(resize_norm_kernel<0>, dim3(grid_for(total)), dim3(256), 0, st, in, out, n, H, W, Hr, Wr, oy, ox, S, mean[0],
                           mean[1], mean[2], sd[0], sd[1], sd[2]);
    return hipGetLastError();
}
