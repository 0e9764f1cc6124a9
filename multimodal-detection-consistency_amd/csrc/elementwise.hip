// HBM-bound row kernels of the CLIP towers and the bank path: LayerNorm,
// patch im2col, token assembly, embedding gather, L2 normalise, split-bf16
// planes, bank row gather.  One 64-lane wave per row, 16-byte accesses,
// wave-shuffle reductions (no LDS).
#include "common.hpp"
#include "kernels.hpp"

#define LN_EPS 1e-5f
// Non-temporal hints of the LayerNorm pass (bit mask: 1 = x loads, 2 = x stores, 4 = delta loads, 8 = y stores).  The
// residual stream and the projection output are read ONCE here and re-read only after the next GEMMs have streamed
// > 256 MB through the caches, so they are marked streaming: 7 measured 14.0 -> 12.6 ms of LayerNorm per step against 0
// (13.0 with 5, 13.1 with 15: the normalised rows ARE re-read, by the next GEMM, and stay cacheable).
#ifndef TVC_LN_NT
#define TVC_LN_NT 7
#endif
#define ROWS_PER_BLOCK 4   // 256 threads = 4 waves = 4 rows

// ---------------------------------------------------------------------------
// (residual add +) LayerNorm: x fp32 row (+ delta bf16 row) -> bf16 row (GEMM
// operand).  With `delta` the residual GEMMs stay store-only: the previous
// projection's output is folded into the fp32 residual stream here, in the same
// streaming pass that normalises it (x is written back when `write_x`).
// d % 4 == 0, d <= 1024.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, int64_t x_row_stride,
                                                        const int32_t* __restrict__ row_idx,
                                                        const uint16_t* __restrict__ delta,
                                                        const uint16_t* __restrict__ delta2, int write_x,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b,
                                                        uint16_t* __restrict__ y, int rows, int d,
                                                        int delta_compact, float* __restrict__ xsum_out,
                                                        float* __restrict__ y32) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t src_row = row_idx ? (int64_t)row_idx[row] : (int64_t)row;
    f32x4_t* xr = (f32x4_t*)(x + src_row * x_row_stride);
    // deltas: the same element offset as x, or (delta_compact) [rows, d] in output-row order
    const int64_t doff = delta_compact ? (int64_t)row * d : src_row * x_row_stride;
    const u32x2_t* dr = delta ? (const u32x2_t*)(delta + doff) : nullptr;
    const u32x2_t* dr2 = delta2 ? (const u32x2_t*)(delta2 + doff) : nullptr;
    const int nv = d >> 2;
    f32x4_t v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
#if TVC_LN_NT & 1
            v[i] = __builtin_nontemporal_load(xr + c);
#else
            v[i] = xr[c];
#endif
            if (dr) {
#if TVC_LN_NT & 4
                const u32x2_t dd = __builtin_nontemporal_load(dr + c);
#else
                const u32x2_t dd = dr[c];
#endif
                v[i][0] += __uint_as_float(dd[0] << 16);
                v[i][1] += __uint_as_float(dd[0] & 0xffff0000u);
                v[i][2] += __uint_as_float(dd[1] << 16);
                v[i][3] += __uint_as_float(dd[1] & 0xffff0000u);
            }
            if (dr2) {
                const u32x2_t dd = dr2[c];
                v[i][0] += __uint_as_float(dd[0] << 16);
                v[i][1] += __uint_as_float(dd[0] & 0xffff0000u);
                v[i][2] += __uint_as_float(dd[1] << 16);
                v[i][3] += __uint_as_float(dd[1] & 0xffff0000u);
            }
#if TVC_LN_NT & 2
            if (write_x && (dr || dr2)) __builtin_nontemporal_store(v[i], xr + c);
#else
            if (write_x && (dr || dr2)) xr[c] = v[i];
#endif
            if (xsum_out) ((f32x4_t*)(xsum_out + (int64_t)row * d))[c] = v[i];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float dlt = v[i][t] - mean;
                q += dlt * dlt;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + LN_EPS);
    u32x2_t* yr = (u32x2_t*)(y + (int64_t)row * d);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4_t gg = ((const f32x4_t*)g)[c];
            const f32x4_t bb = ((const f32x4_t*)b)[c];
            f32x4_t o;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = (v[i][t] - mean) * rstd * gg[t] + bb[t];
            if (y32) ((f32x4_t*)(y32 + (int64_t)row * d))[c] = o;      // fp32 copy (hidden-state outputs)
            if (y) {
                u32x2_t pk;
                pk[0] = pack_bf16x2(o[0], o[1]);
                pk[1] = pack_bf16x2(o[2], o[3]);
#if TVC_LN_NT & 8
                __builtin_nontemporal_store(pk, yr + c);
#else
                yr[c] = pk;
#endif
            }
        }
    }
}

hipError_t launch_layernorm(float* x, int64_t x_row_stride, const int32_t* row_idx, const uint16_t* delta,
                            int write_x, const float* g, const float* b, uint16_t* y, int rows, int d,
                            hipStream_t stream, const uint16_t* delta2, int delta_compact, float* xsum_out, float* y32) {
    if (d % 4 != 0 || d > 1024 || rows < 0) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const int grid = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(layernorm_kernel, dim3(grid), dim3(256), 0, stream, x, x_row_stride, row_idx, delta, delta2,
                       write_x, g, b, y, rows, d, delta_compact, xsum_out, y32);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// im2col for the stride=patch conv: pix fp32 [B,3,S,S] -> bf16 [B*P, Kp],
// column order (c, ky, kx) = the conv weight's flatten order; zero padded.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ pix,
                                                     uint16_t* __restrict__ out, int B, int S,
                                                     int patch, int Kp) {
    const int g = S / patch;           // patches per side
    const int P = g * g;
    const int chunks = Kp >> 3;        // 8 columns (16 B) per thread
    const int64_t total = (int64_t)B * P * chunks;
    const int K = 3 * patch * patch;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(t % chunks);
        const int64_t row = t / chunks;
        const int p = (int)(row % P);
        const int bimg = (int)(row / P);
        const int py = p / g, px = p - py * g;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float f[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = ch * 8 + e * 2 + h;
                float val = 0.f;
                if (col < K) {
                    const int c = col / (patch * patch);
                    const int rem = col - c * patch * patch;
                    const int ky = rem / patch, kx = rem - ky * patch;
                    val = pix[(((int64_t)bimg * 3 + c) * S + (py * patch + ky)) * S + (px * patch + kx)];
                }
                f[h] = val;
            }
            w[e] = pack_bf16x2(f[0], f[1]);
        }
        *(u32x4_t*)(out + row * Kp + ch * 8) = u32x4_t{w[0], w[1], w[2], w[3]};
    }
}

// The same through LDS: one workgroup per (image, patch row).  The 3 * patch image rows of that patch row are read as whole
// rows (S floats contiguous: coalesced), converted and scattered into the LDS image of the g output rows, which are then
// written as whole rows (Kp * 2 bytes contiguous).  The per-element kernel above reads 56-byte runs (patch 14) with one
// scalar load per element: 1.75 TB/s against ~4.5 here.  S % 4 == 0 and g * Kp * 2 bytes of LDS (<= 64 KiB) required.
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ pix, uint16_t* __restrict__ out,
                                                          int S, int patch, int Kp) {
    extern __shared__ __attribute__((aligned(16))) char im_smem[];
    uint16_t* tile = (uint16_t*)im_smem;               // [g][Kp]
    const int g = S / patch;
    const int bimg = blockIdx.x / g, py = blockIdx.x - bimg * g;
    const int K = 3 * patch * patch;
    // zero the padding columns
    for (int i = threadIdx.x; i < g * (Kp - K); i += 256) {
        const int px = i / (Kp - K), c = K + (i - px * (Kp - K));
        tile[px * Kp + c] = 0;
    }
    const int s4 = S >> 2;                              // float4 per image row
    const int total = 3 * patch * s4;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int rowi = i / s4, x4 = i - rowi * s4;    // rowi = c * patch + ky
        const int c = rowi / patch, ky = rowi - c * patch;
        const f32x4_t v = __builtin_nontemporal_load(
            (const f32x4_t*)(pix + (((int64_t)bimg * 3 + c) * S + (py * patch + ky)) * S) + x4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x4 * 4 + e;
            const int px = x / patch, kx = x - px * patch;
            tile[px * Kp + c * patch * patch + ky * patch + kx] = f32_to_bf16_bits(v[e]);
        }
    }
    __syncthreads();
    const int chunks = (g * Kp) >> 3;                   // 16-byte pieces of the g contiguous output rows
    u32x4_t* dst = (u32x4_t*)(out + ((int64_t)bimg * g * g + (int64_t)py * g) * Kp);
    for (int i = threadIdx.x; i < chunks; i += 256) dst[i] = ((const u32x4_t*)tile)[i];
}

hipError_t launch_im2col(const float* pix, uint16_t* out, int B, int image, int patch, int Kp,
                         hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const int g = image / patch;
    const size_t lds = (size_t)g * Kp * 2;
    if (image % 4 == 0 && Kp % 8 == 0 && lds <= 64 * 1024) {
        hipLaunchKernelGGL(im2col_rows_kernel, dim3(B * g), dim3(256), lds, stream, pix, out, image, patch, Kp);
        return hipGetLastError();
    }
    const int64_t total = (int64_t)B * g * g * (Kp >> 3);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, stream, pix, out, B, image, patch, Kp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// x[b, t, :] = ln_pre( (t == 0 ? cls : patch_out[b, t-1, :]) + pos[t, :] )
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_lnpre_kernel(const float* __restrict__ patch_out,
                                                             const float* __restrict__ cls,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ b,
                                                             float* __restrict__ x, int B, int T, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= (int64_t)B * T) return;
    const int t = (int)(row % T);
    const int64_t bimg = row / T;
    const f32x4_t* src = (t == 0) ? (const f32x4_t*)cls
                                  : (const f32x4_t*)(patch_out + (bimg * (T - 1) + (t - 1)) * d);
    const f32x4_t* pr = (const f32x4_t*)(pos + (int64_t)t * d);
    const int nv = d >> 2;
    f32x4_t v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[i] = src[c] + pr[c];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dl = v[i][e] - mean;
                q += dl * dl;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + LN_EPS);
    f32x4_t* xr = (f32x4_t*)(x + row * d);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4_t gg = ((const f32x4_t*)g)[c];
            const f32x4_t bb = ((const f32x4_t*)b)[c];
            f32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
            xr[c] = o;
        }
    }
}

hipError_t launch_assemble_lnpre(const float* patch_out, const float* cls, const float* pos,
                                 const float* g, const float* b, float* x, int B, int T, int d,
                                 hipStream_t stream) {
    if (d % 4 != 0 || d > 1024) return hipErrorInvalidValue;
    const int64_t rows = (int64_t)B * T;
    if (rows == 0) return hipSuccess;
    const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL(assemble_lnpre_kernel, dim3(grid), dim3(256), 0, stream, patch_out, cls, pos, g, b, x, B, T, d);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// text lengths: len[n] = argmax_t tok[n, t] + 1 (tokens up to and including EOT;
// later positions cannot influence the pooled output under the causal mask);
// starts = exclusive scan, starts[n_text] = total rows, starts[n_text+1] = max len.
// n_text is a few thousand.
// ---------------------------------------------------------------------------
// Prefix sharing (pfx != nullptr, texts in consecutive groups of G, first of a group = base text): a
// causal tower gives two texts identical hidden states on their common prefix, so text n keeps only
// the rows from its first token that differs from the base text on: starts[] then scans the OWN row
// counts, pfx[n] = shared prefix length, pfx[n_text + n] = packed row of the base text's position 0.
//
// Two kernels: `text_lens_kernel` (one wave per text, coalesced token reads, any number of
// workgroups) parks own[n] in starts[n] and len[n] in lens[n] (= pfx[n_text + n] when sharing);
// `text_scan_kernel` (one workgroup) turns them into the exclusive scan, the maximum length and the
// base rows.  (One workgroup doing both spent 0.8 ms per step waiting on its own token reads.)
__global__ __launch_bounds__(256) void text_lens_kernel(const int32_t* __restrict__ tok, int32_t* __restrict__ starts,
                                                        int32_t* __restrict__ pfx, int32_t* __restrict__ lens,
                                                        int n_text, int ctx, int G) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_text) return;
    // len = position of the first maximum id + 1
    int best = -1, best_t = 0;
    for (int tt = lane; tt < ctx; tt += 64) {
        const int v = tok[(int64_t)n * ctx + tt];
        if (v > best) { best = v; best_t = tt; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int ov = __shfl_xor(best, o, 64);
        const int ot = __shfl_xor(best_t, o, 64);
        if (ov > best || (ov == best && ot < best_t)) { best = ov; best_t = ot; }
    }
    int len = best_t + 1, p = 0;
    if (pfx) {
        const int bn = n / G * G;
        if (bn != n) {
            // length of the base text, then the first position where the two differ
            int bb = -1, bt = 0, mis = 0x7fffffff;
            for (int tt = lane; tt < ctx; tt += 64) {
                const int vb = tok[(int64_t)bn * ctx + tt], vn = tok[(int64_t)n * ctx + tt];
                if (vb > bb) { bb = vb; bt = tt; }
                if (vb != vn && tt < mis) mis = tt;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int ov = __shfl_xor(bb, o, 64);
                const int ot = __shfl_xor(bt, o, 64);
                if (ov > bb || (ov == bb && ot < bt)) { bb = ov; bt = ot; }
                const int om = __shfl_xor(mis, o, 64);
                mis = om < mis ? om : mis;
            }
            p = mis;
            if (p > len) p = len;
            if (p > bt + 1) p = bt + 1;
        }
        if (lane == 0) pfx[n] = p;
    }
    if (lane == 0) { starts[n] = len - p; lens[n] = len; }
}

__global__ __launch_bounds__(1024) void text_scan_kernel(int32_t* __restrict__ starts, int32_t* __restrict__ pfx,
                                                         const int32_t* __restrict__ lens, int n_text, int G) {
    __shared__ int part[1024];
    __shared__ int carry_s;
    __shared__ int maxlen_s;
    const int t = threadIdx.x;
    if (t == 0) { carry_s = 0; maxlen_s = 0; }
    __syncthreads();
    for (int base = 0; base < n_text; base += 1024) {
        const int n = base + t;
        int own = 0;
        if (n < n_text) {
            own = starts[n];
            atomicMax(&maxlen_s, lens[n]);                      // attention length = prefix + own rows
        }
        part[t] = own;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = (t >= o) ? part[t - o] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        const int carry = carry_s;
        if (n < n_text) starts[n] = carry + part[t] - own;
        __syncthreads();
        if (t == 1023) carry_s = carry + part[1023];
        __syncthreads();
    }
    if (t == 0) { starts[n_text] = carry_s; starts[n_text + 1] = maxlen_s; }
    if (pfx) {
        __syncthreads();
        for (int n = t; n < n_text; n += 1024) pfx[n_text + n] = starts[n / G * G];
    }
}

// lens_ws: int32 [n_text] scratch (ignored with pfx, whose second half is used)
hipError_t launch_text_lens_scan(const int32_t* tok, int32_t* starts, int32_t* pfx, int n_text, int ctx, int G,
                                 hipStream_t stream, int32_t* lens_ws) {
    if (pfx && G < 2) return hipErrorInvalidValue;
    int32_t* lens = pfx ? pfx + n_text : lens_ws;
    if (!lens) return hipErrorInvalidValue;
    hipLaunchKernelGGL(text_lens_kernel, dim3((n_text + 3) / 4), dim3(256), 0, stream, tok, starts, pfx, lens, n_text,
                       ctx, G);
    hipLaunchKernelGGL(text_scan_kernel, dim3(1), dim3(1024), 0, stream, starts, pfx, lens, n_text, G);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// text embedding: x[row(n, t), :] = tok_emb[tok[n, t], :] + pos[t, :];
// eot_row[n] = row of the arg-max id (first maximum, as torch.argmax).
// Dense rows (starts == nullptr): row = n*ctx + t.  Packed rows: row =
// starts[n] + t for t < len[n] only.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ tok,
                                                         const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos,
                                                         float* __restrict__ x,
                                                         int32_t* __restrict__ eot_row,
                                                         const int32_t* __restrict__ starts,
                                                         const int32_t* __restrict__ pfx, int n_text,
                                                         int ctx, int d, int vocab) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= (int64_t)n_text * ctx) return;
    const int t = (int)(row % ctx);
    const int64_t n = row / ctx;
    int64_t out_row = row;
    if (starts) {
        const int s0 = starts[n], own = starts[n + 1] - s0;
        const int p = pfx ? pfx[n] : 0;
        // the EOT row: the last own row, or (text identical to its base up to EOT) the base's row
        if (t == 0 && lane == 0) eot_row[n] = own > 0 ? s0 + own - 1 : pfx[n_text + n] + p - 1;
        if (t < p || t >= p + own) return;
        out_row = s0 + (t - p);
    }
    int id = tok[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f32x4_t* er = (const f32x4_t*)(tok_emb + (int64_t)id * d);
    const f32x4_t* pr = (const f32x4_t*)(pos + (int64_t)t * d);
    f32x4_t* xr = (f32x4_t*)(x + out_row * d);
    for (int c = lane; c < (d >> 2); c += 64) xr[c] = er[c] + pr[c];
    if (t == 0 && !starts) {
        // arg-max over the ctx ids of text n; ties -> lowest position
        int best = -1, best_t = 0;
        for (int tt = lane; tt < ctx; tt += 64) {
            const int v = tok[n * ctx + tt];
            if (v > best) { best = v; best_t = tt; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(best, o, 64);
            const int ot = __shfl_xor(best_t, o, 64);
            if (ov > best || (ov == best && ot < best_t)) { best = ov; best_t = ot; }
        }
        if (lane == 0) eot_row[n] = (int32_t)(n * ctx + best_t);
    }
}

hipError_t launch_text_embed(const int32_t* tok, const float* tok_emb, const float* pos, float* x,
                             int32_t* eot_row, const int32_t* starts, int n_text, int ctx, int d, int vocab,
                             hipStream_t stream, const int32_t* pfx) {
    if (d % 4 != 0) return hipErrorInvalidValue;
    const int64_t rows = (int64_t)n_text * ctx;
    if (rows == 0) return hipSuccess;
    const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL(text_embed_kernel, dim3(grid), dim3(256), 0, stream, tok, tok_emb, pos, x, eot_row, starts,
                       pfx, n_text, ctx, d, vocab);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// in-place L2 normalise of fp32 rows: x / ||x||  (no epsilon, as
// `x / x.norm(dim=-1, keepdim=True)` -- retrieval_ref.py:243)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + (int64_t)row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c] * xr[c];
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int c = lane; c < d; c += 64) xr[c] *= inv;
}

hipError_t launch_l2norm_rows(float* x, int rows, int d, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    const int grid = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(grid), dim3(256), 0, stream, x, rows, d);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// exact GELU 0.5 x (1 + erf(x / sqrt 2)) in place (towers with TVC_ACT_GELU: n % 8 == 0 / n % 4 == 0 -- widths are
// multiples of 64)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__global__ __launch_bounds__(256) void gelu_erf_bf16_kernel(uint16_t* __restrict__ x, int64_t n8) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (int64_t)gridDim.x * blockDim.x) {
        u32x4_t v = ((u32x4_t*)x)[t];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = pack_bf16x2(gelu_erf(__uint_as_float(v[e] << 16)), gelu_erf(__uint_as_float(v[e] & 0xffff0000u)));
        ((u32x4_t*)x)[t] = v;
    }
}
__global__ __launch_bounds__(256) void gelu_erf_f32_kernel(float* __restrict__ x, int64_t n4) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (int64_t)gridDim.x * blockDim.x) {
        f32x4_t v = ((f32x4_t*)x)[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        ((f32x4_t*)x)[t] = v;
    }
}
hipError_t launch_gelu_erf_bf16(uint16_t* x, int64_t n, hipStream_t stream) {
    if (n % 8 != 0 || n < 0) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(gelu_erf_bf16_kernel, dim3((int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384)), dim3(256), 0, stream, x, n8);
    return hipGetLastError();
}
hipError_t launch_gelu_erf_f32(float* x, int64_t n, hipStream_t stream) {
    if (n % 4 != 0 || n < 0) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(gelu_erf_f32_kernel, dim3((int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384)), dim3(256), 0, stream, x, n4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// split-bf16 planes: out[r, 0:d] = bf16(x), out[r, d:2d] = bf16(x - hi)
// (planes == 1: hi only).  x ~= hi + lo to ~2^-17 relative.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x,
                                                           uint16_t* __restrict__ out, int64_t rows,
                                                           int d, int planes) {
    const int nv = d >> 2;
    const int64_t total = rows * nv;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / nv;
        const int c = (int)(t - r * nv);
        const f32x4_t v = ((const f32x4_t*)(x + r * d))[c];
        float hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = bf16_bits_to_f32(f32_to_bf16_bits(v[e]));
            lo[e] = v[e] - hi[e];
        }
        uint16_t* o = out + r * (int64_t)(planes * d) + c * 4;
        *(u32x2_t*)o = u32x2_t{pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
        if (planes > 1) *(u32x2_t*)(o + d) = u32x2_t{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3])};
    }
}

hipError_t launch_split_planes(const float* x, uint16_t* out, int64_t rows, int d, int planes,
                               hipStream_t stream) {
    if (d % 4 != 0 || planes < 1 || planes > 2) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const int64_t total = rows * (d >> 2);
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, stream, x, out, rows, d, planes);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// bank row gather: out[n, :] = fp32(bank[idx[n] - idx_offset]) (hi + lo)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* __restrict__ bank, int64_t ld,
                                                          int planes, int D, int64_t R,
                                                          const int32_t* __restrict__ idx,
                                                          int64_t idx_offset, int n,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t src = (int64_t)idx[row] - idx_offset;
    float* o = out + (int64_t)row * D;
    if (idx[row] < 0 || src < 0 || src >= R) {
        for (int c = lane; c < D; c += 64) o[c] = 0.f;
        return;
    }
    const uint16_t* br = bank + src * ld;
    for (int c = lane; c < D; c += 64) {
        float v = bf16_bits_to_f32(br[c]);
        if (planes > 1) v += bf16_bits_to_f32(br[D + c]);
        o[c] = v;
    }
}

hipError_t launch_gather_rows(const uint16_t* bank, int64_t ld, int planes, int D, int64_t R,
                              const int32_t* idx, int64_t idx_offset, int n, float* out,
                              hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int grid = (n + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, stream, bank, ld, planes, D, R, idx, idx_offset, n, out);
    return hipGetLastError();
}
