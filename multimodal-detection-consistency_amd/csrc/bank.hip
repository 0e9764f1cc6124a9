// K5: exact top-k inner-product search of query rows against the reference
// bank, fused with per-row moments.  The [M, R] similarity matrix is never
// written: it lives only in MFMA accumulators.
//
//   pass 0  a strided SAMPLE of the bank (n_sample rows) is multiplied with the
//           query planes by the dense GEMM (fp32 store) and `kth_bound` turns
//           each query's sample row into tau[q] = a LOWER bound of its final
//           k-th best similarity (the k-th largest of 256 disjoint group
//           maxima: k distinct sample elements are >= it).
//   pass 1  `bank_search_kernel`: persistent workgroups own (query tile, bank
//           chunk); per 256x256 tile they run the GEMM main loop (bank rows on
//           the MFMA row dimension, queries on the lane dimension, so every
//           lane filters against its OWN query's tau), accumulate sum / sumsq /
//           max / count in registers and append the few survivors (v > tau) to
//           a per-(chunk, query) list: slot from an LDS counter, no global
//           atomics.
//   pass 2  `bank_select_kernel`: one workgroup (1 024 threads) per query gathers its lists
//           (~0.3-2k entries) into LDS, re-scores them exactly in the fast form and extracts the k
//           best, ordered by (similarity desc, index asc); reduces the moments over chunks.
//
//   small query batches (M <= 64, the reference's own call pattern): pass 1 is `bank_filter_skinny_kernel` (the bank
//           streamed once from HBM into MFMA operand registers) and, over a bf16 bank, pass 0 is
//           `bank_sample_skinny_kernel` + `kth_groups_kernel` (the sample's 256 group maxima straight from the matrix
//           pipe: no [M, n_sample] similarities).
//
//   fast form (no moments requested): pass 1 multiplies ONE product (bank hi x query hi) and
//           filters against tau - margin[q], margin = a Cauchy-Schwarz bound of the dropped
//           products (max bank-row norms x the query's plane norms), so every row whose exact
//           similarity exceeds tau is still listed; pass 2 re-scores the listed rows in fp32
//           (bank hi + lo against the fp32 query) before selecting.  Same result set, half /
//           a third of the MFMA work.
//
// Precision: queries are split into bf16 (hi, lo) planes, so with a bf16 bank
// the products are exact and the result is the fp32-accumulated cosine of the
// stored values (|err| ~ 1e-6); an fp32 bank is split the same way and uses the
// three products hi.hi + lo.hi + hi.lo.
#include "gemm_core.hpp"
#include "gemm_ring4.hpp"
#include "kernels.hpp"
#include <cstdlib>
#include <mutex>

#define BANK_POOL 6144
#define BANK_MAX_RESCORE_D 2048   // fp32 query row staged in LDS by the re-scoring select
#define BANK_LDS_BYTES (GEMM_LDS_BYTES + 256 * 4 + 2 * 256 * 4 * 4)

struct Cand {
    float v;
    int32_t idx;
};

// (bank_plan: host_plan.hpp)

// tau[q] = (k-th largest of 256 group maxima of the query's sample similarities) minus a safety margin
// FILTER form: tau is lowered by the bound of what the one-product pass leaves out,
//   |b.q - bhi.qhi| <= |bhi||qlo| + |blo||qhi| + |blo||qlo|   (bank_bounds = max |bhi|, max |blo|)
// Shared tail of the two tau kernels: thread t < 256 brings group t's maximum `m`; needs blockDim.x >= 256.
__device__ __forceinline__ void kth_finish(float m, int q, int k, float* __restrict__ tau, const uint16_t* __restrict__ qplanes,
                                           int D, const float* __restrict__ bank_bounds, float* gm, float (*nrm)[4]) {
    const int t = threadIdx.x;
    float margin = 0.f;
    if (qplanes) {
        float h2 = 0.f, l2 = 0.f;
        if (t < 256) {
            const uint16_t* qr = qplanes + (int64_t)q * 2 * D;
            for (int c = t; c < D; c += 256) {
                const float h = bf16_bits_to_f32(qr[c]), l = bf16_bits_to_f32(qr[D + c]);
                h2 = fmaf(h, h, h2); l2 = fmaf(l, l, l2);
            }
            h2 = wave_sum(h2); l2 = wave_sum(l2);
            if ((t & 63) == 0) { nrm[0][t >> 6] = h2; nrm[1][t >> 6] = l2; }
        }
        __syncthreads();
        const float qh = sqrtf(nrm[0][0] + nrm[0][1] + nrm[0][2] + nrm[0][3]);
        const float ql = sqrtf(nrm[1][0] + nrm[1][1] + nrm[1][2] + nrm[1][3]);
        const float bh = bank_bounds[0], bl = bank_bounds[1];
        margin = (bh * ql + bl * qh + bl * ql) * 1.001f + 4e-6f * (1.f + bh * qh);
    }
    if (t < 256) gm[t] = m;
    __syncthreads();
    if (t >= 256) return;
    int rank = 0;
    for (int u = 0; u < 256; ++u) {
        const float o = gm[u];
        rank += (o > m) || (o == m && u < t);
    }
    const int keff = k > BANK_KEFF ? k : BANK_KEFF;
    if (rank == keff - 1) {
        float v = m;
        if (v > -INFINITY) v = v - 1e-6f - 1e-6f * fabsf(v) - margin;
        // a NaN / inf margin (NaN or inf bank rows reach bank_bounds) must list every finite row:
        // the lists overflow, tvc_bank_status reports it and the brute-force path takes over
        if (!(v == v)) v = -INFINITY;
        tau[q] = v;
    }
}

// s0 [M, n_sample] (the dense sample GEMM's output) -> tau.  1 024 threads scan the row with 16-byte loads, four in flight
// per thread (one workgroup of 256 threads with scalar loads took 44-51 us per launch whatever M was: 245 dependent
// round trips); thread t's elements belong to group t & 255 -- any partition into 256 disjoint groups gives a valid bound.
__global__ __launch_bounds__(1024) void kth_bound_kernel(const float* __restrict__ s0, int n_sample,
                                                         int k, float* __restrict__ tau,
                                                         const uint16_t* __restrict__ qplanes, int D,
                                                         const float* __restrict__ bank_bounds) {
    __shared__ float gm[256];
    __shared__ float part[3][256];
    __shared__ float nrm[2][4];
    const int q = blockIdx.x, t = threadIdx.x;
    const float* row = s0 + (int64_t)q * n_sample;
    float m = -INFINITY;
    if ((n_sample & 3) == 0) {
        const f32x4_t* row4 = (const f32x4_t*)row;
        const int n4 = n_sample >> 2;
        int i = t;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            const f32x4_t a = row4[i], b = row4[i + 1024], c = row4[i + 2048], d = row4[i + 3072];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])), fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]))));
        }
        for (; i < n4; i += 1024) {
            const f32x4_t a = row4[i];
            m = fmaxf(m, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
        }
    } else {
        for (int i = t; i < n_sample; i += 1024) m = fmaxf(m, row[i]);
    }
    if (t >= 256) part[(t >> 8) - 1][t & 255] = m;
    __syncthreads();
    if (t < 256) m = fmaxf(fmaxf(m, part[0][t]), fmaxf(part[1][t], part[2][t]));
    kth_finish(m, q, k, tau, qplanes, D, bank_bounds, gm, nrm);
}

// gmax [M, 256] (bank_sample_skinny_kernel's group maxima) -> tau; also clears the search's overflow flag
__global__ __launch_bounds__(256) void kth_groups_kernel(const float* __restrict__ gmax, int k, float* __restrict__ tau,
                                                         const uint16_t* __restrict__ qplanes, int D,
                                                         const float* __restrict__ bank_bounds, int32_t* __restrict__ overflow) {
    __shared__ float gm[256];
    __shared__ float nrm[2][4];
    const int q = blockIdx.x;
    if (q == 0 && threadIdx.x == 0) *overflow = 0;
    kth_finish(gmax[(int64_t)q * 256 + threadIdx.x], q, k, tau, qplanes, D, bank_bounds, gm, nrm);
}

// max over rows of |hi plane| and |lo plane| (squared, as ordered uint bit patterns)
__global__ __launch_bounds__(256) void bank_bounds_kernel(const uint16_t* __restrict__ bank, int64_t ld,
                                                          int planes, int D, int64_t R,
                                                          uint32_t* __restrict__ out_sq) {
    const int lane = threadIdx.x & 63;
    float mh = 0.f, ml = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (int64_t)gridDim.x * 4) {
        const uint16_t* br = bank + r * ld;
        float h2 = 0.f, l2 = 0.f;
        for (int c = lane * 8; c < D; c += 512) {
            const u32x4_t h = *(const u32x4_t*)(br + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __uint_as_float(h[e] << 16), b = __uint_as_float(h[e] & 0xffff0000u);
                h2 = fmaf(a, a, fmaf(b, b, h2));
            }
            if (planes > 1) {
                const u32x4_t l = *(const u32x4_t*)(br + D + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = __uint_as_float(l[e] << 16), b = __uint_as_float(l[e] & 0xffff0000u);
                    l2 = fmaf(a, a, fmaf(b, b, l2));
                }
            }
        }
        mh = fmaxf(mh, wave_sum(h2)); ml = fmaxf(ml, wave_sum(l2));
    }
    if (lane == 0) {
        // NaN/inf rows: the uint order puts them on top, the margin becomes inf/NaN, tau = -inf and the
        // filter lists every finite row (-> overflow flag -> brute-force path); never a wrong result
        atomicMax(out_sq, __float_as_uint(mh));
        atomicMax(out_sq + 1, __float_as_uint(ml));
    }
}
__global__ void bank_bounds_finish_kernel(float* b) {
    b[0] = sqrtf(b[0]) * 1.0001f; b[1] = sqrtf(b[1]) * 1.0001f;
}

hipError_t launch_bank_bounds(const uint16_t* bank, int64_t ld, int planes, int D, int64_t R, float* bounds,
                              hipStream_t stream) {
    hipError_t st = hipMemsetAsync(bounds, 0, 8, stream);
    if (st != hipSuccess || R == 0) return st;
    int64_t grid = (R + 3) / 4;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bank_bounds_kernel, dim3((int)grid), dim3(256), 0, stream, bank, ld, planes, D, R,
                       (uint32_t*)bounds);
    hipLaunchKernelGGL(bank_bounds_finish_kernel, dim3(1), dim3(1), 0, stream, bounds);
    return hipGetLastError();
}

struct BankEpilogue {
    const float* tau;      // [M]
    Cand* cand;            // [S, M, cap]
    int32_t* cand_cnt;     // [S, M]
    float* mom_part;       // [S, M, 4]
    int32_t* overflow;
    int64_t R;
    int64_t idx_offset;
    int M;
    float count_thr;
};

template <bool FULL, bool FILTER>
__device__ __forceinline__ void bank_tile_epilogue(const gemm_acc_t& acc, const BankEpilogue& e,
                                                   int64_t tile_row0, int chunk, int j0, int wm, int wn,
                                                   int lane, const float (&tau)[4], float (&sum)[4],
                                                   float (&sq)[4], float (&mx)[4], float (&cn)[4],
                                                   int* lds_cnt) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int ql = wn * 64 + n * 16 + (lane & 15);
        const int q = j0 + ql;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int64_t row = tile_row0 + wm * 128 + m * 16 + (lane >> 4) * 4;
            f32x4_t v = acc[m][n];
            if (!FULL) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row + r >= e.R) v[r] = -INFINITY;
            }
            const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            if (!FILTER) {
                if (FULL) {
                    sum[n] += (v[0] + v[1]) + (v[2] + v[3]);
                    sq[n] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], sq[n]))));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row + r < e.R) { sum[n] += v[r]; sq[n] = fmaf(v[r], v[r], sq[n]); }
                }
                mx[n] = fmaxf(mx[n], m4);
                if (m4 >= e.count_thr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cn[n] += (v[r] >= e.count_thr) ? 1.f : 0.f;
                }
            }
            // NaN similarities (zero-norm query row, NaN bank row) are never listed: every comparison
            // with a NaN is false, and fmaxf drops NaN operands; tau itself is never NaN (kth_bound_kernel)
            if (m4 > tau[n]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (v[r] > tau[n] && (FULL || row + r < e.R)) {
                        // rare path (a few survivors per lane per bank chunk): keep
                        // its address arithmetic inside the branch, not hoisted into
                        // registers that stay live across the main loop
                        int qlo = ql;
                        asm volatile("" : "+v"(qlo));
                        // padded query lanes (q >= M) re-read the last query row with tau = +inf and never get
                        // here; the guard keeps a stray append out of the lists of (chunk + 1, q - M)
                        if (j0 + qlo < e.M) {
                            const int slot = atomicAdd(&lds_cnt[qlo], 1);
                            if (slot < BANK_CAP) {
                                Cand c;
                                c.v = v[r];
                                c.idx = (int32_t)(row + r + e.idx_offset);
                                e.cand[((int64_t)chunk * e.M + (j0 + qlo)) * BANK_CAP + slot] = c;
                            }
                        }
                    }
                }
            }
        }
    }
}

template <bool FILTER>
__global__ __launch_bounds__(GEMM_THREADS) void bank_search_kernel(GemmOperands g, BankEpilogue e,
                                                                   int nQt, int S, int tiles_per_chunk,
                                                                   int n_bank_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lds_cnt = (int*)(smem + GEMM_LDS_BYTES);
    float* lds_mom = (float*)(smem + GEMM_LDS_BYTES + 256 * 4);
    // Order of the (query tile, bank chunk) items inside an XCD (consecutive `lin` = concurrently
    // resident workgroups): blocks of 4 query tiles x 8 chunks, so the 32 workgroups of a block keep
    // 4 query tiles (1.6 MB of planes) in the XCD's L2 AND read every bank tile four times from L2 for
    // one HBM read (query-tile-slowest order streamed the whole bank once per query tile: 30 GB per
    // launch at cfg 2).  Ragged counts fall back to query tile slowest.
    const int lin = xcd_contiguous(blockIdx.x, nQt * S);
    int qt, chunk;
    if ((nQt & 3) == 0 && (S & 7) == 0) {
        const int blk = lin >> 5, r = lin & 31;
        const int ncg = S >> 3;
        const int qg = blk / ncg, cg = blk - qg * ncg;
        qt = qg * 4 + (r & 3);
        chunk = cg * 8 + (r >> 2);
    } else {
        qt = lin / S; chunk = lin - qt * S;
    }
    const int j0 = qt * GEMM_BN;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    if (threadIdx.x < 256) lds_cnt[threadIdx.x] = 0;
    float tau[4], sum[4], sq[4], mx[4], cn[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int q = j0 + wn * 64 + n * 16 + (lane & 15);
        tau[n] = (q < e.M) ? e.tau[q] : INFINITY;
        sum[n] = 0.f; sq[n] = 0.f; mx[n] = -INFINITY; cn[n] = 0.f;
    }
    __syncthreads();

    const int bt0 = chunk * tiles_per_chunk;
    int bt1 = bt0 + tiles_per_chunk;
    if (bt1 > n_bank_tiles) bt1 = n_bank_tiles;
    for (int bt = bt0; bt < bt1; ++bt) {
        gemm_acc_t acc;
        gemm_zero_acc(acc);
        const int64_t tile_row0 = (int64_t)bt * GEMM_BM;
        gemm_mainloop(acc, g, (int)tile_row0, j0, smem);
        if (tile_row0 + GEMM_BM <= e.R)
            bank_tile_epilogue<true, FILTER>(acc, e, tile_row0, chunk, j0, wm, wn, lane, tau, sum, sq, mx, cn, lds_cnt);
        else
            bank_tile_epilogue<false, FILTER>(acc, e, tile_row0, chunk, j0, wm, wn, lane, tau, sum, sq, mx, cn, lds_cnt);
    }

    if (FILTER) {
        __syncthreads();
        if (threadIdx.x < 256) {
            const int q = j0 + threadIdx.x;
            if (q < e.M) {
                int c = lds_cnt[threadIdx.x];
                if (c > BANK_CAP) { c = BANK_CAP; atomicOr(e.overflow, 1); }
                e.cand_cnt[(int64_t)chunk * e.M + q] = c;
            }
        }
        return;
    }
    // moments: the 4 lane groups of a wave and the 2 wm-waves share a query
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float s = sum[n], s2 = sq[n], m = mx[n], c = cn[n];
        s += __shfl_xor(s, 16, 64);  s += __shfl_xor(s, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        c += __shfl_xor(c, 16, 64);  c += __shfl_xor(c, 32, 64);
        m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
        if ((lane >> 4) == 0) {
            float* p = lds_mom + ((wm * 256) + wn * 64 + n * 16 + (lane & 15)) * 4;
            p[0] = s; p[1] = s2; p[2] = m; p[3] = c;
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int q = j0 + threadIdx.x;
        if (q < e.M) {
            const float* a = lds_mom + threadIdx.x * 4;
            const float* b = lds_mom + (256 + threadIdx.x) * 4;
            float* o = e.mom_part + ((int64_t)chunk * e.M + q) * 4;
            o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = fmaxf(a[2], b[2]); o[3] = a[3] + b[3];
            int c = lds_cnt[threadIdx.x];
            if (c > BANK_CAP) { c = BANK_CAP; atomicOr(e.overflow, 1); }
            e.cand_cnt[(int64_t)chunk * e.M + q] = c;
        }
    }
}

// The filter pass on GEMM form 4 (gemm_ring4.hpp): the workgroup's bank tiles are ONE stream through the LDS-DMA ring --
// counted waits, barrier-staggered ping-pong of the two wave groups, the next tile's K-tiles loading while this tile is
// filtered -- instead of gemm_mainloop's drain-and-barrier per K-tile.  Both operands of this product come out of the L2
// (a query tile is resident, a bank tile is read by four workgroups for one HBM read), where the CU's request path
// delivers 2-3 x what it does for first-touch rows, so the loop is bound by the matrix pipe.  Same products, summed in
// the same order: the listed candidates are identical.  Preconditions (launch_bank_search): the query planes are
// readable up to the next multiple of 256 rows, row pitches are multiples of 128 bytes; the stream takes the FULL bank
// tiles, a ragged last tile goes through gemm_mainloop as before.
__global__ __launch_bounds__(GEMM_THREADS) void bank_filter_ring_kernel(GemmOperands g, BankEpilogue e, int nQt, int S,
                                                                        int tiles_per_chunk, int n_bank_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lds_cnt = (int*)(smem + GEMM_LDS_BYTES);
    const int lin = xcd_contiguous(blockIdx.x, nQt * S);
    int qt, chunk;
    if ((nQt & 3) == 0 && (S & 7) == 0) {       // (the item order of bank_search_kernel)
        const int blk = lin >> 5, r = lin & 31;
        const int ncg = S >> 3;
        const int qg = blk / ncg, cg = blk - qg * ncg;
        qt = qg * 4 + (r & 3);
        chunk = cg * 8 + (r >> 2);
    } else {
        qt = lin / S; chunk = lin - qt * S;
    }
    const int j0 = qt * GEMM_BN;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    if (threadIdx.x < 256) lds_cnt[threadIdx.x] = 0;
    float tau[4], sum[4], sq[4], mx[4], cn[4];      // (only tau is live in the filter form)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int q = j0 + wn * 64 + n * 16 + (lane & 15);
        tau[n] = (q < e.M) ? e.tau[q] : INFINITY;
        sum[n] = 0.f; sq[n] = 0.f; mx[n] = -INFINITY; cn[n] = 0.f;
    }
    __syncthreads();

    const int bt0 = chunk * tiles_per_chunk;
    int bt1 = bt0 + tiles_per_chunk;
    if (bt1 > n_bank_tiles) bt1 = n_bank_tiles;
    const int n_full = (int)(e.R / GEMM_BM);
    const int bt_ring_end = bt1 < n_full ? bt1 : n_full;
    ring4_stream(
        g, smem, bt_ring_end - bt0,
        [&](int n, int& i0, int& jj0) __attribute__((always_inline)) { i0 = (bt0 + n) * GEMM_BM; jj0 = j0; },
        [&](int n, const gemm_acc_t& acc) __attribute__((always_inline)) {
            bank_tile_epilogue<true, true>(acc, e, (int64_t)(bt0 + n) * GEMM_BM, chunk, j0, wm, wn, lane, tau, sum, sq, mx,
                                           cn, lds_cnt);
        });
    for (int bt = (bt_ring_end > bt0 ? bt_ring_end : bt0); bt < bt1; ++bt) {      // the ragged last bank tile
        gemm_acc_t acc;
        gemm_zero_acc(acc);
        const int64_t tile_row0 = (int64_t)bt * GEMM_BM;
        gemm_mainloop(acc, g, (int)tile_row0, j0, smem);
        bank_tile_epilogue<false, true>(acc, e, tile_row0, chunk, j0, wm, wn, lane, tau, sum, sq, mx, cn, lds_cnt);
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int q = j0 + threadIdx.x;
        if (q < e.M) {
            int c = lds_cnt[threadIdx.x];
            if (c > BANK_CAP) { c = BANK_CAP; atomicOr(e.overflow, 1); }
            e.cand_cnt[(int64_t)chunk * e.M + q] = c;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The filter pass for SMALL query batches (M <= 64: the reference searches ONE query at a time, src/retrieval.py:636-680;
// BASELINE configs[0] has 48 rows) -- the HBM-bound regime of SURVEY.md 8(d).  The 256-query tile of the kernels above
// multiplies 256 query columns whatever M is (R x 256 x D x 2 FLOP: 0.28 ms at R = 1 M even at the matrix peak) and
// re-reads the bank through the L2 5 x; here the bank is streamed ONCE, straight from HBM into MFMA operand registers,
// and multiplied with 16 * NQT query columns held in LDS:
//   * a workgroup (8 waves, one per CU) owns a contiguous chunk of bank rows; each wave walks its own 16-row groups:
//     all D / 64 x 2 sixteen-byte pieces of a group are requested before the previous group is multiplied (two register
//     sets: ~50 KB of loads in flight per wave, 400 KB per CU -- latency is covered by bytes, not by occupancy);
//   * a lane (row r = lane & 15, k-group g = lane >> 4) loads the 32 contiguous bytes [g * 32, g * 32 + 32) of every
//     128-byte line of its row: the two MFMAs of a 64-deep block take k = 64 kb + 16 g + {0..7} and {8..15} -- any
//     k permutation is a valid inner product as long as the query fragments use the same one (they do: the LDS image of
//     the queries is read with the same offsets);
//   * queries: hi planes [16 NQT, D] bf16 in LDS, row pitch D * 2 + 16 bytes (16 lanes of a ds_read_b128 group hit 16
//     distinct 16-byte bank groups: conflict-free);
//   * epilogue: as bank_tile_epilogue<FILTER> -- every lane compares its 4 rows x NQT queries with its queries' tau and
//     appends survivors to the per-(chunk, query) lists (LDS slot counters, no global atomics).  The products are summed
//     in another k order than in the ring kernel (fp32: ~1e-7), which the margin of kth_bound_kernel covers; the select
//     pass re-scores the listed rows exactly, so the top-k set and values are the same.
// ---------------------------------------------------------------------------------------------------------------
template <int KB, int NQT>
__global__ __launch_bounds__(512) void bank_filter_skinny_kernel(const uint16_t* __restrict__ bank, int64_t ldb,
                                                                  const uint16_t* __restrict__ qplanes, BankEpilogue e,
                                                                  int rows_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = KB * 64;
    constexpr int QPITCH = D * 2 + 16;
    int* lds_cnt = (int*)(smem + NQT * 16 * QPITCH);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.x;
    const int r16 = lane & 15, g = lane >> 4;
    // ---- queries' hi planes -> LDS (rows >= M: zeros, tau = +inf below)
    for (int idx = tid; idx < NQT * 16 * (D / 8); idx += 512) {
        const int q = idx / (D / 8), c = idx - q * (D / 8);
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (q < e.M) v = *(const u32x4_t*)(qplanes + (int64_t)q * 2 * D + c * 8);
        *(u32x4_t*)(smem + q * QPITCH + c * 16) = v;
    }
    if (tid < NQT * 16) lds_cnt[tid] = 0;
    float tau[NQT];
#pragma unroll
    for (int n = 0; n < NQT; ++n) {
        const int q = n * 16 + r16;
        tau[n] = (q < e.M) ? e.tau[q] : INFINITY;
    }
    __syncthreads();

    const int64_t row_lo = (int64_t)chunk * rows_per_chunk;
    int64_t row_hi = row_lo + rows_per_chunk;
    if (row_hi > e.R) row_hi = e.R;
    const int n_groups = row_hi > row_lo ? (int)((row_hi - row_lo + 15) >> 4) : 0;
    const char* qbase = smem + r16 * QPITCH + g * 32;

    auto load_group = [&](int grp, u32x4_t (&a)[2 * KB]) {
        int64_t row = row_lo + (int64_t)grp * 16 + r16;
        if (row >= e.R) row = e.R - 1;                        // clamped: masked in the epilogue
        const uint16_t* src = bank + row * ldb + g * 16;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            a[2 * kb] = __builtin_nontemporal_load((const u32x4_t*)(src + kb * 64));
            a[2 * kb + 1] = __builtin_nontemporal_load((const u32x4_t*)(src + kb * 64 + 8));
        }
    };
    auto compute_group = [&](int grp, const u32x4_t (&a)[2 * KB]) {
        f32x4_t acc[NQT];
#pragma unroll
        for (int n = 0; n < NQT; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, a[2 * kb + hf]);
#pragma unroll
                for (int n = 0; n < NQT; ++n) {
                    const bf16x8_t bfr = *(const bf16x8_t*)(qbase + n * 16 * QPITCH + kb * 128 + hf * 16);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[n], 0, 0, 0);
                }
            }
        }
        const int64_t row0 = row_lo + (int64_t)grp * 16 + 4 * g;
#pragma unroll
        for (int n = 0; n < NQT; ++n) {
            const f32x4_t v = acc[n];
            const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            if (m4 > tau[n]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (v[r] > tau[n] && row0 + r < row_hi) {
                        int q = n * 16 + r16;
                        asm volatile("" : "+v"(q));
                        if (q < e.M) {
                            const int slot = atomicAdd(&lds_cnt[q], 1);
                            if (slot < BANK_CAP) {
                                Cand c;
                                c.v = v[r];
                                c.idx = (int32_t)(row0 + r + e.idx_offset);
                                e.cand[((int64_t)chunk * e.M + q) * BANK_CAP + slot] = c;
                            }
                        }
                    }
                }
            }
        }
    };
    // two register sets: group i + 8 is in flight while group i is multiplied
    u32x4_t a0[2 * KB], a1[2 * KB];
    int grp = wave;
    if (grp < n_groups) load_group(grp, a0);
    while (grp < n_groups) {
        if (grp + 8 < n_groups) load_group(grp + 8, a1);
        compute_group(grp, a0);
        grp += 8;
        if (grp >= n_groups) break;
        if (grp + 8 < n_groups) load_group(grp + 8, a0);
        compute_group(grp, a1);
        grp += 8;
    }
    __syncthreads();
    if (tid < NQT * 16 && tid < e.M) {
        int c = lds_cnt[tid];
        if (c > BANK_CAP) { c = BANK_CAP; atomicOr(e.overflow, 1); }
        e.cand_cnt[(int64_t)chunk * e.M + tid] = c;
    }
}

template <int KB>
static hipError_t launch_skinny_kb(const uint16_t* bank, int64_t ldb, const uint16_t* qplanes, const BankEpilogue& e, int S,
                                   int rows_per_chunk, hipStream_t stream) {
    const int nqt = (e.M + 15) / 16;
    const size_t lds = (size_t)nqt * 16 * (KB * 128 + 16) + 64 * 4 + 64;
#define SKINNY_CASE(N)                                                                                                        \
    case N: {                                                                                                                 \
        static std::once_flag once;                                                                                           \
        static hipError_t ast = hipSuccess;                                                                                   \
        std::call_once(once, [] {                                                                                             \
            ast = hipFuncSetAttribute((const void*)bank_filter_skinny_kernel<KB, N>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      4 * 16 * (KB * 128 + 16) + 64 * 4 + 64);                                                \
        });                                                                                                                   \
        if (ast != hipSuccess) return ast;                                                                                    \
        hipLaunchKernelGGL((bank_filter_skinny_kernel<KB, N>), dim3(S), dim3(512), lds, stream, bank, ldb, qplanes, e,        \
                           rows_per_chunk);                                                                                   \
        break;                                                                                                                \
    }
    switch (nqt) {
        SKINNY_CASE(1)
        SKINNY_CASE(2)
        SKINNY_CASE(3)
        SKINNY_CASE(4)
        default: return hipErrorInvalidValue;
    }
#undef SKINNY_CASE
    return hipGetLastError();
}

// D in {128, 512, 768} (two register sets of D / 64 x 2 sixteen-byte pieces: D = 1024 would spill) and M <= 64; false = shape
// not covered (the caller takes the 256-query-tile kernels)
static bool skinny_covers(int D, int M) { return M >= 1 && M <= 64 && (D == 128 || D == 512 || D == 768); }
static hipError_t launch_bank_filter_skinny(const uint16_t* bank, int64_t ldb, int D, const uint16_t* qplanes, const BankEpilogue& e,
                                            int S, int rows_per_chunk, hipStream_t stream) {
    switch (D) {
        case 128: return launch_skinny_kb<2>(bank, ldb, qplanes, e, S, rows_per_chunk, stream);
        case 512: return launch_skinny_kb<8>(bank, ldb, qplanes, e, S, rows_per_chunk, stream);
        case 768: return launch_skinny_kb<12>(bank, ldb, qplanes, e, S, rows_per_chunk, stream);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Pass 0 for small query batches over a bf16 bank (M <= 64): the sample's 256 group maxima straight from the matrix
// pipe -- no [M, n_sample] similarities, no scan of them.  (The dense sample GEMM ran one 256 x 256 tile per workgroup
// for M query columns, 49 us, and kth_bound_kernel another 46 us: a third of a 1 M-row search's time that is not the
// bank stream.)  Workgroup (G, qb): sample rows [G * per, (G + 1) * per) -- bank rows i * sample_stride -- against
// the 16 * NQT queries of block qb, whose [hi | lo] planes sit in LDS (pitch 4 D + 16 bytes, conflict-free as in the
// filter kernel); both products of a bf16 bank, b.qhi + b.qlo, go into one accumulator (exact bf16 x bf16 products,
// fp32 sums: the value the dense sample GEMM computed, in another k order -- inside kth_finish's slack).
// Lane (r16, g) owns query n * 16 + r16 and the group's rows 4 g .. 4 g + 3: a running maximum per lane, reduced over
// g by two shuffles and over the eight waves through LDS.  Rows past the group's end are clamped for the load and
// masked to -inf (a duplicated row in two groups would break "k distinct sample rows are >= tau").
// ---------------------------------------------------------------------------------------------------------------
template <int KB, int NQT>
__global__ __launch_bounds__(512) void bank_sample_skinny_kernel(const uint16_t* __restrict__ bank, int64_t ldb,
                                                                  const uint16_t* __restrict__ qplanes, int M, int n_sample,
                                                                  int sample_stride, int per, float* __restrict__ gmax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = KB * 64;
    constexpr int QPITCH = 4 * D + 16;
    float* red = (float*)(smem + NQT * 16 * QPITCH);          // [8][NQT * 16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = blockIdx.x, q0 = blockIdx.y * NQT * 16;
    const int r16 = lane & 15, g = lane >> 4;
    const int i_lo = G * per;
    int i_hi = i_lo + per;
    if (i_hi > n_sample) i_hi = n_sample;
    const int n_groups = i_hi > i_lo ? (i_hi - i_lo + 15) >> 4 : 0;
    const char* qbase = smem + r16 * QPITCH + g * 32;
    float mx[NQT];
#pragma unroll
    for (int n = 0; n < NQT; ++n) mx[n] = -INFINITY;

    auto load_group = [&](int grp, u32x4_t (&a)[2 * KB]) {
        int i = i_lo + grp * 16 + r16;
        if (i >= i_hi) i = i_hi - 1;
        const uint16_t* src = bank + (int64_t)i * sample_stride * ldb + g * 16;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            a[2 * kb] = *(const u32x4_t*)(src + kb * 64);
            a[2 * kb + 1] = *(const u32x4_t*)(src + kb * 64 + 8);
        }
    };
    auto compute_group = [&](int grp, const u32x4_t (&a)[2 * KB]) {
        f32x4_t acc[NQT];
#pragma unroll
        for (int n = 0; n < NQT; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, a[2 * kb + hf]);
#pragma unroll
                for (int n = 0; n < NQT; ++n) {
                    const char* qp = qbase + n * 16 * QPITCH + kb * 128 + hf * 16;
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, *(const bf16x8_t*)qp, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, *(const bf16x8_t*)(qp + 2 * D), acc[n], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // keeps hipcc from hoisting every fragment read of the group (154 spills)
        }
        const int i0 = i_lo + grp * 16 + 4 * g;
#pragma unroll
        for (int n = 0; n < NQT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i0 + r < i_hi) mx[n] = fmaxf(mx[n], acc[n][r]);
    };
    // two register sets: a wave has two or three groups (n_sample / 256 rows per workgroup), both requested up front -- and
    // before the queries' planes are copied to LDS (NQT * 16 * D / 4 sixteen-byte pieces, 3 per thread and query at D = 768,
    // unrolled so that a thread's pieces are in flight together): the launch is latency-bound, the two latencies overlap
    u32x4_t a0[2 * KB], a1[2 * KB];
    int grp = wave;
    if (grp < n_groups) load_group(grp, a0);
    {
        constexpr int PIECES = NQT * 16 * (D / 4);
        constexpr int ROUNDS = (PIECES + 511) / 512;
#pragma unroll 6
        for (int u = 0; u < ROUNDS; ++u) {
            const int idx = tid + u * 512;
            const int q = idx / (D / 4), c = idx - q * (D / 4);
            u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
            if (idx < PIECES && q0 + q < M) v = *(const u32x4_t*)(qplanes + (int64_t)(q0 + q) * 2 * D + c * 8);
            if (idx < PIECES) *(u32x4_t*)(smem + q * QPITCH + c * 16) = v;
        }
    }
    __syncthreads();
    while (grp < n_groups) {
        if (grp + 8 < n_groups) load_group(grp + 8, a1);
        compute_group(grp, a0);
        grp += 8;
        if (grp >= n_groups) break;
        if (grp + 8 < n_groups) load_group(grp + 8, a0);
        compute_group(grp, a1);
        grp += 8;
    }
#pragma unroll
    for (int n = 0; n < NQT; ++n) {
        float v = mx[n];
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        if (g == 0) red[wave * (NQT * 16) + n * 16 + r16] = v;
    }
    __syncthreads();
    if (tid < NQT * 16 && q0 + tid < M) {
        float v = red[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) v = fmaxf(v, red[w * (NQT * 16) + tid]);
        gmax[(int64_t)(q0 + tid) * 256 + G] = v;
    }
}

// 16 * NQT queries per workgroup, NQT <= 3: both planes of 48 queries are 146 KB of LDS at D = 768 (64 queries: two blocks of 32)
template <int KB>
static hipError_t launch_sample_skinny_kb(const uint16_t* bank, int64_t ldb, const uint16_t* qplanes, int M, int n_sample,
                                          int sample_stride, float* gmax, hipStream_t stream) {
    const int nqt = M <= 16 ? 1 : (M > 32 && M <= 48) ? 3 : 2;
    const int per = (n_sample + 255) / 256;
    const dim3 grid(256, (M + nqt * 16 - 1) / (nqt * 16));
    const size_t lds = (size_t)nqt * 16 * (KB * 256 + 16) + 8 * nqt * 16 * 4;
    static std::once_flag once;
    static hipError_t ast = hipSuccess;
    std::call_once(once, [] {
        ast = hipFuncSetAttribute((const void*)bank_sample_skinny_kernel<KB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  16 * (KB * 256 + 16) + 8 * 16 * 4);
        if (ast == hipSuccess)
            ast = hipFuncSetAttribute((const void*)bank_sample_skinny_kernel<KB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      32 * (KB * 256 + 16) + 8 * 32 * 4);
        if (ast == hipSuccess)
            ast = hipFuncSetAttribute((const void*)bank_sample_skinny_kernel<KB, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      48 * (KB * 256 + 16) + 8 * 48 * 4);
    });
    if (ast != hipSuccess) return ast;
#define SAMPLE_CASE(N)                                                                                                          \
    case N:                                                                                                                     \
        hipLaunchKernelGGL((bank_sample_skinny_kernel<KB, N>), grid, dim3(512), lds, stream, bank, ldb, qplanes, M, n_sample,   \
                           sample_stride, per, gmax);                                                                           \
        break;
    switch (nqt) {
        SAMPLE_CASE(1)
        SAMPLE_CASE(2)
        SAMPLE_CASE(3)
    }
#undef SAMPLE_CASE
    return hipGetLastError();
}
static hipError_t launch_bank_sample_skinny(const uint16_t* bank, int64_t ldb, int D, const uint16_t* qplanes, int M, int n_sample,
                                            int sample_stride, float* gmax, hipStream_t stream) {
    switch (D) {
        case 128: return launch_sample_skinny_kb<2>(bank, ldb, qplanes, M, n_sample, sample_stride, gmax, stream);
        case 512: return launch_sample_skinny_kb<8>(bank, ldb, qplanes, M, n_sample, sample_stride, gmax, stream);
        case 768: return launch_sample_skinny_kb<12>(bank, ldb, qplanes, M, n_sample, sample_stride, gmax, stream);
        default: return hipErrorInvalidValue;
    }
}

// (v desc, idx asc) ordering
__device__ __forceinline__ bool cand_better(float v, int idx, float ov, int oidx) {
    return (v > ov) || (v == ov && idx < oidx);
}

// One workgroup of SEL_T threads per query.  Round 4: 1 024 threads (16 waves x 4 rows in flight in the re-scoring: the
// ~250-350 listed rows of a query took 16-22 round trips of ~2 us with 4 waves, a fifth of a small search), and the list
// gather reads a thread's chunk counts in one batch of independent loads instead of one dependent load per chunk.
#define SEL_T 1024
#define SEL_W (SEL_T / 64)
#define SEL_CPT 8          // chunks per thread kept in registers (S <= SEL_T * SEL_CPT; more: the counts are read twice)
__global__ __launch_bounds__(SEL_T) void bank_select_kernel(const Cand* __restrict__ cand,
                                                            const int32_t* __restrict__ cand_cnt,
                                                            const float* __restrict__ mom_part, int S, int M,
                                                            int k, int32_t* __restrict__ topk_idx,
                                                            float* __restrict__ topk_sim,
                                                            float* __restrict__ moments,
                                                            int32_t* __restrict__ overflow,
                                                            const uint16_t* __restrict__ rs_bank, int64_t rs_ld,
                                                            int rs_planes, int D,
                                                            const float* __restrict__ rs_rows,
                                                            int64_t idx_offset) {
    __shared__ Cand pool[BANK_POOL];
    __shared__ __attribute__((aligned(16))) float qs[BANK_MAX_RESCORE_D];
    __shared__ int wsum[SEL_W];
    __shared__ float red_v[2][SEL_W];
    __shared__ int red_i[2][SEL_W];
    __shared__ int red_p[2][SEL_W];
    __shared__ float mom_red[SEL_W][4];
    const int q = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;

    // chunks handled by this thread: c = t, t + SEL_T, ...
    const bool few = S <= SEL_T * SEL_CPT;
    int cnt[SEL_CPT];
    int mine = 0;
    if (few) {
#pragma unroll
        for (int u = 0; u < SEL_CPT; ++u) {
            const int c = t + u * SEL_T;
            cnt[u] = c < S ? cand_cnt[(int64_t)c * M + q] : 0;
        }
#pragma unroll
        for (int u = 0; u < SEL_CPT; ++u) mine += cnt[u];
    } else {
        for (int c = t; c < S; c += SEL_T) mine += cand_cnt[(int64_t)c * M + q];
    }
    // exclusive prefix of `mine` over the workgroup: wave scan + the waves' totals
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0, total_all = 0;
#pragma unroll
    for (int w = 0; w < SEL_W; ++w) {
        const int v = wsum[w];
        if (w < wave) wbase += v;
        total_all += v;
    }
    int off = wbase + incl - mine;
    if (few) {
#pragma unroll
        for (int u = 0; u < SEL_CPT; ++u) {
            const int n = cnt[u];
            if (n > 0) {
                const Cand* src = cand + ((int64_t)(t + u * SEL_T) * M + q) * BANK_CAP;
                for (int i = 0; i < n; ++i) {
                    if (off < BANK_POOL) pool[off] = src[i];
                    ++off;
                }
            }
        }
    } else {
        for (int c = t; c < S; c += SEL_T) {
            const int n = cand_cnt[(int64_t)c * M + q];
            const Cand* src = cand + ((int64_t)c * M + q) * BANK_CAP;
            for (int i = 0; i < n; ++i) {
                if (off < BANK_POOL) pool[off] = src[i];
                ++off;
            }
        }
    }
    if (t == 0 && total_all > BANK_POOL) atomicOr(overflow, 2);
    const int total = total_all < BANK_POOL ? total_all : BANK_POOL;
    if (rs_bank) {
        for (int c = t; c < D; c += SEL_T) qs[c] = rs_rows[(int64_t)q * D + c];
    }
    __syncthreads();

    if (rs_bank) {
        // exact fp32 re-scoring of the listed rows: one wave per row, four rows in flight per wave
        for (int i0 = wave; i0 < total; i0 += 4 * SEL_W) {
            float part[4];
            const uint16_t* br[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int i = i0 + u * SEL_W;
                i = i < total ? i : total - 1;
                br[u] = rs_bank + ((int64_t)pool[i].idx - idx_offset) * rs_ld;
                part[u] = 0.f;
            }
            for (int c = lane * 8; c < D; c += 512) {
                u32x4_t h[4], l[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    h[u] = *(const u32x4_t*)(br[u] + c);
                    if (rs_planes > 1) l[u] = *(const u32x4_t*)(br[u] + D + c);
                }
                const f32x4_t q0 = *(const f32x4_t*)(qs + c), q1 = *(const f32x4_t*)(qs + c + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        float a = __uint_as_float(h[u][e2] << 16), b = __uint_as_float(h[u][e2] & 0xffff0000u);
                        if (rs_planes > 1) {
                            a += __uint_as_float(l[u][e2] << 16);
                            b += __uint_as_float(l[u][e2] & 0xffff0000u);
                        }
                        const float qa = e2 < 2 ? q0[e2 * 2] : q1[e2 * 2 - 4];
                        const float qb = e2 < 2 ? q0[e2 * 2 + 1] : q1[e2 * 2 - 3];
                        part[u] = fmaf(a, qa, fmaf(b, qb, part[u]));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = wave_sum(part[u]);
                const int i = i0 + u * SEL_W;
                if (lane == 0 && i < total) pool[i].v = v;
            }
        }
        __syncthreads();
    }

    // k rounds of arg-max, ONE barrier each: the waves' bests go to a slot pair indexed by the round's parity, every wave
    // reduces the SEL_W entries again by shuffles (same result in every wave), and the winner is struck out by the thread
    // that owns its pool slot (slot % SEL_T: the only thread that ever reads it again)
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff, bp = -1;
        for (int i = t; i < total; i += SEL_T) {
            const Cand c = pool[i];
            if (c.idx >= 0 && cand_better(c.v, c.idx, bv, bi)) { bv = c.v; bi = c.idx; bp = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            const int op = __shfl_xor(bp, o, 64);
            if (op >= 0 && (bp < 0 || cand_better(ov, oi, bv, bi))) { bv = ov; bi = oi; bp = op; }
        }
        const int par = r & 1;
        if (lane == 0) { red_v[par][wave] = bv; red_i[par][wave] = bi; red_p[par][wave] = bp; }
        __syncthreads();
        float fv = red_v[par][lane & (SEL_W - 1)];
        int fi = red_i[par][lane & (SEL_W - 1)], fp = red_p[par][lane & (SEL_W - 1)];
#pragma unroll
        for (int o = SEL_W / 2; o > 0; o >>= 1) {
            const float ov = __shfl_xor(fv, o, 64);
            const int oi = __shfl_xor(fi, o, 64);
            const int op = __shfl_xor(fp, o, 64);
            if (op >= 0 && (fp < 0 || cand_better(ov, oi, fv, fi))) { fv = ov; fi = oi; fp = op; }
        }
        if (t == 0) {
            topk_idx[(int64_t)q * k + r] = fp >= 0 ? fi : -1;
            topk_sim[(int64_t)q * k + r] = fp >= 0 ? fv : -INFINITY;
        }
        if (fp >= 0 && (fp & (SEL_T - 1)) == t) pool[fp].idx = -1;     // taken
    }

    if (moments) {
        float s = 0.f, s2 = 0.f, m = -INFINITY, c = 0.f;
        for (int ch = t; ch < S; ch += SEL_T) {
            const float* p = mom_part + ((int64_t)ch * M + q) * 4;
            s += p[0]; s2 += p[1]; m = fmaxf(m, p[2]); c += p[3];
        }
        s = wave_sum(s); s2 = wave_sum(s2); c = wave_sum(c); m = wave_max(m);
        if (lane == 0) { mom_red[wave][0] = s; mom_red[wave][1] = s2; mom_red[wave][2] = m; mom_red[wave][3] = c; }
        __syncthreads();
        if (t == 0) {
            float* o = moments + (int64_t)q * 4;
            float s0 = 0.f, s1 = 0.f, m2 = -INFINITY, s3 = 0.f;
            for (int w = 0; w < SEL_W; ++w) { s0 += mom_red[w][0]; s1 += mom_red[w][1]; m2 = fmaxf(m2, mom_red[w][2]); s3 += mom_red[w][3]; }
            o[0] = s0; o[1] = s1; o[2] = m2; o[3] = s3;
        }
    }
}

// ---------------------------------------------------------------------------
// Dense fallback for degenerate banks (candidate lists overflowed): the similarities of a
// block of query rows are materialised ([m, R] fp32, m <= 64) by the dense GEMM and every row
// is reduced by k rounds of block arg-max.  Slow (R reads per round) but unconditional.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void row_topk_kernel(float* __restrict__ sims, int64_t R, int k,
                                                        float count_thr, int64_t idx_offset,
                                                        int32_t* __restrict__ topk_idx,
                                                        float* __restrict__ topk_sim,
                                                        float* __restrict__ moments) {
    __shared__ float rv[16];
    __shared__ int ri[16];
    __shared__ float rs[16], rq[16], rc[16];
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float* row = sims + (int64_t)q * R;
    if (moments) {
        float s = 0.f, s2 = 0.f, c = 0.f;
        for (int64_t i = t; i < R; i += 1024) { const float v = row[i]; s += v; s2 = fmaf(v, v, s2); c += (v >= count_thr) ? 1.f : 0.f; }
        s = wave_sum(s); s2 = wave_sum(s2); c = wave_sum(c);
        if (lane == 0) { rs[wave] = s; rq[wave] = s2; rc[wave] = c; }
        __syncthreads();
        if (t == 0) {
            float a = 0.f, b = 0.f, d = 0.f;
            for (int w = 0; w < 16; ++w) { a += rs[w]; b += rq[w]; d += rc[w]; }
            moments[(int64_t)q * 4 + 0] = a; moments[(int64_t)q * 4 + 1] = b; moments[(int64_t)q * 4 + 3] = d;
        }
        __syncthreads();
    }
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        bool has = false;
        for (int64_t i = t; i < R; i += 1024) {
            const float v = row[i];
            // NaN similarities (NaN bank rows / zero-norm queries) are never returned
            if (v == v && v != -INFINITY && (!has || v > bv || (v == bv && (int)i < bi))) { bv = v; bi = (int)i; has = true; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
        __syncthreads();
        if (t == 0) {
            float fv = rv[0]; int fi = ri[0];
            for (int w = 1; w < 16; ++w)
                if (ri[w] != 0x7fffffff && (fi == 0x7fffffff || rv[w] > fv || (rv[w] == fv && ri[w] < fi))) { fv = rv[w]; fi = ri[w]; }
            if (fi != 0x7fffffff) {
                topk_idx[(int64_t)q * k + r] = (int32_t)(fi + idx_offset);
                topk_sim[(int64_t)q * k + r] = fv;
                if (r == 0 && moments) moments[(int64_t)q * 4 + 2] = fv;
                row[fi] = -INFINITY;      // taken
            } else {
                topk_idx[(int64_t)q * k + r] = -1;
                topk_sim[(int64_t)q * k + r] = -INFINITY;
                if (r == 0 && moments) moments[(int64_t)q * 4 + 2] = -INFINITY;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_bank_search_dense(const BankSearchLaunch& L, float* sims_ws, int block_rows, hipStream_t stream) {
    const int D = L.D;
    const int planes = (L.bank_planes == 2) ? 3 : 2;
    const int a_off[4] = {0, 0, D, 0};
    const int b_off[4] = {0, D, 0, 0};
    for (int m0 = 0; m0 < L.M; m0 += block_rows) {
        const int m = (L.M - m0 < block_rows) ? L.M - m0 : block_rows;
        GemmLaunch G;
        G.A = L.bank; G.lda = L.ldb; G.I = (int)L.R;
        G.B = L.qplanes + (int64_t)m0 * 2 * D; G.ldb = 2 * (int64_t)D; G.J = m; G.K = D; G.planes = planes;
        for (int p = 0; p < 4; ++p) { G.a_plane_off[p] = a_off[p]; G.b_plane_off[p] = b_off[p]; }
        G.out = sims_ws; G.ldo = L.R; G.epilogue = TVC_EPI_F32;
        hipError_t st = launch_gemm_bf16(G, stream);
        if (st != hipSuccess) return st;
        hipLaunchKernelGGL(row_topk_kernel, dim3(m), dim3(1024), 0, stream, sims_ws, L.R, L.k, L.count_thr,
                           L.idx_offset, L.topk_idx + (int64_t)m0 * L.k, L.topk_sim + (int64_t)m0 * L.k,
                           L.moments ? L.moments + (int64_t)m0 * 4 : nullptr);
        st = hipGetLastError();
        if (st != hipSuccess) return st;
    }
    return hipSuccess;
}

hipError_t launch_bank_search(const BankSearchLaunch& L, hipStream_t stream) {
    // thread-safe one-time setup (two engines may launch their first search from two threads)
    static std::once_flag attr_once;
    static hipError_t attr_st = hipSuccess;
    std::call_once(attr_once, [] {
        attr_st = hipFuncSetAttribute((const void*)bank_search_kernel<false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BANK_LDS_BYTES);
        if (attr_st == hipSuccess)
            attr_st = hipFuncSetAttribute((const void*)bank_search_kernel<true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, BANK_LDS_BYTES);
        if (attr_st == hipSuccess)
            attr_st = hipFuncSetAttribute((const void*)bank_filter_ring_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, BANK_LDS_BYTES);
    });
    if (attr_st != hipSuccess) return attr_st;
    const int D = L.D;
    // products: (bank plane, query plane) pairs accumulated into one tile
    //   bf16 bank : b.qhi + b.qlo
    //   fp32 bank : bhi.qhi + bhi.qlo + blo.qhi
    const int planes = (L.bank_planes == 2) ? 3 : 2;
    const int a_off[4] = {0, 0, D, 0};          // bank plane offsets
    const int b_off[4] = {0, D, 0, 0};          // query plane offsets
    // fast form: one-product filter + exact re-scoring (needs no per-row moments)
    const bool filter = (L.moments == nullptr) && L.bank_bounds && L.rows && D <= BANK_MAX_RESCORE_D && L.allow_filter;

    // small query batches: the bank streamed once from HBM against <= 64 query columns (TVC_BANK_SKINNY=0: off, for A/B runs)
    static const bool skinny_on = [] { const char* v = getenv("TVC_BANK_SKINNY"); return !v || atoi(v) != 0; }();
    const bool skinny = filter && skinny_on && skinny_covers(D, L.M) && L.ldb % 8 == 0;
    // ... and, over a bf16 bank, the sample's group maxima by the same route (TVC_BANK_SKINNY_SAMPLE=0: the dense sample GEMM)
    static const bool skinny_sample_on = [] { const char* v = getenv("TVC_BANK_SKINNY_SAMPLE"); return !v || atoi(v) != 0; }();

    // ---- pass 0: sample GEMM + tau ----------------------------------------
    hipError_t st;
    if (skinny && skinny_sample_on && L.bank_planes == 1) {
        // s0 holds >= M * n_sample floats, n_sample >= 256 or the whole bank: the first [M, 256] of it take the group maxima
        st = launch_bank_sample_skinny(L.bank, L.ldb, D, L.qplanes, L.M, L.n_sample, L.sample_stride, L.gmax, stream);
        if (st != hipSuccess) return st;
        hipLaunchKernelGGL(kth_groups_kernel, dim3(L.M), dim3(256), 0, stream, L.gmax, L.k, L.tau, L.qplanes, D, L.bank_bounds,
                           L.overflow);
        st = hipGetLastError();
        if (st != hipSuccess) return st;
    } else {
        GemmLaunch G;
        G.A = L.bank; G.lda = L.ldb * (int64_t)L.sample_stride; G.I = L.n_sample;
        G.B = L.qplanes; G.ldb = 2 * (int64_t)D; G.J = L.M; G.K = D; G.planes = planes;
        for (int p = 0; p < 4; ++p) { G.a_plane_off[p] = a_off[p]; G.b_plane_off[p] = b_off[p]; }
        G.out = L.s0; G.ldo = L.n_sample; G.epilogue = TVC_EPI_F32;
        st = launch_gemm_bf16(G, stream);
        if (st != hipSuccess) return st;
        hipLaunchKernelGGL(kth_bound_kernel, dim3(L.M), dim3(1024), 0, stream, L.s0, L.n_sample, L.k, L.tau,
                           filter ? L.qplanes : nullptr, D, L.bank_bounds);
        st = hipGetLastError();
        if (st != hipSuccess) return st;
        st = hipMemsetAsync(L.overflow, 0, sizeof(int32_t), stream);
        if (st != hipSuccess) return st;
    }

    // ---- pass 1: fused GEMM + filter ---------------------------------------
    GemmOperands g;
    g.A = L.bank; g.lda = L.ldb; g.I = (int)L.R;
    g.B = L.qplanes; g.ldb = 2 * (int64_t)D; g.J = L.M;
    g.ksteps_per_plane = D / GEMM_BK; g.planes = filter ? 1 : planes;
    for (int p = 0; p < 4; ++p) { g.a_plane_off[p] = a_off[p]; g.b_plane_off[p] = b_off[p]; }
    BankEpilogue e;
    e.tau = L.tau; e.cand = (Cand*)L.cand; e.cand_cnt = L.cand_cnt; e.mom_part = L.mom_part;
    e.overflow = L.overflow; e.R = L.R; e.idx_offset = L.idx_offset; e.M = L.M; e.count_thr = L.count_thr;
    const int nQt = (L.M + GEMM_BN - 1) / GEMM_BN;
    const int nbt = (int)((L.R + GEMM_BM - 1) / GEMM_BM);
    const int tpc = (nbt + L.S - 1) / L.S;
    // the filter pass streams its bank tiles through GEMM form 4 where that form's preconditions hold (TVC_BANK_RING=0:
    // the one-tile-at-a-time loop, for A/B runs); the ragged last bank tile is handled inside the kernel
    static const bool ring_on = [] { const char* v = getenv("TVC_BANK_RING"); return !v || atoi(v) != 0; }();
    const bool ring = filter && ring_on && L.q_rows_padded && (g.lda % 64 == 0) && (g.ldb % 64 == 0) && L.R >= GEMM_BM;
    if (skinny) {
        st = launch_bank_filter_skinny(L.bank, L.ldb, D, L.qplanes, e, L.S, tpc * GEMM_BM, stream);
        if (st != hipSuccess) return st;
    } else if (ring)
        hipLaunchKernelGGL(bank_filter_ring_kernel, dim3(nQt * L.S), dim3(GEMM_THREADS), BANK_LDS_BYTES, stream,
                           g, e, nQt, L.S, tpc, nbt);
    else if (filter)
        hipLaunchKernelGGL(bank_search_kernel<true>, dim3(nQt * L.S), dim3(GEMM_THREADS), BANK_LDS_BYTES, stream,
                           g, e, nQt, L.S, tpc, nbt);
    else
        hipLaunchKernelGGL(bank_search_kernel<false>, dim3(nQt * L.S), dim3(GEMM_THREADS), BANK_LDS_BYTES, stream,
                           g, e, nQt, L.S, tpc, nbt);
    st = hipGetLastError();
    if (st != hipSuccess) return st;

    // ---- pass 2: select -----------------------------------------------------
    hipLaunchKernelGGL(bank_select_kernel, dim3(L.M), dim3(SEL_T), 0, stream, (const Cand*)L.cand,
                       L.cand_cnt, L.mom_part, L.S, L.M, L.k, L.topk_idx, L.topk_sim, L.moments, L.overflow,
                       filter ? L.bank : nullptr, L.ldb, L.bank_planes, D, L.rows, L.idx_offset);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// merge of W per-shard partial top-k lists (bank sharded over GPUs)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_merge_kernel(const int32_t* __restrict__ idx_parts,
                                                         const float* __restrict__ sim_parts,
                                                         const float* __restrict__ feat_parts,
                                                         const float* __restrict__ mom_parts, int W,
                                                         int M, int k, int kf, int D,
                                                         int32_t* __restrict__ idx_out,
                                                         float* __restrict__ sim_out,
                                                         float* __restrict__ feat_out,
                                                         float* __restrict__ mom_out) {
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ int src_of_rank[32];
    const int q = blockIdx.x, t = threadIdx.x;
    const int n = W * k;   // <= 256
    float v = -INFINITY;
    int id = -1;
    if (t < n) {
        const int w = t / k, j = t - w * k;
        id = idx_parts[((int64_t)w * M + q) * k + j];
        v = (id >= 0) ? sim_parts[((int64_t)w * M + q) * k + j] : -INFINITY;
    }
    sv[t] = v; si[t] = id;
    if (t < 32) src_of_rank[t] = -1;
    __syncthreads();
    if (t < n) {
        int rank = 0;
        for (int u = 0; u < n; ++u) {
            if (u == t) continue;
            const float ov = sv[u]; const int oi = si[u];
            // entries with idx < 0 sort last; among valid: (v desc, idx asc); stable by position
            bool before;
            if (oi < 0) before = false;
            else if (id < 0) before = true;
            else before = (ov > v) || (ov == v && (oi < id || (oi == id && u < t)));
            rank += before ? 1 : 0;
        }
        if (id < 0) {
            // padded entries: count how many padded entries precede (keep ranks unique)
            int valid = 0, pad_before = 0;
            for (int u = 0; u < n; ++u) { valid += (si[u] >= 0); pad_before += (si[u] < 0 && u < t); }
            rank = valid + pad_before;
        }
        if (rank < k) {
            idx_out[(int64_t)q * k + rank] = id;
            sim_out[(int64_t)q * k + rank] = v;
            if (rank < kf) src_of_rank[rank] = t;
        }
    }
    __syncthreads();
    if (feat_out && feat_parts) {
        for (int r = 0; r < kf; ++r) {
            const int s = src_of_rank[r];
            float* o = feat_out + ((int64_t)q * kf + r) * D;
            if (s < 0 || si[s] < 0) {
                for (int c = t; c < D; c += 256) o[c] = 0.f;
            } else {
                const int w = s / k, j = s - w * k;
                // a global rank < kf implies a shard-local rank < kf
                const float* f = feat_parts + (((int64_t)w * M + q) * kf + (j < kf ? j : kf - 1)) * D;
                for (int c = t; c < D; c += 256) o[c] = f[c];
            }
        }
    }
    if (mom_out && mom_parts && t == 0) {
        float s = 0.f, s2 = 0.f, m = -INFINITY, c = 0.f;
        for (int w = 0; w < W; ++w) {
            const float* p = mom_parts + ((int64_t)w * M + q) * 4;
            s += p[0]; s2 += p[1]; m = fmaxf(m, p[2]); c += p[3];
        }
        float* o = mom_out + (int64_t)q * 4;
        o[0] = s; o[1] = s2; o[2] = m; o[3] = c;
    }
}

hipError_t launch_topk_merge(const int32_t* idx_parts, const float* sim_parts, const float* feat_parts,
                             const float* mom_parts, int W, int M, int k, int kf, int D,
                             int32_t* idx_out, float* sim_out, float* feat_out, float* mom_out,
                             hipStream_t stream) {
    if (M == 0) return hipSuccess;
    if (W < 1 || k < 1 || k > 128 || W * k > 256 || kf < 0 || kf > k || kf > 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(M), dim3(256), 0, stream, idx_parts, sim_parts, feat_parts,
                       mom_parts, W, M, k, kf, D, idx_out, sim_out, feat_out, mom_out);
    return hipGetLastError();
}
