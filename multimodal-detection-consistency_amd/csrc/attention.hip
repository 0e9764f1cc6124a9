// Multi-head self-attention for the CLIP towers (head_dim 64, T <= 288).
//
// One workgroup (4 waves) per (sequence, head).  The head's K and V live in
// LDS for the whole workgroup (T <= 288 rows x 64), so the softmax is exact
// and single-pass: no online rescale.  Each wave walks 16-query blocks:
//
//   S^T[key, q]  = K . Q^T      MFMA 16x16x32, A = K rows from LDS (ds_read_b128,
//                               XOR-swizzled 128-B rows), B = Q straight from HBM
//   softmax over keys           lane-local over the accumulator registers, then
//                               two xor-shuffles across the 4 lane groups
//   O^T[dh, q]   = V^T . P^T    A = V^T via ds_read_b64_tr_b16 (hardware
//                               transpose of the row-major V image), B = P^T packed
//                               from the S^T accumulators with NO lane movement
//                               (k-slot order chosen to match the accumulator map)
//
// "Swapped" products keep the query on the lane (column) dimension, so row
// statistics are per-lane scalars; a lane's output is 4 consecutive head dims per 16-wide tile, swapped between
// neighbouring tiles (v_permlane16_swap) into 16-byte store pieces.
//
// A persistent form (item i + 1's K / V rows requested before item i's query blocks and written to LDS after them) was
// built and measured in round 4 (git show 410b4a7:multimodal-detection-consistency_amd/csrc/attention.hip; EXPERIMENTS.md): bit-identical,
// 365-400 us against 277-292 us per ViT-L/14 layer call -- its 72 staging registers leave room for 0-4 pinned K tiles instead
// of 15 and the block loop becomes LDS-bound.
//
// A persistent LDS-DMA stream (round 4, second session; git show 47e248b:multimodal-detection-consistency_amd/csrc/attention.hip has the
// bit-identical form, EXPERIMENTS.md the four cuts): one 8-wave workgroup per CU walks 32 items, item i + 1's K / V images are
// requested by global_load_lds_dwordx4 into the other half of the LDS (no staging registers) under item i's query blocks.
// 269-282 us against 281-288 us at best, the step unchanged -- and with the fill fully hidden, the 17th block's query split
// over the waves and no compiler-inserted wait left, still 279-293 us: the kernel is bound by the vector and matrix work of
// its 17 query blocks (~2 700 VALU clocks each, 40 % of them the quarter-rate exp), not by the fill.  Not kept.
//
// A start stagger (round 4, second session: of the first two workgroups of a compute unit, the one in wave slot 1 sleeps
// 6 400 / 12 800 / 19 200 / 25 600 clocks before its fill, so that one workgroup fills while the other multiplies;
// scripts/attn_stagger_ab.sh, profiles/r04_attention_stagger_ab.log): 292-293 -> 283-286 us whatever the delay, -2.5 %:
// the two residents' lock-step is not what separates 285 us from the ~170 us floor.  Not kept.
//
// What bounds it (in-kernel clock stamps, round 3): the memory system's rate on the packed rows' 128-byte
// per-head pieces, not vector issue.  Hence the XCD-contiguous item order, the unconditional one-latency fill, the wait
// for the next block's Q fragments placed BEFORE the block's stores (vmcnt counts stores), and the 16-byte stores.

#include "common.hpp"
#include "kernels.hpp"
#include <mutex>

#define ATT_DH 64
#define ATT_KROW 128     // K image: 64 bf16 per row
#define ATT_VROW 160     // V image: 64 bf16 + 32 B pad (conflict-free tr reads)

// EXACT: every sequence has exactly MAXT key tiles (fixed-length, non-causal: the vision tower).
// The per-tile guards become compile-time true, so the 16-key tiles of a query block are
// straight-line code and their LDS reads / MFMAs / exps interleave instead of running as MAXT
// dependent chains separated by scalar branches.
// NW: waves per workgroup.  4 everywhere: a 6-wave form for the 257-token vision case (17 query
// blocks in 3 rounds instead of 5) needs 3 waves per SIMD = 168 VGPRs, spills 94 of them and runs
// 3.5x slower (1201 vs 343 us) - the 17 score tiles of a query block want the 256-register budget.
// Non-temporal hints (bit mask: 1 = K / V loads, 2 = Q loads, 4 = output stores): every 128-byte line of the packed QKV
// rows is read by exactly one workgroup, once.
#ifndef TVC_ATT_NT
#define TVC_ATT_NT 3      // measured 9.17 -> 9.08 ms of attention per step (7: 9.26, the out-projection re-reads the output)
#endif
#if TVC_ATT_NT & 1
#define ATT_LD_KV(p_) __builtin_nontemporal_load(p_)
#else
#define ATT_LD_KV(p_) (*(p_))
#endif
#if TVC_ATT_NT & 2
#define ATT_LD_Q(p_) __builtin_nontemporal_load(p_)
#else
#define ATT_LD_Q(p_) (*(p_))
#endif
#if TVC_ATT_NT & 4
#define ATT_ST_O(p_, v_) __builtin_nontemporal_store(v_, p_)
#else
#define ATT_ST_O(p_, v_) (*(p_) = (v_))
#endif
template <int MAXT, bool CAUSAL, int WPS, bool EXACT = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW / 2) void attention_kernel(const uint16_t* __restrict__ qkv,
                                                        uint16_t* __restrict__ out,
                                                        const int32_t* __restrict__ starts, int T_fixed,
                                                        int heads, int n_items, int k_bytes, int region_bytes,
                                                        const int32_t* __restrict__ pfx, int n_seq,
                                                        int pool_mode, const int32_t* __restrict__ pool_row) {
    // WPS waves cooperate on one (sequence, head) item; a workgroup holds 4 / WPS items,
    // each with its own K/V region in LDS.  Short text sequences use WPS = 1.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IPW = NW / WPS;
    const int width = heads * ATT_DH;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // everything below is wave-uniform: keep it in SGPRs so that the per-tile guards are scalar
    // branches, not exec masks
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item_local = wave / WPS, wsub = wave - item_local * WPS;
    // XCD-contiguous item order: consecutive workgroup ids alternate over the 8 XCDs, so with item = blockIdx the 16 heads
    // of a sequence -- 16 adjacent 128-byte pieces of every packed row -- were fetched by 8 different L2s at unrelated
    // times.  With a contiguous item range per XCD the heads of a sequence run on neighbouring CUs of ONE XCD at the same
    // time and every 2 KiB row is fetched whole within a short window: 337 -> 310 us per ViT-L/14 layer call.
    int item = xcd_contiguous(blockIdx.x, gridDim.x) * IPW + item_local;
    const bool active = item < n_items;
    if (!active) item = n_items - 1;
    const int seq = item / heads, h = item - seq * heads;
    int64_t row0;      // first of the sequence's own rows
    int T;             // keys: P shared-prefix rows (from prow0) + the own rows
    int P = 0;         // shared prefix length: queries are the own rows only (positions P .. T-1)
    int64_t prow0 = 0;
    if (starts) {
        const int s0 = __builtin_amdgcn_readfirstlane(starts[seq]);
        const int s1 = __builtin_amdgcn_readfirstlane(starts[seq + 1]);
        row0 = s0; T = s1 - s0;
        if (CAUSAL && pfx) {
            P = __builtin_amdgcn_readfirstlane(pfx[seq]);
            prow0 = __builtin_amdgcn_readfirstlane(pfx[n_seq + seq]);
            T += P;
        }
    } else { row0 = (int64_t)seq * T_fixed; T = T_fixed; }
    // packed row of key / position t
    auto key_row = [&](int t) -> int64_t { return t < P ? prow0 + t : row0 + (t - P); };
    if (T > MAXT * 16) T = MAXT * 16;      // host guarantees this; never index past the region
    const int NT = EXACT ? MAXT : (T + 15) >> 4;          // key tiles of 16
    const int NP = (NT + 1) >> 1;          // key pairs of 32
    const int KT = NT * 16, VT = EXACT ? NT * 16 : NP * 32;   // EXACT: an unpaired last tile multiplies its own V rows by zeros
    char* ldsK = smem + item_local * region_bytes;
    char* ldsV = ldsK + k_bytes;
    const int64_t ld = 3 * (int64_t)width;
    const int tsub = wsub * 64 + lane;     // thread index within the item's waves

    const int g = lane >> 4, r16 = lane & 15;
    const int sw0 = ((0 + g) ^ ((lane >> 1) & 7)) << 4;
    const int sw1 = ((4 + g) ^ ((lane >> 1) & 7)) << 4;
    // transposed-read lane address: lane i of a 16-lane group supplies row (i>>2),
    // columns 4*(i&3) .. +3 of a [4 keys][16 dh] block
    const int tr_off = (4 * g + (r16 >> 2)) * ATT_VROW + ((r16 & 3) << 3);
    const float scale_log2 = 0.125f * 1.4426950408889634f;   // dh^-0.5 * log2(e)
    const int own = T - P;                     // query rows (all of them unless a prefix is shared)
    // Pooled form (the LAST layer of a tower: only the pooled token's output is ever read -- the class token
    // of the vision tower, the EOT token of the text tower): ONE query per sequence, at absolute position
    // `pool_pos`, its output written to the compact row `seq`.  All 16 query columns of the one block carry
    // that query; column 0 stores.
    int pool_pos = 0;
    if (pool_mode == 2) pool_pos = starts ? T - 1 : __builtin_amdgcn_readfirstlane(pool_row[seq]) - (int)row0;
    const int NQ = pool_mode ? 1 : (own + 15) >> 4;
    // Masking costs nothing after the MFMA: the accumulator is INITIALISED with 0 or -inf
    // (-inf + q.k = -inf).  Only the last key tile holds keys >= T.
    f32x4_t pen_tail;
#pragma unroll
    for (int r = 0; r < 4; ++r) pen_tail[r] = ((NT - 1) * 16 + 4 * g + r >= T) ? -INFINITY : 0.f;

    // Q fragments come straight from HBM/L2: fetch the NEXT block's while this one computes
    auto q_ptr = [&](int qb) {
        if (pool_mode) return qkv + key_row(pool_pos) * ld + h * ATT_DH + 8 * g;
        int qrow = qb * 16 + r16;
        qrow = qrow < own ? qrow : own - 1;
        return qkv + (row0 + qrow) * ld + h * ATT_DH + 8 * g;
    };
    bf16x8_t nq0 = {}, nq1 = {};
    if (wsub < NQ) { const uint16_t* qp = q_ptr(wsub); nq0 = ATT_LD_Q((const bf16x8_t*)qp); nq1 = ATT_LD_Q((const bf16x8_t*)(qp + 32)); }

    // ---- fill K / V images (zero beyond T).  ALL global loads of the item -- both images and the first query block --
    // are issued before the first LDS write, so the workgroup pays ONE memory latency per item.  The loads are
    // UNCONDITIONAL (row index clamped, zeros selected at the LDS write): behind a per-lane `if (key < T)` every load was
    // its own basic block and hipcc put an `s_waitcnt vmcnt(0)` between the K loads and the V loads -- two latencies in
    // series, 11.7 k of the 40 k clocks of a workgroup's life (in-kernel stamps, round 3).
    constexpr int STEP = WPS * 64;
    constexpr int KIT = (MAXT * 16 * 8 + STEP - 1) / STEP;
    constexpr int VIT = (((MAXT + 1) / 2) * 32 * 8 + STEP - 1) / STEP;
    {
        u32x4_t kv[KIT], vv[VIT];
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
            const int idx = tsub + i * STEP;
            const int key = idx >> 3, c = idx & 7;
            kv[i] = ATT_LD_KV((const u32x4_t*)(qkv + key_row(key < T ? key : T - 1) * ld + width + h * ATT_DH + c * 8));
        }
#pragma unroll
        for (int i = 0; i < VIT; ++i) {
            const int idx = tsub + i * STEP;
            const int key = idx >> 3, c = idx & 7;
            vv[i] = ATT_LD_KV((const u32x4_t*)(qkv + key_row(key < T ? key : T - 1) * ld + 2 * width + h * ATT_DH + c * 8));
        }
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
            const int idx = tsub + i * STEP;
            const int key = idx >> 3, c = idx & 7;
            if (idx < KT * 8) *(u32x4_t*)(ldsK + key * ATT_KROW + ((c ^ ((key >> 1) & 7)) << 4)) = key < T ? kv[i] : u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < VIT; ++i) {
            const int idx = tsub + i * STEP;
            const int key = idx >> 3, c = idx & 7;
            if (idx < VT * 8) *(u32x4_t*)(ldsV + key * ATT_VROW + (c << 4)) = key < T ? vv[i] : u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();
    if (!active) return;

    // EXACT: the K fragments of a lane do not depend on the query block.  The first KPIN key tiles' fragments stay in
    // registers for the whole item (120 VGPRs at 15 tiles); the last two tiles of the 17 are read per block through a
    // pointer the compiler cannot prove loop-invariant -- with all 17 pinned (what hipcc's own hoisting did) the block
    // below needs 262 registers once its running max is the three-operand form, and spills.
    constexpr int KPIN = EXACT ? (MAXT > 15 ? 15 : MAXT) : 0;
    bf16x8_t ka[KPIN > 0 ? KPIN : 1], kb[KPIN > 0 ? KPIN : 1];
    if (EXACT) {
#pragma unroll
        for (int t = 0; t < KPIN; ++t) {
            ka[t] = *(const bf16x8_t*)(ldsK + (t * 16 + r16) * ATT_KROW + sw0);
            kb[t] = *(const bf16x8_t*)(ldsK + (t * 16 + r16) * ATT_KROW + sw1);
        }
    }
    for (int qb = wsub; qb < NQ; qb += WPS) {
        const int qr = qb * 16 + r16;
        const bf16x8_t bq0 = nq0, bq1 = nq1;
        if (qb + WPS < NQ) { const uint16_t* qp = q_ptr(qb + WPS); nq0 = ATT_LD_Q((const bf16x8_t*)qp); nq1 = ATT_LD_Q((const bf16x8_t*)(qp + 32)); }
        // absolute positions of the block's first / last query and of this lane's query
        const int qmin = pool_mode ? pool_pos : P + qb * 16;
        const int qmax = pool_mode ? pool_pos : qmin + 15;
        const int qpos = pool_mode ? pool_pos : P + qr;
        // causal: keys up to the block's last query position
        const int nt_c = (qmax >> 4) + 1;
        const int nt_q = EXACT ? MAXT : (CAUSAL ? (nt_c < NT ? nt_c : NT) : NT);

        int zoff = 0;
        asm volatile("" : "+v"(zoff));          // opaque 0: keeps the un-pinned K reads inside the loop
        const char* ldsK_i = ldsK + zoff;
        f32x4_t s[MAXT];
        float mx = -INFINITY;          // max of the RAW scores (scaling by a positive constant is monotonic)
        if (EXACT) {
            // two passes over the key tiles (dh 0-31, then dh 32-63): MAXT independent MFMAs per
            // pass instead of MAXT dependent pairs
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const f32x4_t c0 = (t == MAXT - 1) ? pen_tail : f32x4_t{0.f, 0.f, 0.f, 0.f};
                const bf16x8_t a0 = t < KPIN ? ka[t < KPIN ? t : 0] : *(const bf16x8_t*)(ldsK_i + (t * 16 + r16) * ATT_KROW + sw0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bq0, c0, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const bf16x8_t a1 = t < KPIN ? kb[t < KPIN ? t : 0] : *(const bf16x8_t*)(ldsK_i + (t * 16 + r16) * ATT_KROW + sw1);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq1, s[t], 0, 0, 0);
            }
            // IEEE-754-2019 maximum (NaN-propagating): hipcc emits v_maximum3_f32 with no canonicalising copy of the
            // MFMA outputs -- 54 max instructions per block where fmaxf costs 121.  A NaN score makes the query's
            // output NaN either way.
            float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                m0 = __builtin_elementwise_maximum(m0, __builtin_elementwise_maximum(s[t][0], s[t][1]));
                m1 = __builtin_elementwise_maximum(m1, __builtin_elementwise_maximum(s[t][2], s[t][3]));
            }
            mx = __builtin_elementwise_maximum(m0, m1);
        } else {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (t < nt_q) {
                f32x4_t c0 = (t == NT - 1) ? pen_tail : f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (CAUSAL && t * 16 + 15 > qmin) {      // tile reaches past the block's first query position
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * 16 + 4 * g + r > qpos) c0[r] = -INFINITY;
                }
                const char* kr = ldsK + (t * 16 + r16) * ATT_KROW;
                const bf16x8_t a0 = *(const bf16x8_t*)(kr + sw0);
                const bf16x8_t a1 = *(const bf16x8_t*)(kr + sw1);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bq0, c0, 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq1, c0, 0, 0, 0);
                mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
            }
        }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale_log2;

#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < nt_q) {
#pragma unroll
                for (int r = 0; r < 4; ++r)     // exp2(s * c - max * c): one FMA + raw v_exp_f32 (args <= 0; -inf -> 0)
                    s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], scale_log2, -mxs));
            }
        }
        // The softmax denominator comes out of the matrix pipe: a fifth "V^T tile" of ones gives
        // sum_k 1 * P[k, q] in every accumulator row (the pipe is ~20 % busy in this kernel, its 68 fp32 adds
        // and two cross-lane shuffles per query block sat on the vector issue slots that bound it).  The sum is
        // over the SAME bf16-rounded probabilities the numerator uses.
        f32x4_t osum = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

        f32x4_t o[4];
#pragma unroll
        for (int md = 0; md < 4; ++md) o[md] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < (MAXT + 1) / 2; ++u) {
            const int t0 = 2 * u, t1 = 2 * u + 1;
            if (t0 < nt_q) {
                // k-slots j<4: keys 16*t0 + 4g + j ; j>=4: keys 16*t1 + 4g + (j-4)
                const f32x4_t p0 = s[t0];
                const f32x4_t p1 = (t1 < MAXT) ? s[t1 < MAXT ? t1 : 0] : f32x4_t{0.f, 0.f, 0.f, 0.f};
                u32x4_t pk;
                pk[0] = pack_bf16x2(p0[0], p0[1]);
                pk[1] = pack_bf16x2(p0[2], p0[3]);
                pk[2] = pack_bf16x2(p1[0], p1[1]);
                pk[3] = pack_bf16x2(p1[2], p1[3]);
                const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pk);
#pragma unroll
                for (int md = 0; md < 4; ++md) {
                    const char* vb = ldsV + tr_off + md * 32;
                    const bf16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4_t*)(vb + t0 * 16 * ATT_VROW));
                    const bf16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4_t*)(vb + ((EXACT && t1 >= MAXT) ? t0 : t1) * 16 * ATT_VROW));
                    bf16x8_t a;
                    a[0] = v0[0]; a[1] = v0[1]; a[2] = v0[2]; a[3] = v0[3];
                    a[4] = v1[0]; a[5] = v1[1]; a[6] = v1[2]; a[7] = v1[3];
                    o[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb, o[md], 0, 0, 0);
                }
                osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb, osum, 0, 0, 0);
            }
        }
        // The NEXT block's query fragments are waited for HERE, before this block's stores are issued: vmcnt counts
        // stores too, and the stores sit in a per-lane conditional, so a wait for the (older) Q loads at the loop's
        // back-edge was an `s_waitcnt vmcnt(0)` that also waited out the four stores just issued -- 1 500 of the 5 000
        // clocks of a query block (in-kernel stamps, round 3).  Now nobody waits for the stores until the next block's
        // products are done.
        asm volatile("" : "+v"(nq0), "+v"(nq1));
        const float lsum = osum[0];          // every row of the ones tile holds the column (= query) sum
        // A lane's 4 features per 16-wide tile are 8 B; swapping 16-lane rows between the tiles md and md + 1
        // (v_permlane16_swap, as the GEMM's bf16 epilogue does) leaves every lane with 8 consecutive features: two 16-byte
        // stores per lane and 64 contiguous bytes per query row and instruction, instead of four 8-byte stores that
        // scatter 32-byte pieces.  The partner lanes carry the same query column, so the swap sits outside the mask.
        const float inv = 1.0f / lsum;
        u32x4_t ow[2];
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            const f32x4_t v0 = o[2 * mp] * inv, v1 = o[2 * mp + 1] * inv;
            const auto r0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
            ow[mp][0] = r0[0]; ow[mp][1] = r1[0]; ow[mp][2] = r0[1]; ow[mp][3] = r1[1];
        }
        if (pool_mode ? (r16 == 0) : (qr < own)) {
            uint16_t* op = out + (pool_mode ? (int64_t)seq : row0 + qr) * (int64_t)width + h * ATT_DH + (g & 1) * 16 + (g >> 1) * 8;
            ATT_ST_O((u32x4_t*)op, ow[0]);
            ATT_ST_O((u32x4_t*)(op + 32), ow[1]);
        }
    }
}

template <int MAXT, bool CAUSAL, int WPS, bool EXACT = false, int NW = 4>
static hipError_t launch_one(const uint16_t* qkv, uint16_t* out, const int32_t* starts, int n_seq, int T,
                             int max_T, int heads, hipStream_t stream, const int32_t* pfx = nullptr,
                             int pool_mode = 0, const int32_t* pool_row = nullptr) {
    const int NT = (max_T + 15) / 16, NP = (NT + 1) / 2;
    const int k_bytes = NT * 16 * ATT_KROW;
    // EXACT: V rows of whole tiles only (an unpaired last tile needs no zero partner rows).  257 tokens: 34 816 + 43 520 =
    // 78 336 B, two workgroups per CU.
    const int region = k_bytes + (EXACT ? NT * 16 : NP * 32) * ATT_VROW;
    constexpr int IPW = NW / WPS;
    const size_t lds = (size_t)region * IPW;
    static std::once_flag attr_once;          // per instantiation; thread-safe
    static hipError_t attr_st = hipSuccess;
    std::call_once(attr_once, [] {
        attr_st = hipFuncSetAttribute((const void*)attention_kernel<MAXT, CAUSAL, WPS, EXACT, NW>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    if (attr_st != hipSuccess) return attr_st;
    const int n_items = n_seq * heads;
    hipLaunchKernelGGL((attention_kernel<MAXT, CAUSAL, WPS, EXACT, NW>), dim3((n_items + IPW - 1) / IPW), dim3(NW * 64), lds, stream,
                       qkv, out, starts, T, heads, n_items, k_bytes, region, pfx, n_seq, pool_mode, pool_row);
    return hipGetLastError();
}

// starts == nullptr: n_seq sequences of seq_len rows each; otherwise sequence s owns rows
// [starts[s], starts[s+1]) (device array of n_seq + 1 ints) and seq_len is the MAXIMUM length.
// pool_mode 0: every row's output, token-major [rows, width].  1 / 2: only the pooled token of every sequence
// (1 = its first token, 2 = its EOT token: the last packed row, or packed row pool_row[s] for dense rows), written
// to the compact row s of `out` [n_seq, width] -- what the LAST layer of a tower needs.
hipError_t launch_attention(const uint16_t* qkv, uint16_t* out, const int32_t* starts, int n_seq, int seq_len,
                            int heads, int causal, hipStream_t stream, const int32_t* pfx, int pool_mode,
                            const int32_t* pool_row) {
    if (n_seq <= 0) return hipSuccess;
    if (seq_len < 1 || seq_len > 288 || heads < 1 || pool_mode < 0 || pool_mode > 2) return hipErrorInvalidValue;
    if (pool_mode == 2 && !starts && !pool_row) return hipErrorInvalidValue;
    const int NT = (seq_len + 15) / 16;
    if (causal) {
        if (pfx && !starts) return hipErrorInvalidValue;
        if (NT <= 2) return launch_one<2, true, 1>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, pfx, pool_mode, pool_row);
        if (NT <= 6) return launch_one<6, true, 2>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, pfx, pool_mode, pool_row);
        return launch_one<18, true, 4>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, pfx, pool_mode, pool_row);
    }
    if (!starts && NT == 17) return launch_one<17, false, 4, true>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, nullptr, pool_mode, pool_row);   // ViT-L/14: 257 tokens
    if (!starts && NT == 4) return launch_one<4, false, 2, true>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, nullptr, pool_mode, pool_row);     // ViT-B/32: 50 tokens
    if (NT <= 2) return launch_one<2, false, 1>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, nullptr, pool_mode, pool_row);
    if (NT <= 4) return launch_one<4, false, 2>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, nullptr, pool_mode, pool_row);
    if (NT <= 6) return launch_one<6, false, 4>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, nullptr, pool_mode, pool_row);
    return launch_one<18, false, 4>(qkv, out, starts, n_seq, seq_len, seq_len, heads, stream, nullptr, pool_mode, pool_row);
}
