// Host-side launch interface of the TVC HIP kernels (internal; the public
// boundary is include/tvc.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "host_plan.hpp"

enum { TVC_EPI_F32 = 0, TVC_EPI_BF16 = 1, TVC_EPI_GELU_BF16 = 2, TVC_EPI_RESID_F32 = 3 };

struct GemmLaunch {
    const uint16_t* A = nullptr;   // [I, lda] bf16  (weights / bank rows)
    const uint16_t* B = nullptr;   // [J, ldb] bf16  (tokens / query rows)
    int64_t lda = 0, ldb = 0;
    int I = 0, J = 0, K = 0;       // K per plane, multiple of 64
    int planes = 1;
    int a_plane_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // up to GEMM_MAX_PLANES
    int b_plane_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool splitk_small = false;     // launches of fewer tiles than CUs may split K over the idle CUs (needs splitk_ws; the fp32 sums are
                                   // then taken in a different order than in the one-pass kernels)
    int splitk_fixed = 0;          // 0: the launch decides (above).  1: never split.  S >= 2: split K exactly S ways for EVERY tile
                                   // (needs splitk_ws of tiles * S * 256 KiB) -- callers that promise results independent of the
                                   // launch size (the latent-diffusion model: an image must not depend on its batch mates) pick S
                                   // from the per-sample shape, never from J
    bool a_rows_padded = false;    // A has readable rows up to the next multiple of 256 beyond I (form 4 with a ragged last row tile)
    const float* bias = nullptr;   // [I]
    void* out = nullptr;           // [J, ldo]
    int64_t ldo = 0;
    int epilogue = TVC_EPI_F32;
    bool no_solo = false;          // internal: remainder launch of a split GEMM
    bool b_rows_padded = false;    // B has readable (garbage) rows up to the next multiple of 256 beyond J
    // optional scratch for the split-K tail (see launch_gemm_bf16); nullptr disables it
    float* splitk_ws = nullptr;
    size_t splitk_ws_bytes = 0;
};
hipError_t launch_gemm_bf16(const GemmLaunch& L, hipStream_t stream);

// ---- elementwise.hip
// y = LN(x [+ delta [+ delta2]]); with a delta and write_x the sum is written back to x (residual stream)
// delta_compact: the delta rows are [rows, d] in OUTPUT row order (a pooled last layer) instead of x's row layout
hipError_t launch_layernorm(float* x, int64_t x_row_stride, const int32_t* row_idx, const uint16_t* delta,
                            int write_x, const float* g, const float* b, uint16_t* y, int rows, int d,
                            hipStream_t stream, const uint16_t* delta2 = nullptr, int delta_compact = 0,
                            float* xsum_out = nullptr,      // xsum_out: fp32 [rows, d] receives x (+ deltas), compact
                            float* y32 = nullptr);          // y32: fp32 copy of the output rows (y may then be nullptr)
hipError_t launch_im2col(const float* pix, uint16_t* out, int B, int image, int patch, int Kp,
                         hipStream_t stream);
hipError_t launch_assemble_lnpre(const float* patch_out, const float* cls, const float* pos,
                                 const float* g, const float* b, float* x, int B, int T, int d,
                                 hipStream_t stream);
// starts == nullptr: dense [n_text, ctx] rows; else packed rows (text n owns rows
// [starts[n], starts[n+1]), i.e. its tokens up to and including EOT)
// pfx (optional, with starts): prefix sharing - text n owns only the rows of positions >= pfx[n]; see
// text_lens_scan_kernel
hipError_t launch_text_embed(const int32_t* tok, const float* tok_emb, const float* pos, float* x,
                             int32_t* eot_row, const int32_t* starts, int n_text, int ctx, int d, int vocab,
                             hipStream_t stream, const int32_t* pfx = nullptr);
// starts[0..n_text] = exclusive scan of the own row counts (argmax position + 1 [- shared prefix]);
// starts[n_text + 1] = max length; pfx (optional, int32 [2 * n_text], groups of G texts): [n] = shared
// prefix length, [n_text + n] = packed row of the base text's position 0
hipError_t launch_text_lens_scan(const int32_t* tok, int32_t* starts, int32_t* pfx, int n_text, int ctx, int G,
                                 hipStream_t stream, int32_t* lens_ws);
hipError_t launch_l2norm_rows(float* x, int rows, int d, hipStream_t stream);
// exact (erf) GELU in place: the activation of towers with TVC_ACT_GELU, after a store-only FC1
hipError_t launch_gelu_erf_bf16(uint16_t* x, int64_t n, hipStream_t stream);
hipError_t launch_gelu_erf_f32(float* x, int64_t n, hipStream_t stream);
hipError_t launch_split_planes(const float* x, uint16_t* out, int64_t rows, int d, int planes,
                               hipStream_t stream);
hipError_t launch_gather_rows(const uint16_t* bank, int64_t ld, int planes, int D, int64_t R,
                              const int32_t* idx, int64_t idx_offset, int n, float* out,
                              hipStream_t stream);

// ---- attention.hip
// pfx (optional, causal packed rows only): sequence s = pfx[s] rows starting at packed row
// pfx[n_seq + s] (shared prefix, keys only) followed by its own rows [starts[s], starts[s+1]) (keys + queries)
// pool_mode 1 / 2: only the pooled token's output per sequence (first token / EOT token), compact [n_seq, width]
hipError_t launch_attention(const uint16_t* qkv, uint16_t* out, const int32_t* starts, int n_seq,
                            int seq_len, int heads, int causal, hipStream_t stream, const int32_t* pfx = nullptr,
                            int pool_mode = 0, const int32_t* pool_row = nullptr);

// ---- bank.hip
struct BankSearchLaunch {
    const uint16_t* bank = nullptr;   // [R, ldb] bf16 planes
    int64_t ldb = 0;
    int64_t R = 0;
    int D = 0;
    int bank_planes = 1;              // 1 = bf16 bank, 2 = (hi | lo) split of an fp32 bank
    const uint16_t* qplanes = nullptr;// [M, 2*D] bf16 (hi | lo) query planes
    const float* rows = nullptr;      // [M, D] the fp32 query rows (exact re-scoring of the fast form)
    const float* bank_bounds = nullptr;// [2] device: max |hi plane row|, max |lo plane row|
    bool allow_filter = true;         // one-product filter + re-scoring when no moments are requested
    bool q_rows_padded = false;       // qplanes is readable up to the next multiple of 256 rows (the ring-form filter pass)
    int M = 0;
    int k = 0;
    float count_thr = 0.f;
    int64_t idx_offset = 0;
    // workspace (sized by bank_workspace_bytes)
    float* s0 = nullptr;              // [M, n_sample] pre-pass similarities
    float* gmax = nullptr;            // [M, 256] group maxima of the sample (small-M form: bank_sample_skinny_kernel)
    float* tau = nullptr;             // [M]
    void* cand = nullptr;             // [S, M, CAP] {float, int}
    int32_t* cand_cnt = nullptr;      // [S, M]
    float* mom_part = nullptr;        // [S, M, 4]
    int32_t* overflow = nullptr;      // [1]
    int n_sample = 0, sample_stride = 1, S = 0, cap = 0;
    // outputs
    int32_t* topk_idx = nullptr;
    float* topk_sim = nullptr;
    float* moments = nullptr;
};
hipError_t launch_bank_search(const BankSearchLaunch& L, hipStream_t stream);
hipError_t launch_bank_bounds(const uint16_t* bank, int64_t ld, int planes, int D, int64_t R, float* bounds,
                              hipStream_t stream);
// brute force: sims_ws fp32 [block_rows, R]
hipError_t launch_bank_search_dense(const BankSearchLaunch& L, float* sims_ws, int block_rows, hipStream_t stream);
hipError_t launch_topk_merge(const int32_t* idx_parts, const float* sim_parts, const float* feat_parts,
                             const float* mom_parts, int W, int M, int k, int kf, int D,
                             int32_t* idx_out, float* sim_out, float* feat_out, float* mom_out,
                             hipStream_t stream);

// ---- consistency.hip
struct ConsistencyParams {
    int reference_count;
    float similarity_threshold;
    int retrieval_top_k;
    float dup_threshold;
    float w_text_variants, w_consistency;
    float w_exp[4];
};
hipError_t launch_consistency(const float* img, const float* txt, int B, int N, int D,
                              const int32_t* ref_idx, const float* ref_sim, const float* ref_feat,
                              int ks, int kf, const ConsistencyParams& p, float* rec, int rec_stride,
                              hipStream_t stream);

// ---- backward.hip / attention_bwd.hip: input gradient (dX only) of the vision tower
hipError_t launch_layernorm_bwd(const float* x, int64_t x_row_stride, const uint16_t* delta, const void* dy, int dy_fp32,
                                const float* gamma, const float* dres, float* dx, uint16_t* dx16, int rows, int d,
                                int64_t out_row_stride, hipStream_t stream);
hipError_t launch_lnpre_bwd(const float* patch_out, const float* pos, const float* gamma, const float* dy,
                            uint16_t* dpatch, int B, int T, int d, hipStream_t stream);
hipError_t launch_gelu_bwd(uint16_t* dm, const uint16_t* u, int64_t n, hipStream_t stream);
hipError_t launch_gelu_fwd(const uint16_t* u, uint16_t* out, int64_t n, hipStream_t stream);
hipError_t launch_l2norm_bwd(const float* x, const float* dy, uint16_t* dx16, int rows, int d, int normalize, hipStream_t stream);
hipError_t launch_col2im(const float* dcols, float* dpix, int B, int S, int patch, int Kp, hipStream_t stream);
hipError_t launch_transpose_bf16(const uint16_t* in, uint16_t* out, int R, int C, hipStream_t stream);
hipError_t launch_pgd_step(float* adv, const float* clean, const float* grad, float* mom, int B, int64_t n, float eps,
                           float alpha, float mu, float lo, float hi, int targeted, hipStream_t stream);
hipError_t launch_l2_step(float* adv, const float* clean, const float* grad, int B, int64_t n, float eps, float step, float lo,
                          float hi, int descent, hipStream_t stream);
hipError_t launch_attention_bwd(const uint16_t* qkv, const uint16_t* dao, uint16_t* dqkv, float* stats_ws, int n_seq, int T,
                                int heads, hipStream_t stream);

// ---- precise.hip: fp32-grade towers (exact-f32 MFMA GEMM, fp32 attention)
// out[j, i] (op)= sum_k X[j, k] W[i, k] + bias[i]; epi 0 store, 1 QuickGELU, 2 out += ; K, ldw, ldx multiples of 4
hipError_t launch_gemm_f32(const float* W, int64_t ldw, const float* X, int64_t ldx, const float* bias, float* out,
                           int64_t ldo, int I, int J, int K, int epi, hipStream_t stream);
hipError_t launch_attention_f32(const float* qkv, float* out, int n_seq, int T, int heads, int causal, hipStream_t stream);
hipError_t launch_im2col_f32(const float* pix, float* out, int B, int image, int patch, hipStream_t stream);
hipError_t launch_gather_f32_rows(const float* x, int64_t ld, const int32_t* idx, int64_t idx_mul, float* out, int n,
                                  int d, hipStream_t stream);

// ---- split.hip: split-bf16 towers (hi | lo bf16 planes [rows, 2K], three MFMA products per element)
// LayerNorm of x (+ fp32 deltas d1, d2; written back to x when write_x) -> planes [rows, 2d] and / or fp32 rows y32
hipError_t launch_ln_split(float* x, int64_t x_row_stride, const int32_t* row_idx, const float* d1, const float* d2, int write_x,
                           const float* g, const float* b, uint16_t* planes, float* y32, int rows, int d, hipStream_t stream);
// fp32 [rows, ld_in] (K columns) -> planes [rows, 2 * Kp] zero padded; gelu 1: QuickGELU first, 2: erf GELU first
hipError_t launch_rows_split(const float* x, int64_t ld_in, uint16_t* out, int64_t rows, int K, int Kp, int gelu, hipStream_t stream);
// qkv fp32 [rows, 3 * width] -> attention output planes [rows, 2 * width]; ragged / prefix-sharing as launch_attention
hipError_t launch_attention_split(const float* qkv, uint16_t* out, const int32_t* starts, int n_seq, int seq_len, int heads,
                                  int causal, hipStream_t stream, const int32_t* pfx = nullptr);

// ---- sd_ops.hip / sd_attention.hip: latent-diffusion reference generator (bf16 token-major activations)
hipError_t sd_im2col3x3(const uint16_t* in, uint16_t* out, int n, int Hi, int Wi, int C, int stride, int up, hipStream_t st);
hipError_t sd_im2col_in(const float* in, uint16_t* out, int n, int Cin, int H, int W, int Kp, float scale, hipStream_t st);
hipError_t sd_groupnorm(const uint16_t* x, const float* tadd, int64_t ld_t, const float* gamma, const float* beta, uint16_t* y,
                        int n, int H, int W, int C, int groups, float eps, int silu, int in_pad, int out_pad, float* ws,
                        hipStream_t st);
hipError_t sd_relayout(const uint16_t* in, uint16_t* out, int n, int H, int W, int C, int in_pad, int out_pad, int up, hipStream_t st);
hipError_t sd_add_padded(const uint16_t* a, const uint16_t* b_padded, uint16_t* out, int n, int H, int W, int C, hipStream_t st);
hipError_t sd_layernorm_bf16(const uint16_t* x, const float* g, const float* b, uint16_t* y, int64_t rows, int C, float eps, hipStream_t st,
                             const uint16_t* add = nullptr, uint16_t* sum_out = nullptr);     // add: y = LN(bf16(x + add)), the sum also to sum_out
hipError_t sd_geglu(const uint16_t* in, uint16_t* out, int64_t rows, int Ch, hipStream_t st);
hipError_t sd_add_bf16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, hipStream_t st);
hipError_t sd_concat(const uint16_t* a, int Ca, const uint16_t* b, int Cb, uint16_t* out, int64_t tokens, hipStream_t st);
hipError_t sd_cast_silu(const float* in, uint16_t* out, int64_t n, int silu, hipStream_t st);
hipError_t sd_tokens_to_nchw(const float* in, int64_t ld, float* out, int n, int C, int H, int W, float mul, float add, int clamp01,
                             int in_pad, hipStream_t st);
hipError_t sd_pointwise_small(const float* in, const float* w, const float* bias, float* out, int n, int C, int HW, float in_scale, hipStream_t st);
hipError_t sd_cfg(const float* e, float* out, int64_t n, float g, hipStream_t st);
hipError_t sd_lincomb(float* out, const float* sample, float cs, float ce, const float* e0, float c0, const float* e1, float c1,
                      const float* e2, float c2, const float* e3, float c3, int64_t n, hipStream_t st);
hipError_t sd_softmax_rows(const float* s, uint16_t* p, int64_t rows, int T, float scale, hipStream_t st);
hipError_t sd_nchw_to_tokens(const float* in, uint16_t* out, int n, int C, int HW, hipStream_t st);
hipError_t sd_tokens_bf16_to_nchw(const uint16_t* in, float* out, int n, int C, int HW, hipStream_t st);
hipError_t sd_timestep_embed(uint16_t* out, int n, int dim, float t, hipStream_t st);
// Q [n * Tq, ldq], K / V [n * Tk, ldk / ldv], O [n * Tq, ldo] bf16; head h = columns [h * dh, (h + 1) * dh); dh % 8 == 0, <= 160
hipError_t sd_flash_attention(const uint16_t* Q, int64_t ldq, const uint16_t* K, int64_t ldk, const uint16_t* V, int64_t ldv,
                              uint16_t* O, int64_t ldo, int n, int heads, int Tq, int Tk, int dh, hipStream_t st);
hipError_t sd_resize_norm(const float* in, float* out, int n, int H, int W, int Hr, int Wr, int oy, int ox, int S, int cubic,
                          const float* mean, const float* sd, hipStream_t st);
