/*
 * tvc.h -- C ABI of the MI355X-native text-variant-consistency (TVC) hot path.
 *
 * The reference (Zhang-Xin-Duke/multimodal-detection-consistency) is pure
 * Python and has no FFI; its boundary for this path is duck-typed Python
 * (SURVEY.md section 8b).  This header is the C-ABI that sits underneath the
 * Python mirror in multimodal-detection-consistency_amd/: plain pointers and
 * sizes, no torch types.  Each entry point names the reference call sites it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer owned by the caller
 *     (in practice a PyTorch-ROCm tensor's data_ptr());
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and
 *     the call never synchronises the device (exceptions are documented);
 *   - return value 0 = TVC_OK, otherwise an error code; the message is
 *     available from tvc_last_error();
 *   - a handle belongs to one GPU and is not thread-safe (the Python shim
 *     holds a lock around submission); entry points that share an internal
 *     workspace (two calls into the same tower, any two tvc_sd_* calls, two
 *     bank searches) must be issued on ONE stream or be ordered by the caller
 *     -- the vision tower, the text tower and the bank search each have their
 *     own workspace and may run on three streams side by side;
 *   - bf16 buffers are raw uint16_t bit patterns (round-to-nearest-even).
 */
#ifndef TVC_H_
#define TVC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVC_ABI_VERSION 4
#define TVC_MAX_BANKS 8      /* bank slots per handle (tvc_bank_select)          */
#define TVC_MAX_TOPK 128     /* largest k of tvc_bank_search / tvc_topk_merge    */

enum {
    TVC_OK = 0,
    TVC_E_INVALID = 1,     /* bad argument / unsupported geometry            */
    TVC_E_HIP = 2,         /* a HIP runtime call failed                      */
    TVC_E_NOMEM = 3,       /* workspace allocation failed                    */
    TVC_E_STATE = 4,       /* e.g. bank search before tvc_bank_set           */
    TVC_E_OVERFLOW = 5     /* candidate lists overflowed (see tvc_bank_status) */
};

enum { TVC_DTYPE_BF16 = 0, TVC_DTYPE_F32 = 1 };

typedef struct tvc_handle tvc_handle;

/* Transformer tower geometry (CLIP style, pre-LN). */
enum { TVC_ACT_QUICK_GELU = 0, TVC_ACT_GELU = 1 };
typedef struct {
    int32_t width;    /* d_model                          */
    int32_t layers;
    int32_t heads;    /* head_dim = width / heads, must be 64 */
    int32_t mlp;      /* hidden size of the MLP           */
    int32_t act;      /* TVC_ACT_QUICK_GELU: x * sigmoid(1.702 x) (OpenAI CLIP, every tower the detector uses);
                         TVC_ACT_GELU: the exact erf GELU (OpenCLIP ViT-H/14, the text encoder of Stable Diffusion 2.x) --
                         applied by a row kernel after a store-only FC1, not in the GEMM epilogue; not available to
                         tvc_encode_image_grad */
} tvc_tower_arch;

typedef struct {
    int32_t image_size;   /* 224                                  */
    int32_t patch;        /* 32 (ViT-B/32), 14 (ViT-L/14)         */
    int32_t vocab;        /* 49408                                */
    int32_t ctx;          /* 77                                   */
    int32_t embed_dim;    /* D: 512 (B/32), 768 (L/14)            */
    tvc_tower_arch vision;
    tvc_tower_arch text;
} tvc_model_desc;

/* One residual attention block.  GEMM weights are bf16 [out, in] row-major
 * (the nn.Linear layout); LayerNorm parameters and biases are fp32. */
typedef struct {
    const float*    ln1_g;  const float* ln1_b;
    const uint16_t* wqkv;   /* [3d, d]  rows: q | k | v            */
    const float*    bqkv;   /* [3d]                                 */
    const uint16_t* wo;     /* [d, d]                               */
    const float*    bo;
    const float*    ln2_g;  const float* ln2_b;
    const uint16_t* w1;     /* [mlp, d]                             */
    const float*    b1;
    const uint16_t* w2;     /* [d, mlp]                             */
    const float*    b2;
} tvc_layer_weights;

typedef struct {
    const uint16_t* patch_w;   /* bf16 [d, Kp], Kp = round_up(3*patch*patch, 64), zero padded;
                                  columns ordered (c, ky, kx) like the conv weight          */
    const float* cls;          /* [d]                                                       */
    const float* pos;          /* [T, d], T = (image_size/patch)^2 + 1                      */
    const float* ln_pre_g;  const float* ln_pre_b;
    const float* ln_post_g; const float* ln_post_b;
    const uint16_t* proj;      /* bf16 [D, d]                                               */
    const tvc_layer_weights* layers;   /* HOST array of `vision.layers` structs            */
} tvc_vision_weights;

typedef struct {
    const float* tok_emb;      /* fp32 [vocab, d]                                           */
    const float* pos;          /* fp32 [ctx, d]                                             */
    const float* ln_final_g; const float* ln_final_b;
    const uint16_t* proj;      /* bf16 [D, d]                                               */
    const tvc_layer_weights* layers;   /* HOST array of `text.layers` structs              */
} tvc_text_weights;

/* fp32 copies of the same weights for the fp32-grade tower mode (TVC_OPT_TOWER_PRECISION): identical field order,
 * every tensor fp32, nothing rounded to bf16.  patch_w is the UNPADDED conv weight [d, 3*patch*patch]. */
typedef struct {
    const float* ln1_g;  const float* ln1_b;
    const float* wqkv;   const float* bqkv;
    const float* wo;     const float* bo;
    const float* ln2_g;  const float* ln2_b;
    const float* w1;     const float* b1;
    const float* w2;     const float* b2;
} tvc_layer_weights_f32;

typedef struct {
    const float* patch_w;      /* fp32 [d, 3*patch*patch], columns ordered (c, ky, kx) */
    const float* cls;  const float* pos;
    const float* ln_pre_g;  const float* ln_pre_b;
    const float* ln_post_g; const float* ln_post_b;
    const float* proj;         /* fp32 [D, d] */
    const tvc_layer_weights_f32* layers;   /* HOST array of `vision.layers` structs */
} tvc_vision_weights_f32;

typedef struct {
    const float* tok_emb;  const float* pos;
    const float* ln_final_g; const float* ln_final_b;
    const float* proj;         /* fp32 [D, d] */
    const tvc_layer_weights_f32* layers;   /* HOST array of `text.layers` structs */
} tvc_text_weights_f32;

/* ---- lifetime -------------------------------------------------------- */

uint32_t tvc_abi_version(void);

/* Create a handle on the current HIP device.  `vision`/`text` may be NULL
 * (no tower of that kind: tvc_encode_* then fails with TVC_E_STATE; `desc`
 * may be NULL when both are).  The weight buffers are referenced, not copied:
 * the caller keeps them alive.
 * Replaces: constructing the (absent) src.models CLIPModel(CLIPConfig(...))
 * -- src/detector.py:258-271, src/retrieval.py:356-361. */
int tvc_create(const tvc_model_desc* desc, const tvc_vision_weights* vision,
               const tvc_text_weights* text, tvc_handle** out);
void tvc_destroy(tvc_handle* h);

/* Message of the last error on `h` (or of the last failed tvc_create when
 * h == NULL).  Never NULL. */
const char* tvc_last_error(tvc_handle* h);

/* Options.  TVC_OPT_TEXT_PACKING (default 1): the text tower processes only the
 * tokens up to and including EOT of every text (ragged, packed rows).  Under the
 * causal mask later positions cannot influence the pooled EOT row, so the output
 * is bit-identical to the dense [T, ctx] computation while the work drops by
 * ctx / mean length.  With packing on, tvc_encode_text synchronises `stream`
 * once per call (it reads back the packed row count that sizes the GEMM grids).
 * TVC_OPT_MAX_CHUNK_IMAGES / _TEXTS: rows of a batch processed per pass
 * (workspace bound; defaults 512 / 4608).
 * TVC_OPT_BANK_FILTER (default 1): when tvc_bank_search is called without a moments
 * buffer, the bank pass multiplies one bf16 product per element, keeps every row whose
 * similarity could exceed the running bound (Cauchy-Schwarz margin from the bank's
 * largest row norms) and re-scores the kept rows in fp32: the same top-k set, values at
 * least as accurate, 1/2 (bf16 bank) or 1/3 (fp32 bank) of the matrix-core work.  0 forces
 * the all-products pass.
 * TVC_OPT_TEXT_GROUP (default 0 = off): G >= 2 declares that the rows passed to tvc_encode_text
 * come in consecutive groups of G texts whose first text is the original and the others its
 * variants (the layout of pipeline.detect: original + N variants).  A causal tower gives two texts
 * identical hidden states on their common token prefix, so a variant keeps only the rows from its
 * first differing token on and attends to the original's rows for the shared prefix.  Outputs are
 * bit-identical; needs text packing and T % G == 0, otherwise it is ignored for that call.
 * TVC_OPT_POOLED_LAST_LAYER (default 1): both towers are pooled at ONE token (the class token / the EOT token), so
 * in their LAST layer only that token's attention output, out-projection, ln_2 and MLP are ever read.  With the
 * option on, those are computed for the pooled rows only (K and V still come from every token): the embeddings
 * are unchanged (same fp32 sums in the same order) while 10/12 of the last layer's GEMM work is not done.  0 runs
 * the last layer over every token like the others.
 * TVC_OPT_TOWER_PRECISION (default 0): 0 = the towers multiply bf16 x bf16 on the MFMA units with fp32 accumulation
 * (the benchmarked path: embeddings within ~1e-3 of an fp32 CPU tower).  1 = fp32-grade towers: tvc_encode_image /
 * tvc_encode_text / tvc_encode_text_hidden run every GEMM on the exact-f32 matrix instruction (v_mfma_f32_32x32x2_f32)
 * with the fp32 weights registered by tvc_set_weights_f32, fp32 activations and fp32 attention -- embeddings within
 * ~1e-6 of the reference's fp32 CPU path (src/detector.py:461-485; configs/attacks/pgd.yaml:80 asks for fp32), so the
 * consistency scores meet the 1e-4 bar END TO END.  About 10x slower; validation and attack-generation mode, never the
 * benchmarked one.  Needs tvc_set_weights_f32 first (TVC_E_STATE otherwise).  The input-gradient entry points
 * (tvc_encode_image_grad / _backward) always run the bf16 path.
 * 2 = split-bf16 towers, the FAST <= 1e-4 mode: every activation and weight travels to the matrix cores as hi | lo bf16
 * planes (x = hi + lo to ~2^-17) and every product is three bf16 MFMAs (hi hi + hi lo + lo hi, fp32 accumulation); fp32
 * residual stream, LayerNorm, softmax and QuickGELU.  Scores within 1e-4 of the reference's fp32 CPU path END TO END at
 * about a third of the bf16 mode's matrix rate (mode 1: 1/16); EOT packing and prefix sharing are kept, the pooled last
 * layer is not.  Needs tvc_set_weights_f32 first; setting the option builds the weight planes (synchronises the device).
 * Geometry limits: head_dim 64 and sequences of at most 288 tokens in modes 0 and 1 (tvc_create refuses other towers),
 * at most 272 tokens in mode 2 (TVC_E_INVALID when the option is set).
 * TVC_OPT_SD_ARENA_BYTES (default 48 GiB): budget of the activation arena of ONE UNet evaluation inside tvc_sd_generate.
 * The arena grows linearly with the samples of an evaluation (about 0.75 GB per image at 64 x 64 latents: both halves of
 * classifier-free guidance); a batch that would exceed the budget is generated in chunks of whole sampling loops -- every
 * image is independent of its batch mates, so chunking changes no pixel -- instead of failing with TVC_E_NOMEM.
 * TVC_OPT_SD_STREAMS (default 2; 1 = off): inside tvc_sd_generate the unconditional and the conditional half of a UNet
 * evaluation run on two HIP streams (the caller's and one the handle owns, forked / joined by events), each in its own
 * half of the arena: a launch of one half fills the compute units the other half's partial tile round leaves idle.  Every
 * sample's arithmetic is independent of its batch mates, so the images are bit-identical with 1 and 2. */
enum { TVC_OPT_TEXT_PACKING = 1, TVC_OPT_MAX_CHUNK_IMAGES = 2, TVC_OPT_MAX_CHUNK_TEXTS = 3,
       TVC_OPT_BANK_FILTER = 4, TVC_OPT_TEXT_GROUP = 5, TVC_OPT_POOLED_LAST_LAYER = 6, TVC_OPT_TOWER_PRECISION = 7,
       TVC_OPT_SD_ARENA_BYTES = 8, TVC_OPT_SD_STREAMS = 9 };
int tvc_set_option(tvc_handle* h, int32_t option, int64_t value);

/* Register fp32 copies of the tower weights for TVC_OPT_TOWER_PRECISION = 1 (either may be NULL).  Referenced, not
 * copied: the caller keeps the buffers alive.  Geometry = the desc given to tvc_create.
 * Replaces: loading the fp32 checkpoint in the (absent) src.models CLIPModel -- src/detector.py:258-271. */
int tvc_set_weights_f32(tvc_handle* h, const tvc_vision_weights_f32* vision, const tvc_text_weights_f32* text);

/* Bytes of device workspace currently held by the handle. */
uint64_t tvc_workspace_bytes(tvc_handle* h);

/* ---- encoders (K1, K2, K3 of SURVEY.md 2.3) -------------------------- */

/* pix_dev: fp32 [B, 3, image_size, image_size] (already preprocessed);
 * out_dev: fp32 [B, D], L2-normalised when `normalize` != 0.
 * Replaces clip_model.encode_image / encode_image_tensor(x, requires_grad=False):
 * src/detector.py:626,633; experiments/defenses/detector.py:238;
 * src/retrieval.py:407,609. */
int tvc_encode_image(tvc_handle* h, const float* pix_dev, int32_t B,
                     float* out_dev, int32_t normalize, void* stream);

/* tok_dev: int32 [T, ctx] CLIP BPE ids (SOT ... EOT, 0-padded); pooled at the
 * arg-max id (the EOT token).  out_dev: fp32 [T, D].
 * Replaces clip_model.encode_text: experiments/defenses/detector.py:239,247;
 * experiments/defenses/retrieval_ref.py:238-244; src/retrieval.py:451,551. */
int tvc_encode_text(tvc_handle* h, const int32_t* tok_dev, int32_t T,
                    float* out_dev, int32_t normalize, void* stream);

/* ---- reference bank (K5) --------------------------------------------- */

/* A handle holds TVC_MAX_BANKS independent bank slots so that several owners sharing one engine (the
 * retriever's image index src/retrieval.py:225, a ReferenceBank src/ref_bank.py:86, the defense detector's
 * reference features retrieval_ref.py:99) cannot replace each other's rows.  tvc_bank_select picks the slot
 * that tvc_bank_set / _search / _search_dense / _gather address from then on (default: slot 0). */
int tvc_bank_select(tvc_handle* h, int32_t slot);

/* Register the bank: dense row-major [R, D], rows L2-normalised
 * (scripts/build_faiss_indices.py:108-109,138; retrieval_ref.py:99,156).
 * dtype TVC_DTYPE_BF16: used in place (caller keeps it alive).
 * dtype TVC_DTYPE_F32: split once into bf16 (hi, lo) planes inside the handle
 * (3-product split-bf16 GEMM, fp32-grade cosines).
 * Replaces faiss.IndexFlatIP(d).add(features): src/retrieval.py:225-226,
 * retrieval_ref.py:140,156. */
int tvc_bank_set(tvc_handle* h, const void* bank_dev, int64_t R, int32_t D,
                 int32_t dtype, void* stream);

/* Exact top-k inner-product search of M query rows against the bank, fused
 * with per-row moments; the [M, R] matrix is never materialised.
 *   rows_dev   fp32 [M, D] (L2-normalised by the caller when cosines are wanted)
 *   k          1..TVC_MAX_TOPK (128)
 *   topk_idx   int32 [M, k]  global row index (local + idx_offset), -1 = none
 *   topk_sim   fp32  [M, k]  descending; ties broken by ascending index
 *   moments    fp32  [M, 4]  sum, sum of squares, max, count(sim >= count_thr)
 *                            over ALL R rows (may be NULL)
 * Replaces index.search(q, k) / np.dot + argpartition + argsort:
 * src/retrieval.py:636-673, retrieval_ref.py:246-290, src/ref_bank.py:475-484.
 * Returns TVC_OK after enqueueing; overflow of the internal candidate lists is
 * reported by tvc_bank_status() after the stream has been synchronised. */
int tvc_bank_search(tvc_handle* h, const float* rows_dev, int32_t M, int32_t k,
                    float count_thr, int64_t idx_offset,
                    int32_t* topk_idx_dev, float* topk_sim_dev, float* moments_dev,
                    void* stream);

/* Same contract as tvc_bank_search, computed by brute force (the similarities of up to 64 query
 * rows at a time are materialised and reduced): the fallback for degenerate banks on which
 * tvc_bank_status reports TVC_E_OVERFLOW.  Orders of magnitude slower; never overflows. */
int tvc_bank_search_dense(tvc_handle* h, const float* rows_dev, int32_t M, int32_t k,
                          float count_thr, int64_t idx_offset,
                          int32_t* topk_idx_dev, float* topk_sim_dev, float* moments_dev,
                          void* stream);

/* Synchronises `stream` and returns TVC_E_OVERFLOW if the last
 * tvc_bank_search dropped candidates (degenerate banks, e.g. thousands of
 * identical rows); TVC_OK otherwise. */
int tvc_bank_status(tvc_handle* h, void* stream);

/* out_dev fp32 [n, D]: bank rows for LOCAL indices idx_dev[n] (global index -
 * idx_offset); rows with idx < 0 or >= R are zero-filled.
 * Replaces self.reference_features[idx] (retrieval_ref.py:262,286). */
int tvc_bank_gather(tvc_handle* h, const int32_t* idx_dev, int32_t n,
                    int64_t idx_offset, float* out_dev, void* stream);

/* Merge W per-shard partial results (the RCCL all-gather / all-to-all output)
 * into the global top-k: parts are [W, M, k] (idx, sim) each sorted descending,
 * feat_parts fp32 [W, M, kf, D] or NULL carries the rows of the first kf
 * entries of every part; outputs as tvc_bank_search plus feat_out [M, kf, D].
 * mom_parts [W, M, 4] / mom_out [M, 4] may be NULL.  Limits: W * k <= 256, kf <= min(k, 32).
 * Replaces dist.all_gather + host merge (src/utils/multi_gpu_processor.py:595-612). */
int tvc_topk_merge(tvc_handle* h, const int32_t* idx_parts_dev, const float* sim_parts_dev,
                   const float* feat_parts_dev, const float* mom_parts_dev,
                   int32_t W, int32_t M, int32_t k, int32_t kf, int32_t D,
                   int32_t* idx_out_dev, float* sim_out_dev, float* feat_out_dev,
                   float* mom_out_dev, void* stream);

/* All-pairs cosine matrix out[n, m] = cos(x[n], y[m]), fp32-grade (both sides
 * L2-normalised on device, split into bf16 (hi, lo) planes, three MFMA products).
 * x fp32 [N, D], y fp32 [M, D], out fp32 [N, M]; D % 64 == 0.
 * Replaces sklearn cosine_similarity / F.normalize + mm:
 * src/retrieval.py:706 (compute_similarity_matrix), src/utils/metrics.py:144-164. */
int tvc_cosine_matrix(tvc_handle* h, const float* x_dev, int32_t N, const float* y_dev, int32_t M,
                      int32_t D, float* out_dev, void* stream);

/* Token-level output of the text tower: out fp32 [T, ctx, width] = ln_final(hidden states) at EVERY position
 * (no pooling, no projection, no EOT packing; causal mask as in tvc_encode_text) -- what a latent-diffusion
 * pipeline conditions its UNet on (`CLIPTextModel(...).last_hidden_state`; SURVEY.md 8f rank 1: the text encoder of
 * the SD reference generator, src/sd_ref.py:389-412 via StableDiffusionModel.generate_image). */
int tvc_encode_text_hidden(tvc_handle* h, const int32_t* tok_dev, int32_t T, float* out_dev, void* stream);

/* ---- input gradient of the vision tower -------- */

/* What `encode_image_tensor(x, requires_grad=True)` + `loss.backward()` give a white-box attack
 * (src/attacks/pgd_attack.py:456-486, src/attacks/hubness_attack.py:269-424, src/models/clip_model.py:200-221):
 * the embedding AND d(loss)/d(pixels).  Only the INPUT gradient exists -- the weights are frozen in every
 * attack of the reference -- so no weight-gradient GEMM is ever run.
 *
 * tvc_encode_image_grad: the tower of tvc_encode_image (every layer over every token; the FC1 pre-activation is
 *   rounded to bf16 before QuickGELU because it is kept: embeddings agree with tvc_encode_image within the tower's
 *   bf16 tolerance, ~1e-3, not bitwise; B must fit one pass, B <= TVC_OPT_MAX_CHUNK_IMAGES) and KEEPS, inside the handle, what the backward of each layer reads: its
 *   fp32 input rows, its QKV rows, its out-projection output and the FC1 pre-activation (20 * width bytes per
 *   token row and layer: 3.9 GB at ViT-L/14 and 32 images), plus the un-normalised embedding and the ln_post
 *   input.  pix_dev must stay valid and unchanged until the matching backward (the stem is recomputed from it).
 * tvc_encode_image_backward: grad_out fp32 [B, D] = d(loss)/d(out of the LAST tvc_encode_image_grad on this
 *   handle) -> grad_pix fp32 [B, 3, S, S] in the preprocessed (normalised) pixel space.  Four dX GEMMs per layer
 *   (= the forward's GEMM work once more) with fused row kernels between them and a two-pass attention backward;
 *   nothing of the forward is recomputed.  Gradients travel between GEMMs in bf16 and accumulate along the
 *   residual stream in fp32.  Deterministic (no atomics).  May be called repeatedly for different grad_out. */
int tvc_encode_image_grad(tvc_handle* h, const float* pix_dev, int32_t B, float* out_dev, int32_t normalize, void* stream);
int tvc_encode_image_backward(tvc_handle* h, const float* grad_out_dev, float* grad_pix_dev, void* stream);

/* One projected-gradient step on a batch of B images of n elements each, in place on adv_dev
 * (src/attacks/pgd_attack.py:500-521):
 *   momentum_dev != NULL:  m = mu * m + grad / |grad|_1 (per image);  step direction sign(m);  else sign(grad)
 *   adv = clamp(clean + clamp(adv +- alpha * sign - clean, -eps, eps), clip_min, clip_max)   ('-' when targeted) */
int tvc_pgd_step(tvc_handle* h, float* adv_dev, const float* clean_dev, const float* grad_dev, float* momentum_dev,
                 int32_t B, int64_t n, float eps, float alpha, float mu, float clip_min, float clip_max,
                 int32_t targeted, void* stream);

/* The L2-constrained step of the Hubness attack (src/attacks/hubness_attack.py:378-386), in place on adv_dev, per image:
 *   adv += (descent ? -1 : +1) * step * grad / (|grad|_2 + 1e-8);  d = adv - clean;  d *= min(|d|_2, eps) / (|d|_2 + 1e-8);
 *   adv = clamp(clean + d, clip_min, clip_max) */
int tvc_l2_step(tvc_handle* h, float* adv_dev, const float* clean_dev, const float* grad_dev, int32_t B, int64_t n, float eps,
                float step, float clip_min, float clip_max, int32_t descent, void* stream);

/* ---- per-query consistency (K4, K6, K7) ------------------------------ */

typedef struct {
    int32_t reference_count;      /* refs kept per text row, retrieval_ref.py:23 (5)   */
    float   similarity_threshold; /* retrieval_ref.py:24 (0.3)                         */
    int32_t retrieval_top_k;      /* experiments/defenses/detector.py:29 (10), <= 16   */
    float   dup_threshold;        /* experiments/defenses/detector.py:318 (0.95)       */
    float   w_text_variants;      /* src/detector.py:667 (0.4); 0 = method off.  With N == 0 and a weight > 0 the
                                     method still enters the weighted mean with score 0.0 (augmenter present but
                                     no variants, src/detector.py:375-378,457-458)                            */
    float   w_consistency;        /* src/detector.py:669 (0.2)                         */
    float   w_exp[4];             /* consistency_checker.py:61-66 (0.25 each): original,
                                     text_variant, retrieval, generative               */
} tvc_consistency_params;

/* Record layout, fp32 words per query (tvc_consistency writes rec_stride words):
 *  [0] original_similarity s0      [1] mean(sv)       [2] std(sv) (ddof 0)
 *  [3] text-variant score  (src/detector.py:479-485)
 *  [4] consistency score 1-s0 (src/detector.py:579)
 *  [5] aggregated src score (weighted_mean over the two, src/detector.py:664-680)
 *  [6] retrieval_consistency  [7] retrieval_std  [8] number of refs kept
 *  [9] cross_modal_variance (experiments/defenses/detector.py:295-300)
 * [10] overall score, weighted voting (consistency_checker.py:147-160)
 * [11] reserved
 * [12 .. 12+N)        sv[n]
 * [12+N .. +16)       kept reference indices (int32 bit patterns, -1 padded)
 * [12+N+16 .. +16)    cos(image, kept reference) */
#define TVC_REC_HEAD 12
#define TVC_REC_MAXREF 16
static inline int32_t tvc_rec_stride(int32_t N) { return TVC_REC_HEAD + N + 2 * TVC_REC_MAXREF; }

/* img_dev fp32 [B, D]; txt_dev fp32 [B, N+1, D] (row 0 = original text);
 * ref_idx/ref_sim [B*(N+1), ks] = bank search results of the text rows
 * (global indices), ref_feat fp32 [B*(N+1), kf, D] = rows of the first kf
 * (>= reference_count) results; pass ks = 0 / NULLs for "no bank".
 * rec_dev fp32 [B, tvc_rec_stride(N)].
 * Replaces src/detector.py:461-485,573-579,643-682 and
 * experiments/defenses/detector.py:184-204,228-300,302-325. */
int tvc_consistency(tvc_handle* h, const float* img_dev, const float* txt_dev,
                    int32_t B, int32_t N, int32_t D,
                    const int32_t* ref_idx_dev, const float* ref_sim_dev,
                    const float* ref_feat_dev, int32_t ks, int32_t kf,
                    const tvc_consistency_params* params, float* rec_dev, void* stream);

/* ---- in-process kernel timing (HIP events on the launch stream) ------- */

enum { TVC_PROF_GEMM = 0, TVC_PROF_ATTENTION = 1, TVC_PROF_BANK = 2, TVC_PROF_ROWOPS = 3, TVC_PROF_NCAT = 4 };

/* After tvc_profile_begin every kernel launch made through `h` is bracketed by
 * a pair of hipEvents on its stream.  tvc_profile_end synchronises, then fills
 * per category (index = TVC_PROF_*): ms[c] = summed kernel time, work[c] =
 * summed algorithmic FLOPs (GEMM, attention, bank) or bytes (row ops),
 * launches[c] = number of launches; profiling is switched off again.  big_gemm (double[3], may be NULL): over the GEMM
 * launches of >= 64 output tiles (the persistent ring kernels, the `roofline` kernel of bench.py) -- [0] their summed
 * COMPULSORY HBM bytes (every distinct operand plane read once, the output written once), [1] their count, [2] their ms:
 * the denominator that roofline.traffic (PMC bytes per launch) is compared with. */
int tvc_profile_begin(tvc_handle* h);
int tvc_profile_end(tvc_handle* h, double* ms, double* work, int64_t* launches, double* big_gemm);

/* ---- building blocks exported for parity tests and profiling --------- */

/* out[j, i] = sum_k a[i, k] * b[j, k]  (+ bias[i]); a bf16 [I, lda] ("weights"),
 * b bf16 [J, ldb] ("tokens"); K % 64 == 0; lda / ldb = row strides in elements (0 = K, i.e. dense rows;
 * otherwise >= K and a multiple of 8).  epilogue: 0 = fp32 store, 1 = bf16 store, 2 = bf16 quick-GELU,
 * 3 = fp32 residual add (out += ...). */
int tvc_gemm_bf16(tvc_handle* h, const uint16_t* a_dev, const uint16_t* b_dev,
                  const float* bias_dev, void* out_dev, int32_t I, int32_t J, int32_t K,
                  int64_t lda, int64_t ldb, int32_t ld_out, int32_t epilogue, void* stream);

/* Multi-head attention over packed sequences: qkv bf16 [rows, 3*width]
 * (q | k | v), sequences of `seq_len` consecutive rows, out bf16 [rows, width]. */
int tvc_attention(tvc_handle* h, const uint16_t* qkv_dev, uint16_t* out_dev,
                  int32_t n_seq, int32_t seq_len, int32_t heads, int32_t causal, void* stream);

/* y bf16 [rows, d] = LayerNorm(x fp32 [rows, d]) * g + b, eps 1e-5. */
int tvc_layernorm(tvc_handle* h, const float* x_dev, const float* g_dev, const float* b_dev,
                  uint16_t* y_dev, int32_t rows, int32_t d, void* stream);

/* Backward of tvc_attention (non-causal, fixed-length sequences, head_dim 64): qkv as the forward saw it,
 * dout bf16 [rows, width] = gradient w.r.t. the attention output, dqkv bf16 [rows, 3*width] (dq | dk | dv). */
int tvc_attention_backward(tvc_handle* h, const uint16_t* qkv_dev, const uint16_t* dout_dev, uint16_t* dqkv_dev,
                           int32_t n_seq, int32_t seq_len, int32_t heads, void* stream);

/* LayerNorm input gradient: x fp32 [rows, d] (the forward's input), dy bf16 [rows, d], gamma fp32 [d],
 * dres fp32 [rows, d] or NULL (added: the residual path's gradient), dx fp32 [rows, d]. */
int tvc_layernorm_backward(tvc_handle* h, const float* x_dev, const uint16_t* dy_dev, const float* g_dev,
                           const float* dres_dev, float* dx_dev, int32_t rows, int32_t d, void* stream);

/* ---- latent-diffusion reference generator (BASELINE configs[4]) -------------------------
 * What src/sd_ref.py:389-399 (StableDiffusionModel.generate_image) and experiments/defenses/generative_ref.py:139-147
 * (sd_model.generate) reach through diffusers.StableDiffusionPipeline: the UNet2DConditionModel denoising loop with
 * classifier-free guidance under the PNDM (PLMS) scheduler, then AutoencoderKL.decode.  Geometry = the config.json
 * files the reference holds (cache/sd/models--runwayml--stable-diffusion-v1-5/snapshots/<rev>/{unet,vae,scheduler}), or
 * Stable Diffusion 2.x (src/__init__.py:110-113 lists stable-diffusion-2-1): per-level head counts, linear proj_in /
 * proj_out ([C, C] weights -- a 1 x 1 convolution on token rows is the same GEMM), cross-attention width 1024,
 * optionally v-prediction.
 * Activations are bf16 token-major (NHWC) between GEMMs, statistics / softmax / scheduler arithmetic fp32. */
typedef struct {
    int32_t in_channels, out_channels;           /* 4, 4                                            */
    int32_t n_blocks;                            /* 4                                               */
    int32_t block_out_channels[4];               /* 320, 640, 1280, 1280                            */
    int32_t down_block_attn[4];                  /* 1, 1, 1, 0 (CrossAttnDownBlock2D x3, DownBlock2D) */
    int32_t layers_per_block;                    /* 2                                               */
    int32_t heads;                               /* 8 ("attention_head_dim": 8 = the head count)    */
    int32_t heads_per_block[4];                  /* SD 2.x: 5, 10, 20, 20 (head dim 64 at every level); all 0 = `heads` everywhere */
    int32_t prediction_type;                     /* 0 = epsilon (SD 1.x, 2.1-base), 1 = v_prediction (2.1 at 768 px)  */
    int32_t cross_attention_dim;                 /* 768                                             */
    int32_t norm_groups;                         /* 32                                              */
    float   norm_eps;                            /* 1e-5                                            */
    int32_t vae_n_blocks;                        /* 4                                               */
    int32_t vae_block_out_channels[4];           /* 128, 256, 512, 512                              */
    int32_t vae_layers_per_block;                /* 2                                               */
    int32_t latent_channels;                     /* 4                                               */
    float   vae_scaling;                         /* 0.18215                                         */
    int32_t ctx;                                 /* 77                                              */
    float   beta_start, beta_end;                /* 0.00085, 0.012 (scaled_linear)                  */
    int32_t num_train_timesteps, steps_offset;   /* 1000, 1                                         */
} tvc_sd_desc;

typedef struct { const char* name; const void* ptr; } tvc_named_tensor;

/* Register the model.  `tensors` = device pointers keyed by the diffusers state-dict names (UNet names as they are,
 * VAE names as in AutoencoderKL: "decoder....", "post_quant_conv...."), prepared by the host as follows:
 *   3x3 conv weight [Co, Ci, 3, 3]  -> bf16 [Co, 9 * Ci], column (ky * 3 + kx) * Ci + ci ("conv_in": zero padded to 64 columns)
 *   1x1 conv / linear weight        -> bf16 [Co, Ci]
 *   attn1.to_q|to_k|to_v            -> one bf16 "....attn1.to_qkv.weight" [3C, C];  attn2.to_k|to_v -> "....attn2.to_kv.weight" [2C, 768]
 *   VAE query|key|value             -> "....to_qkv.weight" bf16 [3C, C] and "....to_qkv.bias" fp32 [3C]
 *   biases, norm gains / offsets, post_quant_conv.weight [4, 4]  -> fp32
 *   every bf16 matrix must be READABLE up to the next multiple of 256 rows (zero rows appended by the host): the GEMM
 *   stages whole 256-row tiles of it even where Co (320, 640, 4 ...) is not a multiple of 256
 * Referenced, not copied (the caller keeps them alive), except the resnets' time projections, which are gathered into
 * one matrix inside the handle.  Either half may be absent (UNet-only / VAE-only handles). */
int tvc_sd_load(tvc_handle* h, const tvc_sd_desc* desc, const tvc_named_tensor* tensors, int32_t n_tensors, void* stream);

/* One UNet evaluation: latents fp32 [n, 4, H, W], ctx fp32 [n, ctx, cross_attention_dim] (tvc_encode_text_hidden of
 * the CLIP ViT-L/14 text tower), scalar timestep -> predicted noise fp32 [n, 4, H, W].  H, W multiples of 8. */
int tvc_sd_unet(tvc_handle* h, const float* latents_dev, int32_t n, int32_t H, int32_t W, float timestep,
                const float* ctx_dev, float* eps_dev, void* stream);

/* AutoencoderKL.decode(latents / scaling_factor) -> images fp32 [n, 3, 8H, 8W], (x / 2 + 0.5) clamped to [0, 1]. */
int tvc_sd_vae_decode(tvc_handle* h, const float* latents_dev, int32_t n, int32_t H, int32_t W, float* images_dev, void* stream);

/* The whole sampling loop for n images: cond / uncond fp32 [n, ctx, cross_attention_dim], latents fp32 [n, 4, H, W]
 * (in: the initial noise; out: the final latents), `steps` PNDM steps (steps + 1 UNet evaluations on 2n samples each),
 * guidance scale as in the pipeline; images_dev may be NULL (latents only). */
int tvc_sd_generate(tvc_handle* h, const float* cond_dev, const float* uncond_dev, float* latents_dev, int32_t n, int32_t H,
                    int32_t W, int32_t steps, float guidance, float* images_dev, void* stream);

/* One block of the model on fp32 NCHW tensors (parity tests): kind 0 = ResnetBlock2D, 1 = Transformer2DModel, 2 = the
 * VAE AttentionBlock, 3 = 3x3 conv (stride 1), 4 = stride-2 downsample conv, 5 = nearest-2x upsample + conv;
 * `prefix` = the block's state-dict prefix with the trailing dot (conv kinds: up to and including "conv." / "conv_in.").
 * x [n, Cin, H, W]; temb fp32 [n, time_dim] (resnets of the UNet; NULL for the VAE's); ctx as tvc_sd_unet (kind 1);
 * out [n, Cout, H', W'].  vae != 0 uses the VAE's eps (1e-6). */
int tvc_sd_block(tvc_handle* h, int32_t kind, const char* prefix, const float* x_dev, int32_t n, int32_t Cin, int32_t H,
                 int32_t W, const float* temb_dev, const float* ctx_dev, int32_t Cout, int32_t vae, float* out_dev, void* stream);

/* Streaming attention of the UNet (parity tests): q [n * Tq, heads * dh], k / v [n * Tk, heads * dh], out like q; bf16. */
int tvc_sd_attention(tvc_handle* h, const uint16_t* q_dev, const uint16_t* k_dev, const uint16_t* v_dev, uint16_t* out_dev,
                     int32_t n, int32_t heads, int32_t Tq, int32_t Tk, int32_t dh, void* stream);

/* Image preprocessing on the device: images fp32 [n, 3, H, W] with values in [0, 1] -> out fp32 [n, 3, S, S]:
 * antialiased resize of the short side to S (filter 0 = bilinear: torchvision Resize as in
 * experiments/defenses/generative_ref.py:55-59 -- which resizes BOTH sides to S; 1 = bicubic: the CLIP preprocess of
 * src/attacks/hubness_attack.py:223), centre crop, (x - mean[c]) / std[c].  keep_aspect 0 resizes both sides to S.
 * PIL / torch(antialias) filter semantics on float pixels (no uint8 rounding). */
int tvc_preprocess_images(tvc_handle* h, const float* images_dev, int32_t n, int32_t H, int32_t W, int32_t S, int32_t filter,
                          int32_t keep_aspect, const float* mean3, const float* std3, float* out_dev, void* stream);

/* Building blocks of the fp32-grade tower mode (TVC_OPT_TOWER_PRECISION = 1), exported for parity tests:
 * out[j, i] (op)= sum_k x[j, k] * w[i, k] + bias[i] on the exact-f32 matrix instruction; w fp32 [I, K], x fp32 [J, K],
 * out fp32 [J, ld_out]; K % 4 == 0; epilogue 0 = store, 1 = QuickGELU, 2 = out += (residual add). */
int tvc_gemm_f32(tvc_handle* h, const float* w_dev, const float* x_dev, const float* bias_dev, float* out_dev,
                 int32_t I, int32_t J, int32_t K, int32_t ld_out, int32_t epilogue, void* stream);
/* fp32 multi-head attention, head_dim 64, seq_len <= 288: qkv fp32 [n_seq * seq_len, 3 * width], out fp32 [rows, width]. */
int tvc_attention_f32(tvc_handle* h, const float* qkv_dev, float* out_dev, int32_t n_seq, int32_t seq_len,
                      int32_t heads, int32_t causal, void* stream);

/* Building blocks of the split-bf16 tower mode (TVC_OPT_TOWER_PRECISION = 2), exported for parity tests:
 * tvc_gemm_split: out fp32 [J, ld_out] = x [J, K] w[I, K]^T + bias with both operands split into hi | lo bf16 planes and
 * three MFMA products per element (K % 4 == 0).
 * tvc_attention_split: qkv fp32 [rows, 3 * width] (head_dim 64, seq_len <= 272; starts_dev int32 [n_seq + 1] = packed
 * sequences of at most seq_len rows, or NULL = n_seq x seq_len dense rows) -> the attention output as hi | lo bf16 planes
 * [rows, 2 * width] (value = hi + lo); seq_len <= 272. */
int tvc_gemm_split(tvc_handle* h, const float* w_dev, const float* x_dev, const float* bias_dev, float* out_dev,
                   int32_t I, int32_t J, int32_t K, int32_t ld_out, void* stream);
int tvc_attention_split(tvc_handle* h, const float* qkv_dev, uint16_t* out_planes_dev, const int32_t* starts_dev,
                        int32_t n_seq, int32_t seq_len, int32_t heads, int32_t causal, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TVC_H_ */
