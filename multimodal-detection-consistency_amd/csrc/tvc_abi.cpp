// C-ABI of the TVC hot path (include/tvc.h): handle, workspace and the launch
// sequences of the CLIP towers, the bank search and the consistency kernel.
// No torch types, no exceptions across the boundary, no device synchronisation
// except where tvc.h says so.
#include "handle.hpp"

#include <cstdlib>

// tvc_precise.cpp
int tvc_precise_encode_image(tvc_handle* h, const float* pix_dev, int32_t B, float* out_dev, int32_t normalize, hipStream_t st);
int tvc_precise_encode_text(tvc_handle* h, const int32_t* tok_dev, int32_t Tn, float* out_dev, int32_t normalize,
                            float* hidden_out, hipStream_t st);

// tvc_split.cpp
int tvc_split_encode_image(tvc_handle* h, const float* pix_dev, int32_t B, float* out_dev, int32_t normalize, hipStream_t st);
int tvc_split_encode_text(tvc_handle* h, const int32_t* tok_dev, int32_t Tn, float* out_dev, int32_t normalize, float* hidden_out,
                          hipStream_t st);

namespace {

bool tower_ok(const tvc_tower_arch& a) {
    return a.width > 0 && a.layers > 0 && a.heads > 0 && a.width == a.heads * 64 && a.width % 64 == 0 &&
           a.width <= 1024 && a.mlp > 0 && a.mlp % 64 == 0 && (a.act == TVC_ACT_QUICK_GELU || a.act == TVC_ACT_GELU);
}

// One transformer tower over `rows` packed token rows (n_seq sequences of seq_len).
// Sequences: n_seq x seq_len dense rows, or (starts != nullptr) packed rows with
// `total_rows` rows in all and seq_len = the maximum length.
// pool_mode (TVC_OPT_POOLED_LAST_LAYER): 1 / 2 = the caller only reads the pooled token of every sequence (first
// token / EOT token at packed row pool_row[s]).  The LAST layer then computes attention, out-proj, ln_2 and the
// MLP for those n_seq rows only (its K and V still need every token): outputs identical, 10/12 of the layer's
// GEMM work and all but one query of its attention dropped.  Its deltas are left COMPACT ([n_seq, d], WS_POOL)
// for the caller's final LayerNorm; `pooled_out` reports that.
struct PoolBufs { uint16_t *Hc, *D1c, *H2c, *MLPc, *D2c; };
PoolBufs pool_bufs(tvc_handle* h, const tvc_tower_arch& a, int n_seq, int wso) {
    uint16_t* p = (uint16_t*)h->ws[WS_POOL + wso].p;
    const size_t nd = ((size_t)n_seq + 256) * a.width;            // rows padded to a GEMM tile
    PoolBufs b;
    b.Hc = p; b.D1c = p + nd; b.H2c = p + 2 * nd; b.D2c = p + 3 * nd; b.MLPc = p + 4 * nd;
    return b;
}
size_t pool_bytes(const tvc_tower_arch& a, int n_seq) {
    return (((size_t)n_seq + 256) * a.width * 4 + ((size_t)n_seq + 256) * a.mlp) * 2;
}

// Input-gradient mode: what the backward of a layer reads is KEPT per layer instead of recomputed (288 GB of HBM:
// 20 * width bytes per token row and layer -- 3.9 GB at ViT-L/14, 32 images): the layer's fp32 input rows, its QKV
// rows, its out-projection output (ln_2's second operand) and the FC1 pre-activation.
struct GradSave {
    float* x;          // [layers][rows, d]      fp32
    uint16_t* qkv;     // [layers][rows, 3d]     bf16
    uint16_t* d1;      // [layers][rows, d]      bf16
    uint16_t* u;       // [layers][rows, mlp]    bf16
};

int run_layers(tvc_handle* h, const tvc_tower_arch& a, const tvc_layer_weights* lw, int n_seq, int seq_len,
               int causal, const int32_t* starts, int total_rows, int wso, hipStream_t st,
               const int32_t* pfx = nullptr, int pool_mode = 0, const int32_t* pool_row = nullptr,
               int64_t pool_x_stride = 0, const GradSave* gs = nullptr) {
    const int d = a.width;
    const int rows = starts ? total_rows : n_seq * seq_len;
    float* X = (float*)h->ws[WS_X + wso].p;
    uint16_t* H = (uint16_t*)h->ws[WS_H + wso].p;
    uint16_t* QKV = (uint16_t*)h->ws[WS_QKV + wso].p;
    uint16_t* MLP = (uint16_t*)h->ws[WS_MLP + wso].p;
    uint16_t* D1 = (uint16_t*)h->ws[WS_DELTA + wso].p;     // attention out-proj output
    uint16_t* D2 = (uint16_t*)h->ws[WS_DELTA2 + wso].p;    // MLP fc2 output
    // The residual projections (attention out-proj, MLP fc2) are store-only GEMMs writing bf16
    // deltas; LayerNorm passes fold them into the fp32 residual stream X while they normalise (a
    // streaming kernel at the HBM roofline instead of a read-modify-write epilogue inside an
    // MFMA-bound kernel).  X is rewritten ONCE per layer: ln_2 normalises X + D1 without storing
    // it, the next ln_1 reads X + D1 + D2 (same fp32 sums, same order) and stores that.  The last
    // layer's two deltas are left pending for the caller's final LayerNorm (ln_post / ln_final).
    bool pending = false;
    for (int l = 0; l < a.layers; ++l) {
        const tvc_layer_weights& w = lw[l];
        {
            // D1 still points at the PREVIOUS layer's out-projection output here
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)rows * d * (pending ? 14.0 : 6.0));
            // gradient mode keeps the residual stream as layer l sees it (X + the pending deltas): ln_1 writes it to the
            // kept buffer while it normalises (`xsum_out`) -- no separate device-to-device copy of X per layer
            HIP_TRY(launch_layernorm(X, d, nullptr, pending ? D1 : nullptr, 1, w.ln1_g, w.ln1_b, H, rows, d, st,
                                     pending ? D2 : nullptr, 0, gs ? gs->x + (size_t)l * rows * d : nullptr));
        }
        if (gs) {
            QKV = gs->qkv + (size_t)l * rows * 3 * d;
            D1 = gs->d1 + (size_t)l * rows * d;
        }
        GemmLaunch g;
        g.A = w.wqkv; g.lda = d; g.I = 3 * d; g.B = H; g.ldb = d; g.J = rows; g.K = d;
        g.bias = w.bqkv; g.out = QKV; g.ldo = 3 * d; g.epilogue = TVC_EPI_BF16; g.b_rows_padded = true;
        g.splitk_small = gs != nullptr;            // gradient mode only (see tvc_encode_image_backward)
        HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
        if (pool_mode && l == a.layers - 1) {
            const PoolBufs pb = pool_bufs(h, a, n_seq, wso);
            {
                const double avg_len = starts ? (double)rows / n_seq : (double)seq_len;
                ProfScope ps(h, st, TVC_PROF_ATTENTION, 4.0 * n_seq * a.heads * avg_len * 64);
                HIP_TRY(launch_attention(QKV, pb.Hc, starts, n_seq, seq_len, a.heads, causal, st, pfx, pool_mode, pool_row));
            }
            g = GemmLaunch();
            g.A = w.wo; g.lda = d; g.I = d; g.B = pb.Hc; g.ldb = d; g.J = n_seq; g.K = d;
            g.bias = w.bo; g.out = pb.D1c; g.ldo = d; g.epilogue = TVC_EPI_BF16; g.b_rows_padded = true;
            HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
            HIP_TRY(launch_layernorm(X, pool_row ? d : pool_x_stride, pool_row, pb.D1c, 0, w.ln2_g, w.ln2_b, pb.H2c, n_seq, d,
                                     st, nullptr, 1));
            g = GemmLaunch();
            g.A = w.w1; g.lda = d; g.I = a.mlp; g.B = pb.H2c; g.ldb = d; g.J = n_seq; g.K = d;
            g.bias = w.b1; g.out = pb.MLPc; g.ldo = a.mlp; g.b_rows_padded = true;
            g.epilogue = a.act == TVC_ACT_GELU ? TVC_EPI_BF16 : TVC_EPI_GELU_BF16;
            HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
            if (a.act == TVC_ACT_GELU) HIP_TRY(launch_gelu_erf_bf16(pb.MLPc, (int64_t)n_seq * a.mlp, st));
            g = GemmLaunch();
            g.A = w.w2; g.lda = a.mlp; g.I = d; g.B = pb.MLPc; g.ldb = a.mlp; g.J = n_seq; g.K = a.mlp;
            g.bias = w.b2; g.out = pb.D2c; g.ldo = d; g.epilogue = TVC_EPI_BF16; g.b_rows_padded = true;
            HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
            return TVC_OK;
        }
        {
            const double avg_len = starts ? (double)rows / n_seq : (double)seq_len;
            const double fl = 4.0 * n_seq * a.heads * avg_len * avg_len * 64 * (causal ? 0.5 : 1.0);
            ProfScope ps(h, st, TVC_PROF_ATTENTION, fl);
            HIP_TRY(launch_attention(QKV, H, starts, n_seq, seq_len, a.heads, causal, st, pfx));
        }
        g = GemmLaunch();
        g.A = w.wo; g.lda = d; g.I = d; g.B = H; g.ldb = d; g.J = rows; g.K = d;
        g.bias = w.bo; g.out = D1; g.ldo = d; g.epilogue = TVC_EPI_BF16; g.b_rows_padded = true;
        g.splitk_small = gs != nullptr;
        HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
        {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)rows * d * 8.0);
            HIP_TRY(launch_layernorm(X, d, nullptr, D1, 0, w.ln2_g, w.ln2_b, H, rows, d, st));
        }
        g = GemmLaunch();
        g.A = w.w1; g.lda = d; g.I = a.mlp; g.B = H; g.ldb = d; g.J = rows; g.K = d;
        g.bias = w.b1; g.b_rows_padded = true; g.ldo = a.mlp;
        g.splitk_small = gs != nullptr;
        if (gs) {
            // keep the pre-activation (what gelu' needs); the activation is one streaming pass over it
            uint16_t* U = gs->u + (size_t)l * rows * a.mlp;
            g.out = U; g.epilogue = TVC_EPI_BF16;
            HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
            HIP_TRY(launch_gelu_fwd(U, MLP, (int64_t)rows * a.mlp, st));
        } else if (a.act == TVC_ACT_GELU) {
            // erf GELU (the SD-2.x text encoder): store-only FC1, the activation as a streaming pass in place
            g.out = MLP; g.epilogue = TVC_EPI_BF16;
            HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)rows * a.mlp * 4.0);
            HIP_TRY(launch_gelu_erf_bf16(MLP, (int64_t)rows * a.mlp, st));
        } else {
            g.out = MLP; g.epilogue = TVC_EPI_GELU_BF16;
            HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
        }
        g = GemmLaunch();
        g.A = w.w2; g.lda = a.mlp; g.I = d; g.B = MLP; g.ldb = a.mlp; g.J = rows; g.K = a.mlp;
        g.bias = w.b2; g.out = D2; g.ldo = d; g.epilogue = TVC_EPI_BF16; g.b_rows_padded = true;
        g.splitk_small = gs != nullptr;
        HIP_TRY(timed_gemm(h, g, st, WS_SPLITK + wso));
        pending = true;
    }
    return TVC_OK;
}

int ensure_tower_ws(tvc_handle* h, const tvc_tower_arch& a, int64_t rows, int n_seq, int wso) {
    int rc;
    // GEMM operands get readable rows up to the next multiple of 256 (+ one tile): the 64-deep ring form stages
    // whole 256-row tiles without clamping; what it reads beyond the last row only reaches outputs never stored
    rows = (rows + 255) / 256 * 256 + 256;
    if ((rc = ensure(h, (Slot)(WS_X + wso), (size_t)rows * a.width * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_H + wso), (size_t)rows * a.width * 2))) return rc;
    if ((rc = ensure(h, (Slot)(WS_QKV + wso), (size_t)rows * a.width * 3 * 2))) return rc;
    if ((rc = ensure(h, (Slot)(WS_MLP + wso), (size_t)rows * a.mlp * 2))) return rc;
    if ((rc = ensure(h, (Slot)(WS_CLS + wso), (size_t)n_seq * a.width * 2))) return rc;
    if ((rc = ensure(h, (Slot)(WS_DELTA + wso), (size_t)rows * a.width * 2))) return rc;
    if ((rc = ensure(h, (Slot)(WS_DELTA2 + wso), (size_t)rows * a.width * 2))) return rc;
    // fp32 partial tiles of the split-K paths (<= 256 partial tiles of 256 KiB): small batches split
    // every tile over K, big ones (opt-in) the left-over tile columns
    if ((rc = ensure(h, (Slot)(WS_SPLITK + wso), (size_t)256 * 256 * 256 * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_POOL + wso), pool_bytes(a, n_seq)))) return rc;
    return TVC_OK;
}

}  // namespace

extern "C" {

uint32_t tvc_abi_version(void) { return TVC_ABI_VERSION; }

const char* tvc_last_error(tvc_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tvc_create(const tvc_model_desc* desc, const tvc_vision_weights* vision, const tvc_text_weights* text,
               tvc_handle** out) {
    tvc_handle* h = nullptr;   // for HIP_TRY / fail before the handle exists
    if (!out) return fail(h, TVC_E_INVALID, "tvc_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(h, TVC_E_HIP, "tvc_create: no HIP device visible (the TVC path has no CPU fallback)");
    if ((vision || text) && !desc) return fail(h, TVC_E_INVALID, "tvc_create: weights given without a model desc");
    tvc_handle* nh = new tvc_handle();
    if (desc) nh->desc = *desc;
    if (vision) {
        const tvc_model_desc& d = *desc;
        const int Kraw = 3 * d.patch * d.patch;
        if (!tower_ok(d.vision) || d.patch <= 0 || d.image_size % d.patch != 0 || d.embed_dim % 64 != 0 ||
            Kraw <= 0 || !vision->layers) {
            delete nh;
            return fail(h, TVC_E_INVALID, "tvc_create: unsupported vision geometry (head_dim must be 64, width<=1024)");
        }
        const int T = (d.image_size / d.patch) * (d.image_size / d.patch) + 1;
        if (T > 288) { delete nh; return fail(h, TVC_E_INVALID, "tvc_create: vision sequence longer than 288 tokens"); }
        nh->vw = *vision;
        nh->vlayers.assign(vision->layers, vision->layers + d.vision.layers);
        nh->vw.layers = nh->vlayers.data();
        nh->has_vision = true;
    }
    if (text) {
        const tvc_model_desc& d = *desc;
        if (!tower_ok(d.text) || d.ctx <= 0 || d.ctx > 288 || d.vocab <= 0 || d.embed_dim % 64 != 0 || !text->layers) {
            delete nh;
            return fail(h, TVC_E_INVALID, "tvc_create: unsupported text geometry");
        }
        nh->tw = *text;
        nh->tlayers.assign(text->layers, text->layers + d.text.layers);
        nh->tw.layers = nh->tlayers.data();
        nh->has_text = true;
    }
    *out = nh;
    return TVC_OK;
}

void tvc_destroy(tvc_handle* h) {
    if (!h) return;
    tvc_sd_free(h);
    tvc_split_free(h);
    for (auto& b : h->ws) if (b.p) (void)hipFree(b.p);
    for (void* p : h->wT) if (p) (void)hipFree(p);
    for (auto& bk : h->banks) {
        if (bk.owned) (void)hipFree(bk.owned);
        if (bk.bounds) (void)hipFree(bk.bounds);
    }
    delete h;
}

uint64_t tvc_workspace_bytes(tvc_handle* h) {
    if (!h) return 0;
    uint64_t n = 0;
    for (auto& b : h->ws) n += b.n;
    return n;
}

int tvc_encode_image(tvc_handle* h, const float* pix_dev, int32_t B, float* out_dev, int32_t normalize,
                     void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!h->has_vision) return fail(h, TVC_E_STATE, "tvc_encode_image: handle has no vision tower");
    if (B < 0 || (B > 0 && (!pix_dev || !out_dev))) return fail(h, TVC_E_INVALID, "tvc_encode_image: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.vision;
    const int gside = m.image_size / m.patch, P = gside * gside, T = P + 1, d = a.width;
    const int Kp = (3 * m.patch * m.patch + 63) / 64 * 64;
    const int chunk = B < h->max_chunk_images ? B : h->max_chunk_images;
    if (B == 0) return TVC_OK;
    if (h->tower_precision == 1) return tvc_precise_encode_image(h, pix_dev, B, out_dev, normalize, st);
    if (h->tower_precision == 2) return tvc_split_encode_image(h, pix_dev, B, out_dev, normalize, st);
    int rc;
    if ((rc = ensure_tower_ws(h, a, (int64_t)chunk * T, chunk, 0))) return rc;
    if ((rc = ensure(h, WS_PATCH, ((size_t)chunk * P + 512) * Kp * 2))) return rc;      // + tile padding, as ensure_tower_ws
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int n = (B - b0 < chunk) ? B - b0 : chunk;
        const float* pix = pix_dev + (size_t)b0 * 3 * m.image_size * m.image_size;
        uint16_t* Pm = (uint16_t*)h->ws[WS_PATCH].p;
        float* patch_out = (float*)h->ws[WS_MLP].p;    // [n*P, d] fp32 fits: mlp >= 2*d
        if ((size_t)n * P * d * 4 > h->ws[WS_MLP].n) return fail(h, TVC_E_INVALID, "tvc_encode_image: mlp < 2*width unsupported");
        HIP_TRY(launch_im2col(pix, Pm, n, m.image_size, m.patch, Kp, st));
        GemmLaunch g;
        g.A = h->vw.patch_w; g.lda = Kp; g.I = d; g.B = Pm; g.ldb = Kp; g.J = n * P; g.K = Kp;
        g.out = patch_out; g.ldo = d; g.epilogue = TVC_EPI_F32; g.b_rows_padded = true;
        HIP_TRY(timed_gemm(h, g, st, WS_SPLITK));
        HIP_TRY(launch_assemble_lnpre(patch_out, h->vw.cls, h->vw.pos, h->vw.ln_pre_g, h->vw.ln_pre_b,
                                      (float*)h->ws[WS_X].p, n, T, d, st));
        const int pool = h->pooled_last ? 1 : 0;                  // only the class token (row b*T) is pooled
        if ((rc = run_layers(h, a, h->vw.layers, n, T, 0, nullptr, 0, 0, st, nullptr, pool, nullptr, (int64_t)T * d))) return rc;
        // ln_post on the class rows, projection, L2 norm
        uint16_t* Hc = (uint16_t*)h->ws[WS_CLS].p;
        // ln_post on the class rows (row b*T), folding in the last layer's two pending deltas
        if (pool) {
            const PoolBufs pb = pool_bufs(h, a, n, 0);
            HIP_TRY(launch_layernorm((float*)h->ws[WS_X].p, (int64_t)T * d, nullptr, pb.D1c, 0, h->vw.ln_post_g,
                                     h->vw.ln_post_b, Hc, n, d, st, pb.D2c, 1));
        } else {
            HIP_TRY(launch_layernorm((float*)h->ws[WS_X].p, (int64_t)T * d, nullptr,
                                     (const uint16_t*)h->ws[WS_DELTA].p, 0, h->vw.ln_post_g, h->vw.ln_post_b, Hc, n, d, st,
                                     (const uint16_t*)h->ws[WS_DELTA2].p));
        }
        g = GemmLaunch();
        g.A = h->vw.proj; g.lda = d; g.I = m.embed_dim; g.B = Hc; g.ldb = d; g.J = n; g.K = d;
        g.out = out_dev + (size_t)b0 * m.embed_dim; g.ldo = m.embed_dim; g.epilogue = TVC_EPI_F32;
        HIP_TRY(timed_gemm(h, g, st, WS_SPLITK));
        if (normalize) HIP_TRY(launch_l2norm_rows(out_dev + (size_t)b0 * m.embed_dim, n, m.embed_dim, st));
    }
    return TVC_OK;
}

int tvc_encode_text(tvc_handle* h, const int32_t* tok_dev, int32_t Tn, float* out_dev, int32_t normalize,
                    void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!h->has_text) return fail(h, TVC_E_STATE, "tvc_encode_text: handle has no text tower");
    if (Tn < 0 || (Tn > 0 && (!tok_dev || !out_dev))) return fail(h, TVC_E_INVALID, "tvc_encode_text: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.text;
    const int d = a.width, ctx = m.ctx;
    if (Tn == 0) return TVC_OK;
    if (h->tower_precision == 1) return tvc_precise_encode_text(h, tok_dev, Tn, out_dev, normalize, nullptr, st);
    if (h->tower_precision == 2) return tvc_split_encode_text(h, tok_dev, Tn, out_dev, normalize, nullptr, st);
    int chunk = Tn < h->max_chunk_texts ? Tn : h->max_chunk_texts;
    // prefix sharing needs whole groups in a pass
    const int G = (h->pack_text && h->text_group >= 2 && Tn % h->text_group == 0) ? h->text_group : 0;
    if (G && chunk >= G) chunk = chunk / G * G;
    const bool share = G && chunk % G == 0;
    int rc;
    if ((rc = ensure_tower_ws(h, a, (int64_t)chunk * ctx, chunk, WS_TOWER_N))) return rc;
    if (share && (rc = ensure(h, WS_PFX, (size_t)chunk * 2 * 4))) return rc;
    if ((rc = ensure(h, WS_EOT, (size_t)chunk * 4))) return rc;
    if ((rc = ensure(h, WS_STARTS, (size_t)(chunk + 2) * 4))) return rc;
    if ((rc = ensure(h, WS_LENS, (size_t)chunk * 4))) return rc;
    for (int t0 = 0; t0 < Tn; t0 += chunk) {
        const int n = (Tn - t0 < chunk) ? Tn - t0 : chunk;
        int32_t* eot = (int32_t*)h->ws[WS_EOT].p;
        const int32_t* tok = tok_dev + (size_t)t0 * ctx;
        const int32_t* starts = nullptr;
        const int32_t* pfx = nullptr;
        int total_rows = n * ctx, max_len = ctx;
        if (h->pack_text) {
            // Keep only the tokens up to and including EOT: under the causal mask the later
            // positions cannot reach the pooled (EOT) row, so the result is bit-identical.
            // The row count sizes the GEMM grids, hence ONE 8-byte read-back per call.
            int32_t* sd = (int32_t*)h->ws[WS_STARTS].p;
            int32_t* pd = share ? (int32_t*)h->ws[WS_PFX].p : nullptr;
            HIP_TRY(launch_text_lens_scan(tok, sd, pd, n, ctx, G, st, (int32_t*)h->ws[WS_LENS].p));
            pfx = pd;
            int32_t tail[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(tail, sd + n, sizeof tail, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            total_rows = tail[0]; max_len = tail[1];
            if (total_rows < (share ? n / G : n) || total_rows > n * ctx || max_len < 1 || max_len > ctx)
                return fail(h, TVC_E_HIP, "tvc_encode_text: inconsistent sequence lengths");
            starts = sd;
        }
        HIP_TRY(launch_text_embed(tok, h->tw.tok_emb, h->tw.pos, (float*)h->ws[WS_TX].p, eot, starts, n, ctx, d,
                                  m.vocab, st, pfx));
        const int pool = h->pooled_last ? 2 : 0;                  // only the EOT row of every text is pooled
        if ((rc = run_layers(h, a, h->tw.layers, n, max_len, 1, starts, total_rows, WS_TOWER_N, st, pfx, pool, eot, d))) return rc;
        uint16_t* Hc = (uint16_t*)h->ws[WS_TCLS].p;
        if (pool) {
            const PoolBufs pb = pool_bufs(h, a, n, WS_TOWER_N);
            HIP_TRY(launch_layernorm((float*)h->ws[WS_TX].p, d, eot, pb.D1c, 0, h->tw.ln_final_g, h->tw.ln_final_b, Hc,
                                     n, d, st, pb.D2c, 1));
        } else {
            HIP_TRY(launch_layernorm((float*)h->ws[WS_TX].p, d, eot, (const uint16_t*)h->ws[WS_TDELTA].p, 0,
                                     h->tw.ln_final_g, h->tw.ln_final_b, Hc, n, d, st,
                                     (const uint16_t*)h->ws[WS_TDELTA2].p));
        }
        GemmLaunch g;
        g.A = h->tw.proj; g.lda = d; g.I = m.embed_dim; g.B = Hc; g.ldb = d; g.J = n; g.K = d;
        g.out = out_dev + (size_t)t0 * m.embed_dim; g.ldo = m.embed_dim; g.epilogue = TVC_EPI_F32;
        HIP_TRY(timed_gemm(h, g, st, WS_TSPLITK));
        if (normalize) HIP_TRY(launch_l2norm_rows(out_dev + (size_t)t0 * m.embed_dim, n, m.embed_dim, st));
    }
    return TVC_OK;
}

int tvc_encode_text_hidden(tvc_handle* h, const int32_t* tok_dev, int32_t Tn, float* out_dev, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!h->has_text) return fail(h, TVC_E_STATE, "tvc_encode_text_hidden: handle has no text tower");
    if (Tn < 0 || (Tn > 0 && (!tok_dev || !out_dev))) return fail(h, TVC_E_INVALID, "tvc_encode_text_hidden: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.text;
    const int d = a.width, ctx = m.ctx;
    if (Tn == 0) return TVC_OK;
    if (h->tower_precision == 1) return tvc_precise_encode_text(h, tok_dev, Tn, nullptr, 0, out_dev, st);
    if (h->tower_precision == 2) return tvc_split_encode_text(h, tok_dev, Tn, nullptr, 0, out_dev, st);
    const int chunk = Tn < h->max_chunk_texts ? Tn : h->max_chunk_texts;
    int rc;
    if ((rc = ensure_tower_ws(h, a, (int64_t)chunk * ctx, chunk, WS_TOWER_N))) return rc;
    if ((rc = ensure(h, WS_EOT, (size_t)chunk * 4))) return rc;
    for (int t0 = 0; t0 < Tn; t0 += chunk) {
        const int n = (Tn - t0 < chunk) ? Tn - t0 : chunk;
        // every position is an output here (the conditioning sequence of a latent-diffusion UNet): dense rows, no
        // EOT packing, no pooling; the causal mask is the tower's own
        HIP_TRY(launch_text_embed(tok_dev + (size_t)t0 * ctx, h->tw.tok_emb, h->tw.pos, (float*)h->ws[WS_TX].p,
                                  (int32_t*)h->ws[WS_EOT].p, nullptr, n, ctx, d, m.vocab, st, nullptr));
        if ((rc = run_layers(h, a, h->tw.layers, n, ctx, 1, nullptr, 0, WS_TOWER_N, st))) return rc;
        HIP_TRY(launch_layernorm((float*)h->ws[WS_TX].p, d, nullptr, (const uint16_t*)h->ws[WS_TDELTA].p, 0,
                                 h->tw.ln_final_g, h->tw.ln_final_b, nullptr, n * ctx, d, st,
                                 (const uint16_t*)h->ws[WS_TDELTA2].p, 0, nullptr, out_dev + (size_t)t0 * ctx * d));
    }
    return TVC_OK;
}

int tvc_bank_set(tvc_handle* h, const void* bank_dev, int64_t R, int32_t D, int32_t dtype, void* stream) {
    if (!h) return TVC_E_INVALID;
    BankSlot& bk = h->banks[h->cur_bank];
    if (R < 0 || R > 0x7fffffffLL || D <= 0 || D % 64 != 0 || (R > 0 && !bank_dev))
        return fail(h, TVC_E_INVALID, "tvc_bank_set: need 0 <= R < 2^31 and D % 64 == 0");
    if (bk.owned) { HIP_TRY(hipFree(bk.owned)); bk.owned = nullptr; }
    bk.bank = nullptr; bk.R = 0; bk.D = D;
    if (dtype == TVC_DTYPE_BF16) {
        bk.bank = (const uint16_t*)bank_dev; bk.planes = 1;
    } else if (dtype == TVC_DTYPE_F32) {
        if (R > 0) {
            HIP_TRY(hipMalloc(&bk.owned, (size_t)R * 2 * D * 2));
            HIP_TRY(launch_split_planes((const float*)bank_dev, (uint16_t*)bk.owned, R, D, 2, (hipStream_t)stream));
        }
        bk.bank = (const uint16_t*)bk.owned; bk.planes = 2;
    } else {
        return fail(h, TVC_E_INVALID, "tvc_bank_set: dtype must be TVC_DTYPE_BF16 or TVC_DTYPE_F32");
    }
    if (!bk.bounds) HIP_TRY(hipMalloc((void**)&bk.bounds, 8));
    HIP_TRY(launch_bank_bounds(bk.bank, (int64_t)bk.planes * D, bk.planes, D, R, bk.bounds,
                               (hipStream_t)stream));
    bk.R = R;
    return TVC_OK;
}

int tvc_bank_select(tvc_handle* h, int32_t slot) {
    if (!h) return TVC_E_INVALID;
    if (slot < 0 || slot >= TVC_MAX_BANKS) return fail(h, TVC_E_INVALID, "tvc_bank_select: slot must be in [0, TVC_MAX_BANKS)");
    h->cur_bank = slot;
    return TVC_OK;
}

int tvc_bank_search(tvc_handle* h, const float* rows_dev, int32_t M, int32_t k, float count_thr,
                    int64_t idx_offset, int32_t* topk_idx_dev, float* topk_sim_dev, float* moments_dev,
                    void* stream) {
    if (!h) return TVC_E_INVALID;
    BankSlot& bk = h->banks[h->cur_bank];
    if (!bk.bank && bk.R != 0) return fail(h, TVC_E_STATE, "tvc_bank_search: no bank registered");
    if (bk.D == 0) return fail(h, TVC_E_STATE, "tvc_bank_search: call tvc_bank_set first");
    if (M < 0 || k < 1 || k > TVC_MAX_TOPK || (M > 0 && (!rows_dev || !topk_idx_dev || !topk_sim_dev)))
        return fail(h, TVC_E_INVALID, "tvc_bank_search: need 1 <= k <= 128 and non-NULL buffers");
    if (M == 0) return TVC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int D = bk.D;
    if (bk.R == 0) {
        // empty bank: retrieval_ref.py:195-197 returns no references
        HIP_TRY(hipMemsetAsync(topk_idx_dev, 0xff, (size_t)M * k * 4, st));
        HIP_TRY(hipMemsetAsync(topk_sim_dev, 0, (size_t)M * k * 4, st));
        if (moments_dev) HIP_TRY(hipMemsetAsync(moments_dev, 0, (size_t)M * 16, st));
        return TVC_OK;
    }
    BankSearchLaunch L;
    bank_plan(bk.R, M, k, &L.n_sample, &L.sample_stride, &L.S, &L.cap);
    int rc;
    if ((rc = ensure(h, WS_QPLANES, (size_t)((M + 255) / 256 * 256) * 2 * D * 2))) return rc;
    if ((rc = ensure(h, WS_S0, (size_t)M * (L.n_sample > 256 ? L.n_sample : 256) * 4))) return rc;
    if ((rc = ensure(h, WS_TAU, (size_t)M * 4))) return rc;
    if ((rc = ensure(h, WS_CAND, (size_t)L.S * M * L.cap * 8))) return rc;
    if ((rc = ensure(h, WS_CAND_CNT, (size_t)L.S * M * 4))) return rc;
    if ((rc = ensure(h, WS_MOM_PART, (size_t)L.S * M * 16))) return rc;
    if ((rc = ensure(h, WS_OVERFLOW, 16))) return rc;
    HIP_TRY(launch_split_planes(rows_dev, (uint16_t*)h->ws[WS_QPLANES].p, M, D, 2, st));
    L.bank = bk.bank; L.ldb = (int64_t)bk.planes * D; L.R = bk.R; L.D = D; L.bank_planes = bk.planes;
    L.qplanes = (const uint16_t*)h->ws[WS_QPLANES].p; L.q_rows_padded = true; L.M = M; L.k = k; L.count_thr = count_thr;
    L.idx_offset = idx_offset;
    L.rows = rows_dev; L.bank_bounds = bk.bounds; L.allow_filter = h->bank_filter;
    L.s0 = (float*)h->ws[WS_S0].p; L.gmax = L.s0; L.tau = (float*)h->ws[WS_TAU].p; L.cand = h->ws[WS_CAND].p;
    L.cand_cnt = (int32_t*)h->ws[WS_CAND_CNT].p; L.mom_part = (float*)h->ws[WS_MOM_PART].p;
    L.overflow = (int32_t*)h->ws[WS_OVERFLOW].p;
    L.topk_idx = topk_idx_dev; L.topk_sim = topk_sim_dev; L.moments = moments_dev;
    {
        const bool filter = !moments_dev && h->bank_filter && D <= 2048;
        const int planes = filter ? 1 : (bk.planes == 2 ? 3 : 2);
        ProfScope ps(h, st, TVC_PROF_BANK, 2.0 * (double)bk.R * M * D * planes);
        HIP_TRY(launch_bank_search(L, st));
    }
    return TVC_OK;
}

int tvc_bank_search_dense(tvc_handle* h, const float* rows_dev, int32_t M, int32_t k, float count_thr,
                          int64_t idx_offset, int32_t* topk_idx_dev, float* topk_sim_dev, float* moments_dev,
                          void* stream) {
    if (!h) return TVC_E_INVALID;
    BankSlot& bk = h->banks[h->cur_bank];
    if (bk.D == 0 || (!bk.bank && bk.R != 0)) return fail(h, TVC_E_STATE, "tvc_bank_search_dense: call tvc_bank_set first");
    if (M < 0 || k < 1 || k > TVC_MAX_TOPK || (M > 0 && (!rows_dev || !topk_idx_dev || !topk_sim_dev)))
        return fail(h, TVC_E_INVALID, "tvc_bank_search_dense: need 1 <= k <= 128 and non-NULL buffers");
    if (M == 0) return TVC_OK;
    if (bk.R == 0) return tvc_bank_search(h, rows_dev, M, k, count_thr, idx_offset, topk_idx_dev, topk_sim_dev, moments_dev, stream);
    hipStream_t st = (hipStream_t)stream;
    const int D = bk.D;
    int block = (int)((size_t)1 << 28) / (int)(bk.R > 0 ? bk.R : 1);      // <= 1 GiB of similarities at a time
    if (block > 64) block = 64;
    if (block < 1) block = 1;
    int rc;
    if ((rc = ensure(h, WS_QPLANES, (size_t)((M + 255) / 256 * 256) * 2 * D * 2))) return rc;
    if ((rc = ensure(h, WS_S0, (size_t)block * bk.R * 4))) return rc;
    HIP_TRY(launch_split_planes(rows_dev, (uint16_t*)h->ws[WS_QPLANES].p, M, D, 2, st));
    BankSearchLaunch L;
    L.bank = bk.bank; L.ldb = (int64_t)bk.planes * D; L.R = bk.R; L.D = D; L.bank_planes = bk.planes;
    L.qplanes = (const uint16_t*)h->ws[WS_QPLANES].p; L.q_rows_padded = true; L.M = M; L.k = k; L.count_thr = count_thr;
    L.idx_offset = idx_offset;
    L.topk_idx = topk_idx_dev; L.topk_sim = topk_sim_dev; L.moments = moments_dev;
    HIP_TRY(launch_bank_search_dense(L, (float*)h->ws[WS_S0].p, block, st));
    return TVC_OK;
}

int tvc_bank_status(tvc_handle* h, void* stream) {
    if (!h) return TVC_E_INVALID;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (!h->ws[WS_OVERFLOW].p) return TVC_OK;
    int32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, h->ws[WS_OVERFLOW].p, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) {
        char m[160];
        snprintf(m, sizeof m, "tvc_bank_search: candidate lists overflowed (flag %d): bank has too many "
                              "near-identical rows for the sampled bound", flag);
        return fail(h, TVC_E_OVERFLOW, m);
    }
    return TVC_OK;
}

int tvc_bank_gather(tvc_handle* h, const int32_t* idx_dev, int32_t n, int64_t idx_offset, float* out_dev,
                    void* stream) {
    if (!h) return TVC_E_INVALID;
    BankSlot& bk = h->banks[h->cur_bank];
    if (bk.D == 0) return fail(h, TVC_E_STATE, "tvc_bank_gather: call tvc_bank_set first");
    if (n < 0 || (n > 0 && (!idx_dev || !out_dev))) return fail(h, TVC_E_INVALID, "tvc_bank_gather: bad arguments");
    if (n == 0) return TVC_OK;
    if (bk.R == 0) { HIP_TRY(hipMemsetAsync(out_dev, 0, (size_t)n * bk.D * 4, (hipStream_t)stream)); return TVC_OK; }
    HIP_TRY(launch_gather_rows(bk.bank, (int64_t)bk.planes * bk.D, bk.planes, bk.D, bk.R, idx_dev,
                               idx_offset, n, out_dev, (hipStream_t)stream));
    return TVC_OK;
}

int tvc_topk_merge(tvc_handle* h, const int32_t* idx_parts_dev, const float* sim_parts_dev,
                   const float* feat_parts_dev, const float* mom_parts_dev, int32_t W, int32_t M, int32_t k,
                   int32_t kf, int32_t D, int32_t* idx_out_dev, float* sim_out_dev, float* feat_out_dev,
                   float* mom_out_dev, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (M < 0 || !idx_parts_dev || !sim_parts_dev || !idx_out_dev || !sim_out_dev)
        return fail(h, TVC_E_INVALID, "tvc_topk_merge: bad arguments");
    if (W < 1 || k < 1 || k > TVC_MAX_TOPK || W * k > 256 || kf < 0 || kf > k || kf > 32 ||
        (kf > 0 && feat_parts_dev && (!feat_out_dev || D < 1)))
        return fail(h, TVC_E_INVALID, "tvc_topk_merge: limits are W * k <= 256, kf <= min(k, 32)");
    hipError_t st = launch_topk_merge(idx_parts_dev, sim_parts_dev, feat_parts_dev, mom_parts_dev, W, M, k, kf, D,
                                      idx_out_dev, sim_out_dev, feat_out_dev, mom_out_dev, (hipStream_t)stream);
    if (st != hipSuccess) return fail(h, st == hipErrorInvalidValue ? TVC_E_INVALID : TVC_E_HIP,
                                      std::string("tvc_topk_merge: ") + hipGetErrorString(st));
    return TVC_OK;
}

int tvc_cosine_matrix(tvc_handle* h, const float* x_dev, int32_t N, const float* y_dev, int32_t M, int32_t D,
                      float* out_dev, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (N < 0 || M < 0 || D <= 0 || D % 64 != 0 || ((N > 0 && M > 0) && (!x_dev || !y_dev || !out_dev)))
        return fail(h, TVC_E_INVALID, "tvc_cosine_matrix: need D % 64 == 0 and non-NULL buffers");
    if (N == 0 || M == 0) return TVC_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = ensure(h, WS_COSX, (size_t)N * D * 4))) return rc;
    if ((rc = ensure(h, WS_COSY, (size_t)M * D * 4))) return rc;
    if ((rc = ensure(h, WS_COSXP, (size_t)N * D * 4))) return rc;
    if ((rc = ensure(h, WS_COSYP, (size_t)M * D * 4))) return rc;
    float* xn = (float*)h->ws[WS_COSX].p;
    float* yn = (float*)h->ws[WS_COSY].p;
    HIP_TRY(hipMemcpyAsync(xn, x_dev, (size_t)N * D * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(yn, y_dev, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(launch_l2norm_rows(xn, N, D, st));
    HIP_TRY(launch_l2norm_rows(yn, M, D, st));
    uint16_t* xp = (uint16_t*)h->ws[WS_COSXP].p;
    uint16_t* yp = (uint16_t*)h->ws[WS_COSYP].p;
    HIP_TRY(launch_split_planes(xn, xp, N, D, 2, st));
    HIP_TRY(launch_split_planes(yn, yp, M, D, 2, st));
    // out[n, m]: "A rows" (fast output dim) = y, "B rows" = x; hi.hi + hi.lo + lo.hi
    GemmLaunch g;
    g.A = yp; g.lda = 2 * (int64_t)D; g.I = M; g.B = xp; g.ldb = 2 * (int64_t)D; g.J = N; g.K = D; g.planes = 3;
    g.a_plane_off[0] = 0; g.a_plane_off[1] = 0; g.a_plane_off[2] = D;
    g.b_plane_off[0] = 0; g.b_plane_off[1] = D; g.b_plane_off[2] = 0;
    g.out = out_dev; g.ldo = M; g.epilogue = TVC_EPI_F32;
    HIP_TRY(timed_gemm(h, g, st));
    return TVC_OK;
}

int tvc_consistency(tvc_handle* h, const float* img_dev, const float* txt_dev, int32_t B, int32_t N, int32_t D,
                    const int32_t* ref_idx_dev, const float* ref_sim_dev, const float* ref_feat_dev, int32_t ks,
                    int32_t kf, const tvc_consistency_params* params, float* rec_dev, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (B < 0 || N < 0 || D <= 0 || !params || (B > 0 && (!img_dev || !txt_dev || !rec_dev)))
        return fail(h, TVC_E_INVALID, "tvc_consistency: bad arguments");
    if (ks > 0 && (!ref_idx_dev || !ref_sim_dev || !ref_feat_dev || kf < 1))
        return fail(h, TVC_E_INVALID, "tvc_consistency: ks > 0 needs ref_idx, ref_sim, ref_feat and kf >= 1");
    ConsistencyParams p;
    p.reference_count = params->reference_count;
    p.similarity_threshold = params->similarity_threshold;
    p.retrieval_top_k = params->retrieval_top_k;
    p.dup_threshold = params->dup_threshold;
    p.w_text_variants = params->w_text_variants;
    p.w_consistency = params->w_consistency;
    for (int i = 0; i < 4; ++i) p.w_exp[i] = params->w_exp[i];
    if (p.retrieval_top_k > TVC_REC_MAXREF || p.retrieval_top_k < 0 || p.reference_count < 0)
        return fail(h, TVC_E_INVALID, "tvc_consistency: retrieval_top_k must be in [0, 16]");
    hipError_t st = launch_consistency(img_dev, txt_dev, B, N, D, ref_idx_dev, ref_sim_dev, ref_feat_dev, ks, kf, p,
                                       rec_dev, tvc_rec_stride(N), (hipStream_t)stream);
    if (st != hipSuccess) return fail(h, st == hipErrorInvalidValue ? TVC_E_INVALID : TVC_E_HIP,
                                      std::string("tvc_consistency: ") + hipGetErrorString(st));
    return TVC_OK;
}

// ---- input gradient of the vision tower -------------------------------------------
namespace {
int build_transposed_weights(tvc_handle* h, hipStream_t st) {
    if (!h->wT.empty()) return TVC_OK;
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.vision;
    const int d = a.width, Kp = (3 * m.patch * m.patch + 63) / 64 * 64;
    auto tr = [&](const uint16_t* w, int R, int C, void** out) -> int {
        HIP_TRY(hipMalloc(out, (size_t)R * C * 2));
        HIP_TRY(launch_transpose_bf16(w, (uint16_t*)*out, R, C, st));
        return TVC_OK;
    };
    std::vector<void*> t((size_t)a.layers * 4 + 2, nullptr);
    int rc = TVC_OK;
    for (int l = 0; l < a.layers && !rc; ++l) {
        const tvc_layer_weights& w = h->vw.layers[l];
        if ((rc = tr(w.wqkv, 3 * d, d, &t[l * 4 + 0]))) break;      // [3d, d] -> [d, 3d]
        if ((rc = tr(w.wo, d, d, &t[l * 4 + 1]))) break;
        if ((rc = tr(w.w1, a.mlp, d, &t[l * 4 + 2]))) break;        // [mlp, d] -> [d, mlp]
        if ((rc = tr(w.w2, d, a.mlp, &t[l * 4 + 3]))) break;        // [d, mlp] -> [mlp, d]
    }
    if (!rc) rc = tr(h->vw.proj, m.embed_dim, d, &t[(size_t)a.layers * 4]);           // [D, d] -> [d, D]
    if (!rc) rc = tr(h->vw.patch_w, d, Kp, &t[(size_t)a.layers * 4 + 1]);             // [d, Kp] -> [Kp, d]
    if (rc) { for (void* p : t) if (p) (void)hipFree(p); return rc; }
    h->wT = t;
    return TVC_OK;
}
}  // namespace

int tvc_encode_image_grad(tvc_handle* h, const float* pix_dev, int32_t B, float* out_dev, int32_t normalize, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!h->has_vision) return fail(h, TVC_E_STATE, "tvc_encode_image_grad: handle has no vision tower");
    if (B < 1 || !pix_dev || !out_dev) return fail(h, TVC_E_INVALID, "tvc_encode_image_grad: bad arguments");
    if (B > h->max_chunk_images) return fail(h, TVC_E_INVALID, "tvc_encode_image_grad: B exceeds TVC_OPT_MAX_CHUNK_IMAGES (one pass only)");
    if (h->desc.vision.act != TVC_ACT_QUICK_GELU) return fail(h, TVC_E_INVALID, "tvc_encode_image_grad: only QuickGELU towers have a backward pass");
    hipStream_t st = (hipStream_t)stream;
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.vision;
    const int gside = m.image_size / m.patch, P = gside * gside, T = P + 1, d = a.width;
    const int Kp = (3 * m.patch * m.patch + 63) / 64 * 64;
    const int64_t rows = (int64_t)B * T;
    int rc;
    if ((rc = ensure_tower_ws(h, a, rows, B, 0))) return rc;
    if ((rc = ensure(h, WS_PATCH, ((size_t)B * P + 512) * Kp * 2))) return rc;
    const size_t per_layer = (size_t)rows * ((size_t)d * 4 + (size_t)3 * d * 2 + (size_t)d * 2 + (size_t)a.mlp * 2);
    if ((rc = ensure(h, WS_GSAVE, (size_t)a.layers * per_layer + 256))) return rc;
    GradSave gs;
    gs.x = (float*)h->ws[WS_GSAVE].p;
    gs.qkv = (uint16_t*)(gs.x + (size_t)a.layers * rows * d);
    gs.d1 = gs.qkv + (size_t)a.layers * rows * 3 * d;
    gs.u = gs.d1 + (size_t)a.layers * rows * d;
    if ((rc = ensure(h, WS_GOUT, (size_t)B * m.embed_dim * 4))) return rc;
    if ((rc = ensure(h, WS_GXL, (size_t)B * d * 4))) return rc;
    uint16_t* Pm = (uint16_t*)h->ws[WS_PATCH].p;
    float* patch_out = (float*)h->ws[WS_MLP].p;
    if ((size_t)B * P * d * 4 > h->ws[WS_MLP].n) return fail(h, TVC_E_INVALID, "tvc_encode_image_grad: mlp < 2*width unsupported");
    HIP_TRY(launch_im2col(pix_dev, Pm, B, m.image_size, m.patch, Kp, st));
    GemmLaunch g;
    g.A = h->vw.patch_w; g.lda = Kp; g.I = d; g.B = Pm; g.ldb = Kp; g.J = B * P; g.K = Kp;
    g.out = patch_out; g.ldo = d; g.epilogue = TVC_EPI_F32; g.b_rows_padded = true;
    HIP_TRY(timed_gemm(h, g, st, WS_SPLITK));
    HIP_TRY(launch_assemble_lnpre(patch_out, h->vw.cls, h->vw.pos, h->vw.ln_pre_g, h->vw.ln_pre_b, (float*)h->ws[WS_X].p, B, T, d, st));
    if ((rc = run_layers(h, a, h->vw.layers, B, T, 0, nullptr, 0, 0, st, nullptr, 0, nullptr, 0, &gs))) return rc;
    uint16_t* Hc = (uint16_t*)h->ws[WS_CLS].p;
    HIP_TRY(launch_layernorm((float*)h->ws[WS_X].p, (int64_t)T * d, nullptr, gs.d1 + (size_t)(a.layers - 1) * rows * d, 0,
                             h->vw.ln_post_g, h->vw.ln_post_b, Hc, B, d, st, (const uint16_t*)h->ws[WS_DELTA2].p, 0,
                             (float*)h->ws[WS_GXL].p));
    g = GemmLaunch();
    g.A = h->vw.proj; g.lda = d; g.I = m.embed_dim; g.B = Hc; g.ldb = d; g.J = B; g.K = d;
    g.out = h->ws[WS_GOUT].p; g.ldo = m.embed_dim; g.epilogue = TVC_EPI_F32;
    HIP_TRY(timed_gemm(h, g, st, WS_SPLITK));
    HIP_TRY(hipMemcpyAsync(out_dev, h->ws[WS_GOUT].p, (size_t)B * m.embed_dim * 4, hipMemcpyDeviceToDevice, st));
    if (normalize) HIP_TRY(launch_l2norm_rows(out_dev, B, m.embed_dim, st));
    h->grad_B = B; h->grad_normalize = normalize; h->grad_pix = pix_dev;
    return TVC_OK;
}

int tvc_encode_image_backward(tvc_handle* h, const float* grad_out_dev, float* grad_pix_dev, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!h->has_vision || h->grad_B < 1) return fail(h, TVC_E_STATE, "tvc_encode_image_backward: call tvc_encode_image_grad first");
    if (!grad_out_dev || !grad_pix_dev) return fail(h, TVC_E_INVALID, "tvc_encode_image_backward: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.vision;
    const int B = h->grad_B, gside = m.image_size / m.patch, P = gside * gside, T = P + 1, d = a.width, D = m.embed_dim;
    const int Kp = (3 * m.patch * m.patch + 63) / 64 * 64;
    const int64_t rows = (int64_t)B * T;
    const int64_t rows_pad = (rows + 255) / 256 * 256 + 256;
    int rc;
    if ((rc = build_transposed_weights(h, st))) return rc;
    if ((rc = ensure(h, WS_GDX, (size_t)rows_pad * d * 4))) return rc;
    if ((rc = ensure(h, WS_G16, (size_t)rows_pad * d * 2))) return rc;
    if ((rc = ensure(h, WS_GMLP2, (size_t)rows_pad * a.mlp * 2))) return rc;
    if ((rc = ensure(h, WS_GDQKV, (size_t)rows_pad * 3 * d * 2))) return rc;
    if ((rc = ensure(h, WS_GSTATS, (size_t)rows * a.heads * 16))) return rc;
    if ((rc = ensure(h, WS_GSMALL, ((size_t)B + 256) * (D + d) * 2))) return rc;
    if ((rc = ensure(h, WS_GPATCH, ((size_t)B * P + 512) * (size_t)(Kp > d ? Kp : d) * 4))) return rc;
    uint16_t* H = (uint16_t*)h->ws[WS_H].p;
    uint16_t* D2 = (uint16_t*)h->ws[WS_DELTA2].p;
    float* dX = (float*)h->ws[WS_GDX].p;
    uint16_t* G16 = (uint16_t*)h->ws[WS_G16].p;
    uint16_t* dM = (uint16_t*)h->ws[WS_GMLP2].p;
    uint16_t* dQKV = (uint16_t*)h->ws[WS_GDQKV].p;
    auto gemm = [&](const void* A, int64_t lda, int I, const uint16_t* Bm, int64_t ldb, int J, int K, void* out, int64_t ldo,
                    int epi) -> int {
        GemmLaunch g;
        g.A = (const uint16_t*)A; g.lda = lda; g.I = I; g.B = Bm; g.ldb = ldb; g.J = J; g.K = K;
        g.out = out; g.ldo = ldo; g.epilogue = epi; g.b_rows_padded = true;
        // the gradient path promises no batch-position invariance of its last bits: a partial last round of tiles (FC1-shaped
        // GEMMs at 32 images: 528 tiles = 2 rounds + 16 tiles) is split over K instead of taking a third round
        g.splitk_small = true;
        HIP_TRY(timed_gemm(h, g, st, WS_SPLITK));
        return TVC_OK;
    };
    // ---- head: L2 normalise, projection, ln_post (gradient lives on the class rows only)
    uint16_t* dP16 = (uint16_t*)h->ws[WS_GSMALL].p;                 // [B(+pad), D]
    uint16_t* dHc = dP16 + ((size_t)B + 256) * D;                   // [B(+pad), d]
    HIP_TRY(launch_l2norm_bwd((const float*)h->ws[WS_GOUT].p, grad_out_dev, dP16, B, D, h->grad_normalize, st));
    if ((rc = gemm(h->wT[(size_t)a.layers * 4], D, d, dP16, D, B, D, dHc, d, TVC_EPI_BF16))) return rc;
    HIP_TRY(hipMemsetAsync(dX, 0, (size_t)rows * d * 4, st));
    HIP_TRY(hipMemsetAsync(G16, 0, (size_t)rows * d * 2, st));
    HIP_TRY(launch_layernorm_bwd((const float*)h->ws[WS_GXL].p, d, nullptr, dHc, 0, h->vw.ln_post_g, nullptr, dX, G16, B, d,
                                 (int64_t)T * d, st));
    // ---- layers, last to first, on what the forward kept (GradSave)
    const float* sx = (const float*)h->ws[WS_GSAVE].p;
    const uint16_t* sqkv = (const uint16_t*)(sx + (size_t)a.layers * rows * d);
    const uint16_t* sd1 = sqkv + (size_t)a.layers * rows * 3 * d;
    const uint16_t* su = sd1 + (size_t)a.layers * rows * d;
    for (int l = a.layers - 1; l >= 0; --l) {
        const tvc_layer_weights& w = h->vw.layers[l];
        void* const* wt = &h->wT[(size_t)l * 4];
        const float* X = sx + (size_t)l * rows * d;
        const uint16_t* QKV = sqkv + (size_t)l * rows * 3 * d;
        const uint16_t* D1 = sd1 + (size_t)l * rows * d;
        const uint16_t* U = su + (size_t)l * rows * a.mlp;
        // MLP branch: dM = dOut W2, dU = dM gelu'(U), dH2 = dU W1, ln_2 backward (+ residual)
        if ((rc = gemm(wt[3], d, a.mlp, G16, d, (int)rows, d, dM, a.mlp, TVC_EPI_BF16))) return rc;
        HIP_TRY(launch_gelu_bwd(dM, U, rows * a.mlp, st));
        if ((rc = gemm(wt[2], a.mlp, d, dM, a.mlp, (int)rows, a.mlp, D2, d, TVC_EPI_BF16))) return rc;
        HIP_TRY(launch_layernorm_bwd(X, d, D1, D2, 0, w.ln2_g, dX, dX, G16, (int)rows, d, d, st));
        // attention branch: dAO = dMid Wo, attention backward, dH1 = dQKV Wqkv, ln_1 backward (+ residual)
        if ((rc = gemm(wt[1], d, d, G16, d, (int)rows, d, H, d, TVC_EPI_BF16))) return rc;
        HIP_TRY(launch_attention_bwd(QKV, H, dQKV, (float*)h->ws[WS_GSTATS].p, B, T, a.heads, st));
        if ((rc = gemm(wt[0], 3 * d, d, dQKV, 3 * d, (int)rows, 3 * d, D2, d, TVC_EPI_BF16))) return rc;
        HIP_TRY(launch_layernorm_bwd(X, d, nullptr, D2, 0, w.ln1_g, dX, dX, G16, (int)rows, d, d, st));
    }
    // ---- stem: ln_pre, patch embedding, col2im
    uint16_t* Pm = (uint16_t*)h->ws[WS_PATCH].p;
    float* patch_out = (float*)h->ws[WS_GPATCH].p;                  // fp32 [B*P, d], then reused as dcols [B*P, Kp]
    HIP_TRY(launch_im2col(h->grad_pix, Pm, B, m.image_size, m.patch, Kp, st));
    if ((rc = gemm(h->vw.patch_w, Kp, d, Pm, Kp, B * P, Kp, patch_out, d, TVC_EPI_F32))) return rc;
    uint16_t* dpatch = (uint16_t*)h->ws[WS_GDQKV].p;               // bf16 [B*P, d]
    HIP_TRY(launch_lnpre_bwd(patch_out, h->vw.pos, h->vw.ln_pre_g, dX, dpatch, B, T, d, st));
    if ((rc = gemm(h->wT[(size_t)a.layers * 4 + 1], d, Kp, dpatch, d, B * P, d, patch_out, Kp, TVC_EPI_F32))) return rc;
    HIP_TRY(launch_col2im(patch_out, grad_pix_dev, B, m.image_size, m.patch, Kp, st));
    return TVC_OK;
}

int tvc_pgd_step(tvc_handle* h, float* adv_dev, const float* clean_dev, const float* grad_dev, float* momentum_dev, int32_t B,
                 int64_t n, float eps, float alpha, float mu, float clip_min, float clip_max, int32_t targeted, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (B < 0 || n < 0 || (B > 0 && n > 0 && (!adv_dev || !clean_dev || !grad_dev)))
        return fail(h, TVC_E_INVALID, "tvc_pgd_step: bad arguments");
    HIP_TRY(launch_pgd_step(adv_dev, clean_dev, grad_dev, momentum_dev, B, n, eps, alpha, mu, clip_min, clip_max, targeted,
                            (hipStream_t)stream));
    return TVC_OK;
}

int tvc_l2_step(tvc_handle* h, float* adv_dev, const float* clean_dev, const float* grad_dev, int32_t B, int64_t n, float eps,
                float step, float clip_min, float clip_max, int32_t descent, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (B < 0 || n < 0 || (B > 0 && n > 0 && (!adv_dev || !clean_dev || !grad_dev)))
        return fail(h, TVC_E_INVALID, "tvc_l2_step: bad arguments");
    HIP_TRY(launch_l2_step(adv_dev, clean_dev, grad_dev, B, n, eps, step, clip_min, clip_max, descent, (hipStream_t)stream));
    return TVC_OK;
}

int tvc_attention_backward(tvc_handle* h, const uint16_t* qkv_dev, const uint16_t* dout_dev, uint16_t* dqkv_dev, int32_t n_seq,
                           int32_t seq_len, int32_t heads, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!qkv_dev || !dout_dev || !dqkv_dev) return fail(h, TVC_E_INVALID, "tvc_attention_backward: NULL buffer");
    int rc;
    if ((rc = ensure(h, WS_GSTATS, (size_t)n_seq * seq_len * heads * 16))) return rc;
    HIP_TRY(launch_attention_bwd(qkv_dev, dout_dev, dqkv_dev, (float*)h->ws[WS_GSTATS].p, n_seq, seq_len, heads, (hipStream_t)stream));
    return TVC_OK;
}

int tvc_layernorm_backward(tvc_handle* h, const float* x_dev, const uint16_t* dy_dev, const float* g_dev, const float* dres_dev,
                           float* dx_dev, int32_t rows, int32_t d, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!x_dev || !dy_dev || !g_dev || !dx_dev) return fail(h, TVC_E_INVALID, "tvc_layernorm_backward: NULL buffer");
    HIP_TRY(launch_layernorm_bwd(x_dev, d, nullptr, dy_dev, 0, g_dev, dres_dev, dx_dev, nullptr, rows, d, d, (hipStream_t)stream));
    return TVC_OK;
}

int tvc_set_option(tvc_handle* h, int32_t option, int64_t value) {
    if (!h) return TVC_E_INVALID;
    switch (option) {
        case TVC_OPT_TEXT_PACKING: h->pack_text = value != 0; return TVC_OK;
        case TVC_OPT_BANK_FILTER: h->bank_filter = value != 0; return TVC_OK;
        case TVC_OPT_POOLED_LAST_LAYER: h->pooled_last = value != 0; return TVC_OK;
        case TVC_OPT_TOWER_PRECISION: {
            if (value < 0 || value > 2)
                return fail(h, TVC_E_INVALID, "tvc_set_option: TVC_OPT_TOWER_PRECISION must be 0 (bf16), 1 (fp32) or 2 (split-bf16)");
            if (value != 0 && !h->has_vision32 && !h->has_text32)
                return fail(h, TVC_E_STATE, "tvc_set_option: TVC_OPT_TOWER_PRECISION = 1 / 2 needs tvc_set_weights_f32 first");
            if (value == 2) {
                const int rc = tvc_split_prepare(h);        // builds the weight planes once (synchronises the device)
                if (rc) return rc;
            }
            h->tower_precision = (int)value; return TVC_OK;
        }
        case TVC_OPT_TEXT_GROUP:
            if (value < 0 || value > 4096) return fail(h, TVC_E_INVALID, "tvc_set_option: TVC_OPT_TEXT_GROUP out of range");
            h->text_group = (int)value; return TVC_OK;
        case TVC_OPT_MAX_CHUNK_IMAGES:
            if (value < 1) return fail(h, TVC_E_INVALID, "tvc_set_option: chunk must be >= 1");
            h->max_chunk_images = (int)value; return TVC_OK;
        case TVC_OPT_MAX_CHUNK_TEXTS:
            if (value < 1) return fail(h, TVC_E_INVALID, "tvc_set_option: chunk must be >= 1");
            h->max_chunk_texts = (int)value; return TVC_OK;
        case TVC_OPT_SD_ARENA_BYTES:
            if (value < ((int64_t)1 << 28)) return fail(h, TVC_E_INVALID, "tvc_set_option: TVC_OPT_SD_ARENA_BYTES must be >= 256 MiB");
            h->sd_arena_bytes = (size_t)value; return TVC_OK;
        case TVC_OPT_SD_STREAMS:
            if (value < 1 || value > 2) return fail(h, TVC_E_INVALID, "tvc_set_option: TVC_OPT_SD_STREAMS must be 1 or 2");
            h->sd_streams = (int)value; return TVC_OK;
        default: return fail(h, TVC_E_INVALID, "tvc_set_option: unknown option");
    }
}

int tvc_profile_begin(tvc_handle* h) {
    if (!h) return TVC_E_INVALID;
    for (auto& r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    h->prof_recs.clear();
    h->prof = true;
    return TVC_OK;
}

int tvc_profile_end(tvc_handle* h, double* ms, double* work, int64_t* launches, double* big_gemm) {
    if (!h) return TVC_E_INVALID;
    if (!ms || !work || !launches) return fail(h, TVC_E_INVALID, "tvc_profile_end: NULL output");
    h->prof = false;
    HIP_TRY(hipDeviceSynchronize());
    for (int c = 0; c < TVC_PROF_NCAT; ++c) { ms[c] = 0; work[c] = 0; launches[c] = 0; }
    if (big_gemm) big_gemm[0] = big_gemm[1] = big_gemm[2] = 0;
    // diagnostics (scripts/gemm_shape_table.py): TVC_PROF_DUMP=<file> appends one line per GEMM launch -- I J K planes split ms
    FILE* dump = nullptr;
    if (const char* path = getenv("TVC_PROF_DUMP")) dump = fopen(path, "a");
    for (auto& r : h->prof_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.cat >= 0 && r.cat < TVC_PROF_NCAT) {
            ms[r.cat] += t; work[r.cat] += r.work; launches[r.cat] += 1;
            if (big_gemm && r.big_bytes > 0) { big_gemm[0] += r.big_bytes; big_gemm[1] += 1; big_gemm[2] += t; }
            if (dump && r.cat == TVC_PROF_GEMM) fprintf(dump, "%d %d %d %d %d %.6f\n", r.gI, r.gJ, r.gK, r.gP, r.gS, t);
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    h->prof_recs.clear();
    if (dump) fclose(dump);
    return TVC_OK;
}

int tvc_gemm_bf16(tvc_handle* h, const uint16_t* a_dev, const uint16_t* b_dev, const float* bias_dev, void* out_dev,
                  int32_t I, int32_t J, int32_t K, int64_t lda, int64_t ldb, int32_t ld_out, int32_t epilogue,
                  void* stream) {
    if (!h) return TVC_E_INVALID;
    if (lda == 0) lda = K;
    if (ldb == 0) ldb = K;
    if (I <= 0 || J <= 0 || K <= 0 || K % 64 != 0 || !a_dev || !b_dev || !out_dev || ld_out < I ||
        epilogue < 0 || epilogue > 3 || lda < K || ldb < K || lda % 8 != 0 || ldb % 8 != 0)
        return fail(h, TVC_E_INVALID, "tvc_gemm_bf16: need K % 64 == 0, ld_out >= I, lda / ldb >= K and multiples of 8");
    GemmLaunch g;
    g.A = a_dev; g.lda = lda; g.I = I; g.B = b_dev; g.ldb = ldb; g.J = J; g.K = K;
    g.bias = bias_dev; g.out = out_dev; g.ldo = ld_out; g.epilogue = epilogue;
    {
        int rc = ensure(h, WS_SPLITK, (size_t)256 * 256 * 256 * 4);       // split-K scratch (small or tail tiles)
        if (rc) return rc;
        g.splitk_ws = (float*)h->ws[WS_SPLITK].p; g.splitk_ws_bytes = h->ws[WS_SPLITK].n;
    }
    HIP_TRY(launch_gemm_bf16(g, (hipStream_t)stream));
    return TVC_OK;
}

int tvc_attention(tvc_handle* h, const uint16_t* qkv_dev, uint16_t* out_dev, int32_t n_seq, int32_t seq_len,
                  int32_t heads, int32_t causal, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!qkv_dev || !out_dev) return fail(h, TVC_E_INVALID, "tvc_attention: NULL buffer");
    HIP_TRY(launch_attention(qkv_dev, out_dev, nullptr, n_seq, seq_len, heads, causal, (hipStream_t)stream));
    return TVC_OK;
}

int tvc_layernorm(tvc_handle* h, const float* x_dev, const float* g_dev, const float* b_dev, uint16_t* y_dev,
                  int32_t rows, int32_t d, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!x_dev || !g_dev || !b_dev || !y_dev) return fail(h, TVC_E_INVALID, "tvc_layernorm: NULL buffer");
    HIP_TRY(launch_layernorm(const_cast<float*>(x_dev), d, nullptr, nullptr, 0, g_dev, b_dev, y_dev, rows, d,
                             (hipStream_t)stream));
    return TVC_OK;
}

}  // extern "C"
