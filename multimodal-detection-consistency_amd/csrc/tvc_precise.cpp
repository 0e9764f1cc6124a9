// fp32-grade CLIP towers behind tvc_encode_image / tvc_encode_text / tvc_encode_text_hidden when
// TVC_OPT_TOWER_PRECISION = 1 (include/tvc.h).  Same arithmetic as the bf16 towers of tvc_abi.cpp -- pre-LN blocks,
// QuickGELU, class / EOT pooling, projection, optional L2 normalisation -- but every tensor stays fp32 and every GEMM
// runs on the exact-f32 matrix instruction (precise.hip), so the result equals the reference's fp32 CPU towers up to
// the order of fp32 additions.  No EOT packing, prefix sharing or pooled last layer here: the mode exists for parity
// (validation, attack generation), not for speed.
#include "handle.hpp"

namespace {

struct P32Bufs { float *X, *H, *QKV, *MLP, *CLS; };

int ensure_p32(tvc_handle* h, const tvc_tower_arch& a, int64_t rows, int n_seq, int wso, P32Bufs* b, size_t qkv_min = 0) {
    int rc;
    if ((rc = ensure(h, (Slot)(WS_PX + wso), (size_t)rows * a.width * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_PH + wso), (size_t)rows * a.width * 4))) return rc;
    size_t qkv_bytes = (size_t)rows * a.width * 3 * 4;
    if (qkv_bytes < qkv_min) qkv_bytes = qkv_min;           // the vision stem parks its im2col rows here
    if ((rc = ensure(h, (Slot)(WS_PQKV + wso), qkv_bytes))) return rc;
    if ((rc = ensure(h, (Slot)(WS_PMLP + wso), (size_t)rows * a.mlp * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_PCLS + wso), (size_t)n_seq * a.width * 4 * 2))) return rc;
    b->X = (float*)h->ws[WS_PX + wso].p; b->H = (float*)h->ws[WS_PH + wso].p;
    b->QKV = (float*)h->ws[WS_PQKV + wso].p; b->MLP = (float*)h->ws[WS_PMLP + wso].p;
    b->CLS = (float*)h->ws[WS_PCLS + wso].p;
    return TVC_OK;
}

int gemm32(tvc_handle* h, const float* W, int I, int K, const float* X, int64_t J, const float* bias, float* out,
           int64_t ldo, int epi, hipStream_t st) {
    ProfScope ps(h, st, TVC_PROF_GEMM, 2.0 * I * (double)J * K);
    HIP_TRY(launch_gemm_f32(W, K, X, K, bias, out, ldo, I, (int)J, K, epi, st));
    return TVC_OK;
}

int run_layers_f32(tvc_handle* h, const tvc_tower_arch& a, const tvc_layer_weights_f32* lw, int n_seq, int T, int causal,
                   const P32Bufs& b, hipStream_t st) {
    const int d = a.width;
    const int64_t rows = (int64_t)n_seq * T;
    int rc;
    for (int l = 0; l < a.layers; ++l) {
        const tvc_layer_weights_f32& w = lw[l];
        HIP_TRY(launch_layernorm(b.X, d, nullptr, nullptr, 0, w.ln1_g, w.ln1_b, nullptr, (int)rows, d, st, nullptr, 0, nullptr, b.H));
        if ((rc = gemm32(h, w.wqkv, 3 * d, d, b.H, rows, w.bqkv, b.QKV, 3 * d, 0, st))) return rc;
        {
            ProfScope ps(h, st, TVC_PROF_ATTENTION, 4.0 * n_seq * a.heads * (double)T * T * 64 * (causal ? 0.5 : 1.0));
            HIP_TRY(launch_attention_f32(b.QKV, b.H, n_seq, T, a.heads, causal, st));
        }
        if ((rc = gemm32(h, w.wo, d, d, b.H, rows, w.bo, b.X, d, 2, st))) return rc;          // X += out-proj
        HIP_TRY(launch_layernorm(b.X, d, nullptr, nullptr, 0, w.ln2_g, w.ln2_b, nullptr, (int)rows, d, st, nullptr, 0, nullptr, b.H));
        if ((rc = gemm32(h, w.w1, a.mlp, d, b.H, rows, w.b1, b.MLP, a.mlp, a.act == TVC_ACT_GELU ? 0 : 1, st))) return rc; // QuickGELU in the epilogue
        if (a.act == TVC_ACT_GELU) HIP_TRY(launch_gelu_erf_f32(b.MLP, rows * a.mlp, st));                                   // erf GELU: a row pass
        if ((rc = gemm32(h, w.w2, d, a.mlp, b.MLP, rows, w.b2, b.X, d, 2, st))) return rc;     // X += fc2
    }
    return TVC_OK;
}

}  // namespace

int tvc_precise_encode_image(tvc_handle* h, const float* pix_dev, int32_t B, float* out_dev, int32_t normalize, hipStream_t st) {
    if (!h->has_vision32) return fail(h, TVC_E_STATE, "tvc_encode_image: TVC_OPT_TOWER_PRECISION = 1 needs tvc_set_weights_f32 (vision)");
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.vision;
    const int gside = m.image_size / m.patch, P = gside * gside, T = P + 1, d = a.width, K = 3 * m.patch * m.patch;
    const int chunk = B < h->max_chunk_images ? B : h->max_chunk_images;
    P32Bufs b;
    int rc;
    if ((rc = ensure_p32(h, a, (int64_t)chunk * T, chunk, 0, &b, (size_t)chunk * P * K * 4))) return rc;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int n = (B - b0 < chunk) ? B - b0 : chunk;
        const float* pix = pix_dev + (size_t)b0 * 3 * m.image_size * m.image_size;
        float* cols = b.QKV;                 // [n*P, K] fp32
        float* patch_out = b.MLP;            // [n*P, d]
        HIP_TRY(launch_im2col_f32(pix, cols, n, m.image_size, m.patch, st));
        if ((rc = gemm32(h, h->vw32.patch_w, d, K, cols, (int64_t)n * P, nullptr, patch_out, d, 0, st))) return rc;
        HIP_TRY(launch_assemble_lnpre(patch_out, h->vw32.cls, h->vw32.pos, h->vw32.ln_pre_g, h->vw32.ln_pre_b, b.X, n, T, d, st));
        if ((rc = run_layers_f32(h, a, h->vw32.layers, n, T, 0, b, st))) return rc;
        // ln_post on the class rows (row b*T), projection, L2 norm
        HIP_TRY(launch_layernorm(b.X, (int64_t)T * d, nullptr, nullptr, 0, h->vw32.ln_post_g, h->vw32.ln_post_b, nullptr, n, d,
                                 st, nullptr, 0, nullptr, b.CLS));
        float* out = out_dev + (size_t)b0 * m.embed_dim;
        if ((rc = gemm32(h, h->vw32.proj, m.embed_dim, d, b.CLS, n, nullptr, out, m.embed_dim, 0, st))) return rc;
        if (normalize) HIP_TRY(launch_l2norm_rows(out, n, m.embed_dim, st));
    }
    return TVC_OK;
}

// hidden_out != nullptr: ln_final at every position -> [Tn, ctx, width] (tvc_encode_text_hidden); else pooled embeddings
int tvc_precise_encode_text(tvc_handle* h, const int32_t* tok_dev, int32_t Tn, float* out_dev, int32_t normalize,
                            float* hidden_out, hipStream_t st) {
    if (!h->has_text32) return fail(h, TVC_E_STATE, "tvc_encode_text: TVC_OPT_TOWER_PRECISION = 1 needs tvc_set_weights_f32 (text)");
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.text;
    const int d = a.width, ctx = m.ctx;
    const int chunk = Tn < h->max_chunk_texts ? Tn : h->max_chunk_texts;
    P32Bufs b;
    int rc;
    if ((rc = ensure_p32(h, a, (int64_t)chunk * ctx, chunk, WS_P_N_END - WS_PX, &b))) return rc;
    if ((rc = ensure(h, WS_PEOT, (size_t)chunk * 4))) return rc;
    int32_t* eot = (int32_t*)h->ws[WS_PEOT].p;
    for (int t0 = 0; t0 < Tn; t0 += chunk) {
        const int n = (Tn - t0 < chunk) ? Tn - t0 : chunk;
        HIP_TRY(launch_text_embed(tok_dev + (size_t)t0 * ctx, h->tw32.tok_emb, h->tw32.pos, b.X, eot, nullptr, n, ctx, d,
                                  m.vocab, st, nullptr));
        if ((rc = run_layers_f32(h, a, h->tw32.layers, n, ctx, 1, b, st))) return rc;
        if (hidden_out) {
            HIP_TRY(launch_layernorm(b.X, d, nullptr, nullptr, 0, h->tw32.ln_final_g, h->tw32.ln_final_b, nullptr, n * ctx, d, st,
                                     nullptr, 0, nullptr, hidden_out + (size_t)t0 * ctx * d));
            continue;
        }
        HIP_TRY(launch_layernorm(b.X, d, eot, nullptr, 0, h->tw32.ln_final_g, h->tw32.ln_final_b, nullptr, n, d, st, nullptr, 0,
                                 nullptr, b.CLS));
        float* out = out_dev + (size_t)t0 * m.embed_dim;
        if ((rc = gemm32(h, h->tw32.proj, m.embed_dim, d, b.CLS, n, nullptr, out, m.embed_dim, 0, st))) return rc;
        if (normalize) HIP_TRY(launch_l2norm_rows(out, n, m.embed_dim, st));
    }
    return TVC_OK;
}

extern "C" int tvc_set_weights_f32(tvc_handle* h, const tvc_vision_weights_f32* vision, const tvc_text_weights_f32* text) {
    if (!h) return TVC_E_INVALID;
    if (vision) {
        if (!h->has_vision) return fail(h, TVC_E_STATE, "tvc_set_weights_f32: handle was created without a vision tower");
        if (!vision->layers || !vision->patch_w || !vision->proj) return fail(h, TVC_E_INVALID, "tvc_set_weights_f32: NULL vision weights");
        h->vw32 = *vision;
        h->vlayers32.assign(vision->layers, vision->layers + h->desc.vision.layers);
        h->vw32.layers = h->vlayers32.data();
        h->has_vision32 = true;
    }
    if (text) {
        if (!h->has_text) return fail(h, TVC_E_STATE, "tvc_set_weights_f32: handle was created without a text tower");
        if (!text->layers || !text->tok_emb || !text->proj) return fail(h, TVC_E_INVALID, "tvc_set_weights_f32: NULL text weights");
        h->tw32 = *text;
        h->tlayers32.assign(text->layers, text->layers + h->desc.text.layers);
        h->tw32.layers = h->tlayers32.data();
        h->has_text32 = true;
    }
    // the split-bf16 planes (TVC_OPT_TOWER_PRECISION = 2) are derived from these tensors: rebuild them when in use
    if (vision || text) {
        tvc_split_free(h);
        if (h->tower_precision == 2) return tvc_split_prepare(h);
    }
    return TVC_OK;
}

extern "C" int tvc_gemm_f32(tvc_handle* h, const float* w_dev, const float* x_dev, const float* bias_dev, float* out_dev, int32_t I,
                            int32_t J, int32_t K, int32_t ld_out, int32_t epilogue, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (I <= 0 || J <= 0 || K <= 0 || K % 4 != 0 || !w_dev || !x_dev || !out_dev || ld_out < I || epilogue < 0 || epilogue > 2)
        return fail(h, TVC_E_INVALID, "tvc_gemm_f32: need K % 4 == 0, ld_out >= I, epilogue in [0, 2]");
    HIP_TRY(launch_gemm_f32(w_dev, K, x_dev, K, bias_dev, out_dev, ld_out, I, J, K, epilogue, (hipStream_t)stream));
    return TVC_OK;
}

extern "C" int tvc_attention_f32(tvc_handle* h, const float* qkv_dev, float* out_dev, int32_t n_seq, int32_t seq_len, int32_t heads,
                                 int32_t causal, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!qkv_dev || !out_dev || seq_len < 1 || seq_len > 288 || heads < 1 || n_seq < 0)
        return fail(h, TVC_E_INVALID, "tvc_attention_f32: need 1 <= seq_len <= 288 and non-NULL buffers");
    HIP_TRY(launch_attention_f32(qkv_dev, out_dev, n_seq, seq_len, heads, causal, (hipStream_t)stream));
    return TVC_OK;
}
