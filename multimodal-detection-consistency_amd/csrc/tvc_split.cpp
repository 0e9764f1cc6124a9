// Split-bf16 CLIP towers behind tvc_encode_image / tvc_encode_text / tvc_encode_text_hidden when
// TVC_OPT_TOWER_PRECISION = 2 (include/tvc.h): the fast <= 1e-4 mode.  Same arithmetic as the reference's fp32 towers
// (src/detector.py:461-485 via CLIPModel.encode_*) with every activation and weight carried to the matrix cores as
// hi | lo bf16 planes and every product formed from three bf16 MFMAs (split.hip): embeddings within ~1e-5 of the fp32
// CPU path at about a third of the bf16 mode's matrix rate -- against the 1/16 of the exact-f32 mode
// (tvc_precise.cpp, TVC_OPT_TOWER_PRECISION = 1), which stays the exact reference.
//
// Per layer:   Hs = split(LN1(X [+ D1 + D2 of the previous layer, written back]))       ln_split
//              QKV = Wqkv_s x Hs + b            (3 planes, fp32 out)                      gemm_ring4_kernel<F32>
//              Hs = split(attention(QKV))                                                 attention_split
//              D1 = Wo_s x Hs + b               (fp32)
//              Hs = split(LN2(X + D1))
//              U = W1_s x Hs + b                (fp32);  Ms = split(QuickGELU(U))         rows_split(gelu)
//              D2 = W2_s x Ms + b               (fp32)
// The residual projections are store-only GEMMs (as in the bf16 tower): the LayerNorm passes fold the fp32 deltas into X.
// Text rows are EOT-packed with variant prefix sharing exactly as in the bf16 mode (TVC_OPT_TEXT_PACKING / _GROUP): the
// work skipped is work whose results the causal tower never reads.  No pooled last layer (the mode is for parity).
#include "handle.hpp"

namespace {

struct SBufs { float *X, *QKV, *U, *D1, *D2, *CLS; uint16_t *Hs, *Ms; };

int64_t pad_tile_rows(int64_t rows) { return (rows + 255) / 256 * 256 + 256; }

int ensure_split_ws(tvc_handle* h, const tvc_tower_arch& a, int64_t rows, int n_seq, int wso, SBufs* b, size_t u_min = 0, size_t m_min = 0) {
    int rc;
    const int64_t rp = pad_tile_rows(rows);             // GEMM operands: readable rows to the next tile (+ one)
    size_t u_bytes = (size_t)rp * a.mlp * 4, m_bytes = (size_t)rp * a.mlp * 2 * 2;
    if (u_bytes < u_min) u_bytes = u_min;
    if (m_bytes < m_min) m_bytes = m_min;
    if ((rc = ensure(h, (Slot)(WS_SX + wso), (size_t)rp * a.width * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_SH + wso), (size_t)rp * a.width * 2 * 2))) return rc;
    if ((rc = ensure(h, (Slot)(WS_SQKV + wso), (size_t)rp * a.width * 3 * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_SU + wso), u_bytes))) return rc;
    if ((rc = ensure(h, (Slot)(WS_SM + wso), m_bytes))) return rc;
    if ((rc = ensure(h, (Slot)(WS_SDELTA1 + wso), (size_t)rp * a.width * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_SDELTA2 + wso), (size_t)rp * a.width * 4))) return rc;
    if ((rc = ensure(h, (Slot)(WS_SCLS + wso), (size_t)(n_seq + 8) * a.width * 4))) return rc;
    b->X = (float*)h->ws[WS_SX + wso].p; b->Hs = (uint16_t*)h->ws[WS_SH + wso].p;
    b->QKV = (float*)h->ws[WS_SQKV + wso].p; b->U = (float*)h->ws[WS_SU + wso].p;
    b->Ms = (uint16_t*)h->ws[WS_SM + wso].p; b->D1 = (float*)h->ws[WS_SDELTA1 + wso].p;
    b->D2 = (float*)h->ws[WS_SDELTA2 + wso].p; b->CLS = (float*)h->ws[WS_SCLS + wso].p;
    return TVC_OK;
}

// out fp32 [J, ldo] = (W_hi + W_lo)[I, K] x (B_hi + B_lo)[J, K]^T + bias, without the lo x lo term
int gemm3(tvc_handle* h, const uint16_t* Ws, int I, int K, const uint16_t* Bs, int64_t J, const float* bias, float* out, int64_t ldo,
          hipStream_t st, int splitk_slot) {
    GemmLaunch g;
    g.A = Ws; g.lda = 2 * (int64_t)K; g.I = I; g.B = Bs; g.ldb = 2 * (int64_t)K; g.J = (int)J; g.K = K;
    g.planes = 3;                                       // small terms first, the hi x hi products last
    g.a_plane_off[0] = 0; g.b_plane_off[0] = K;         // A_hi x B_lo
    g.a_plane_off[1] = K; g.b_plane_off[1] = 0;         // A_lo x B_hi
    g.a_plane_off[2] = 0; g.b_plane_off[2] = 0;         // A_hi x B_hi
    g.bias = bias; g.out = out; g.ldo = ldo; g.epilogue = TVC_EPI_F32;
    g.a_rows_padded = true; g.b_rows_padded = true;     // split weights / workspaces are allocated to whole tiles
    HIP_TRY(timed_gemm(h, g, st, splitk_slot));
    return TVC_OK;
}

int run_layers_split(tvc_handle* h, const tvc_tower_arch& a, const SplitLayer* sw, const tvc_layer_weights_f32* lw, int n_seq,
                     int seq_len, int causal, const int32_t* starts, int total_rows, const SBufs& b, hipStream_t st,
                     const int32_t* pfx, int splitk_slot) {
    const int d = a.width;
    const int rows = starts ? total_rows : n_seq * seq_len;
    int rc;
    bool pending = false;
    for (int l = 0; l < a.layers; ++l) {
        const tvc_layer_weights_f32& w = lw[l];
        {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)rows * d * (pending ? 24.0 : 8.0));
            HIP_TRY(launch_ln_split(b.X, d, nullptr, pending ? b.D1 : nullptr, pending ? b.D2 : nullptr, 1, w.ln1_g, w.ln1_b, b.Hs,
                                    nullptr, rows, d, st));
        }
        if ((rc = gemm3(h, sw[l].wqkv, 3 * d, d, b.Hs, rows, w.bqkv, b.QKV, 3 * d, st, splitk_slot))) return rc;
        {
            const double avg_len = starts ? (double)rows / n_seq : (double)seq_len;
            ProfScope ps(h, st, TVC_PROF_ATTENTION, 3.0 * 4.0 * n_seq * a.heads * avg_len * avg_len * 64 * (causal ? 0.5 : 1.0));
            HIP_TRY(launch_attention_split(b.QKV, b.Hs, starts, n_seq, seq_len, a.heads, causal, st, pfx));
        }
        if ((rc = gemm3(h, sw[l].wo, d, d, b.Hs, rows, w.bo, b.D1, d, st, splitk_slot))) return rc;
        {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)rows * d * 12.0);
            HIP_TRY(launch_ln_split(b.X, d, nullptr, b.D1, nullptr, 0, w.ln2_g, w.ln2_b, b.Hs, nullptr, rows, d, st));
        }
        if ((rc = gemm3(h, sw[l].w1, a.mlp, d, b.Hs, rows, w.b1, b.U, a.mlp, st, splitk_slot))) return rc;
        {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)rows * a.mlp * 8.0);
            HIP_TRY(launch_rows_split(b.U, a.mlp, b.Ms, rows, a.mlp, a.mlp, a.act == TVC_ACT_GELU ? 2 : 1, st));
        }
        if ((rc = gemm3(h, sw[l].w2, d, a.mlp, b.Ms, rows, w.b2, b.D2, d, st, splitk_slot))) return rc;
        pending = true;
    }
    return TVC_OK;
}

// fp32 [I, K] -> handle-owned hi | lo planes [round_up(I, 256), 2 * Kp] (zero rows / columns beyond I / K)
int split_weight(tvc_handle* h, const float* w, int I, int K, int Kp, uint16_t** out, hipStream_t st) {
    const size_t rows = ((size_t)I + 255) / 256 * 256;
    void* p = nullptr;
    if (hipMalloc(&p, rows * 2 * Kp * 2) != hipSuccess) return fail(h, TVC_E_NOMEM, "split weights: allocation failed");
    h->split_owned.push_back(p);
    HIP_TRY(hipMemsetAsync(p, 0, rows * 2 * Kp * 2, st));
    HIP_TRY(launch_rows_split(w, K, (uint16_t*)p, I, K, Kp, 0, st));
    *out = (uint16_t*)p;
    return TVC_OK;
}

int split_tower(tvc_handle* h, const tvc_tower_arch& a, const tvc_layer_weights_f32* lw, std::vector<SplitLayer>& out, hipStream_t st) {
    out.resize(a.layers);
    int rc;
    for (int l = 0; l < a.layers; ++l) {
        if ((rc = split_weight(h, lw[l].wqkv, 3 * a.width, a.width, a.width, &out[l].wqkv, st))) return rc;
        if ((rc = split_weight(h, lw[l].wo, a.width, a.width, a.width, &out[l].wo, st))) return rc;
        if ((rc = split_weight(h, lw[l].w1, a.mlp, a.width, a.width, &out[l].w1, st))) return rc;
        if ((rc = split_weight(h, lw[l].w2, a.width, a.mlp, a.mlp, &out[l].w2, st))) return rc;
    }
    return TVC_OK;
}

}  // namespace

void tvc_split_free(tvc_handle* h) {
    for (void* p : h->split_owned) if (p) (void)hipFree(p);
    h->split_owned.clear();
    h->vsplit.clear(); h->tsplit.clear();
    h->vsplit_patch = nullptr;
    h->split_ready = false;
}

// Build the hi | lo planes of every GEMM weight from the fp32 copies registered by tvc_set_weights_f32 (1.7 GB at
// ViT-L/14); synchronises the device once.  Called by tvc_set_option(TVC_OPT_TOWER_PRECISION, 2).
int tvc_split_prepare(tvc_handle* h) {
    if (h->split_ready) return TVC_OK;
    {
        // the split attention keeps K and V of a head as hi | lo images in LDS: at most 272 tokens per sequence (the bf16
        // and fp32 modes take 288, which tvc_create already enforces together with head_dim 64)
        const tvc_model_desc& md = h->desc;
        const int Tv = h->has_vision ? (md.image_size / md.patch) * (md.image_size / md.patch) + 1 : 0;
        if (Tv > 272 || (h->has_text && md.ctx > 272))
            return fail(h, TVC_E_INVALID, "TVC_OPT_TOWER_PRECISION = 2: sequences longer than 272 tokens are not supported by the split attention");
    }
    tvc_split_free(h);
    hipStream_t st = nullptr;
    int rc;
    const tvc_model_desc& m = h->desc;
    if (h->has_vision32) {
        if ((rc = split_tower(h, m.vision, h->vw32.layers, h->vsplit, st))) { tvc_split_free(h); return rc; }
        const int K = 3 * m.patch * m.patch, Kp = (K + 63) / 64 * 64;
        if ((rc = split_weight(h, h->vw32.patch_w, m.vision.width, K, Kp, &h->vsplit_patch, st))) { tvc_split_free(h); return rc; }
    }
    if (h->has_text32 && (rc = split_tower(h, m.text, h->tw32.layers, h->tsplit, st))) { tvc_split_free(h); return rc; }
    HIP_TRY(hipStreamSynchronize(st));
    h->split_ready = true;
    return TVC_OK;
}

int tvc_split_encode_image(tvc_handle* h, const float* pix_dev, int32_t B, float* out_dev, int32_t normalize, hipStream_t st) {
    if (!h->has_vision32 || !h->split_ready || h->vsplit.empty())
        return fail(h, TVC_E_STATE, "tvc_encode_image: TVC_OPT_TOWER_PRECISION = 2 needs tvc_set_weights_f32 (vision) before the option is set");
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.vision;
    const int gside = m.image_size / m.patch, P = gside * gside, T = P + 1, d = a.width, K = 3 * m.patch * m.patch;
    const int Kp = (K + 63) / 64 * 64;
    const int chunk = B < h->max_chunk_images ? B : h->max_chunk_images;
    SBufs b;
    int rc;
    // the stem parks its fp32 im2col rows in U, their planes in Ms and the patch embeddings in QKV ([n * P, d] fits [rows, 3d])
    if ((rc = ensure_split_ws(h, a, (int64_t)chunk * T, chunk, 0, &b, (size_t)pad_tile_rows((int64_t)chunk * P) * K * 4,
                              (size_t)pad_tile_rows((int64_t)chunk * P) * 2 * Kp * 2))) return rc;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int n = (B - b0 < chunk) ? B - b0 : chunk;
        const float* pix = pix_dev + (size_t)b0 * 3 * m.image_size * m.image_size;
        float* cols = b.U;                   // [n * P, K] fp32
        float* patch_out = b.QKV;            // [n * P, d] fp32
        HIP_TRY(launch_im2col_f32(pix, cols, n, m.image_size, m.patch, st));
        HIP_TRY(launch_rows_split(cols, K, b.Ms, (int64_t)n * P, K, Kp, 0, st));
        if ((rc = gemm3(h, h->vsplit_patch, d, Kp, b.Ms, (int64_t)n * P, nullptr, patch_out, d, st, -1))) return rc;
        HIP_TRY(launch_assemble_lnpre(patch_out, h->vw32.cls, h->vw32.pos, h->vw32.ln_pre_g, h->vw32.ln_pre_b, b.X, n, T, d, st));
        if ((rc = run_layers_split(h, a, h->vsplit.data(), h->vw32.layers, n, T, 0, nullptr, 0, b, st, nullptr, -1))) return rc;
        // ln_post on the class rows (row b * T) with the last layer's two pending deltas, projection (exact f32: n rows), L2 norm
        HIP_TRY(launch_ln_split(b.X, (int64_t)T * d, nullptr, b.D1, b.D2, 0, h->vw32.ln_post_g, h->vw32.ln_post_b, nullptr, b.CLS, n, d, st));
        float* out = out_dev + (size_t)b0 * m.embed_dim;
        {
            ProfScope ps(h, st, TVC_PROF_GEMM, 2.0 * m.embed_dim * (double)n * d);
            HIP_TRY(launch_gemm_f32(h->vw32.proj, d, b.CLS, d, nullptr, out, m.embed_dim, m.embed_dim, n, d, 0, st));
        }
        if (normalize) HIP_TRY(launch_l2norm_rows(out, n, m.embed_dim, st));
    }
    return TVC_OK;
}

// hidden_out != nullptr: ln_final at every position of the DENSE rows -> [Tn, ctx, width]; else pooled embeddings
int tvc_split_encode_text(tvc_handle* h, const int32_t* tok_dev, int32_t Tn, float* out_dev, int32_t normalize, float* hidden_out,
                          hipStream_t st) {
    if (!h->has_text32 || !h->split_ready || h->tsplit.empty())
        return fail(h, TVC_E_STATE, "tvc_encode_text: TVC_OPT_TOWER_PRECISION = 2 needs tvc_set_weights_f32 (text) before the option is set");
    const tvc_model_desc& m = h->desc;
    const tvc_tower_arch& a = m.text;
    const int d = a.width, ctx = m.ctx;
    const int wso = WS_S_N;
    int chunk = Tn < h->max_chunk_texts ? Tn : h->max_chunk_texts;
    const bool pack = h->pack_text && !hidden_out;
    const int G = (pack && h->text_group >= 2 && Tn % h->text_group == 0) ? h->text_group : 0;
    if (G && chunk >= G) chunk = chunk / G * G;
    const bool share = G && chunk % G == 0;
    SBufs b;
    int rc;
    if ((rc = ensure_split_ws(h, a, (int64_t)chunk * ctx, chunk, wso, &b))) return rc;
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
        if (pack) {
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
        HIP_TRY(launch_text_embed(tok, h->tw32.tok_emb, h->tw32.pos, b.X, eot, starts, n, ctx, d, m.vocab, st, pfx));
        if ((rc = run_layers_split(h, a, h->tsplit.data(), h->tw32.layers, n, max_len, 1, starts, total_rows, b, st, pfx, -1))) return rc;
        if (hidden_out) {
            HIP_TRY(launch_ln_split(b.X, d, nullptr, b.D1, b.D2, 0, h->tw32.ln_final_g, h->tw32.ln_final_b, nullptr,
                                    hidden_out + (size_t)t0 * ctx * d, n * ctx, d, st));
            continue;
        }
        HIP_TRY(launch_ln_split(b.X, d, eot, b.D1, b.D2, 0, h->tw32.ln_final_g, h->tw32.ln_final_b, nullptr, b.CLS, n, d, st));
        float* out = out_dev + (size_t)t0 * m.embed_dim;
        {
            ProfScope ps(h, st, TVC_PROF_GEMM, 2.0 * m.embed_dim * (double)n * d);
            HIP_TRY(launch_gemm_f32(h->tw32.proj, d, b.CLS, d, nullptr, out, m.embed_dim, m.embed_dim, n, d, 0, st));
        }
        if (normalize) HIP_TRY(launch_l2norm_rows(out, n, m.embed_dim, st));
    }
    return TVC_OK;
}

// ---- building blocks exported for parity tests (include/tvc.h)
extern "C" int tvc_gemm_split(tvc_handle* h, const float* w_dev, const float* x_dev, const float* bias_dev, float* out_dev, int32_t I,
                              int32_t J, int32_t K, int32_t ld_out, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (I <= 0 || J <= 0 || K <= 0 || K % 4 != 0 || !w_dev || !x_dev || !out_dev || ld_out < I || ld_out % 4 != 0)
        return fail(h, TVC_E_INVALID, "tvc_gemm_split: need K % 4 == 0, ld_out >= I and a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    const int Kp = (K + 63) / 64 * 64;
    const int64_t Ip = ((int64_t)I + 255) / 256 * 256, Jp = pad_tile_rows(J);
    int rc;
    if ((rc = ensure(h, WS_SH, (size_t)Ip * 2 * Kp * 2))) return rc;
    if ((rc = ensure(h, WS_SM, (size_t)Jp * 2 * Kp * 2))) return rc;
    uint16_t* ws = (uint16_t*)h->ws[WS_SH].p;
    uint16_t* xs = (uint16_t*)h->ws[WS_SM].p;
    HIP_TRY(hipMemsetAsync(ws, 0, (size_t)Ip * 2 * Kp * 2, st));
    HIP_TRY(launch_rows_split(w_dev, K, ws, I, K, Kp, 0, st));
    HIP_TRY(launch_rows_split(x_dev, K, xs, J, K, Kp, 0, st));
    return gemm3(h, ws, I, Kp, xs, J, bias_dev, out_dev, ld_out, st, -1);
}

extern "C" int tvc_attention_split(tvc_handle* h, const float* qkv_dev, uint16_t* out_planes_dev, const int32_t* starts_dev,
                                   int32_t n_seq, int32_t seq_len, int32_t heads, int32_t causal, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!qkv_dev || !out_planes_dev || seq_len < 1 || seq_len > 272 || heads < 1 || n_seq < 0)
        return fail(h, TVC_E_INVALID, "tvc_attention_split: need 1 <= seq_len <= 272 and non-NULL buffers");
    HIP_TRY(launch_attention_split(qkv_dev, out_planes_dev, starts_dev, n_seq, seq_len, heads, causal, (hipStream_t)stream, nullptr));
    return TVC_OK;
}
