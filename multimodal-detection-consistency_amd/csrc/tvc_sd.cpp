// Latent-diffusion reference generator behind the C-ABI (include/tvc.h, "latent-diffusion reference generator"):
// UNet2DConditionModel evaluation, AutoencoderKL.decode and the PNDM (PLMS) sampling loop with classifier-free
// guidance -- what src/sd_ref.py:389-399 / experiments/defenses/generative_ref.py:139-147 reach through
// diffusers.StableDiffusionPipeline.  Geometry from the config.json files the reference holds
// (cache/sd/models--runwayml--stable-diffusion-v1-5/snapshots/*/{unet,vae,scheduler}).
//
// Layout: activations are bf16 token-major [n * H * W, C] (NHWC); every convolution and linear layer is ONE call of
// the tower GEMM (gemm.hip; 3x3 convolutions gather their rows with sd_im2col3x3 -- stride-2 and nearest-2x-upsample
// forms included --, 1x1 convolutions read the rows as they are); GroupNorm / LayerNorm / GEGLU / residual adds are
// streaming row kernels (sd_ops.hip); attention is the streaming kernel of sd_attention.hip (VAE: one 512-wide head,
// done as two GEMMs around a row softmax).  All activations of one evaluation live in ONE arena sized by a dry run
// (288 GB of HBM: nothing is recomputed, nothing is freed inside an evaluation except im2col scratch).
#include "handle.hpp"

#include <algorithm>
#include <cmath>
#include <unordered_map>

struct SdState {
    tvc_sd_desc d{};
    std::unordered_map<std::string, const void*> w;
    bool has_unet = false, has_vae = false;
    void* temb_w = nullptr;      // bf16 [temb_total, time_dim]: every resnet's time_emb_proj.weight, gathered
    void* temb_b = nullptr;      // fp32 [temb_total]
    int temb_total = 0;
    std::unordered_map<std::string, int> temb_off;
    std::vector<float> alphas_cumprod;
};

void tvc_sd_free(tvc_handle* h) {
    if (!h) return;
    if (h->sd_aux) { (void)hipStreamSynchronize(h->sd_aux); (void)hipStreamDestroy(h->sd_aux); h->sd_aux = nullptr; }
    if (h->sd_fork) { (void)hipEventDestroy(h->sd_fork); h->sd_fork = nullptr; }
    if (h->sd_join) { (void)hipEventDestroy(h->sd_join); h->sd_join = nullptr; }
    if (!h->sd) return;
    if (h->sd->temb_w) (void)hipFree(h->sd->temb_w);
    if (h->sd->temb_b) (void)hipFree(h->sd->temb_b);
    delete h->sd;
    h->sd = nullptr;
}

namespace {

struct Act {             // bf16 token-major activation
    uint16_t* p = nullptr;
    int n = 0, H = 0, W = 0, C = 0;
    bool pad = false;    // PADDED layout (sd_ops.hip, tok_row): (H + 2) x (W + 2) rows per image -- what the 9-plane conv GEMM reads / writes
    int64_t tok() const { return (int64_t)n * H * W; }
    int64_t rows() const { return pad ? (int64_t)n * (H + 2) * (W + 2) : tok(); }
};

// One evaluation: arena bump allocator + first-error latch (after an error every op is a no-op).
struct Run {
    tvc_handle* h;
    SdState* S;
    hipStream_t st;
    bool dry;                 // sizing pass: allocate, launch nothing
    char* base = nullptr;
    size_t off = 0, high = 0;
    size_t cap = 0;           // bytes behind `base` (the real pass): an allocation beyond it is an error, never a wild pointer
    int rc = TVC_OK;
    // Step-invariant tensors of a sampling loop (the bf16 text states and every cross-attention's key / value projection
    // of them: functions of the prompt only) live in their OWN bump region, walked in the same order by every evaluation:
    // keep = 1 computes them (the loop's first evaluation), keep = 2 finds them there; keep = 0 (a lone evaluation): arena.
    int keep = 0;
    char* keep_base = nullptr;
    size_t keep_off = 0, keep_high = 0, keep_cap = 0;
    void* alloc_keep(size_t bytes) {
        if (!keep) return alloc(bytes);
        keep_off = (keep_off + 255) & ~(size_t)255;
        void* p = keep_base ? keep_base + keep_off : nullptr;
        keep_off += bytes;
        if (keep_off > keep_high) keep_high = keep_off;
        if (keep_base && keep_off > keep_cap) {
            if (rc == TVC_OK) rc = fail(h, TVC_E_STATE, "tvc_sd: step-invariant region overrun");
            return nullptr;
        }
        return p;
    }
    bool keep_fill() const { return keep != 2; }      // these tensors are computed in this evaluation
    // K split of a GEMM, chosen from the PER-SAMPLE shape only (never from the launch size): every sample's arithmetic is
    // then the same whatever batch it is generated in, so an image is bit-for-bit independent of its batch mates and of the
    // chunking of tvc_sd_generate (the reference's seed policy, src/sd_ref.py:389-412, promises reproducible references).
    // Nominal batch: 24 samples (12 images x classifier-free guidance, the batch bench.py generates) -- K is split only
    // where even that batch leaves most CUs without a tile (the 8 x 8 level, the time / text projections).  A split that
    // filled the chip for a single image (S = 4 .. 16 at the 16 x 16 level) cost the 12-image batch 20 % of its GEMM time
    // (r04_bench1: 602 against 497 ms) -- throughput of the batched generator is what configs[4] measures; one image alone
    // runs its low-resolution levels on fewer workgroups than CUs.
#ifndef TVC_SD_NOMINAL
#define TVC_SD_NOMINAL 24
#endif
    static constexpr int NOMINAL_SAMPLES = TVC_SD_NOMINAL;
    static int fixed_split(int I, int K, int planes, int64_t rows_per_sample) {
        const int64_t tiles0 = (int64_t)((I + 255) / 256) * ((NOMINAL_SAMPLES * rows_per_sample + 255) / 256);
        const int nk64 = (int)((int64_t)K * planes / 64);
        int64_t S = tiles0 > 0 ? 256 / tiles0 : 1;
        if (S > nk64 / 4) S = nk64 / 4;
        if (S > 16) S = 16;
        // the split kernels run the one-tile loop at about half the ring kernel's rate: a two-way split of a shallow K
        // (the 16 x 16 level's linear layers: 120 tiles x 20 K-tiles) loses to 120 ring workgroups (measured 418 against
        // ~550 TFLOP/s); it pays from three ways on, or on a deep K (>= 32 K-tiles per slice)
        if (S == 2 && nk64 / 2 < 32) S = 1;
        // slices of equal depth (S divides the K-tile count) run in the ring kernel (gemm.hip, gemm_ring4_split_kernel)
        while (S > 2 && nk64 % S != 0) --S;
        if (S == 2 && (nk64 % 2 != 0 || nk64 / 2 < 32)) S = 1;
        return S >= 2 ? (int)S : 1;
    }
    // launch with the fixed split; the fp32 partial tiles are scratch of the arena
    void launch_fixed(GemmLaunch& g, int64_t rows_per_sample, const char* what) {
        const int S = fixed_split(g.I, g.K, g.planes, rows_per_sample);
        g.splitk_fixed = S;
        const size_t mark = off;
        if (S >= 2) {
            const size_t tiles = (size_t)((g.I + 255) / 256) * (((size_t)g.J + 255) / 256);
            g.splitk_ws_bytes = tiles * S * 256 * 256 * 4;
            g.splitk_ws = (float*)alloc(g.splitk_ws_bytes);
        }
        if (live()) hip(timed_gemm(h, g, st), what);
        off = mark;
    }

    bool live() const { return rc == TVC_OK && !dry; }
    void hip(hipError_t e, const char* what) {
        if (rc == TVC_OK && e != hipSuccess) rc = fail(h, TVC_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        if (off > high) high = off;
        if (base && off > cap) {          // the sizing pass and the real pass must walk the same allocations
            if (rc == TVC_OK) rc = fail(h, TVC_E_STATE, "tvc_sd: arena overrun (the dry run sized " + std::to_string(cap) + " bytes)");
            return nullptr;
        }
        return p;
    }
    // GEMM operands get readable rows up to the next multiple of 256 (+ one tile): gemm.hip's ring form stages whole tiles
    static int64_t pad_rows_of(int64_t r) { return (r + 255) / 256 * 256 + 256; }
    Act act(int n, int H, int W, int C) {
        Act a; a.n = n; a.H = H; a.W = W; a.C = C;
        const int64_t rows = pad_rows_of(a.tok());
        a.p = (uint16_t*)alloc((size_t)rows * C * 2);
        return a;
    }
    // padded layout: + the rows a shifted tap reads beyond the last tile
    Act act_padded(int n, int H, int W, int C) {
        Act a; a.n = n; a.H = H; a.W = W; a.C = C; a.pad = true;
        const int64_t rows = (a.rows() + 255) / 256 * 256 + 256 + 3 * (W + 2);
        a.p = (uint16_t*)alloc((size_t)rows * C * 2);
        return a;
    }
    const void* W(const std::string& name) {
        auto it = S->w.find(name);
        if (it == S->w.end()) {
            if (rc == TVC_OK) rc = fail(h, TVC_E_STATE, "tvc_sd: tensor '" + name + "' was not given to tvc_sd_load");
            return nullptr;
        }
        return it->second;
    }
    bool has(const std::string& name) const { return S->w.count(name) != 0; }

    // out[j, i] = sum_k B[j, k] A[i, k] + bias[i]
    // rows_per_sample: B rows of ONE sample (what the K split is chosen from)
    void gemm(const void* A, int I, int K, const uint16_t* B, int64_t J, const float* bias, void* out, int64_t ldo, int epi,
              int64_t rows_per_sample) {
        GemmLaunch g;
        g.A = (const uint16_t*)A; g.lda = K; g.I = I; g.B = B; g.ldb = K; g.J = (int)J; g.K = K;
        g.bias = bias; g.out = out; g.ldo = ldo; g.epilogue = epi; g.b_rows_padded = true;
        g.a_rows_padded = true;            // tvc_sd_load's contract: GEMM weights are readable to the next multiple of 256 rows
        // the 16 x 16 / 8 x 8 levels and the time / text projections are a few tiles with a long K: split it over the idle CUs
        launch_fixed(g, rows_per_sample, "sd gemm");
    }
    // 3x3 convolution, stride 1, padding 1, as ONE GEMM of 9 K-planes: xp is in the padded layout with ZERO border rows
    // (a GroupNorm output), so tap (ky, kx) is the same token rows shifted by ky * (W + 2) + kx -- no im2col rows, the
    // activations are read once (from L2 for eight of the nine taps).  Output row q = input row q + (W + 3) (the tap
    // centre), i.e. the result is in the padded layout too; its border rows hold garbage nobody reads.
    // out_f32: fp32 output [rows, Cout] (the model's last convolution) instead of a bf16 activation.
    Act conv3x3_planes(const Act& xp, const std::string& prefix, int Cout, float** out_f32 = nullptr) {
        Act y;
        float* of = nullptr;
        if (out_f32) {
            y = xp; y.C = Cout; y.p = nullptr;
            of = (float*)alloc((size_t)(xp.rows() + 256) * Cout * 4);
            *out_f32 = of;
        } else {
            y = act_padded(xp.n, xp.H, xp.W, Cout);
        }
        const void* w = W(prefix + "weight");
        const float* b = (const float*)W(prefix + "bias");
        const int Wp = xp.W + 2;
        GemmLaunch g;
        g.A = (const uint16_t*)w; g.lda = 9 * (int64_t)xp.C; g.I = Cout; g.B = xp.p; g.ldb = xp.C;
        g.J = (int)(xp.rows() - 2 * (Wp + 1)); g.K = xp.C; g.planes = 9;
        for (int t = 0; t < 9; ++t) { g.a_plane_off[t] = t * xp.C; g.b_plane_off[t] = ((t / 3) * Wp + t % 3) * xp.C; }
        g.bias = b; g.ldo = Cout; g.b_rows_padded = true; g.a_rows_padded = true;
        if (out_f32) { g.out = of ? of + (size_t)(Wp + 1) * Cout : nullptr; g.epilogue = TVC_EPI_F32; }
        else { g.out = y.p ? y.p + (size_t)(Wp + 1) * Cout : nullptr; g.epilogue = TVC_EPI_BF16; }
        launch_fixed(g, (int64_t)(xp.H + 2) * Wp, "sd conv gemm");
        return y;
    }
    // linear / 1x1 convolution on token rows
    Act linear(const Act& x, const std::string& wname, const std::string& bname, int Cout) {
        Act y = act(x.n, x.H, x.W, Cout);
        const void* w = W(wname);
        const float* b = bname.empty() ? nullptr : (const float*)W(bname);
        gemm(w, Cout, x.C, x.p, x.tok(), b, y.p, Cout, TVC_EPI_BF16, (int64_t)x.H * x.W);
        return y;
    }
    // 3x3 convolution, padding 1: stride 1 / 2, or on the nearest-2x upsampling of x
    // Upsample2D: nearest-2x then a 3x3 convolution -- the upsampled tensor is written ONCE in the padded layout (4x the
    // input's rows, not the 36x of im2col rows) and convolved as nine planes; the dense result is what the next block reads
    Act upsample_conv(const Act& x, const std::string& prefix, int Cout) {
        Act y = act(x.n, 2 * x.H, 2 * x.W, Cout);
        const size_t mark = off;
        Act xp = act_padded(x.n, 2 * x.H, 2 * x.W, x.C);
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)xp.rows() * x.C * 2.5);
            hip(sd_relayout(x.p, xp.p, x.n, 2 * x.H, 2 * x.W, x.C, x.pad, 1, 1, st), "sd_relayout(up)");
        }
        Act yp = conv3x3_planes(xp, prefix, Cout);
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)y.tok() * Cout * 4.0);
            hip(sd_relayout(yp.p, y.p, y.n, y.H, y.W, Cout, 1, 0, 0, st), "sd_relayout");
        }
        off = mark;
        return y;
    }
    Act conv3x3(const Act& x, const std::string& prefix, int Cout, int stride = 1, int up = 0) {
        const int Hs = up ? 2 * x.H : x.H, Ws = up ? 2 * x.W : x.W;
        const int Ho = (Hs - 1) / stride + 1, Wo = (Ws - 1) / stride + 1;
        Act y = act(x.n, Ho, Wo, Cout);
        const size_t mark = off;
        Act col = act(x.n, Ho, Wo, 9 * x.C);
        const void* w = W(prefix + "weight");
        const float* b = (const float*)W(prefix + "bias");
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)col.tok() * 9 * x.C * 4.0);      // rows written once, read once by the GEMM
            hip(sd_im2col3x3(x.p, col.p, x.n, x.H, x.W, x.C, stride, up, st), "sd_im2col3x3");
        }
        gemm(w, Cout, 9 * x.C, col.p, col.tok(), b, y.p, Cout, TVC_EPI_BF16, (int64_t)Ho * Wo);
        off = mark;
        return y;
    }
    Act groupnorm(const Act& x, const std::string& prefix, float eps, int silu, const float* tadd = nullptr, int64_t ld_t = 0,
                  bool out_pad = false) {
        Act y = out_pad ? act_padded(x.n, x.H, x.W, x.C) : act(x.n, x.H, x.W, x.C);
        const size_t mark = off;
        float* ws = (float*)alloc(sd_groupnorm_ws_floats(x.n, x.H * x.W, S->d.norm_groups) * 4);
        const float* g = (const float*)W(prefix + "weight");
        const float* b = (const float*)W(prefix + "bias");
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)x.tok() * x.C * 6.0);
            hip(sd_groupnorm(x.p, tadd, ld_t, g, b, y.p, x.n, x.H, x.W, x.C, S->d.norm_groups, eps, silu, x.pad, out_pad, ws, st),
                "sd_groupnorm");
        }
        off = mark;
        return y;
    }
    Act layernorm(const Act& x, const std::string& prefix) {
        Act y = act(x.n, x.H, x.W, x.C);
        const float* g = (const float*)W(prefix + "weight");
        const float* b = (const float*)W(prefix + "bias");
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)x.tok() * x.C * 4.0);
            hip(sd_layernorm_bf16(x.p, g, b, y.p, x.tok(), x.C, 1e-5f, st), "sd_layernorm");
        }
        return y;
    }
    // sum = a + b (the residual stream) and LayerNorm(sum) in one pass; bit-identical to add() followed by layernorm()
    Act add_layernorm(const Act& a, const Act& b, const std::string& prefix, Act& sum) {
        sum = act(a.n, a.H, a.W, a.C);
        Act y = act(a.n, a.H, a.W, a.C);
        const float* g = (const float*)W(prefix + "weight");
        const float* bb = (const float*)W(prefix + "bias");
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)a.tok() * a.C * 8.0);
            hip(sd_layernorm_bf16(a.p, g, bb, y.p, a.tok(), a.C, 1e-5f, st, b.p, sum.p), "sd_add_layernorm");
        }
        return y;
    }
    Act add(const Act& a, const Act& b) {
        Act y = act(a.n, a.H, a.W, a.C);
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)a.tok() * a.C * 6.0);
            hip(sd_add_bf16(a.p, b.p, y.p, a.tok() * a.C, st), "sd_add");
        }
        return y;
    }
    void attention(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, uint16_t* o,
                   int64_t ldo, int n, int heads, int Tq, int Tk, int dh) {
        if (!live()) return;
        ProfScope ps(h, st, TVC_PROF_ATTENTION, 4.0 * n * heads * (double)Tq * Tk * dh);
        hip(sd_flash_attention(q, ldq, k, ldk, v, ldv, o, ldo, n, heads, Tq, Tk, dh, st), "sd_flash_attention");
    }

    // ResnetBlock2D: x + conv2(silu(gn2(conv1(silu(gn1(x))) + time projection))), 1x1 shortcut when the widths differ
    Act resnet(const Act& x, const std::string& p, int Cout, const float* tadd_all, float eps) {
        Act out = act(x.n, x.H, x.W, Cout);
        const size_t mark = off;
        Act h1 = groupnorm(x, p + "norm1.", eps, 1, nullptr, 0, true);
        Act h2 = conv3x3_planes(h1, p + "conv1.", Cout);
        const float* tadd = nullptr;
        if (tadd_all) {
            auto it = S->temb_off.find(p);
            if (it == S->temb_off.end()) { if (rc == TVC_OK) rc = fail(h, TVC_E_STATE, "tvc_sd: no time projection for " + p); }
            else tadd = tadd_all + it->second;
        }
        Act h3 = groupnorm(h2, p + "norm2.", eps, 1, tadd, S->temb_total, true);
        Act h4 = conv3x3_planes(h3, p + "conv2.", Cout);
        Act sc = x;
        if (x.C != Cout) sc = linear(x, p + "conv_shortcut.weight", p + "conv_shortcut.bias", Cout);
        if (live()) {
            ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)x.tok() * Cout * 6.0);
            hip(sd_add_padded(sc.p, h4.p, out.p, x.n, x.H, x.W, Cout, st), "sd_add_padded");
        }
        off = mark;
        return out;
    }

    // Transformer2DModel with one BasicTransformerBlock; ctx16 bf16 [n * ctx, cross_attention_dim]
    Act transformer(const Act& x, const std::string& p, const uint16_t* ctx16, int heads) {
        const int C = x.C, dh = C / heads, T = x.H * x.W, n = x.n;
        const std::string t = p + "transformer_blocks.0.";
        Act out = act(n, x.H, x.W, C);
        const size_t mark = off;
        Act g = groupnorm(x, p + "norm.", 1e-6f, 0);
        Act hs = linear(g, p + "proj_in.weight", p + "proj_in.bias", C);
        Act n2, n3, hs1;
        {   // self-attention
            Act n1 = layernorm(hs, t + "norm1.");
            Act qkv = linear(n1, t + "attn1.to_qkv.weight", "", 3 * C);
            Act a = act(n, x.H, x.W, C);
            attention(qkv.p, 3 * C, qkv.p + C, 3 * C, qkv.p + 2 * C, 3 * C, a.p, C, n, heads, T, T, dh);
            Act o = linear(a, t + "attn1.to_out.0.weight", t + "attn1.to_out.0.bias", C);
            n2 = add_layernorm(hs, o, t + "norm2.", hs1);      // hs1 = hs + o; n2 = norm2(hs1)
            hs = hs1;
        }
        {   // cross-attention onto the text states
            Act q = linear(n2, t + "attn2.to_q.weight", "", C);
            Act cx; cx.n = n; cx.H = 1; cx.W = S->d.ctx; cx.C = S->d.cross_attention_dim; cx.p = const_cast<uint16_t*>(ctx16);
            Act kv = cx; kv.C = 2 * C;                       // step-invariant (alloc_keep)
            kv.p = (uint16_t*)alloc_keep((size_t)pad_rows_of(cx.tok()) * 2 * C * 2);
            if (keep_fill()) gemm(W(t + "attn2.to_kv.weight"), 2 * C, cx.C, cx.p, cx.tok(), nullptr, kv.p, 2 * C, TVC_EPI_BF16, S->d.ctx);
            else (void)W(t + "attn2.to_kv.weight");
            Act a = act(n, x.H, x.W, C);
            attention(q.p, C, kv.p, 2 * C, kv.p + C, 2 * C, a.p, C, n, heads, T, S->d.ctx, dh);
            Act o = linear(a, t + "attn2.to_out.0.weight", t + "attn2.to_out.0.bias", C);
            n3 = add_layernorm(hs, o, t + "norm3.", hs1);
            hs = hs1;
        }
        {   // GEGLU feed-forward
            Act gg = linear(n3, t + "ff.net.0.proj.weight", t + "ff.net.0.proj.bias", 8 * C);
            Act ge = act(n, x.H, x.W, 4 * C);
            if (live()) {
                ProfScope ps(h, st, TVC_PROF_ROWOPS, (double)x.tok() * C * 24.0);
                hip(sd_geglu(gg.p, ge.p, gg.tok(), 4 * C, st), "sd_geglu");
            }
            Act o = linear(ge, t + "ff.net.2.weight", t + "ff.net.2.bias", C);
            hs = add(hs, o);
        }
        Act po = linear(hs, p + "proj_out.weight", p + "proj_out.bias", C);
        if (live()) hip(sd_add_bf16(x.p, po.p, out.p, out.tok() * C, st), "sd_add");
        off = mark;
        return out;
    }

    // VAE AttentionBlock: one head over all C channels, scores materialised per image (T x T fp32)
    Act vae_attention(const Act& x, const std::string& p) {
        const int C = x.C, T = x.H * x.W, n = x.n;
        Act out = act(n, x.H, x.W, C);
        const size_t mark = off;
        Act g = groupnorm(x, p + "group_norm.", 1e-6f, 0);
        Act qkv = linear(g, p + "to_qkv.weight", p + "to_qkv.bias", 3 * C);
        Act a = act(n, x.H, x.W, C);
        const int64_t Tp = ((int64_t)T + 255) / 256 * 256 + 256;
        uint16_t* qc = (uint16_t*)alloc((size_t)Tp * C * 2);      // compact q / k rows of one image
        uint16_t* kc = (uint16_t*)alloc((size_t)Tp * C * 2);
        uint16_t* vT = (uint16_t*)alloc((size_t)(C + 512) * (size_t)((T + 63) / 64 * 64) * 2);     // [C, T]
        float* sc = (float*)alloc((size_t)T * T * 4);
        // `pr` holds the probabilities [T (readable to Tp), T] -- and, before them, the image's compact V rows [T, C]
        uint16_t* pr = (uint16_t*)alloc(std::max((size_t)Tp * T, (size_t)T * C) * 2);
        if (T % 64 != 0 && rc == TVC_OK) rc = fail(h, TVC_E_INVALID, "tvc_sd: VAE attention needs H * W % 64 == 0");
        for (int i = 0; i < (dry ? 1 : n) && rc == TVC_OK; ++i) {          // (dry run: one image sizes the per-image scratch)
            const uint16_t* rows = qkv.p + (int64_t)i * T * 3 * C;
            if (live()) {
                hip(hipMemcpy2DAsync(qc, (size_t)C * 2, rows, (size_t)3 * C * 2, (size_t)C * 2, T, hipMemcpyDeviceToDevice, st), "copy q");
                hip(hipMemcpy2DAsync(kc, (size_t)C * 2, rows + C, (size_t)3 * C * 2, (size_t)C * 2, T, hipMemcpyDeviceToDevice, st), "copy k");
                hip(hipMemcpy2DAsync(pr, (size_t)C * 2, rows + 2 * C, (size_t)3 * C * 2, (size_t)C * 2, T, hipMemcpyDeviceToDevice, st), "copy v");
                hip(launch_transpose_bf16(pr, vT, T, C, st), "transpose v");
            }
            // scores[q, key] = q . k  (A = keys, B = queries), probabilities, out[q, c] = sum_key P[q, key] V^T[c, key]
            gemm(kc, T, C, qc, T, nullptr, sc, T, TVC_EPI_F32, T);
            if (live()) hip(sd_softmax_rows(sc, pr, T, T, 1.0f / sqrtf((float)C), st), "sd_softmax_rows");
            gemm(vT, C, T, pr, T, nullptr, a.p + (int64_t)i * T * C, C, TVC_EPI_BF16, T);
        }
        Act o = linear(a, p + "proj_attn.weight", p + "proj_attn.bias", C);
        if (live()) hip(sd_add_bf16(x.p, o.p, out.p, out.tok() * C, st), "sd_add");
        off = mark;
        return out;
    }
};

int64_t pad_rows(int64_t r) { return (r + 255) / 256 * 256 + 256; }

// ---- UNet2DConditionModel.forward on 2-D latents
void unet_forward(Run& R, const float* latents, int n, int H, int W, float timestep, const float* ctx, float* eps_out) {
    SdState* S = R.S;
    const tvc_sd_desc& d = S->d;
    const int nb = d.n_blocks, c0 = d.block_out_channels[0], Tdim = 4 * c0;
    // text states -> bf16 rows
    uint16_t* ctx16 = (uint16_t*)R.alloc_keep((size_t)pad_rows((int64_t)n * d.ctx) * d.cross_attention_dim * 2);
    if (R.live() && R.keep_fill()) R.hip(sd_cast_silu(ctx, ctx16, (int64_t)n * d.ctx * d.cross_attention_dim, 0, R.st), "ctx cast");
    // time embedding MLP, then every resnet's time projection in one GEMM: tadd fp32 [n, temb_total]
    uint16_t* te = (uint16_t*)R.alloc((size_t)pad_rows(n) * c0 * 2);
    float* t1 = (float*)R.alloc((size_t)n * Tdim * 4);
    uint16_t* t1b = (uint16_t*)R.alloc((size_t)pad_rows(n) * Tdim * 2);
    float* t2 = (float*)R.alloc((size_t)n * Tdim * 4);
    uint16_t* t2b = (uint16_t*)R.alloc((size_t)pad_rows(n) * Tdim * 2);
    float* tadd = (float*)R.alloc((size_t)n * S->temb_total * 4);
    const void* w1 = R.W("time_embedding.linear_1.weight"); const float* b1 = (const float*)R.W("time_embedding.linear_1.bias");
    const void* w2 = R.W("time_embedding.linear_2.weight"); const float* b2 = (const float*)R.W("time_embedding.linear_2.bias");
    if (R.live()) R.hip(sd_timestep_embed(te, n, c0, timestep, R.st), "timestep embed");
    R.gemm(w1, Tdim, c0, te, n, b1, t1, Tdim, TVC_EPI_F32, 1);
    if (R.live()) R.hip(sd_cast_silu(t1, t1b, (int64_t)n * Tdim, 1, R.st), "silu");
    R.gemm(w2, Tdim, Tdim, t1b, n, b2, t2, Tdim, TVC_EPI_F32, 1);
    if (R.live()) R.hip(sd_cast_silu(t2, t2b, (int64_t)n * Tdim, 1, R.st), "silu");       // resnets apply SiLU to temb first
    R.gemm(S->temb_w, S->temb_total, Tdim, t2b, n, (const float*)S->temb_b, tadd, S->temb_total, TVC_EPI_F32, 1);

    // conv_in on the fp32 NCHW latents
    Act x = R.act(n, H, W, c0);
    {
        const size_t mark = R.off;
        uint16_t* col = (uint16_t*)R.alloc((size_t)pad_rows((int64_t)n * H * W) * 64 * 2);
        if (R.live()) R.hip(sd_im2col_in(latents, col, n, d.in_channels, H, W, 64, 1.0f, R.st), "im2col_in");
        R.gemm(R.W("conv_in.weight"), c0, 64, col, (int64_t)n * H * W, (const float*)R.W("conv_in.bias"), x.p, c0, TVC_EPI_BF16,
               (int64_t)H * W);
        R.off = mark;
    }
    std::vector<Act> skips;
    skips.push_back(x);
    const float eps = d.norm_eps;
    auto heads_at = [&](int level) { return d.heads_per_block[level] > 0 ? d.heads_per_block[level] : d.heads; };
    for (int i = 0; i < nb; ++i) {
        const int c = d.block_out_channels[i];
        for (int j = 0; j < d.layers_per_block; ++j) {
            const std::string pi = "down_blocks." + std::to_string(i);
            x = R.resnet(x, pi + ".resnets." + std::to_string(j) + ".", c, tadd, eps);
            if (d.down_block_attn[i]) x = R.transformer(x, pi + ".attentions." + std::to_string(j) + ".", ctx16, heads_at(i));
            skips.push_back(x);
        }
        if (i != nb - 1) {
            x = R.conv3x3(x, "down_blocks." + std::to_string(i) + ".downsamplers.0.conv.", c, 2, 0);
            skips.push_back(x);
        }
    }
    const int cm = d.block_out_channels[nb - 1];
    x = R.resnet(x, "mid_block.resnets.0.", cm, tadd, eps);
    x = R.transformer(x, "mid_block.attentions.0.", ctx16, heads_at(nb - 1));
    x = R.resnet(x, "mid_block.resnets.1.", cm, tadd, eps);
    for (int i = 0; i < nb; ++i) {
        const int c = d.block_out_channels[nb - 1 - i];
        const bool attn = d.down_block_attn[nb - 1 - i] != 0;
        const std::string pi = "up_blocks." + std::to_string(i);
        for (int j = 0; j < d.layers_per_block + 1; ++j) {
            const Act sk = skips.back();
            skips.pop_back();
            Act cat = R.act(x.n, x.H, x.W, x.C + sk.C);
            if (R.live()) R.hip(sd_concat(x.p, x.C, sk.p, sk.C, cat.p, cat.tok(), R.st), "sd_concat");
            x = R.resnet(cat, pi + ".resnets." + std::to_string(j) + ".", c, tadd, eps);
            if (attn) x = R.transformer(x, pi + ".attentions." + std::to_string(j) + ".", ctx16, heads_at(nb - 1 - i));
        }
        if (i != nb - 1) x = R.upsample_conv(x, pi + ".upsamplers.0.conv.", c);
    }
    Act y = R.groupnorm(x, "conv_norm_out.", eps, 1, nullptr, 0, true);
    {
        float* o = nullptr;
        R.conv3x3_planes(y, "conv_out.", d.out_channels, &o);
        if (R.live()) R.hip(sd_tokens_to_nchw(o, d.out_channels, eps_out, n, d.out_channels, H, W, 1.0f, 0.0f, 0, 1, R.st), "to nchw");
    }
}

// ---- AutoencoderKL.decode(z / scaling) -> (x / 2 + 0.5).clamp(0, 1), fp32 NCHW
void vae_forward(Run& R, const float* latents, int n, int H, int W, float* images) {
    SdState* S = R.S;
    const tvc_sd_desc& d = S->d;
    const int nb = d.vae_n_blocks, L = d.latent_channels, top = d.vae_block_out_channels[nb - 1];
    float* z = (float*)R.alloc((size_t)n * L * H * W * 4);
    if (R.live())
        R.hip(sd_pointwise_small(latents, (const float*)R.W("post_quant_conv.weight"), (const float*)R.W("post_quant_conv.bias"), z,
                                 n, L, H * W, 1.0f / d.vae_scaling, R.st), "post_quant_conv");
    else { (void)R.W("post_quant_conv.weight"); (void)R.W("post_quant_conv.bias"); }
    Act x = R.act(n, H, W, top);
    {
        const size_t mark = R.off;
        uint16_t* col = (uint16_t*)R.alloc((size_t)pad_rows((int64_t)n * H * W) * 64 * 2);
        if (R.live()) R.hip(sd_im2col_in(z, col, n, L, H, W, 64, 1.0f, R.st), "im2col_in");
        R.gemm(R.W("decoder.conv_in.weight"), top, 64, col, (int64_t)n * H * W, (const float*)R.W("decoder.conv_in.bias"), x.p, top,
               TVC_EPI_BF16, (int64_t)H * W);
        R.off = mark;
    }
    const float eps = 1e-6f;
    x = R.resnet(x, "decoder.mid_block.resnets.0.", top, nullptr, eps);
    x = R.vae_attention(x, "decoder.mid_block.attentions.0.");
    x = R.resnet(x, "decoder.mid_block.resnets.1.", top, nullptr, eps);
    for (int i = 0; i < nb; ++i) {
        const int c = d.vae_block_out_channels[nb - 1 - i];
        const std::string pi = "decoder.up_blocks." + std::to_string(i);
        for (int j = 0; j < d.vae_layers_per_block + 1; ++j) x = R.resnet(x, pi + ".resnets." + std::to_string(j) + ".", c, nullptr, eps);
        if (i != nb - 1) x = R.upsample_conv(x, pi + ".upsamplers.0.conv.", c);
    }
    Act y = R.groupnorm(x, "decoder.conv_norm_out.", eps, 1, nullptr, 0, true);
    float* o = nullptr;
    R.conv3x3_planes(y, "decoder.conv_out.", 3, &o);
    if (R.live()) R.hip(sd_tokens_to_nchw(o, 3, images, y.n, 3, y.H, y.W, 0.5f, 0.5f, 1, 1, R.st), "to nchw");
}

// run `body` twice: a dry pass that sizes the arena, then the real one
template <class F>
int with_arena(tvc_handle* h, hipStream_t st, Slot slot, F&& body) {
    Run dry{h, h->sd, st, true};
    body(dry);
    if (dry.rc != TVC_OK) return dry.rc;
    int rc = ensure(h, slot, dry.high + 4096);
    if (rc) return rc;
    Run run{h, h->sd, st, false};
    run.base = (char*)h->ws[slot].p;
    run.cap = h->ws[slot].n;
    body(run);
    return run.rc;
}

// UNet2DConditionModel halves H, W (blocks - 1) times with ceil sizes and doubles them back: any size that is not a multiple
// of 2^(blocks - 1) gives skip tensors whose token counts do not match the upsampled ones
int check_latent_hw(tvc_handle* h, int H, int W, bool unet, bool vae, const char* who) {
    const int ds = unet ? 1 << (h->sd->d.n_blocks - 1) : 1;
    if (H < ds || W < ds || H % ds || W % ds)
        return fail(h, TVC_E_INVALID, std::string(who) + ": H and W must be multiples of 2^(blocks - 1) = " + std::to_string(ds));
    if (vae && ((int64_t)H * W) % 64 != 0)
        return fail(h, TVC_E_INVALID, std::string(who) + ": the VAE's attention block needs H * W % 64 == 0");
    return TVC_OK;
}

// heads of the Transformer2DModel whose state-dict prefix is `p` ("down_blocks.<i>.", "up_blocks.<i>.", "mid_block.")
int block_heads(const tvc_sd_desc& d, const std::string& p) {
    int level = d.n_blocks - 1;
    if (p.compare(0, 12, "down_blocks.") == 0 && p.size() > 12) level = p[12] - '0';
    else if (p.compare(0, 10, "up_blocks.") == 0 && p.size() > 10) level = d.n_blocks - 1 - (p[10] - '0');
    if (level < 0 || level >= d.n_blocks) level = d.n_blocks - 1;
    return d.heads_per_block[level] > 0 ? d.heads_per_block[level] : d.heads;
}

int need_sd(tvc_handle* h, bool unet, bool vae, const char* who) {
    if (!h) return TVC_E_INVALID;
    if (!h->sd || (unet && !h->sd->has_unet) || (vae && !h->sd->has_vae))
        return fail(h, TVC_E_STATE, std::string(who) + ": call tvc_sd_load first (with the " + (unet ? "UNet" : "VAE") + " tensors)");
    return TVC_OK;
}

}  // namespace

extern "C" {

int tvc_sd_load(tvc_handle* h, const tvc_sd_desc* desc, const tvc_named_tensor* tensors, int32_t n_tensors, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!desc || !tensors || n_tensors <= 0) return fail(h, TVC_E_INVALID, "tvc_sd_load: NULL desc / tensors");
    const tvc_sd_desc& d = *desc;
    if (d.n_blocks < 1 || d.n_blocks > 4 || d.vae_n_blocks < 1 || d.vae_n_blocks > 4 || d.norm_groups < 1 || d.norm_groups > 32 ||
        d.heads < 1 || d.prediction_type < 0 || d.prediction_type > 1 || d.in_channels * 9 > 64 || d.latent_channels * 9 > 64 || d.latent_channels > 8 || d.ctx < 1 ||
        d.cross_attention_dim % 64 != 0 || d.layers_per_block < 1 || d.vae_layers_per_block < 1 || d.out_channels < 1)
        return fail(h, TVC_E_INVALID, "tvc_sd_load: unsupported geometry");
    for (int i = 0; i < d.n_blocks; ++i) {
        const int c = d.block_out_channels[i];
        const int nh = d.heads_per_block[i] > 0 ? d.heads_per_block[i] : d.heads;
        const int dh = c % nh == 0 ? c / nh : 0;
        const bool dh_ok = dh == 8 || dh == 16 || dh == 24 || dh == 32 || dh == 40 || dh == 48 || dh == 56 || dh == 64 || dh == 80 ||
                           dh == 96 || dh == 128 || dh == 160;         // the instantiations of sd_flash_attention_kernel
        if (c % 64 != 0 || c % d.norm_groups != 0 || (c / d.norm_groups) % 2 != 0 || !dh_ok || c > 1536)
            return fail(h, TVC_E_INVALID, "tvc_sd_load: UNet widths must be multiples of 64 with head_dim in {8..64 step 8, 80, 96, 128, 160}");
    }
    for (int i = 0; i < d.vae_n_blocks; ++i) {
        const int c = d.vae_block_out_channels[i];
        if (c % 64 != 0 || c % d.norm_groups != 0 || (c / d.norm_groups) % 2 != 0)
            return fail(h, TVC_E_INVALID, "tvc_sd_load: VAE widths must be multiples of 64");
    }
    tvc_sd_free(h);
    SdState* S = new SdState();
    S->d = d;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || !tensors[i].ptr) { delete S; return fail(h, TVC_E_INVALID, "tvc_sd_load: NULL tensor entry"); }
        S->w[tensors[i].name] = tensors[i].ptr;
    }
    S->has_unet = S->w.count("conv_in.weight") != 0;
    S->has_vae = S->w.count("decoder.conv_in.weight") != 0;
    h->sd = S;
    hipStream_t st = (hipStream_t)stream;
    if (S->has_unet) {
        // gather every resnet's time projection [Cout, time_dim] into one matrix (one GEMM per evaluation)
        const int Tdim = 4 * d.block_out_channels[0];
        std::vector<std::pair<std::string, int>> order;
        const std::string suffix = "time_emb_proj.weight";
        for (auto& kv : S->w)
            if (kv.first.size() > suffix.size() && kv.first.compare(kv.first.size() - suffix.size(), suffix.size(), suffix) == 0)
                order.push_back({kv.first.substr(0, kv.first.size() - suffix.size()), 0});
        std::sort(order.begin(), order.end());
        // widths: the resnet's conv2 is [Cout, 9 * Cout]; the host tells us Cout through "<prefix>time_emb_proj.rows"? No --
        // derive it from the architecture walk instead.
        std::unordered_map<std::string, int> width;
        {
            const int nb = d.n_blocks;
            for (int i = 0; i < nb; ++i)
                for (int j = 0; j < d.layers_per_block; ++j)
                    width["down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + "."] = d.block_out_channels[i];
            width["mid_block.resnets.0."] = width["mid_block.resnets.1."] = d.block_out_channels[nb - 1];
            for (int i = 0; i < nb; ++i)
                for (int j = 0; j < d.layers_per_block + 1; ++j)
                    width["up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + "."] = d.block_out_channels[nb - 1 - i];
        }
        int total = 0;
        for (auto& e : order) {
            auto it = width.find(e.first);
            if (it == width.end()) { tvc_sd_free(h); return fail(h, TVC_E_INVALID, "tvc_sd_load: unexpected resnet " + e.first); }
            e.second = it->second;
            S->temb_off[e.first] = total;
            total += e.second;
        }
        if ((int)order.size() != (int)width.size()) { tvc_sd_free(h); return fail(h, TVC_E_STATE, "tvc_sd_load: a resnet's time_emb_proj is missing"); }
        S->temb_total = total;
        HIP_TRY(hipMalloc(&S->temb_w, (size_t)(total + 512) * Tdim * 2));
        HIP_TRY(hipMalloc(&S->temb_b, (size_t)total * 4));
        for (auto& e : order) {
            const int o = S->temb_off[e.first];
            auto bi = S->w.find(e.first + "time_emb_proj.bias");
            if (bi == S->w.end()) { tvc_sd_free(h); return fail(h, TVC_E_STATE, "tvc_sd_load: missing " + e.first + "time_emb_proj.bias"); }
            HIP_TRY(hipMemcpyAsync((char*)S->temb_w + (size_t)o * Tdim * 2, S->w[e.first + suffix], (size_t)e.second * Tdim * 2,
                                   hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync((char*)S->temb_b + (size_t)o * 4, bi->second, (size_t)e.second * 4, hipMemcpyDeviceToDevice, st));
        }
    }
    // scheduler table: betas = linspace(sqrt(b0), sqrt(b1), T)^2 in fp32, alphas_cumprod = cumprod(1 - betas)
    {
        const int T = d.num_train_timesteps > 0 ? d.num_train_timesteps : 1000;
        S->alphas_cumprod.resize(T);
        const float s0 = sqrtf(d.beta_start), s1 = sqrtf(d.beta_end);
        const float step = (s1 - s0) / (float)(T - 1);
        float prod = 1.0f;
        for (int i = 0; i < T; ++i) {
            const float r = i < T / 2 ? s0 + step * (float)i : s1 - step * (float)(T - 1 - i);
            prod *= 1.0f - r * r;
            S->alphas_cumprod[i] = prod;
        }
    }
    return TVC_OK;
}

int tvc_sd_unet(tvc_handle* h, const float* latents_dev, int32_t n, int32_t H, int32_t W, float timestep, const float* ctx_dev,
                float* eps_dev, void* stream) {
    int rc = need_sd(h, true, false, "tvc_sd_unet");
    if (rc) return rc;
    if (n < 1 || !latents_dev || !ctx_dev || !eps_dev) return fail(h, TVC_E_INVALID, "tvc_sd_unet: need n >= 1, non-NULL buffers");
    if ((rc = check_latent_hw(h, H, W, true, false, "tvc_sd_unet"))) return rc;
    return with_arena(h, (hipStream_t)stream, WS_SD0,
                      [&](Run& R) { unet_forward(R, latents_dev, n, H, W, timestep, ctx_dev, eps_dev); });
}

int tvc_sd_vae_decode(tvc_handle* h, const float* latents_dev, int32_t n, int32_t H, int32_t W, float* images_dev, void* stream) {
    int rc = need_sd(h, false, true, "tvc_sd_vae_decode");
    if (rc) return rc;
    if (n < 1 || H < 1 || W < 1 || !latents_dev || !images_dev) return fail(h, TVC_E_INVALID, "tvc_sd_vae_decode: bad arguments");
    if ((rc = check_latent_hw(h, H, W, false, true, "tvc_sd_vae_decode"))) return rc;
    const tvc_sd_desc& d = h->sd->d;
    const int up = 1 << (d.vae_n_blocks - 1);
    // images in chunks that keep the im2col rows of the full-resolution layers within a few GB
    const size_t per_image = (size_t)H * up * W * up * 9 * d.vae_block_out_channels[d.vae_n_blocks > 1 ? 1 : 0] * 2;
    int chunk = (int)(((size_t)6 << 30) / (per_image ? per_image : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = n - i0 < chunk ? n - i0 : chunk;
        rc = with_arena(h, (hipStream_t)stream, WS_SD1, [&](Run& R) {
            vae_forward(R, latents_dev + (size_t)i0 * d.latent_channels * H * W, m, H, W,
                        images_dev + (size_t)i0 * 3 * H * up * W * up);
        });
        if (rc) return rc;
    }
    return TVC_OK;
}

// the sampling loop for m images whose conditioning / latents start at the given pointers
static int sd_generate_chunk(tvc_handle* h, const float* cond_dev, const float* uncond_dev, float* latents_dev, int n, int H, int W,
                             int steps, float guidance, hipStream_t st) {
    SdState* S = h->sd;
    const tvc_sd_desc& d = S->d;
    const int T = (int)S->alphas_cumprod.size();
    int rc;
    const int64_t ne = (int64_t)n * d.in_channels * H * W;          // elements of the latents
    const size_t ctx_elems = (size_t)n * d.ctx * d.cross_attention_dim;
    // [uncond | cond] text states, doubled latents, eps of both halves, 4 history slots, the saved sample of the 2nd step
    if ((rc = ensure(h, WS_SD2, (2 * ctx_elems + 2 * ne + 2 * ne + 4 * ne + ne + ne) * 4))) return rc;
    float* ctx2 = (float*)h->ws[WS_SD2].p;
    float* lat2 = ctx2 + 2 * ctx_elems;
    float* eps2 = lat2 + 2 * ne;
    float* ets = eps2 + 2 * ne;             // 4 slots
    float* cur = ets + 4 * ne;
    float* tmp = cur + ne;
    HIP_TRY(hipMemcpyAsync(ctx2, uncond_dev, ctx_elems * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx2 + ctx_elems, cond_dev, ctx_elems * 4, hipMemcpyDeviceToDevice, st));
    // PNDMScheduler.set_timesteps with skip_prk_steps: the second-to-last value is visited twice
    const int ratio = T / steps;
    std::vector<int> ts;
    for (int i = 0; i < steps; ++i) ts.push_back(i * ratio + d.steps_offset);
    std::vector<int> plms(ts.begin(), ts.end() - 1);
    plms.push_back(ts[steps - 2]);
    plms.push_back(ts[steps - 1]);
    std::vector<int> order(plms.rbegin(), plms.rend());
    auto acp = [&](int t) { return t >= 0 ? S->alphas_cumprod[t < T ? t : T - 1] : S->alphas_cumprod[0]; };
    int counter = 0, n_ets = 0, head = 0;            // ets ring: slot (head - 1 - k) mod 4 = k-th newest
    auto slot = [&](int k) { return ets + (size_t)((head - 1 - k + 8) % 4) * ne; };
    // the step-invariant tensors (bf16 text states, every cross-attention's K / V of them) are computed by the first
    // evaluation and kept for the others: 16 projections per evaluation less, bit-identical
    // Two streams (TVC_OPT_SD_STREAMS): the unconditional and the conditional half of an evaluation are independent until
    // the guidance step, so each runs on its own stream in its own half of the arena (and of the step-invariant region).
    // A half's launches have half the tiles -- at 12 images the 32 x 32 level's 327-tile convolutions were two tile rounds
    // for 1.28 rounds of work --, and whatever compute units one half's partial round leaves idle take the other half's
    // next launch.  Per-sample arithmetic does not depend on the batch (Run::fixed_split), so the images do not change.
    const int parts = h->sd_streams >= 2 ? 2 : 1;
    const int pn = 2 * n / parts;                     // samples per part
    if (parts == 2 && !h->sd_aux) {
        HIP_TRY(hipStreamCreateWithFlags(&h->sd_aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->sd_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->sd_join, hipEventDisableTiming));
    }
    size_t keep_part = 0, arena_part[2] = {0, 0};     // arena: [0] the loop's first evaluation (fills the kept tensors), [1] the others
    for (int pass = 0; pass < 2; ++pass) {
        Run dry{h, S, st, true};
        dry.keep = pass == 0 ? 1 : 2;
        unet_forward(dry, nullptr, pn, H, W, 0.f, nullptr, nullptr);
        if (dry.rc != TVC_OK) return dry.rc;
        if (pass == 0) keep_part = (dry.keep_high + 4096 + 255) & ~(size_t)255;
        arena_part[pass] = (dry.high + 4096 + 255) & ~(size_t)255;
    }
    if ((rc = ensure(h, WS_SD3, keep_part * parts))) return rc;
    if ((rc = ensure(h, WS_SD0, std::max(arena_part[0], arena_part[1]) * parts))) return rc;
    bool first = true;
    for (int t : order) {
        HIP_TRY(hipMemcpyAsync(lat2, latents_dev, ne * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(lat2 + ne, latents_dev, ne * 4, hipMemcpyDeviceToDevice, st));
        const size_t ap = arena_part[first ? 0 : 1];
        if (parts == 2) {
            HIP_TRY(hipEventRecord(h->sd_fork, st));
            HIP_TRY(hipStreamWaitEvent(h->sd_aux, h->sd_fork, 0));
        }
        rc = TVC_OK;
        for (int part = 0; part < parts; ++part) {
            Run run{h, S, part == 0 ? st : h->sd_aux, false};
            run.base = (char*)h->ws[WS_SD0].p + (size_t)part * ap;
            run.cap = ap;
            run.keep = first ? 1 : 2;
            run.keep_base = (char*)h->ws[WS_SD3].p + (size_t)part * keep_part;
            run.keep_cap = keep_part;
            const size_t so = (size_t)part * pn;          // first sample of this part
            unet_forward(run, lat2 + so * (ne / n), pn, H, W, (float)t, ctx2 + so * (ctx_elems / n), eps2 + so * (ne / n));
            if (run.rc != TVC_OK && rc == TVC_OK) rc = run.rc;
        }
        if (parts == 2) {          // joined even after an error: the caller's stream must not run ahead of the other half
            HIP_TRY(hipEventRecord(h->sd_join, h->sd_aux));
            HIP_TRY(hipStreamWaitEvent(st, h->sd_join, 0));
        }
        if (rc) return rc;
        first = false;
        // classifier-free guidance, then PNDMScheduler.step_plms
        int prev_t = t - ratio, tt = t;
        float* e_new = tmp;
        if (counter != 1) {
            e_new = ets + (size_t)head * ne;         // append (the ring drops the oldest of 4)
            head = (head + 1) % 4;
            if (n_ets < 4) ++n_ets;
        } else {
            prev_t = t; tt = t + ratio;
        }
        HIP_TRY(sd_cfg(eps2, e_new, ne, guidance, st));
        const float* sample = latents_dev;
        const float *e0 = nullptr, *e1 = nullptr, *e2 = nullptr, *e3 = nullptr;
        float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (n_ets == 1 && counter == 0) {
            e0 = slot(0); c0 = 1.f;
            HIP_TRY(hipMemcpyAsync(cur, latents_dev, ne * 4, hipMemcpyDeviceToDevice, st));
        } else if (n_ets == 1 && counter == 1) {
            e0 = e_new; c0 = 0.5f; e1 = slot(0); c1 = 0.5f;
            sample = cur;
        } else if (n_ets == 2) {
            e0 = slot(0); c0 = 1.5f; e1 = slot(1); c1 = -0.5f;
        } else if (n_ets == 3) {
            e0 = slot(0); c0 = 23.f / 12.f; e1 = slot(1); c1 = -16.f / 12.f; e2 = slot(2); c2 = 5.f / 12.f;
        } else {
            e0 = slot(0); c0 = 55.f / 24.f; e1 = slot(1); c1 = -59.f / 24.f; e2 = slot(2); c2 = 37.f / 24.f; e3 = slot(3); c3 = -9.f / 24.f;
        }
        const float a_t = acp(tt), a_prev = acp(prev_t);
        const float b_t = 1.f - a_t, b_prev = 1.f - a_prev;
        float cs = sqrtf(a_prev / a_t);
        const float denom = a_t * sqrtf(b_prev) + sqrtf(a_t * b_t * a_prev);
        float ce = (a_prev - a_t) / denom;
        if (d.prediction_type == 1) {
            // v-prediction (PNDMScheduler._get_prev_sample): the combined model output E stands for
            // sqrt(a_t) E + sqrt(b_t) sample -- folded into the two coefficients of prev = cs * sample - ce * E
            cs -= ce * sqrtf(b_t);
            ce *= sqrtf(a_t);
        }
        HIP_TRY(sd_lincomb(latents_dev, sample, cs, ce, e0, c0, e1, c1, e2, c2, e3, c3, ne, st));
        ++counter;
    }
    return TVC_OK;
}

int tvc_sd_generate(tvc_handle* h, const float* cond_dev, const float* uncond_dev, float* latents_dev, int32_t n, int32_t H,
                    int32_t W, int32_t steps, float guidance, float* images_dev, void* stream) {
    int rc = need_sd(h, true, images_dev != nullptr, "tvc_sd_generate");
    if (rc) return rc;
    SdState* S = h->sd;
    const tvc_sd_desc& d = S->d;
    const int T = (int)S->alphas_cumprod.size();
    if (n < 1 || steps < 2 || steps > T || !cond_dev || !uncond_dev || !latents_dev)
        return fail(h, TVC_E_INVALID, "tvc_sd_generate: need n >= 1, 2 <= steps <= num_train_timesteps, non-NULL buffers");
    if ((rc = check_latent_hw(h, H, W, true, images_dev != nullptr, "tvc_sd_generate"))) return rc;
    hipStream_t st = (hipStream_t)stream;
    // Samples per pass: the activations of ONE UNet evaluation live in one arena (nothing is recomputed), whose size is
    // linear in the sample count -- a dry run of one image's (unconditional | conditional) pair measures it, and the batch
    // runs in chunks that keep the arena within TVC_OPT_SD_ARENA_BYTES.  A chunk is a whole sampling loop of its images
    // (every sample's arithmetic is independent of its batch mates), so chunking changes no image.
    Run dry{h, S, st, true};
    unet_forward(dry, nullptr, 2, H, W, 0.f, nullptr, nullptr);
    if (dry.rc != TVC_OK) return dry.rc;
    const size_t per_image = dry.high;
    int64_t chunk = (int64_t)(h->sd_arena_bytes / (per_image ? per_image : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    const size_t lat_per = (size_t)d.in_channels * H * W, ctx_per = (size_t)d.ctx * d.cross_attention_dim;
    for (int i0 = 0; i0 < n; i0 += (int)chunk) {
        const int m = n - i0 < chunk ? n - i0 : (int)chunk;
        rc = sd_generate_chunk(h, cond_dev + (size_t)i0 * ctx_per, uncond_dev + (size_t)i0 * ctx_per, latents_dev + (size_t)i0 * lat_per,
                               m, H, W, steps, guidance, st);
        if (rc) return rc;
    }
    if (images_dev) return tvc_sd_vae_decode(h, latents_dev, n, H, W, images_dev, stream);
    return TVC_OK;
}

int tvc_sd_block(tvc_handle* h, int32_t kind, const char* prefix, const float* x_dev, int32_t n, int32_t Cin, int32_t H, int32_t W,
                 const float* temb_dev, const float* ctx_dev, int32_t Cout, int32_t vae, float* out_dev, void* stream) {
    int rc = need_sd(h, false, false, "tvc_sd_block");
    if (rc) return rc;
    if (kind < 0 || kind > 5 || !prefix || !x_dev || !out_dev || n < 1 || Cin < 8 || Cin % 8 || Cout < 1 || H < 1 || W < 1)
        return fail(h, TVC_E_INVALID, "tvc_sd_block: bad arguments");
    if (kind == 1 && !ctx_dev) return fail(h, TVC_E_INVALID, "tvc_sd_block: a transformer block needs ctx");
    SdState* S = h->sd;
    const tvc_sd_desc& d = S->d;
    const std::string p(prefix);
    return with_arena(h, (hipStream_t)stream, WS_SD0, [&](Run& R) {
        Act x = R.act(n, H, W, Cin);
        if (R.live()) R.hip(sd_nchw_to_tokens(x_dev, x.p, n, Cin, H * W, R.st), "nchw_to_tokens");
        Act y;
        if (kind == 0) {
            // the block's own time projection: tadd[n, Cout] = silu(temb) W^T + b, laid out as one slice of temb_total
            const float* tadd_all = nullptr;
            if (temb_dev) {
                const int Tdim = 4 * d.block_out_channels[0];
                uint16_t* tb = (uint16_t*)R.alloc((size_t)pad_rows(n) * Tdim * 2);
                float* tadd = (float*)R.alloc((size_t)n * S->temb_total * 4);
                if (R.live()) R.hip(sd_cast_silu(temb_dev, tb, (int64_t)n * Tdim, 1, R.st), "silu");
                R.gemm(S->temb_w, S->temb_total, Tdim, tb, n, (const float*)S->temb_b, tadd, S->temb_total, TVC_EPI_F32, 1);
                tadd_all = tadd;
            }
            y = R.resnet(x, p, Cout, tadd_all, vae ? 1e-6f : d.norm_eps);
        } else if (kind == 1) {
            uint16_t* ctx16 = (uint16_t*)R.alloc((size_t)pad_rows((int64_t)n * d.ctx) * d.cross_attention_dim * 2);
            if (R.live()) R.hip(sd_cast_silu(ctx_dev, ctx16, (int64_t)n * d.ctx * d.cross_attention_dim, 0, R.st), "ctx cast");
            y = R.transformer(x, p, ctx16, block_heads(d, p));
        } else if (kind == 2) {
            y = R.vae_attention(x, p);
        } else if (kind == 3) {
            // stride 1: the 9-plane GEMM on the padded layout (dense -> padded with zero borders -> conv -> dense)
            Act xp = R.act_padded(n, H, W, Cin);
            if (R.live()) R.hip(sd_relayout(x.p, xp.p, n, H, W, Cin, 0, 1, 0, R.st), "relayout");
            Act yp = R.conv3x3_planes(xp, p, Cout);
            y = R.act(n, H, W, Cout);
            if (R.live()) R.hip(sd_relayout(yp.p, y.p, n, H, W, Cout, 1, 0, 0, R.st), "relayout");
        } else {
            y = kind == 5 ? R.upsample_conv(x, p, Cout) : R.conv3x3(x, p, Cout, 2, 0);
        }
        if (R.live()) R.hip(sd_tokens_bf16_to_nchw(y.p, out_dev, y.n, y.C, y.H * y.W, R.st), "tokens_to_nchw");
    });
}

int tvc_preprocess_images(tvc_handle* h, const float* images_dev, int32_t n, int32_t H, int32_t W, int32_t S, int32_t filter,
                          int32_t keep_aspect, const float* mean3, const float* std3, float* out_dev, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (n < 0 || H < 1 || W < 1 || S < 1 || !mean3 || !std3 || (n > 0 && (!images_dev || !out_dev)) || filter < 0 || filter > 1)
        return fail(h, TVC_E_INVALID, "tvc_preprocess_images: bad arguments");
    int Hr = S, Wr = S;
    if (keep_aspect) {               // short side -> S (rounded as PIL does), centre crop
        if (H <= W) Wr = (int)((double)W * S / H + 0.5); else Hr = (int)((double)H * S / W + 0.5);
        if (Hr < S) Hr = S;
        if (Wr < S) Wr = S;
    }
    HIP_TRY(sd_resize_norm(images_dev, out_dev, n, H, W, Hr, Wr, (Hr - S) / 2, (Wr - S) / 2, S, filter, mean3, std3, (hipStream_t)stream));
    return TVC_OK;
}

int tvc_sd_attention(tvc_handle* h, const uint16_t* q_dev, const uint16_t* k_dev, const uint16_t* v_dev, uint16_t* out_dev, int32_t n,
                     int32_t heads, int32_t Tq, int32_t Tk, int32_t dh, void* stream) {
    if (!h) return TVC_E_INVALID;
    if (!q_dev || !k_dev || !v_dev || !out_dev || n < 1 || heads < 1 || Tq < 1 || Tk < 1 || dh < 8 || dh > 160 || dh % 8)
        return fail(h, TVC_E_INVALID, "tvc_sd_attention: need head_dim % 8 == 0 and <= 160, non-NULL buffers");
    const int64_t ld = (int64_t)heads * dh;
    HIP_TRY(sd_flash_attention(q_dev, ld, k_dev, ld, v_dev, ld, out_dev, ld, n, heads, Tq, Tk, dh, (hipStream_t)stream));
    return TVC_OK;
}

}  // extern "C"
