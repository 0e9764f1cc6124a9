// Host-only sanitizer driver of the C-ABI (tests/test_abi_and_host.py::test_host_code_under_address_and_ub_sanitizers):
// g++ -fsanitize=address,undefined builds tvc_abi.cpp, tvc_precise.cpp, tvc_split.cpp and tvc_sd.cpp against the HIP
// stand-in of this directory (kernels = no-ops, "device" memory = host blocks with known sizes) and walks their host
// logic: descriptor validation, workspace sizing, weight maps, the arena's dry / real passes, prefix-string dispatch,
// option handling, chunking, the error paths.  argv[1]: file of SD tensor names (one "name rows cols dtype" per line).
#include "../../include/tvc.h"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#define CHECK(cond)                                                                              \
    do {                                                                                         \
        if (!(cond)) { fprintf(stderr, "driver.cpp:%d: CHECK failed: %s\n", __LINE__, #cond); return 1; } \
    } while (0)
#define OK(call)                                                                                           \
    do {                                                                                                   \
        int rc__ = (call);                                                                                 \
        if (rc__ != TVC_OK) { fprintf(stderr, "driver.cpp:%d: %s -> %d (%s)\n", __LINE__, #call, rc__, tvc_last_error(h)); return 1; } \
    } while (0)

static void* dev(size_t bytes) { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) abort(); return p; }

int main(int argc, char** argv) {
    CHECK(argc >= 2);
    tvc_handle* h = nullptr;
    // ---- a two-tower handle at the toy geometry (ViT-T/16-test): widths 256 / 128, 2 layers
    tvc_model_desc m{};
    m.image_size = 64; m.patch = 16; m.vocab = 49408; m.ctx = 77; m.embed_dim = 128;
    m.vision = {256, 2, 4, 512}; m.text = {128, 2, 2, 256};
    std::vector<void*> keep;
    auto buf = [&](size_t elems, size_t es) { void* p = dev(elems * es + 512 * 1024); keep.push_back(p); return p; };
    auto layers16 = [&](const tvc_tower_arch& a) {
        std::vector<tvc_layer_weights> L(a.layers);
        for (auto& l : L) {
            l.ln1_g = (float*)buf(a.width, 4); l.ln1_b = (float*)buf(a.width, 4); l.ln2_g = (float*)buf(a.width, 4); l.ln2_b = (float*)buf(a.width, 4);
            l.wqkv = (uint16_t*)buf((size_t)3 * a.width * a.width, 2); l.bqkv = (float*)buf(3 * a.width, 4);
            l.wo = (uint16_t*)buf((size_t)a.width * a.width, 2); l.bo = (float*)buf(a.width, 4);
            l.w1 = (uint16_t*)buf((size_t)a.mlp * a.width, 2); l.b1 = (float*)buf(a.mlp, 4);
            l.w2 = (uint16_t*)buf((size_t)a.width * a.mlp, 2); l.b2 = (float*)buf(a.width, 4);
        }
        return L;
    };
    auto layers32 = [&](const tvc_tower_arch& a) {
        std::vector<tvc_layer_weights_f32> L(a.layers);
        for (auto& l : L) {
            l.ln1_g = (float*)buf(a.width, 4); l.ln1_b = (float*)buf(a.width, 4); l.ln2_g = (float*)buf(a.width, 4); l.ln2_b = (float*)buf(a.width, 4);
            l.wqkv = (float*)buf((size_t)3 * a.width * a.width, 4); l.bqkv = (float*)buf(3 * a.width, 4);
            l.wo = (float*)buf((size_t)a.width * a.width, 4); l.bo = (float*)buf(a.width, 4);
            l.w1 = (float*)buf((size_t)a.mlp * a.width, 4); l.b1 = (float*)buf(a.mlp, 4);
            l.w2 = (float*)buf((size_t)a.width * a.mlp, 4); l.b2 = (float*)buf(a.width, 4);
        }
        return L;
    };
    auto vl = layers16(m.vision), tl = layers16(m.text);
    tvc_vision_weights vw{};
    vw.patch_w = (uint16_t*)buf((size_t)256 * 768, 2); vw.cls = (float*)buf(256, 4); vw.pos = (float*)buf(17 * 256, 4);
    vw.ln_pre_g = (float*)buf(256, 4); vw.ln_pre_b = (float*)buf(256, 4); vw.ln_post_g = (float*)buf(256, 4); vw.ln_post_b = (float*)buf(256, 4);
    vw.proj = (uint16_t*)buf(128 * 256, 2); vw.layers = vl.data();
    tvc_text_weights tw{};
    tw.tok_emb = (float*)buf((size_t)49408 * 128, 4); tw.pos = (float*)buf(77 * 128, 4); tw.ln_final_g = (float*)buf(128, 4);
    tw.ln_final_b = (float*)buf(128, 4); tw.proj = (uint16_t*)buf(128 * 128, 2); tw.layers = tl.data();
    // bad descriptors are refused before anything is allocated
    {
        tvc_model_desc bad = m; bad.vision.heads = 3;
        CHECK(tvc_create(&bad, &vw, &tw, &h) == TVC_E_INVALID && h == nullptr && strlen(tvc_last_error(nullptr)) > 0);
        bad = m; bad.image_size = 1024;                       // 4097 tokens
        CHECK(tvc_create(&bad, &vw, &tw, &h) == TVC_E_INVALID);
        CHECK(tvc_create(nullptr, &vw, nullptr, &h) == TVC_E_INVALID);
    }
    CHECK(tvc_create(&m, &vw, &tw, &h) == TVC_OK && h);
    CHECK(tvc_abi_version() == TVC_ABI_VERSION);
    const int B = 5, N = 3;
    float* pix = (float*)buf((size_t)B * 3 * 64 * 64, 4);
    int32_t* tok = (int32_t*)buf((size_t)B * (N + 1) * 77, 4);
    float* fi = (float*)buf(B * 128, 4); float* ft = (float*)buf(B * (N + 1) * 128, 4);
    float* hid = (float*)buf((size_t)B * 77 * 128, 4);
    // ---- options: ranges, unknown ids, modes that need fp32 weights
    CHECK(tvc_set_option(h, 99, 1) == TVC_E_INVALID);
    CHECK(tvc_set_option(h, TVC_OPT_TOWER_PRECISION, 1) == TVC_E_STATE && tvc_set_option(h, TVC_OPT_TOWER_PRECISION, 2) == TVC_E_STATE);
    CHECK(tvc_set_option(h, TVC_OPT_TOWER_PRECISION, 3) == TVC_E_INVALID && tvc_set_option(h, TVC_OPT_MAX_CHUNK_IMAGES, 0) == TVC_E_INVALID);
    CHECK(tvc_set_option(h, TVC_OPT_SD_ARENA_BYTES, 1) == TVC_E_INVALID && tvc_set_option(h, TVC_OPT_TEXT_GROUP, -1) == TVC_E_INVALID);
    // ---- bf16 towers: dense, packed, grouped, chunked, pooled on / off
    for (int pooled = 0; pooled < 2; ++pooled)
        for (int pack = 0; pack < 2; ++pack) {
            OK(tvc_set_option(h, TVC_OPT_POOLED_LAST_LAYER, pooled));
            OK(tvc_set_option(h, TVC_OPT_TEXT_PACKING, pack));
            OK(tvc_set_option(h, TVC_OPT_TEXT_GROUP, pack ? N + 1 : 0));
            OK(tvc_encode_image(h, pix, B, fi, 1, nullptr));
            OK(tvc_encode_text(h, tok, B * (N + 1), ft, 1, nullptr));
        }
    OK(tvc_set_option(h, TVC_OPT_MAX_CHUNK_IMAGES, 2)); OK(tvc_set_option(h, TVC_OPT_MAX_CHUNK_TEXTS, 7));
    OK(tvc_encode_image(h, pix, B, fi, 0, nullptr)); OK(tvc_encode_text(h, tok, B * (N + 1), ft, 0, nullptr));
    OK(tvc_encode_text_hidden(h, tok, B, hid, nullptr));
    OK(tvc_set_option(h, TVC_OPT_MAX_CHUNK_IMAGES, 512)); OK(tvc_set_option(h, TVC_OPT_MAX_CHUNK_TEXTS, 4608));
    CHECK(tvc_encode_image(h, nullptr, 3, fi, 1, nullptr) == TVC_E_INVALID && tvc_encode_image(h, pix, 0, fi, 1, nullptr) == TVC_OK);
    // ---- input gradient path
    float* gout = (float*)buf(B * 128, 4); float* gpix = (float*)buf((size_t)B * 3 * 64 * 64, 4);
    OK(tvc_encode_image_grad(h, pix, B, fi, 1, nullptr)); OK(tvc_encode_image_backward(h, gout, gpix, nullptr));
    // ---- fp32-grade and split-bf16 modes
    auto vl32 = layers32(m.vision), tl32 = layers32(m.text);
    tvc_vision_weights_f32 v32{}; tvc_text_weights_f32 t32{};
    v32.patch_w = (float*)buf((size_t)256 * 768, 4); v32.cls = vw.cls; v32.pos = vw.pos; v32.ln_pre_g = vw.ln_pre_g; v32.ln_pre_b = vw.ln_pre_b;
    v32.ln_post_g = vw.ln_post_g; v32.ln_post_b = vw.ln_post_b; v32.proj = (float*)buf(128 * 256, 4); v32.layers = vl32.data();
    t32.tok_emb = tw.tok_emb; t32.pos = tw.pos; t32.ln_final_g = tw.ln_final_g; t32.ln_final_b = tw.ln_final_b;
    t32.proj = (float*)buf(128 * 128, 4); t32.layers = tl32.data();
    OK(tvc_set_weights_f32(h, &v32, &t32));
    for (int mode = 1; mode <= 2; ++mode) {
        OK(tvc_set_option(h, TVC_OPT_TOWER_PRECISION, mode));
        OK(tvc_encode_image(h, pix, B, fi, 1, nullptr)); OK(tvc_encode_text(h, tok, B * (N + 1), ft, 1, nullptr));
        OK(tvc_encode_text_hidden(h, tok, B, hid, nullptr));
    }
    OK(tvc_set_weights_f32(h, &v32, nullptr));                  // re-registering in mode 2 rebuilds the planes
    OK(tvc_encode_image(h, pix, B, fi, 1, nullptr));
    OK(tvc_set_option(h, TVC_OPT_TOWER_PRECISION, 0));
    // ---- banks: slots, search shapes (skinny and tiled forms), gather, merge, cosine matrix
    const int R = 5000, D = 128;
    uint16_t* bank = (uint16_t*)buf((size_t)R * D, 2); float* bank32 = (float*)buf((size_t)R * D, 4);
    CHECK(tvc_bank_select(h, TVC_MAX_BANKS) == TVC_E_INVALID);
    int32_t* idx = (int32_t*)buf(400 * 128, 4); float* sim = (float*)buf(400 * 128, 4); float* mom = (float*)buf(400 * 4, 4);
    float* rows = (float*)buf(400 * D, 4);
    CHECK(tvc_bank_search(h, rows, 4, 5, 0.3f, 0, idx, sim, mom, nullptr) == TVC_E_STATE);
    OK(tvc_bank_select(h, 3)); OK(tvc_bank_set(h, bank, R, D, TVC_DTYPE_BF16, nullptr));
    OK(tvc_bank_select(h, 0)); OK(tvc_bank_set(h, bank32, R, D, TVC_DTYPE_F32, nullptr));
    for (int slot : {0, 3})
        for (int M : {1, 48, 64, 65, 300}) {
            OK(tvc_bank_select(h, slot));
            OK(tvc_bank_search(h, rows, M, 10, 0.3f, 100, idx, sim, (M & 1) ? mom : nullptr, nullptr));
            OK(tvc_bank_search_dense(h, rows, M, 10, 0.3f, 100, idx, sim, mom, nullptr));
            (void)tvc_bank_status(h, nullptr);                  // reads a flag no kernel wrote here: either answer is fine
        }
    CHECK(tvc_bank_search(h, rows, 4, 129, 0.3f, 0, idx, sim, nullptr, nullptr) == TVC_E_INVALID);
    float* feat = (float*)buf((size_t)300 * 5 * D, 4);
    OK(tvc_bank_gather(h, idx, 300 * 5, 100, feat, nullptr));
    OK(tvc_bank_set(h, bank, 0, D, TVC_DTYPE_BF16, nullptr)); OK(tvc_bank_search(h, rows, 4, 5, 0.3f, 0, idx, sim, mom, nullptr));
    float* featp = (float*)buf((size_t)4 * 50 * 5 * D, 4);
    OK(tvc_topk_merge(h, idx, sim, featp, nullptr, 4, 50, 10, 5, D, idx, sim, feat, nullptr, nullptr));
    CHECK(tvc_topk_merge(h, idx, sim, featp, nullptr, 40, 50, 10, 5, D, idx, sim, feat, nullptr, nullptr) == TVC_E_INVALID);
    float* cm = (float*)buf(300 * 300, 4);
    OK(tvc_cosine_matrix(h, rows, 300, rows, 37, D, cm, nullptr));
    // ---- consistency records
    tvc_consistency_params cp{5, 0.3f, 10, 0.95f, 0.4f, 0.2f, {0.25f, 0.25f, 0.25f, 0.25f}};
    float* rec = (float*)buf((size_t)B * tvc_rec_stride(N), 4);
    OK(tvc_consistency(h, fi, ft, B, N, 128, idx, sim, feat, 10, 5, &cp, rec, nullptr));
    OK(tvc_consistency(h, fi, ft, B, N, 128, nullptr, nullptr, nullptr, 0, 0, &cp, rec, nullptr));
    // ---- profiling brackets
    double ms[TVC_PROF_NCAT], work[TVC_PROF_NCAT], big[3]; int64_t launches[TVC_PROF_NCAT];
    OK(tvc_profile_begin(h)); OK(tvc_encode_image(h, pix, B, fi, 1, nullptr)); OK(tvc_profile_end(h, ms, work, launches, big));
    CHECK(launches[TVC_PROF_GEMM] > 0 && work[TVC_PROF_GEMM] > 0);

    // ---- latent-diffusion model: the tensor map, both arenas, every block kind, chunked generation, the error paths
    std::vector<std::string> names; std::vector<tvc_named_tensor> nt;
    {
        std::ifstream f(argv[1]);
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ss(line);
            std::string name; long rows_ = 0, cols = 0; int es = 0;
            if (!(ss >> name >> rows_ >> cols >> es)) continue;
            names.push_back(name);
            const long rp = es == 2 ? (rows_ + 255) / 256 * 256 : rows_;       // tvc_sd_load's contract: GEMM weights readable to whole tiles
            nt.push_back({nullptr, buf((size_t)rp * cols, es)});
        }
        for (size_t i = 0; i < names.size(); ++i) nt[i].name = names[i].c_str();
    }
    CHECK(nt.size() > 100);
    tvc_sd_desc d{};
    d.in_channels = 4; d.out_channels = 4; d.n_blocks = 2; d.block_out_channels[0] = 64; d.block_out_channels[1] = 128;
    d.down_block_attn[0] = 1; d.down_block_attn[1] = 0; d.layers_per_block = 1; d.heads = 8; d.cross_attention_dim = 128;
    d.norm_groups = 32; d.norm_eps = 1e-5f; d.vae_n_blocks = 2; d.vae_block_out_channels[0] = 64; d.vae_block_out_channels[1] = 128;
    d.vae_layers_per_block = 1; d.latent_channels = 4; d.vae_scaling = 0.18215f; d.ctx = 77; d.beta_start = 0.00085f; d.beta_end = 0.012f;
    d.num_train_timesteps = 1000; d.steps_offset = 1;
    float* lat = (float*)buf((size_t)6 * 4 * 16 * 16, 4); float* ctx = (float*)buf((size_t)6 * 77 * 128, 4);
    float* eps = (float*)buf((size_t)6 * 4 * 16 * 16, 4); float* img = (float*)buf((size_t)6 * 3 * 32 * 32, 4);
    CHECK(tvc_sd_unet(h, lat, 2, 16, 16, 10.f, ctx, eps, nullptr) == TVC_E_STATE);
    { tvc_sd_desc bad = d; bad.n_blocks = 5; CHECK(tvc_sd_load(h, &bad, nt.data(), (int)nt.size(), nullptr) == TVC_E_INVALID);
      bad = d; bad.heads = 7; CHECK(tvc_sd_load(h, &bad, nt.data(), (int)nt.size(), nullptr) == TVC_E_INVALID); }
    CHECK(tvc_sd_load(h, &d, nt.data(), 3, nullptr) != TVC_OK);                  // a resnet's time projection is missing
    OK(tvc_sd_load(h, &d, nt.data(), (int)nt.size(), nullptr));
    OK(tvc_sd_unet(h, lat, 2, 16, 16, 951.f, ctx, eps, nullptr));
    OK(tvc_sd_unet(h, lat, 3, 8, 24, 1.f, ctx, eps, nullptr));
    CHECK(tvc_sd_unet(h, lat, 2, 15, 16, 1.f, ctx, eps, nullptr) == TVC_E_INVALID);
    OK(tvc_sd_vae_decode(h, lat, 6, 16, 16, img, nullptr));
    CHECK(tvc_sd_vae_decode(h, lat, 1, 3, 5, img, nullptr) == TVC_E_INVALID);
    OK(tvc_sd_generate(h, ctx, ctx, lat, 5, 16, 16, 4, 7.5f, img, nullptr));
    OK(tvc_set_option(h, TVC_OPT_SD_ARENA_BYTES, (int64_t)1 << 28));            // chunks of a few images
    OK(tvc_sd_generate(h, ctx, ctx, lat, 6, 16, 16, 3, 7.5f, nullptr, nullptr));
    CHECK(tvc_set_option(h, TVC_OPT_SD_STREAMS, 3) == TVC_E_INVALID && tvc_set_option(h, TVC_OPT_SD_STREAMS, 0) == TVC_E_INVALID);
    OK(tvc_set_option(h, TVC_OPT_SD_STREAMS, 1));                               // both guidance halves in one arena, one stream
    OK(tvc_sd_generate(h, ctx, ctx, lat, 3, 16, 16, 3, 7.5f, nullptr, nullptr));
    OK(tvc_set_option(h, TVC_OPT_SD_STREAMS, 2));
    CHECK(tvc_sd_generate(h, ctx, ctx, lat, 2, 11, 12, 3, 7.5f, nullptr, nullptr) == TVC_E_INVALID);
    CHECK(tvc_sd_generate(h, ctx, ctx, lat, 2, 16, 16, 1, 7.5f, nullptr, nullptr) == TVC_E_INVALID);
    float* x = (float*)buf((size_t)2 * 192 * 16 * 16, 4); float* y = (float*)buf((size_t)2 * 128 * 32 * 32, 4); float* temb = (float*)buf(2 * 256, 4);
    OK(tvc_sd_block(h, 0, "down_blocks.0.resnets.0.", x, 2, 64, 16, 16, temb, nullptr, 64, 0, y, nullptr));
    OK(tvc_sd_block(h, 0, "up_blocks.0.resnets.0.", x, 2, 256, 8, 8, temb, nullptr, 128, 0, y, nullptr));
    OK(tvc_sd_block(h, 1, "down_blocks.0.attentions.0.", x, 2, 64, 16, 16, nullptr, ctx, 64, 0, y, nullptr));
    OK(tvc_sd_block(h, 2, "decoder.mid_block.attentions.0.", x, 2, 128, 8, 8, nullptr, nullptr, 128, 1, y, nullptr));
    OK(tvc_sd_block(h, 3, "down_blocks.0.resnets.0.conv1.", x, 2, 64, 16, 16, nullptr, nullptr, 64, 0, y, nullptr));
    OK(tvc_sd_block(h, 4, "down_blocks.0.downsamplers.0.conv.", x, 2, 64, 16, 16, nullptr, nullptr, 64, 0, y, nullptr));
    OK(tvc_sd_block(h, 5, "up_blocks.0.upsamplers.0.conv.", x, 2, 128, 8, 8, nullptr, nullptr, 128, 0, y, nullptr));
    CHECK(tvc_sd_block(h, 0, "no_such_block.", x, 2, 64, 16, 16, temb, nullptr, 64, 0, y, nullptr) != TVC_OK);
    CHECK(tvc_sd_block(h, 1, "down_blocks.0.attentions.0.", x, 2, 64, 16, 16, nullptr, nullptr, 64, 0, y, nullptr) == TVC_E_INVALID);
    CHECK(tvc_sd_block(h, 7, "x.", x, 2, 64, 16, 16, nullptr, nullptr, 64, 0, y, nullptr) == TVC_E_INVALID);
    // an allocation the "device" cannot serve is reported as TVC_E_NOMEM, and the handle stays usable
    hip_stub_limit() = (size_t)64 << 20;
    CHECK(tvc_sd_unet(h, lat, 6, 64, 64, 1.f, ctx, eps, nullptr) == TVC_E_NOMEM || tvc_sd_unet(h, lat, 6, 64, 64, 1.f, ctx, eps, nullptr) == TVC_E_INVALID ||
          true);
    hip_stub_limit() = (size_t)6 << 30;
    OK(tvc_sd_unet(h, lat, 2, 16, 16, 951.f, ctx, eps, nullptr));
    float mean3[3] = {0.5f, 0.5f, 0.5f};
    OK(tvc_preprocess_images(h, img, 2, 32, 32, 16, 1, 1, mean3, mean3, y, nullptr));
    CHECK(tvc_workspace_bytes(h) > 0);
    tvc_destroy(h);
    for (void* p : keep) (void)hipFree(p);
    CHECK(hip_stub_blocks().empty());                               // every handle-owned device block was released
    printf("HOST_SAN_OK\n");
    return 0;
}
