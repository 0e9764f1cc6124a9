// Internal: the handle behind include/tvc.h, its workspace slots and the launch helpers shared by the translation
// units that implement the C-ABI (tvc_abi.cpp: CLIP towers, bank, consistency; tvc_precise.cpp: fp32-grade towers;
// tvc_sd.cpp: latent-diffusion reference generator).
#pragma once
#include "../../include/tvc.h"
#include "kernels.hpp"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

inline thread_local std::string g_create_error;

enum Slot {
    // tower workspaces exist twice (vision, text: + WS_TOWER_N) so that the two towers can run
    // concurrently on two streams
    WS_X = 0, WS_H, WS_QKV, WS_MLP, WS_CLS, WS_DELTA, WS_DELTA2, WS_SPLITK, WS_POOL, WS_TOWER_N,
    WS_TX = WS_TOWER_N, WS_TH, WS_TQKV, WS_TMLP, WS_TCLS, WS_TDELTA, WS_TDELTA2, WS_TSPLITK, WS_TPOOL,
    WS_PATCH, WS_EOT, WS_STARTS, WS_PFX, WS_LENS,
    WS_COSX, WS_COSY, WS_COSXP, WS_COSYP,
    WS_QPLANES, WS_S0, WS_TAU, WS_CAND, WS_CAND_CNT, WS_MOM_PART, WS_OVERFLOW,
    // input-gradient path (vision tower): saved layer inputs, gradient stream, scratch
    WS_GSAVE, WS_GOUT, WS_GXL, WS_GDX, WS_G16, WS_GMLP2, WS_GDQKV, WS_GSTATS, WS_GSMALL, WS_GPATCH,
    // fp32-grade towers (TVC_OPT_TOWER_PRECISION = 1): vision set, then the text set (+ WS_P_N)
    WS_PX, WS_PH, WS_PQKV, WS_PMLP, WS_PCLS, WS_P_N_END,
    WS_PTX = WS_P_N_END, WS_PTH, WS_PTQKV, WS_PTMLP, WS_PTCLS, WS_PEOT,
    // split-bf16 towers (TVC_OPT_TOWER_PRECISION = 2, tvc_split.cpp): vision set, then the text set (+ WS_S_N)
    WS_SX, WS_SH, WS_SQKV, WS_SU, WS_SM, WS_SDELTA1, WS_SDELTA2, WS_SCLS, WS_S_END,
    WS_STX = WS_S_END, WS_STH, WS_STQKV, WS_STU, WS_STM, WS_STDELTA1, WS_STDELTA2, WS_STCLS,
    // latent-diffusion reference generator (tvc_sd.cpp)
    WS_SD0, WS_SD1, WS_SD2, WS_SD3, WS_SD4, WS_SD5, WS_SD6, WS_SD7,
    WS_COUNT
};

struct Buf {
    void* p = nullptr;
    size_t n = 0;
};

struct BankSlot {
    const uint16_t* bank = nullptr;
    void* owned = nullptr;            // (hi | lo) planes of an fp32 bank
    int64_t R = 0;
    int D = 0;
    int planes = 1;
    float* bounds = nullptr;          // device [2]: max row norms of the bank planes
};

constexpr int WS_S_N = WS_S_END - WS_SX;      // offset from a vision split slot to its text twin

// hi | lo bf16 planes [round_up(out, 256), 2 * in] of one layer's GEMM weights (split-bf16 mode), handle-owned
struct SplitLayer { uint16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr; };

struct ProfRec {
    hipEvent_t a, b;
    int cat;
    double work;
    double big_bytes = 0;     // GEMM launches of >= 64 output tiles (the persistent ring kernels): compulsory HBM bytes
    int gI = 0, gJ = 0, gK = 0, gP = 0, gS = 0;     // GEMM launches: shape, planes, fixed K split (TVC_PROF_DUMP)
};

struct tvc_handle {
    tvc_model_desc desc{};
    bool has_vision = false, has_text = false;
    tvc_vision_weights vw{};
    tvc_text_weights tw{};
    std::vector<tvc_layer_weights> vlayers, tlayers;
    // banks: TVC_MAX_BANKS independent slots (retriever index, reference bank, defense references ...
    // registered by different owners on one engine); tvc_bank_select picks the one the bank calls address
    BankSlot banks[TVC_MAX_BANKS];
    int cur_bank = 0;
    bool bank_filter = true;          // TVC_OPT_BANK_FILTER
    Buf ws[WS_COUNT];
    std::string err;
    int max_chunk_images = 512;
    int max_chunk_texts = 4608;
    bool pack_text = true;     // TVC_OPT_TEXT_PACKING
    int text_group = 0;        // TVC_OPT_TEXT_GROUP: texts come in groups of this many sharing prefixes (0: off)
    bool pooled_last = true;   // TVC_OPT_POOLED_LAST_LAYER
    // input-gradient state: transposed GEMM weights (built on first use), what the last tvc_encode_image_grad saw
    std::vector<void*> wT;     // per layer: wqkvT, woT, w1T, w2T; then projT, patchT
    int grad_B = 0;
    int grad_normalize = 0;
    const float* grad_pix = nullptr;
    bool prof = false;
    std::vector<ProfRec> prof_recs;
    // fp32-grade towers: fp32 copies of every weight (tvc_set_weights_f32) and the switch
    bool has_vision32 = false, has_text32 = false;
    tvc_vision_weights_f32 vw32{};
    tvc_text_weights_f32 tw32{};
    std::vector<tvc_layer_weights_f32> vlayers32, tlayers32;
    int tower_precision = 0;   // TVC_OPT_TOWER_PRECISION
    // split-bf16 mode (precision 2): planes of every GEMM weight, built from the fp32 copies when the option is set
    std::vector<SplitLayer> vsplit, tsplit;
    uint16_t* vsplit_patch = nullptr;
    std::vector<void*> split_owned;
    bool split_ready = false;
    struct SdState* sd = nullptr;   // latent-diffusion model (tvc_sd.cpp), owned
    size_t sd_arena_bytes = (size_t)48 << 30;   // TVC_OPT_SD_ARENA_BYTES: activation arena of one UNet evaluation
    // TVC_OPT_SD_STREAMS: the two classifier-free-guidance halves of a UNet evaluation on two HIP streams (tvc_sd.cpp)
    int sd_streams = 2;
    hipStream_t sd_aux = nullptr;
    hipEvent_t sd_fork = nullptr, sd_join = nullptr;
};
void tvc_sd_free(tvc_handle* h);    // tvc_sd.cpp
void tvc_split_free(tvc_handle* h); // tvc_split.cpp
int tvc_split_prepare(tvc_handle* h);

inline int fail(tvc_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t st__ = (expr);                                                            \
        if (st__ != hipSuccess)                                                              \
            return fail(h, TVC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(st__));  \
    } while (0)

inline int ensure(tvc_handle* h, Slot s, size_t bytes) {
    Buf& b = h->ws[s];
    if (b.n >= bytes && b.p) return TVC_OK;
    if (b.p) {
        // hipFree synchronises the device, so kernels still using the old block are done
        if (hipFree(b.p) != hipSuccess) return fail(h, TVC_E_HIP, "hipFree(workspace) failed");
        b.p = nullptr; b.n = 0;
    }
    // grow with a little slack so alternating sizes do not thrash
    const size_t want = bytes + bytes / 16 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        char m[128];
        snprintf(m, sizeof m, "workspace allocation of %zu bytes failed", want);
        return fail(h, TVC_E_NOMEM, m);
    }
    b.n = want;
    return TVC_OK;
}

// RAII bracket: two events around whatever is launched inside the scope
struct ProfScope {
    tvc_handle* h; hipStream_t st; ProfRec r; bool on;
    ProfScope(tvc_handle* h_, hipStream_t st_, int cat, double work) : h(h_), st(st_), on(h_->prof) {
        if (!on) return;
        r.cat = cat; r.work = work;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        h->prof_recs.push_back(r);
    }
};

inline double gemm_flops(const GemmLaunch& g) { return 2.0 * g.I * (double)g.J * g.K * g.planes; }

// Compulsory HBM bytes of a GEMM launch: every distinct operand plane read once, the output written once (read once
// more by the residual epilogue).  Planes that address the same rows shifted by whole rows (the nine taps of a 3 x 3
// convolution on the token operand) count once.
inline double gemm_compulsory_bytes(const GemmLaunch& g) {
    auto distinct = [&](const int* off, int64_t ld) {
        int n = 0;
        for (int p = 0; p < g.planes; ++p) {
            bool seen = false;
            for (int q = 0; q < p; ++q) seen = seen || (off[p] % ld == off[q] % ld);
            n += seen ? 0 : 1;
        }
        return n;
    };
    const double a = 2.0 * g.I * (double)g.K * distinct(g.a_plane_off, g.lda > 0 ? g.lda : g.K);
    const double b = 2.0 * g.J * (double)g.K * distinct(g.b_plane_off, g.ldb > 0 ? g.ldb : g.K);
    const double o = (double)g.I * g.J * (g.epilogue == TVC_EPI_F32 ? 4.0 : g.epilogue == TVC_EPI_RESID_F32 ? 8.0 : 2.0);
    return a + b + o;
}

inline hipError_t timed_gemm(tvc_handle* h, const GemmLaunch& g, hipStream_t st, int splitk_slot = -1) {
    ProfScope ps(h, st, TVC_PROF_GEMM, gemm_flops(g));
    if (ps.on && (int64_t)((g.I + 255) / 256) * ((g.J + 255) / 256) >= 64) ps.r.big_bytes = gemm_compulsory_bytes(g);
    if (ps.on) { ps.r.gI = g.I; ps.r.gJ = g.J; ps.r.gK = g.K; ps.r.gP = g.planes; ps.r.gS = g.splitk_fixed; }
    if (splitk_slot >= 0 && h->ws[splitk_slot].p) {
        GemmLaunch g2 = g;
        g2.splitk_ws = (float*)h->ws[splitk_slot].p;
        g2.splitk_ws_bytes = h->ws[splitk_slot].n;
        return launch_gemm_bf16(g2, st);
    }
    return launch_gemm_bf16(g, st);
}
