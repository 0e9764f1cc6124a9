"""ctypes binding of the TVC C-ABI (``include/tvc.h``) -- the only way the Python
mirror reaches the HIP kernels.  There is no CPU fallback: if ``libtvc_hip.so``
is missing or no GPU is visible, calls raise ``TVCError``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC_DIR = PKG_DIR / "csrc"
LIB_PATH = Path(os.environ["TVC_LIB_PATH"]) if os.environ.get("TVC_LIB_PATH") else PKG_DIR / "libtvc_hip.so"   # override: kernel experiments only
HEADER_PATH = PKG_DIR.parent / "include" / "tvc.h"

TVC_OK, TVC_E_INVALID, TVC_E_HIP, TVC_E_NOMEM, TVC_E_STATE, TVC_E_OVERFLOW = range(6)
TVC_DTYPE_BF16, TVC_DTYPE_F32 = 0, 1
TVC_REC_HEAD, TVC_REC_MAXREF = 12, 16
TVC_ABI_VERSION, TVC_MAX_BANKS, TVC_MAX_TOPK = 4, 8, 128
TVC_OPT_TEXT_PACKING, TVC_OPT_MAX_CHUNK_IMAGES, TVC_OPT_MAX_CHUNK_TEXTS, TVC_OPT_BANK_FILTER = 1, 2, 3, 4
TVC_OPT_TEXT_GROUP = 5
TVC_OPT_POOLED_LAST_LAYER = 6
TVC_OPT_TOWER_PRECISION = 7
TVC_OPT_SD_ARENA_BYTES = 8
TVC_OPT_SD_STREAMS = 9


class TVCError(RuntimeError):
    """An error reported by the C-ABI (code + tvc_last_error message)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"TVC error {code}: {message}")
        self.code = code


class TowerArch(C.Structure):
    _fields_ = [("width", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("mlp", C.c_int32), ("act", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("patch", C.c_int32), ("vocab", C.c_int32), ("ctx", C.c_int32),
                ("embed_dim", C.c_int32), ("vision", TowerArch), ("text", TowerArch)]


_P = C.c_void_p


class LayerWeights(C.Structure):
    _fields_ = [(n, _P) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo",
                                  "ln2_g", "ln2_b", "w1", "b1", "w2", "b2")]


class VisionWeights(C.Structure):
    _fields_ = [(n, _P) for n in ("patch_w", "cls", "pos", "ln_pre_g", "ln_pre_b",
                                  "ln_post_g", "ln_post_b", "proj")] + [("layers", C.POINTER(LayerWeights))]


class TextWeights(C.Structure):
    _fields_ = [(n, _P) for n in ("tok_emb", "pos", "ln_final_g", "ln_final_b", "proj")] + \
               [("layers", C.POINTER(LayerWeights))]


class ConsistencyParams(C.Structure):
    _fields_ = [("reference_count", C.c_int32), ("similarity_threshold", C.c_float),
                ("retrieval_top_k", C.c_int32), ("dup_threshold", C.c_float),
                ("w_text_variants", C.c_float), ("w_consistency", C.c_float),
                ("w_exp", C.c_float * 4)]


class SDDesc(C.Structure):
    """``tvc_sd_desc`` (include/tvc.h)."""
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("n_blocks", C.c_int32),
                ("block_out_channels", C.c_int32 * 4), ("down_block_attn", C.c_int32 * 4),
                ("layers_per_block", C.c_int32), ("heads", C.c_int32), ("heads_per_block", C.c_int32 * 4),
                ("prediction_type", C.c_int32), ("cross_attention_dim", C.c_int32),
                ("norm_groups", C.c_int32), ("norm_eps", C.c_float),
                ("vae_n_blocks", C.c_int32), ("vae_block_out_channels", C.c_int32 * 4),
                ("vae_layers_per_block", C.c_int32), ("latent_channels", C.c_int32), ("vae_scaling", C.c_float),
                ("ctx", C.c_int32), ("beta_start", C.c_float), ("beta_end", C.c_float),
                ("num_train_timesteps", C.c_int32), ("steps_offset", C.c_int32)]


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p)]


# name -> (restype, argtypes); must list every symbol include/tvc.h declares
SIGNATURES = {
    "tvc_abi_version": (C.c_uint32, []),
    "tvc_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(VisionWeights), C.POINTER(TextWeights), C.POINTER(_P)]),
    "tvc_destroy": (None, [_P]),
    "tvc_last_error": (C.c_char_p, [_P]),
    "tvc_workspace_bytes": (C.c_uint64, [_P]),
    "tvc_set_option": (C.c_int, [_P, C.c_int32, C.c_int64]),
    "tvc_encode_image": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    "tvc_encode_text": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    "tvc_bank_set": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "tvc_bank_select": (C.c_int, [_P, C.c_int32]),
    "tvc_bank_search": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int64, _P, _P, _P, _P]),
    "tvc_bank_search_dense": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int64, _P, _P, _P, _P]),
    "tvc_bank_status": (C.c_int, [_P, _P]),
    "tvc_bank_gather": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P]),
    "tvc_topk_merge": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 _P, _P, _P, _P, _P]),
    "tvc_cosine_matrix": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    "tvc_consistency": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32,
                                  C.POINTER(ConsistencyParams), _P, _P]),
    "tvc_profile_begin": (C.c_int, [_P]),
    "tvc_profile_end": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), _P]),
    "tvc_gemm_bf16": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                C.c_int32, _P]),
    "tvc_attention": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "tvc_layernorm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "tvc_encode_text_hidden": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "tvc_encode_image_grad": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    "tvc_encode_image_backward": (C.c_int, [_P, _P, _P, _P]),
    "tvc_pgd_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_int32, _P]),
    "tvc_l2_step": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, _P]),
    "tvc_attention_backward": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "tvc_layernorm_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    # fp32-grade towers: the *_f32 weight structs have the field order of LayerWeights / VisionWeights / TextWeights
    "tvc_set_weights_f32": (C.c_int, [_P, C.POINTER(VisionWeights), C.POINTER(TextWeights)]),
    "tvc_gemm_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "tvc_attention_f32": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "tvc_gemm_split": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "tvc_attention_split": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    # latent-diffusion reference generator
    "tvc_sd_load": (C.c_int, [_P, C.POINTER(SDDesc), C.POINTER(NamedTensor), C.c_int32, _P]),
    "tvc_sd_unet": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, _P]),
    "tvc_sd_vae_decode": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "tvc_sd_generate": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P]),
    "tvc_sd_block": (C.c_int, [_P, C.c_int32, C.c_char_p, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32,
                               C.c_int32, _P, _P]),
    "tvc_preprocess_images": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "tvc_sd_attention": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
}

_lib = None
_lib_lock = threading.Lock()


def build(force: bool = False) -> Path:
    """Compile the HIP sources for gfx950 in-tree (``make`` drives hipcc;
    cross-compiles without a GPU).  Returns the path of the shared library."""
    if force:
        subprocess.run(["make", "-C", str(CSRC_DIR), "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", str(CSRC_DIR), "-j", str(min(8, os.cpu_count() or 1))],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libtvc_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if not LIB_PATH.exists():
        raise RuntimeError(f"make succeeded but {LIB_PATH} is missing")
    return LIB_PATH


def load() -> C.CDLL:
    """Load ``libtvc_hip.so`` and attach prototypes.  Raises if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise TVCError(TVC_E_STATE,
                           f"{LIB_PATH} not found: the TVC path has no CPU fallback. Build it with "
                           f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C {CSRC_DIR}`.")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(handle, code: int) -> None:
    if code != TVC_OK:
        msg = load().tvc_last_error(handle)
        raise TVCError(code, msg.decode("utf-8", "replace") if msg else "")


def rec_stride(n_variants: int) -> int:
    return TVC_REC_HEAD + n_variants + 2 * TVC_REC_MAXREF
