"""``TVCEngine``: one C-ABI handle on one GPU, plus the tensor plumbing around it.

PyTorch-ROCm is used for device memory and streams only: every computation on
the hot path is a HIP kernel launched through ``include/tvc.h``.
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .arch import ClipArch


@dataclass
class ConsistencyConfig:
    """Fields of the reference configs the consistency stage reads."""
    reference_count: int = 5             # experiments/defenses/retrieval_ref.py:23
    similarity_threshold: float = 0.3    # retrieval_ref.py:24
    retrieval_top_k: int = 10            # experiments/defenses/detector.py:29
    dup_threshold: float = 0.95          # experiments/defenses/detector.py:318
    w_text_variants: float = 0.4         # src/detector.py:667
    w_consistency: float = 0.2           # src/detector.py:669
    w_exp: Tuple[float, float, float, float] = (0.25, 0.25, 0.25, 0.25)   # consistency_checker.py:61-66
    search_k: int = 5                    # rows searched per text; >= reference_count

    def to_c(self) -> _lib.ConsistencyParams:
        p = _lib.ConsistencyParams()
        p.reference_count = self.reference_count
        p.similarity_threshold = self.similarity_threshold
        p.retrieval_top_k = self.retrieval_top_k
        p.dup_threshold = self.dup_threshold
        p.w_text_variants = self.w_text_variants
        p.w_consistency = self.w_consistency
        for i, w in enumerate(self.w_exp):
            p.w_exp[i] = w
        return p


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _require_cuda(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); the TVC path has no CPU fallback")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class TVCEngine:
    """Owns a ``tvc_handle`` and the device copies of the tower weights."""

    def __init__(self, arch: Optional[ClipArch] = None, vision_w: Optional[Dict] = None,
                 text_w: Optional[Dict] = None, device: str = "cuda:0", precision: str = "bf16"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.TVCError(_lib.TVC_E_HIP, "no GPU visible to PyTorch-ROCm; the TVC path has no CPU fallback")
        self.device = torch.device(device)
        self.arch = arch
        self._lock = threading.Lock()
        self.dense_fallbacks = 0           # searches that overflowed the candidate lists and were redone brute force
        self._keep = []          # device tensors / ctypes objects referenced by the handle
        # named bank slots: every owner (retriever index, ReferenceBank, defense references ...) registers
        # its rows under its own name, so owners sharing one engine cannot replace each other's bank
        self._banks: Dict[str, Dict] = {}
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            desc = vis = txt = None
            if arch is not None:
                desc = _lib.ModelDesc(arch.image_size, arch.patch, arch.vocab, arch.ctx, arch.embed_dim,
                                      _lib.TowerArch(arch.vision.width, arch.vision.layers, arch.vision.heads, arch.vision.mlp,
                                                     int(arch.vision.act == "gelu")),
                                      _lib.TowerArch(arch.text.width, arch.text.layers, arch.text.heads, arch.text.mlp,
                                                     int(arch.text.act == "gelu")))
                if vision_w is not None:
                    vis = self._vision_struct(arch, vision_w)
                if text_w is not None:
                    txt = self._text_struct(arch, text_w)
            rc = self.lib.tvc_create(C.byref(desc) if desc is not None else None,
                                     C.byref(vis) if vis is not None else None,
                                     C.byref(txt) if txt is not None else None, C.byref(self.handle))
            _lib.check(None, rc)
        self._w_host = (vision_w, text_w)     # the caller's (fp32) weight dicts: the fp32-grade mode uploads them as they are
        self._has_f32 = False
        self.precision = "bf16"
        if precision != "bf16":
            self.set_precision(precision)

    # ---- weights -------------------------------------------------------
    def _layers_f32(self, layers) -> "C.Array":
        arr = (_lib.LayerWeights * len(layers))()
        for i, lw in enumerate(layers):
            for name in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2"):
                setattr(arr[i], name, self._dev(lw[name], torch.float32).data_ptr())
        self._keep.append(arr)
        return arr

    def _upload_f32_weights(self) -> None:
        """fp32 copies of every weight for ``TVC_OPT_TOWER_PRECISION = 1`` (nothing rounded to bf16)."""
        vw, tw = self._w_host
        vis = txt = None
        if vw is not None:
            vis = _lib.VisionWeights()
            for name in ("patch_w", "cls", "pos", "ln_pre_g", "ln_pre_b", "ln_post_g", "ln_post_b", "proj"):
                setattr(vis, name, self._dev(vw[name], torch.float32).data_ptr())
            vis.layers = C.cast(self._layers_f32(vw["layers"]), C.POINTER(_lib.LayerWeights))
            self._keep.append(vis)
        if tw is not None:
            txt = _lib.TextWeights()
            for name in ("tok_emb", "pos", "ln_final_g", "ln_final_b", "proj"):
                setattr(txt, name, self._dev(tw[name], torch.float32).data_ptr())
            txt.layers = C.cast(self._layers_f32(tw["layers"]), C.POINTER(_lib.LayerWeights))
            self._keep.append(txt)
        self._check(self.lib.tvc_set_weights_f32(self.handle, C.byref(vis) if vis is not None else None,
                                                 C.byref(txt) if txt is not None else None))
        self._has_f32 = True

    PRECISIONS = {"bf16": 0, "fp32": 1, "split": 2}

    def set_precision(self, precision: str) -> None:
        """Tower arithmetic (``TVC_OPT_TOWER_PRECISION``):

        * ``"bf16"`` (default, the benchmarked path): bf16 x bf16 products, fp32 accumulation; scores within ~1e-3 of the
          reference's fp32 CPU path;
        * ``"split"``: fp32-grade at about a third of the bf16 matrix rate -- every activation and weight travels as
          hi | lo bf16 planes, three MFMA products per element; scores within 1e-4 of the fp32 CPU path END TO END (the
          bar of BASELINE.json), EOT packing / prefix sharing kept;
        * ``"fp32"``: the exact reference -- every GEMM on the exact-f32 matrix instruction (1/16 of the bf16 rate),
          fp32 attention; embeddings within ~1e-6 of the fp32 CPU path.

        ``"split"`` and ``"fp32"`` upload fp32 copies of the caller's weights.  The input-gradient entry points
        (``encode_image_grad`` / ``encode_image_backward``: the PGD / Hubness loops) ALWAYS run the bf16 path, whatever
        this is set to."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)} (got {precision!r})")
        with self._lock, torch.cuda.device(self.device):
            if precision != "bf16" and not self._has_f32:
                if self._w_host[0] is None and self._w_host[1] is None:
                    raise _lib.TVCError(_lib.TVC_E_STATE, "this engine has no towers to run in fp32")
                self._upload_f32_weights()
            self._check(self.lib.tvc_set_option(self.handle, _lib.TVC_OPT_TOWER_PRECISION, self.PRECISIONS[precision]))
        self.precision = precision

    def _dev(self, t: torch.Tensor, dtype) -> torch.Tensor:
        d = t.detach().to(device=self.device, dtype=dtype).contiguous()
        self._keep.append(d)
        return d

    def _layers(self, layers) -> "C.Array":
        arr = (_lib.LayerWeights * len(layers))()
        for i, lw in enumerate(layers):
            for name in ("ln1_g", "ln1_b", "bqkv", "bo", "ln2_g", "ln2_b", "b1", "b2"):
                setattr(arr[i], name, self._dev(lw[name], torch.float32).data_ptr())
            for name in ("wqkv", "wo", "w1", "w2"):
                setattr(arr[i], name, self._dev(lw[name], torch.bfloat16).data_ptr())
        self._keep.append(arr)
        return arr

    def _vision_struct(self, arch: ClipArch, w: Dict) -> _lib.VisionWeights:
        s = _lib.VisionWeights()
        pw = torch.zeros((arch.vision.width, arch.patch_k_padded), dtype=torch.float32)
        pw[:, :arch.patch_k] = w['patch_w'].float().cpu()
        s.patch_w = self._dev(pw, torch.bfloat16).data_ptr()
        for name in ("cls", "pos", "ln_pre_g", "ln_pre_b", "ln_post_g", "ln_post_b"):
            setattr(s, name, self._dev(w[name], torch.float32).data_ptr())
        s.proj = self._dev(w['proj'], torch.bfloat16).data_ptr()
        s.layers = C.cast(self._layers(w['layers']), C.POINTER(_lib.LayerWeights))
        self._keep.append(s)
        return s

    def _text_struct(self, arch: ClipArch, w: Dict) -> _lib.TextWeights:
        s = _lib.TextWeights()
        for name in ("tok_emb", "pos", "ln_final_g", "ln_final_b"):
            setattr(s, name, self._dev(w[name], torch.float32).data_ptr())
        s.proj = self._dev(w['proj'], torch.bfloat16).data_ptr()
        s.layers = C.cast(self._layers(w['layers']), C.POINTER(_lib.LayerWeights))
        self._keep.append(s)
        return s

    def close(self) -> None:
        if getattr(self, "handle", None) and self.handle.value:
            torch.cuda.synchronize(self.device)
            self.lib.tvc_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        _lib.check(self.handle, rc)

    # ---- encoders ------------------------------------------------------
    def encode_image(self, pixels: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """pixels [B, 3, S, S] (preprocessed, on the GPU) -> fp32 [B, D]."""
        a = self.arch
        if pixels.dim() == 3:
            pixels = pixels.unsqueeze(0)
        if pixels.dim() != 4 or pixels.shape[1:] != (3, a.image_size, a.image_size):
            raise ValueError(f"expected [B, 3, {a.image_size}, {a.image_size}], got {tuple(pixels.shape)}")
        pixels = _require_cuda(pixels, torch.float32, "pixels")
        out = torch.empty((pixels.shape[0], a.embed_dim), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_encode_image(self.handle, _ptr(pixels), pixels.shape[0], _ptr(out),
                                                  int(normalize), _stream()))
        return out

    def encode_text_hidden(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens int [T, ctx] (on the GPU) -> fp32 [T, ctx, width]: ln_final of the hidden state at every position
        (``CLIPTextModel.last_hidden_state``; the conditioning a latent-diffusion UNet takes)."""
        a = self.arch
        if tokens.dim() != 2 or tokens.shape[1] != a.ctx:
            raise ValueError(f"expected [T, {a.ctx}] tokens, got {tuple(tokens.shape)}")
        tokens = _require_cuda(tokens, torch.int32, "tokens")
        out = torch.empty((tokens.shape[0], a.ctx, a.text.width), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_encode_text_hidden(self.handle, _ptr(tokens), tokens.shape[0], _ptr(out), _stream()))
        return out

    # ---- input gradient of the vision tower (attacks: PGD / Hubness inner loop) ----------------
    def encode_image_grad(self, pixels: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """As ``encode_image`` but keeps what ``encode_image_backward`` needs inside the handle.  ``pixels`` must stay
        alive and unchanged until the backward (the engine holds a reference)."""
        a = self.arch
        if pixels.dim() != 4 or pixels.shape[1:] != (3, a.image_size, a.image_size):
            raise ValueError(f"expected [B, 3, {a.image_size}, {a.image_size}], got {tuple(pixels.shape)}")
        pixels = _require_cuda(pixels, torch.float32, "pixels")
        out = torch.empty((pixels.shape[0], a.embed_dim), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_encode_image_grad(self.handle, _ptr(pixels), pixels.shape[0], _ptr(out),
                                                       int(normalize), _stream()))
            self._grad_pixels = pixels
            self._grad_generation = getattr(self, "_grad_generation", 0) + 1
        return out

    def encode_image_backward(self, grad_out: torch.Tensor, generation: Optional[int] = None) -> torch.Tensor:
        """d(loss)/d(out of the last ``encode_image_grad``) [B, D] -> d(loss)/d(pixels) [B, 3, S, S]."""
        px = getattr(self, "_grad_pixels", None)
        if px is None:
            raise RuntimeError("encode_image_backward: no encode_image_grad call to differentiate")
        if generation is not None and generation != self._grad_generation:
            raise RuntimeError("encode_image_backward: a later encode_image_grad call replaced the saved activations "
                               "(one differentiable image batch at a time per engine)")
        grad_out = _require_cuda(grad_out, torch.float32, "grad_out")
        if tuple(grad_out.shape) != (px.shape[0], self.arch.embed_dim):
            raise ValueError(f"grad_out must be [{px.shape[0]}, {self.arch.embed_dim}], got {tuple(grad_out.shape)}")
        gp = torch.empty_like(px)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_encode_image_backward(self.handle, _ptr(grad_out), _ptr(gp), _stream()))
        return gp

    def pgd_step(self, adv: torch.Tensor, clean: torch.Tensor, grad: torch.Tensor, momentum: Optional[torch.Tensor],
                 eps: float, alpha: float, mu: float = 0.9, clip_min: float = 0.0, clip_max: float = 1.0,
                 targeted: bool = False) -> None:
        """One projected sign-gradient step, in place on ``adv`` (and ``momentum``)."""
        for name, t in (("adv", adv), ("clean", clean), ("grad", grad), ("momentum", momentum)):
            if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == adv.shape):
                raise ValueError(f"{name}: contiguous fp32 device tensor shaped like adv expected")
        B = adv.shape[0]
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_pgd_step(self.handle, _ptr(adv), _ptr(clean), _ptr(grad), _ptr(momentum), B,
                                              adv.numel() // max(B, 1), eps, alpha, mu, clip_min, clip_max,
                                              int(targeted), _stream()))

    def l2_step(self, adv: torch.Tensor, clean: torch.Tensor, grad: torch.Tensor, eps: float, step: float,
                clip_min: float = 0.0, clip_max: float = 1.0, descent: bool = True) -> None:
        """One L2-normalised gradient step + projection onto the eps L2 ball + clamp, in place on ``adv``
        (src/attacks/hubness_attack.py:378-386)."""
        for name, t in (("adv", adv), ("clean", clean), ("grad", grad)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == adv.shape):
                raise ValueError(f"{name}: contiguous fp32 device tensor shaped like adv expected")
        B = adv.shape[0]
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_l2_step(self.handle, _ptr(adv), _ptr(clean), _ptr(grad), B, adv.numel() // max(B, 1),
                                             eps, step, clip_min, clip_max, int(descent), _stream()))

    def encode_text(self, tokens: torch.Tensor, normalize: bool = True, group: int = 0) -> torch.Tensor:
        """tokens int [T, ctx] (on the GPU) -> fp32 [T, D].  ``group`` = N + 1 declares that the rows
        are consecutive (original, variant_1 .. variant_N) groups: variants then share the rows of
        the token prefix they have in common with their original (``TVC_OPT_TEXT_GROUP``; the
        embeddings are bit-identical, the text tower does less work)."""
        a = self.arch
        if tokens.dim() != 2 or tokens.shape[1] != a.ctx:
            raise ValueError(f"expected [T, {a.ctx}] token ids, got {tuple(tokens.shape)}")
        tokens = _require_cuda(tokens, torch.int32, "tokens")
        out = torch.empty((tokens.shape[0], a.embed_dim), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            if group >= 2:
                self._check(self.lib.tvc_set_option(self.handle, _lib.TVC_OPT_TEXT_GROUP, int(group)))
            try:
                self._check(self.lib.tvc_encode_text(self.handle, _ptr(tokens), tokens.shape[0], _ptr(out),
                                                     int(normalize), _stream()))
            finally:
                if group >= 2:
                    self._check(self.lib.tvc_set_option(self.handle, _lib.TVC_OPT_TEXT_GROUP, 0))
        return out

    # ---- bank ----------------------------------------------------------
    DEFAULT_BANK = "default"

    def _slot(self, name: str, create: bool = False) -> Dict:
        b = self._banks.get(name)
        if b is None:
            if not create:
                raise _lib.TVCError(_lib.TVC_E_STATE, f"no bank registered under the name {name!r}")
            used = {v["slot"] for v in self._banks.values()}
            free = [i for i in range(_lib.TVC_MAX_BANKS) if i not in used]
            if not free:
                raise _lib.TVCError(_lib.TVC_E_STATE, f"all {_lib.TVC_MAX_BANKS} bank slots of this engine are in use "
                                                      f"({sorted(self._banks)}); release_bank() one first")
            b = self._banks[name] = {"slot": free[0], "rows": 0, "dim": 0, "keep": None}
        return b

    def _select(self, name: str, create: bool = False) -> Dict:
        """(lock held) point the handle's bank calls at the slot of ``name``."""
        b = self._slot(name, create)
        self._check(self.lib.tvc_bank_select(self.handle, b["slot"]))
        return b

    @property
    def bank_rows(self) -> int:
        return self.bank_size(self.DEFAULT_BANK)

    @property
    def bank_dim(self) -> int:
        b = self._banks.get(self.DEFAULT_BANK)
        return b["dim"] if b else 0

    def bank_size(self, name: str = DEFAULT_BANK) -> int:
        b = self._banks.get(name)
        return b["rows"] if b else 0

    def has_bank(self, name: str = DEFAULT_BANK) -> bool:
        return name in self._banks

    def bank_is_bf16(self, name: str = DEFAULT_BANK) -> bool:
        """True when the bank of ``name`` was registered as bf16 (its rows are exact in bf16)."""
        b = self._banks.get(name)
        return bool(b and b.get("bf16"))

    def set_bank(self, bank: torch.Tensor, name: str = DEFAULT_BANK) -> None:
        """bank [R, D], rows L2-normalised; bf16 is used in place, fp32 is split
        into (hi, lo) bf16 planes inside the handle.  ``name`` = the owner's slot."""
        if bank.dim() != 2:
            raise ValueError("bank must be [R, D]")
        if bank.dtype not in (torch.bfloat16, torch.float32):
            bank = bank.float()
        bank = _require_cuda(bank, bank.dtype, "bank")
        dt = _lib.TVC_DTYPE_BF16 if bank.dtype == torch.bfloat16 else _lib.TVC_DTYPE_F32
        with self._lock, torch.cuda.device(self.device):
            b = self._select(name, create=True)
            self._check(self.lib.tvc_bank_set(self.handle, _ptr(bank), bank.shape[0], bank.shape[1], dt, _stream()))
            if dt == _lib.TVC_DTYPE_F32:
                torch.cuda.current_stream().synchronize()   # the split read `bank`; it may now be freed
            b["keep"] = bank if dt == _lib.TVC_DTYPE_BF16 else None
            b["bf16"] = dt == _lib.TVC_DTYPE_BF16
            b["rows"], b["dim"] = bank.shape

    def release_bank(self, name: str) -> None:
        """Forget the bank of ``name`` and free its slot (an fp32 bank's planes are freed too)."""
        with self._lock, torch.cuda.device(self.device):
            if name in self._banks:
                self._select(name)
                self._check(self.lib.tvc_bank_set(self.handle, None, 0, 64, _lib.TVC_DTYPE_BF16, _stream()))
                del self._banks[name]

    def bank_search(self, rows: torch.Tensor, k: int, count_thr: float = 0.3, idx_offset: int = 0,
                    want_moments: bool = True, bank: str = DEFAULT_BANK):
        """rows fp32 [M, D] -> (idx int32 [M, k], sim fp32 [M, k], moments fp32 [M, 4] | None)."""
        if not 1 <= k <= _lib.TVC_MAX_TOPK:
            raise ValueError(f"top-k must be in [1, {_lib.TVC_MAX_TOPK}] (got {k}): tvc_bank_search never truncates silently")
        rows = _require_cuda(rows, torch.float32, "rows")
        M = rows.shape[0]
        idx = torch.empty((M, k), dtype=torch.int32, device=self.device)
        sim = torch.empty((M, k), dtype=torch.float32, device=self.device)
        mom = torch.empty((M, 4), dtype=torch.float32, device=self.device) if want_moments else None
        with self._lock, torch.cuda.device(self.device):
            self._select(bank)
            self._check(self.lib.tvc_bank_search(self.handle, _ptr(rows), M, k, count_thr, idx_offset,
                                                 _ptr(idx), _ptr(sim), _ptr(mom), _stream()))
        return idx, sim, mom

    def bank_search_robust(self, rows: torch.Tensor, k: int, count_thr: float = 0.3, idx_offset: int = 0,
                           want_moments: bool = True, bank: str = DEFAULT_BANK):
        """``bank_search`` + status check; on candidate overflow (degenerate bank) the same
        contract is recomputed by the brute-force kernel.  Synchronises the stream."""
        out = self.bank_search(rows, k, count_thr, idx_offset, want_moments, bank)
        try:
            self.bank_status()
            return out
        except _lib.TVCError as e:
            if e.code != _lib.TVC_E_OVERFLOW:
                raise
        idx, sim, mom = out
        rows = _require_cuda(rows, torch.float32, "rows")
        with self._lock, torch.cuda.device(self.device):
            self._select(bank)
            self._check(self.lib.tvc_bank_search_dense(self.handle, _ptr(rows), rows.shape[0], k, count_thr, idx_offset,
                                                       _ptr(idx), _ptr(sim), _ptr(mom), _stream()))
        self.dense_fallbacks += 1
        return idx, sim, mom

    def bank_status(self) -> None:
        """Synchronise and raise ``TVCError(TVC_E_OVERFLOW)`` if the last search dropped candidates."""
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_bank_status(self.handle, _stream()))

    def bank_gather(self, idx: torch.Tensor, idx_offset: int = 0, bank: str = DEFAULT_BANK) -> torch.Tensor:
        idx = _require_cuda(idx, torch.int32, "idx")
        n = idx.numel()
        with self._lock, torch.cuda.device(self.device):
            b = self._select(bank)
            out = torch.empty((n, b["dim"]), dtype=torch.float32, device=self.device)
            self._check(self.lib.tvc_bank_gather(self.handle, _ptr(idx), n, idx_offset, _ptr(out), _stream()))
        return out.view(*idx.shape, b["dim"])

    def topk_merge(self, idx_parts, sim_parts, feat_parts=None, mom_parts=None):
        """parts [W, M, k] (+ feat [W, M, kf, D], mom [W, M, 4]) -> merged (idx, sim, feat, mom)."""
        idx_parts = _require_cuda(idx_parts, torch.int32, "idx_parts")
        sim_parts = _require_cuda(sim_parts, torch.float32, "sim_parts")
        W, M, k = idx_parts.shape
        kf, D = (feat_parts.shape[2], feat_parts.shape[3]) if feat_parts is not None else (0, 0)
        idx = torch.empty((M, k), dtype=torch.int32, device=self.device)
        sim = torch.empty((M, k), dtype=torch.float32, device=self.device)
        feat = torch.empty((M, kf, D), dtype=torch.float32, device=self.device) if feat_parts is not None else None
        mom = torch.empty((M, 4), dtype=torch.float32, device=self.device) if mom_parts is not None else None
        if feat_parts is not None:
            feat_parts = _require_cuda(feat_parts, torch.float32, "feat_parts")
        if mom_parts is not None:
            mom_parts = _require_cuda(mom_parts, torch.float32, "mom_parts")
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_topk_merge(self.handle, _ptr(idx_parts), _ptr(sim_parts), _ptr(feat_parts),
                                                _ptr(mom_parts), W, M, k, kf, D, _ptr(idx), _ptr(sim), _ptr(feat),
                                                _ptr(mom), _stream()))
        return idx, sim, feat, mom

    # ---- consistency ---------------------------------------------------
    def consistency(self, img: torch.Tensor, txt: torch.Tensor, cfg: ConsistencyConfig,
                    ref_idx: Optional[torch.Tensor] = None, ref_sim: Optional[torch.Tensor] = None,
                    ref_feat: Optional[torch.Tensor] = None) -> torch.Tensor:
        """img [B, D], txt [B, N+1, D]; optional refs for the text rows:
        ref_idx / ref_sim [B*(N+1), ks], ref_feat [B*(N+1), kf, D].  Returns the
        record tensor fp32 [B, rec_stride(N)] (layout: include/tvc.h)."""
        img = _require_cuda(img, torch.float32, "img")
        txt = _require_cuda(txt, torch.float32, "txt")
        B, N1, D = txt.shape
        N = N1 - 1
        ks = kf = 0
        if ref_idx is not None:
            ref_idx = _require_cuda(ref_idx, torch.int32, "ref_idx")
            ref_sim = _require_cuda(ref_sim, torch.float32, "ref_sim")
            ref_feat = _require_cuda(ref_feat, torch.float32, "ref_feat")
            ks, kf = ref_idx.shape[-1], ref_feat.shape[-2]
        rec = torch.empty((B, _lib.rec_stride(N)), dtype=torch.float32, device=self.device)
        p = cfg.to_c()
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_consistency(self.handle, _ptr(img), _ptr(txt), B, N, D, _ptr(ref_idx),
                                                 _ptr(ref_sim), _ptr(ref_feat), ks, kf, C.byref(p), _ptr(rec), _stream()))
        return rec

    def detect_embeddings(self, img: torch.Tensor, txt: torch.Tensor, cfg: ConsistencyConfig,
                          use_bank: bool = True, robust: bool = False, bank: str = DEFAULT_BANK) -> torch.Tensor:
        """Bank search of the text rows -> gather -> consistency; all on the current stream.
        ``robust=False``: no host synchronisation, the caller checks ``bank_status()`` later;
        ``robust=True``: synchronises once and falls back to the brute-force search on overflow."""
        B, N1, D = txt.shape
        if use_bank and self.bank_size(bank) > 0:
            k = max(cfg.search_k, cfg.reference_count)
            search = self.bank_search_robust if robust else self.bank_search
            idx, sim, _ = search(txt.reshape(B * N1, D), k, cfg.similarity_threshold, want_moments=False, bank=bank)
            kf = cfg.reference_count
            feat = self.bank_gather(idx[:, :kf].contiguous(), bank=bank)
            return self.consistency(img, txt, cfg, idx, sim, feat)
        return self.consistency(img, txt, cfg)

    # ---- building blocks (parity tests / profiling) ----------------------
    def gemm(self, a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = 0,
             out: Optional[torch.Tensor] = None, k: Optional[int] = None) -> torch.Tensor:
        """out[j, i] = sum_k a[i, k] b[j, k] (+ bias[i]); a, b bf16.  ``k``: multiply only the first k columns
        (the operands' row strides stay their full widths: padded leading dimensions)."""
        a = _require_cuda(a, torch.bfloat16, "a")
        b = _require_cuda(b, torch.bfloat16, "b")
        I, lda = a.shape
        J, ldb = b.shape
        K = k if k is not None else lda
        if k is None and lda != ldb:
            raise ValueError("a and b must have the same number of columns (or pass k)")
        if out is None:
            out = torch.empty((J, I), dtype=torch.float32 if epilogue in (0, 3) else torch.bfloat16, device=self.device)
            if epilogue == 3:
                out.zero_()
        if bias is not None:
            bias = _require_cuda(bias, torch.float32, "bias")
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_gemm_bf16(self.handle, _ptr(a), _ptr(b), _ptr(bias), _ptr(out), I, J, K, lda, ldb,
                                               out.shape[1], epilogue, _stream()))
        return out

    def attention(self, qkv: torch.Tensor, n_seq: int, seq_len: int, heads: int, causal: bool) -> torch.Tensor:
        qkv = _require_cuda(qkv, torch.bfloat16, "qkv")
        out = torch.empty((qkv.shape[0], heads * 64), dtype=torch.bfloat16, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_attention(self.handle, _ptr(qkv), _ptr(out), n_seq, seq_len, heads,
                                               int(causal), _stream()))
        return out

    def attention_backward(self, qkv: torch.Tensor, dout: torch.Tensor, n_seq: int, seq_len: int, heads: int) -> torch.Tensor:
        qkv = _require_cuda(qkv, torch.bfloat16, "qkv")
        dout = _require_cuda(dout, torch.bfloat16, "dout")
        if qkv.shape != (n_seq * seq_len, 3 * heads * 64) or dout.shape != (n_seq * seq_len, heads * 64):
            raise ValueError("attention_backward: qkv [rows, 3*heads*64] and dout [rows, heads*64] expected")
        dqkv = torch.empty_like(qkv)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_attention_backward(self.handle, _ptr(qkv), _ptr(dout), _ptr(dqkv), n_seq, seq_len,
                                                        heads, _stream()))
        return dqkv

    def layernorm_backward(self, x: torch.Tensor, dy: torch.Tensor, g: torch.Tensor,
                           dres: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _require_cuda(x, torch.float32, "x")
        dy = _require_cuda(dy, torch.bfloat16, "dy")
        g = _require_cuda(g, torch.float32, "g")
        if dres is not None:
            dres = _require_cuda(dres, torch.float32, "dres")
        dx = torch.empty_like(x)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_layernorm_backward(self.handle, _ptr(x), _ptr(dy), _ptr(g), _ptr(dres), _ptr(dx),
                                                        x.shape[0], x.shape[1], _stream()))
        return dx

    def preprocess_images(self, images: torch.Tensor, size: int, mean, std, bicubic: bool = True,
                          keep_aspect: bool = True) -> torch.Tensor:
        """images fp32 [n, 3, H, W] in [0, 1] (on the GPU) -> fp32 [n, 3, size, size]: antialiased resize (+ centre crop)
        and (x - mean) / std in one kernel (``tvc_preprocess_images``)."""
        images = _require_cuda(images, torch.float32, "images")
        n, _, H, W = images.shape
        out = torch.empty((n, 3, size, size), dtype=torch.float32, device=self.device)
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_preprocess_images(self.handle, _ptr(images), n, H, W, size, int(bicubic), int(keep_aspect),
                                                       m3, s3, _ptr(out), _stream()))
        return out

    def gemm_f32(self, w: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = 0,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out[j, i] (op)= sum_k x[j, k] w[i, k] + bias[i] on the exact-f32 matrix instruction (fp32-grade mode)."""
        w = _require_cuda(w, torch.float32, "w")
        x = _require_cuda(x, torch.float32, "x")
        if w.shape[1] != x.shape[1]:
            raise ValueError("w [I, K] and x [J, K] expected")
        if out is None:
            out = torch.zeros((x.shape[0], w.shape[0]), dtype=torch.float32, device=self.device)
        if bias is not None:
            bias = _require_cuda(bias, torch.float32, "bias")
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_gemm_f32(self.handle, _ptr(w), _ptr(x), _ptr(bias), _ptr(out), w.shape[0], x.shape[0],
                                              w.shape[1], out.shape[1], epilogue, _stream()))
        return out

    def attention_f32(self, qkv: torch.Tensor, n_seq: int, seq_len: int, heads: int, causal: bool) -> torch.Tensor:
        qkv = _require_cuda(qkv, torch.float32, "qkv")
        out = torch.empty((qkv.shape[0], heads * 64), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_attention_f32(self.handle, _ptr(qkv), _ptr(out), n_seq, seq_len, heads, int(causal),
                                                   _stream()))
        return out

    def gemm_split(self, w: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
        """out [J, I] = x [J, K] w [I, K]^T + bias on hi | lo bf16 planes, three MFMA products per element (split-bf16 mode)."""
        w = _require_cuda(w, torch.float32, "w")
        x = _require_cuda(x, torch.float32, "x")
        if w.shape[1] != x.shape[1]:
            raise ValueError("w [I, K] and x [J, K] expected")
        ld = (w.shape[0] + 3) // 4 * 4
        out = torch.zeros((x.shape[0], ld), dtype=torch.float32, device=self.device)
        if bias is not None:
            bias = _require_cuda(bias, torch.float32, "bias")
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_gemm_split(self.handle, _ptr(w), _ptr(x), _ptr(bias), _ptr(out), w.shape[0], x.shape[0],
                                                w.shape[1], ld, _stream()))
        return out[:, :w.shape[0]]

    def attention_split(self, qkv: torch.Tensor, n_seq: int, seq_len: int, heads: int, causal: bool,
                        starts: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 qkv rows -> fp32 attention output (hi + lo of the planes the split-bf16 tower's out-projection reads)."""
        qkv = _require_cuda(qkv, torch.float32, "qkv")
        w = heads * 64
        planes = torch.zeros((qkv.shape[0], 2 * w), dtype=torch.bfloat16, device=self.device)
        if starts is not None:
            starts = _require_cuda(starts, torch.int32, "starts")
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_attention_split(self.handle, _ptr(qkv), _ptr(planes), _ptr(starts), n_seq, seq_len, heads,
                                                     int(causal), _stream()))
        return planes[:, :w].float() + planes[:, w:].float()

    def layernorm(self, x: torch.Tensor, g: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        x = _require_cuda(x, torch.float32, "x")
        g = _require_cuda(g, torch.float32, "g")
        b = _require_cuda(b, torch.float32, "b")
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            self._check(self.lib.tvc_layernorm(self.handle, _ptr(x), _ptr(g), _ptr(b), _ptr(out), x.shape[0],
                                               x.shape[1], _stream()))
        return out

    PROF_CATEGORIES = ("gemm", "attention", "bank", "rowops")

    def profile_begin(self) -> None:
        self._check(self.lib.tvc_profile_begin(self.handle))

    def profile_end(self) -> Dict[str, Dict[str, float]]:
        """Per kernel category: summed kernel ms (HIP events on the launch stream),
        summed algorithmic work (FLOPs; bytes for row ops) and launch count."""
        ms = (C.c_double * 4)()
        work = (C.c_double * 4)()
        n = (C.c_int64 * 4)()
        big = (C.c_double * 3)()
        self._check(self.lib.tvc_profile_end(self.handle, ms, work, n, big))
        out = {c: {"ms": ms[i], "work": work[i], "launches": int(n[i])} for i, c in enumerate(self.PROF_CATEGORIES)}
        # the GEMM launches of >= 64 tiles (the ring kernels): compulsory HBM bytes, count, ms
        out["gemm"].update({"big_bytes": big[0], "big_launches": int(big[1]), "big_ms": big[2]})
        return out

    def set_option(self, option: int, value: int) -> None:
        """``_lib.TVC_OPT_*`` (e.g. text packing on / off)."""
        self._check(self.lib.tvc_set_option(self.handle, option, value))

    def workspace_bytes(self) -> int:
        return int(self.lib.tvc_workspace_bytes(self.handle))
