"""``CLIPModel`` / ``CLIPConfig``: mirror of the reference's (absent)
``src/models/clip_model.py`` wrapper, reconstructed from its call sites
(SURVEY.md section 8b): ``CLIPConfig(model_name, device, batch_size, normalize)``
(``src/retrieval.py:356-361``), ``encode_image`` / ``encode_text`` /
``encode_image_tensor(x, requires_grad)`` (``src/detector.py:626``),
``get_text_image_similarity(text, image)`` (``src/detector.py:461``),
``preprocess`` (``src/attacks/hubness_attack.py:223``), ``tokenize``
(``src/attacks/cw_attack.py:656``), ``.model`` / ``.to`` / ``.eval``.

Both towers run as HIP kernels behind ``include/tvc.h``; nothing here computes
on the CPU except image decoding and tokenisation.
"""
from __future__ import annotations

import hashlib
import re
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import synth
from .arch import ClipArch, get_arch
from .engine import TVCEngine

# OpenAI-CLIP preprocessing constants (the `preprocess` the wrapper exposes)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class CLIPConfig:
    model_name: str = "ViT-B/32"          # src/retrieval.py:357, src/detector.py:175
    device: str = "cuda"
    batch_size: int = 256                 # src/retrieval.py:359
    normalize: bool = True                # src/retrieval.py:360
    # --- additions (no network in this environment) ---------------------
    weights: Optional[str] = None         # None/"random": seeded random init; else a .safetensors
    #                                       file holding a transformers.CLIPModel state dict
    seed: int = 0
    tokenizer_dir: Optional[str] = None   # directory with CLIP BPE vocab.json + merges.txt
    precision: str = "bf16"               # "split": fp32-grade towers at ~1/3 of the bf16 rate (scores within 1e-4 end to end);
                                          # "fp32": the exact-f32 reference mode (~1e-6; ~15x slower)


class HashTokenizer:
    """Deterministic stand-in for the CLIP BPE tokenizer when no vocab files are
    available (no network): lower-cased word / punctuation pieces are hashed into
    the id range [1, 49405]; SOT / EOT / padding follow the CLIP layout."""

    _PIECES = re.compile(r"[a-z0-9]+|[^\sa-z0-9]")

    def __init__(self, ctx: int = 77):
        self.ctx = ctx
        self._ids: Dict[str, int] = {}          # piece -> id (the hash is computed once per distinct piece)

    def _id(self, piece: str) -> int:
        v = self._ids.get(piece)
        if v is None:
            v = 1 + int.from_bytes(hashlib.blake2s(piece.encode(), digest_size=4).digest(), "little") % (synth.SOT - 1)
            if len(self._ids) < 1_000_000:
                self._ids[piece] = v
        return v

    def __call__(self, texts: Sequence[str]) -> torch.Tensor:
        out = np.zeros((len(texts), self.ctx), dtype=np.int32)
        out[:, 0] = synth.SOT
        find, ident, cap = self._PIECES.findall, self._id, self.ctx - 2
        for i, t in enumerate(texts):
            ids = [ident(p) for p in find(t.lower())[:cap]]
            m = len(ids)
            out[i, 1:1 + m] = ids
            out[i, 1 + m] = synth.EOT
        return torch.from_numpy(out)


class BPETokenizer:
    """CLIP BPE through ``transformers.CLIPTokenizer`` on local vocab files."""

    def __init__(self, directory: str, ctx: int = 77):
        from transformers import CLIPTokenizer
        d = Path(directory)
        self.tok = CLIPTokenizer(vocab_file=str(d / "vocab.json"), merges_file=str(d / "merges.txt"))
        self.ctx = ctx

    def __call__(self, texts: Sequence[str]) -> torch.Tensor:
        enc = self.tok(list(texts), padding="max_length", truncation=True, max_length=self.ctx, return_tensors="np")
        ids = enc["input_ids"].astype(np.int32)
        # CLIP pads with 0 after EOT (argmax pooling needs EOT to be the largest id)
        eot = ids.argmax(1)
        for i, e in enumerate(eot):
            ids[i, e + 1:] = 0
        return torch.from_numpy(ids)


def weights_from_hf_state_dict(sd: Dict[str, torch.Tensor], arch: ClipArch):
    """``transformers.CLIPModel`` state dict -> (vision, text) weight dicts of the
    layout documented in include/tvc.h (q/k/v concatenated, conv flattened)."""
    def layers(prefix: str, n: int):
        out = []
        for i in range(n):
            p = f"{prefix}.encoder.layers.{i}."
            out.append({
                "ln1_g": sd[p + "layer_norm1.weight"], "ln1_b": sd[p + "layer_norm1.bias"],
                "wqkv": torch.cat([sd[p + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0),
                "bqkv": torch.cat([sd[p + f"self_attn.{x}_proj.bias"] for x in "qkv"], 0),
                "wo": sd[p + "self_attn.out_proj.weight"], "bo": sd[p + "self_attn.out_proj.bias"],
                "ln2_g": sd[p + "layer_norm2.weight"], "ln2_b": sd[p + "layer_norm2.bias"],
                "w1": sd[p + "mlp.fc1.weight"], "b1": sd[p + "mlp.fc1.bias"],
                "w2": sd[p + "mlp.fc2.weight"], "b2": sd[p + "mlp.fc2.bias"],
            })
        return out

    v, t = "vision_model", "text_model"
    pw = sd[f"{v}.embeddings.patch_embedding.weight"]
    vision = {
        "patch_w": pw.reshape(pw.shape[0], -1), "cls": sd[f"{v}.embeddings.class_embedding"],
        "pos": sd[f"{v}.embeddings.position_embedding.weight"],
        "ln_pre_g": sd[f"{v}.pre_layrnorm.weight"], "ln_pre_b": sd[f"{v}.pre_layrnorm.bias"],
        "layers": layers(v, arch.vision.layers),
        "ln_post_g": sd[f"{v}.post_layernorm.weight"], "ln_post_b": sd[f"{v}.post_layernorm.bias"],
        "proj": sd["visual_projection.weight"],
    }
    text = {
        "tok_emb": sd[f"{t}.embeddings.token_embedding.weight"],
        "pos": sd[f"{t}.embeddings.position_embedding.weight"],
        "layers": layers(t, arch.text.layers),
        "ln_final_g": sd[f"{t}.final_layer_norm.weight"], "ln_final_b": sd[f"{t}.final_layer_norm.bias"],
        "proj": sd["text_projection.weight"],
    }
    return vision, text


class _EncodeImageFn(torch.autograd.Function):
    """pixels -> embedding with the HIP forward / input-gradient kernels behind torch.autograd."""

    @staticmethod
    def forward(ctx, x, engine, normalize):
        xc = x.detach().to(torch.float32).contiguous()
        out = engine.encode_image_grad(xc, normalize)
        ctx.engine, ctx.generation, ctx.in_dtype = engine, engine._grad_generation, x.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.engine.encode_image_backward(grad_out.contiguous(), ctx.generation)
        return g.to(ctx.in_dtype), None, None


class CLIPModel:
    """The encoder object the detector / retriever / runners are handed."""

    def __init__(self, config: Optional[CLIPConfig] = None, weights: Optional[tuple] = None):
        self.config = config or CLIPConfig()
        self.arch: ClipArch = get_arch(self.config.model_name)
        dev = self.config.device
        self.device = torch.device("cuda:0" if dev in ("cuda", "auto") else dev)
        if weights is None:
            if self.config.weights in (None, "random"):
                weights = synth.make_clip_weights(self.arch, self.config.seed)
            else:
                from safetensors.torch import load_file
                weights = weights_from_hf_state_dict(load_file(self.config.weights), self.arch)
        self.engine = TVCEngine(self.arch, weights[0], weights[1], device=str(self.device),
                                precision=self.config.precision)
        self.tokenizer = (BPETokenizer(self.config.tokenizer_dir, self.arch.ctx) if self.config.tokenizer_dir
                          else HashTokenizer(self.arch.ctx))
        self.model = self          # `.model` is handed to nn.DataParallel by attacks (out of scope)

    # -- module-ish no-ops the runners call (experiments/runners/run_attack.py:55-56)
    def to(self, device):
        if torch.device(device).type != "cuda":
            raise ValueError("the TVC encoders only run on the GPU (no CPU fallback)")
        return self

    def eval(self):
        return self

    # -- host-side helpers ------------------------------------------------
    def tokenize(self, texts: Union[str, Sequence[str]]) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        return self.tokenizer(texts).long()

    def preprocess(self, image) -> torch.Tensor:
        """PIL image -> fp32 [3, S, S]: bicubic resize of the short side, centre
        crop, CLIP mean/std."""
        from PIL import Image
        S = self.arch.image_size
        img = image.convert("RGB")
        w, h = img.size
        s = S / min(w, h)
        img = img.resize((max(S, round(w * s)), max(S, round(h * s))), Image.BICUBIC)
        w, h = img.size
        l, t = (w - S) // 2, (h - S) // 2
        img = img.crop((l, t, l + S, t + S))
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
        std = torch.tensor(CLIP_STD).view(3, 1, 1)
        return ((x - mean) / std).contiguous()

    def preprocess_tensor(self, images01: torch.Tensor) -> torch.Tensor:
        """Device images fp32 [n, 3, H, W] with values in [0, 1] -> [n, 3, S, S]: the same bicubic resize of the short side,
        centre crop and CLIP mean / std as ``preprocess``, in one HIP kernel (``tvc_preprocess_images``) -- generated
        references never leave the GPU.  (``preprocess`` works on 8-bit PIL pixels: the two agree to the 8-bit rounding.)"""
        return self.engine.preprocess_images(images01.to(self.device, torch.float32), self.arch.image_size, CLIP_MEAN, CLIP_STD,
                                             bicubic=True, keep_aspect=True)

    def _images_to_device(self, images):
        if isinstance(images, torch.Tensor):
            x = images if images.dim() == 4 else images.unsqueeze(0)
            return x.to(self.device, torch.float32), images.is_cuda
        if not isinstance(images, (list, tuple)):
            images = [images]
        x = torch.stack([im if isinstance(im, torch.Tensor) else self.preprocess(im) for im in images])
        return x.to(self.device, torch.float32), False

    # -- encoders -----------------------------------------------------------
    def encode_image(self, images, normalize: Optional[bool] = None) -> torch.Tensor:
        """Tensor [B,3,S,S] (or [3,S,S]) or list of PIL images -> [B, D].  Device
        tensors in -> device tensor out; host inputs -> CPU tensor (callers such as
        src/retrieval.py:413 call ``.numpy()`` on it)."""
        normalize = self.config.normalize if normalize is None else normalize
        x, on_dev = self._images_to_device(images)
        outs = [self.engine.encode_image(x[i:i + self.config.batch_size], normalize)
                for i in range(0, x.shape[0], self.config.batch_size)]
        out = torch.cat(outs) if len(outs) != 1 else outs[0]
        return out if on_dev else out.cpu()

    def encode_image_beside(self, x: torch.Tensor, normalize: bool = True):
        """The image tower of a batch enqueued on the model's SIDE stream, so that the text tower the caller enqueues next
        (on the current stream) runs beside it -- the two towers of a query are independent until the cosines, and for
        one query each is a chain of ~300 latency-bound launches.  Returns ``join``: call it before the first consumer;
        it makes the current stream wait for the side stream and returns the rows ``[B, D]``."""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(self.device)
        side = self._side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fi = self.engine.encode_image(x, normalize)

        def join() -> torch.Tensor:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(side)
            fi.record_stream(cur)
            return fi
        return join

    def encode_image_tensor(self, x: torch.Tensor, requires_grad: bool = False) -> torch.Tensor:
        """Tensor in, tensor out (src/detector.py:626).  ``requires_grad=True`` (src/attacks/pgd_attack.py:254,459,480,
        src/attacks/hubness_attack.py:586): the result carries an autograd node whose backward is the HIP input-gradient
        pass (``tvc_encode_image_backward``) -- ``loss.backward()`` fills ``x.grad`` as it does in the reference.  Only
        the pixels receive a gradient (the weights are frozen); one differentiable batch at a time per model."""
        if not requires_grad:
            return self.encode_image(x)
        if not x.is_cuda:
            raise ValueError("encode_image_tensor(requires_grad=True) needs a device tensor (no CPU fallback)")
        if x.dim() == 3:
            x = x.unsqueeze(0)
        return _EncodeImageFn.apply(x, self.engine, bool(self.config.normalize))

    def encode_tokens(self, tokens: torch.Tensor, normalize: Optional[bool] = None, group: int = 0) -> torch.Tensor:
        """int [T, ctx] -> device tensor [T, D].  ``group`` = N + 1 when the rows are consecutive
        (original, N variants) groups (see ``TVCEngine.encode_text``)."""
        normalize = self.config.normalize if normalize is None else normalize
        return self.engine.encode_text(tokens.to(self.device, torch.int32), normalize, group=group)

    def encode_text(self, texts: Union[str, Sequence[str], torch.Tensor], normalize: Optional[bool] = None) -> torch.Tensor:
        """list[str] -> CPU tensor [B, D] (src/retrieval.py:551-554 calls ``.numpy()``);
        a token tensor on the GPU -> device tensor."""
        if isinstance(texts, torch.Tensor):
            out = self.encode_tokens(texts, normalize)
            return out if texts.is_cuda else out.cpu()
        return self.encode_tokens(self.tokenize(texts), normalize).cpu()

    def get_text_image_similarity(self, text, image) -> torch.Tensor:
        """0-d tensor cos(image, text) (src/detector.py:461 calls ``.item()``)."""
        x, _ = self._images_to_device(image)
        fi = self.engine.encode_image(x[:1], True)
        ft = self.encode_tokens(self.tokenize(text) if not isinstance(text, torch.Tensor) else text, True)
        from .engine import ConsistencyConfig
        rec = self.engine.consistency(fi[:1], ft[:1].unsqueeze(0), ConsistencyConfig())   # K4 kernel; word 0 = cos
        return rec[0, 0]
