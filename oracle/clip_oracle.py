"""CPU restatement (PyTorch fp32) of the CLIP towers the hot path encodes with.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

**Parity unpinned** against the reference: the reference calls
``clip_model.encode_image / encode_text / encode_image_tensor`` on a wrapper
(``src/models/clip_model.py``) that is absent from the snapshot
(``/root/reference/.gitignore:52``), over third-party towers
(``clip-by-openai>=1.0`` / ``open-clip-torch>=2.20.0``,
``/root/reference/requirements.txt:8-10``, not vendored) whose weights are
missing too.  This file restates the published OpenAI-CLIP architecture
(ViT with pre-LN blocks, quick-GELU, class token, ``ln_pre``/``ln_post``,
bias-free patch conv and projections; causal text transformer pooled at the
EOT token = ``argmax`` of the ids) and is cross-checked against
``transformers.CLIPModel`` built from a local config in
``tests/test_clip_oracle_vs_hf.py``.  Call sites anchoring the interface:
``/root/reference/src/detector.py:461,626``,
``/root/reference/experiments/defenses/detector.py:238-247``,
``/root/reference/experiments/defenses/retrieval_ref.py:238-244``.

Weight dictionary layout (shared with the product's loader; all fp32 here):

vision: ``patch_w [d, 3*p*p]`` (conv weight flattened c,ky,kx), ``cls [d]``,
``pos [T, d]``, ``ln_pre_g/b``, ``layers[i]{ln1_g, ln1_b, wqkv [3d,d], bqkv,
wo [d,d], bo, ln2_g, ln2_b, w1 [mlp,d], b1, w2 [d,mlp], b2}``, ``ln_post_g/b``,
``proj [D, d]``.
text: ``tok_emb [V, d]``, ``pos [ctx, d]``, ``layers`` as above,
``ln_final_g/b``, ``proj [D, d]``.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

LN_EPS = 1e-5


def _quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def _block(x: torch.Tensor, lw: Dict[str, torch.Tensor], heads: int, causal: bool, act: str = "quick_gelu") -> torch.Tensor:
    """One pre-LN residual attention block.  x: [B, T, d].  act: "quick_gelu" (OpenAI CLIP) or "gelu" (nn.GELU(), the
    exact erf form: OpenCLIP ViT-H/14, the text encoder of Stable Diffusion 2.x)."""
    B, T, d = x.shape
    dh = d // heads
    h = F.layer_norm(x, (d,), lw['ln1_g'], lw['ln1_b'], LN_EPS)
    qkv = h @ lw['wqkv'].t() + lw['bqkv']
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(B, T, heads, dh).transpose(1, 2)
    k = k.view(B, T, heads, dh).transpose(1, 2)
    v = v.view(B, T, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    if causal:
        mask = torch.full((T, T), float('-inf'), dtype=s.dtype, device=s.device).triu(1)
        s = s + mask
    p = s.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, d)
    x = x + o @ lw['wo'].t() + lw['bo']
    h = F.layer_norm(x, (d,), lw['ln2_g'], lw['ln2_b'], LN_EPS)
    u = h @ lw['w1'].t() + lw['b1']
    h = F.gelu(u) if act == "gelu" else _quick_gelu(u)
    x = x + h @ lw['w2'].t() + lw['b2']
    return x


def vision_forward(w: Dict, pixels: torch.Tensor, heads: int, patch: int,
                   normalize: bool = True, act: str = "quick_gelu") -> torch.Tensor:
    """pixels [B, 3, H, W] fp32 -> [B, D] (L2-normalised when ``normalize``)."""
    B, C, H, W = pixels.shape
    gh, gw = H // patch, W // patch
    d = w['cls'].shape[0]
    # stride=patch conv == unfold + GEMM; flatten order (c, ky, kx) as conv weight
    cols = pixels.unfold(2, patch, patch).unfold(3, patch, patch)          # [B,C,gh,gw,p,p]
    cols = cols.permute(0, 2, 3, 1, 4, 5).reshape(B, gh * gw, C * patch * patch)
    x = cols @ w['patch_w'].t()                                            # [B, P, d]
    x = torch.cat([w['cls'].expand(B, 1, d), x], dim=1) + w['pos']
    x = F.layer_norm(x, (d,), w['ln_pre_g'], w['ln_pre_b'], LN_EPS)
    for lw in w['layers']:
        x = _block(x, lw, heads, causal=False, act=act)
    x = F.layer_norm(x[:, 0], (d,), w['ln_post_g'], w['ln_post_b'], LN_EPS)
    x = x @ w['proj'].t()
    if normalize:
        x = x / x.norm(dim=-1, keepdim=True)
    return x


def text_forward(w: Dict, tokens: torch.Tensor, heads: int,
                 normalize: bool = True, act: str = "quick_gelu") -> torch.Tensor:
    """tokens [B, ctx] int -> [B, D].  Pooled at ``tokens.argmax(-1)`` (EOT)."""
    B, T = tokens.shape
    d = w['tok_emb'].shape[1]
    x = w['tok_emb'][tokens.long()] + w['pos'][:T]
    for lw in w['layers']:
        x = _block(x, lw, heads, causal=True, act=act)
    x = F.layer_norm(x, (d,), w['ln_final_g'], w['ln_final_b'], LN_EPS)
    x = x[torch.arange(B), tokens.long().argmax(dim=-1)]
    x = x @ w['proj'].t()
    if normalize:
        x = x / x.norm(dim=-1, keepdim=True)
    return x


def text_hidden(w: Dict, tokens: torch.Tensor, heads: int, act: str = "quick_gelu") -> torch.Tensor:
    """tokens [B, ctx] int -> [B, ctx, d]: ln_final of the hidden state at every position =
    ``transformers.CLIPTextModel(...).last_hidden_state`` (the conditioning of the SD UNet; the reference reaches
    it through the absent ``StableDiffusionModel.generate_image``, /root/reference/src/sd_ref.py:389-412)."""
    B, T = tokens.shape
    d = w['tok_emb'].shape[1]
    x = w['tok_emb'][tokens.long()] + w['pos'][:T]
    for lw in w['layers']:
        x = _block(x, lw, heads, causal=True, act=act)
    return F.layer_norm(x, (d,), w['ln_final_g'], w['ln_final_b'], LN_EPS)


def round_gemm_weights_to_bf16(w: Dict) -> Dict:
    """Copy of ``w`` whose GEMM operands are rounded to bf16 (and back to fp32),
    the exact values the HIP path multiplies with -- isolates kernel arithmetic
    from the weight-storage rounding in parity tests."""
    gemm_keys = {'patch_w', 'proj', 'wqkv', 'wo', 'w1', 'w2'}

    def rnd(t):
        return t.to(torch.bfloat16).to(torch.float32)

    out = {}
    for k, v in w.items():
        if k == 'layers':
            out[k] = [{kk: (rnd(vv) if kk in gemm_keys else vv) for kk, vv in lw.items()} for lw in v]
        else:
            out[k] = rnd(v) if k in gemm_keys else v
    return out


def from_hf_state_dict(sd: Dict[str, torch.Tensor], n_vision_layers: int, n_text_layers: int):
    """Convert a ``transformers.CLIPModel`` state dict into (vision, text) weight
    dicts of the layout above (used only to cross-check this oracle against HF)."""
    def layers(prefix: str, n: int) -> List[Dict]:
        out = []
        for i in range(n):
            p = f'{prefix}.encoder.layers.{i}.'
            out.append({
                'ln1_g': sd[p + 'layer_norm1.weight'], 'ln1_b': sd[p + 'layer_norm1.bias'],
                'wqkv': torch.cat([sd[p + 'self_attn.q_proj.weight'], sd[p + 'self_attn.k_proj.weight'],
                                   sd[p + 'self_attn.v_proj.weight']], 0),
                'bqkv': torch.cat([sd[p + 'self_attn.q_proj.bias'], sd[p + 'self_attn.k_proj.bias'],
                                   sd[p + 'self_attn.v_proj.bias']], 0),
                'wo': sd[p + 'self_attn.out_proj.weight'], 'bo': sd[p + 'self_attn.out_proj.bias'],
                'ln2_g': sd[p + 'layer_norm2.weight'], 'ln2_b': sd[p + 'layer_norm2.bias'],
                'w1': sd[p + 'mlp.fc1.weight'], 'b1': sd[p + 'mlp.fc1.bias'],
                'w2': sd[p + 'mlp.fc2.weight'], 'b2': sd[p + 'mlp.fc2.bias'],
            })
        return out

    v = 'vision_model'
    pw = sd[f'{v}.embeddings.patch_embedding.weight']
    vision = {
        'patch_w': pw.reshape(pw.shape[0], -1),
        'cls': sd[f'{v}.embeddings.class_embedding'],
        'pos': sd[f'{v}.embeddings.position_embedding.weight'],
        'ln_pre_g': sd[f'{v}.pre_layrnorm.weight'], 'ln_pre_b': sd[f'{v}.pre_layrnorm.bias'],
        'layers': layers(v, n_vision_layers),
        'ln_post_g': sd[f'{v}.post_layernorm.weight'], 'ln_post_b': sd[f'{v}.post_layernorm.bias'],
        'proj': sd['visual_projection.weight'],
    }
    t = 'text_model'
    text = {
        'tok_emb': sd[f'{t}.embeddings.token_embedding.weight'],
        'pos': sd[f'{t}.embeddings.position_embedding.weight'],
        'layers': layers(t, n_text_layers),
        'ln_final_g': sd[f'{t}.final_layer_norm.weight'], 'ln_final_b': sd[f'{t}.final_layer_norm.bias'],
        'proj': sd['text_projection.weight'],
    }
    return vision, text
