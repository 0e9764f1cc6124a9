"""PGD-perturbed evaluation inputs (TEST INFRASTRUCTURE ONLY).

Restates ``/root/reference/src/attacks/pgd_attack.py:406-523`` (``_batch_pgd_attack``)
on the oracle's CPU tower with autograd, including the reference's quirks:
random start U[-eps, eps] (:437-442), momentum 0.9 on the per-sample
L1-normalised gradient (:505-507), the untargeted loss ``mean cos(adv, text)``
stepped with ``+ alpha * sign(grad)`` (:492,514), projection to the eps ball
(:517-518) and the clamp of the NORMALISED tensor to [0, 1] (:34-35,519-520).
Defaults eps 8/255, alpha 2/255, 10 steps, seed 42 (:22-29,80).
"""
from __future__ import annotations

import torch

from . import clip_oracle


def pgd_images(vision_w, text_w, images: torch.Tensor, tokens: torch.Tensor, v_heads: int, t_heads: int,
               patch: int, eps: float = 8 / 255, alpha: float = 2 / 255, steps: int = 10,
               momentum: float = 0.9, seed: int = 42, clip_min: float = 0.0, clip_max: float = 1.0) -> torch.Tensor:
    """images [Q,3,S,S] (normalised), tokens [Q, ctx] (the original texts) -> adversarial images."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        tf = clip_oracle.text_forward(text_w, tokens, t_heads)            # unit rows (:424-425)
    adv = images.clone()
    if steps > 1:
        noise = (torch.rand(adv.shape, generator=gen) * 2 - 1) * eps
        adv = torch.clamp(adv + noise, clip_min, clip_max)
    mom = torch.zeros_like(adv)
    for _ in range(steps):
        adv.requires_grad_(True)
        f = clip_oracle.vision_forward(vision_w, adv, v_heads, patch)
        loss = torch.nn.functional.cosine_similarity(f, tf, dim=-1).mean()
        grad, = torch.autograd.grad(loss, adv)
        mom = momentum * mom + grad / grad.abs().sum(dim=(1, 2, 3), keepdim=True)
        with torch.no_grad():
            adv = adv + alpha * mom.sign()
            delta = torch.clamp(adv - images, -eps, eps)
            adv = torch.clamp(images + delta, clip_min, clip_max)
    return adv.detach()
