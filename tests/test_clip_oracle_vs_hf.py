"""CPU: the CLIP tower restatement (oracle/clip_oracle.py) against
``transformers.CLIPModel`` built from a LOCAL config (no download), shared
random weights.  This is the cross-check behind the oracle's "parity unpinned"
note: the reference ships neither the wrapper nor weights nor tests at this
boundary, so the published architecture as implemented by HF is the anchor."""
import pytest
import torch

from oracle import clip_oracle


def _hf(width_v=128, width_t=64):
    from transformers import CLIPConfig, CLIPModel
    cfg = CLIPConfig(
        vision_config=dict(hidden_size=width_v, intermediate_size=2 * width_v, num_hidden_layers=2,
                           num_attention_heads=2, image_size=32, patch_size=8, hidden_act="quick_gelu"),
        text_config=dict(hidden_size=width_t, intermediate_size=2 * width_t, num_hidden_layers=2,
                         num_attention_heads=1, vocab_size=49408, max_position_embeddings=77,
                         hidden_act="quick_gelu", eos_token_id=49407, bos_token_id=49406, pad_token_id=0),
        projection_dim=64)
    torch.manual_seed(0)
    m = CLIPModel(cfg).eval()
    with torch.no_grad():       # HF zero-inits biases: randomise so that they are exercised
        for n, p in m.named_parameters():
            if n.endswith("bias") or "layer_norm" in n or "layrnorm" in n:
                p.add_(0.05 * torch.randn_like(p))
    return m


def test_vision_tower_matches_hf():
    m = _hf()
    vw, _ = clip_oracle.from_hf_state_dict(m.state_dict(), 2, 2)
    x = torch.randn(3, 3, 32, 32)
    with torch.no_grad():
        want = m.get_image_features(pixel_values=x)
        want = getattr(want, "pooler_output", want)
        got = clip_oracle.vision_forward(vw, x, heads=2, patch=8, normalize=False)
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4), (got - want).abs().max()


def test_text_tower_matches_hf():
    m = _hf()
    _, tw = clip_oracle.from_hf_state_dict(m.state_dict(), 2, 2)
    tok = torch.zeros((4, 77), dtype=torch.long)
    for i, L in enumerate((3, 9, 20, 75)):
        tok[i, 0] = 49406
        tok[i, 1:1 + L] = torch.randint(1, 49405, (L,))
        tok[i, 1 + L] = 49407
    with torch.no_grad():
        want = m.get_text_features(input_ids=tok, attention_mask=(tok != 0).long() | 1)
        want = getattr(want, "pooler_output", want)
        got = clip_oracle.text_forward(tw, tok, heads=1, normalize=False)
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4), (got - want).abs().max()


def test_causality_makes_padding_irrelevant():
    """Tokens after EOT cannot influence the pooled output (causal mask): the
    property that lets the HIP text tower skip them."""
    m = _hf()
    _, tw = clip_oracle.from_hf_state_dict(m.state_dict(), 2, 2)
    tok = torch.zeros((1, 77), dtype=torch.long)
    tok[0, 0] = 49406
    tok[0, 1:6] = torch.tensor([5, 6, 7, 8, 9])
    tok[0, 6] = 49407
    tok2 = tok.clone()
    tok2[0, 7:] = torch.randint(1, 1000, (70,))
    with torch.no_grad():
        a = clip_oracle.text_forward(tw, tok, heads=1)
        b = clip_oracle.text_forward(tw, tok2, heads=1)
    assert torch.allclose(a, b, atol=1e-6)
