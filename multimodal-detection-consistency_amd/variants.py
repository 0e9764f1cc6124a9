"""Text variants are INPUTS to the hot path.  The reference produces them with an
LLM / WordNet augmenter (``src/text_augment.py``, ``experiments/defenses/
text_variants.py``) that is out of scope.
Callers pass their own generator (any callable ``text -> list[str]``); the
default below is a deterministic prompt-template generator in the spirit of the
reference's rule-based branch (``experiments/defenses/text_variants.py:305-315``)
so that the string API works stand-alone."""
from __future__ import annotations

from typing import Callable, List

_TEMPLATES = (
    "a photo of {t}", "an image showing {t}", "a picture of {t}", "a scene with {t}",
    "a view of {t}", "an image featuring {t}", "a photograph showing {t}", "a snapshot of {t}",
    "a depiction of {t}", "a representation of {t}", "a close-up of {t}", "a rendering of {t}",
)


class TemplateVariantGenerator:
    def __init__(self, num_variants: int = 5):
        self.num_variants = num_variants

    def generate_variants(self, text: str) -> List[str]:
        out = []
        for tpl in _TEMPLATES:
            v = tpl.format(t=text)
            if v != text:
                out.append(v)
            if len(out) >= self.num_variants:
                break
        return out

    __call__ = generate_variants


def as_generator(obj, num_variants: int) -> Callable[[str], List[str]]:
    """Accept ``None`` (default templates), a callable, or an object exposing
    ``generate_variants`` / ``augment`` (the two names the reference uses:
    src/pipeline.py:430 vs src/text_augment.py:491)."""
    if obj is None:
        return TemplateVariantGenerator(num_variants)
    for name in ("generate_variants", "augment"):
        if hasattr(obj, name):
            fn = getattr(obj, name)
            return lambda t: list(fn(t))[:num_variants]
    if callable(obj):
        return lambda t: list(obj(t))[:num_variants]
    raise TypeError("variant generator must be callable or expose generate_variants()/augment()")


def batch_variants(obj, num_variants: int, texts) -> List[List[str]]:
    """Variants of many texts: ONE call when the generator can batch (``batch_generate_variants``,
    experiments/defenses/text_variants.py:383 -- the CLIP-filtered generator then encodes all candidates
    in one launch), otherwise one call per text."""
    if obj is not None and hasattr(obj, "batch_generate_variants"):
        return [list(v)[:num_variants] for v in obj.batch_generate_variants(list(texts))]
    gen = as_generator(obj, num_variants)
    return [gen(t) for t in texts]
