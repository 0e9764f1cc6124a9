"""Rule-based text variants + CLIP semantic filter.

Mirror of ``experiments/defenses/text_variants.py`` (``TextVariantGenerator``, ``TextVariantConfig``):
synonym replacement from a small dictionary (``:110-135``), paraphrase templates + simple descriptive
rewrites (``:137-157,305-343``), adjacent-word reordering for short texts (``:159-176``), then the
quality filter (``:206-284``): basic checks, the CLIP text-text similarity window
``diversity_threshold < cos(original, variant) < similarity_threshold``, de-duplication, ranking by
similarity (descending), cut to ``variant_count``.

What is different, on purpose:
* the LLM branch (``:178-204``, Qwen) only runs when a ``qwen_model`` with ``generate(prompt=, max_length=,
  temperature=)`` is injected -- text generation is not part of the path;
* the filter encodes ALL candidates of ALL texts in one ``tvc_encode_text`` launch and takes the
  cosines with the K4 kernel (``tvc_consistency`` record words 0 / 12..), instead of one
  ``encode_text([variant])`` + ``.item()`` per candidate (``:256-258``) and a second round for the
  ranking (``:270-273``);
* de-duplication keeps the first occurrence and the ranking is a stable sort (the reference goes
  through ``set()``, ``:229``, whose order is not defined);
* word reordering draws from a seeded ``random.Random`` instead of the global RNG (``:169``).
"""
from __future__ import annotations

import random
import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from .engine import ConsistencyConfig


@dataclass
class TextVariantConfig:
    """experiments/defenses/text_variants.py:17-28 (same field names and defaults)."""
    variant_count: int = 5
    max_length: int = 77
    temperature: float = 0.7
    diversity_threshold: float = 0.1
    similarity_threshold: float = 0.8
    use_synonyms: bool = True
    use_paraphrasing: bool = True
    use_reordering: bool = True
    filter_quality: bool = True
    seed: int = 0                      # addition: the reordering RNG


# experiments/defenses/text_variants.py:345-369 -- the simplified dictionary is DATA of the rule, kept as is
SYNONYMS: Dict[str, List[str]] = {
    "cat": ["feline", "kitten", "kitty"], "dog": ["canine", "puppy", "hound"],
    "car": ["vehicle", "automobile", "auto"], "house": ["home", "building", "residence"],
    "person": ["individual", "human", "people"], "man": ["male", "gentleman", "guy"],
    "woman": ["female", "lady", "girl"], "child": ["kid", "youngster", "youth"],
    "big": ["large", "huge", "enormous"], "small": ["tiny", "little", "mini"],
    "beautiful": ["pretty", "lovely", "gorgeous"], "old": ["elderly", "aged", "ancient"],
    "young": ["youthful", "juvenile", "new"], "red": ["crimson", "scarlet", "cherry"],
    "blue": ["azure", "navy", "cobalt"], "green": ["emerald", "lime", "forest"],
    "happy": ["joyful", "cheerful", "glad"], "sad": ["unhappy", "sorrowful", "melancholy"],
    "fast": ["quick", "rapid", "swift"], "slow": ["sluggish", "gradual", "leisurely"],
}
PARAPHRASE_TEMPLATES = ("a view of {text}", "an image featuring {text}", "a photograph showing {text}",
                        "a snapshot of {text}", "a depiction of {text}", "a representation of {text}")     # :371-381
DESCRIPTIVE = ("a photo of {t}", "an image showing {t}", "a picture of {t}", "a scene with {t}")            # :310-315
CORE_PATTERNS = (r"a photo of (.+)", r"an image (?:showing|of) (.+)", r"a picture of (.+)", r"a scene with (.+)")   # :325-330


class TextVariantGenerator:
    def __init__(self, qwen_model=None, clip_model=None, config: Optional[TextVariantConfig] = None,
                 synonym_dict: Optional[Dict[str, List[str]]] = None):
        self.qwen_model = qwen_model
        self.clip_model = clip_model
        self.config = config or TextVariantConfig()
        self.synonym_dict = synonym_dict if synonym_dict is not None else SYNONYMS
        self.paraphrase_templates = list(PARAPHRASE_TEMPLATES)
        self._rng = random.Random(self.config.seed)

    # ---- candidate rules (host string operations) -------------------------------------------------
    def _generate_synonym_variants(self, text: str) -> List[str]:
        """:110-135: for every word with an entry, up to 3 synonyms, one replacement per variant."""
        out, words, limit = [], text.split(), self.config.max_length * 4
        for i, w in enumerate(words):
            key = w.lower().strip(".,!?;:")
            for syn in self.synonym_dict.get(key, [])[:3]:
                nw = list(words)
                nw[i] = syn.upper() if w.isupper() else syn.capitalize() if w.istitle() else syn
                v = " ".join(nw)
                if v != text and len(v) <= limit:
                    out.append(v)
        return out

    def _simple_paraphrases(self, text: str) -> List[str]:
        """:305-343."""
        limit = self.config.max_length * 4
        out = [v for v in (d.format(t=text) for d in DESCRIPTIVE) if v != text and len(v) <= limit]
        if text.startswith(("a photo of", "an image", "a picture", "a scene")):
            for pat in CORE_PATTERNS:
                m = re.match(pat, text, re.IGNORECASE)
                if m:
                    core = m.group(1).strip()
                    if core:
                        out.append(core)
                    break
        return out

    def _generate_paraphrase_variants(self, text: str) -> List[str]:
        """:137-157."""
        limit = self.config.max_length * 4
        out = [v for v in (t.format(text=text) for t in self.paraphrase_templates) if v != text and len(v) <= limit]
        return out + self._simple_paraphrases(text)

    def _generate_reorder_variants(self, text: str) -> List[str]:
        """:159-176: texts of <= 8 words, min(3, len-1) random adjacent swaps."""
        out, words = [], text.split()
        if len(words) <= 8:
            for _ in range(min(3, len(words) - 1)):
                nw = list(words)
                i = self._rng.randint(0, len(words) - 2)
                nw[i], nw[i + 1] = nw[i + 1], nw[i]
                v = " ".join(nw)
                if v != text:
                    out.append(v)
        return out

    def _generate_llm_variants(self, text: str) -> List[str]:
        """:178-204 -- only with an injected language model."""
        if self.qwen_model is None:
            return []
        out = []
        for prompt in (f"请改写以下句子，保持原意不变：{text}", f"用不同的表达方式重新描述：{text}"):
            try:
                resp = self.qwen_model.generate(prompt=prompt, max_length=self.config.max_length,
                                                temperature=self.config.temperature)
            except Exception:
                continue
            for line in (resp or "").strip().split("\n"):
                line = line.strip().strip(".,!?;: \"'")
                if line and line != text and not line.startswith(("请", "用")):
                    out.append(line)
                    break
        return out

    def candidates(self, text: str) -> List[str]:
        """:74-91: rule outputs in the reference's order (synonyms, paraphrases, reorderings, LLM)."""
        c = self.config
        out: List[str] = []
        if c.use_synonyms:
            out += self._generate_synonym_variants(text)
        if c.use_paraphrasing:
            out += self._generate_paraphrase_variants(text)
        if c.use_reordering:
            out += self._generate_reorder_variants(text)
        return out + self._generate_llm_variants(text)

    def _basic_filter(self, variant: str, original: str) -> bool:
        """:236-254."""
        if len(variant) > self.config.max_length * 4 or not variant.strip():
            return False
        if variant.strip().lower() == original.strip().lower():
            return False
        return re.search(r"[a-zA-Z]", variant) is not None

    # ---- CLIP filter + ranking, batched (K2 + K4) ---------------------------------------------------
    def similarities(self, texts: Sequence[str], cands: Sequence[Sequence[str]]) -> List[np.ndarray]:
        """cos(encode_text(original), encode_text(candidate)) for every candidate of every text: ONE text
        encode of all strings, one consistency launch per distinct candidate count."""
        clip = self.clip_model
        flat = list(texts) + [v for c in cands for v in c]
        f = clip.encode_tokens(clip.tokenize(flat), True)                       # device [n + sum, D]
        n = len(texts)
        offs = np.concatenate([[n], n + np.cumsum([len(c) for c in cands])]).astype(np.int64)
        out: List[Optional[np.ndarray]] = [np.zeros(0)] * n
        by_count: Dict[int, List[int]] = {}
        for i, c in enumerate(cands):
            if c:
                by_count.setdefault(len(c), []).append(i)
        cfg = ConsistencyConfig()
        for J, ids in by_count.items():
            rows = torch.as_tensor(np.concatenate([np.arange(offs[i], offs[i] + J) for i in ids]), device=f.device)
            q = f[torch.as_tensor(ids, device=f.device)].contiguous()
            rec = clip.engine.consistency(q, f[rows].view(len(ids), J, -1), cfg).cpu().numpy().astype(np.float64)
            sims = np.concatenate([rec[:, 0:1], rec[:, 12:12 + J - 1]], axis=1)  # cos(original, candidate_j)
            for j, i in enumerate(ids):
                out[i] = sims[j]
        return out

    def batch_generate_variants(self, texts: Sequence[str]) -> List[List[str]]:
        """:383-399, with the filter of all texts in one pass."""
        c = self.config
        cands = [self.candidates(t) for t in texts]
        if not c.filter_quality:
            return [v[:c.variant_count] for v in cands]
        kept = [[v for v in cand if self._basic_filter(v, t)] for t, cand in zip(texts, cands)]
        if self.clip_model is None:
            raise ValueError("filter_quality=True needs a clip_model (the semantic filter is a CLIP text-text cosine)")
        sims = self.similarities(texts, kept)
        out = []
        for cand, s in zip(kept, sims):
            seen, pairs = set(), []
            for v, x in zip(cand, s):
                if c.diversity_threshold < x < c.similarity_threshold and v not in seen:      # :256-261, :229
                    seen.add(v)
                    pairs.append((v, x))
            if len(pairs) > 1:
                pairs.sort(key=lambda p: p[1], reverse=True)                                  # :281 (stable)
            out.append([v for v, _ in pairs][:c.variant_count])
        return out

    def generate_variants(self, text: str) -> List[str]:
        """:63-108."""
        return self.batch_generate_variants([text])[0]

    __call__ = generate_variants

    def evaluate_variant_quality(self, original: str, variants: List[str]) -> Dict[str, Any]:
        """:401-452: similarity to the original, pairwise diversity (1 - cos) among the variants, quality
        score (:454-468).  The pairwise cosines come from ONE all-pairs launch (``tvc_cosine_matrix``)."""
        if not variants:
            return {"message": "no variants"}
        clip = self.clip_model
        f = clip.encode_tokens(clip.tokenize([original] + list(variants)), True)
        from .metrics import SimilarityCalculator
        C = SimilarityCalculator.batch_cosine_similarity(f, f, engine=clip.engine).astype(np.float64)
        sims = C[0, 1:]
        iu = np.triu_indices(len(variants), k=1)
        div = 1.0 - C[1:, 1:][iu]
        quality = float(np.clip((div.mean() if div.size else 0.0) - np.mean(np.abs(sims - 0.7)), 0.0, 1.0))
        return {"variant_count": len(variants),
                "similarity_stats": {"mean": float(sims.mean()), "std": float(sims.std()), "min": float(sims.min()),
                                     "max": float(sims.max())},
                "diversity_stats": {"mean": float(div.mean()) if div.size else 0.0, "std": float(div.std()) if div.size else 0.0},
                "quality_score": quality}
