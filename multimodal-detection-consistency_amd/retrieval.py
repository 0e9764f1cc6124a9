"""Retrieval over the reference bank: mirrors of ``MultiModalRetriever``
(``src/retrieval.py:316``) and ``RetrievalReferenceGenerator``
(``experiments/defenses/retrieval_ref.py:34``).  FAISS ``IndexFlatIP`` /
``index_cpu_to_gpu`` (``src/retrieval.py:100-112``) is replaced by the fused
exact top-k kernel behind ``tvc_bank_search``; IVF / HNSW approximations are not
needed (the search is exact at 1M-10M rows).
"""
from __future__ import annotations

import json
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .clip import CLIPConfig, CLIPModel


@dataclass
class RetrievalConfig:
    """src/retrieval.py:290-314 (fields the path reads)."""
    clip_model: str = "ViT-B/32"
    device: str = "cuda"
    batch_size: int = 256
    top_k: int = 10
    similarity_metric: str = "cosine"     # cosine, dot_product, euclidean
    index_type: str = "faiss"             # accepted for compatibility; the search is always exact
    faiss_index_type: str = "IndexFlatIP"
    n_clusters: int = 100
    enable_cache: bool = True
    cache_dir: Optional[str] = None
    normalize_features: bool = True
    use_gpu_index: bool = True
    bank_dtype: str = "bfloat16"          # "bfloat16" (1.5 GB / 1M x 768) or "float32" (split-bf16 planes)


class MultiModalRetriever:
    def __init__(self, config: Optional[RetrievalConfig] = None, clip_model: Optional[CLIPModel] = None):
        self.config = config or RetrievalConfig()
        self.clip_model = clip_model or CLIPModel(CLIPConfig(model_name=self.config.clip_model,
                                                             device=self.config.device,
                                                             batch_size=self.config.batch_size,
                                                             normalize=self.config.normalize_features))
        self.image_features: Optional[np.ndarray] = None
        self.image_paths: List[str] = []
        self.text_features: Optional[np.ndarray] = None
        self.texts: List[str] = []
        self.retrieval_cache: Dict[str, Tuple[List[str], List[float]]] = {}
        self._bank_is = None        # "image" / "text": which index lives on the engine
        self.bank_name = f"retriever:{id(self):x}"      # this retriever's own bank slot on the (shared) engine

    def __del__(self):          # give the bank slot back to the shared engine
        try:
            self.clip_model.engine.release_bank(self.bank_name)
        except Exception:
            pass

    # -- index construction (src/retrieval.py:372-432, scripts/build_faiss_indices.py:59-158)
    def _set_bank(self, feats: torch.Tensor, kind: str) -> None:
        dt = torch.bfloat16 if self.config.bank_dtype == "bfloat16" else torch.float32
        self.clip_model.engine.set_bank(feats.to(self.clip_model.device, dt), name=self.bank_name)
        self._bank_is = kind

    def build_image_index(self, images: Union[Sequence[str], torch.Tensor], batch_size: Optional[int] = None) -> np.ndarray:
        """paths (PIL-loaded + preprocessed) or a preprocessed tensor [R,3,S,S] ->
        L2-normalised features [R, D]; registers them as the bank."""
        bs = batch_size or self.config.batch_size
        feats = []
        if isinstance(images, torch.Tensor):
            self.image_paths = [f"image_{i}" for i in range(images.shape[0])]
            for i in range(0, images.shape[0], bs):
                feats.append(self.clip_model.engine.encode_image(images[i:i + bs].to(self.clip_model.device), True))
        else:
            from PIL import Image
            self.image_paths = list(images)
            for i in range(0, len(images), bs):
                batch = torch.stack([self.clip_model.preprocess(Image.open(p)) for p in images[i:i + bs]])
                feats.append(self.clip_model.engine.encode_image(batch.to(self.clip_model.device), True))
        f = torch.cat(feats) if feats else torch.empty((0, self.clip_model.arch.embed_dim), device=self.clip_model.device)
        self._set_bank(f, "image")
        self.image_features = f.cpu().numpy()
        self.retrieval_cache.clear()
        return self.image_features

    def set_image_features(self, features: Union[np.ndarray, torch.Tensor], paths: Optional[Sequence[str]] = None) -> None:
        """Register a pre-computed ``features.npy`` bank (rows L2-normalised)."""
        f = torch.as_tensor(features)
        self.image_paths = list(paths) if paths is not None else [f"image_{i}" for i in range(f.shape[0])]
        self._set_bank(f, "image")
        self.image_features = f.float().cpu().numpy() if f.shape[0] <= 200_000 else None
        self.retrieval_cache.clear()

    def build_text_index(self, texts: Sequence[str]) -> np.ndarray:
        self.texts = list(texts)
        f = self.clip_model.encode_tokens(self.clip_model.tokenize(self.texts), True)
        self.text_features = f.cpu().numpy()
        return self.text_features

    # -- queries (src/retrieval.py:527-680) ----------------------------------
    def _search(self, q: torch.Tensor, top_k: int) -> Tuple[np.ndarray, np.ndarray]:
        eng = self.clip_model.engine
        # the reference accepts any top_k (src/retrieval.py:636); the kernel's limit is TVC_MAX_TOPK and the
        # engine raises beyond it -- never a silent truncation.  More than R rows cannot be returned (-1 padded).
        idx, sim, _ = eng.bank_search_robust(q, top_k, want_moments=False, bank=self.bank_name)
        idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
        return idx, sim

    def retrieve_images_by_text(self, query_text: str, top_k: Optional[int] = None) -> Tuple[List[str], List[float]]:
        top_k = top_k or self.config.top_k
        key = f"text2img_{query_text}_{top_k}"
        if self.config.enable_cache and key in self.retrieval_cache:
            return self.retrieval_cache[key]
        if self._bank_is != "image":
            raise ValueError("image index not built")                      # src/retrieval.py:547-548
        q = self.clip_model.encode_tokens(self.clip_model.tokenize(query_text), self.config.normalize_features)
        idx, sim = self._search(q, top_k)
        keep = idx[0] >= 0
        out = ([self.image_paths[i] for i in idx[0][keep]], sim[0][keep].astype(float).tolist())
        if self.config.enable_cache:
            self.retrieval_cache[key] = out
        return out

    def batch_retrieve_images_by_texts(self, query_texts: Sequence[str], top_k: Optional[int] = None):
        top_k = top_k or self.config.top_k
        q = self.clip_model.encode_tokens(self.clip_model.tokenize(list(query_texts)), self.config.normalize_features)
        idx, sim = self._search(q, top_k)
        return [([self.image_paths[i] for i in r[r >= 0]], s[r >= 0].astype(float).tolist()) for r, s in zip(idx, sim)]

    def batch_retrieve_images_by_features(self, text_features: torch.Tensor, top_k: Optional[int] = None):
        """The same search for text rows that are already encoded (the pipeline hands over the original
        texts' rows of the detection step instead of encoding them a second time)."""
        idx, sim = self._search(text_features.contiguous(), top_k or self.config.top_k)
        paths = self.image_paths
        return [([paths[i] for i in r[r >= 0]], s[r >= 0].astype(float).tolist()) for r, s in zip(idx, sim)]

    def retrieve(self, text: str, k: int = 5):
        """Name used by the efficiency harness (experiments/run_experiments.py:3143)."""
        return self.retrieve_images_by_text(text, top_k=k)

    def compute_similarity_matrix(self, text_features: Optional[np.ndarray] = None,
                                  image_features: Optional[np.ndarray] = None) -> np.ndarray:
        """src/retrieval.py:682-722: [N_text, N_image] cosine (HIP split-bf16 GEMM);
        dot_product / euclidean are derived from the same kernel on the host side
        of the boundary only through their norms."""
        tf = self.text_features if text_features is None else text_features
        imf = self.image_features if image_features is None else image_features
        if tf is None or imf is None:
            raise ValueError("missing text or image features")
        from .metrics import SimilarityCalculator
        cos = SimilarityCalculator.batch_cosine_similarity(np.asarray(tf), np.asarray(imf), engine=self.clip_model.engine)
        if self.config.similarity_metric == "cosine":
            return cos
        tn = np.linalg.norm(tf, axis=1, keepdims=True)
        inn = np.linalg.norm(imf, axis=1, keepdims=True)
        dot = cos * tn * inn.T
        if self.config.similarity_metric == "dot_product":
            return dot
        if self.config.similarity_metric == "euclidean":
            d2 = np.maximum(tn ** 2 + (inn ** 2).T - 2 * dot, 0.0)
            return 1.0 / (1.0 + np.sqrt(d2))
        raise ValueError(f"unsupported similarity metric: {self.config.similarity_metric}")


def extract_features(clip_model: CLIPModel, dataloader, encode_batch: int = 512):
    """scripts/build_faiss_indices.py:59-120 (``IndexBuilder.extract_features``): batches of
    ``{'image': Tensor [b,3,S,S], 'text': list[str], 'image_id': list}`` -> (image_features [R, D],
    text_features [R, D], image_ids), both L2-normalised fp32 numpy (the arrays ``build_dataset_indices``
    saves as ``image_features.npy`` / ``text_features.npy``, :186-192).  Images and texts are regrouped into
    ``encode_batch`` rows per tower launch."""
    imgs, toks, ids = [], [], []
    fi, ft = [], []

    def flush():
        if imgs:
            x = torch.cat(imgs)
            fi.append(clip_model.engine.encode_image(x, True))
            ft.append(clip_model.encode_tokens(torch.cat(toks), True))
            imgs.clear(); toks.clear()

    n = 0
    for batch in dataloader:
        x, _ = clip_model._images_to_device(batch["image"])
        imgs.append(x); toks.append(clip_model.tokenize(list(batch["text"]))); ids.extend(list(batch["image_id"]))
        n += x.shape[0]
        if n >= encode_batch:
            flush(); n = 0
    flush()
    D = clip_model.arch.embed_dim
    f_i = torch.cat(fi).cpu().numpy() if fi else np.zeros((0, D), np.float32)
    f_t = torch.cat(ft).cpu().numpy() if ft else np.zeros((0, D), np.float32)
    return f_i, f_t, ids


def create_retriever(config: Optional[RetrievalConfig] = None, **kw) -> MultiModalRetriever:
    """src/retrieval.py:915."""
    return MultiModalRetriever(config, **kw)


# ---------------------------------------------------------------------------
@dataclass
class RetrievalRefConfig:
    """experiments/defenses/retrieval_ref.py:20-32."""
    reference_count: int = 5
    similarity_threshold: float = 0.3
    use_faiss: bool = True           # accepted; the HIP search is exact either way
    device: str = "cuda"
    cache_size: int = 1000
    enable_reranking: bool = True
    rerank_top_k: int = 20


class RetrievalReferenceGenerator:
    """experiments/defenses/retrieval_ref.py:34-236: ``features.npy`` + ``metadata.json``
    database, ``retrieve_references(text)`` -> list of dicts."""

    def __init__(self, clip_model: CLIPModel, reference_db_path: Optional[str] = None,
                 config: Optional[RetrievalRefConfig] = None, features: Optional[Union[np.ndarray, torch.Tensor]] = None,
                 metadata: Optional[list] = None):
        self.clip_model = clip_model
        self.config = config or RetrievalRefConfig()
        self.reference_features = None
        self.reference_metadata: list = []
        self.feature_cache: Dict[int, List[Dict[str, Any]]] = {}
        self.bank_name = f"retrieval_ref:{id(self):x}"  # own bank slot on the (shared) engine
        self.retrieval_stats = {"total_queries": 0, "successful_retrievals": 0, "cache_hits": 0}
        self.reference_db_path = Path(reference_db_path) if reference_db_path is not None else None
        if features is not None:
            self._register(torch.as_tensor(features), metadata or [])
        elif reference_db_path is not None:
            d = Path(reference_db_path)
            if (d / "features.npy").exists() and (d / "metadata.json").exists():
                feats = np.load(d / "features.npy")                        # allow_pickle=False (default)
                with open(d / "metadata.json", "r", encoding="utf-8") as f:
                    meta = json.load(f)
                self._register(torch.from_numpy(feats), meta)

    def __del__(self):
        try:
            self.clip_model.engine.release_bank(self.bank_name)
        except Exception:
            pass

    def _register(self, feats: torch.Tensor, meta: list) -> None:
        self.reference_features = feats
        self.reference_metadata = meta
        self.clip_model.engine.set_bank(feats.float().to(self.clip_model.device) if feats.dtype != torch.bfloat16
                                        else feats.to(self.clip_model.device), name=self.bank_name)

    # -- bank construction -------------------------------------------------------
    def _save_reference_database(self, path: Optional[str] = None) -> Path:
        """experiments/defenses/retrieval_ref.py:442-457: ``features.npy`` (dense fp32 [R, D], rows
        L2-normalised) + ``metadata.json`` -- the only on-disk bank format the hot path reads."""
        d = Path(path or self.reference_db_path)
        d.mkdir(parents=True, exist_ok=True)
        feats = self.reference_features
        feats = feats.float().cpu().numpy() if isinstance(feats, torch.Tensor) else np.asarray(feats, np.float32)
        np.save(d / "features.npy", feats.astype(np.float32, copy=False))
        with open(d / "metadata.json", "w", encoding="utf-8") as f:
            json.dump(self.reference_metadata, f, ensure_ascii=False, indent=2)
        return d

    def build_reference_database(self, dataset_loader, max_samples: Optional[int] = None, save_interval: int = 1000,
                                 encode_batch: int = 512, save: bool = True) -> bool:
        """experiments/defenses/retrieval_ref.py:459-540 and scripts/build_faiss_indices.py:59-120: stream a
        dataset through the image tower, L2-normalise, stack, save, register as this generator's bank.
        ``dataset_loader`` yields ``{'images': Tensor [b,3,S,S] | list of Tensor/PIL, 'texts': list[str]}``.
        The reference encodes ONE image per call (:482-487); here the loader's batches are regrouped into
        ``encode_batch`` images per ``tvc_encode_image`` launch (the tower's efficient batch), metadata
        entries are the reference's ``{'text', 'index', 'batch_idx'}``."""
        clip = self.clip_model
        feats: List[torch.Tensor] = []
        meta: List[Dict[str, Any]] = []
        pend: List[torch.Tensor] = []
        n_pend = 0
        count = 0

        def flush():
            nonlocal pend, n_pend
            if n_pend:
                x = torch.cat(pend) if len(pend) > 1 else pend[0]
                feats.append(clip.engine.encode_image(x, True))              # L2-normalised rows (:489)
                pend, n_pend = [], 0

        for batch_idx, batch in enumerate(dataset_loader):
            images, texts = batch["images"], batch["texts"]
            if max_samples is not None and count >= max_samples:
                break
            take = len(texts) if max_samples is None else min(len(texts), max_samples - count)
            x, _ = clip._images_to_device(images[:take] if isinstance(images, torch.Tensor) else list(images)[:take])
            for j in range(take):
                meta.append({"text": texts[j], "index": count + j, "batch_idx": batch_idx})
            count += take
            while x.shape[0]:
                room = encode_batch - n_pend
                pend.append(x[:room]); n_pend += min(room, x.shape[0])
                x = x[room:]
                if n_pend >= encode_batch:
                    flush()
        flush()
        if not feats:
            return False                                                      # :534-536
        self._register(torch.cat(feats) if len(feats) > 1 else feats[0], meta)
        if save and getattr(self, "reference_db_path", None) is not None:
            self._save_reference_database()
        return True

    def retrieve_references_batch(self, texts: Sequence[str]) -> List[List[Dict[str, Any]]]:
        eng = self.clip_model.engine
        rows = eng.bank_size(self.bank_name)
        if self.reference_features is None or rows == 0:
            return [[] for _ in texts]                                     # retrieval_ref.py:195-197
        c = self.config
        q = self.clip_model.encode_tokens(self.clip_model.tokenize(list(texts)), True)    # :238-244
        search_k = min(c.rerank_top_k if c.enable_reranking else c.reference_count, rows)   # :249
        idx, sim, _ = eng.bank_search_robust(q, search_k, c.similarity_threshold, want_moments=False, bank=self.bank_name)
        keep = min(c.reference_count, search_k)
        feats = eng.bank_gather(idx[:, :keep].contiguous(), bank=self.bank_name).cpu().numpy()
        idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
        out = []
        for r in range(len(texts)):
            refs = []
            for j in range(keep):                       # sorted desc already (= :292-299), threshold :210-213, cut :216
                i, s = int(idx[r, j]), float(sim[r, j])
                if i >= 0 and s >= c.similarity_threshold:
                    refs.append({"index": i, "similarity": s,
                                 "metadata": self.reference_metadata[i] if i < len(self.reference_metadata) else {},
                                 "features": feats[r, j]})
            out.append(refs)
        return out

    def retrieve_references(self, text: str) -> List[Dict[str, Any]]:
        key = hash(text)
        if key in self.feature_cache:
            self.retrieval_stats["cache_hits"] += 1
            return self.feature_cache[key]
        refs = self.retrieve_references_batch([text])[0]
        if len(self.feature_cache) < self.config.cache_size:
            self.feature_cache[key] = refs
        self.retrieval_stats["total_queries"] += 1
        if refs:
            self.retrieval_stats["successful_retrievals"] += 1
        return refs
