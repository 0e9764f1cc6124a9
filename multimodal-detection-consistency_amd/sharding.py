"""Multi-GPU execution of the hot path: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on ROCm).  Replaces
``src/utils/multi_gpu_processor.py`` (thread-per-GPU queues ``:49-491``, NCCL
helpers ``:494-620``) and the single-GPU FAISS index (``src/retrieval.py:111-112``).

Two independent axes:

1. **Query batch (pure data parallel).**  Queries are independent: every rank
   holds the encoder weights and takes ``B / W`` queries; no collective on the
   data path.  This is what ``bench.py --gpus N`` runs (weak scaling).
2. **Bank rows (BASELINE configs[3], 10M rows).**  Rank r owns the contiguous
   rows ``shard_bounds(R, W, r)``.  Per batch:
     a. all-gather of the ``[m, D]`` query-side embeddings of every rank
        (5120 x 768 fp32 = 15.7 MB in total at config 4);
     b. every rank searches ALL ``W*m`` rows against its shard (exact local
        top-k with global indices) and gathers the ``kf`` best rows' features;
     c. all-to-all: rank r receives, for ITS OWN m rows, the W partial lists
        (``m x k x 8 B`` + ``m x kf x D x 4 B`` per peer: 0.2 MB + 78.6 MB per pair at
        m = 5120, kf = 5, D = 768) -- xGMI is point-to-point, every link carries a
        distinct peer's slice;
     d. ``tvc_topk_merge`` merges the W sorted partials (HIP kernel).
   **Default form ("fused", no host synchronisation at all):** (c) is ONE ``all_to_all_single`` of fixed-size
   slots -- per peer: the m x k index / similarity lists and the rows of the kf best of every row that needs
   references (none for the image rows of a detection batch, ``feat_from``), in bf16 when the bank is bf16
   (exact): 35 MB per link and step at m = 5120, kf = 5, D = 768, against 7 links x ~50 GB/s.  The local
   search is the asynchronous one; ``check_status`` (one read-back per STEP, next to the records' own copy)
   reports a candidate-list overflow on any rank to every rank.  Every tensor size is known on the host in
   advance, so a step enqueues its collectives behind the kernels and returns: the exchange of batch i runs
   under the towers of batch i + 1 (``bench.py --shard-bank`` pipelines exactly that).
   With ``rows_per_shard`` given AND ``mode="two_phase"``, (c) carries indices + similarities only and two
   variable-size all-to-alls (e, f) fetch just the DISTINCT winners' rows from their owners (fewest bytes,
   but three host read-backs per step: for links far slower than xGMI).
   The ``[M, R]`` similarity rows are never exchanged.

The collectives move small tensors; all arithmetic stays in the HIP kernels.
``ShardOps`` is the seam that lets the world_size-2 ``gloo`` CPU test drive the
same orchestration with a numpy stand-in for the kernels.
"""
from __future__ import annotations

from typing import Optional, Protocol, Tuple

import torch
import torch.distributed as dist


def shard_bounds(R: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of ``rank`` (ceil split, last shards may be short)."""
    per = (R + world - 1) // world
    lo = min(R, rank * per)
    return lo, min(R, lo + per)


def split_queries(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Query range [lo, hi) of ``rank`` for the data-parallel axis."""
    return shard_bounds(n, world, rank)


class ShardOps(Protocol):
    def search(self, rows: torch.Tensor, k: int): ...            # -> idx [M,k] (global), sim [M,k]
    def gather(self, idx: torch.Tensor): ...                      # global idx [M,kf] -> feat [M,kf,D]
    def merge(self, idx_parts, sim_parts, feat_parts): ...        # [W,M,*] -> idx, sim, feat (feat_parts may be None)


class HipShardOps:
    """The product ops: C-ABI kernels on this rank's bank shard."""

    def __init__(self, engine, row_offset: int, count_thr: float = 0.3):
        self.engine = engine
        self.row_offset = row_offset
        self.count_thr = count_thr

    def search(self, rows: torch.Tensor, k: int):
        # the status-checked form: a candidate-list overflow on THIS shard (clustered / duplicate-heavy
        # rows) would otherwise drop true top-k rows of every rank's queries with nothing to show for it
        # after the merge.  The brute-force fallback is local to the shard, so it is safe collectively.
        idx, sim, _ = self.engine.bank_search_robust(rows, k, self.count_thr, idx_offset=self.row_offset,
                                                     want_moments=False)
        return idx, sim

    def search_async(self, rows: torch.Tensor, k: int):
        """The same search enqueued without a status read-back (``check`` reports an overflow later)."""
        idx, sim, _ = self.engine.bank_search(rows, k, self.count_thr, idx_offset=self.row_offset, want_moments=False)
        return idx, sim

    def check(self) -> bool:
        """Synchronises; True if the last ``search_async`` overflowed its candidate lists on this shard."""
        from . import _lib
        try:
            self.engine.bank_status()
            return False
        except _lib.TVCError as e:
            if e.code != _lib.TVC_E_OVERFLOW:
                raise
            return True

    def gather(self, idx: torch.Tensor):
        return self.engine.bank_gather(idx.contiguous(), idx_offset=self.row_offset)

    @property
    def transport_dtype(self):
        """dtype the winners' rows travel in: a bf16 bank's rows are exact in bf16 (half the bytes on the links)."""
        return torch.bfloat16 if getattr(self.engine, "bank_is_bf16", lambda: False)() else torch.float32

    def merge(self, idx_parts, sim_parts, feat_parts):
        idx, sim, feat, _ = self.engine.topk_merge(idx_parts, sim_parts, feat_parts)
        return idx, sim, feat


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    """inp/out [W, ...]: out[w] = what rank w sent to me.  RCCL: one all_to_all_single;
    gloo (CPU tests) lacks it for some dtypes -> all_gather + slice."""
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(out, inp.contiguous(), group=group)
        return
    W = dist.get_world_size(group)
    me = dist.get_rank(group)
    bufs = [torch.empty_like(inp) for _ in range(W)]
    dist.all_gather(bufs, inp.contiguous(), group=group)
    for w in range(W):
        out[w].copy_(bufs[w][me])


def _exchange_lists(send, group):
    """send[w] = 1-D/2-D tensor for rank w (any lengths) -> recv[w] = what rank w sent to me.
    RCCL: counts by all_to_all_single, then ONE all_to_all_single with split sizes; gloo (CPU
    tests): all_gather_object."""
    W = dist.get_world_size(group)
    me = dist.get_rank(group)
    if dist.get_backend(group) != "nccl":
        box = [None] * W
        dist.all_gather_object(box, [t.cpu() for t in send], group=group)
        return [box[w][me].to(send[0].device) for w in range(W)]
    dev = send[0].device
    cnt_out = torch.tensor([t.shape[0] for t in send], dtype=torch.int64, device=dev)
    cnt_in = torch.empty_like(cnt_out)
    dist.all_to_all_single(cnt_in, cnt_out, group=group)
    n_in = cnt_in.tolist()
    tail = tuple(send[0].shape[1:])
    out = torch.empty((sum(n_in),) + tail, dtype=send[0].dtype, device=dev)
    dist.all_to_all_single(out, torch.cat(send).contiguous(), output_split_sizes=n_in,
                           input_split_sizes=cnt_out.tolist(), group=group)
    return list(out.split(n_in))


class ShardedBankSearch:
    """Exact global top-k over a row-sharded bank for data-parallel query rows.

    ``rows_per_shard`` (= ``ceil(R / W)``, the split of ``shard_bounds``) enables the two-phase
    exchange: indices + similarities first, merge, then ONLY the winners' rows travel (each from the
    shard that owns it): ``m x kf x D x 4 B`` per rank instead of W times that.  Without it the
    single-phase form (every shard ships the rows of its own kf best) is used."""

    def __init__(self, ops: ShardOps, group=None, rows_per_shard: Optional[int] = None, mode: Optional[str] = None):
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rows_per_shard = rows_per_shard
        # "fused": one fixed-size all-to-all, no host synchronisation (default); "two_phase": fewest bytes (needs
        # rows_per_shard); "single_phase": the round-1 form (fp32 rows of every shard's kf best)
        self.mode = mode or "fused"
        if self.mode not in ("fused", "two_phase", "single_phase"):
            raise ValueError(f"unknown exchange mode {self.mode!r}")
        if self.mode == "two_phase" and rows_per_shard is None:
            raise ValueError("the two-phase exchange needs rows_per_shard")
        self.last_exchange = {}
        self._unchecked = False     # a fused search whose overflow status has not been read yet (check_status clears it)

    # ---- the sync-free form ------------------------------------------------------------------------------
    def search_fused(self, rows: torch.Tensor, k: int, kf: int, feat_from: int = 0):
        """(a) all-gather of the query rows, (b) asynchronous local search + gather of the kf best rows of every row
        that needs references, (c) ONE all-to-all of fixed-size slots [idx | sim | rows], (d) merge.  Nothing here
        reads a value back to the host: call ``check_status`` once per step."""
        if self._unchecked:
            # the fused search never reads its overflow flag: a caller that skips check_status would get silently
            # truncated candidate lists from a degenerate shard -- refuse to run a second step on top of an unread one
            raise RuntimeError("ShardedBankSearch(mode='fused'): call check_status() once per step (after search) -- the "
                               "previous search's overflow status was never read; mode='two_phase' / 'single_phase' check per call")
        W, (m, D), dev = self.world, rows.shape, rows.device
        need = m - feat_from
        allrows = torch.empty((W * m, D), dtype=rows.dtype, device=dev)
        dist.all_gather_into_tensor(allrows, rows.contiguous(), group=self.group)          # (a)
        search = getattr(self.ops, "search_async", self.ops.search)
        idx, sim = search(allrows, k)                                                       # (b) [W*m, k], global indices
        tdt = getattr(self.ops, "transport_dtype", torch.float32)
        esz = 2 if tdt == torch.bfloat16 else 4
        n_is = m * k * 4
        n_ft = (need * kf * D * esz + 15) // 16 * 16
        slot = 2 * ((n_is + 15) // 16 * 16) + n_ft
        o_sim, o_ft = (n_is + 15) // 16 * 16, 2 * ((n_is + 15) // 16 * 16)
        send = torch.empty((W, slot), dtype=torch.uint8, device=dev)
        send[:, :n_is].view(torch.int32).copy_(idx.view(W, m * k))
        send[:, o_sim:o_sim + n_is].view(torch.float32).copy_(sim.view(W, m * k))
        if need > 0:
            widx = idx.view(W, m, k)[:, feat_from:, :kf].reshape(W * need, kf)
            feat = self.ops.gather(widx.contiguous())                                       # [W*need, kf, D] fp32
            send[:, o_ft:o_ft + need * kf * D * esz].view(tdt).copy_(feat.view(W, need * kf * D))
        recv = torch.empty_like(send)
        _all_to_all(recv, send, self.group)                                                 # (c)
        idx_in = recv[:, :n_is].view(torch.int32).reshape(W, m, k)
        sim_in = recv[:, o_sim:o_sim + n_is].view(torch.float32).reshape(W, m, k)
        out_feat = torch.zeros((m, kf, D), dtype=torch.float32, device=dev)
        if feat_from > 0 or need == 0:
            hi, hs, _ = self.ops.merge(idx_in[:, :m - need].contiguous(), sim_in[:, :m - need].contiguous(), None)
        if need > 0:
            feat_in = recv[:, o_ft:o_ft + need * kf * D * esz].view(tdt).reshape(W, need, kf, D).to(torch.float32)
            ti, ts, tf = self.ops.merge(idx_in[:, feat_from:].contiguous(), sim_in[:, feat_from:].contiguous(), feat_in)   # (d)
            out_feat[feat_from:] = tf
            midx, msim = (torch.cat([hi, ti]), torch.cat([hs, ts])) if feat_from > 0 else (ti, ts)
        else:
            midx, msim = hi, hs
        self.last_exchange = {"mode": "fused", "bytes_per_peer": int(slot), "host_syncs": 0,
                              "rows_sent": int(W * need * kf), "bytes_per_row": D * esz}
        self._unchecked = True
        return midx, msim, out_feat

    def check_status(self) -> None:
        """One read-back per step: raise ``TVCError(TVC_E_OVERFLOW)`` on EVERY rank if the asynchronous search of any
        rank dropped candidates (degenerate shard); the caller then repeats the step with ``mode="two_phase"`` /
        the status-checked search, whose brute-force fallback is local to the shard."""
        from . import _lib
        # every rank must reach the all_reduce whatever its local check did: a rank that raised before it would leave the
        # others hanging in the collective.  Two flags travel: [overflow, any other error].
        over, other = False, None
        try:
            over = bool(getattr(self.ops, "check", lambda: False)())
        except Exception as e:                                   # noqa: BLE001 -- re-raised below, after the collective
            other = e
        finally:
            self._unchecked = False
        if self.world > 1:
            t = torch.tensor([1.0 if over else 0.0, 1.0 if other is not None else 0.0])
            t = t.cuda() if dist.get_backend(self.group) == "nccl" else t
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            over = bool(t[0].item() > 0)
            if other is None and t[1].item() > 0:
                other = RuntimeError("sharded bank search: another rank's status check failed")
        if other is not None:
            raise other
        if over:
            raise _lib.TVCError(_lib.TVC_E_OVERFLOW, "sharded bank search: a shard's candidate lists overflowed")

    def search(self, rows: torch.Tensor, k: int, kf: int, feat_from: int = 0):
        """rows [m, D] (this rank's query-side rows; m equal on every rank) ->
        idx [m, k] global, sim [m, k], feat [m, kf, D] of the kf best.  ``feat_from``: the first ``feat_from``
        rows need no feature rows (the image rows of a detection batch: only the text rows' references are
        compared with the image) -- their feat stays zero and nothing travels for them."""
        if self.mode == "fused":
            return self.search_fused(rows, k, kf, feat_from)
        W, (m, D) = self.world, rows.shape
        allrows = torch.empty((W * m, D), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(allrows, rows.contiguous(), group=self.group)         # (a)
        idx, sim = self.ops.search(allrows, k)                                            # (b)
        idx_in = torch.empty((W, m, k), dtype=idx.dtype, device=rows.device)
        sim_in = torch.empty((W, m, k), dtype=sim.dtype, device=rows.device)
        _all_to_all(idx_in, idx.view(W, m, k), self.group)                                 # (c)
        _all_to_all(sim_in, sim.view(W, m, k), self.group)
        if self.mode == "single_phase":
            feat = self.ops.gather(idx[:, :kf])
            feat_in = torch.empty((W, m, kf, D), dtype=feat.dtype, device=rows.device)
            _all_to_all(feat_in, feat.view(W, m, kf, D), self.group)
            return self.ops.merge(idx_in, sim_in, feat_in)                                # (d)
        # ---- two-phase: merge the lists, then fetch the winners' rows from their owners
        midx, msim, _ = self.ops.merge(idx_in, sim_in, None)                              # (d)
        win = midx[feat_from:, :kf].reshape(-1)                                           # global, -1 = none
        # every DISTINCT winner travels once: the N+1 text rows of a query mostly retrieve the same references
        uniq, inv = torch.unique(win, return_inverse=True)
        owner = torch.where(uniq >= 0, torch.clamp(uniq // self.rows_per_shard, max=W - 1), torch.full_like(uniq, -1))
        pos = [torch.nonzero(owner == w).flatten() for w in range(W)]
        asked = _exchange_lists([uniq[p_] for p_ in pos], self.group)                     # (e) who wants which of my rows
        tdt = getattr(self.ops, "transport_dtype", torch.float32)
        sent = [self.ops.gather(a.view(-1, 1)).view(-1, D).to(tdt) if a.numel() else
                torch.empty((0, D), dtype=tdt, device=rows.device) for a in asked]
        got = _exchange_lists(sent, self.group)                                           # (f) the rows come back
        ufeat = torch.zeros((uniq.numel(), D), dtype=torch.float32, device=rows.device)   # (a -1 entry stays zero)
        for w in range(W):
            if pos[w].numel():
                ufeat[pos[w]] = got[w].to(torch.float32)
        feat = torch.zeros((m, kf, D), dtype=torch.float32, device=rows.device)
        feat[feat_from:] = ufeat[inv].view(m - feat_from, kf, D)
        self.last_exchange = {"mode": "two_phase", "host_syncs": 3,
                              "rows_sent": int(sum(t.shape[0] for t in sent)), "distinct_winners": int((uniq >= 0).sum()),
                              "winner_slots": int(win.numel()), "bytes_per_row": D * (2 if tdt == torch.bfloat16 else 4)}
        return midx, msim, feat


def detect_sharded(engine, search: ShardedBankSearch, img: torch.Tensor, txt: torch.Tensor, cfg) -> torch.Tensor:
    """Bank-sharded variant of ``TVCEngine.detect_embeddings`` for this rank's B
    queries: img [B, D], txt [B, N+1, D] -> records [B, rec_stride]."""
    B, N1, D = txt.shape
    k = max(cfg.search_k, cfg.reference_count)
    idx, sim, feat = search.search(txt.reshape(B * N1, D), k, cfg.reference_count)
    return engine.consistency(img, txt, cfg, idx, sim, feat)
