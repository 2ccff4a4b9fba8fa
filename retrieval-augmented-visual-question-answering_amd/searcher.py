"""`Searcher` -- drop-in for the reference's `colbert.Searcher` (TPC/searcher.py:22-132) on the search side.

What is kept: constructor signature and index resolution ({root}/{experiment}/indexes/{index} under the
active `Run().context`, searcher.py:26-31), `.config`, `.ranker` (IndexScorer), `.collection`, `configure`,
`search`, `search_all`, `_search_all_Q`, `dense_search`, the k -> (ncells, centroid_score_threshold, ndocs)
policy INCLUDING its stickiness (the first call's k fixes the values on self.config, searcher.py:92-118),
`remove_zero_tensors`, and the Ranking layout {qid: [(pid, rank, score)]}.

What differs: `_search_all_Q` runs all queries as device batches (the reference loops query by query,
searcher.py:75-79) unless a `filter_fn` is supplied, in which case it stages per query so the callable sees the
same ascending int32 pid tensor.  Query ENCODING (Checkpoint / BERT forward) is outside this build's scope: the
reference's `Checkpoint` is built lazily on the first `encode()` when its class is available (`ravqa_amd.install()`
binds it), or pass `query_encoder=callable(list[str]) -> Tensor[n, Nq, dim]`.

Boundary value types are class attributes (`ColBERTConfig`, `Run`, `Collection`, `Queries`, `Ranking`, `Provenance`,
`IndexScorer`, `Checkpoint`): this module binds this package's own host-side mirrors; `ravqa_amd.dropin.install()`
derives a subclass bound to the reference package's classes, so that inside the RA-VQA executors the objects that
go in (`ColBERTConfig`, `Queries`, the `Run()` context) and come out (`Ranking`, `.config`) ARE the reference's.

Numerics: the DEFAULT is the reference's CPU-path arithmetic (fp32, zero-clamped MaxSim) whatever
`config.total_visible_gpus` says -- it is the arithmetic the golden vectors pin, and `total_visible_gpus == 0` never
meant "stay on the host" here: the HIP path always runs.  In the reference `total_visible_gpus > 0` selects its CUDA
branch (fp16 centroids / embeddings, -9999 padding, `index_storage.py:113-149`).  This build has that arithmetic as a mode
("gpu-fp16"), checked against the reference's expressions on CPU half tensors but NOT against its CUDA kernels, so it is
opt-in: `Searcher(..., numerics="reference")` or `FLMR_NUMERICS=reference` follows the reference's selection (gpu-fp16 when
the caller assigned total_visible_gpus > 0, as FLMR_executor.py:784 does on one GPU), `numerics="gpu-fp16"` forces it,
`"cpu"` (default) keeps the pinned arithmetic.  An index the fp16 mode cannot serve (centroids not fp16-representable, K not
a multiple of 64) falls back to "cpu" with a warning.  The chosen mode is logged (logger "ravqa_amd") and kept in `.numerics`.
"""
import logging
import os
import warnings

import torch

from . import config as _config
from . import data as _data
from .scorer import IndexScorer as _IndexScorer


class Searcher:
    # boundary types (see module docstring)
    ColBERTConfig = _config.ColBERTConfig
    Run = _config.Run
    Collection = _data.Collection
    Queries = _data.Queries
    Ranking = _data.Ranking
    Provenance = _data.Provenance
    IndexScorer = _IndexScorer
    Checkpoint = None   # the reference's colbert.modeling.checkpoint.Checkpoint when installed over the reference package
    _warned_cpu_numerics = False

    def __init__(self, index, checkpoint=None, collection=None, config=None, disable_gpu=True, query_encoder=None,
                 max_batch=256, numerics=None, pipelined=None):
        cfg_cls = self.ColBERTConfig
        initial_config = cfg_cls.from_existing(config, self.Run().config)
        if config is not None:  # searcher.py:27 (the reference dereferences `config` unconditionally)
            initial_config.total_visible_gpus = config.total_visible_gpus
        self.index = os.path.join(initial_config.index_root_, index)
        self.index_config = cfg_cls.load_from_index(self.index)
        self.checkpoint = checkpoint or self.index_config.checkpoint
        self.checkpoint_config = cfg_cls.load_from_checkpoint(self.checkpoint) if isinstance(self.checkpoint, str) else None
        self.config = cfg_cls.from_existing(self.checkpoint_config, self.index_config, initial_config)
        self.collection = self._cast_collection(collection if collection is not None else self.config.collection)
        self.configure(checkpoint=self.checkpoint, collection=self.collection)
        self.query_encoder = query_encoder
        self._checkpoint_model = None
        # _search_all_Q returns while the device still works and hands out lists that wait for their sub-batch (FLMR_PIPELINED=0 /
        # pipelined=False: every result on the host, and every deferred device error raised, before it returns)
        self.pipelined = (os.environ.get("FLMR_PIPELINED", "1") != "0") if pipelined is None else bool(pipelined)
        use_gpu = (self.config.total_visible_gpus or 0) > 0
        # searcher.py:42-45: total_visible_gpus > 0 selects the reference's CUDA branch.  Here that arithmetic is an opt-in
        # mode (module docstring): "reference" follows the reference's selection when the CALLER assigned the value (the
        # executor does on a single GPU, FLMR_executor.py:784; a mere config default does not count), "gpu-fp16" forces it,
        # "cpu" -- the default, what total_visible_gpus = 0 (every multi-GPU run, FLMR_executor.py:779-781) means and what
        # the golden vectors pin -- keeps the CPU-path arithmetic.
        requested = numerics or os.environ.get("FLMR_NUMERICS") or "cpu"
        if requested not in ("cpu", "gpu-fp16", "reference"):
            raise ValueError(f"numerics must be 'cpu', 'gpu-fp16' or 'reference', got {requested!r}")
        explicit = config is not None and "total_visible_gpus" in getattr(config, "assigned", {})
        mode = ("gpu-fp16" if (use_gpu and explicit) else "cpu") if requested == "reference" else requested
        self.ranker = self.IndexScorer(self.index, use_gpu, max_batch=max_batch, numerics=mode)
        self.numerics = getattr(self.ranker, "numerics", mode) or mode    # (the scorer falls back to "cpu" where fp16 is unsupported)
        logging.getLogger("ravqa_amd").info("Searcher(%s): numerics %s (requested %s, total_visible_gpus=%s)", self.index,
                                            self.numerics, requested, self.config.total_visible_gpus)
        if use_gpu and explicit and self.numerics == "cpu" and not Searcher._warned_cpu_numerics:
            # the caller asked for the reference's GPU branch (FLMR_executor.py:784 on one GPU): this build answers in the CPU-path
            # arithmetic unless told otherwise -- near-ties can rank differently from a reference single-GPU run.  Said once.
            Searcher._warned_cpu_numerics = True
            warnings.warn("ravqa_amd.Searcher: config.total_visible_gpus > 0 was assigned, which selects the reference's CUDA-branch "
                          "arithmetic (fp16 scores, -9999 padding); this build runs its default CPU-path arithmetic (fp32, the one its "
                          "parity tests pin) -- pass numerics='reference' (or FLMR_NUMERICS=reference) to follow the reference's selection",
                          RuntimeWarning, stacklevel=2)

    def _cast_collection(self, obj):
        """Collection.cast, but lazy: the search path never reads passage text, so a missing / unset collection is not
        an error here (the reference would fail in Collection.cast(None), searcher.py:39)."""
        if obj is None:
            return None
        if isinstance(obj, str) and not os.path.exists(obj):
            return obj
        try:
            return self.Collection.cast(obj)
        except AssertionError:
            return obj

    def configure(self, **kw_args):
        self.config.configure(**kw_args)

    # ---- text entry points (need an encoder) ------------------------------------------------------------------
    def encode(self, text):
        queries = text if isinstance(text, list) else [text]
        if self.query_encoder is not None:
            return self.query_encoder(queries)
        if self.Checkpoint is None:
            raise NotImplementedError("query encoding (Checkpoint.queryFromText) is outside the retrieval hot path; "
                                      "construct Searcher(..., query_encoder=fn), call ravqa_amd.install() so the "
                                      "reference's Checkpoint is used, or call _search_all_Q with embeddings")
        if self._checkpoint_model is None:  # searcher.py:41-45, deferred to the first text query
            self._checkpoint_model = self.Checkpoint(self.checkpoint, colbert_config=self.config)
            if torch.cuda.is_available():
                self._checkpoint_model = self._checkpoint_model.cuda()
        bsize = 128 if len(queries) > 128 else None
        self._checkpoint_model.query_tokenizer.query_maxlen = self.config.query_maxlen
        return self._checkpoint_model.queryFromText(queries, bsize=bsize, to_cpu=True)

    def search(self, text, k=10, filter_fn=None):
        return self.dense_search(self.encode(text), k, filter_fn=filter_fn)

    def search_all(self, queries, k=10, filter_fn=None):
        queries = self.Queries.cast(queries)
        Q = self.encode(list(queries.values()))
        return self._search_all_Q(queries, Q, k, filter_fn=filter_fn)

    # ---- policy (searcher.py:92-118) ----------------------------------------------------------------------------
    def _apply_k_policy(self, k):
        if k <= 100:
            ncells, thr, ndocs = 2, 0.45, 1024
        else:
            ncells, thr, ndocs = 4, 0.4, max(k * 4, 4096)
        if self.config.ncells is None:
            self.configure(ncells=ncells)
        if self.config.centroid_score_threshold is None:
            self.configure(centroid_score_threshold=thr)
        if self.config.ndocs is None:
            self.configure(ndocs=ndocs)

    @staticmethod
    def _compact_nonzero_rows(Q):
        """remove_zero_tensors (searcher.py:120-126) for a batch: move each query's non-zero rows to the front,
        return (Q_compacted, q_lens)."""
        nz = Q.abs().sum(dim=-1) > 0                                   # [n, Nq]
        order = torch.argsort((~nz).to(torch.int8), dim=1, stable=True)  # non-zero rows first, original order kept
        Qc = torch.gather(Q, 1, order.unsqueeze(-1).expand_as(Q))
        lens = nz.sum(dim=1).to(torch.int32)
        Qc = Qc * (torch.arange(Q.size(1), device=Q.device).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
        return Qc, lens

    @staticmethod
    def ranking_lists(pids, scores, counts, k, lazy=True):
        """Device results [n, k] -> the Ranking layout [[(pid, rank, score)] * count] (searcher.py:81-89, :132: ranks are 1..k).
        One bulk transfer per array; each query's list is a `data.RankedList` over its two numpy rows, which builds the tuples
        when they are read (lazy=False: plain lists now, built by a structured array's tolist() -- 10 ms per 1024 x 100, more than
        the device path's 7 ms, which is why it is no longer what `_search_all_Q` does)."""
        import numpy as np
        P, S, C = pids.cpu().numpy(), scores.cpu().numpy(), counts.cpu().tolist()
        if P.ndim != 2 or P.shape[1] == 0:
            return [[] for _ in C]
        C = [min(max(int(n), 0), P.shape[1]) for n in C]
        if lazy:
            return _data.ranked_lists(P, S, C)
        rec = np.empty(P.shape, dtype=[("pid", np.int32), ("rank", np.int32), ("score", np.float32)])
        rec["pid"], rec["score"] = P, S
        rec["rank"] = np.arange(1, P.shape[1] + 1, dtype=np.int32)
        rows = rec.tolist()
        return [row if n >= len(row) else row[:n] for row, n in zip(rows, C)]

    @staticmethod
    def pending_lists(pend):
        """scorer.PendingBatch -> the Ranking layout, one `data.ChunkRankedList` per query: rows become readable sub-batch by
        sub-batch (FLMR_executor.py:852-858 reads every tuple of every query -- it reads sub-batch 0 while the device still works
        on the others)."""
        rows = []
        for j, (b0, b1) in enumerate(pend.chunks()):
            chunk = _data.RankedChunk((lambda j=j: pend.wait(j)), pend.pids[b0:b1], pend.scores[b0:b1], pend.counts[b0:b1])
            rows += [_data.ChunkRankedList(chunk, i) for i in range(b1 - b0)]
        return rows

    # ---- embedding entry points -----------------------------------------------------------------------------------
    def _search_all_Q(self, queries, Q, k, filter_fn=None, progress=True, remove_zero_tensors=False):
        qids = list(queries.keys())
        if filter_fn is not None:
            all_scored = [list(zip(*self.dense_search(Q[i:i + 1], k, filter_fn=filter_fn,
                                                      remove_zero_tensors=remove_zero_tensors))) for i in range(Q.size(0))]
        else:
            self._apply_k_policy(k)
            c = self.config
            q_lens = None
            all_scored = None
            Qb = Q
            if remove_zero_tensors:
                Qb, q_lens = self._compact_nonzero_rows(Q)
            kk = min(k, max(c.ndocs // 4, 1))
            if self.pipelined and hasattr(self.ranker, "search_batch_pending"):
                # device sub-batch i+1 beside the host's reading of sub-batch i: the call returns when the kernels are queued, every
                # sub-batch's rows land in pinned memory behind its kernels, and a list waits for ITS sub-batch when it is first read
                pend = self.ranker.search_batch_pending(Qb, kk, c.ncells, c.centroid_score_threshold, c.ndocs, c.query_maxlen, q_lens=q_lens)
                all_scored = self.pending_lists(pend)
            elif hasattr(self.ranker, "search_batch_checked"):   # + the deferred device-side errors (candidate bound, q_lens range;
                pids, scores, counts = self.ranker.search_batch_checked(Qb, kk, c.ncells, c.centroid_score_threshold, c.ndocs,   # score-row capacity: redone)
                                                                        c.query_maxlen, q_lens=q_lens)
            else:
                pids, scores, counts = self.ranker.search_batch(Qb, kk, c.ncells, c.centroid_score_threshold, c.ndocs,
                                                                c.query_maxlen, q_lens=q_lens)
                if hasattr(self.ranker, "check"):
                    self.ranker.check()
            if all_scored is None:
                all_scored = self.ranking_lists(pids, scores, counts, k)
        data = dict(zip(qids, all_scored))
        provenance = self.Provenance()
        provenance.source = "Searcher::search_all"
        provenance.queries = queries.provenance() if hasattr(queries, "provenance") else None
        provenance.config = self.config.export()
        provenance.k = k
        return self.Ranking(data=data, provenance=provenance)

    def dense_search(self, Q, k=10, filter_fn=None, remove_zero_tensors=False):
        self._apply_k_policy(k)
        if remove_zero_tensors:
            nonzero = torch.abs(Q).sum(dim=-1) > 0
            Q = Q[nonzero].unsqueeze(0)
        pids, scores = self.ranker.rank(self.config, Q, filter_fn=filter_fn)
        return pids[:k], list(range(1, k + 1)), scores[:k]
