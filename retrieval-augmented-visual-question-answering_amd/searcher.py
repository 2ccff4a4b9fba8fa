"""`Searcher` -- drop-in for the reference's `colbert.Searcher` (TPC/searcher.py:22-132) on the search side.

What is kept: constructor signature and index resolution ({root}/{experiment}/indexes/{index} under the
active `Run().context`, searcher.py:26-31), `.config`, `.ranker` (IndexScorer), `.collection`, `configure`,
`search`, `search_all`, `_search_all_Q`, `dense_search`, the k -> (ncells, centroid_score_threshold, ndocs)
policy INCLUDING its stickiness (the first call's k fixes the values on self.config, searcher.py:92-118),
`remove_zero_tensors`, and the Ranking layout {qid: [(pid, rank, score)]}.

What differs: `_search_all_Q` runs all queries as device batches (the reference loops query by query,
searcher.py:75-79) unless a `filter_fn` is supplied, in which case it stages per query so the callable sees the
same ascending int32 pid tensor.  Query ENCODING (Checkpoint / BERT forward) is outside this build's scope: pass
`query_encoder=callable(list[str]) -> Tensor[n, Nq, dim]` if `search()` / `search_all()` on text are needed.
`config.total_visible_gpus == 0` does NOT mean "stay on the host": the HIP path always runs, with the CPU
path's numerics.
"""
import os

import torch

from .config import ColBERTConfig, Run
from .data import Collection, Provenance, Queries, Ranking
from .scorer import IndexScorer


class Searcher:
    def __init__(self, index, checkpoint=None, collection=None, config=None, disable_gpu=True, query_encoder=None,
                 max_batch=256):
        initial_config = ColBERTConfig.from_existing(config, Run().config)
        if config is not None:
            initial_config.total_visible_gpus = config.total_visible_gpus
        self.index = os.path.join(initial_config.index_root_, index)
        self.index_config = ColBERTConfig.load_from_index(self.index)
        self.checkpoint = checkpoint or self.index_config.checkpoint
        self.checkpoint_config = ColBERTConfig.load_from_checkpoint(self.checkpoint) if isinstance(self.checkpoint, str) else None
        self.config = ColBERTConfig.from_existing(self.checkpoint_config, self.index_config, initial_config)
        self.collection = Collection.cast(collection if collection is not None else
                                          (self.config.collection if isinstance(self.config.collection, (str, list)) else None))
        self.configure(checkpoint=self.checkpoint)
        self.query_encoder = query_encoder
        self.ranker = IndexScorer(self.index, use_gpu=True, max_batch=max_batch)

    def configure(self, **kw_args):
        self.config.configure(**kw_args)

    # ---- text entry points (need an encoder) ------------------------------------------------------------------
    def encode(self, text):
        if self.query_encoder is None:
            raise NotImplementedError("query encoding (Checkpoint.queryFromText) is outside the retrieval hot path; "
                                      "construct Searcher(..., query_encoder=fn) or call _search_all_Q with embeddings")
        queries = text if isinstance(text, list) else [text]
        return self.query_encoder(queries)

    def search(self, text, k=10, filter_fn=None):
        return self.dense_search(self.encode(text), k, filter_fn=filter_fn)

    def search_all(self, queries, k=10, filter_fn=None):
        queries = Queries.cast(queries)
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
            Qb = Q
            if remove_zero_tensors:
                Qb, q_lens = self._compact_nonzero_rows(Q)
            kk = min(k, max(c.ndocs // 4, 1))
            pids, scores, counts = self.ranker.search_batch(Qb, kk, c.ncells, c.centroid_score_threshold, c.ndocs,
                                                            c.query_maxlen, q_lens=q_lens)
            pids, scores, counts = pids.cpu(), scores.cpu(), counts.cpu().tolist()
            all_scored = []
            for i, n in enumerate(counts):
                all_scored.append(list(zip(pids[i, :n].tolist(), range(1, k + 1), scores[i, :n].tolist())))
        data = dict(zip(qids, all_scored))
        provenance = Provenance()
        provenance.source = "Searcher::search_all"
        provenance.queries = queries.provenance() if hasattr(queries, "provenance") else None
        provenance.config = self.config.export()
        provenance.k = k
        return Ranking(data=data, provenance=provenance)

    def dense_search(self, Q, k=10, filter_fn=None, remove_zero_tensors=False):
        self._apply_k_policy(k)
        if remove_zero_tensors:
            nonzero = torch.abs(Q).sum(dim=-1) > 0
            Q = Q[nonzero].unsqueeze(0)
        pids, scores = self.ranker.rank(self.config, Q, filter_fn=filter_fn)
        return pids[:k], list(range(1, k + 1)), scores[:k]
