"""The FLMR / ColBERT scoring head -- host-side mirror of `colbert_score`, `colbert_score_packed`,
`colbert_score_reduce` and `ColBERT.segmented_maxsim` (TPC/modeling/colbert.py:235-311) and of the inherited
`score(Q, D_padded, D_mask)` of the FLMR model classes (src/models/retriever/FLMR.py; TPC/modeling/colbert.py:217-224).

Forward only.  When autograd needs the score (training, in-batch negatives) the caller must keep using its torch
expression: these functions raise if an input requires grad instead of silently computing without a graph.
"""
import torch

from . import ops


def _no_grad_only(*tensors):
    if torch.is_grad_enabled() and any(getattr(t, "requires_grad", False) for t in tensors):
        raise RuntimeError("the HIP MaxSim scorer is forward-only; call it under torch.no_grad()/inference_mode()")


def _interaction(config):
    interaction = getattr(config, "interaction", "colbert") if config is not None else "colbert"
    assert interaction in ("colbert", "flipr"), interaction   # colbert.py:244
    return interaction


def reduce_colmax(colmax, config=None):
    """[B, Nq] per-column maxima -> [B] (the second half of colbert_score_reduce, colbert.py:244-263): their sum for
    'colbert'; for 'flipr' the 32 largest of the first 64 columns plus, when there are that many, the 8 largest of the rest."""
    if _interaction(config) == "flipr":
        assert config.query_maxlen == 64, ("for now", config)
        qm, K2 = config.query_maxlen, 8
        out = colmax[:, :qm].topk(qm // 2, dim=-1).values.sum(-1)
        if K2 <= colmax.size(1) - qm:
            out = out + colmax[:, qm:].topk(K2, dim=-1).values.sum(1)
        return out
    return colmax.sum(-1)


def colbert_score_reduce(scores_padded, D_mask, config=None):
    """[B, Ld, Nq] scores + mask -> [B] (colbert.py:235-263, both interactions): tiny torch epilogue kept for callers that
    already hold a padded score tensor on the device."""
    pad = ~D_mask.view(scores_padded.size(0), scores_padded.size(1)).bool()
    scores_padded = scores_padded.masked_fill(pad.unsqueeze(-1), -9999)
    return reduce_colmax(scores_padded.max(1).values, config)


def colbert_score(Q, D_padded, D_mask, config=None, use_gpu=False):
    """Padded late-interaction score (colbert.py:268-286): Q [1|B, Nq, d] x D [B, Ld, d] -> [B] on the device; the 'flipr'
    interaction takes the kernel's per-column maxima and reduces them as colbert.py:246-261 does."""
    _no_grad_only(Q, D_padded)
    assert Q.dim() == 3 and D_padded.dim() == 3 and Q.size(0) in (1, D_padded.size(0))
    if _interaction(config) == "flipr":
        return reduce_colmax(ops.colbert_colmax_padded(Q, D_padded, D_mask), config)
    return ops.colbert_score_padded(Q, D_padded, D_mask)


def colbert_score_packed(Q, D_packed, D_lengths, config=None):
    """Single-query packed score (colbert.py:289-311) from an already materialised D_packed: GEMM by torch on the
    device (plumbing), zero-clamped segmented MaxSim by the HIP op."""
    _no_grad_only(Q, D_packed)
    Qd = Q.squeeze(0).to("cuda", torch.float32)
    scores = D_packed.to("cuda", torch.float32) @ Qd.T
    return ops.segmented_maxsim(scores, torch.as_tensor(D_lengths).to("cuda"))


class ColBERT:
    """Only the class attribute the reference's packed scorer looks up (colbert.py:44-62, :311)."""
    segmented_maxsim = staticmethod(ops.segmented_maxsim)

    @classmethod
    def try_load_torch_extensions(cls, use_gpu):
        from . import _native
        _native.load(require_device=True)
        cls.loaded_extensions = True


class FLMRScoringHead:
    """`score(Q, D_padded, D_mask)` with the semantics the four FLMR model classes inherit from ColBERT.score
    (colbert.py:217-224, similarity == 'cosine' path): exhaustive-search and RAG re-scoring callers
    (src/executors/FLMR_executor.py:833, src/models/rag/rag_model_blip.py:435) can bind this in place of the
    torch expression when gradients are off."""

    def score(self, Q, D_padded, D_mask):
        return colbert_score(Q, D_padded, D_mask)


def exhaustive_search(query_embeddings, item_embeddings, item_embedding_mask, k, item_chunk=None):
    """Brute-force late-interaction search: every query against every item, top-k by score -- the `else` branch of
    `FLMRExecutor.evaluate_outputs` (src/executors/FLMR_executor.py:799-847), which fills `rate_batch[nq, n_items]` four
    items at a time through `model.score` and then sorts each row descending.

    query_embeddings [nq, Nq, d]; item_embeddings [n_items, Ld, d]; item_embedding_mask [n_items, Ld(,1)] (1 = real token).
    Returns (indices int64 [nq, k'], scores f32 [nq, k'], rate_batch f32 [nq, n_items]) on the device, k' = min(k, n_items).
    One launch of the padded MaxSim kernel per item chunk scores EVERY query against the chunk's items (the cross form,
    `flmr_colbert_score_cross`); `item_chunk` bounds the items resident per launch for corpora larger than HBM headroom."""
    _no_grad_only(query_embeddings, item_embeddings)
    Q = torch.as_tensor(query_embeddings).to("cuda", torch.float32)
    D = torch.as_tensor(item_embeddings)
    M = torch.as_tensor(item_embedding_mask)
    n_items = D.size(0)
    M = M.reshape(n_items, -1)
    item_chunk = item_chunk or n_items
    rate = torch.empty((Q.size(0), n_items), dtype=torch.float32, device="cuda")
    for i0 in range(0, n_items, item_chunk):
        Dc = D[i0:i0 + item_chunk].to("cuda", torch.float32).contiguous()
        Mc = M[i0:i0 + item_chunk].to("cuda")
        for q0 in range(0, Q.size(0), 65535):   # (grid dimension limit of the cross form)
            rate[q0:q0 + 65535, i0:i0 + Dc.size(0)] = ops.colbert_score_cross(Q[q0:q0 + 65535], Dc, Mc)
    scores, indices = torch.sort(rate, dim=-1, descending=True)
    kk = min(int(k), n_items)
    return indices[:, :kk], scores[:, :kk], rate


def exhaustive_ranking_dict(indices, scores):
    """(indices, scores) -> {query_index: [(item index, rank from 0, int(score))]} exactly as FLMR_executor.py:838-845
    builds it (the reference truncates the score to an int there; kept so downstream records match)."""
    idx, sc = indices.cpu(), scores.cpu()
    return {q: [(int(idx[q, i]), i, int(sc[q, i])) for i in range(idx.size(1))] for q in range(idx.size(0))}
