#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz by RUNNING THE REFERENCE's CPU path.

Run only in the build container (needs /root/reference; the GPU box has no reference):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What it does
  * imports third_party/ColBERT from /root/reference with three scratch shims (ujson -> json,
    a stub `git`, transformers.AdamW alias) -- see SURVEY.md Appendix C;
  * builds small synthetic indexes with the reference's OWN codec / IVF code
    (ResidualCodec.compress, ResidualCodec.save, optimize_ivf);
  * loads them with the reference IndexScorer(use_gpu=False) and taps every stage of
    IndexScorer.rank (retrieve -> filter_pids -> decompress_residuals -> normalize ->
    colbert_score_packed -> sort) plus the four C++ ops in isolation;
  * stores inputs + expected outputs as plain numpy arrays.

Only DATA is written to the repo: no reference source text is copied.
"""
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FLMR_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, os.path.join(REF, "third_party", "ColBERT"))
sys.path.insert(0, os.path.join(HERE, "_shims"))
os.environ.setdefault("TORCH_EXTENSIONS_DIR", "/tmp/flmr_ref_torch_ext")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import transformers  # noqa: E402

if not hasattr(transformers, "AdamW"):
    transformers.AdamW = torch.optim.AdamW

from colbert.infra.config import ColBERTConfig  # noqa: E402
from colbert.indexing.codecs.residual import ResidualCodec  # noqa: E402
from colbert.indexing.utils import optimize_ivf  # noqa: E402
from colbert.modeling.colbert import ColBERT, colbert_score, colbert_score_packed  # noqa: E402
from colbert.search.index_storage import IndexScorer  # noqa: E402
from colbert.search.strided_tensor import StridedTensor  # noqa: E402

DIM = 128


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def make_corpus(seed, n_docs, K, max_len, sigma=0.05):
    """Clustered token embeddings: token = normalize(proto[c] + sigma*N(0,I))."""
    rng = np.random.default_rng(seed)
    protos = unit(rng.standard_normal((K, DIM)).astype(np.float32))
    doclens = rng.integers(1, max_len + 1, size=n_docs).astype(np.int64)
    # edge cases: a few 1-token docs, a few empty docs, one maximal doc
    doclens[3] = 1
    doclens[7] = 0
    doclens[n_docs // 2] = 0
    doclens[n_docs - 1] = max_len
    N = int(doclens.sum())
    # documents draw most tokens from a small per-doc topic set so that docs share centroids
    tok_c = np.empty(N, dtype=np.int64)
    off = 0
    for d in range(n_docs):
        L = int(doclens[d])
        topics = rng.integers(0, K, size=max(2, L // 3 + 1))
        tok_c[off:off + L] = topics[rng.integers(0, len(topics), size=L)]
        off += L
    embs = unit(protos[tok_c] + sigma * rng.standard_normal((N, DIM)).astype(np.float32)).astype(np.float32)
    return protos, doclens, tok_c, embs


def build_reference_index(index_dir, protos, doclens, embs, nbits, seed):
    """Write an index directory in the reference's on-disk format using the reference's own code."""
    os.makedirs(index_dir, exist_ok=True)
    cfg = ColBERTConfig(dim=DIM, nbits=nbits, total_visible_gpus=0)
    centroids = torch.from_numpy(protos).half().float()  # save() stores half: pre-round
    embs_t = torch.from_numpy(embs)

    # bucket tables from held-out residual quantiles, as collection_indexer.py:286-308 does
    rng = np.random.default_rng(seed + 99)
    held = embs_t[torch.from_numpy(rng.choice(embs.shape[0], size=min(4096, embs.shape[0]), replace=False))]
    tmp_codec = ResidualCodec(config=cfg, centroids=centroids, avg_residual=None)
    held_codes = tmp_codec.compress_into_codes(held, out_device="cpu")
    held_res = held - tmp_codec.lookup_centroids(held_codes, out_device="cpu")
    avg_residual = torch.abs(held_res).mean(dim=0)
    num_options = 2 ** nbits
    quantiles = torch.arange(0, num_options) * (1 / num_options)
    cut_q, w_q = quantiles[1:], quantiles + (0.5 / num_options)
    bucket_cutoffs = held_res.float().quantile(cut_q)
    bucket_weights = held_res.float().quantile(w_q)

    codec = ResidualCodec(config=cfg, centroids=centroids, avg_residual=avg_residual.mean(),
                          bucket_cutoffs=bucket_cutoffs, bucket_weights=bucket_weights)
    codec.save(index_dir)
    ce = codec.compress(embs_t)
    ce.save(os.path.join(index_dir, "0"))
    with open(os.path.join(index_dir, "doclens.0.json"), "w") as f:
        json.dump([int(x) for x in doclens], f)
    K = protos.shape[0]
    with open(os.path.join(index_dir, "metadata.json"), "w") as f:
        json.dump({"config": {"dim": DIM, "nbits": nbits, "query_maxlen": 32}, "num_chunks": 1,
                   "num_partitions": K, "num_embeddings": int(embs.shape[0]),
                   "avg_doclen": float(embs.shape[0]) / len(doclens)}, f)
    s = ce.codes.long().sort()
    optimize_ivf(s.indices, torch.bincount(s.values, minlength=K), index_dir)
    return codec, ce


def make_queries(seed, protos, doclens, tok_c, spec, sigma=0.05):
    """spec: list of (kind, Nq). Returns list of float32 [Nq,128] arrays."""
    rng = np.random.default_rng(seed)
    K = protos.shape[0]
    offs = np.concatenate([[0], np.cumsum(doclens)])
    out = []
    for kind, Nq in spec:
        if kind == "doc":  # tokens near the centroids of one target document
            t = int(rng.integers(0, len(doclens)))
            while doclens[t] < 4:
                t = int(rng.integers(0, len(doclens)))
            cs = tok_c[offs[t]:offs[t + 1]]
            c = cs[rng.integers(0, len(cs), size=Nq)]
        else:  # random centroids
            c = rng.integers(0, K, size=Nq)
        q = unit(protos[c] + sigma * rng.standard_normal((Nq, DIM)).astype(np.float32)).astype(np.float32)
        if kind == "doc" and Nq > 32:
            q[40:44] = 0.0  # zero rows beyond the candidate-generation window (FLMR pads with zeros)
        out.append(q)
    return out


def tap_rank(scorer, Q, ncells, thr, ndocs, nq_cand=32):
    """Re-compose IndexScorer.rank stage by stage (index_storage.py:86-182) and check it equals rank()."""
    cfg = ColBERTConfig(total_visible_gpus=0, ncells=ncells, centroid_score_threshold=thr, ndocs=ndocs,
                        query_maxlen=nq_cand)
    Qt = torch.from_numpy(Q).unsqueeze(0)
    with torch.inference_mode():
        pids, cs = scorer.retrieve(cfg, Qt)
        cells, _ = scorer.get_cells(Qt[0, :nq_cand], ncells)
        idx = cs.max(-1).values >= thr
        offsets = scorer.embeddings_strided.codes_strided.offsets
        rec = {"Q": Q, "ncells": ncells, "thr": np.float32(thr), "ndocs": ndocs, "nq_cand": nq_cand,
               "centroid_scores": cs.numpy().copy(), "cells": np.sort(cells.numpy()),
               "cand_pids": pids.numpy().copy(), "idx": idx.numpy().copy()}
        if len(pids) < ndocs:
            rec["undefined"] = True  # filter_pids.cpp:119-123 pops an empty heap: reference UB
            return rec
        filt = IndexScorer.filter_pids(pids, cs, scorer.embeddings.codes, scorer.doclens, offsets, idx, ndocs)
        D = IndexScorer.decompress_residuals(
            filt, scorer.doclens, offsets, scorer.codec.bucket_weights, scorer.codec.reversed_bit_map,
            scorer.codec.decompression_lookup_table, scorer.embeddings.residuals, scorer.embeddings.codes,
            scorer.codec.centroids, scorer.codec.dim, scorer.codec.nbits)
        Dn = torch.nn.functional.normalize(D.to(torch.float32), p=2, dim=-1)
        lens = scorer.doclens[filt.long()]
        scores = colbert_score_packed(Qt, Dn, lens, cfg)
        srt = scores.sort(descending=True)
        final_pids = filt[srt.indices]
        r_pids, r_scores = scorer.rank(cfg, Qt)
        assert r_pids == final_pids.tolist() and r_scores == srt.values.tolist(), "stage taps != rank()"
    rec.update({"filtered_pids": filt.numpy().copy(), "doc_scores": scores.numpy().copy(),
                "final_pids": final_pids.numpy().copy(), "final_scores": srt.values.numpy().copy(),
                # first 4 finalists' decompressed (un-normalised) and normalised rows -- full D is too big
                "D_head_len": int(lens[:4].sum()),
                "D_head": D[: int(lens[:4].sum())].numpy().copy(),
                "Dn_head": Dn[: int(lens[:4].sum())].numpy().copy()})
    return rec


def flat(prefix, rec):
    return {f"{prefix}.{k}": np.asarray(v) for k, v in rec.items()}


def gen_index_fixture(name, seed, n_docs, K, nbits, max_len, qspec, configs, workdir):
    protos, doclens, tok_c, embs = make_corpus(seed, n_docs, K, max_len)
    idir = os.path.join(workdir, name)
    codec, ce = build_reference_index(idir, protos, doclens, embs, nbits, seed)
    scorer = IndexScorer(idir, use_gpu=False)
    ColBERT.try_load_torch_extensions(False)
    ivf, ivf_lengths = torch.load(os.path.join(idir, "ivf.pid.pt"))
    N = int(doclens.sum())
    out = {
        "meta.dim": np.int32(DIM), "meta.nbits": np.int32(nbits), "meta.K": np.int32(K),
        "meta.num_embeddings": np.int64(N),
        "index.centroids_f16": torch.load(os.path.join(idir, "centroids.pt")).numpy(),
        "index.bucket_cutoffs": codec.bucket_cutoffs.numpy(), "index.bucket_weights": codec.bucket_weights.numpy(),
        "index.avg_residual": np.float32(float(codec.avg_residual)),
        "index.codes": ce.codes.numpy(), "index.residuals": ce.residuals.numpy(),
        "index.doclens": doclens, "index.ivf": ivf.numpy(), "index.ivf_lengths": ivf_lengths.numpy(),
        "codec.reversed_bit_map": codec.reversed_bit_map.numpy(),
        "codec.decompression_lookup_table": codec.decompression_lookup_table.numpy(),
        # compress() parity vectors for the index-build row: first 256 embeddings
        "compress.embs": embs[:256], "compress.codes": ce.codes[:256].numpy(),
        "compress.residuals": ce.residuals[:256].numpy(),
    }
    # op-level decompress: 24 docs incl. empty / 1-token ones; cross-checked with the torch path
    pids = torch.tensor([3, 7, n_docs // 2, n_docs - 1] + list(range(10, 30)), dtype=torch.int32)
    offsets = scorer.embeddings_strided.codes_strided.offsets
    D = IndexScorer.decompress_residuals(
        pids, scorer.doclens, offsets, codec.bucket_weights, codec.reversed_bit_map,
        codec.decompression_lookup_table, scorer.embeddings.residuals, scorer.embeddings.codes,
        codec.centroids, DIM, nbits)
    eids = torch.cat([torch.arange(int(offsets[p]), int(offsets[p]) + int(scorer.doclens[p])) for p in pids.tolist()])
    D_torch = codec.lookup_centroids(ce.codes[eids], out_device="cpu") + codec.bucket_weights[
        codec.decompression_lookup_table[codec.reversed_bit_map[ce.residuals[eids].long()].long()]
        .reshape(len(eids), -1).long()]
    assert torch.equal(D, D_torch), "decompress_residuals_cpp != torch decompress path"
    out["op_decompress.pids"] = pids.numpy()
    out["op_decompress.D"] = D.numpy()

    queries = make_queries(seed + 1, protos, doclens, tok_c, qspec)
    n_rec = 0
    for qi, Q in enumerate(queries):
        for (ncells, thr, ndocs) in configs:
            rec = tap_rank(scorer, Q, ncells, thr, ndocs)
            out.update(flat(f"rank{n_rec}", rec))
            n_rec += 1
    # dense_search(remove_zero_tensors=True) case (searcher.py:120-126): zero rows INSIDE the first 32
    Qz = make_queries(seed + 2, protos, doclens, tok_c, [("doc", 40)])[0]
    Qz[[1, 5, 6, 11, 17, 20, 21, 22, 30, 31, 36, 39]] = 0.0
    keep = np.abs(Qz).sum(-1) > 0
    rec = tap_rank(scorer, Qz[keep], *configs[0])
    rec["Q_raw"] = Qz
    out.update(flat("rank_rz", rec))
    out["meta.n_rank"] = np.int32(n_rec)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {n_rec} rank records, N={N}")


def gen_ops_fixture():
    """Op-level vectors for segmented_lookup / segmented_maxsim / colbert_score (padded)."""
    rng = np.random.default_rng(7)
    out = {}
    StridedTensor.try_load_torch_extensions(False)
    ColBERT.try_load_torch_extensions(False)
    # segmented_lookup_cpp for each supported dtype (segmented_lookup.cpp:127-144), 1-D and 2-D inputs
    lengths = rng.integers(0, 9, size=40).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    M = int(lengths.sum())
    pids = np.array([5, 0, 39, 17, 17, 3, 22], dtype=np.int64)
    for tag, arr in [("u8", rng.integers(0, 256, size=(M, 16)).astype(np.uint8)),
                     ("i32", rng.integers(-2**31, 2**31 - 1, size=M).astype(np.int32)),
                     ("i64", rng.integers(-2**62, 2**62, size=M).astype(np.int64)),
                     ("f32", rng.standard_normal((M, 8)).astype(np.float32)),
                     ("f16", rng.standard_normal((M, 4)).astype(np.float16))]:
        res = StridedTensor.segmented_lookup(torch.from_numpy(arr), torch.from_numpy(pids),
                                             torch.from_numpy(lengths[pids]), torch.from_numpy(offsets[pids]))
        out[f"lookup.{tag}.input"] = arr
        out[f"lookup.{tag}.output"] = res.numpy()
    out["lookup.pids"] = pids
    out["lookup.lengths"] = lengths
    out["lookup.offsets"] = offsets
    # segmented_maxsim_cpp incl. all-negative docs (zero clamp, segmented_maxsim.cpp:58-59) and empty docs
    dl = np.array([3, 1, 0, 7, 2, 5, 0, 4], dtype=np.int64)
    sc = rng.standard_normal((int(dl.sum()), 32)).astype(np.float32)
    sc[0:3] = -np.abs(sc[0:3])  # doc 0 all negative -> 0
    res = ColBERT.segmented_maxsim(torch.from_numpy(sc), torch.from_numpy(dl))
    out["maxsim.scores"], out["maxsim.lengths"], out["maxsim.output"] = sc, dl, res.numpy()
    sc2 = rng.standard_normal((int(dl.sum()), 45)).astype(np.float32)  # Nq not a multiple of 32
    res2 = ColBERT.segmented_maxsim(torch.from_numpy(sc2), torch.from_numpy(dl))
    out["maxsim45.scores"], out["maxsim45.output"] = sc2, res2.numpy()
    # colbert_score padded (colbert.py:268-286): Q [1,Nq,d] vs D [B,Ld,d] with mask; -9999 padding, no clamp
    Qp = unit(rng.standard_normal((1, 32, DIM)).astype(np.float32))
    Dp = unit(rng.standard_normal((6, 9, DIM)).astype(np.float32))
    lens = np.array([9, 1, 4, 0, 7, 9])
    mask = (np.arange(9)[None, :] < lens[:, None])
    Dp = Dp * mask[..., None]
    res = colbert_score(torch.from_numpy(Qp), torch.from_numpy(Dp.copy()), torch.from_numpy(mask), ColBERTConfig())
    out["padded.Q"], out["padded.D"], out["padded.mask"], out["padded.output"] = Qp, Dp.astype(np.float32), mask, res.numpy()
    Qb = unit(rng.standard_normal((6, 20, DIM)).astype(np.float32))  # per-doc aligned queries
    res = colbert_score(torch.from_numpy(Qb), torch.from_numpy(Dp.astype(np.float32).copy()), torch.from_numpy(mask), ColBERTConfig())
    out["padded_aligned.Q"], out["padded_aligned.output"] = Qb, res.numpy()
    path = os.path.join(HERE, "ops.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


def main():
    torch.set_num_threads(8)
    workdir = tempfile.mkdtemp(prefix="flmr_golden_")
    try:
        std = [(2, 0.45, 256), (1, 0.5, 64)]
        gen_index_fixture("idx_nb2", 11, 3000, 512, 2, 48,
                          [("doc", 32), ("rand", 32), ("doc", 48), ("doc", 96)],
                          [(2, 0.45, 1024), (4, 0.4, 256), (1, 0.5, 64)], workdir)
        gen_index_fixture("idx_nb1", 21, 800, 64, 1, 48, [("doc", 32), ("doc", 48)], std, workdir)
        small = [(2, 0.45, 128), (1, 0.5, 16)]
        gen_index_fixture("idx_nb4", 41, 600, 256, 4, 40, [("doc", 32), ("rand", 32)], small, workdir)
        # last config has P < ndocs on purpose: recorded as "undefined" (reference UB)
        gen_index_fixture("idx_nb8", 81, 400, 256, 8, 40, [("doc", 32), ("doc", 96)], small + [(2, 0.45, 1024)],
                          workdir)
        gen_ops_fixture()
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


if __name__ == "__main__":
    main()
