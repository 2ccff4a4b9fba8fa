#!/usr/bin/env python3
"""Have the REFERENCE write index directories, and keep them (data files only) as loader fixtures.

Run only in the build container (needs /root/reference):

    python tests/golden/make_refindex.py        # writes tests/golden/refindex_2chunk/, refindex_legacy/, refindex.npz

What is written, and by whom
  * refindex_2chunk/ : every file is produced by the reference's own code --
      `IndexSaver.save_codec` -> `ResidualCodec.save`                     (centroids.pt, avg_residual.pt, buckets.pt)
      `IndexSaver.thread()/save_chunk` -> `_write_chunk_to_disk`          ({0,1}.codes.pt, {0,1}.residuals.pt,
                                                                           doclens.{0,1}.json, {0,1}.metadata.json)
      `CollectionIndexer._collect_embedding_id_offset/_build_ivf/_update_metadata`
                                                                          (ivf.pid.pt via optimize_ivf, metadata.json)
    i.e. the tail of `CollectionIndexer.run` (collection_indexer.py:57-76) with the encoder replaced by synthetic
    embeddings (no BERT checkpoint / FAISS in this image: centroids are supplied).
  * refindex_legacy/ : the same chunks, but the variants an index built by an older / GPU run holds:
      legacy `ivf.pt` = (embedding ids sorted by code, lengths) instead of ivf.pid.pt   (index_loader.py:33-36)
      `avg_residual.pt` as an fp16 [dim] vector and `buckets.pt` as fp16 tensors         (residual.py:32-40, :164-167)
  * refindex.npz : what the reference's OWN loader (`IndexScorer(index, use_gpu=False)`: IndexLoader +
    ResidualCodec.load + ResidualEmbeddings.load_chunks) holds in memory after reading refindex_2chunk/, plus one
    `rank()` result on it -- the expected values of tests/test_host_logic.py::test_loader_on_reference_written_index.

Only data files are committed; no reference source text is copied.
"""
import json
import os
import shutil
import sys

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
from colbert.indexing.collection_indexer import CollectionIndexer  # noqa: E402
from colbert.indexing.index_saver import IndexSaver  # noqa: E402
from colbert.modeling.colbert import ColBERT  # noqa: E402
from colbert.search.index_storage import IndexScorer  # noqa: E402

DIM, NBITS, K, NDOCS, SPLIT = 128, 2, 32, 90, 50


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def main():
    rng = np.random.default_rng(4242)
    protos = unit(rng.standard_normal((K, DIM)).astype(np.float32))
    doclens = rng.integers(1, 13, size=NDOCS).astype(np.int64)
    doclens[5] = 0
    doclens[SPLIT] = 1
    N = int(doclens.sum())
    tok_c = rng.integers(0, K, size=N)
    embs = torch.from_numpy(unit(protos[tok_c] + 0.05 * rng.standard_normal((N, DIM)).astype(np.float32)).astype(np.float32))
    centroids = torch.from_numpy(protos).half().float()
    n0 = int(doclens[:SPLIT].sum())

    out = os.path.join(HERE, "refindex_2chunk")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    collection = [f"passage {i}" for i in range(NDOCS)]
    config = ColBERTConfig(dim=DIM, nbits=NBITS, total_visible_gpus=0, index_path=out, collection=collection,
                           checkpoint="synthetic-no-checkpoint", query_maxlen=32, doc_maxlen=16)

    # --- codec exactly as CollectionIndexer.train() builds it (collection_indexer.py:286-316) ------------------
    ci = CollectionIndexer.__new__(CollectionIndexer)
    ci.config, ci.rank, ci.nranks, ci.use_gpu = config, 0, 1, False
    ci.collection, ci.saver = collection, IndexSaver(config)
    ci.num_chunks, ci.num_partitions = 2, K
    ci.num_embeddings_est, ci.avg_doclen_est = float(N), float(N) / NDOCS
    ci._save_plan()                       # setup() writes plan.json first (collection_indexer.py:78-113,187-204)
    heldout = embs[torch.from_numpy(rng.choice(N, size=min(256, N), replace=False))]
    bucket_cutoffs, bucket_weights, avg_residual = ci._compute_avg_residual(centroids, heldout)
    codec = ResidualCodec(config=config, centroids=centroids, avg_residual=avg_residual,
                          bucket_cutoffs=bucket_cutoffs, bucket_weights=bucket_weights)
    ci.saver.save_codec(codec)

    # --- chunks through the saver thread (collection_indexer.py:318-340, index_saver.py:55-90) -------------------
    # _write_chunk_to_disk asks get_dim_and_nbits-free code only; metadata.json does not exist yet, as in a real run
    with ci.saver.thread():
        ci.saver.save_chunk(0, 0, embs[:n0], [int(x) for x in doclens[:SPLIT]])
        ci.saver.save_chunk(1, SPLIT, embs[n0:], [int(x) for x in doclens[SPLIT:]])

    # --- finalize (collection_indexer.py:342-444) ---------------------------------------------------------------------
    ci._collect_embedding_id_offset()
    ci._build_ivf()
    ci._update_metadata()

    # --- what the reference's own loader reads back, and one ranking on it ---------------------------------------
    ColBERT.try_load_torch_extensions(False)
    scorer = IndexScorer(out, use_gpu=False)
    N_ = scorer.num_embeddings
    q = torch.from_numpy(unit(protos[tok_c[: 32]] + 0.05 * rng.standard_normal((32, DIM)).astype(np.float32)).astype(np.float32))
    cfg = ColBERTConfig(total_visible_gpus=0, ncells=2, centroid_score_threshold=0.45, ndocs=64, query_maxlen=32)
    pids, scores = scorer.rank(cfg, q.unsqueeze(0))
    exp = {
        "codes": scorer.embeddings.codes[:N_].numpy(), "residuals": scorer.embeddings.residuals[:N_].numpy(),
        "doclens": scorer.doclens.numpy(), "ivf": scorer.ivf.tensor.numpy(), "ivf_lengths": scorer.ivf.lengths.numpy(),
        "centroids": scorer.codec.centroids.numpy(), "bucket_weights": scorer.codec.bucket_weights.numpy(),
        "bucket_cutoffs": scorer.codec.bucket_cutoffs.numpy(), "avg_residual": np.float32(scorer.codec.avg_residual),
        "num_chunks": np.int64(scorer.num_chunks), "num_embeddings": np.int64(N_),
        "rank.Q": q.numpy(), "rank.pids": np.asarray(pids, dtype=np.int32), "rank.scores": np.asarray(scores, dtype=np.float32),
        "rank.ncells": np.int64(2), "rank.thr": np.float32(0.45), "rank.ndocs": np.int64(64),
    }
    assert exp["codes"].shape[0] == N and scorer.num_chunks == 2
    np.savez_compressed(os.path.join(HERE, "refindex.npz"), **exp)

    # --- legacy / GPU-built variants of the same index -----------------------------------------------------------
    leg = os.path.join(HERE, "refindex_legacy")
    shutil.rmtree(leg, ignore_errors=True)
    shutil.copytree(out, leg)
    os.remove(os.path.join(leg, "ivf.pid.pt"))
    codes = torch.cat([ResidualCodec.Embeddings.load_codes(out, i) for i in range(2)]).long().sort()
    torch.save((codes.indices, torch.bincount(codes.values, minlength=K)), os.path.join(leg, "ivf.pt"))  # _build_ivf's input
    gpu_like = ResidualCodec(config=config, centroids=centroids, avg_residual=torch.full((DIM,), float(avg_residual)).half(),
                             bucket_cutoffs=bucket_cutoffs.half(), bucket_weights=bucket_weights.half())
    gpu_like.bucket_weights = bucket_weights.half()   # the CPU constructor widened it again (residual.py:44-45)
    gpu_like.save(leg)
    for d in (out, leg):
        print(d, sorted(os.listdir(d)), sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)), "bytes")


if __name__ == "__main__":
    main()
