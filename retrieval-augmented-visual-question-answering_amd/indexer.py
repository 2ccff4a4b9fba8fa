"""`Indexer` -- the reference's index-build entry point (TPC/indexer.py:15-85) over this build's GPU index build
(`indexing.build_index`: k-means in torch, nearest-centroid argmax + residual bit-packing on the HIP kernels of
csrc/flmr_build.hip, IVF), writing the reference's on-disk format.

Kept from the reference: `Indexer(checkpoint, config)`, config merge order (checkpoint config < `config` < the active
`Run().config`, indexer.py:25-26), `configure`, `get_index`, `erase` (same file selection, indexer.py:35-56), and
`index(name, collection, overwrite)` with `overwrite in {True, False, 'reuse', 'resume'}` and the same assertions /
path resolution (`config.index_path_` = {root}/{experiment}/indexes/{name}).

What differs: passage ENCODING (the BERT / ViT forward of `CollectionEncoder`, collection_indexer.py:318-340) is outside
this build's scope, so the embeddings come from `doc_encoder(list_of_passages) -> (embeddings [N, 128], doclens [P])` given
at construction, or `collection` may already be an `(embeddings, doclens)` pair.  There is no multi-process launcher: one
process builds the index on its GPU (the reference spawns `nranks` encoder processes, indexer.py:78-84).  'resume' has
nothing to resume from (single chunk) and behaves like `True`.  When `ravqa_amd.install()` is active the reference's own
`colbert.Indexer` is left in place -- this class is for running without the reference package.
"""
import os
import time

import torch

from . import config as _config
from .indexing import build_index


class Indexer:
    ColBERTConfig = _config.ColBERTConfig
    Run = _config.Run

    def __init__(self, checkpoint=None, config=None, doc_encoder=None, build_backend=None):
        self.index_path = None
        self.build_backend = build_backend   # None: the HIP kernels (indexing.HipBackend); tests of the host logic inject a torch one
        self.checkpoint = checkpoint
        self.checkpoint_config = self.ColBERTConfig.load_from_checkpoint(checkpoint) if isinstance(checkpoint, str) else None
        self.config = self.ColBERTConfig.from_existing(self.checkpoint_config, config, self.Run().config)
        self.configure(checkpoint=checkpoint)
        self.doc_encoder = doc_encoder

    def configure(self, **kw_args):
        self.config.configure(**kw_args)

    def get_index(self):
        return self.index_path

    def erase(self, wait_seconds=0):
        assert self.index_path is not None
        directory, deleted = self.index_path, []
        for filename in sorted(os.listdir(directory)):
            path = os.path.join(directory, filename)
            delete = path.endswith(".json") and ("metadata" in path or "doclen" in path or "plan" in path)
            if delete or path.endswith(".pt"):
                deleted.append(path)
        if deleted:
            if wait_seconds:
                time.sleep(wait_seconds)   # the reference waits 3 s before deleting (indexer.py:50-51)
            for path in deleted:
                os.remove(path)
        return deleted

    def _embeddings_of(self, collection):
        if isinstance(collection, (tuple, list)) and len(collection) == 2 and torch.is_tensor(collection[0]):
            return collection
        if self.doc_encoder is None:
            raise NotImplementedError("passage encoding (CollectionEncoder.encode_passages) is outside the retrieval hot path: "
                                      "construct Indexer(..., doc_encoder=fn) or pass collection=(embeddings, doclens)")
        passages = list(collection.data) if hasattr(collection, "data") else list(collection)
        return self.doc_encoder(passages)

    def index(self, name, collection, overwrite=False):
        assert overwrite in [True, False, "reuse", "resume"]
        self.configure(index_name=name, resume=overwrite == "resume")
        self.index_path = self.config.index_path_
        index_does_not_exist = not os.path.exists(self.index_path)
        assert (overwrite in [True, "reuse", "resume"]) or index_does_not_exist, self.index_path
        os.makedirs(self.index_path, exist_ok=True)
        if overwrite is True or overwrite == "resume":
            self.erase()
        if index_does_not_exist or overwrite != "reuse":
            embeddings, doclens = self._embeddings_of(collection)
            if self.build_backend is None and not embeddings.is_cuda:
                embeddings = embeddings.cuda()   # the build runs on the MI355X (no host path: fails loudly without a device)
            cfg = {"query_maxlen": self.config.query_maxlen, "doc_maxlen": self.config.doc_maxlen,
                   "checkpoint": self.checkpoint if isinstance(self.checkpoint, str) else None, "index_name": name}
            kw = {} if self.build_backend is None else {"backend": self.build_backend}
            arrays = build_index(embeddings, torch.as_tensor(doclens), nbits=self.config.nbits,
                                 kmeans_niters=self.config.kmeans_niters, config=cfg, **kw)
            arrays.save(self.index_path)
        return self.index_path
