"""Index loading: the reference's on-disk format -> host arrays -> HBM.

Host-side mirror of IndexLoader (TPC/search/index_loader.py:14-86), ResidualCodec.load
(TPC/indexing/codecs/residual.py:134-150), ResidualEmbeddings.load_chunks
(TPC/indexing/codecs/residual_embeddings.py:27-52) and optimize_ivf (TPC/indexing/utils.py:8-53, for
legacy `ivf.pt` files).  Files are parsed with torch.load / json on the host; `DeviceIndex` hands the
arrays to libflmr_hip (flmr_index_open), which owns the HBM copies.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _native


def codec_tables(nbits):
    """reversed_bit_map u8[256] and decompression_lookup_table u8[256, 8/nbits] as ResidualCodec.__init__
    builds them (residual.py:51-95): bits reversed inside each nbits-wide group; base-2^nbits digits, most
    significant first."""
    vpb, mask = 8 // nbits, (1 << nbits) - 1
    i = np.arange(256)
    rev = np.zeros(256, dtype=np.int64)
    lut = np.zeros((256, vpb), dtype=np.uint8)
    for g in range(vpb):
        sh = 8 - nbits * (g + 1)
        x = (i >> sh) & mask
        y = np.zeros(256, dtype=np.int64)
        for b in range(nbits):
            y |= ((x >> b) & 1) << (nbits - 1 - b)
        rev |= y << sh
        lut[:, g] = x
    return rev.astype(np.uint8), lut


class IndexArrays:
    """Host arrays of one index (or one shard of it) in the reference's CPU layout."""

    def __init__(self, dim, nbits, codes, residuals, doclens, ivf, ivf_lengths, centroids, bucket_weights,
                 bucket_cutoffs=None, avg_residual=None, pid_base=0, config=None):
        self.dim, self.nbits = int(dim), int(nbits)
        self.codes = np.ascontiguousarray(codes, dtype=np.int32)
        self.residuals = np.ascontiguousarray(residuals, dtype=np.uint8)
        self.doclens = np.ascontiguousarray(doclens, dtype=np.int64)
        self.doc_offsets = np.concatenate([[0], np.cumsum(self.doclens)]).astype(np.int64)
        self.ivf = np.ascontiguousarray(ivf, dtype=np.int32)
        self.ivf_lengths = np.ascontiguousarray(ivf_lengths, dtype=np.int64)
        self.ivf_offsets = np.concatenate([[0], np.cumsum(self.ivf_lengths)]).astype(np.int64)
        self.centroids = np.ascontiguousarray(centroids, dtype=np.float32)
        self.bucket_weights = np.ascontiguousarray(bucket_weights, dtype=np.float32)
        self.bucket_cutoffs = bucket_cutoffs
        self.avg_residual = avg_residual
        self.pid_base = int(pid_base)
        self.config = config or {}
        if self.residuals.ndim != 2 or self.residuals.shape[1] != self.dim * self.nbits // 8:
            raise ValueError(f"residuals shape {self.residuals.shape} != [N, {self.dim * self.nbits // 8}]")
        if self.codes.shape[0] != self.doc_offsets[-1] or self.residuals.shape[0] != self.codes.shape[0]:
            raise ValueError("codes / residuals / doclens disagree on the number of embeddings")
        if self.ivf_lengths.shape[0] != self.centroids.shape[0]:
            raise ValueError("ivf_lengths must have one entry per centroid")

    @property
    def num_passages(self):
        return int(self.doclens.shape[0])

    @property
    def num_embeddings(self):
        return int(self.codes.shape[0])

    @property
    def num_centroids(self):
        return int(self.centroids.shape[0])

    @classmethod
    def from_golden(cls, z):
        """Build from a tests/golden/*.npz fixture dict."""
        return cls(int(z["meta.dim"]), int(z["meta.nbits"]), z["index.codes"], z["index.residuals"], z["index.doclens"],
                   z["index.ivf"], z["index.ivf_lengths"], z["index.centroids_f16"].astype(np.float32),
                   z["index.bucket_weights"], bucket_cutoffs=z["index.bucket_cutoffs"])

    def shard(self, rank, world_size):
        """Contiguous pid-range shard `rank` of `world_size`: slices of codes/residuals/doclens,
        an IVF restricted to the shard's pids (lists stay sorted; local pid = global pid - pid_base);
        centroids and bucket tables are replicated."""
        n = self.num_passages
        lo, hi = (n * rank) // world_size, (n * (rank + 1)) // world_size
        t0, t1 = int(self.doc_offsets[lo]), int(self.doc_offsets[hi])
        keep = (self.ivf >= lo) & (self.ivf < hi)
        owner = np.repeat(np.arange(self.num_centroids), self.ivf_lengths)
        ivf_lengths = np.bincount(owner[keep], minlength=self.num_centroids).astype(np.int64)
        return IndexArrays(self.dim, self.nbits, self.codes[t0:t1], self.residuals[t0:t1], self.doclens[lo:hi],
                           self.ivf[keep] - lo, ivf_lengths, self.centroids, self.bucket_weights,
                           bucket_cutoffs=self.bucket_cutoffs, avg_residual=self.avg_residual,
                           pid_base=self.pid_base + lo, config=self.config)

    def check_ivf_invariant(self):
        """True iff `ivf[c]` is exactly the sorted set of passages whose codes contain centroid c -- what `optimize_ivf`
        (colbert/indexing/utils.py:8-53) produces.  Candidate generation (here and in the reference) trusts these lists,
        and this build's default stage 1 derives the per-passage centroid sets from them instead of scanning the codes."""
        n = int(self.doclens.shape[0])
        ntok = int(np.asarray(self.doclens, dtype=np.int64).sum())
        pid_of = np.repeat(np.arange(n, dtype=np.int64), np.asarray(self.doclens, dtype=np.int64))
        key = np.unique(np.asarray(self.codes[:ntok], dtype=np.int64) * n + pid_of)
        lens = np.bincount(key // n, minlength=int(self.centroids.shape[0]))
        total = int(np.asarray(self.ivf_lengths, dtype=np.int64).sum())
        return bool(np.array_equal(lens, np.asarray(self.ivf_lengths, dtype=np.int64)) and
                    np.array_equal((key % n).astype(np.int64), np.asarray(self.ivf[:total], dtype=np.int64)))

    def save(self, index_path):
        """Write the reference's on-disk format (single chunk): used by tests and by the synthetic bench."""
        import torch
        os.makedirs(index_path, exist_ok=True)
        torch.save(torch.from_numpy(self.centroids).half(), os.path.join(index_path, "centroids.pt"))
        cut = self.bucket_cutoffs if self.bucket_cutoffs is not None else np.zeros(2 ** self.nbits - 1, np.float32)
        torch.save((torch.from_numpy(np.asarray(cut, dtype=np.float32)), torch.from_numpy(self.bucket_weights)),
                   os.path.join(index_path, "buckets.pt"))
        torch.save(torch.tensor([float(self.avg_residual or 0.0)]), os.path.join(index_path, "avg_residual.pt"))
        torch.save(torch.from_numpy(self.codes), os.path.join(index_path, "0.codes.pt"))
        torch.save(torch.from_numpy(self.residuals), os.path.join(index_path, "0.residuals.pt"))
        with open(os.path.join(index_path, "doclens.0.json"), "w") as f:
            json.dump([int(x) for x in self.doclens], f)
        torch.save((torch.from_numpy(self.ivf), torch.from_numpy(self.ivf_lengths)), os.path.join(index_path, "ivf.pid.pt"))
        cfg = dict(self.config)
        cfg.update({"dim": self.dim, "nbits": self.nbits})
        cfg.setdefault("query_maxlen", 32)
        with open(os.path.join(index_path, "metadata.json"), "w") as f:
            json.dump({"config": cfg, "num_chunks": 1, "num_partitions": self.num_centroids,
                       "num_embeddings": self.num_embeddings,
                       "avg_doclen": self.num_embeddings / max(1, self.num_passages)}, f)


def _ivf_to_pid_lists(eids, lengths, doclens):
    """optimize_ivf (indexing/utils.py:8-53): embedding-id lists -> sorted unique pid lists per centroid."""
    emb2pid = np.repeat(np.arange(len(doclens), dtype=np.int32), doclens)
    pid_of = emb2pid[eids]
    out, out_len, off = [], [], 0
    for ln in lengths.tolist():
        u = np.unique(pid_of[off:off + ln])
        out.append(u)
        out_len.append(len(u))
        off += ln
    return (np.concatenate(out) if out else np.zeros(0, np.int32)), np.asarray(out_len, dtype=np.int64)


def load_index_arrays(index_path):
    """Parse an index directory written by the reference's indexer (Appendix A)."""
    import torch
    meta_path = os.path.join(index_path, "metadata.json")
    if not os.path.exists(meta_path):
        meta_path = os.path.join(index_path, "plan.json")
    with open(meta_path) as f:
        meta = json.load(f)
    cfg = meta.get("config", meta)
    dim, nbits = int(cfg["dim"]), int(cfg["nbits"])
    num_chunks = int(meta["num_chunks"])

    def tload(name):
        return torch.load(os.path.join(index_path, name), map_location="cpu")

    centroids = tload("centroids.pt").float().numpy()
    bucket_cutoffs, bucket_weights = tload("buckets.pt")
    avg_residual = tload("avg_residual.pt")
    avg_residual = float(avg_residual.float().mean()) if avg_residual.numel() else 0.0
    codes, residuals, doclens = [], [], []
    for i in range(num_chunks):
        codes.append(tload(f"{i}.codes.pt").to(torch.int32).numpy())
        residuals.append(tload(f"{i}.residuals.pt").numpy())
        with open(os.path.join(index_path, f"doclens.{i}.json")) as f:
            doclens.extend(json.load(f))
    codes = np.concatenate(codes)
    residuals = np.concatenate(residuals)
    doclens = np.asarray(doclens, dtype=np.int64)
    if os.path.exists(os.path.join(index_path, "ivf.pid.pt")):
        ivf, ivf_lengths = tload("ivf.pid.pt")
        ivf, ivf_lengths = ivf.to(torch.int32).numpy(), torch.as_tensor(ivf_lengths).long().numpy()
    else:
        eids, lengths = tload("ivf.pt")
        ivf, ivf_lengths = _ivf_to_pid_lists(eids.long().numpy(), torch.as_tensor(lengths).long().numpy(), doclens)
    if "num_embeddings" in meta and int(meta["num_embeddings"]) != codes.shape[0]:
        raise ValueError("metadata.json num_embeddings does not match the chunk files")
    return IndexArrays(dim, nbits, codes, residuals, doclens, ivf, ivf_lengths, centroids,
                       bucket_weights.float().numpy(), bucket_cutoffs=bucket_cutoffs.float().numpy(),
                       avg_residual=avg_residual, config=cfg)


class DeviceIndex:
    """An index resident in HBM (handle to a `flmr_index_t`).  `arrays` may be host numpy arrays (the library
    copies them) or torch CUDA tensors with the same field names (borrowed, kept alive here)."""

    def __init__(self, arrays: IndexArrays, device_tensors=None):
        lib = _native.load(require_device=True)
        self.arrays = arrays
        self._keep = device_tensors
        d = _native.IndexDesc()
        d.dim, d.nbits, d.num_centroids = arrays.dim, arrays.nbits, arrays.num_centroids
        d.num_embeddings, d.num_passages, d.pid_base = arrays.num_embeddings, arrays.num_passages, arrays.pid_base
        bw = np.ascontiguousarray(arrays.bucket_weights, dtype=np.float32)
        self._bw = bw
        d.bucket_weights = bw.ctypes.data
        if device_tensors is None:
            d.memory = _native.FLMR_MEM_HOST
            for field, arr in (("codes", arrays.codes), ("residuals", arrays.residuals),
                               ("doc_offsets", arrays.doc_offsets), ("ivf_pids", arrays.ivf),
                               ("ivf_offsets", arrays.ivf_offsets), ("centroids", arrays.centroids)):
                setattr(d, field, arr.ctypes.data)
        else:
            d.memory = _native.FLMR_MEM_DEVICE
            for field in ("codes", "residuals", "doc_offsets", "ivf_pids", "ivf_offsets", "centroids"):
                t = device_tensors[field]
                assert t.is_cuda and t.is_contiguous(), field
                setattr(d, field, t.data_ptr())
        h = C.c_void_p()
        _native.check(lib.flmr_index_open(C.byref(d), C.byref(h)))
        self.handle = h
        self._lib = lib

    def info(self):
        """dict of what flmr_index_open derived on this device (flmr_index_info): stage-2 slice count, the XCD dispatch
        probe, derived HBM bytes, ..."""
        out = _native.IndexInfo()
        _native.check(self._lib.flmr_index_info(self.handle, C.byref(out)))
        return {name: getattr(out, name) for name, _ in out._fields_ if name != "reserved"}

    def close(self):
        if getattr(self, "handle", None):
            self._lib.flmr_index_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
