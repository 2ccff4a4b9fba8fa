"""CPU stand-in for `ravqa_amd.scorer.IndexScorer` in the multi-process tests of the exact sharded protocol: the same
method surface `ShardedSearcher.search_batch_exact` drives (probe_dims / supports_query_split / probe / phase1 /
phase1_probed / phase2 / phase3), each phase computed with the CPU oracle's primitives on this rank's passage shard, and
numpy restatements of the key ops (`topn_keys`, `unpack_keys`).  TEST INFRASTRUCTURE ONLY: what the tests exercise is the
product's host code in distributed.py -- slice arithmetic, the real torch.distributed collectives, slot alignment, the
capability vote -- with the kernels replaced by the checker."""
import numpy as np
import torch

from oracle import oracle as orc


def f2ord(scores):
    """order-preserving u32 image of fp32 scores (csrc/flmr_common.h: flmr_f2ord)."""
    s = np.asarray(scores, dtype=np.float32).copy()
    s[s == 0.0] = 0.0
    u = s.view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def ord2f(o):
    o = np.asarray(o, dtype=np.uint32)
    u = np.where(o & np.uint32(0x80000000), o & np.uint32(0x7FFFFFFF), ~o).astype(np.uint32)
    return u.view(np.float32)


def make_keys(scores, pids):
    return (f2ord(scores).astype(np.uint64) << np.uint64(32)) | np.asarray(pids, dtype=np.int64).astype(np.uint64)


def topn_keys(keys, n, ordered=True):
    """keys int64 [B, m] (u64 bit patterns, 0 = empty) -> the n largest per row, descending, 0 padded."""
    k = np.ascontiguousarray(keys.numpy()).view(np.uint64)
    out = np.zeros((k.shape[0], n), dtype=np.uint64)
    srt = np.sort(k, axis=1)[:, ::-1]
    m = min(n, k.shape[1])
    out[:, :m] = srt[:, :m]
    return torch.from_numpy(out.view(np.int64))


def unpack_keys(keys, k):
    u = np.ascontiguousarray(keys.numpy()).view(np.uint64)[:, :k]
    B = u.shape[0]
    pids = np.full((B, k), -1, dtype=np.int32)
    scores = np.zeros((B, k), dtype=np.float32)
    valid = u != 0
    pids[:, : u.shape[1]][valid] = (u[valid] & np.uint64(0xFFFFFFFF)).astype(np.int64).astype(np.int32)
    scores[:, : u.shape[1]][valid] = ord2f((u[valid] >> np.uint64(32)).astype(np.uint32))
    return torch.from_numpy(pids), torch.from_numpy(scores), torch.from_numpy(valid.sum(axis=1).astype(np.int32))


class OracleShardScorer:
    probe_device = "cpu"

    def __init__(self, shard_arrays, query_split=True):
        a = shard_arrays
        self.a = a
        self.oi = orc.OracleIndex(a.dim, a.nbits, a.codes, a.residuals, a.doclens, a.ivf, a.ivf_lengths, a.centroids,
                                  a.bucket_weights)
        self.query_split = query_split
        self.calls = []

    def clone(self):
        """Another scorer on the same shard with its own phase state (a second in-flight sub-batch of the pipelined protocol)."""
        other = OracleShardScorer(self.a, self.query_split)
        other.calls = self.calls     # one call log per rank
        return other

    # ---- stage 0 ------------------------------------------------------------------------------------------------
    def _stage0(self, q, qlen, nq_cand, ncells, thr):
        nqc = min(nq_cand, qlen)
        cs = self.oi.centroid_scores(q[:nqc])
        return cs, orc.idx_mask(cs, thr), orc.select_cells(cs, ncells)

    def supports_query_split(self, Q, k, ncells, thr, ndocs, nq_cand=32):
        return self.query_split

    def probe_dims(self, Q, k, ncells, thr, ndocs, nq_cand=32):
        return (self.oi.K + 31) // 32, min(nq_cand, Q.size(1)) * ncells

    def probe(self, Q, k, ncells, thr, ndocs, q_begin, q_count, nq_cand=32, q_lens=None, out=None):
        self.calls.append(("probe", q_begin, q_count))
        bits, cells, ncell = out
        Qn = Q.numpy()
        for j in range(q_count):
            b = q_begin + j
            qlen = int(q_lens[b]) if q_lens is not None else Qn.shape[1]
            _, idx, cl = self._stage0(Qn[b], qlen, nq_cand, ncells, thr)
            packed = np.packbits(np.pad(idx, (0, bits.size(1) * 32 - len(idx))), bitorder="little").view(np.int32)
            bits[j] = torch.from_numpy(packed.copy())
            cells[j, : len(cl)] = torch.from_numpy(cl)
            ncell[j] = len(cl)
        return out

    # ---- phases ---------------------------------------------------------------------------------------------------
    def _phase1_one(self, q, qlen, nq_cand, ncells, thr, ndocs, idx=None, cells=None):
        cs, idx0, cells0 = self._stage0(q, qlen, nq_cand, ncells, thr)
        idx = idx0 if idx is None else idx
        cells = cells0 if cells is None else cells
        cand = self.oi.candidates(cells)
        p, s = self.oi.filter_pass(cand, cs, idx, ndocs)
        row = np.zeros(ndocs, dtype=np.uint64)
        row[: len(p)] = make_keys(s, p.astype(np.int64) + self.a.pid_base)
        return row, cs

    def phase1(self, Q, k, ncells, thr, ndocs, nq_cand=32, q_lens=None):
        self.calls.append(("phase1",))
        return self._phase1(Q, ncells, thr, ndocs, nq_cand, q_lens, None, None, None)

    def phase1_probed(self, Q, k, ncells, thr, ndocs, idx_bits, cells, ncell, nq_cand=32, q_lens=None):
        self.calls.append(("phase1_probed",))
        return self._phase1(Q, ncells, thr, ndocs, nq_cand, q_lens, idx_bits, cells, ncell)

    def _phase1(self, Q, ncells, thr, ndocs, nq_cand, q_lens, bits, cells, ncell):
        Qn = Q.numpy()
        self._state = dict(Q=Qn, q_lens=q_lens, nq_cand=nq_cand, ndocs=ndocs, cs=[])
        out = np.zeros((Qn.shape[0], ndocs), dtype=np.uint64)
        for b in range(Qn.shape[0]):
            qlen = int(q_lens[b]) if q_lens is not None else Qn.shape[1]
            idx = cl = None
            if bits is not None:
                idx = np.unpackbits(bits[b].numpy().view(np.uint8), bitorder="little")[: self.oi.K].astype(bool)
                cl = cells[b, : int(ncell[b])].numpy().astype(np.int32)
            out[b], cs = self._phase1_one(Qn[b], qlen, nq_cand, ncells, thr, ndocs, idx, cl)
            self._state["cs"].append(cs)
        return torch.from_numpy(out.view(np.int64))

    def _mine(self, key_row):
        u = np.ascontiguousarray(key_row.numpy()).view(np.uint64)
        pid = (u & np.uint64(0xFFFFFFFF)).astype(np.int64)
        mine = (u != 0) & (pid >= self.a.pid_base) & (pid < self.a.pid_base + self.a.num_passages)
        return np.nonzero(mine)[0], (pid[mine] - self.a.pid_base).astype(np.int32)

    def phase2(self, global_s1):
        st = self._state
        out = np.zeros((global_s1.size(0), st["ndocs"]), dtype=np.uint64)
        for b in range(global_s1.size(0)):
            slots, local = self._mine(global_s1[b])
            if len(local):
                p, s = self.oi.filter_pass(local, st["cs"][b], None, len(local))   # all centroids, every member scored
                score_of = dict(zip(p.tolist(), s.tolist()))
                out[b, slots] = make_keys([score_of[int(x)] for x in local], local.astype(np.int64) + self.a.pid_base)
        return torch.from_numpy(out.view(np.int64))

    def phase3(self, global_s2):
        st = self._state
        out = np.zeros((global_s2.size(0), st["ndocs"] // 4), dtype=np.uint64)
        for b in range(global_s2.size(0)):
            slots, local = self._mine(global_s2[b])
            if len(local):
                qlen = int(st["q_lens"][b]) if st["q_lens"] is not None else st["Q"].shape[1]
                D = orc.normalize_rows(self.oi.decompress(local))
                sc = orc.maxsim_packed(D, st["Q"][b][:qlen], self.oi.doclens[local])
                out[b, slots] = make_keys(sc, local.astype(np.int64) + self.a.pid_base)
        return torch.from_numpy(out.view(np.int64))
