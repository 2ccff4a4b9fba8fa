"""Boundary value types -- host-side mirror of `Queries`, `Ranking`, `Collection`, `Provenance`
(TPC/data/queries.py:11-48, TPC/data/ranking.py:25-55, TPC/data/collection.py, TPC/infra/provenance.py).
Plain containers: dict in, dict out.  File formats: tab-separated `qid<TAB>text` queries and
`qid<TAB>pid<TAB>rank<TAB>score` rankings, as the reference reads / writes them."""
from collections.abc import Sequence as _Sequence
import inspect
import os


class Provenance:
    def __init__(self):
        self.initial_stacktrace = [f"{f.filename}:{f.lineno}:{f.function}" for f in inspect.stack()[1:6]]

    def toDict(self):
        return dict(self.__dict__)


class Queries:
    def __init__(self, path=None, data=None):
        self.path = path
        if data is not None:
            if not isinstance(data, dict):
                raise AssertionError(type(data))
            self.data, qas = {}, {}
            for qid, content in data.items():
                if isinstance(content, dict):
                    self.data[qid] = content["question"]
                    qas[qid] = content
                else:
                    self.data[qid] = content
            if qas:
                self._qas = qas
        else:
            self.data = {}
            with open(path) as f:
                for line in f:
                    qid, text, *_ = line.rstrip("\n").split("\t")
                    self.data[int(qid)] = text

    def __len__(self):
        return len(self.data)

    def __iter__(self):
        return iter(self.data.items())

    def __getitem__(self, key):
        return self.data[key]

    def keys(self):
        return self.data.keys()

    def values(self):
        return self.data.values()

    def items(self):
        return self.data.items()

    def provenance(self):
        return self.path

    def toDict(self):
        return {"provenance": self.provenance()}

    def qas(self):
        return dict(self._qas)

    @classmethod
    def cast(cls, obj):
        if isinstance(obj, str):
            return cls(path=obj)
        if isinstance(obj, dict):
            return cls(data=obj)
        if isinstance(obj, list):
            return cls(data=dict(enumerate(obj)))
        if isinstance(obj, cls):
            return obj
        raise AssertionError(f"obj has type {type(obj)} which is not compatible with cast()")


class RankedList(_Sequence):
    """One query's ranked list, `[(pid, rank, score), ...]` (searcher.py:81-89, :132: ranks are 1..k), as a read-only sequence over
    two numpy rows of the bulk device->host copy.  Tuples are built when they are READ -- an element, a slice, an iteration --
    and hold the same Python ints / floats `Tensor.tolist()` gives (float32 widened).  Building 1024 x 100 tuples eagerly costs
    10 ms per 1024 queries, more than the whole device path (7 ms): a caller that reads `ranking.todict()[qid][:5]` never pays it.
    Compares equal to the list of tuples it stands for; `+` and slicing return plain lists."""
    __slots__ = ("_p", "_s", "_rows")

    def __init__(self, pids, scores):
        self._p, self._s, self._rows = pids, scores, None

    def _all(self):
        if self._rows is None:
            self._rows = list(zip(self._p.tolist(), range(1, len(self._p) + 1), self._s.tolist()))
        return self._rows

    def __len__(self):
        return len(self._p)

    def __getitem__(self, i):
        if isinstance(i, slice):
            start, stop, step = i.indices(len(self._p))
            if self._rows is None and step == 1:     # the common read: a prefix
                stop = max(stop, start)
                return list(zip(self._p[start:stop].tolist(), range(start + 1, stop + 1), self._s[start:stop].tolist()))
            return self._all()[i]
        if self._rows is not None:
            return self._rows[i]
        n = len(self._p)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("ranked list index out of range")
        return (int(self._p[i]), i + 1, float(self._s[i]))

    def __iter__(self):
        return iter(self._all())

    def __eq__(self, other):
        if isinstance(other, RankedList):
            other = other._all()
        return isinstance(other, (list, tuple)) and self._all() == list(other)

    __hash__ = None

    def __add__(self, other):
        return self._all() + list(other)

    def __radd__(self, other):
        return list(other) + self._all()

    def __repr__(self):
        return repr(self._all())

    def tolist(self):
        return list(self._all())

    # vectorised access for callers that can use arrays (no tuples are built): the executor's loop over
    # `(pid, _, score)` (FLMR_executor.py:852-866) is `row.pids` / `row.scores`
    @property
    def pids(self):
        return self._p

    @property
    def scores(self):
        return self._s


def ranked_lists(pids, scores, counts):
    """numpy [n, k] pids / scores + counts -> [RankedList] (rows cut to their count)."""
    return [RankedList(pids[i, :c], scores[i, :c]) for i, c in enumerate(counts)]


class RankedChunk:
    """The rows of ONE device sub-batch of a `_search_all_Q` pass (scorer.PendingBatch): they become readable when the sub-batch's
    copy to pinned host memory has completed, which the first read of any of them waits for -- the device works on the later
    sub-batches meanwhile.  A full read of a row (iteration, `tolist`, comparison) builds the tuples of the WHOLE sub-batch at once:
    one `tolist()` per array and one `zip` with the cyclic collector paused (it otherwise runs a dozen times over 25 k new tuples
    that cannot form cycles), 2-3 x cheaper per tuple than row by row -- a caller that iterates one row iterates them all
    (FLMR_executor.py:852-858).  Prefix slices and single elements are served from the arrays without building anything."""
    __slots__ = ("_wait", "_P", "_S", "_C", "_lists")

    def __init__(self, wait, pids, scores, counts):
        self._wait, self._P, self._S, self._C, self._lists = wait, pids, scores, counts, None

    def arrays(self):
        if self._wait is not None:
            self._wait()           # blocks until this sub-batch's rows are in host memory (raises its deferred device errors)
            self._wait = None
            k = self._P.shape[1] if self._P.ndim == 2 else 0
            self._C = [min(max(int(c), 0), k) for c in self._C.tolist()]
        return self._P, self._S, self._C

    def lists(self):
        if self._lists is None:
            import gc
            P, S, C = self.arrays()
            n, k = P.shape
            was = gc.isenabled()
            gc.disable()
            try:
                flat = list(zip(P.ravel().tolist(), list(range(1, k + 1)) * n, S.ravel().tolist()))
                self._lists = [flat[i * k:i * k + c] for i, c in enumerate(C)]
            finally:
                if was:
                    gc.enable()
        return self._lists


class ChunkRankedList(RankedList):
    """RankedList whose numpy rows arrive with its sub-batch (RankedChunk)."""
    __slots__ = ("_chunk", "_j")

    def __init__(self, chunk, j):
        self._p = self._s = self._rows = None
        self._chunk, self._j = chunk, j

    def _bind(self):
        if self._p is None:
            P, S, C = self._chunk.arrays()
            c = C[self._j]
            self._p, self._s = P[self._j, :c], S[self._j, :c]

    def _all(self):
        if self._rows is None:
            self._rows = self._chunk.lists()[self._j]
            self._bind()
        return self._rows

    def __len__(self):
        self._bind()
        return len(self._p)

    def __getitem__(self, i):
        self._bind()
        return RankedList.__getitem__(self, i)

    @property
    def pids(self):
        self._bind()
        return self._p

    @property
    def scores(self):
        self._bind()
        return self._s


def lazy_flat_ranking(base):
    """A subclass of a Ranking class (this package's or the reference's `colbert.data.Ranking`, ranking.py:25-55) whose
    `flat_ranking` -- [(qid, pid, rank, score)] over all queries, which the reference builds in its constructor -- is built on
    first use (`tolist()` / `save()`): the executors read `todict()` only (FLMR_executor.py:794).  Same name, module and
    behaviour otherwise; `isinstance(r, base)` holds."""
    class Ranking(base):
        def _prepare_data(self, data):
            if isinstance(data, dict):
                self._lazy_flat = None
                return data
            return super()._prepare_data(data)

        @property
        def flat_ranking(self):
            if self.__dict__.get("_lazy_flat") is None:
                self._lazy_flat = [(qid, *rest) for qid, sub in self.data.items() for rest in sub]
            return self._lazy_flat

        @flat_ranking.setter
        def flat_ranking(self, value):
            self._lazy_flat = value

    Ranking.__module__, Ranking.__qualname__, Ranking.__doc__ = base.__module__, base.__qualname__, base.__doc__
    Ranking.reference_class = base
    return Ranking


class Ranking:
    """data = {qid: [(pid, rank, score), ...]}  (searcher.py:81-89)."""

    def __init__(self, path=None, data=None, metrics=None, provenance=None):
        self._provenance = provenance or path or Provenance()
        if data is None:
            rows = []
            with open(path) as f:
                for line in f:
                    rows.append([float(v) if "." in v else int(v) for v in line.strip().split("\t")])
            data = rows
        if isinstance(data, dict):
            self._flat = None        # built on first use (tolist / save): the executors read todict() only
            self.data = data
        else:
            self._flat = data
            grouped = {}
            for qid, *rest in data:
                grouped.setdefault(qid, []).append(tuple(rest))
            self.data = grouped

    @property
    def flat_ranking(self):
        if self._flat is None:
            self._flat = [(qid, *rest) for qid, sub in self.data.items() for rest in sub]
        return self._flat

    def provenance(self):
        return self._provenance

    def toDict(self):
        return {"provenance": self.provenance()}

    def todict(self):
        return dict(self.data)

    def tolist(self):
        return list(self.flat_ranking)

    def items(self):
        return self.data.items()

    def save(self, new_path):
        os.makedirs(os.path.dirname(os.path.abspath(new_path)), exist_ok=True)
        with open(new_path, "w") as f:
            for items in self.flat_ranking:
                f.write("\t".join(str(int(x) if isinstance(x, bool) else x) for x in items) + "\n")
        return new_path

    @classmethod
    def cast(cls, obj):
        if isinstance(obj, str):
            return cls(path=obj)
        if isinstance(obj, (dict, list)):
            return cls(data=obj)
        if isinstance(obj, cls):
            return obj
        raise AssertionError(f"obj has type {type(obj)} which is not compatible with cast()")


class Collection:
    def __init__(self, path=None, data=None):
        self.path = path
        if data is None and path is not None:
            data = []
            with open(path) as f:
                for line in f:
                    pid, passage, *rest = line.rstrip("\n\r").split("\t")
                    data.append((rest[0] + " | " + passage) if rest else passage)
        self.data = data if data is not None else []

    def __iter__(self):
        return iter(self.data)

    def __getitem__(self, item):
        return self.data[item]

    def __len__(self):
        return len(self.data)

    def provenance(self):
        return self.path

    @classmethod
    def cast(cls, obj):
        if obj is None:
            return cls(data=[])
        if isinstance(obj, str):
            return cls(path=obj)
        if isinstance(obj, list):
            return cls(data=obj)
        if isinstance(obj, cls):
            return obj
        raise AssertionError(f"obj has type {type(obj)} which is not compatible with cast()")
