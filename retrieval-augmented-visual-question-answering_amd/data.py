"""Boundary value types -- host-side mirror of `Queries`, `Ranking`, `Collection`, `Provenance`
(TPC/data/queries.py:11-48, TPC/data/ranking.py:25-55, TPC/data/collection.py, TPC/infra/provenance.py).
Plain containers: dict in, dict out.  File formats: tab-separated `qid<TAB>text` queries and
`qid<TAB>pid<TAB>rank<TAB>score` rankings, as the reference reads / writes them."""
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
