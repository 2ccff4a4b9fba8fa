"""Configuration objects read by the search path -- host-side mirror of the reference's
`ColBERTConfig` / `RunConfig` / `Run` (TPC/infra/config/{core_config,base_config,settings,config}.py,
TPC/infra/run.py).  Only the fields the retrieval path reads are modelled; unknown keys found in an
index's metadata.json are ignored exactly as `from_deprecated_args(ignore_unrecognized=True)` does
(base_config.py:33-38).

Semantics kept from the reference:
  * every field has a default; a field counts as "assigned" only if the caller passed a non-None value
    (core_config.py:21-34), and `from_existing(*sources)` merges ONLY assigned fields, later sources
    winning (base_config.py:17-31);
  * `configure(**kw)` / `set()` mark fields assigned (core_config.py:41-66);
  * `index_root_` = index_root or {root}/{experiment}/indexes/ (settings.py:50-52);
  * `Run()` is a process-wide singleton holding a stack of RunConfig; `Run().context(cfg)` pushes a
    config merged over the current one (run.py:10-60).
"""
import json
import os
from contextlib import contextmanager

_RUN_FIELDS = {
    "overwrite": False, "root": None, "experiment": "default", "index_root": None, "name": None,
    "rank": 0, "nranks": 1, "amp": True, "total_visible_gpus": None, "gpus": None,
}
_COLBERT_FIELDS = dict(_RUN_FIELDS, **{
    "checkpoint": None, "triples": None, "collection": None, "queries": None, "index_name": None,
    "dim": 128, "doc_maxlen": 220, "mask_punctuation": True,
    "query_maxlen": 32, "attend_to_mask_tokens": False, "interaction": "colbert",
    "index_path": None, "nbits": 1, "kmeans_niters": 4, "resume": False,
    "ncells": None, "centroid_score_threshold": None, "ndocs": None,
    # TrainingSettings (settings.py:114-144): never read by the search path, but the executors construct ColBERTConfig with
    # them (FLMR_executor.py:129-134: bsize, use_ib_negatives) and index metadata carries them
    "similarity": "cosine", "bsize": 32, "accumsteps": 1, "lr": 3e-06, "maxsteps": 500_000, "save_every": None,
    "warmup": None, "warmup_bert": None, "relu": False, "nway": 2, "use_ib_negatives": False, "reranker": False,
    "distillation_alpha": 1.0, "ignore_scores": False,
})


def _default_of(key):
    if key == "root":
        return os.path.join(os.getcwd(), "experiments")
    if key in ("total_visible_gpus", "gpus"):
        try:
            import torch
            return torch.cuda.device_count()
        except Exception:  # pragma: no cover
            return 0
    return _COLBERT_FIELDS[key]


class _Config:
    _FIELDS = _COLBERT_FIELDS

    def __init__(self, **kw):
        unknown = set(kw) - set(self._FIELDS)
        if unknown:
            raise TypeError(f"{type(self).__name__}: unexpected fields {sorted(unknown)}")
        self.assigned = {}
        for key in self._FIELDS:
            val = kw.get(key)
            if val is None:
                val = _default_of(key)
            else:
                self.assigned[key] = True
            object.__setattr__(self, key, val)

    # --- core_config.py:36-66 ---------------------------------------------------------------
    def assign_defaults(self):
        for key in self._FIELDS:
            object.__setattr__(self, key, _default_of(key))
            self.assigned[key] = True

    def set(self, key, value, ignore_unrecognized=False):
        if key in self._FIELDS:
            object.__setattr__(self, key, value)
            self.assigned[key] = True
            return True
        if not ignore_unrecognized:
            raise Exception(f"Unrecognized key `{key}` for {type(self)}")
        return False

    def configure(self, ignore_unrecognized=True, **kw):
        ignored = set()
        for key, value in kw.items():
            if not self.set(key, value, ignore_unrecognized):
                ignored.add(key)
        return ignored

    def export(self):
        return {k: getattr(self, k) for k in self._FIELDS}

    # --- base_config.py:17-87 ---------------------------------------------------------------
    @classmethod
    def from_existing(cls, *sources):
        kw = {}
        for src in sources:
            if src is None:
                continue
            for k in src.assigned:
                if k in cls._FIELDS:
                    kw[k] = getattr(src, k)
        return cls(**kw)

    @classmethod
    def from_deprecated_args(cls, args):
        obj = cls()
        ignored = obj.configure(ignore_unrecognized=True, **args)
        return obj, ignored

    @classmethod
    def from_path(cls, name):
        with open(name) as f:
            args = json.load(f)
        if "config" in args:
            args = args["config"]
        return cls.from_deprecated_args(args)

    @classmethod
    def load_from_index(cls, index_path):
        try:
            cfg, _ = cls.from_path(os.path.join(index_path, "metadata.json"))
        except Exception:
            cfg, _ = cls.from_path(os.path.join(index_path, "plan.json"))
        return cfg

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path):
        if checkpoint_path is None:
            return None
        p = os.path.join(checkpoint_path, "artifact.metadata")
        if os.path.exists(p):
            cfg, _ = cls.from_path(p)
            cfg.set("checkpoint", checkpoint_path)
            return cfg
        return None

    # --- settings.py:50-52,157-159 --------------------------------------------------------------
    @property
    def index_root_(self):
        return self.index_root or os.path.join(self.root, self.experiment, "indexes/")

    @property
    def index_path_(self):
        return self.index_path or os.path.join(self.index_root_, self.index_name)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={getattr(self, k)!r}' for k in sorted(self.assigned))})"


class ColBERTConfig(_Config):
    _FIELDS = _COLBERT_FIELDS


class RunConfig(_Config):
    _FIELDS = _RUN_FIELDS


class Run:
    """Process-wide stack of RunConfig (TPC/infra/run.py)."""
    _instance = None

    def __new__(cls):
        if cls._instance is None:
            cls._instance = super().__new__(cls)
            base = RunConfig()
            base.assign_defaults()
            cls._instance.stack = [base]
        return cls._instance

    @property
    def config(self):
        return self.stack[-1]

    def __getattr__(self, name):
        cfg = object.__getattribute__(self, "stack")[-1]
        if hasattr(cfg, name):
            return getattr(cfg, name)
        raise AttributeError(name)

    @contextmanager
    def context(self, runconfig, inherit_config=True):
        if inherit_config:
            runconfig = RunConfig.from_existing(self.config, runconfig)
        self.stack.append(runconfig)
        try:
            yield
        finally:
            self.stack.pop()
