"""Level-1 drop-in (`ravqa_amd.install()`): the reference's `colbert` package patched in place.

* test_install_over_reference_package: build container only (needs /root/reference).  A fresh interpreter imports the
  REFERENCE's colbert (with the scratch shims of tests/golden/_shims), calls install(), executes the exact import
  lines of src/executors/FLMR_executor.py:46-54,99 and src/models/retriever/FLMR.py:7 (read from the checkout at test
  time), builds the executor's ColBERTConfig, and constructs Searcher(index=..., config=<reference ColBERTConfig>)
  inside the reference's Run().context up to the point where the native scorer would be created.
* test_install_mechanics_on_stand_in_package: same bindings against tests/fake_colbert.py (runs anywhere).
The `-m gpu` counterpart that really searches through the patched names is tests/test_hip_parity.py::
test_installed_searcher_through_patched_names.
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, load_golden

REF = os.environ.get("FLMR_REFERENCE_ROOT", "/root/reference")
HAVE_REF = os.path.isdir(os.path.join(REF, "third_party", "ColBERT", "colbert"))

_SCRIPT = r"""
import json, os, sys, re
REF, ROOT, TMP = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, os.path.join(REF, "third_party", "ColBERT"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "_shims"))
sys.path.insert(0, ROOT)
import numpy as np, torch, transformers
if not hasattr(transformers, "AdamW"):
    transformers.AdamW = torch.optim.AdamW
import colbert                                   # the REFERENCE package
ref_searcher, ref_indexer = colbert.Searcher, colbert.Indexer
import colbert.search.index_storage as ixs
ref_scorer = ixs.IndexScorer
import ravqa_amd
Installed = ravqa_amd.install()
assert ravqa_amd.install() is Installed           # idempotent

# ---- the executors' own import lines, verbatim from the checkout ----------------------------------------------
def lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        src = f.read().split("\n")
    return [l.strip() for l in src[lo - 1:hi] if re.match(r"\s*(from colbert|import colbert)", l)]
imports = lines("src/executors/FLMR_executor.py", 46, 54) + lines("src/executors/FLMR_executor.py", 99, 99) + \
          lines("src/models/retriever/FLMR.py", 7, 7) + lines("src/models/rag/rag_model_blip.py", 30, 34)
assert len(imports) >= 8, imports
ns = {}
for l in imports:
    exec(l, ns)
assert ns["Searcher"] is Installed and colbert.searcher.Searcher is Installed
assert ns["Indexer"] is ref_indexer                                   # indexing stays with the reference
assert issubclass(ns["ColBERT"], torch.nn.Module)                      # FLMR.py:7 can still subclass it
assert ns["ColBERTConfig"].__module__.startswith("colbert.infra")      # the reference's own config class
assert ixs.IndexScorer is ravqa_amd.IndexScorer and colbert.searcher.IndexScorer is ravqa_amd.IndexScorer
# FLMR_executor.py:129-134
cc = ns["ColBERTConfig"](bsize=None, use_ib_negatives=True, checkpoint="bert-base-uncased", rank=0)
assert cc.use_ib_negatives is True and cc.checkpoint == "bert-base-uncased"

# ---- FLMR_executor.py:774-794 up to the native scorer ------------------------------------------------------------
Run, RunConfig, ColBERTConfig, Queries = ns["Run"], ns["RunConfig"], ns["ColBERTConfig"], ns["Queries"]
z = dict(np.load(os.path.join(ROOT, "tests", "golden", "idx_nb2.npz")))
index_dir = os.path.join(TMP, "temp_index_0", "indexes", "temp_index.nbits=2")
ravqa_amd.IndexArrays.from_golden(z).save(index_dir)
calls = []
class RecordingScorer:
    def __init__(self, index_path, use_gpu=True, max_batch=256, numerics=None):
        calls.append((index_path, use_gpu))
        self.numerics = numerics
    def search_batch(self, Q, k, ncells, thr, ndocs, nq_cand=32, q_lens=None):
        calls.append(("search_batch", tuple(Q.shape), k, ncells, thr, ndocs, nq_cand))
        n = Q.size(0)
        return (torch.arange(n * k, dtype=torch.int32).view(n, k), torch.ones(n, k), torch.full((n,), k, dtype=torch.int32))
Installed.IndexScorer = RecordingScorer
with Run().context(RunConfig(nranks=1, rank=0, root=TMP, experiment="temp_index_0")):
    config = ColBERTConfig(total_visible_gpus=0)
    searcher = ns["Searcher"](index="temp_index.nbits=2", config=config)
    assert calls[0] == (index_dir, False), calls
    assert isinstance(searcher.config, ColBERTConfig) and searcher.config.nbits == 2 and searcher.config.dim == 128
    assert searcher.config.total_visible_gpus == 0 and searcher.config.root == TMP
    queries = Queries(data={7: "what is this", 9: "and this"})
    ranking = searcher._search_all_Q(queries, torch.zeros(2, 32, 128), k=5)
    assert type(ranking).__module__.startswith("colbert.data")         # the reference's Ranking comes back
    d = ranking.todict()
    assert list(d) == [7, 9] and d[9][0] == (5, 1, 1.0) and len(d[7]) == 5
    assert calls[1] == ("search_batch", (2, 32, 128), 5, 2, 0.45, 1024, 32), calls
    assert searcher.config.ndocs == 1024                                 # the policy is sticky on the config (searcher.py:92-118)
    assert searcher.ranker.numerics == "cpu"                             # total_visible_gpus=0: the CPU-path arithmetic
    # FLMR_executor.py:784 on a single GPU (total_visible_gpus=1): still the pinned CPU-path arithmetic by default; the
    # reference's CUDA-branch arithmetic (SURVEY 8f-4) is opt-in -- numerics="reference" follows the reference's selection
    s1 = ns["Searcher"](index="temp_index.nbits=2", config=ColBERTConfig(total_visible_gpus=1))
    assert calls[-1] == (index_dir, True) and s1.ranker.numerics == "cpu" and s1.numerics == "cpu"
    s2 = ns["Searcher"](index="temp_index.nbits=2", config=ColBERTConfig(total_visible_gpus=1), numerics="reference")
    assert s2.ranker.numerics == "gpu-fp16" and s2.numerics == "gpu-fp16"
    s3 = ns["Searcher"](index="temp_index.nbits=2", config=ColBERTConfig(total_visible_gpus=0), numerics="reference")
    assert s3.numerics == "cpu"

# ---- the scoring head: ColBERT.score -> colbert_score (colbert/modeling/colbert.py:217-224,268-286), what
# FLMR_executor.py:833 (exhaustive search) and rag_model_blip.py:435 (RAG re-score) call ---------------------------
import colbert.modeling.colbert as mc
from oracle import oracle as orc
assert getattr(mc.colbert_score, "__ravqa_amd__", False) and ixs.colbert_score is mc.colbert_score   # index_storage.py:12 imported the name
ref_colbert_score = mc.colbert_score.__wrapped__
zo = dict(np.load(os.path.join(ROOT, "tests", "golden", "ops.npz")))
Qp, Dp, Mp = torch.from_numpy(zo["padded.Q"]), torch.from_numpy(zo["padded.D"]), torch.from_numpy(zo["padded.mask"])
hip_calls = []
def device_stand_in(Q, D, M):          # this container has no GPU: the checker plays the kernel, the DISPATCH is under test
    hip_calls.append((tuple(Q.shape), tuple(D.shape), tuple(M.shape), Q.requires_grad))
    return torch.from_numpy(orc.colbert_score_padded(Q.numpy(), D.numpy(), np.asarray(M).reshape(D.shape[0], D.shape[1])))
import ravqa_amd.ops as rops
real_op, rops.colbert_score_padded = rops.colbert_score_padded, device_stand_in
import ravqa_amd._native as rnat
real_visible, rnat.device_visible = rnat.device_visible, (lambda: True)   # (pretend the device the stand-in plays is there)
class Model:                           # the attributes ColBERT.score reads (colbert.py:217-224)
    colbert_config = ColBERTConfig()
    use_gpu = False
with torch.no_grad():
    got = ns["ColBERT"].score(Model(), Qp, Dp, Mp)                              # the executors' call shape
assert len(hip_calls) == 1 and got.dtype == Dp.dtype and got.device == Dp.device
assert np.max(np.abs(got.numpy() - zo["padded.output"]) / (1.0 + np.abs(zo["padded.output"]))) <= 2e-6
Qa = torch.from_numpy(zo["padded_aligned.Q"])
with torch.inference_mode():
    got_a = mc.colbert_score(Qa, Dp, Mp.unsqueeze(-1), config=ColBERTConfig())   # [B, Ld, 1] masks, aligned queries
assert len(hip_calls) == 2 and np.max(np.abs(got_a.numpy() - zo["padded_aligned.output"]) / (1.0 + np.abs(zo["padded_aligned.output"]))) <= 2e-6
Qg = Qp.clone().requires_grad_(True)
out_g = ns["ColBERT"].score(Model(), Qg, Dp, Mp)                                   # training: the reference's own expression
assert len(hip_calls) == 2 and out_g.requires_grad and torch.equal(out_g.detach(), ref_colbert_score(Qp, Dp, Mp))
out_g.sum().backward()
assert Qg.grad is not None and float(Qg.grad.abs().sum()) > 0
with torch.no_grad():
    mc.colbert_score(Qg, Dp, Mp)                                                   # grad mode off: forward-only again
assert len(hip_calls) == 3 and hip_calls[-1][3] is False
rops.colbert_score_padded = real_op
rnat.device_visible = real_visible
if not torch.cuda.is_available():      # no device in the process: the call is NOT intercepted (a CPU-only run of the
    import warnings                    # reference keeps working), and says so once
    with warnings.catch_warnings(record=True) as w, torch.no_grad():
        warnings.simplefilter("always")
        got = mc.colbert_score(Qp, Dp, Mp)
        mc.colbert_score(Qp, Dp, Mp)
    assert torch.equal(got, ref_colbert_score(Qp, Dp, Mp))
    assert sum("no HIP device" in str(x.message) for x in w) == 1, [str(x.message) for x in w]

ravqa_amd.uninstall()
assert colbert.Searcher is ref_searcher and ixs.IndexScorer is ref_scorer and colbert.searcher.IndexScorer is ref_scorer
assert mc.colbert_score is ref_colbert_score and ixs.colbert_score is ref_colbert_score

# ---- level "ops": the four extension attributes --------------------------------------------------------------
from colbert.search.strided_tensor import StridedTensor
from colbert.modeling.colbert import ColBERT
names = ravqa_amd.install(level="ops")
assert sorted(names) == ["decompress_residuals", "filter_pids", "segmented_lookup", "segmented_maxsim"]
from ravqa_amd import ops
assert ixs.IndexScorer.filter_pids is ops.filter_pids and ixs.IndexScorer.decompress_residuals is ops.decompress_residuals
assert StridedTensor.segmented_lookup is ops.segmented_lookup and ColBERT.segmented_maxsim is ops.segmented_maxsim
assert ixs.IndexScorer.loaded_extensions and StridedTensor.loaded_extensions and ColBERT.loaded_extensions
ravqa_amd.uninstall()
assert "filter_pids" not in ixs.IndexScorer.__dict__ and "loaded_extensions" not in ColBERT.__dict__
print("DROPIN-OK")
"""


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference checkout (build container only)")
def test_install_over_reference_package(tmp_path):
    env = dict(os.environ, TORCH_EXTENSIONS_DIR=str(tmp_path / "ext"), PYTHONPATH="")
    r = subprocess.run([sys.executable, "-c", _SCRIPT, REF, ROOT, str(tmp_path)], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_install_mechanics_on_stand_in_package(tmp_path):
    import fake_colbert
    import ravqa_amd
    cleanup = fake_colbert.make(str(tmp_path / "pkg"))
    try:
        import colbert
        import colbert.search.index_storage as ixs
        ref_searcher = colbert.Searcher
        late = type(sys)("late_importer")            # a module that did `from colbert import Searcher` before install()
        late.Searcher = colbert.Searcher
        sys.modules["late_importer"] = late
        Installed = ravqa_amd.install()
        assert colbert.Searcher is Installed and colbert.searcher.Searcher is Installed and late.Searcher is Installed
        assert ixs.IndexScorer is ravqa_amd.IndexScorer and colbert.Indexer.marker == "reference-indexer"
        assert issubclass(Installed, ravqa_amd.Searcher) and Installed.reference_class is ref_searcher
        assert Installed.Queries is colbert.data.Queries and Installed.Run is colbert.infra.Run
        # the scoring head (scoring=True is the default): module global + the name index_storage imported
        import colbert.modeling.colbert as mc
        import torch
        assert getattr(mc.colbert_score, "__ravqa_amd__", False) and ixs.colbert_score is mc.colbert_score
        assert mc.colbert_score.__wrapped__.marker == "reference-colbert-score"
        Q = torch.randn(1, 5, 128, requires_grad=True)
        D, M = torch.randn(3, 7, 128), torch.ones(3, 7, dtype=torch.bool)
        out = mc.ColBERT().score(Q, D, M)                     # autograd needs it: the reference expression
        assert out.requires_grad and out.shape == (3,)
        seen = []
        from ravqa_amd import _native, ops
        real, real_visible = ops.colbert_score_padded, _native.device_visible
        ops.colbert_score_padded = lambda q, d, m: (seen.append((q.requires_grad, q.dtype, tuple(m.shape))), mc.colbert_score.__wrapped__(q.float(), d, m))[1]
        _native.device_visible = lambda: True      # (the stand-in plays the device)
        try:
            with torch.no_grad():
                out2 = mc.ColBERT().score(Q, D.half(), M.unsqueeze(-1))
        finally:
            ops.colbert_score_padded, _native.device_visible = real, real_visible
        # Q reaches the scorer rounded to D's dtype (colbert.py:280), detached; the result is in D's dtype on D's device
        assert seen == [(False, torch.float16, (3, 7, 1))] and out2.dtype == torch.float16 and out2.device == D.device
        ravqa_amd.uninstall()
        assert colbert.Searcher is ref_searcher and late.Searcher is ref_searcher and ixs.IndexScorer.marker == "reference-index-scorer"
        assert mc.colbert_score.marker == "reference-colbert-score" and ixs.colbert_score is mc.colbert_score
    finally:
        ravqa_amd.uninstall()
        sys.modules.pop("late_importer", None)
        cleanup()


def test_install_needs_the_reference_package(tmp_path):
    import ravqa_amd
    assert "colbert" not in sys.modules
    with pytest.raises(ImportError, match="patches that package in place"):
        ravqa_amd.install(package="colbert_not_there")
