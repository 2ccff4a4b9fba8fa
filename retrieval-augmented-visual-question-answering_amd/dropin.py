"""`install()` -- Level-1 drop-in: patch the REFERENCE's `colbert` package in place so that the RA-VQA executors
(`src/executors/FLMR_executor.py:46-54,99,774-798`, `FLMR_vision_pretraining_executor.py:177-185`,
`src/models/rag/rag_model_blip.py:30-34,301,397`, `src/models/retriever/FLMR.py:7`) pick up the MI355X search path
without source changes.

Nothing is shadowed: the reference package stays the one that is imported, and everything this build does not
replace -- `Indexer`, `Trainer`, `Checkpoint`, `colbert.modeling.*` (the `ColBERT` nn.Module the FLMR models
subclass), the tokenizers, `colbert.infra.*` (`Run`, `RunConfig`, `ColBERTConfig`), `colbert.data.*` -- is left
untouched.  What is rebound:

    colbert.Searcher, colbert.searcher.Searcher          -> a `ravqa_amd.searcher.Searcher` subclass whose boundary
                                                            types are the reference's own classes
    colbert.search.index_storage.IndexScorer,
    colbert.searcher.IndexScorer                         -> `ravqa_amd.scorer.IndexScorer`

`install(level="ops")` instead keeps the reference's Python and swaps only the four pybind extensions for the torch
front-ends of the C ABI (same signatures; INTEGRATION.md level 2).  `uninstall()` restores every binding.

`scoring=True` (default, either level) also rebinds the scoring head:

    colbert.modeling.colbert.colbert_score               -> a dispatcher that runs the HIP padded MaxSim
    (and the name in every module that imported it)         (`flmr_colbert_score_padded`) for forward-only calls and
                                                            the reference's own torch expression when autograd needs
                                                            the result

so `ColBERT.score` (TPC/modeling/colbert.py:217-224) -- what the FLMR model classes inherit and what the exhaustive
search (`src/executors/FLMR_executor.py:833`) and the RAG re-scoring (`src/models/rag/rag_model_blip.py:435`) call --
reaches the MI355X kernel in a drop-in run, while training (in-batch negatives under autograd) is untouched.
"""
import importlib
import warnings
import sys

_saved = []          # [(object, attribute name, previous value or _MISSING)]
_installed = None    # the installed Searcher subclass (level "searcher") or True (level "ops")
_MISSING = object()


def _bind(obj, name, value):
    _saved.append((obj, name, obj.__dict__.get(name, _MISSING) if hasattr(obj, "__dict__") else getattr(obj, name, _MISSING)))
    setattr(obj, name, value)


def installed():
    return _installed


def uninstall():
    global _installed
    while _saved:
        obj, name, prev = _saved.pop()
        if prev is _MISSING:
            try:
                delattr(obj, name)
            except AttributeError:
                pass
        else:
            setattr(obj, name, prev)
    _installed = None


def _import_reference(package):
    try:
        return importlib.import_module(package)
    except ImportError as e:
        raise ImportError(
            f"ravqa_amd.install(): the reference package `{package}` is not importable ({e}).  Put "
            "third_party/ColBERT of the RA-VQA checkout on sys.path first -- install() patches that package in place, "
            "it does not replace it.") from e


def make_colbert_score_dispatch(reference_colbert_score):
    """The function bound over the reference's `colbert_score` (TPC/modeling/colbert.py:268-286).

    HIP path (flmr_colbert_score_padded: fp16-split MFMA, -9999 padding, no clamp, fp32 accumulation) when the call is
    forward-only -- grad mode off, or no input requires grad -- and the shape is the kernel's ('colbert' interaction,
    or 'flipr' interaction, 3-D Q / D with Q.size(0) in {1, B}, floating inputs).  Everything else -- autograd (training), odd
    ranks -- goes to the reference's own expression, unchanged.

    This function replaces a GLOBAL of the caller's package, so it must not break flows that never had a GPU: when NO HIP
    device is visible in the process (a CPU-only validation run of the reference) the call is not intercepted at all -- it
    goes to the reference's expression, with one RuntimeWarning.  With a device visible there is no host fallback: a missing
    or broken libflmr_hip.so raises FlmrNativeError.  (The search path -- Searcher / IndexScorer -- never passes through.)

    Arithmetic: as in the reference Q is first rounded to D_padded's dtype (colbert.py:280: `Q.to(dtype=D_padded.dtype)`);
    the products of those values are then exact and accumulated in fp32, and only the final per-passage score is rounded to
    D_padded's dtype -- for half / bf16 inputs the reference rounds the [B, Ld, Nq] score matrix and the column sums as well,
    so scores can differ from it in the last fp16 / bf16 digit (fp32 inputs: within 1e-4, tests).  The result lives where the
    reference would have put it (the inputs' device; cuda when use_gpu)."""
    import torch
    state = {"warned": False}

    def reference(Q, D_padded, D_mask, config, use_gpu):
        if config is None:
            return reference_colbert_score(Q, D_padded, D_mask, use_gpu=use_gpu)
        return reference_colbert_score(Q, D_padded, D_mask, config=config, use_gpu=use_gpu)

    def colbert_score(Q, D_padded, D_mask, config=None, use_gpu=False):
        from . import _native, ops
        interaction = getattr(config, "interaction", "colbert") if config is not None else "colbert"
        forward_only = not (torch.is_grad_enabled() and (Q.requires_grad or D_padded.requires_grad))
        shape_ok = (torch.is_tensor(Q) and torch.is_tensor(D_padded) and Q.dim() == 3 and D_padded.dim() == 3
                    and Q.size(0) in (1, D_padded.size(0)) and Q.size(-1) == D_padded.size(-1)
                    and Q.is_floating_point() and D_padded.is_floating_point() and D_padded.size(0) > 0 and D_padded.size(1) > 0)
        if not (forward_only and shape_ok and interaction in ("colbert", "flipr")):
            return reference(Q, D_padded, D_mask, config, use_gpu)
        if not _native.device_visible():
            if not state["warned"]:
                state["warned"] = True
                warnings.warn("ravqa_amd: no HIP device is visible in this process -- colbert_score calls are left to the "
                              "reference's torch expression (the HIP scorer takes them when a device is present)", RuntimeWarning)
            return reference(Q, D_padded, D_mask, config, use_gpu)
        Qr = Q.detach().to(dtype=D_padded.dtype)                                   # colbert.py:280
        if interaction == "flipr":   # colbert.py:246-261: the column maxima from the HIP kernel, their top-k sums in scoring.reduce_colmax
            from . import scoring
            out = scoring.reduce_colmax(ops.colbert_colmax_padded(Qr, D_padded.detach(), D_mask), config)
        else:
            out = ops.colbert_score_padded(Qr, D_padded.detach(), D_mask)          # f32 on the device
        dev = torch.device("cuda") if use_gpu else D_padded.device
        return out.to(device=dev, dtype=D_padded.dtype)

    colbert_score.__doc__ = (reference_colbert_score.__doc__ or "") + "\n    [ravqa_amd: forward-only calls run flmr_colbert_score_padded on the MI355X]"
    colbert_score.__wrapped__ = reference_colbert_score
    colbert_score.__ravqa_amd__ = True
    return colbert_score


def _install_scoring(package):
    """Rebind `colbert_score` in colbert.modeling.colbert (ColBERT.score resolves it there at call time) and in every
    module that imported the name before install() (index_storage.py:12 does)."""
    mc = _import_reference(package + ".modeling.colbert")
    reference_fn = getattr(mc, "colbert_score", None)
    if reference_fn is None or getattr(reference_fn, "__ravqa_amd__", False):
        return
    dispatch = make_colbert_score_dispatch(reference_fn)
    _bind(mc, "colbert_score", dispatch)
    for mod in list(sys.modules.values()):
        d = getattr(mod, "__dict__", None)
        if d is None or mod is mc:
            continue
        if d.get("colbert_score") is reference_fn:
            _bind(mod, "colbert_score", dispatch)


def install(level="searcher", package="colbert", require_device=False, scoring=True):
    """Patch the reference's `colbert` package (see module docstring).  Returns the installed Searcher class
    (level "searcher") or the list of patched op names (level "ops").  Idempotent.  `require_device=True` also
    checks that libflmr_hip.so loads and a HIP device is visible (the search path has no CPU fallback).
    `scoring=True` also routes forward-only `colbert_score` / `ColBERT.score` calls to the HIP padded scorer."""
    global _installed
    if _installed is not None:
        return _installed
    if level not in ("searcher", "ops"):
        raise ValueError(f"level must be 'searcher' or 'ops', got {level!r}")
    from . import _native
    if require_device:
        _native.load(require_device=True)
    ref = _import_reference(package)
    if getattr(ref, "__ravqa_amd__", False):
        raise ImportError(f"`{package}` resolves to a ravqa_amd shim, not to the reference package")
    if scoring:
        _install_scoring(package)

    if level == "ops":
        from . import ops
        ixs = _import_reference(package + ".search.index_storage")
        st = _import_reference(package + ".search.strided_tensor")
        mc = _import_reference(package + ".modeling.colbert")
        # the class attributes the reference installs from its JIT-built extensions
        # (index_storage.py:29-60, strided_tensor.py:19-37, modeling/colbert.py:44-62)
        _bind(ixs.IndexScorer, "filter_pids", staticmethod(ops.filter_pids))
        _bind(ixs.IndexScorer, "decompress_residuals", staticmethod(ops.decompress_residuals))
        _bind(ixs.IndexScorer, "loaded_extensions", True)
        _bind(st.StridedTensor, "segmented_lookup", staticmethod(ops.segmented_lookup))
        _bind(st.StridedTensor, "loaded_extensions", True)
        _bind(mc.ColBERT, "segmented_maxsim", staticmethod(ops.segmented_maxsim))
        _bind(mc.ColBERT, "loaded_extensions", True)
        _installed = ["filter_pids", "decompress_residuals", "segmented_lookup", "segmented_maxsim"]
        return _installed

    from .scorer import IndexScorer
    from .searcher import Searcher
    infra = _import_reference(package + ".infra")
    data = _import_reference(package + ".data")
    prov = _import_reference(package + ".infra.provenance")
    searcher_mod = _import_reference(package + ".searcher")
    ixs = _import_reference(package + ".search.index_storage")
    try:
        checkpoint_cls = _import_reference(package + ".modeling.checkpoint").Checkpoint
    except (ImportError, AttributeError):   # text encoding is optional on the search path
        checkpoint_cls = None
    reference_searcher = searcher_mod.Searcher
    from .data import lazy_flat_ranking

    # (Ranking: the reference's class with its eager flat list deferred -- a subclass, so isinstance / save / tolist hold)
    bound = {"ColBERTConfig": infra.ColBERTConfig, "Run": infra.Run, "Collection": data.Collection,
             "Queries": data.Queries, "Ranking": lazy_flat_ranking(data.Ranking), "Provenance": prov.Provenance,
             "IndexScorer": IndexScorer, "Checkpoint": checkpoint_cls, "reference_class": reference_searcher,
             "__doc__": "colbert.Searcher running on libflmr_hip.so (installed by ravqa_amd.install())",
             "__module__": package + ".searcher"}
    InstalledSearcher = type("Searcher", (Searcher,), bound)

    _bind(ref, "Searcher", InstalledSearcher)
    _bind(searcher_mod, "Searcher", InstalledSearcher)
    _bind(searcher_mod, "IndexScorer", IndexScorer)
    _bind(ixs, "IndexScorer", IndexScorer)
    # modules that did `from colbert import Searcher` BEFORE install() hold the old class: rebind those too
    for mod in list(sys.modules.values()):
        d = getattr(mod, "__dict__", None)
        if d is None or mod in (ref, searcher_mod, ixs):
            continue
        if d.get("Searcher") is reference_searcher:
            _bind(mod, "Searcher", InstalledSearcher)
    _installed = InstalledSearcher
    return InstalledSearcher
