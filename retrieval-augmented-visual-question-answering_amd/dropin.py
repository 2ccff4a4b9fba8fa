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
"""
import importlib
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


def install(level="searcher", package="colbert", require_device=False):
    """Patch the reference's `colbert` package (see module docstring).  Returns the installed Searcher class
    (level "searcher") or the list of patched op names (level "ops").  Idempotent.  `require_device=True` also
    checks that libflmr_hip.so loads and a HIP device is visible (the search path has no CPU fallback)."""
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

    bound = {"ColBERTConfig": infra.ColBERTConfig, "Run": infra.Run, "Collection": data.Collection,
             "Queries": data.Queries, "Ranking": data.Ranking, "Provenance": prov.Provenance,
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
