"""MI355X-native late-interaction retrieval path for RA-VQA / FLMR / PreFLMR.

Import name: `ravqa_amd` (see ravqa_amd.py at the repo root; this directory's name contains hyphens).
`ravqa_amd.install()` patches the reference's `colbert` package in place (Searcher / IndexScorer -> this build); see
dropin.py and INTEGRATION.md.
"""
from . import _native
from ._native import FlmrNativeError, build_native
from .config import ColBERTConfig, Run, RunConfig
from .data import Collection, Provenance, Queries, Ranking
from .dropin import install, installed, uninstall
from .index import DeviceIndex, IndexArrays, codec_tables, load_index_arrays

__all__ = ["ColBERTConfig", "RunConfig", "Run", "Queries", "Ranking", "Collection", "Provenance", "IndexArrays",
           "DeviceIndex", "load_index_arrays", "codec_tables", "build_native", "FlmrNativeError", "Searcher",
           "IndexScorer", "install", "uninstall", "installed", "Indexer", "FLMRModelForRetrieval"]


def __getattr__(name):  # torch-dependent pieces are imported lazily
    if name == "Searcher":
        from .searcher import Searcher
        return Searcher
    if name == "IndexScorer":
        from .scorer import IndexScorer
        return IndexScorer
    if name == "Indexer":
        from .indexer import Indexer
        return Indexer
    if name == "FLMRModelForRetrieval":
        from .flmr import FLMRModelForRetrieval
        return FLMRModelForRetrieval
    raise AttributeError(name)
