"""Reference-shaped module tree (`from colbert import Searcher`, `colbert.search.index_storage.IndexScorer`, ...).
Put this package's parent directory on sys.path AHEAD of third_party/ColBERT to make the RA-VQA executors pick up the
MI355X path without source changes.  Every module here only re-exports `ravqa_amd` objects."""
from ravqa_amd.searcher import Searcher  # noqa: F401
