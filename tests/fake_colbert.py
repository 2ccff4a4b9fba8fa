"""A stand-in `colbert` package tree for machines where the reference checkout does not exist (the GPU box): the same
module / attribute names `ravqa_amd.install()` patches, with trivial bodies written for this test suite (no reference
source).  Its infra / data classes re-export this build's host-side mirrors so Run().context / ColBERTConfig behave."""
import os
import sys
import textwrap

_FILES = {
    "colbert/__init__.py": """
        from .indexer import Indexer
        from .searcher import Searcher
    """,
    "colbert/indexer.py": """
        class Indexer:
            marker = "reference-indexer"
    """,
    "colbert/searcher.py": """
        from colbert.search.index_storage import IndexScorer
        class Searcher:
            marker = "reference-searcher"
            def __init__(self, *a, **kw):
                raise RuntimeError("the stand-in reference Searcher must never be constructed once install() ran")
    """,
    "colbert/search/__init__.py": "",
    "colbert/search/index_storage.py": """
        class IndexScorer:
            marker = "reference-index-scorer"
    """,
    "colbert/search/strided_tensor.py": """
        class StridedTensor:
            pass
    """,
    "colbert/modeling/__init__.py": "",
    "colbert/modeling/colbert.py": """
        import torch
        class ColBERT(torch.nn.Module):
            pass
    """,
    "colbert/infra/__init__.py": """
        from ravqa_amd.config import ColBERTConfig, Run, RunConfig
    """,
    "colbert/infra/provenance.py": """
        from ravqa_amd.data import Provenance
    """,
    "colbert/data/__init__.py": """
        from ravqa_amd.data import Collection, Queries, Ranking
    """,
}


def make(root):
    """Write the stand-in package under `root` and put `root` first on sys.path.  Returns a cleanup callable."""
    for rel, body in _FILES.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(textwrap.dedent(body))
    sys.path.insert(0, root)

    def cleanup():
        if root in sys.path:
            sys.path.remove(root)
        for name in [m for m in sys.modules if m == "colbert" or m.startswith("colbert.")]:
            del sys.modules[name]
    return cleanup
