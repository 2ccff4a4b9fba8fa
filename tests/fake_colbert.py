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
        from colbert.modeling.colbert import colbert_score
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
        def colbert_score(Q, D_padded, D_mask, config=None, use_gpu=False):
            # stand-in with the call contract of the reference's function (padded late interaction, -9999 padding)
            s = D_padded @ Q.to(D_padded.dtype).permute(0, 2, 1)
            pad = ~D_mask.view(s.size(0), s.size(1)).bool()
            return s.masked_fill(pad.unsqueeze(-1), -9999).max(1).values.sum(-1)
        colbert_score.marker = "reference-colbert-score"
        class ColBERT(torch.nn.Module):
            use_gpu = False
            colbert_config = None
            def score(self, Q, D_padded, D_mask):
                return colbert_score(Q, D_padded, D_mask, config=self.colbert_config, use_gpu=self.use_gpu)
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
