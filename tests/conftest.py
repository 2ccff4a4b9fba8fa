import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
INDEX_FIXTURES = ["idx_nb1", "idx_nb2", "idx_nb4", "idx_nb8"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return _cache[name]


def rank_records(z):
    """Names of the rank records in an index fixture: rank0..rank{n-1} + rank_rz."""
    return [f"rank{i}" for i in range(int(z["meta.n_rank"]))] + ["rank_rz"]


@pytest.fixture(params=INDEX_FIXTURES)
def golden_index(request):
    return request.param, load_golden(request.param)


@pytest.fixture
def golden_ops():
    return load_golden("ops")


def tie_aware_equal(ref_pids, ref_scores, got_pids, got_scores, gap=1e-5, tol=1e-4):
    """Ranked-list comparison: ids must match position by position except inside runs of
    reference scores closer than `gap` (a different-but-valid fp32 summation order may swap those);
    scores must agree within `tol` after aligning by pid."""
    ref_pids, got_pids = list(map(int, ref_pids)), list(map(int, got_pids))
    assert len(ref_pids) == len(got_pids), (len(ref_pids), len(got_pids))
    got = dict(zip(got_pids, map(float, got_scores)))
    i, n = 0, len(ref_pids)
    while i < n:
        j = i
        while j + 1 < n and abs(float(ref_scores[j]) - float(ref_scores[j + 1])) <= gap:
            j += 1
        assert sorted(ref_pids[i:j + 1]) == sorted(got_pids[i:j + 1]), (i, j, ref_pids[i:j + 1], got_pids[i:j + 1])
        i = j + 1
    for p, s in zip(ref_pids, ref_scores):
        assert abs(got[p] - float(s)) <= tol, (p, got[p], float(s))
