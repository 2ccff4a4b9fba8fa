import os
import sys


def effective_cpus():
    """CPUs this process may actually use: min(affinity, cgroup CPU quota).  The GPU boxes report 256 hardware threads
    under a 16-CPU cgroup quota; libraries that size their pools from the former (torch intra-op, OpenMP, the reference
    ops' at::get_num_threads() pthreads) then spend their time throttled -- round 3's GPU run was lost to exactly that."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


_THREADS = str(max(1, min(8, effective_cpus())))
for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):   # before numpy / torch create their pools
    os.environ.setdefault(_v, _THREADS)

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
INDEX_FIXTURES = ["idx_nb1", "idx_nb2", "idx_nb4", "idx_nb8"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "limit(seconds): per-test watchdog limit (default FLMR_TEST_LIMIT_S, 150 s)")


# ---- collection order: cheapest and most local evidence first ------------------------------------------------------------
# Under `-x -q` one slow or stuck test costs everything behind it, so the op-level / golden / fuzz parity tests run before
# the BASELINE-size corpora, and those run smallest first.
_FILE_RANK = {"test_oracle_golden.py": 0, "test_host_logic.py": 1, "test_split_arithmetic.py": 2, "test_dropin.py": 3,
              "test_native_harness.py": 4, "test_hip_parity.py": 5, "test_hip_regressions.py": 6,
              "test_distributed_gloo.py": 7, "test_baseline_shapes.py": 9}
_SIZE_RANK = ["test_fuzz_sharded", "test_cfg2", "test_cfg3", "test_cfg4_1m", "test_cfg4_sharded", "test_cfg5_one_shard",
              "test_cfg5_6m"]


def pytest_collection_modifyitems(config, items):
    def key(ix_item):
        ix, item = ix_item
        f = os.path.basename(str(item.fspath))
        size = next((i for i, p in enumerate(_SIZE_RANK) if item.name.startswith(p)), 0) if f == "test_baseline_shapes.py" else 0
        return (_FILE_RANK.get(f, 8), size, ix)
    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


# ---- per-test watchdog ---------------------------------------------------------------------------------------------------
# Two stages.  SIGALRM at the limit raises in the main thread, so an ordinary slow test FAILS with its name (and `-x` stops
# there).  A main thread stuck inside a C call (hipDeviceSynchronize on a hung kernel, a join on the reference's pthreads)
# never returns to the interpreter to see the signal: faulthandler's own watchdog thread then dumps every thread's stack
# and ends the process 45 s later, so the log names the test and the frame instead of a driver-side kill at 20 minutes.
_DEFAULT_LIMIT = float(os.environ.get("FLMR_TEST_LIMIT_S", "150"))
# progress + watchdog dumps go to a file as well (pytest's fd capture would swallow a dump written to stderr when the
# process is ended from the watchdog thread); gpurun_out/ is pulled back from the GPU box
_LOGDIR = os.path.join(ROOT, "gpurun_out")
try:
    os.makedirs(_LOGDIR, exist_ok=True)
    _PROGRESS = os.environ.get("FLMR_TEST_PROGRESS", os.path.join(_LOGDIR, "pytest_progress.log"))
    _DUMP = open(os.path.join(_LOGDIR, "pytest_watchdog.log"), "a")
    import atexit
    atexit.register(_DUMP.close)
except OSError:
    _PROGRESS, _DUMP = "", None


class TestTimeout(Exception):
    pass


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    import faulthandler
    import signal
    import time
    m = item.get_closest_marker("limit")
    limit = float(m.args[0]) if m else _DEFAULT_LIMIT
    t0 = time.time()
    if _PROGRESS:
        with open(_PROGRESS, "a") as f:
            f.write(f"START {item.nodeid}\n")

    def on_alarm(signum, frame):
        raise TestTimeout(f"{item.nodeid} exceeded its {limit:.0f} s limit")

    armed = hasattr(signal, "SIGALRM") and limit > 0
    if armed:
        sys.stderr.flush()
        if _DUMP is not None:
            _DUMP.write(f"ARMED {item.nodeid} limit {limit:.0f}s (+45 s hard)\n")
            _DUMP.flush()
            faulthandler.dump_traceback_later(limit + 45, exit=True, file=_DUMP)
        else:
            faulthandler.dump_traceback_later(limit + 45, exit=True)
        old = signal.signal(signal.SIGALRM, on_alarm)
        signal.setitimer(signal.ITIMER_REAL, limit)
    try:
        yield
    finally:
        if armed:
            signal.setitimer(signal.ITIMER_REAL, 0)
            signal.signal(signal.SIGALRM, old)
            faulthandler.cancel_dump_traceback_later()
        if _PROGRESS:
            with open(_PROGRESS, "a") as f:
                f.write(f"END   {item.nodeid} {time.time() - t0:.1f}s\n")


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
