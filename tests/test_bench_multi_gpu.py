"""`bench.py --gpus N` is what the driver runs on the 8-GPU node; this build box has one GPU, so its N > 1 branch -- the
self-launch under torch.distributed.run, shard construction per rank, the exact protocol's pipeline calibration, the fast mode
and the replica mode timed beside it, the MAX-over-ranks timing and rank 0's single JSON line -- is exercised here with
`--single-device-smoke` (every rank on cuda:0, gloo group, host-staged gathers: timings meaningless, control flow real)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_bench_two_ranks_on_one_device(mode):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device-smoke", "--shard-mode", mode,
           "--passages", "40000", "--doclen", "64", "--batch", "128", "--sub-batch", "64", "--query-batches", "2",
           "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-built-index"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    assert rec["metric"] == "queries/sec" and rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["higher_is_better"] is True and rec["scaling"] == "strong"
    assert rec["recall_at_5"] >= 0.95, rec["recall_at_5"]      # planted queries: the sharded search finds its targets
    # the three curves the first real multi-GPU run will yield: the sharded mode asked for, (exact only) the fast mode beside it,
    # and the replica mode (whole index per rank, queries split, no collective)
    rep = rec["shard_mode_replica"]
    assert "failed" not in rep and rep["queries_per_sec"] > 0 and rep["recall_at_5"] >= 0.95 and rep["queries_per_rank"] == 64
    if mode == "exact":
        fast = rec["shard_mode_fast"]
        assert "failed" not in fast and fast["queries_per_sec"] > 0 and fast["recall_at_5"] >= 0.95
        assert rec["shard_pipeline"]["chosen_depth"] in (1, 2) and set(rec["exchange_ms"]) >= {"stage1_keys"} or rec["exchange_ms"]
    assert "2 GPUs" in rec["config"]["parallelism"]
