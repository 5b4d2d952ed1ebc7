"""The N > 1 path of bench.py rehearsed on the one-GPU box: `--gpus 2` starts two ranks itself; both sit on cuda:0 and
talk over gloo (AMC3D_DIST_BACKEND; RCCL refuses two ranks on one device).  What it exercises is everything but the
wire: the launcher, scene shards, SyncBatchNorm as the N > 1 default with its statistics all-reduces issued eagerly
between captured graph segments (amcontrast3d_amd/graphs.py), the flat gradient all-reduce between the captured halves,
max-over-ranks timing, and the replica check."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["AMC3D_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--batch", "2", "--points", "4096", "--no-cpu-baseline"] + extra, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_syncbn_default_segmented_graph():
    line = _run([])
    assert line["n_gpus"] == 2 and line["replicas_in_sync"] is True
    assert "syncbn" in line["config"]["parallelism"] and line["config"]["launch"].startswith("hipGraph")
    assert line["config"]["global_batch"] == 4 and line["value"] > 0


def test_two_ranks_without_syncbn():
    line = _run(["--no-sync-bn"])
    assert line["n_gpus"] == 2 and line["replicas_in_sync"] is True and "syncbn" not in line["config"]["parallelism"]
