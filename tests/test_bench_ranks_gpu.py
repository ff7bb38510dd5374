"""bench.py end to end with two ranks that share the one GPU of the test box (gloo
carries the frame reduce, since RCCL refuses two ranks on one device): the sharded
frame must be the single-rank frame, bit for bit, and the JSON line must be complete."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--spp", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--checksum"]


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    sha = re.search(r"frame sha256 ([0-9a-f]{64})", p.stderr).group(1)
    return line, sha


@pytest.mark.gpu
def test_two_ranks_equal_one_rank():
    one, sha1 = run([sys.executable, "bench.py"] + COMMON)
    two, sha2 = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                     "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2",
                     "--backend", "gloo", "--one-device"] + COMMON)
    assert sha1 == sha2
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["ray_segments_per_frame"] == one["config"]["ray_segments_per_frame"]
    for line in (one, two):
        for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in line
        assert line["scaling"] == "strong" and line["vs_baseline"] is None
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.gpu
def test_one_rank_through_rccl():
    """The collectives of the N > 1 path (frame reduce, barriers, the two all-reduces) through the
    nccl backend = RCCL itself, with the one rank a single-GPU box allows: the frame and the
    segment count must be those of the plain single-process run."""
    one, sha1 = run([sys.executable, "bench.py"] + COMMON)
    rccl, sha2 = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                      "--master-addr", "127.0.0.1", "--master-port", "29534", "bench.py", "--gpus", "1",
                      "--single-rank-collective"] + COMMON)
    assert sha1 == sha2
    assert rccl["n_gpus"] == 1
    assert rccl["config"]["ray_segments_per_frame"] == one["config"]["ray_segments_per_frame"]
    # frames enqueued without a host wait, the reduce right behind them on the stream
    fast, sha3 = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                      "--master-addr", "127.0.0.1", "--master-port", "29535", "bench.py", "--gpus", "1",
                      "--single-rank-collective", "--no-wait"] + COMMON)
    assert sha3 == sha1 and fast["config"]["ray_segments_per_frame"] == one["config"]["ray_segments_per_frame"]
