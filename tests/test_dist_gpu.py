"""N>1 on the GPU path: two processes (gloo, sharing cuda:0) through Engine + GradReducer, and bench.py's N=2 code path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, port, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    env = dict(os.environ, DC_DIST_BACKEND="gloo")
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_reduced_gradients_equal_sum_of_rank_gradients():
    r = _torchrun([os.path.join(ROOT, "tests", "dist_worker.py")], 29621)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_WORKER rel_err" in r.stdout


def test_bench_two_ranks_prints_one_json_line():
    import json
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--local_batch_size", "2",
                   "--height", "128", "--width", "192"], 29622)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 4 and out["value"] > 0
