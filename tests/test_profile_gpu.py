"""The phase-bracketed profiling driver (reference: profile_hdf5_ddp.py:77-94,196-236): runs outside a profiler too and reports
the forward / backward / optimizer split of the step."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("phase", ["Forward", "Backward", "Optimizer"])
def test_profile_driver_brackets_one_phase(phase):
    r = subprocess.run([sys.executable, "-m", "mlperf_deepcam_amd.profile", "--profile", phase, "--local_batch_size", "2", "--height", "64",
                        "--width", "96", "--num_warmup_steps", "2", "--num_profile_steps", "2", "--optimizer", "LAMB", "--amp_opt_level", "O1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["profile"] == phase and out["profile_steps"] == 2 and out["dtype"] == "bf16"
    assert set(out["ms_per_step"]) == {"Forward", "Backward", "Optimizer"} and all(v > 0 for v in out["ms_per_step"].values())
    assert 0 < out["loss"] < 20


def test_profile_driver_rejects_unknown_phase():
    r = subprocess.run([sys.executable, "-m", "mlperf_deepcam_amd.profile", "--profile", "Everything"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "invalid choice" in r.stderr
