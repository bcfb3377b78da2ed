"""bench.py's optional baseline leg on the GPU (the default line and the N > 1 paths are covered in test_dist_gpu.py / test_profile_gpu.py)."""
import json, os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_reports_the_torch_operator_baseline_when_asked():
    """--torch_gpu_baseline nchw: after the timed steps the oracle's train step runs on the same GPU through PyTorch-ROCm's own operators (eager,
    bf16 autocast) and is reported beside the line -- `torch_rocm_baseline`, a baseline like `cpu_baseline`; the line itself is unchanged."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--local_batch_size", "2", "--height", "128", "--width", "192",
           "--no_cpu_baseline", "--torch_gpu_baseline", "nchw"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    base = out["torch_rocm_baseline"]
    assert out["value"] > 0 and out["vs_baseline"] is None and "cpu_baseline" not in out
    assert base["value"] is not None and base["value"] > 0, base
    assert base["unit"] == "samples/s" and base["kind"] == "port" and "nchw" in base["sample"]
    assert abs(base["this_repository_over_it"] - out["value"] / base["value"]) < 0.01 * base["this_repository_over_it"] + 0.01


def test_default_line_carries_the_other_single_gpu_configurations():
    """BASELINE.json configs[1] (local batch 2, fp32, Adam), configs[2] (local batch 4, bf16) and the per-rank shape of configs[3] (local batch 2,
    bf16) ride on the default line as `also`: each timed in the
    same process the same way, with its whole-step and encoder-region fraction.  Small images here; the keys are what is asserted."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--height", "128", "--width", "192", "--no_cpu_baseline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["local_batch"] == 8 and out["dtype"] == "bf16" and out["value"] > 0
    also = out["also"]
    assert [(e["config"]["local_batch"], e["dtype"], e["config"]["optimizer"]) for e in also] == [(2, "fp32", "Adam"), (4, "bf16", "LAMB"), (2, "bf16", "LAMB")]
    for e in also:
        assert e["value"] > 0 and e["unit"] == "samples/s" and e["steps"] == 2 and e["warmup"] == 1, e
        assert abs(e["ms_per_step"] * e["value"] - 1e3 * e["config"]["local_batch"]) < 1.0
        assert 0 < e["whole_step_frac"] < 1 and 0 < e["encoder_region"]["frac"] < 1
        assert e["encoder_region"]["fwd_ms"] > 0 and e["encoder_region"]["bwd_ms"] > 0
        assert e["loss_last_step"] == e["loss_last_step"]      # finite
    # --no_also leaves the line as it was
    r = subprocess.run(cmd + ["--no_also"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "also" not in json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
