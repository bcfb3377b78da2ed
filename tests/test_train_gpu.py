"""The train_hdf5_ddp.py-compatible driver end to end on synthetic data: CLI, MLLOG lines, validation, checkpoint, resume."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=400):
    cmd = [sys.executable, "-m", "mlperf_deepcam_amd.train"] + args
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def _events(stdout):
    ev = []
    for line in stdout.splitlines():
        if line.startswith(":::MLLOG "):
            ev.append(json.loads(line[len(":::MLLOG "):]))
    return ev


def test_driver_trains_validates_saves_and_resumes(tmp_path):
    out = str(tmp_path / "run")
    common = ["--wireup_method", "single", "--run_tag", "t1", "--output_dir", out, "--synthetic_samples", "8", "--local_batch_size", "2",
              "--height", "64", "--width", "96", "--logging_frequency", "1", "--validation_frequency", "2", "--save_frequency", "2",
              "--optimizer", "AdamW", "--weight_decay", "1e-2", "--amp_opt_level", "O1", "--model_prefix", "classifier",
              "--lr_schedule", 'type="multistep",milestones="3 5",decay_rate="0.1"', "--training_visualization_frequency", "0",
              "--validation_visualization_frequency", "0"]
    r = _run(common + ["--max_epochs", "1"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ev = _events(r.stdout)
    keys = [e["key"] for e in ev]
    for k in ("init_start", "seed", "global_batch_size", "opt_name", "train_samples", "eval_samples", "init_stop", "run_start", "epoch_start",
              "learning_rate", "train_accuracy", "train_loss", "eval_start", "eval_accuracy", "eval_loss", "eval_stop", "save_start", "save_stop",
              "epoch_stop", "run_stop"):
        assert k in keys, k
    lrs = [e["value"] for e in ev if e["key"] == "learning_rate"]
    assert lrs == pytest.approx([1e-3, 1e-3, 1e-4, 1e-4])              # milestone 3 takes effect from the 3rd step on
    losses = [e["value"] for e in ev if e["key"] == "train_loss"]
    assert len(losses) == 4 and all(0 < v < 20 for v in losses)
    ck = os.path.join(out, "classifier_step_4.cpt")
    assert os.path.exists(ck) and os.path.exists(os.path.join(out, "classifier_step_2.cpt"))
    c = torch.load(ck, map_location="cpu", weights_only=False)
    assert c["step"] == 4 and c["epoch"] == 0
    assert len(c["model"]) == 532 and all(k.startswith("module.") for k in c["model"])
    assert c["model"]["module.xception_features.bn1.num_batches_tracked"].dtype == torch.int64
    assert int(c["model"]["module.xception_features.bn1.num_batches_tracked"]) == 4
    assert set(c["optimizer"]) == {"state", "param_groups"} and len(c["optimizer"]["state"]) == 301
    assert sorted(c["optimizer"]["state"][0]) == ["exp_avg", "exp_avg_sq", "step"]
    # resume: continues at step 4 with the decayed learning rate restored from the optimizer state
    r2 = _run(common + ["--max_epochs", "2", "--checkpoint", ck, "--max_steps", "6"])
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    ev2 = _events(r2.stdout)
    steps = [e["metadata"]["step_num"] for e in ev2 if e["key"] == "train_loss"]
    assert steps == [5, 6]
    lrs2 = [e["value"] for e in ev2 if e["key"] == "learning_rate"]
    # The checkpointed optimizer already holds 1e-5 (counter 5 was reached by the 4th scheduler.step()); the reference then
    # builds MultiStepLR(last_epoch=4), whose constructor steps to counter 5 again and multiplies once more
    # (parsing_helpers.py:35, train_hdf5_ddp.py:236,246): a resume exactly at a milestone decays twice.  Mirrored, not fixed.
    assert lrs2 == pytest.approx([1e-6, 1e-6])


def test_driver_rejects_unknown_optimizer_and_missing_data(tmp_path):
    r = _run(["--wireup_method", "single", "--run_tag", "t", "--output_dir", str(tmp_path), "--optimizer", "SGD"])
    assert r.returncode != 0 and "invalid choice" in r.stderr
    r = _run(["--wireup_method", "single", "--run_tag", "t", "--output_dir", str(tmp_path), "--local_batch_size", "2",
              "--data_dir_prefix", str(tmp_path / "no_such_data")])
    # no data directory: a clear error (missing directory, or missing HDF5 support), never a silent fallback to synthetic data
    assert r.returncode != 0 and ("No such file or directory" in r.stderr or "h5py" in r.stderr)
