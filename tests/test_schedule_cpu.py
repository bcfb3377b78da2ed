"""The PRODUCT's LR schedule classes (mlperf-deepcam_amd/nn.py) against the golden sequences captured from the reference's
get_lr_schedule (utils/parsing_helpers.py:27-37) driven the way train_hdf5_ddp.py:369-371 drives it: read, then step."""
import json
import os

import numpy as np
import pytest

from mlperf_deepcam_amd import nn as dnn


class _Opt:
    """What a scheduler needs from an optimizer."""

    def __init__(self, lr):
        self.param_groups = [{"lr": lr}]


def test_multistep_schedule_matches_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "lr_schedule.json")))
    for start in (0, 4):
        opt = _Opt(1e-3)
        sched = dnn.get_lr_schedule(1e-3, g["arg"], opt, last_step=start)
        assert opt.param_groups[0]["initial_lr"] == 1e-3                       # parsing_helpers.py:29
        seq = []
        for _ in range(8):
            seq.append(sched.get_last_lr()[0])                                 # read BEFORE step (train_hdf5_ddp.py:370-371)
            sched.step()
            assert opt.param_groups[0]["lr"] == sched.get_last_lr()[0]        # the optimizer's lr is what the kernels read
        np.testing.assert_allclose(seq, g[f"start{start}"], rtol=1e-12)
    assert g["bad_type_raises"]
    with pytest.raises(ValueError) as e:
        dnn.get_lr_schedule(1e-3, {"type": "cosine"}, _Opt(1e-3))
    assert str(e.value) == g["bad_type_message"]


def test_two_decays_on_one_step_and_resume_past_all_milestones():
    opt = _Opt(1.0)
    sched = dnn.MultiStepSchedule(opt, [2, 2, 5], 0.5, last_epoch=0)          # a repeated milestone decays twice (Counter semantics)
    seq = []
    for _ in range(6):
        seq.append(sched.get_last_lr()[0])
        sched.step()
    assert seq == [1.0, 0.25, 0.25, 0.25, 0.125, 0.125]
    opt = _Opt(1e-5)                                                          # resumed: the checkpoint's optimizer carries the decayed lr
    sched = dnn.MultiStepSchedule(opt, [3, 6], 0.1, last_epoch=10)
    assert [sched.get_last_lr()[0] for _ in range(3) if sched.step() is None] == [1e-5] * 3


def test_gradual_warmup_parity_unpinned_definition():
    """PARITY UNPINNED (pytorch-gradual-warmup-lr is not vendored): the definition the driver documents.
    multiplier > 1: base -> base*multiplier linearly over total_epoch steps; multiplier == 1 (the reference's default
    --lr_warmup_factor): 0 -> base; afterwards the wrapped schedule takes over."""
    opt = _Opt(1e-3)
    after = dnn.get_lr_schedule(1e-3, {"type": "multistep", "milestones": "100", "decay_rate": "0.1"}, opt, last_step=0)
    warm = dnn.GradualWarmupScheduler(opt, multiplier=4.0, total_epoch=4, after_scheduler=after)
    seq = []
    for _ in range(7):
        seq.append(warm.get_last_lr()[0])
        warm.step()
    np.testing.assert_allclose(seq[:5], [1e-3, 1.75e-3, 2.5e-3, 3.25e-3, 4e-3], rtol=1e-12)
    np.testing.assert_allclose(seq[5:], [4e-3, 4e-3], rtol=1e-12)

    opt = _Opt(1e-3)
    after = dnn.get_lr_schedule(1e-3, {"type": "multistep", "milestones": "100", "decay_rate": "0.1"}, opt, last_step=0)
    warm = dnn.GradualWarmupScheduler(opt, multiplier=1.0, total_epoch=4, after_scheduler=after)
    seq = []
    for _ in range(7):
        seq.append(warm.get_last_lr()[0])
        warm.step()
    np.testing.assert_allclose(seq, [0.0, 0.25e-3, 0.5e-3, 0.75e-3, 1e-3, 1e-3, 1e-3], rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        dnn.GradualWarmupScheduler(_Opt(1e-3), multiplier=0.5, total_epoch=4)
