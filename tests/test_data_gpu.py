"""Input pipeline on the GPU: staged batches equal the reference's host-side normalisation; training through it equals the NCHW path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mlperf_deepcam_amd import data as ddata  # noqa: E402
from mlperf_deepcam_amd import nn as dnn  # noqa: E402

DEV = torch.device("cuda", 0)


def _reference_batch(ds, first, B):
    """What CamDataset.__getitem__ + DataLoader would hand over: NCHW fp32, normalised on the host (cam_hdf5_dataset.py:122-129)."""
    xs, ys = [], []
    for j in range(B):
        d, l = np.empty(ds.data_shape, np.float32), np.empty(ds.label_shape, np.int64)
        ds.read_into(first + j, d, l)
        data = np.transpose(d[..., ds.channels], (2, 0, 1))
        xs.append(ds.data_scale.reshape(-1, 1, 1) * (data - ds.data_shift.reshape(-1, 1, 1)))
        ys.append(l)
    return torch.from_numpy(np.stack(xs)), torch.from_numpy(np.stack(ys))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("channels", [list(range(16)), [0, 1, 2, 5, 8, 9, 12, 15]], ids=["all16", "subset8"])
def test_pipeline_batches_match_host_normalisation(dtype, channels):
    ds = ddata.SyntheticHWC(7, 16, 24, channels=channels)
    pipe = ddata.InputPipeline(ds, 2, dtype=dtype, depth=2)
    assert len(pipe) == 3                                                  # drop_last
    seen = 0
    for b, (x, y, names) in enumerate(pipe):
        xr, yr = _reference_batch(ds, 2 * b, 2)
        assert x.shape == (2, 16, 24, len(channels)) and x.dtype == dtype
        got = x.float().cpu().permute(0, 3, 1, 2)
        assert torch.equal(got, xr.to(dtype).float())                     # same fp32 arithmetic, one rounding to the activation dtype
        assert torch.equal(y.cpu(), yr)
        assert names == ds.files[2 * b:2 * b + 2]
        seen += 1
    assert seen == 3


def test_training_through_the_pipeline_equals_the_nchw_path():
    H, W, B = 64, 96, 2
    ds = ddata.SyntheticHWC(4, H, W)
    cw = dnn.class_weights()
    losses = []
    for use_pipe in (True, False):
        net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.float32, seed=333)
        net.materialize(B, H, W)
        opt = dnn.make_optimizer("AdamW", net, 1e-3, 1e-8, 1e-2)
        step = dnn.TrainStep(net, opt, cw, B, H, W)
        ls = []
        if use_pipe:
            for x, y, _ in ddata.InputPipeline(ds, B, dtype=torch.float32, depth=2):
                step(x, y)
                ls.append(step.loss())
        else:
            for b in range(2):
                x, y = _reference_batch(ds, b * B, B)
                step(x.to(DEV), y.to(DEV))
                ls.append(step.loss())
        torch.cuda.synchronize()
        losses.append((ls, net.engine.params.clone()))
    # (the loss is a sum of per-block fp64 atomics: the last bit depends on arrival order; the weights must be bit-identical)
    assert losses[0][0] == pytest.approx(losses[1][0], rel=1e-12)
    assert torch.equal(losses[0][1], losses[1][1])


def test_pipeline_over_real_hdf5_files_matches_host_normalisation(tmp_path):
    """CAM5-shaped HDF5 files written and read through the HDF5 C library (no h5py), statistics from the dataset-preparation
    tool, then the whole device path: staged batches equal what the reference's __getitem__ computes on the host."""
    from mlperf_deepcam_amd import h5lite, prep
    if not h5lite.available():
        pytest.skip("no HDF5 C library in this image")
    Hh, Ww, root = 16, 24, str(tmp_path / "train")
    import os
    os.makedirs(root)
    rs = np.random.RandomState(5)
    for i in range(5):
        with h5lite.File(os.path.join(root, f"data-{i:02d}.h5"), "w") as f:
            f.write("climate/data", (rs.rand(Hh, Ww, 16) * np.linspace(1, 300, 16) - 40.0).astype(np.float32))
            f.write("climate/labels_0", rs.randint(0, 3, (Hh, Ww)).astype(np.int64))
    stats = prep.summarize(str(tmp_path), workers=2)
    channels = [0, 1, 2, 10]
    ds = ddata.CamDataset(root, stats, channels, shuffle=True)
    pipe = ddata.InputPipeline(ds, 2, dtype=torch.float32, depth=2, workers=2)
    nb = 0
    for b, (x, y, names) in enumerate(pipe):
        xr, yr = _reference_batch(ds, 2 * b, 2)
        assert torch.equal(x.cpu().permute(0, 3, 1, 2), xr) and torch.equal(y.cpu(), yr)
        assert names == ds.files[2 * b:2 * b + 2]
        nb += 1
    assert nb == 2


def test_abandoned_iteration_does_not_corrupt_the_next_one():
    """train.py breaks out of the validation loader at --max_validation_steps: the next iteration must hand out every batch
    once, with the right contents (a slot of the abandoned iteration used to be queued twice)."""
    ds = ddata.SyntheticHWC(12, 16, 24)
    pipe = ddata.InputPipeline(ds, 2, dtype=torch.float32, depth=2, workers=2)
    for rounds in range(3):
        for b, (x, y, names) in enumerate(pipe):
            if b == 1:
                break                                                     # abandon with a batch staged and one handed out
        seen = []
        for b, (x, y, names) in enumerate(pipe):
            xr, yr = _reference_batch(ds, 2 * b, 2)
            assert torch.equal(x.cpu().permute(0, 3, 1, 2), xr) and torch.equal(y.cpu(), yr), (rounds, b)
            seen += names
        assert seen == ds.files[:12]


@pytest.mark.parametrize("channels", [[0, 1, 2], [0, 1, 2, 3, 4, 5, 6, 10, 12, 15]], ids=["c3", "c10"])
def test_channel_subsets_the_mfma_stem_does_not_take(channels):
    """--channels subsets are legal in the reference (train_hdf5_ddp.py:561): counts that are not a multiple of 8 run the direct
    stem kernel; batches from the pipeline (NHWC) and the reference's NCHW batches give the same step."""
    H, W, B = 32, 48, 2
    ds = ddata.SyntheticHWC(4, H, W, channels=channels)
    cw = dnn.class_weights()
    res = []
    for use_pipe in (True, False):
        net = dnn.DeepLabv3_plus(len(channels), 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
        net.materialize(B, H, W)
        assert net.engine.x0 is None
        opt = dnn.make_optimizer("Adam", net, 1e-3, 1e-8, 1e-6)
        step = dnn.TrainStep(net, opt, cw, B, H, W)
        if use_pipe:
            for b, (x, y, _) in enumerate(ddata.InputPipeline(ds, B, dtype=torch.bfloat16, depth=2, layout="nchw")):
                assert x.shape == (B, len(channels), H, W) and x.dtype == torch.float32
                assert torch.equal(x.cpu(), _reference_batch(ds, b * B, B)[0])
                step(x, y)
        else:
            for b in range(2):
                x, y = _reference_batch(ds, b * B, B)
                step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        assert np.isfinite(step.loss())
        res.append(net.engine.params.clone())
    assert torch.equal(res[0], res[1])
