"""HDF5 side of the input pipeline without h5py: the ctypes binding of libhdf5 (h5lite), CamDataset over CAM5-shaped files, and the
dataset-preparation tools (reference: data/cam_hdf5_dataset.py:36-131, src/utils/summarize_data.py, src/utils/split_data.py)."""
import os

import numpy as np
import pytest

from mlperf_deepcam_amd import data as ddata
from mlperf_deepcam_amd import h5lite, prep

pytestmark = pytest.mark.skipif(not h5lite.available(), reason="no HDF5 C library in this image")

H, W, CF = 12, 20, 16


def make_files(root, n, label_dtype=np.int64, seed=0):
    """n CAM5-shaped samples under root/: climate/data float32 [H, W, 16] with per-channel offsets, climate/labels_0 [H, W]."""
    os.makedirs(root, exist_ok=True)
    rs = np.random.RandomState(seed)
    lo = np.linspace(-40.0, 150.0, CF).astype(np.float32)
    arrays = {}
    for i in range(n):
        d = (lo + rs.rand(H, W, CF).astype(np.float32) * np.linspace(1.0, 90.0, CF).astype(np.float32)).astype(np.float32)
        lab = rs.randint(0, 3, size=(H, W)).astype(label_dtype)
        name = f"data-2000-01-{i:02d}-1.h5"
        with h5lite.File(os.path.join(root, name), "w") as f:
            f.write("climate/data", d)
            f.write("climate/labels_0", lab)
        arrays[name] = (d, lab)
    return arrays


def test_h5lite_round_trip_types_and_errors(tmp_path):
    p = str(tmp_path / "t.h5")
    rs = np.random.RandomState(1)
    payload = {"g/f32": rs.rand(5, 7, 3).astype(np.float32), "g/sub/f64": rs.rand(4), "i64": rs.randint(-5, 5, (6, 2)).astype(np.int64),
               "u8": rs.randint(0, 255, (9,)).astype(np.uint8), "scalar": np.int64(42)}
    with h5lite.File(p, "w") as f:
        for k, v in payload.items():
            f.write(k, v)
    with h5lite.File(p) as f:
        for k, v in payload.items():
            assert k in f
            assert f.shape(k) == np.asarray(v).shape and f.dtype(k) == np.asarray(v).dtype
            assert np.array_equal(f.read(k), v)
        assert "g/nope" not in f
        out = np.empty((5, 7, 3), np.float64)
        f.read_direct("g/f32", out)                               # type conversion goes through H5Dread
        assert np.array_equal(out, payload["g/f32"].astype(np.float64))
        with pytest.raises(h5lite.H5Error):
            f.read_direct("g/f32", np.empty((5, 7), np.float32))  # shape mismatch is refused
        with pytest.raises(h5lite.H5Error):
            f.read("missing")
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(tmp_path / "absent.h5"))


@pytest.mark.parametrize("label_dtype", [np.int64, np.uint8])
def test_cam_dataset_reads_what_was_written_and_shards_like_the_reference(tmp_path, label_dtype):
    root = str(tmp_path / "train")
    arrays = make_files(root, 7, label_dtype)
    allv = np.stack([a[0] for a in arrays.values()])
    stats = str(tmp_path / "stats.h5")
    with h5lite.File(stats, "w") as f:
        f.write("climate/minval", allv.min((0, 1, 2)))
        f.write("climate/maxval", allv.max((0, 1, 2)))
    channels = [0, 1, 2, 10]
    seen = []
    for rank in range(2):
        ds = ddata.CamDataset(root, stats, channels, allow_uneven_distribution=False, shuffle=True, comm_size=2, comm_rank=rank)
        assert len(ds) == 3 and ds.global_size == 6 and ds.data_shape == (H, W, CF) and ds.label_shape == (H, W)
        # normalisation constants exactly as cam_hdf5_dataset.py:96-98
        shift = allv.min((0, 1, 2))[channels]
        assert np.array_equal(ds.data_shift, shift) and np.allclose(ds.data_scale, 1.0 / (allv.max((0, 1, 2))[channels] - shift))
        for i in range(len(ds)):
            d, lab = np.empty((H, W, CF), np.float32), np.empty((H, W), np.int64)
            name = ds.read_into(i, d, lab)
            want_d, want_l = arrays[os.path.basename(name)]
            assert np.array_equal(d, want_d) and np.array_equal(lab, want_l.astype(np.int64))
            seen.append(os.path.basename(name))
    expect = sorted(arrays)
    np.random.RandomState(12345).shuffle(expect)
    assert seen == expect[:6]                                      # one fixed permutation, contiguous slices, remainder dropped


def test_summarize_matches_numpy_and_does_not_depend_on_worker_count(tmp_path):
    arrays = make_files(str(tmp_path / "train"), 5)
    out1 = prep.summarize(str(tmp_path), workers=1)
    with h5lite.File(out1) as f:
        s1 = {k: f.read("climate/" + k) for k in ("count", "mean", "sqmean", "minval", "maxval")}
    out4 = prep.summarize(str(tmp_path), out_path=str(tmp_path / "stats4.h5"), workers=4)
    with h5lite.File(out4) as f:
        s4 = {k: f.read("climate/" + k) for k in s1}
    for k in s1:
        assert np.array_equal(s1[k], s4[k]), k
    allv = np.stack([arrays[k][0] for k in sorted(arrays)]).astype(np.float64)
    assert int(s1["count"]) == 5
    assert np.array_equal(s1["minval"], allv.min((0, 1, 2)).astype(np.float32)) and np.array_equal(s1["maxval"], allv.max((0, 1, 2)).astype(np.float32))
    np.testing.assert_allclose(s1["mean"], allv.mean((0, 1, 2)), rtol=2e-6)
    np.testing.assert_allclose(s1["sqmean"], np.square(allv).mean((0, 1, 2)), rtol=2e-6)
    # the loader accepts the file the tool wrote
    ds = ddata.CamDataset(str(tmp_path / "train"), out1, list(range(16)))
    assert np.array_equal(ds.data_shift, s1["minval"])


def test_split_is_the_reference_permutation_and_links_every_file_once(tmp_path):
    src = tmp_path / "all"
    src.mkdir()
    names = [f"data-{i:03d}.h5" for i in range(23)] + ["notes.txt", "stats.h5"]
    for n in names:
        (src / n).write_bytes(b"x")
    train, val, test = prep.split(str(src), str(tmp_path / "out"))
    files = sorted(n for n in names if n.startswith("data") and n.endswith(".h5"))
    np.random.seed(12345)                                          # the reference seeds the global generator and shuffles in place
    np.random.shuffle(files)
    assert (train, val, test) == (files[:18], files[18:20], files[20:])
    for sub, lst in (("train", train), ("validation", val), ("test", test)):
        d = tmp_path / "out" / sub
        assert sorted(os.listdir(d)) == sorted(lst) and all(os.path.islink(d / n) for n in lst)
