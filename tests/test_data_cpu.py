"""Sharding rule of the input pipeline (host logic) against the reference's init_reader semantics (cam_hdf5_dataset.py:38-65)."""
import numpy as np

from mlperf_deepcam_amd import data as ddata


def _names(n):
    return [f"/d/data-{i:04d}.h5" for i in np.random.RandomState(0).permutation(n)]       # unsorted on purpose


def test_even_sharding_drops_remainder_and_reports_reduced_global_size():
    files = _names(11)
    shards = [ddata.shard_files(files, 4, r, False, False) for r in range(4)]
    assert [len(s[0]) for s in shards] == [2, 2, 2, 2] and all(s[1] == 8 for s in shards)
    flat = [f for s in shards for f in s[0]]
    assert flat == sorted(files)[:8]                       # contiguous slices of the SORTED list


def test_uneven_sharding_gives_remainder_to_last_rank():
    files = _names(11)
    shards = [ddata.shard_files(files, 4, r, True, False) for r in range(4)]
    assert [len(s[0]) for s in shards] == [2, 2, 2, 5] and all(s[1] == 11 for s in shards)
    assert [f for s in shards for f in s[0]] == sorted(files)


def test_shuffle_is_one_fixed_permutation_shared_by_all_ranks():
    files = _names(20)
    a = [ddata.shard_files(files, 2, r, False, True, seed=12345)[0] for r in range(2)]
    b = [ddata.shard_files(list(reversed(files)), 2, r, False, True, seed=12345)[0] for r in range(2)]
    assert a == b                                          # depends on the sorted list and the seed only
    expect = sorted(files)
    np.random.RandomState(12345).shuffle(expect)           # numpy's in-place shuffle of the sorted list, as the reference does
    assert a[0] + a[1] == expect
    assert set(a[0]).isdisjoint(a[1])


def test_synthetic_source_is_deterministic_and_layout_is_hwc():
    ds = ddata.SyntheticHWC(6, 8, 12, channels=[0, 3, 5, 7, 8, 9, 10, 15], comm_size=2, comm_rank=1)
    assert len(ds) == 3 and ds.data_shape == (8, 12, 16) and ds.label_shape == (8, 12)
    d1, l1 = np.empty((8, 12, 16), np.float32), np.empty((8, 12), np.int64)
    d2, l2 = np.empty_like(d1), np.empty_like(l1)
    n1 = ds.read_into(1, d1, l1)
    n2 = ddata.SyntheticHWC(6, 8, 12, comm_size=1, comm_rank=0).read_into(4, d2, l2)      # same global sample
    assert n1 == n2 and np.array_equal(d1, d2) and np.array_equal(l1, l2)
    norm = ds.data_scale * (d1[..., ds.channels] - ds.data_shift)
    assert norm.min() >= 0.0 and norm.max() <= 1.0
    assert set(np.unique(l1)).issubset({0, 1, 2})
