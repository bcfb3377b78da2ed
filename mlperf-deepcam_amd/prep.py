"""Dataset preparation: the statistics file the loader normalises with, and the train / validation /
test split.  Reference: src/utils/summarize_data.py (per-file per-channel mean, mean of squares, min, max over H and W, merged as
count-weighted running means / elementwise min / max, written as stats.h5 `climate/{count,mean,sqmean,minval,maxval}`) and
src/utils/split_data.py (sorted listing, np.random.seed(12345) + shuffle, 80 / 10 / 10 by int() truncation, symbolic links).
The reference spreads the files over MPI ranks; here a thread pool reads them (file reads release the GIL) and the per-file
tokens are merged in file order, so the result does not depend on the number of workers.

    python -m mlperf_deepcam_amd.prep summarize /data            # reads /data/train/data-*.h5, writes /data/stats.h5
    python -m mlperf_deepcam_amd.prep split /data/all /data
"""
from __future__ import annotations

import argparse
import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np

from . import data as D


def _read_data(path: str) -> np.ndarray:
    backend, h5 = D._h5_backend()
    if backend == "h5py":
        with h5.File(path, "r") as f:
            return f["climate/data"][...]
    with h5.File(path) as f:
        return f.read("climate/data")


def file_token(path: str, data_format: str = "nhwc") -> Tuple[int, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(1, mean, mean of squares, min, max) per channel of one sample (summarize_data.py:78-99)."""
    arr = _read_data(path)
    axis = (1, 2) if data_format == "nchw" else (0, 1)
    return 1, np.mean(arr, axis=axis), np.mean(np.square(arr), axis=axis), np.amin(arr, axis=axis), np.amax(arr, axis=axis)


def merge_token(t1, t2):
    """Count-weighted means, elementwise min / max (summarize_data.py:53-75)."""
    n1, n2 = t1[0], t2[0]
    n = n1 + n2
    return (n, float(n1) / float(n) * t1[1] + float(n2) / float(n) * t2[1], float(n1) / float(n) * t1[2] + float(n2) / float(n) * t2[2],
            np.minimum(t1[3], t2[3]), np.maximum(t1[4], t2[4]))


def summarize(data_path_prefix: str, out_path: str = None, data_format: str = "nhwc", workers: int = 8, subdir: str = "train") -> str:
    root = os.path.join(data_path_prefix, subdir)
    files = sorted(os.path.join(root, x) for x in os.listdir(root) if x.endswith(".h5") and x.startswith("data-"))
    if not files:
        raise FileNotFoundError(f"no data-*.h5 files under {root}")
    with ThreadPoolExecutor(max(1, workers)) as pool:
        tokens = list(pool.map(lambda p: file_token(p, data_format), files))
    token = tokens[0]
    for t in tokens[1:]:
        token = merge_token(t, token)
    out_path = out_path or os.path.join(data_path_prefix, "stats.h5")
    backend, h5 = D._h5_backend()
    entries = {"climate/count": np.int64(token[0]), "climate/mean": token[1], "climate/sqmean": token[2], "climate/minval": token[3],
               "climate/maxval": token[4]}
    if backend == "h5py":
        with h5.File(out_path, "w") as f:
            for k, v in entries.items():
                f[k] = v
    else:
        with h5.File(out_path, "w") as f:
            for k, v in entries.items():
                f.write(k, v)
    return out_path


def split_lists(names: Sequence[str], train_fraction: float = 0.8, validation_fraction: float = 0.1, seed: int = 12345):
    """The reference's split of a directory listing (split_data.py:36-72): returns (train, validation, test) name lists."""
    files: List[str] = sorted(x for x in names if x.startswith("data") and x.endswith(".h5"))
    rs = np.random.RandomState(seed)                      # np.random.seed(seed); np.random.shuffle(files)
    rs.shuffle(files)
    num_train = int(len(files) * train_fraction)
    num_validation = int(len(files) * validation_fraction)
    return files[:num_train], files[num_train:num_train + num_validation], files[num_train + num_validation:]


def split(inputdir: str, outputdir: str, train_fraction: float = 0.8, validation_fraction: float = 0.1, seed: int = 12345):
    train, validation, test = split_lists(os.listdir(inputdir), train_fraction, validation_fraction, seed)
    print("Following split will be used: ")
    print("Total files: {}".format(len(train) + len(validation) + len(test)))
    print("Train files: {}".format(len(train)))
    print("Validation files: {}".format(len(validation)))
    print("Test files: {}".format(len(test)))
    for sub, names in (("train", train), ("validation", validation), ("test", test)):
        d = os.path.join(outputdir, sub)
        os.makedirs(d, exist_ok=True)
        for f in names:
            os.symlink(os.path.join(inputdir, f), os.path.join(d, f))
    return train, validation, test


def main(argv=None):
    ap = argparse.ArgumentParser(prog="mlperf_deepcam_amd.prep")
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("summarize")
    a.add_argument("data_path_prefix")
    a.add_argument("--output", default=None)
    a.add_argument("--data_format", default="nhwc", choices=["nhwc", "nchw"])
    a.add_argument("--workers", type=int, default=8)
    b = sub.add_parser("split")
    b.add_argument("inputdir")
    b.add_argument("outputdir")
    b.add_argument("--train_fraction", type=float, default=0.8)
    b.add_argument("--validation_fraction", type=float, default=0.1)
    b.add_argument("--seed", type=int, default=12345)
    args = ap.parse_args(argv)
    if args.cmd == "summarize":
        print(summarize(args.data_path_prefix, args.output, args.data_format, args.workers))
    else:
        split(args.inputdir, args.outputdir, args.train_fraction, args.validation_fraction, args.seed)


if __name__ == "__main__":
    main()
