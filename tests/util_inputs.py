"""Seeded synthetic inputs shared by the tests, smoke() and the golden generator's recipe."""
import numpy as np
import torch

CLASS_FREQ = (0.986267818390377, 0.0004578708870701058, 0.01327431072255291)   # train_hdf5_ddp.py:206


def make_inputs(B, H, W, seed=1234):
    """Same recipe as tests/golden/make_golden.py::make_inputs (uniform labels)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 16, H, W, generator=g)
    y = torch.randint(0, 3, (B, H, W), generator=g)
    return x, y


def sample_index(n, count=64, seed=99):
    rs = np.random.RandomState(seed)
    return rs.randint(0, n, size=count).astype(np.int64)


def digest(t):
    f = t.detach().double().flatten().cpu()
    return {"sum": float(f.sum()), "abs": float(f.abs().sum()), "head": [float(x) for x in f[:4]]}
