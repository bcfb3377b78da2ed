"""N>1 host logic on CPU: bucket planning and the overlapped gradient all-reduce, world_size 2 over gloo."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlperf_deepcam_amd import dist as ddist
from mlperf_deepcam_amd import spec


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeEngine:
    """What GradReducer needs from an engine: the layout, the flat gradient arena and the readiness callback."""

    def __init__(self, layout):
        self.layout = layout
        self.params = torch.zeros(layout.n_params)
        self.grads = torch.zeros(layout.n_params)
        self.buffers = torch.zeros(layout.n_buffers)
        self.nbt = torch.zeros(len(layout.nbt), dtype=torch.int64)
        self.on_grad_ready = None
        self.changed = 0

    def mark_weights_changed(self):
        self.changed += 1


def test_bucket_plan_covers_arena_in_backward_order():
    lay = spec.Layout()
    offs = {n: (p.offset, math.prod(p.shape)) for n, p in lay.params.items()}
    buckets = ddist.plan_buckets(offs, lay.n_params, 8 * 2 ** 20)
    assert buckets[0].hi == lay.n_params and buckets[-1].lo == 0
    for a, b in zip(buckets, buckets[1:]):
        assert a.lo == b.hi                                    # contiguous, descending
    assert sum(len(b.names) for b in buckets) == 301
    assert "upsample.last_deconv.0.weight" in buckets[0].names and "xception_features.conv1.weight" in buckets[-1].names
    assert 5 <= len(buckets) <= 10


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ddist.init("env", backend="gloo")
    assert ddist.get_rank() == rank and ddist.get_size() == world
    lay = spec.Layout()
    eng = FakeEngine(lay)
    eng.params.fill_(float(rank + 1))
    red = ddist.GradReducer(eng, world, bucket_mb=16.0)
    red.broadcast_parameters()
    assert float(eng.params[0]) == 1.0 and float(eng.params[-1]) == 1.0 and eng.changed == 1
    names = list(lay.params)
    for step in range(2):
        # "backward": gradients become ready from the end of the arena, a few tensors at a time
        g = torch.Generator().manual_seed(100 * step + rank)
        local = torch.randn(lay.n_params, generator=g)
        for lo in range(len(names) - 1, -1, -7):
            chunk = names[max(0, lo - 6):lo + 1]
            for n in chunk:
                p = lay.params[n]
                k = math.prod(p.shape)
                eng.grads[p.offset:p.offset + k] = local[p.offset:p.offset + k]
            eng.on_grad_ready(chunk)
        assert red.launched == len(red.buckets)
        # step 0: the fused-step contract (the arena holds the SUM, the optimizer applies 1/world); step 1: the DDP wrapper's
        # contract (finish(average=True) leaves the averaged gradient).  Either way the buckets are re-armed for the next backward.
        red.finish(average=(step == 1))
        expect = sum(torch.randn(lay.n_params, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world))
        if step == 1:
            expect = expect / world
        assert torch.allclose(eng.grads, expect, atol=1e-6)
        assert red.launched == 0 and all(b.remaining == len(b.names) and b.work is None for b in red.buckets)
    # a backward that never reported some gradient must not pass silently
    eng.on_grad_ready(names[:5])
    try:
        red.finish()
        raise AssertionError("finish() accepted an incomplete backward")
    except RuntimeError as e:
        assert "never completed" in str(e)
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


def test_wireup_rejects_unknown_method():
    with pytest.raises(NotImplementedError):
        ddist.init("carrier-pigeon")
