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


def _worker(rank, world, port, q, collective="torch"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ddist.init("env", backend="gloo")
    assert ddist.get_rank() == rank and ddist.get_size() == world
    lay = spec.Layout()
    eng = FakeEngine(lay)
    eng.params.fill_(float(rank + 1))
    # "library": the reductions go through dc_grad_allreduce_enqueue / _wait of libdeepcam_hip.so (csrc/comm.cpp) with the callback transport,
    # i.e. the library calls back into torch.distributed over gloo -- the entry points the RCCL transport shares
    red = ddist.GradReducer(eng, world, bucket_mb=16.0, collective=collective, comm="callback" if collective == "library" else None)
    if collective == "library":
        assert red.comm.info() == {"rank": rank, "world": world, "transport": "callback", "enqueued": 0}
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
    if collective == "library":
        assert red.comm.info()["enqueued"] >= 2 * len(red.buckets)
        red.comm.close()
    q.put((rank, "ok"))
    dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["torch", "library"])
def test_grad_reducer_world2_gloo(collective):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, collective)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


class _P:
    def __init__(self, offset, shape):
        self.offset, self.shape = offset, shape


class SmallLayout:
    """A layout with the attributes GradReducer reads, small enough for many ranks on the CPU."""

    def __init__(self, ntensors=37, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.params, off = {}, 0
        for i in range(ntensors):
            n = int(torch.randint(8, 40000, (1,), generator=g))
            n = (n + 7) // 8 * 8               # arena ranges stay 16-byte aligned in both payload widths
            self.params[f"t{i}"] = _P(off, (n,))
            off += n
        self.n_params, self.n_buffers, self.nbt = off, 16, {"a": 0}


def _wireup_worker(rank, world, port, method, q):
    # what the launcher of each reference wire-up method exports (utils/comm.py:64-108), nothing else
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        os.environ.pop(k, None)
    os.environ["DC_MASTER_PORT"] = str(port)
    os.environ["DC_DIST_BACKEND"] = "gloo"
    if method == "nccl-openmpi":
        os.environ.update(PMIX_SERVER_URI2=f"pmix-server.1234;tcp4://127.0.0.1:{40000 + rank}", OMPI_COMM_WORLD_RANK=str(rank),
                          OMPI_COMM_WORLD_SIZE=str(world))
    elif method == "nccl-slurm":
        os.environ.update(PMIX_RANK=str(rank), SLURM_NTASKS=str(world), SLURM_LAUNCH_NODE_IPADDR="127.0.0.1")
    else:
        os.environ.update(PMI_RANK=str(rank), SLURM_NTASKS=str(world), SLURM_LAUNCH_NODE_IPADDR="127.0.0.1")
    ddist.init(method)
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["MASTER_PORT"] == str(port)
    assert ddist.get_rank() == rank and ddist.get_size() == world
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    q.put((rank, float(t)))
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["nccl-openmpi", "nccl-slurm", "nccl-slurm-pmi"])
def test_reference_wireup_methods_bring_up_two_ranks(method):
    """utils/comm.py:64-108: each launcher's environment variables -> rank, world size, master address (gloo stands in for RCCL)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wireup_worker, args=(r, 2, port, method, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, 3.0), (1, 3.0)]


def test_wireup_default_port_is_the_references():
    import inspect
    src = inspect.getsource(ddist.init)
    assert '"29500"' in src        # comm.py:71,83,96


def _many_worker(rank, world, port, payload, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ddist.init("env", backend="gloo")
    lay = SmallLayout()
    eng = FakeEngine(lay)
    red = ddist.GradReducer(eng, world, bucket_mb=0.25, payload=payload)
    assert len(red.buckets) >= 4
    names = list(lay.params)
    order = []
    for step in range(2):
        local = torch.randn(lay.n_params, generator=torch.Generator().manual_seed(10 * step + rank))
        if payload == "bf16":
            local = local.bfloat16().float()
        eng.grads.copy_(local)
        for lo in range(len(names) - 1, -1, -5):
            before = red.launched
            eng.on_grad_ready(names[max(0, lo - 4):lo + 1])
            order += [step] * (red.launched - before)
        assert red.launched == len(red.buckets)
        red.finish()
        expect = sum(torch.randn(lay.n_params, generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world))
        if payload == "bf16":
            # every rank's contribution is a bf16 value and the collective sums in bf16: rounding of the running sum only
            assert torch.allclose(eng.grads, expect, atol=0.12, rtol=0.02), float((eng.grads - expect).abs().max())
            assert float((eng.grads - expect).norm() / expect.norm()) < 1e-2
        else:
            assert torch.allclose(eng.grads, expect, atol=1e-5)
    dist.barrier()
    q.put((rank, len(order)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,payload", [(8, "fp32"), (2, "bf16"), (8, "bf16")])
def test_grad_reducer_many_ranks_and_bf16_payload(world, payload):
    """The N = 8 configuration (BASELINE configs[3], [4]) on the CPU: bucket order, re-arming and the bf16 payload over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_many_worker, args=(r, world, port, payload, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert [g[0] for g in got] == list(range(world)) and len({g[1] for g in got}) == 1


def test_finish_rearms_after_an_incomplete_backward_and_refuses_double_averaging():
    lay = SmallLayout()
    eng = FakeEngine(lay)
    red = ddist.GradReducer(eng, 1, bucket_mb=0.25)
    eng.on_grad_ready(list(lay.params)[:3])
    with pytest.raises(RuntimeError, match="never completed"):
        red.finish()
    assert all(b.remaining == len(b.names) and b.work is None for b in red.buckets)     # clean for the next backward
    red.averaging_in_optimizer = True
    with pytest.raises(RuntimeError, match="average twice"):
        red.finish(average=True)
    other = FakeEngine(lay)
    other.grads = eng.grads                     # an engine of another batch shape shares the arena
    red.hook(other)
    assert other.on_grad_ready == red.ready
    with pytest.raises(ValueError):
        ddist.GradReducer(eng, 1, payload="fp8")


def test_wireup_rejects_unknown_method():
    with pytest.raises(NotImplementedError):
        ddist.init("carrier-pigeon")
