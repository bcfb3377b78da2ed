"""Data parallelism: rank wire-up (reference utils/comm.py:26-113) and the gradient all-reduce that apex/torch DDP does
implicitly at train_hdf5_ddp.py:227,363 -- here explicit, zero-copy and overlapped with backward.

Design for xGMI (point-to-point links, ring collectives per-link bound): all 301 gradients already live in ONE flat fp32
arena in parameter order, and backward finishes them roughly from the arena's end to its start.  Buckets are therefore
plain contiguous arena ranges (no gather/scatter copies): as soon as the last gradient of a range has been written, the
range is handed to RCCL (``all_reduce(SUM)`` on the process group's own stream, which first waits for the compute stream's
current position) while backward keeps running.  ``finish()`` makes the compute stream wait for the outstanding
collectives and re-arms the buckets for the next backward.  The 1/world averaging is either folded into the optimizer kernel
(``TrainStep.attach_reducer``: grad_scale, no extra pass) or, on the reference's own loop through ``DistributedDataParallel``
(``loss.backward(); optimizer.step()``), applied to the arena by ``finish(average=True)`` at the end of backward, so that
``p.grad`` is the averaged gradient exactly as under apex / torch DDP.  BatchNorm statistics stay per rank, as in the
reference (no SyncBN).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------------------------
# wire-up (same method names as the reference, plus "env" for torchrun-style launches)
# ---------------------------------------------------------------------------------------------------------------------
def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_local_rank() -> int:
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    n = torch.cuda.device_count()
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"]) % n if n > 0 else 0
    return dist.get_rank() % n if n > 0 else 0                       # comm.py:45-46


def init(method: str, backend: Optional[str] = None) -> None:
    """comm.init: derive rank / world / master from the launcher's environment and bring up the process group.
    'nccl' is RCCL on ROCm.  `backend` (or the environment variable DC_DIST_BACKEND) overrides: tests use gloo, on the CPU
    and for several ranks sharing one GPU."""
    backend = backend or os.environ.get("DC_DIST_BACKEND") or None
    port = os.environ.get("DC_MASTER_PORT", "29500")      # the reference hard-codes 29500 (comm.py:71,83,96); tests pick a free one
    if method == "nccl-openmpi":
        addrport = os.getenv("PMIX_SERVER_URI2").split("//")[1]
        os.environ["MASTER_ADDR"] = addrport.split(":")[0]
        os.environ["MASTER_PORT"] = port
        rank = int(os.getenv("OMPI_COMM_WORLD_RANK", 0))
        world = int(os.getenv("OMPI_COMM_WORLD_SIZE", 0))
    elif method == "nccl-slurm":
        rank, world = int(os.getenv("PMIX_RANK")), int(os.getenv("SLURM_NTASKS"))
        os.environ["MASTER_ADDR"] = os.getenv("SLURM_LAUNCH_NODE_IPADDR")
        os.environ["MASTER_PORT"] = port
    elif method == "nccl-slurm-pmi":
        rank, world = int(os.getenv("PMI_RANK")), int(os.getenv("SLURM_NTASKS"))
        os.environ["MASTER_ADDR"] = os.getenv("SLURM_LAUNCH_NODE_IPADDR")
        os.environ["MASTER_PORT"] = port
    elif method == "mpi":
        dist.init_process_group(backend="mpi")
        return
    elif method == "env":
        rank, world = int(os.getenv("RANK", 0)), int(os.getenv("WORLD_SIZE", 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    elif method == "single":
        return
    else:
        raise NotImplementedError()
    dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world)


# ---------------------------------------------------------------------------------------------------------------------
# bucketed, overlapped gradient all-reduce over contiguous arena ranges
# ---------------------------------------------------------------------------------------------------------------------
class Bucket:
    __slots__ = ("lo", "hi", "names", "remaining", "work")

    def __init__(self, lo: int, hi: int, names: List[str]):
        self.lo, self.hi, self.names = lo, hi, names
        self.remaining = len(names)
        self.work = None


def plan_buckets(param_offsets: "Dict[str, tuple]", total: int, bucket_elems: int) -> List[Bucket]:
    """Cut [0,total) at tensor boundaries into ranges of about bucket_elems, walking from the END of the arena (the part
    backward finishes first) so that the last, possibly short, bucket is the one that completes last."""
    items = sorted(((off, n, name) for name, (off, n) in param_offsets.items()), reverse=True)
    buckets: List[Bucket] = []
    hi, names, size = total, [], 0
    for off, n, name in items:
        names.append(name)
        size += n
        if size >= bucket_elems:
            buckets.append(Bucket(off, hi, names))
            hi, names, size = off, [], 0
    if names:
        buckets.append(Bucket(0, hi, names))
    return buckets


class LibraryComm:
    """A communicator of the library's own (csrc/comm.cpp: dc_comm_*, dc_grad_allreduce_enqueue / _wait).  Two transports:
    `rccl()` brings up an RCCL communicator from a unique id that rank 0 creates and torch.distributed (any backend) carries to the other
    ranks; `over_torch()` hands every reduction back to torch.distributed through a host callback -- how the gloo tests (no RCCL between
    CPU ranks, or between several ranks on one GPU) drive the very same entry points.  Both are driven by word-sized C calls only, so a
    recorded launch list (TrainStep.enable_program) replays the collectives with the kernels."""

    def __init__(self, handle, keep=()):
        self.h = handle
        self._keep = list(keep)

    @classmethod
    def rccl(cls, group=None) -> "LibraryComm":
        import ctypes as C
        from . import lib as L
        rank, world = dist.get_rank(group) if dist.is_initialized() else 0, dist.get_world_size(group) if dist.is_initialized() else 1
        box = [None]
        if rank == 0:
            buf = (C.c_char * 128)()
            L.call("dc_comm_unique_id", buf)
            box[0] = bytes(buf)
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        h = C.c_void_p()
        L.call("dc_comm_create", C.c_char_p(box[0]), rank, world, C.byref(h))
        return cls(h)

    @classmethod
    def over_torch(cls, resolve, group=None, sync_stream: bool = True) -> "LibraryComm":
        """resolve(ptr, count, dtype_code) -> the torch tensor (a view of the caller's arena) that starts at device / host address ptr."""
        import ctypes as C
        from . import lib as L
        rank, world = dist.get_rank(group) if dist.is_initialized() else 0, dist.get_world_size(group) if dist.is_initialized() else 1
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)

        def cb(_ctx, ptr, count, dtype_code):
            try:
                t = resolve(ptr, count, dtype_code)
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception:       # an exception must not unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        fn = CB(cb)
        h = C.c_void_p()
        L.call("dc_comm_create_callback", C.cast(fn, C.c_void_p), None, rank, world, 1 if sync_stream else 0, C.byref(h))
        return cls(h, keep=[fn])

    def info(self) -> dict:
        import ctypes as C
        from . import lib as L
        r, w, t, n = C.c_int(), C.c_int(), C.c_int(), C.c_long()
        L.call("dc_comm_info", self.h, C.byref(r), C.byref(w), C.byref(t), C.byref(n))
        return {"rank": r.value, "world": w.value, "transport": ("none", "rccl", "callback")[t.value], "enqueued": n.value}

    def close(self) -> None:
        if self.h:
            from . import lib as L
            L.call("dc_comm_destroy", self.h)
            self.h = None


class _Enqueued:
    """stands in for torch's Work object on the library path: completion is a stream-side wait issued once by finish()"""

    def wait(self):
        return True


class GradReducer:
    """payload: "fp32" all-reduces the arena ranges in place (zero copies); "bf16" rounds each bucket into a bf16 send buffer
    (dc_grad_pack_bf16), sums THAT across ranks and widens the result back into the fp32 arena (dc_grad_unpack_bf16): half the
    bytes on the xGMI links (112.9 MB instead of 225.8 MB per step) for two extra passes over the bucket."""

    def __init__(self, engine, world: int, bucket_mb: float = 32.0, group=None, payload: Optional[str] = None, collective: Optional[str] = None,
                 comm: "Optional[LibraryComm]" = None):
        """collective: "torch" (default) = torch.distributed's all_reduce on the process group's stream; "library" (or DC_GRAD_COLLECTIVE=lib)
        = dc_grad_allreduce_enqueue / _wait of the HIP library over `comm` (default: an RCCL communicator of its own, LibraryComm.rccl):
        every call of the step is then a C call and the step can be replayed from a recorded launch list at any world size."""
        self.eng, self.world, self.group = engine, world, group
        self.collective = (collective or os.environ.get("DC_GRAD_COLLECTIVE", "torch")).lower()
        self.collective = {"lib": "library"}.get(self.collective, self.collective)
        if self.collective not in ("torch", "library"):
            raise ValueError(f"gradient collective must be 'torch' or 'library', got {self.collective!r}")
        self.comm = comm
        self.payload = (payload or os.environ.get("DC_GRAD_PAYLOAD", "fp32")).lower()
        if self.payload not in ("fp32", "bf16"):
            raise ValueError(f"gradient payload must be 'fp32' or 'bf16', got {self.payload!r}")
        lay = engine.layout
        import math
        offs = {name: (p.offset, math.prod(p.shape)) for name, p in lay.params.items()}
        self.buckets = plan_buckets(offs, lay.n_params, int(bucket_mb * 2 ** 20 / 4))
        self.of: Dict[str, Bucket] = {n: b for b in self.buckets for n in b.names}
        covered = sum(b.hi - b.lo for b in self.buckets)
        assert covered == lay.n_params and all(b.hi > b.lo for b in self.buckets)
        self._send = None
        if self.payload == "bf16":
            # one send buffer for the whole arena (bucket b uses [lo, hi) of it, so buckets in flight never alias)
            self._send = torch.empty(lay.n_params, dtype=torch.bfloat16, device=engine.grads.device)
            # dc_grad_pack_bf16 / dc_grad_unpack_bf16 move 16-byte vectors of both arenas: every bucket must start on an 8-element boundary
            bad = [b.lo for b in self.buckets if b.lo % 8]
            if bad:
                raise ValueError(f"bf16 gradient payload needs bucket offsets that are multiples of 8 elements, got {bad[:4]}")
        self.engines = []
        self.hook(engine)
        self.launched = 0
        self.averaging_in_optimizer = False     # set by TrainStep.attach_reducer: the 1/world then rides in the optimizer kernel
        self._owns_comm = False
        if self.collective == "library" and (self.comm is None or isinstance(self.comm, str)):
            self._owns_comm = True
            # default: an RCCL communicator of the library's own; "callback" (or DC_GRAD_COLLECTIVE_TRANSPORT=callback): torch.distributed does
            # the reduction inside a host callback (gloo tests)
            transport = self.comm if isinstance(self.comm, str) else os.environ.get("DC_GRAD_COLLECTIVE_TRANSPORT", "rccl")
            if transport == "callback":
                self.comm = LibraryComm.over_torch(self._resolve, group, sync_stream=engine.grads.is_cuda)
            else:
                self.comm = LibraryComm.rccl(group)

    def close(self) -> None:
        """Give back the communicator this reducer created (an ncclComm_t, a stream and two events on the RCCL transport); one the caller
        passed in stays the caller's."""
        if self._owns_comm and isinstance(self.comm, LibraryComm):
            self.comm.close()
            self._owns_comm = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _resolve(self, ptr: int, count: int, dtype_code: int):
        """callback transport: the arena view an address handed to dc_grad_allreduce_enqueue refers to"""
        arena = self._send if dtype_code == 1 else self.eng.grads
        off = (ptr - arena.data_ptr()) // arena.element_size()
        assert 0 <= off and off + count <= arena.numel(), "address outside the gradient arena"
        return arena[off:off + count]

    def hook(self, engine) -> None:
        """Report gradient readiness from this engine too.  Every engine that shares the arena (nn.DeepLabv3_plus builds one per
        batch shape) must be hooked, or a backward through it would leave the buckets unreduced."""
        if engine not in self.engines:
            assert engine.grads.data_ptr() == self.eng.grads.data_ptr(), "engines of one reducer share one gradient arena"
            engine.on_grad_ready = self.ready
            self.engines.append(engine)

    def reset(self) -> None:
        for b in self.buckets:
            b.remaining = len(b.names)
            b.work = None
        self.launched_last = self.launched       # buckets the step that has just finished sent to the collective (bench.py's comm block)
        self.launched = 0

    def _reduce(self, b: Bucket):
        if self.world <= 1:
            return None                          # a single rank has nothing to reduce
        g = self.eng.grads[b.lo:b.hi]
        lib_path = self.collective == "library"
        if lib_path:
            import ctypes as C
            from . import lib as L
            st = L.stream_ptr() if g.is_cuda else None
        if self._send is None:
            if lib_path:
                L.call("dc_grad_allreduce_enqueue", self.comm.h, C.c_void_p(g.data_ptr()), b.hi - b.lo, L.DC_F32, st)
                return _Enqueued()
            return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        s = self._send[b.lo:b.hi]
        if g.is_cuda:
            from . import lib as L
            L.call("dc_grad_pack_bf16", b.hi - b.lo, L.dptr(g), L.dptr(s), L.stream_ptr())
        else:                                   # host arenas exist only in the CPU tests of this class
            s.copy_(g)
        if lib_path:
            L.call("dc_grad_allreduce_enqueue", self.comm.h, C.c_void_p(s.data_ptr()), b.hi - b.lo, L.DC_BF16, st)
            return _Enqueued()
        return dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _unpack(self, b: Bucket) -> None:
        g, s = self.eng.grads[b.lo:b.hi], self._send[b.lo:b.hi]
        if g.is_cuda:
            from . import lib as L
            L.call("dc_grad_unpack_bf16", b.hi - b.lo, L.dptr(s), L.dptr(g), L.stream_ptr())
        else:
            g.copy_(s)

    def ready(self, names: List[str]) -> None:
        """Engine callback: these parameters' gradients have just been enqueued (final values) on the compute stream."""
        for n in names:
            b = self.of[n]
            b.remaining -= 1
            if b.remaining == 0:
                # the process group's stream waits for everything enqueued so far on the launching stream, then reduces.
                # Weight gradients are produced on the engine's side stream, the BatchNorm ones on the main stream: the
                # launch happens on the side stream after it has been ordered behind the main stream's current position,
                # so it covers both without stalling backward.
                side = getattr(self.eng, "side", None) if getattr(self.eng, "use_side_stream", False) else None
                if side is not None:
                    if self.collective == "library":       # a C call, so that a recorded launch list holds the fence too
                        import ctypes as C
                        from . import lib as L
                        L.call("dc_stream_fence", L.stream_ptr(), C.c_void_p(side.cuda_stream))
                    else:
                        side.wait_event(torch.cuda.current_stream().record_event())
                    with torch.cuda.stream(side):
                        b.work = self._reduce(b)
                else:
                    b.work = self._reduce(b)
                self.launched += 1

    def finish(self, average: bool = False) -> None:
        """Order the optimizer behind all outstanding collectives (stream-side wait on GPU; blocking on CPU/gloo) and re-arm
        the buckets.  average=True also divides the reduced arena by the world size in place (the DDP wrapper's path, where
        the caller's optimizer knows nothing about the reduction); with average=False the arena holds the SUM and the
        optimizer kernel applies 1/world (grad_scale).  The two are exclusive: a reducer attached to a fused TrainStep refuses
        average=True (the gradients would be scaled by 1/world twice)."""
        if average and self.averaging_in_optimizer:
            raise RuntimeError("this reducer is attached to a fused TrainStep whose optimizer already applies 1/world; "
                               "finish(average=True) would average twice")
        missing = [b for b in self.buckets if b.remaining != 0]
        if self.collective == "library" and any(b.work is not None for b in self.buckets):
            from . import lib as L                   # the compute stream waits for every collective enqueued so far
            L.call("dc_grad_allreduce_wait", self.comm.h, L.stream_ptr() if self.eng.grads.is_cuda else None)
        for b in self.buckets:                       # wait for what WAS launched even when the backward was incomplete
            if b.work is not None:
                b.work.wait()
                if self._send is not None:
                    self._unpack(b)
        self.reset()                                 # re-armed for the next backward either way
        if missing:
            b = missing[0]
            raise RuntimeError(f"gradient bucket [{b.lo},{b.hi}) never completed: {b.remaining} gradients missing "
                               f"({len(missing)} of {len(self.buckets)} buckets incomplete)")
        if average and self.world > 1:
            self.eng.grads.mul_(1.0 / self.world)

    def broadcast_parameters(self, src: int = 0) -> None:
        """DDP construction semantics (train_hdf5_ddp.py:227): every rank starts from rank 0's weights and BN buffers."""
        dist.broadcast(self.eng.params, src=src, group=self.group)
        dist.broadcast(self.eng.buffers, src=src, group=self.group)
        dist.broadcast(self.eng.nbt, src=src, group=self.group)
        if hasattr(self.eng, "mark_weights_changed"):
            self.eng.mark_weights_changed()


class DistributedDataParallel(torch.nn.Module):
    """The wrapper the reference applies at train_hdf5_ddp.py:227: forwards to the module, prefixes state-dict keys with
    'module.' (checkpoint format, :521) and averages gradients across ranks during backward."""

    def __init__(self, module, bucket_mb: float = 32.0, payload: Optional[str] = None):
        super().__init__()
        self.module = module
        self.reducer = None
        self._bucket_mb, self._payload = bucket_mb, payload
        if get_size() > 1 and module.engine is not None:
            self._attach()

    def _attach(self):
        self.reducer = GradReducer(self.module.engine, get_size(), self._bucket_mb, payload=self._payload)
        for eng in getattr(self.module, "_engines", {}).values():     # engines of other batch shapes share the arena
            self.reducer.hook(eng)
        self.reducer.broadcast_parameters()
        # loss.backward() through the module (nn._NetFn.backward) ends with reducer.finish(average=True): the gradients the
        # caller's optimizer.step() then reads are final and averaged, as apex / torch DDP guarantee at train_hdf5_ddp.py:363-364
        self.module._ddp_reducer = self.reducer

    def forward(self, *args, **kw):
        if self.reducer is None and get_size() > 1 and self.module.engine is not None:
            self._attach()
        return self.module.forward(*args, **kw)

    def load_state_dict(self, state_dict, strict: bool = True):
        return self.module.load_state_dict(state_dict, strict=strict)

    def state_dict(self, *args, **kw):
        from collections import OrderedDict
        return OrderedDict(("module." + k, v) for k, v in self.module.state_dict(*args, **kw).items())
