"""Phase-bracketed profiling driver: the MI355X counterpart of the reference's profile_hdf5_ddp.py.

The reference wraps ONE phase of the train step (``--profile Forward | Backward | Optimizer``) in
``pyc.driver.start_profiler() / stop_profiler()`` after ``--num_warmup_steps`` un-profiled steps, so that Nsight records only
that phase (profile_hdf5_ddp.py:77-94 the ``Profile`` context manager, :196-236 the three bracketed phases, :270-272 the flags).
Here the same three phases of the HIP engine are bracketed with roctx: a named range (``roctxRangePush/Pop``) and the
profiler's pause / resume control (``roctxProfilerPause/Resume``), which rocprofv3 honours:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --marker-trace --stats -d out -- python3 -m mlperf_deepcam_amd.profile --profile Backward \\
        --local_batch_size 8 --optimizer LAMB --amp_opt_level O1

Only the kernels of the chosen phase of the ``--num_profile_steps`` profiled steps reach the trace.  Independently of any
profiler the driver times the three phases with HIP events on the compute stream and prints one JSON line (phase -> ms per
step), so a plain run already gives the forward / backward / optimizer split of the step.

Phases (train_hdf5_ddp.py:348-364):
  Forward    net.forward(inputs) + fp_loss            (the fused loss kernel also leaves d(loss)/d(logits) behind)
  Backward   optimizer.zero_grad() + loss.backward()  (weight gradients on the side stream are joined inside the phase)
  Optimizer  optimizer.step()                         (Adam / AdamW / LAMB over the flat arena + the weight repack it triggers)
"""
from __future__ import annotations

import argparse as ap
import ctypes
import json
import os

import torch

from . import nn as dnn


class Roctx:
    """ctypes view of the roctx entry points rocprofv3 listens to; every call is a no-op when no roctx library is present."""

    def __init__(self):
        self.lib = None
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                self.lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if self.lib is not None:
            self.lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            self.lib.roctxRangePushA.restype = ctypes.c_int
            self.lib.roctxRangePop.restype = ctypes.c_int
            for fn in ("roctxProfilerPause", "roctxProfilerResume"):
                if hasattr(self.lib, fn):
                    getattr(self.lib, fn).argtypes = [ctypes.c_uint64]
                    getattr(self.lib, fn).restype = ctypes.c_int

    def push(self, name: str) -> None:
        if self.lib is not None:
            self.lib.roctxRangePushA(name.encode())

    def pop(self) -> None:
        if self.lib is not None:
            self.lib.roctxRangePop()

    def pause(self) -> None:
        if self.lib is not None and hasattr(self.lib, "roctxProfilerPause"):
            self.lib.roctxProfilerPause(0)

    def resume(self) -> None:
        if self.lib is not None and hasattr(self.lib, "roctxProfilerResume"):
            self.lib.roctxProfilerResume(0)


class Profile:
    """The reference's ``Profile`` context manager (profile_hdf5_ddp.py:77-94): active when this phase is the selected one and
    the warm-up steps are over.  Entering resumes the profiler and opens a roctx range; leaving drains the GPU (so the phase's
    kernels are inside the bracket), closes the range and pauses the profiler again."""

    def __init__(self, roctx: Roctx, selected: str, flag: str, step: int, num_warmup_steps: int):
        self.roctx, self.flag = roctx, flag
        self.active = (flag == selected) and (step >= num_warmup_steps)

    def __enter__(self):
        if self.active:
            torch.cuda.synchronize()
            self.roctx.resume()
            self.roctx.push(self.flag)
        return self

    def __exit__(self, *exc):
        if self.active:
            torch.cuda.synchronize()
            self.roctx.pop()
            self.roctx.pause()
        return False


def build_parser():
    AP = ap.ArgumentParser()
    AP.add_argument("--local_batch_size", type=int, default=2, help="Number of samples per local minibatch")
    AP.add_argument("--num_warmup_steps", type=int, default=5, help="Number of warmup steps")
    AP.add_argument("--num_profile_steps", type=int, default=1, help="Number of profiling steps")
    AP.add_argument("--profile", type=str, default="Forward", choices=["Forward", "Backward", "Optimizer"], help="Flag which parts to profile")
    AP.add_argument("--channels", type=int, nargs="+", default=list(range(16)), help="Channels used in input")
    AP.add_argument("--optimizer", type=str, default="Adam", choices=["Adam", "AdamW", "LAMB"], help="Optimizer to use")
    AP.add_argument("--start_lr", type=float, default=1e-3, help="Start LR")
    AP.add_argument("--adam_eps", type=float, default=1e-8, help="Adam Epsilon")
    AP.add_argument("--weight_decay", type=float, default=1e-6, help="Weight decay")
    AP.add_argument("--loss_weight_pow", type=float, default=-0.125, help="Decay factor to adjust the weights")
    AP.add_argument("--amp_opt_level", type=str, default="O0", help="O0 = fp32 activations, O1/O2 = bf16 activations")
    AP.add_argument("--height", type=int, default=768)
    AP.add_argument("--width", type=int, default=1152)
    return AP


def main(pargs) -> dict:
    if not torch.cuda.is_available():
        raise RuntimeError("mlperf_deepcam_amd.profile needs an MI355X: there is no CPU path")
    torch.manual_seed(333)
    dev = torch.device("cuda", torch.cuda.current_device())
    dtype = torch.float32 if pargs.amp_opt_level == "O0" else torch.bfloat16
    B, H, W, C_ = pargs.local_batch_size, pargs.height, pargs.width, len(pargs.channels)
    roctx = Roctx()
    roctx.pause()                                        # nothing is recorded until the selected phase of a profiled step

    net = dnn.DeepLabv3_plus(n_input=C_, n_classes=3, os=16, pretrained=False, _print=False, dtype=dtype, seed=333)
    net.materialize(B, H, W)
    net.train()
    eng = net.engine
    opt = dnn.make_optimizer(pargs.optimizer, net, pargs.start_lr, pargs.adam_eps, pargs.weight_decay)
    cw = dnn.class_weights(pargs.loss_weight_pow)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(B, C_, H, W, generator=g).to(dev)
    y = torch.multinomial(torch.tensor(dnn.CLASS_FREQ), B * H * W, replacement=True, generator=g).view(B, H, W).to(dev)
    loss_sum = torch.zeros(1, dtype=torch.float64, device=dev)

    total = pargs.num_warmup_steps + pargs.num_profile_steps
    marks = []
    for step in range(total):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        with Profile(roctx, pargs.profile, "Forward", step, pargs.num_warmup_steps):
            loss_sum.zero_()
            logits = eng.forward(x, train=True)
            dnn.wce_fused(logits, y, cw, dlogits=eng.dlogits, loss_sum=loss_sum)
        ev[1].record()
        with Profile(roctx, pargs.profile, "Backward", step, pargs.num_warmup_steps):
            opt.zero_grad()
            eng.backward()
        ev[2].record()
        with Profile(roctx, pargs.profile, "Optimizer", step, pargs.num_warmup_steps):
            opt.step()
            eng.pack_weights()                           # the repack the next forward would trigger belongs to the update
        ev[3].record()
        if step >= pargs.num_warmup_steps:
            marks.append(ev)
    torch.cuda.synchronize()
    n = len(marks)
    ms = {name: sum(e[i].elapsed_time(e[i + 1]) for e in marks) / n for i, name in enumerate(("Forward", "Backward", "Optimizer"))}
    out = {"profile": pargs.profile, "local_batch": B, "dtype": "fp32" if dtype == torch.float32 else "bf16", "optimizer": pargs.optimizer,
           "height": H, "width": W, "warmup_steps": pargs.num_warmup_steps, "profile_steps": n,
           "ms_per_step": {k: round(v, 3) for k, v in ms.items()}, "loss": float(loss_sum.item()) / (B * H * W),
           "roctx": roctx.lib is not None}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    main(build_parser().parse_args())
