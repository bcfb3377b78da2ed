"""torch custom-op registration of the hot path's operators: ``torch.ops.deepcam.*``.

north_star / SURVEY 8b ask for "hand-written HIP through custom ops".  The kernels live behind the C ABI of libdeepcam_hip.so
(include/deepcam_hip.h, bound by lib.py); this module registers the operators the reference's loop calls
(train_hdf5_ddp.py:348-364,406-407: ``net.forward``, ``criterion``, ``loss.backward``, ``optimizer.step``, ``compute_score``) with
torch's dispatcher, so that nn.py reaches them as ``torch.ops.deepcam.<name>`` -- one dispatcher entry per operator of the step, not
per kernel (a step is ~830 launches; the program that issues them is engine.py's explicit launch list).  Engines and optimizers are
addressed by integer handles (they own device buffers and a compiled launch program; a dispatcher schema has no type for that).

    deepcam::net_forward(Tensor x, int engine, bool train) -> Tensor logits        DeepLabv3_plus.forward (deeplab_xception.py:441-465)
    deepcam::net_backward(Tensor dlogits, int engine) -> ()                        loss.backward() through the network (:363)
    deepcam::wce_fused(...) -> ()                                                  fp_loss + its gradient + argmax + IoU counts (losses.py:28-52)
    deepcam::confusion_counts(Tensor pred, Tensor gt, Tensor(a!) counts) -> ()      compute_score's tp / fp / fn (utils.py:32-60)
    deepcam::optimizer_step(int optimizer) -> ()                                    optimizer.step() (:364)

EAGER ONLY.  The schemas name the operators, not their memory effects: net_forward returns the engine's persistent logits buffer (the
same storage every call; a fused TrainStep does not even write it), net_backward and optimizer_step mutate the engine's arenas behind a
`-> ()` signature.  That is what an eager training loop needs and all this package does with them; under opcheck, functionalization or
torch.compile such operators could be reordered or dropped, so do not trace through them.  The drop-in boundary for another host is the
C ABI (include/deepcam_hip.h), not this registration.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional

import torch

from . import lib as L

_ENGINES: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
_OPTIMIZERS: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()


def engine_handle(engine) -> int:
    h = id(engine)
    _ENGINES[h] = engine
    return h


def optimizer_handle(opt) -> int:
    h = id(opt)
    _OPTIMIZERS[h] = opt
    return h


def _engine(h: int):
    e = _ENGINES.get(h)
    if e is None:
        raise L.DeepcamHipError(f"deepcam custom op: unknown engine handle {h}")
    return e


def _net_forward(x: torch.Tensor, engine: int, train: bool) -> torch.Tensor:
    return _engine(engine).forward(x, train=train)


def _net_backward(dlogits: torch.Tensor, engine: int) -> None:
    eng = _engine(engine)
    if dlogits.data_ptr() != eng.dlogits.data_ptr():
        eng.dlogits.copy_(dlogits)
    eng.backward()


def _wce_fused(logits: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, scale: float, loss_sum: torch.Tensor,
               dlogits: Optional[torch.Tensor], pred: Optional[torch.Tensor], counts: Optional[torch.Tensor]) -> None:
    B, _, H, W = logits.shape
    L.call("dc_wce_fused", B, H, W, L.dptr(logits), L.dptr(target), target.element_size(), L.dptr(weight), float(scale), L.dptr(loss_sum),
           L.dptr(dlogits), L.dptr(pred), L.dptr(counts), L.stream_ptr())


def _confusion_counts(pred: torch.Tensor, gt: torch.Tensor, counts: torch.Tensor) -> None:
    L.call("dc_confusion_counts", pred.numel(), L.dptr(pred), L.dptr(gt), gt.element_size(), L.dptr(counts), L.stream_ptr())


def _optimizer_step(optimizer: int) -> None:
    opt = _OPTIMIZERS.get(optimizer)
    if opt is None:
        raise L.DeepcamHipError(f"deepcam custom op: unknown optimizer handle {optimizer}")
    opt._launch_kernels()


_SCHEMAS = {
    "net_forward": ("(Tensor x, int engine, bool train) -> Tensor", _net_forward),
    "net_backward": ("(Tensor dlogits, int engine) -> ()", _net_backward),
    "wce_fused": ("(Tensor logits, Tensor target, Tensor weight, float scale, Tensor(a!) loss_sum, Tensor(b!)? dlogits, Tensor(c!)? pred, "
                  "Tensor(d!)? counts) -> ()", _wce_fused),
    "confusion_counts": ("(Tensor pred, Tensor gt, Tensor(a!) counts) -> ()", _confusion_counts),
    "optimizer_step": ("(int optimizer) -> ()", _optimizer_step),
}

# one registration per process, whichever name the package was imported under (mlperf_deepcam_amd is an alias of this directory)
import sys as _sys

if getattr(_sys, "_deepcam_ops_library", None) is None:
    _LIB = torch.library.Library("deepcam", "DEF")
    for _name, (_schema, _fn) in _SCHEMAS.items():
        _LIB.define(_name + _schema)
        _LIB.impl(_name, _fn, "CompositeExplicitAutograd")
    _sys._deepcam_ops_library = _LIB             # keeps the registrations alive

OPS = torch.ops.deepcam
