"""The reference's Python call surface for the hot path, backed by the HIP engine.

Mirrors, name for name and argument for argument, what src/deepCam/train_hdf5_ddp.py calls in its step loop:

  architecture/deeplab_xception.py:398-465   DeepLabv3_plus(n_input, n_classes, os, pretrained, rank) / .forward
  utils/losses.py:28-52                      fp_loss(logit, target, weight, fpw_1, fpw_2)
  utils/utils.py:32-60                       compute_score(prediction, gt, num_classes, device_id)
  train_hdf5_ddp.py:213-218                  optim.Adam / optim.AdamW / apex FusedLAMB  -> Adam, AdamW, LAMB below
  utils/parsing_helpers.py:27-37             get_lr_schedule(start_lr, scheduler_arg, optimizer, last_step)
  train_hdf5_ddp.py:251-253                  GradualWarmupScheduler(optimizer, multiplier, total_epoch, after_scheduler)

so ``outputs = net.forward(inputs); loss = fp_loss(...); optimizer.zero_grad(); loss.backward(); optimizer.step()``
runs unmodified.  ``TrainStep`` is the same sequence without the autograd round trip (what bench.py and train.py use).
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import lib as L
from . import ops as _ops
from .engine import Engine

CLASS_FREQ = (0.986267818390377, 0.0004578708870701058, 0.01327431072255291)   # train_hdf5_ddp.py:206


def class_weights(loss_pow: float = -0.125) -> List[float]:
    return [f ** loss_pow for f in CLASS_FREQ]


# ------------------------------------------------------------------------------------------------------------------
# loss / metric
# ------------------------------------------------------------------------------------------------------------------
_cw_cache: Dict[tuple, torch.Tensor] = {}
_dlogits_for: Dict[int, torch.Tensor] = {}      # logits.data_ptr() -> engine-owned gradient buffer (zero-copy hand-over)


def _cw_tensor(weight: Sequence[float], device) -> torch.Tensor:
    key = (tuple(float(w) for w in weight), str(device))
    t = _cw_cache.get(key)
    if t is None:
        # losses.py:35 goes list -> numpy float64 -> float32
        t = torch.from_numpy(np.array(weight)).float().to(device)
        _cw_cache[key] = t
    return t


def wce_fused(logit: torch.Tensor, target: torch.Tensor, weight: Sequence[float], dlogits: Optional[torch.Tensor] = None,
              pred: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None,
              loss_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One pass over the logits: returns the fp64 loss SUM tensor [1]; optionally fills dlogits / pred / counts."""
    if logit.dim() != 4 or logit.size(1) != 3:
        raise L.DeepcamHipError("fp_loss: logits must be [B,3,H,W]")
    if logit.dtype != torch.float32 or not logit.is_contiguous():
        logit = logit.float().contiguous()
    B, _, H, W = logit.shape
    t = target.squeeze(1) if target.dim() == 4 else target
    if t.dtype not in (torch.uint8, torch.int32, torch.int64):
        t = t.long()
    if not t.is_contiguous():
        t = t.contiguous()
    if tuple(t.shape) != (B, H, W):
        raise L.DeepcamHipError(f"fp_loss: target shape {tuple(t.shape)} does not match logits {tuple(logit.shape)}")
    if loss_sum is None:
        loss_sum = torch.zeros(1, dtype=torch.float64, device=logit.device)
    _ops.OPS.wce_fused(logit, t, _cw_tensor(weight, logit.device), 1.0 / float(B * H * W), loss_sum, dlogits, pred, counts)
    return loss_sum


class _FpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, target, weight):
        dl = _dlogits_for.get(logit.data_ptr())
        if dl is None or dl.shape != logit.shape:
            dl = torch.empty_like(logit, dtype=torch.float32)
        s = wce_fused(logit.detach(), target, weight, dlogits=dl)
        ctx.dl = dl
        n = logit.numel() // logit.size(1)
        return (s / n).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, go):
        dl = ctx.dl
        dl.mul_(go)          # the kernel already produced d(mean)/d(logit); autograd's incoming scalar is usually 1
        return dl, None, None


def fp_loss(logit, target, weight, fpw_1=0, fpw_2=0):
    """mean over B*H*W of w[y]*CE(logit, y).  fpw_1 / fpw_2 are accepted and ignored exactly like the reference, whose
    false-positive masks are identically zero (losses.py:41,46)."""
    return _FpLoss.apply(logit, target, weight)


def iou_from_counts(counts) -> float:
    c = [int(v) for v in counts]
    ious = []
    for j in range(3):
        union = c[j] + c[3 + j] + c[6 + j]
        ious.append(np.float32(1.0) if union == 0 else np.float32(c[j]) / np.float32(union))   # utils.py:55-58
    return float(np.float32(ious[0] + ious[1] + ious[2]) / np.float32(3.0))


def compute_score(prediction, gt, num_classes=3, device_id=None, type="iou", weights=None):
    if num_classes != 3:
        raise L.DeepcamHipError("compute_score: the fused kernel is built for the 3 DeepCAM classes")
    pred = prediction if prediction.dtype == torch.int64 else prediction.long()
    pred = pred.contiguous()
    g = gt if gt.dtype in (torch.uint8, torch.int32, torch.int64) else gt.long()
    g = g.contiguous()
    counts = torch.zeros(9, dtype=torch.int64, device=pred.device)
    _ops.OPS.confusion_counts(pred, g, counts)
    return torch.tensor(iou_from_counts(counts.cpu().tolist()), dtype=torch.float32, device=pred.device)


# ------------------------------------------------------------------------------------------------------------------
# model
# ------------------------------------------------------------------------------------------------------------------
class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, module):
        eng = module.engine_for(x)
        ctx.eng, ctx.module = eng, module
        out = _ops.OPS.net_forward(x, _ops.engine_handle(eng), module.training)
        if module.training:
            if len(_dlogits_for) > 64:           # engines come and go with input shapes: drop entries of freed buffers
                _dlogits_for.clear()
            _dlogits_for[out.data_ptr()] = eng.dlogits
        return out

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.eng
        # torch semantics: a backward() that arrives without an optimizer.zero_grad() / step() since the previous one ACCUMULATES into p.grad.
        # The engine's backward program writes every parameter's gradient exactly once (it overwrites), so the gradients of the earlier
        # backward calls are set aside first and added back behind this one -- a 226 MB copy and add on the micro-batch path only; the
        # reference's loop (zero_grad before every backward, train_hdf5_ddp.py:358-364) never takes it.
        stash = None
        if eng.grads_live[0]:
            if getattr(ctx.module, "_grad_stash", None) is None:      # (one per module: its engines share the arena)
                ctx.module._grad_stash = torch.empty_like(eng.grads)
            stash = ctx.module._grad_stash
            stash.copy_(eng.grads)
        eng.grads_live[0] = True
        _ops.OPS.net_backward(dlogits, _ops.engine_handle(eng))
        red = ctx.module._ddp_reducer
        if red is not None:
            # under dist.DistributedDataParallel: wait for the bucketed all-reduces launched during backward, re-arm the
            # buckets and leave the AVERAGED gradient in the arena (train_hdf5_ddp.py:227,363).  A reducer that a fused TrainStep
            # has taken over averages inside the optimizer kernel: finish(average=True) refuses (the gradients would otherwise be
            # scaled twice, or -- had the reference been dropped -- the buckets launched by this backward would never be waited for)
            red.finish(average=True)
        if stash is not None:
            eng.grads.add_(stash)
        ctx.module._attach_grads()
        return None, None, None


class DeepLabv3_plus(torch.nn.Module):
    """Drop-in for architecture.deeplab_xception.DeepLabv3_plus (os=16, DeconvUpsampler).  Parameters are views of the
    engine's flat fp32 arena with the reference's names, so ``state_dict()`` interchanges with reference checkpoints."""

    def __init__(self, n_input=3, n_classes=21, os=16, pretrained=False, normalizer=None, _print=True, rank=0,
                 dtype=torch.bfloat16, seed=None):
        super().__init__()
        if os != 16:
            raise NotImplementedError                       # deeplab_xception.py:140-141,415-416 (train uses os=16)
        if pretrained:
            raise NotImplementedError("pretrained weights need a download (train_hdf5_ddp.py:199 passes False)")
        if _print and rank == 0:
            print("Constructing DeepLabv3+ model...")
            print("Number of output channels: {}".format(n_classes))
            print("Output stride: {}".format(os))
            print("Number of Input Channels: {}".format(n_input))
        self.n_input, self.n_classes, self.act_dtype = n_input, n_classes, dtype
        self._seed = seed
        self._engines: Dict[tuple, Engine] = {}
        self._primary: Optional[Engine] = None
        self._anchor = None
        self._ddp_reducer = None      # set by dist.DistributedDataParallel
        self._init_seed_state = torch.random.get_rng_state() if seed is None else None

    # -- engines are shape-specialised and built lazily; all of them share one parameter store
    def engine_for(self, x: torch.Tensor = None, shape: tuple = None) -> Engine:
        shp = tuple(x.shape) if x is not None else tuple(shape)
        B, Cin, H, W = shp
        key = (B, H, W)
        eng = self._engines.get(key)
        if eng is None:
            if self._primary is None:
                if self._seed is None and self._init_seed_state is not None:
                    # reproduce "torch.manual_seed(333); DeepLabv3_plus(...)": draw the weights from the RNG state
                    # that was current when the module was constructed
                    keep = torch.random.get_rng_state()
                    torch.random.set_rng_state(self._init_seed_state)
                    eng = Engine(B, H, W, self.act_dtype, n_input=self.n_input, n_classes=self.n_classes, seed=None)
                    torch.random.set_rng_state(keep)
                else:
                    eng = Engine(B, H, W, self.act_dtype, n_input=self.n_input, n_classes=self.n_classes, seed=self._seed)
                self._primary = eng
                self._register_arena(eng)
            else:
                eng = Engine(B, H, W, self.act_dtype, n_input=self.n_input, n_classes=self.n_classes, share_from=self._primary)
            self._engines[key] = eng
            if self._ddp_reducer is not None:
                self._ddp_reducer.hook(eng)       # every engine that shares the arena reports its gradients to the reducer
        return eng

    def materialize(self, batch: int, height: int, width: int) -> Engine:
        """Build the engine for a given input shape ahead of the first forward (so that .parameters() exists)."""
        return self.engine_for(shape=(batch, self.n_input, height, width))

    def _register_arena(self, eng: Engine) -> None:
        for name in eng.layout.state_keys:
            mod, _, leaf = name.rpartition(".")
            if name in eng.layout.params:
                t = torch.nn.Parameter(eng.param_view(name), requires_grad=True)
                self._flat_register(name, t, is_param=True)
            else:
                self._flat_register(name, eng.buffer_view(name), is_param=False)
        self._anchor = torch.zeros(1, device=eng.device, requires_grad=True)

    def _flat_register(self, name: str, t: torch.Tensor, is_param: bool) -> None:
        # nn.Module wants dotted names to be a module tree: build empty container modules along the path
        parts = name.split(".")
        m = self
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, torch.nn.Module())
            m = m._modules[p]
        if is_param:
            m.register_parameter(parts[-1], t)
        else:
            m.register_buffer(parts[-1], t)

    def _attach_grads(self) -> None:
        eng = self._primary
        for name, p in self.named_parameters():
            p.grad = eng.grad_view(name)

    def parameters(self, recurse: bool = True):
        if self._primary is None:
            raise L.DeepcamHipError("call net.materialize(batch, H, W) (or run one forward) before asking for parameters: "
                                    "the HIP engine is specialised to the input shape")
        return super().parameters(recurse)

    def to(self, *args, **kwargs):
        # parameters live in HIP memory from the start; .to(device) of the reference is a no-op here
        return self

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if input.dim() != 4 or input.size(1) != self.n_input:
            raise RuntimeError(f"expected input [B,{self.n_input},H,W], got {tuple(input.shape)}")
        if torch.is_grad_enabled() and self.training:
            self.engine_for(input)
            return _NetFn.apply(input, self._anchor, self)
        eng = self.engine_for(input)
        return _ops.OPS.net_forward(input, _ops.engine_handle(eng), self.training)

    def load_state_dict(self, state_dict, strict: bool = True):
        if self._primary is None:
            raise L.DeepcamHipError("materialize the model before load_state_dict")
        sd = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())
        r = super().load_state_dict(sd, strict=strict)
        self._primary.mark_weights_changed()
        return r

    @property
    def engine(self) -> Engine:
        return self._primary


# ------------------------------------------------------------------------------------------------------------------
# optimizers over the flat arena
# ------------------------------------------------------------------------------------------------------------------
class ArenaOptimizer:
    """torch.optim.Optimizer surface (param_groups, state_dict, load_state_dict, zero_grad, step) for the fused
    multi-tensor kernels.  state_dict() uses torch.optim.Adam's layout so checkpoints interchange with the reference."""

    kind = L.DC_ADAM

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, engine: Engine = None):
        params = list(params)
        if engine is None:
            raise L.DeepcamHipError("ArenaOptimizer needs the model's engine (use make_optimizer(net, ...))")
        self.engine = engine
        lay = engine.layout
        if len(params) != len(lay.params):
            raise L.DeepcamHipError("the optimizer must own all model parameters (one arena, one launch)")
        self.param_groups = [dict(params=params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)]
        dev = engine.device
        self.m = torch.zeros(lay.n_params, dtype=torch.float32, device=dev)
        self.v = torch.zeros(lay.n_params, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.max_grad_norm = max_grad_norm
        self.grad_scale = 1.0
        # learning rate (float32) and step count (int32) side by side: ONE 8-byte host-to-device copy per step
        self._scal_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._lr_host = self._scal_host[:1].view(torch.float32)
        self._step_host = self._scal_host[1:]
        self._scalars_copied = None
        self._scal_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        self.lr_dev = self._scal_dev[:1].view(torch.float32)
        self.step_dev = self._scal_dev[1:]
        self._offsets = torch.tensor(lay.offsets(), dtype=torch.int64, device=dev)
        # the LAMB workspace holds an n-element update direction (226 MB): only LAMB allocates it
        self._lamb_ws = (torch.zeros(L.load().dc_lamb_workspace_words(len(lay.params), lay.n_params), dtype=torch.float32, device=dev)
                         if self.kind == L.DC_LAMB else None)
        self.state = {}

    def zero_grad(self, set_to_none: bool = True):
        # every backward overwrites the whole gradient arena (each parameter exactly once): nothing to clear.  What this call does is end the
        # accumulation window: a backward() behind it starts from its own gradients, one that arrives WITHOUT it adds to the previous ones
        # (_NetFn.backward)
        self.engine.grads_live[0] = False
        return None

    def sync_scalars(self) -> None:
        """Push lr / step count to the device scalars the kernels read (keeps a captured hipGraph in step).
        The copies are asynchronous and the host enqueues steps ahead of the GPU, so the pinned source words must not be
        rewritten before the copy that reads them has run: wait for the previous copy's event first.  (Without the wait the
        device could pick up the step count of a LATER step -- wrong bias corrections, different from run to run.)  The host
        can still run one whole step ahead."""
        if self._scalars_copied is not None:
            self._scalars_copied.synchronize()
        self._lr_host[0] = float(self.param_groups[0]["lr"])
        self._step_host[0] = self.step_count
        self._scal_dev.copy_(self._scal_host, non_blocking=True)
        self._scalars_copied = torch.cuda.current_stream().record_event()

    def launch(self) -> None:
        """optimizer.step() through the dispatcher: torch.ops.deepcam.optimizer_step."""
        _ops.OPS.optimizer_step(_ops.optimizer_handle(self))

    def _launch_kernels(self) -> None:
        g = self.param_groups[0]
        eng = self.engine
        n = eng.layout.n_params
        b1, b2 = g["betas"]
        if self.kind == L.DC_LAMB:
            L.call("dc_lamb_step", len(eng.layout.params), L.dptr(self._offsets), n, L.dptr(eng.params), L.dptr(eng.grads),
                   L.dptr(self.m), L.dptr(self.v), L.dptr(self.lr_dev), b1, b2, g["eps"], g["weight_decay"], L.dptr(self.step_dev),
                   self.max_grad_norm, self.grad_scale, L.dptr(self._lamb_ws), L.stream_ptr())
        else:
            L.call("dc_adam_step", self.kind, n, L.dptr(eng.params), L.dptr(eng.grads), L.dptr(self.m), L.dptr(self.v),
                   L.dptr(self.lr_dev), b1, b2, g["eps"], g["weight_decay"], L.dptr(self.step_dev), self.grad_scale, L.stream_ptr())
        eng.mark_weights_changed()

    def step(self, closure=None):
        self.step_count += 1
        self.sync_scalars()
        self.launch()
        self.engine.grads_live[0] = False

    # -- torch.optim.Adam-compatible serialisation
    def state_dict(self):
        lay = self.engine.layout
        state = {}
        for i, p in enumerate(lay.params.values()):
            n = math.prod(p.shape)
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.m[p.offset:p.offset + n].view(p.shape).clone(),
                        "exp_avg_sq": self.v[p.offset:p.offset + n].view(p.shape).clone()}
        if self.step_count == 0:
            state = {}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(lay.params)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        lay = self.engine.layout
        g = sd["param_groups"][0]
        for k, v in g.items():
            if k != "params":
                self.param_groups[0][k] = v
        st = sd.get("state", {})
        self.step_count = 0
        for i, p in enumerate(lay.params.values()):
            s = st.get(i, st.get(str(i)))
            if s is None:
                continue
            n = math.prod(p.shape)
            self.m[p.offset:p.offset + n].copy_(s["exp_avg"].reshape(-1).to(self.m.device, torch.float32))
            self.v[p.offset:p.offset + n].copy_(s["exp_avg_sq"].reshape(-1).to(self.v.device, torch.float32))
            # torch.optim keeps 'step' per parameter; apex FusedLAMB keeps it in the param group
            self.step_count = int(float(s["step"])) if "step" in s else int(g.get("step", 0))


class Adam(ArenaOptimizer):
    kind = L.DC_ADAM


class AdamW(ArenaOptimizer):
    kind = L.DC_ADAMW


class LAMB(ArenaOptimizer):
    """apex.optimizers.FusedLAMB stand-in (train_hdf5_ddp.py:218).  Parity with apex is UNPINNED (apex is not vendored by
    the reference nor installable here); the definition is DESIGN.md's: global grad-norm clip at max_grad_norm, Adam
    moments with bias correction, decoupled weight decay inside the update, per-tensor trust ratio ||w||/||u||."""
    kind = L.DC_LAMB


def make_optimizer(name: str, net: DeepLabv3_plus, lr: float, eps: float, weight_decay: float) -> ArenaOptimizer:
    cls = {"Adam": Adam, "AdamW": AdamW, "LAMB": LAMB}.get(name)
    if cls is None:
        raise NotImplementedError("Error, optimizer {} not supported".format(name))          # train_hdf5_ddp.py:220
    return cls(net.parameters(), lr=lr, eps=eps, weight_decay=weight_decay, engine=net.engine)


# ------------------------------------------------------------------------------------------------------------------
# LR schedules
# ------------------------------------------------------------------------------------------------------------------
class MultiStepSchedule:
    """MultiStepLR exactly as the reference drives it: recursive (multiplies the optimizer's CURRENT lr by gamma when its
    counter lands on a milestone) and already stepped once by its constructor; see oracle/optim.py for the derivation."""

    def __init__(self, optimizer, milestones: Sequence[int], gamma: float, last_epoch: int = -1):
        self.optimizer, self.milestones, self.gamma = optimizer, list(milestones), gamma
        self.last_epoch = last_epoch
        self._last_lr = [g["lr"] for g in optimizer.param_groups]
        self.step()

    def get_last_lr(self):
        return self._last_lr

    def step(self):
        self.last_epoch += 1
        n = self.milestones.count(self.last_epoch)
        for g in self.optimizer.param_groups:
            if n:
                g["lr"] = g["lr"] * self.gamma ** n
        self._last_lr = [g["lr"] for g in self.optimizer.param_groups]


def get_lr_schedule(start_lr, scheduler_arg, optimizer, last_step=-1):
    optimizer.param_groups[0]["initial_lr"] = start_lr                     # parsing_helpers.py:29
    if scheduler_arg["type"] == "multistep":
        milestones = [int(x) for x in scheduler_arg["milestones"].split()]
        gamma = float(scheduler_arg["decay_rate"])
        return MultiStepSchedule(optimizer, milestones, gamma, last_epoch=last_step)
    raise ValueError("Error, scheduler type {} not supported.".format(scheduler_arg["type"]))


class GradualWarmupScheduler:
    """PARITY UNPINNED stand-in for ildoonet/pytorch-gradual-warmup-lr (train_hdf5_ddp.py:251-253): linear ramp of the lr
    from base to base*multiplier over total_epoch steps -- or, for multiplier == 1.0 (the reference's default
    --lr_warmup_factor), from 0 to base as that package's get_lr does -- then hands over to after_scheduler (whose lr is
    scaled by multiplier at the hand-over)."""

    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        if multiplier < 1.0:
            raise ValueError("multiplier should be greater thant or equal to 1.")
        self.optimizer, self.multiplier, self.total_epoch, self.after = optimizer, multiplier, total_epoch, after_scheduler
        self.base_lrs = [g.get("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = 0
        self.finished = False
        self._apply()

    def _apply(self):
        if self.last_epoch > self.total_epoch:
            if not self.finished:
                for g, b in zip(self.optimizer.param_groups, self.base_lrs):
                    g["lr"] = b * self.multiplier
                self.finished = True
            return
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            if self.multiplier == 1.0:
                g["lr"] = b * (float(self.last_epoch) / self.total_epoch)
            else:
                g["lr"] = b * ((self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0)

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self):
        self.last_epoch += 1
        if self.finished and self.after is not None:
            self.after.step()
        else:
            self._apply()


# ------------------------------------------------------------------------------------------------------------------
# fused train step (no autograd round trip)
# ------------------------------------------------------------------------------------------------------------------
class TrainStep:
    """inputs -> forward -> weighted CE (+argmax, IoU counts) -> backward -> [gradient hook] -> optimizer, as one flat
    launch sequence on the current stream.  Equivalent to train_hdf5_ddp.py:348-364."""

    def __init__(self, net: DeepLabv3_plus, optimizer: ArenaOptimizer, weight: Sequence[float], batch: int, height: int,
                 width: int, with_metrics: bool = False):
        self.net, self.opt = net, optimizer
        self.eng = net.engine_for(shape=(batch, net.n_input, height, width))
        self.weight = list(weight)
        dev = self.eng.device
        # loss sum (one double) and the nine confusion counts behind it in ONE 80-byte buffer: one memset per step clears both
        self._acc = torch.zeros(10, dtype=torch.int64, device=dev)
        self.loss_sum = self._acc[:1].view(torch.float64)
        self.with_metrics = with_metrics
        self.pred = torch.empty((batch, height, width), dtype=torch.int64, device=dev) if with_metrics else None
        self.counts = self._acc[1:] if with_metrics else None
        self.npix = batch * height * width
        self.after_backward = None      # set by attach_reducer
        self._eng_handle = _ops.engine_handle(self.eng)
        # loss inside the classifier's kernel (bf16 engine; DC_FUSE_HEAD_LOSS=0: the separate dc_wce_fused pass over stored logits);
        # store_logits: also write the fp32 NCHW logits (nobody reads them in the fused step)
        import os as _os
        self.fuse_head_loss = net.act_dtype == torch.bfloat16 and _os.environ.get("DC_FUSE_HEAD_LOSS", "1") != "0"
        self.store_logits = False

    def attach_reducer(self, reducer) -> None:
        """Data parallelism for the fused step: the reducer's buckets are all-reduced (SUM) while backward runs, the step
        waits for them before the optimizer and the 1/world averaging rides in the optimizer kernel (no extra pass)."""
        self.after_backward = reducer.finish
        self._reducer = reducer
        self.opt.grad_scale = 1.0 / reducer.world
        # one averaging mechanism per module: the autograd path (DistributedDataParallel -> finish(average=True)) is switched off
        reducer.averaging_in_optimizer = True
        reducer.hook(self.eng)
        # (net._ddp_reducer stays set: a loss.backward() through the wrapped module after this point reaches finish(average=True), which
        # raises on a reducer whose averaging lives in the optimizer, instead of silently leaving launched buckets un-waited)

    def launch(self, x: torch.Tensor, labels: torch.Tensor) -> None:
        eng = self.eng
        L.call("dc_memset_async", L.dptr(self._acc), 0, 80 if self.counts is not None else 8, L.stream_ptr())
        if self.fuse_head_loss:
            # the loss pass rides inside the classifier's forward kernel (dc_head_fwd_loss): the logits are neither stored nor re-read
            t = labels.squeeze(1) if labels.dim() == 4 else labels
            if t.dtype not in (torch.uint8, torch.int32, torch.int64):
                t = t.long()
            if not t.is_contiguous():
                t = t.contiguous()
            if tuple(t.shape) != (eng.B, eng.H, eng.W):
                raise L.DeepcamHipError(f"TrainStep: target shape {tuple(t.shape)} does not match the engine {(eng.B, eng.H, eng.W)}")
            eng.loss_args = {"labels": t, "weight": _cw_tensor(self.weight, eng.device), "grad_scale": 1.0 / float(self.npix),
                             "loss_sum": self.loss_sum, "pred": self.pred, "counts": self.counts, "store_logits": self.store_logits}
            try:
                _ops.OPS.net_forward(x, self._eng_handle, True)
            finally:
                eng.loss_args = None
        else:
            logits = _ops.OPS.net_forward(x, self._eng_handle, True)
            wce_fused(logits, labels, self.weight, dlogits=eng.dlogits, pred=self.pred, counts=self.counts, loss_sum=self.loss_sum)
        _ops.OPS.net_backward(eng.dlogits, self._eng_handle)
        if self.after_backward is not None:
            self.after_backward()
        self.opt.launch()

    def enable_graph(self) -> None:
        """Capture the whole step (about 1000 launches) into one hipGraph; later calls copy the batch into static
        buffers and replay.  Single-process only: the RCCL hand-off of the gradient reducer stays eager."""
        if self.after_backward is not None or self.eng.on_grad_ready is not None:
            raise L.DeepcamHipError("hipGraph capture is for the single-GPU step (the all-reduce path launches eagerly)")
        eng = self.eng
        self._gx = eng.x_static
        self._gy = torch.zeros((eng.B, eng.H, eng.W), dtype=torch.int64, device=eng.device)
        if eng.packed_version != eng.version[0]:
            eng.pack_weights()
        # weights change every step: repacking is part of the captured sequence
        eng.packed_version = -1
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.launch(self._gx, self._gy)              # warm-up on the side stream (lazy attribute setup, allocator)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eng.packed_version = -1
        with torch.cuda.graph(self._graph):
            self.launch(self._gx, self._gy)
        # the warm-up applied one optimizer update with whatever was in the static buffers (the capture pass records, it executes nothing):
        # callers enable the graph before training starts (bench) or accept one extra step on stale data
        self.graphed = True

    def enable_program(self) -> None:
        """Record the whole step as a C-side launch list (lib.Program, dc_program_*: every library call of one step with its arguments,
        stream fences included) and replay it with ONE call per step: later calls copy the batch into static buffers, push the
        optimizer's scalars and run the list -- no Python between two launches.  The recording pass is a real step on whatever the
        static buffers hold, as with enable_graph.  With a gradient reducer attached the step can be recorded when the reducer's collective
        is the library's own (GradReducer(collective="library"): dc_grad_allreduce_enqueue / _wait and the stream fences are C calls and
        land in the list); torch.distributed's collectives are Python objects and cannot."""
        red = getattr(self, "_reducer", None)
        if (self.after_backward is not None or self.eng.on_grad_ready is not None) and getattr(red, "collective", None) != "library":
            raise L.DeepcamHipError("the recorded launch list needs the library's own collective (GradReducer(collective='library')): "
                                    "torch.distributed's all-reduce hooks run in Python")
        eng = self.eng
        self._gx = eng.x_static
        self._gy = torch.zeros((eng.B, eng.H, eng.W), dtype=torch.int64, device=eng.device)
        eng.packed_version = -1           # weights change every step: repacking is part of the recorded sequence
        self.opt.step_count += 1
        self.opt.sync_scalars()
        self.launch(self._gx, self._gy)   # warm-up (lazy attribute setup); a real step, like the recording pass below
        torch.cuda.synchronize()
        eng.packed_version = -1
        self.opt.step_count += 1
        self.opt.sync_scalars()
        self._program = L.Program()
        self._program_stream = L.stream_ptr().value      # the recorded launches carry THIS stream; replay must run under the same one
        with self._program.recording():
            self.launch(self._gx, self._gy)
        eng.packed_version = -1
        self.programmed = True

    def __call__(self, x: torch.Tensor, labels: torch.Tensor) -> None:
        self.opt.step_count += 1
        self.opt.sync_scalars()
        if getattr(self, "programmed", False):
            # the batch copy and the optimizer's scalar hand-over above are enqueued on the CURRENT stream, the recorded launches on the stream
            # of the recording pass: under another current stream the step would read a stale or half-copied batch
            if L.stream_ptr().value != self._program_stream:
                raise L.DeepcamHipError("TrainStep: the recorded launch list was recorded under another stream; call the step under that stream "
                                        "or record again (enable_program)")
            self._gx.copy_(x, non_blocking=True)
            self._gy.copy_(labels.squeeze(1) if labels.dim() == 4 else labels, non_blocking=True)
            try:
                self._program.run()
            except L.DeepcamHipError:
                # a list that failed half way may have forked the weight-gradient stream without joining it: close the fork before raising
                if self.eng.use_side_stream:
                    L.call("dc_stream_fence", C.c_void_p(self.eng.side.cuda_stream), L.stream_ptr())
                raise
            self.eng.version[0] += 1
            return
        if getattr(self, "graphed", False):
            self._gx.copy_(x, non_blocking=True)
            self._gy.copy_(labels, non_blocking=True)
            self._graph.replay()
            self.eng.version[0] += 1
            return
        self.launch(x, labels)

    def loss(self) -> float:
        return float(self.loss_sum.item()) / self.npix

    def iou(self) -> float:
        return iou_from_counts(self.counts.cpu().tolist())
