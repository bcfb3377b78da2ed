"""Oracle: DeepLabV3+ / modified aligned Xception-65, functional CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference's architecture file, src/deepCam/architecture/deeplab_xception.py:
  * separable conv with explicit "same" padding          :45-66
  * residual Block incl. the shared in-place ReLU quirk  :69-122
  * Xception entry/middle/exit flow                      :125-242, init :244-252
  * ASPP modules                                         :282-312
  * DeconvUpsampler decoder (the one actually used)      :347-383, chosen at :438-439
  * DeepLabv3_plus wiring                                :398-465

It is written as a *table* (``layer_table``) plus a functional ``forward`` over a
flat ``{reference_state_dict_key: tensor}`` dict, so that the very same keys serve
as the checkpoint format (reference train_hdf5_ddp.py:515-527).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default, deeplab_xception.py:70 (normalizer default)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# layer table
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Conv:
    name: str            # state-dict prefix, weight key is name + ".weight"
    cin: int
    cout: int
    k: int = 1
    stride: int = 1
    pad: int = 0
    dil: int = 1
    groups: int = 1
    bias: bool = False
    transposed: bool = False   # ConvTranspose2d k3 s2 p1 op1
    init: str = "default"      # "default" (PyTorch reset_parameters) | "kaiming_normal"

    @property
    def wshape(self) -> Tuple[int, int, int, int]:
        if self.transposed:
            return (self.cin, self.cout, self.k, self.k)
        return (self.cout, self.cin // self.groups, self.k, self.k)


@dataclass(frozen=True)
class BN:
    name: str
    c: int


@dataclass(frozen=True)
class BlockSpec:
    """One residual unit, deeplab_xception.py:69-109."""
    name: str
    cin: int
    cout: int
    reps: int
    stride: int = 1
    start_with_relu: bool = True
    grow_first: bool = True
    is_last: bool = False

    def rep_items(self) -> List[Tuple[str, int, int, int]]:
        """[(kind, cin, cout, stride)] with kind in relu|sep|bn, in nn.Sequential order."""
        items: List[Tuple[str, int, int, int]] = []
        filters = self.cin
        if self.grow_first:
            items += [("relu", 0, 0, 0), ("sep", self.cin, self.cout, 1), ("bn", self.cout, 0, 0)]
            filters = self.cout
        for _ in range(self.reps - 1):
            items += [("relu", 0, 0, 0), ("sep", filters, filters, 1), ("bn", filters, 0, 0)]
        if not self.grow_first:
            items += [("relu", 0, 0, 0), ("sep", self.cin, self.cout, 1), ("bn", self.cout, 0, 0)]
        if not self.start_with_relu:
            items = items[1:]
        if self.stride != 1:
            items.append(("sep", self.cout, self.cout, 2))
        if self.stride == 1 and self.is_last:
            items.append(("sep", self.cout, self.cout, 1))
        return items

    @property
    def has_skip_conv(self) -> bool:
        return self.cout != self.cin or self.stride != 1


def xception_blocks() -> List[BlockSpec]:
    """deeplab_xception.py:152-177 with os=16 (train_hdf5_ddp.py:199)."""
    b = [BlockSpec("block1", 64, 128, 2, stride=2, start_with_relu=False),
         BlockSpec("block2", 128, 256, 2, stride=2),
         BlockSpec("block3", 256, 728, 2, stride=2, is_last=True)]
    b += [BlockSpec(f"block{i}", 728, 728, 3) for i in range(4, 20)]
    b.append(BlockSpec("block20", 728, 1024, 2, stride=1, grow_first=False, is_last=True))
    return b


def _sep_layers(prefix: str, cin: int, cout: int, stride: int, dil: int, init: str) -> List[Conv]:
    """SeparableConv2d_same = depthwise 3x3 (pad 0, explicit pre-pad) + pointwise 1x1, :54-60."""
    return [Conv(prefix + ".conv1", cin, cin, 3, stride, 0, dil, groups=cin, init=init),
            Conv(prefix + ".pointwise", cin, cout, 1, init=init)]


def layer_table(n_input: int = 16, n_classes: int = 3) -> List[object]:
    """All parameterised layers in the reference's module-registration order.

    The order matters twice: it is the state-dict / optimizer parameter order, and it is
    the order in which the reference draws random numbers at construction.
    """
    X = "xception_features."
    KN = "kaiming_normal"
    t: List[object] = [
        Conv(X + "conv1", n_input, 32, 3, 2, 1, init=KN), BN(X + "bn1", 32),
        Conv(X + "conv2", 32, 64, 3, 1, 1, init=KN), BN(X + "bn2", 64),
    ]
    for blk in xception_blocks():
        p = X + blk.name
        if blk.has_skip_conv:
            t += [Conv(p + ".skip", blk.cin, blk.cout, 1, blk.stride, init=KN), BN(p + ".skipbn", blk.cout)]
        for idx, (kind, a, b, s) in enumerate(blk.rep_items()):
            if kind == "sep":
                t += _sep_layers(f"{p}.rep.{idx}", a, b, s, 1, KN)
            elif kind == "bn":
                t.append(BN(f"{p}.rep.{idx}", a))
    for nm, ci, co in (("conv3", 1024, 1536), ("conv4", 1536, 1536), ("conv5", 1536, 2048)):
        t += _sep_layers(X + nm, ci, co, 1, 2, KN)
        t.append(BN(X + "bn" + nm[-1], co))
    # ASPP, :418-421 (rates 1, 6, 12, 18)
    for i, rate in enumerate((1, 6, 12, 18), start=1):
        k, pad = (1, 0) if rate == 1 else (3, rate)
        t += [Conv(f"aspp{i}.atrous_convolution", 2048, 256, k, 1, pad, rate, init=KN), BN(f"aspp{i}.bn", 256)]
    t += [Conv("global_avg_pool.1", 2048, 256, 1), BN("global_avg_pool.2", 256),
          Conv("conv1", 1280, 256, 1), BN("bn1", 256),
          Conv("conv2", 128, 48, 1), BN("bn2", 48)]
    U = "upsample."
    t += [Conv(U + "deconv1.0", 256, 256, 3, 2, 1, transposed=True), BN(U + "deconv1.1", 256),
          Conv(U + "deconv2.0", 256, 256, 3, 2, 1, transposed=True), BN(U + "deconv2.1", 256),
          Conv(U + "conv1.0", 304, 256, 3, 1, 1), BN(U + "conv1.1", 256),
          Conv(U + "conv1.3", 256, 256, 3, 1, 1), BN(U + "conv1.4", 256),
          Conv(U + "conv1.6", 256, 256, 1, bias=True),
          Conv(U + "deconv3.0", 256, 256, 3, 2, 1, transposed=True), BN(U + "deconv3.1", 256),
          Conv(U + "last_deconv.0", 256, n_classes, 3, 2, 1, transposed=True)]
    return t


# --------------------------------------------------------------------------------------
# initialisation (reproduces torch.manual_seed(333) + reference constructors bit for bit)
# --------------------------------------------------------------------------------------
def _default_conv_init(w: torch.Tensor, b: torch.Tensor | None) -> None:
    # nn.Conv2d / nn.ConvTranspose2d.reset_parameters: kaiming_uniform_(a=sqrt(5)), bias U(-1/sqrt(fan_in), ..)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    if b is not None:
        fan_in = w.size(1) * w.size(2) * w.size(3)
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        torch.nn.init.uniform_(b, -bound, bound)


def init_state(seed: int | None = 333, n_input: int = 16, n_classes: int = 3) -> "OrderedDict[str, torch.Tensor]":
    """Fresh state dict (params + BN buffers) equal to the reference model built after
    ``torch.manual_seed(seed)`` (train_hdf5_ddp.py:113-117,197-200).

    Random-number order: every conv draws its default init at construction; then
    Xception.__init_weight (:244-252) redraws *all* Xception convs with kaiming_normal_
    (fan_in, gain sqrt(2)) in module order; each ASPP module does the same for its own conv
    right after it is constructed (:296,304-312).  Everything else keeps the default draw
    because DeepLabv3_plus.__init_weight / DeconvUpsampler.__init_weight are never called.
    """
    if seed is not None:
        torch.manual_seed(seed)
    table = layer_table(n_input, n_classes)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    xception_convs: List[Conv] = []

    def add_conv(c: Conv) -> None:
        w = torch.empty(c.wshape)
        b = torch.empty(c.cout) if c.bias else None
        _default_conv_init(w, b)
        sd[c.name + ".weight"] = w
        if b is not None:
            sd[c.name + ".bias"] = b

    def add_bn(b: BN) -> None:
        sd[b.name + ".weight"] = torch.ones(b.c)
        sd[b.name + ".bias"] = torch.zeros(b.c)
        sd[b.name + ".running_mean"] = torch.zeros(b.c)
        sd[b.name + ".running_var"] = torch.ones(b.c)
        sd[b.name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    in_xception = True
    for layer in table:
        if in_xception and not layer.name.startswith("xception_features."):
            # end of Xception.__init__: __init_weight over all of its convs
            for c in xception_convs:
                torch.nn.init.kaiming_normal_(sd[c.name + ".weight"])
            in_xception = False
        if isinstance(layer, Conv):
            add_conv(layer)
            if in_xception:
                xception_convs.append(layer)
            elif layer.init == "kaiming_normal":      # ASPP conv: re-drawn immediately
                torch.nn.init.kaiming_normal_(sd[layer.name + ".weight"])
        else:
            add_bn(layer)
    return sd


def param_keys(sd: Dict[str, torch.Tensor]) -> List[str]:
    """Keys that are nn.Parameters (optimizer order), i.e. not BN buffers."""
    return [k for k in sd if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


# --------------------------------------------------------------------------------------
# functional forward
# --------------------------------------------------------------------------------------
class _Ctx:
    def __init__(self, sd, training: bool, update_stats: bool):
        self.sd, self.training, self.update_stats = sd, training, update_stats

    def bn(self, x: torch.Tensor, name: str) -> torch.Tensor:
        sd = self.sd
        if self.training:
            if x.numel() // x.size(1) <= 1:
                # mirrors F.batch_norm's "Expected more than 1 value per channel" 
                raise ValueError("Expected more than 1 value per channel when training, got input size {}".format(tuple(x.shape)))
            dims = (0, 2, 3)
            mean = x.mean(dims)
            var = x.var(dims, unbiased=False)
            if self.update_stats:
                with torch.no_grad():
                    n = x.numel() / x.size(1)
                    sd[name + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(mean.detach(), alpha=BN_MOMENTUM)
                    sd[name + ".running_var"].mul_(1 - BN_MOMENTUM).add_(var.detach() * (n / (n - 1)), alpha=BN_MOMENTUM)
                    sd[name + ".num_batches_tracked"].add_(1)
        else:
            mean, var = sd[name + ".running_mean"], sd[name + ".running_var"]
        inv = torch.rsqrt(var + BN_EPS)
        g, b = sd[name + ".weight"], sd[name + ".bias"]
        return (x - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]

    def sep(self, x: torch.Tensor, prefix: str, stride: int, dil: int) -> torch.Tensor:
        # fixed_padding(k=3, rate=dil): pad dil on every side (:45-51), then dw 3x3 pad 0, then pw 1x1
        x = F.pad(x, (dil, dil, dil, dil))
        x = F.conv2d(x, self.sd[prefix + ".conv1.weight"], None, stride, 0, dil, groups=x.size(1))
        return F.conv2d(x, self.sd[prefix + ".pointwise.weight"])

    def block(self, blk: BlockSpec, inp: torch.Tensor, prefix: str) -> torch.Tensor:
        items = blk.rep_items()
        # The leading ReLU is in-place on the block input (:79,84), so the skip path sees relu(inp).
        if items[0][0] == "relu":
            inp = F.relu(inp)
        x = inp
        for idx, (kind, a, b, s) in enumerate(items):
            if kind == "relu":
                if idx > 0:
                    x = F.relu(x)
            elif kind == "sep":
                x = self.sep(x, f"{prefix}.rep.{idx}", s, 1)
            else:
                x = self.bn(x, f"{prefix}.rep.{idx}")
        if blk.has_skip_conv:
            skip = F.conv2d(inp, self.sd[prefix + ".skip.weight"], None, blk.stride)
            skip = self.bn(skip, prefix + ".skipbn")
        else:
            skip = inp
        return x + skip


def forward(sd: Dict[str, torch.Tensor], inputs: torch.Tensor, training: bool = True,
            update_stats: bool = True, return_intermediates: bool = False):
    """logits[B,3,H,W] = DeepLabv3_plus.forward(inputs[B,16,H,W])  (:441-465)."""
    c = _Ctx(sd, training, update_stats)
    X = "xception_features."
    inter = {}
    x = F.conv2d(inputs, sd[X + "conv1.weight"], None, 2, 1)
    x = F.relu(c.bn(x, X + "bn1"))
    x = F.conv2d(x, sd[X + "conv2.weight"], None, 1, 1)
    x = F.relu(c.bn(x, X + "bn2"))
    low = None
    for blk in xception_blocks():
        x = c.block(blk, x, X + blk.name)
        if blk.name == "block1":
            # low_level_feat aliases block1's output, which block2's in-place ReLU then mutates (:205-207)
            low = F.relu(x)
    # no ReLU between block20 and conv3 (:229-230)
    for i in (3, 4, 5):
        x = F.relu(c.bn(c.sep(x, f"{X}conv{i}", 1, 2), f"{X}bn{i}"))
    inter["encoder_out"] = x
    inter["low_level"] = low

    branches = []
    for i, rate in enumerate((1, 6, 12, 18), start=1):
        w = sd[f"aspp{i}.atrous_convolution.weight"]
        y = F.conv2d(x, w) if rate == 1 else F.conv2d(x, w, None, 1, rate, rate)
        branches.append(F.relu(c.bn(y, f"aspp{i}.bn")))
    g = x.mean((2, 3), keepdim=True)                               # AdaptiveAvgPool2d((1,1))
    g = F.relu(c.bn(F.conv2d(g, sd["global_avg_pool.1.weight"]), "global_avg_pool.2"))
    # bilinear, align_corners=True from a 1x1 source is a constant broadcast (:450)
    branches.append(g.expand(-1, -1, x.size(2), x.size(3)))
    x = torch.cat(branches, 1)
    x = F.relu(c.bn(F.conv2d(x, sd["conv1.weight"]), "bn1"))
    low = F.relu(c.bn(F.conv2d(low, sd["conv2.weight"]), "bn2"))
    inter["aspp_out"] = x

    U = "upsample."

    def deconv(t, name):
        return F.conv_transpose2d(t, sd[U + name + ".weight"], None, 2, 1, 1)

    x = F.relu(c.bn(deconv(x, "deconv1.0"), U + "deconv1.1"))
    x = F.relu(c.bn(deconv(x, "deconv2.0"), U + "deconv2.1"))
    x = torch.cat((x, low), 1)
    x = F.relu(c.bn(F.conv2d(x, sd[U + "conv1.0.weight"], None, 1, 1), U + "conv1.1"))
    x = F.relu(c.bn(F.conv2d(x, sd[U + "conv1.3.weight"], None, 1, 1), U + "conv1.4"))
    x = F.conv2d(x, sd[U + "conv1.6.weight"], sd[U + "conv1.6.bias"])
    x = F.relu(c.bn(deconv(x, "deconv3.0"), U + "deconv3.1"))
    x = deconv(x, "last_deconv.0")
    if return_intermediates:
        return x, inter
    return x
