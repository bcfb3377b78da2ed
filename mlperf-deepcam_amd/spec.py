"""Parameter/buffer layout of DeepLabV3+ (modified aligned Xception-65, os=16, DeconvUpsampler) as the reference
builds it (src/deepCam/architecture/deeplab_xception.py:125-193,282-296,347-374,398-439; chosen at
train_hdf5_ddp.py:197-200), expressed as a flat table so that

  * every parameter lives at a fixed offset of ONE fp32 arena (optimizer, gradient all-reduce and checkpoint
    views all index the same memory), in the reference's ``named_parameters()`` order, and
  * ``state_dict()`` keys / shapes / dtypes are those of the reference (checkpoint interchange,
    train_hdf5_ddp.py:515-527), including the int64 ``num_batches_tracked`` buffers.

The names are the reference's module paths; the structure here is a generated table, not a module tree.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, NamedTuple, Tuple

import torch


class SepSpec(NamedTuple):
    prefix: str       # "....rep.1" / "xception_features.conv3": holds .conv1 (depthwise) and .pointwise
    cin: int
    cout: int
    stride: int
    dil: int
    bn: str           # name of the BatchNorm that follows ("" if none)
    relu_after: bool  # the BN output is consumed through a ReLU


class BlockSpec(NamedTuple):
    name: str
    cin: int
    cout: int
    stride: int
    seps: Tuple[SepSpec, ...]
    skip: bool        # 1x1 conv + BN on the shortcut
    relu_out: bool    # the block output is only ever consumed through the next block's leading in-place ReLU


def _block(name: str, cin: int, cout: int, reps: int, stride: int = 1, start_with_relu: bool = True,
           grow_first: bool = True, is_last: bool = False) -> BlockSpec:
    """Mirror of Block.__init__'s rep list (deeplab_xception.py:80-109), resolved into separable-conv stages."""
    p = "xception_features." + name
    items: List[Tuple[str, int, int, int]] = []
    filters = cin
    if grow_first:
        items += [("relu", 0, 0, 0), ("sep", cin, cout, 1), ("bn", cout, 0, 0)]
        filters = cout
    for _ in range(reps - 1):
        items += [("relu", 0, 0, 0), ("sep", filters, filters, 1), ("bn", filters, 0, 0)]
    if not grow_first:
        items += [("relu", 0, 0, 0), ("sep", cin, cout, 1), ("bn", cout, 0, 0)]
    if not start_with_relu:
        items = items[1:]
    if stride != 1:
        items.append(("sep", cout, cout, 2))
    if stride == 1 and is_last:
        items.append(("sep", cout, cout, 1))
    seps = []
    for i, (kind, a, b, s) in enumerate(items):
        if kind != "sep":
            continue
        has_bn = i + 1 < len(items) and items[i + 1][0] == "bn"
        relu_after = has_bn and i + 2 < len(items) and items[i + 2][0] == "relu"
        seps.append(SepSpec(f"{p}.rep.{i}", a, b, s, 1, f"{p}.rep.{i + 1}" if has_bn else "", relu_after))
    return BlockSpec(name, cin, cout, stride, tuple(seps), cout != cin or stride != 1, name != "block20")


def blocks() -> List[BlockSpec]:
    b = [_block("block1", 64, 128, 2, stride=2, start_with_relu=False),
         _block("block2", 128, 256, 2, stride=2),
         _block("block3", 256, 728, 2, stride=2, is_last=True)]
    b += [_block(f"block{i}", 728, 728, 3) for i in range(4, 20)]
    b.append(_block("block20", 728, 1024, 2, grow_first=False, is_last=True))
    return b


EXIT_SEPS = (SepSpec("xception_features.conv3", 1024, 1536, 1, 2, "xception_features.bn3", True),
             SepSpec("xception_features.conv4", 1536, 1536, 1, 2, "xception_features.bn4", True),
             SepSpec("xception_features.conv5", 1536, 2048, 1, 2, "xception_features.bn5", True))
ASPP_RATES = (1, 6, 12, 18)


class ParamInfo(NamedTuple):
    name: str
    shape: Tuple[int, ...]
    offset: int         # element offset into the fp32 parameter / gradient arenas
    init: str           # conv_default | conv_kaiming | bias | ones | zeros


def _param_table(n_input: int, n_classes: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in registration order; kind: conv|convk (kaiming-normal re-init)|bias|bn."""
    t: List[Tuple[str, Tuple[int, ...], str]] = []
    X = "xception_features."

    def conv(name, cout, cin, k, kind="convk"):
        t.append((name + ".weight", (cout, cin, k, k), kind))

    def bn(name, c):
        t.append((name, (c,), "bn"))

    def sep(s: SepSpec, kind="convk"):
        conv(s.prefix + ".conv1", s.cin, 1, 3, kind)
        conv(s.prefix + ".pointwise", s.cout, s.cin, 1, kind)

    conv(X + "conv1", 32, n_input, 3); bn(X + "bn1", 32)
    conv(X + "conv2", 64, 32, 3); bn(X + "bn2", 64)
    for b in blocks():
        if b.skip:
            conv(X + b.name + ".skip", b.cout, b.cin, 1); bn(X + b.name + ".skipbn", b.cout)
        for s in b.seps:
            sep(s)
            if s.bn:
                bn(s.bn, s.cout)
    for s in EXIT_SEPS:
        sep(s)
        bn(s.bn, s.cout)
    for i, r in enumerate(ASPP_RATES, start=1):
        conv(f"aspp{i}.atrous_convolution", 256, 2048, 1 if r == 1 else 3, "convk_now"); bn(f"aspp{i}.bn", 256)
    conv("global_avg_pool.1", 256, 2048, 1, "conv"); bn("global_avg_pool.2", 256)
    conv("conv1", 256, 1280, 1, "conv"); bn("bn1", 256)
    conv("conv2", 48, 128, 1, "conv"); bn("bn2", 48)
    U = "upsample."
    t.append((U + "deconv1.0.weight", (256, 256, 3, 3), "conv")); bn(U + "deconv1.1", 256)
    t.append((U + "deconv2.0.weight", (256, 256, 3, 3), "conv")); bn(U + "deconv2.1", 256)
    conv(U + "conv1.0", 256, 304, 3, "conv"); bn(U + "conv1.1", 256)
    conv(U + "conv1.3", 256, 256, 3, "conv"); bn(U + "conv1.4", 256)
    conv(U + "conv1.6", 256, 256, 1, "conv_bias")
    t.append((U + "deconv3.0.weight", (256, 256, 3, 3), "conv")); bn(U + "deconv3.1", 256)
    t.append((U + "last_deconv.0.weight", (256, n_classes, 3, 3), "conv"))
    return t


def block_param_table(b: BlockSpec) -> List[Tuple[str, Tuple[int, ...], str]]:
    """The rows of _param_table that belong to one Block (same order): lets a single Block be built as an engine of its own
    (tests/test_block_gpu.py checks the HIP block program against the reference's Block known-answer vectors)."""
    X = "xception_features."
    t: List[Tuple[str, Tuple[int, ...], str]] = []
    if b.skip:
        t.append((X + b.name + ".skip.weight", (b.cout, b.cin, 1, 1), "convk"))
        t.append((X + b.name + ".skipbn", (b.cout,), "bn"))
    for s in b.seps:
        t.append((s.prefix + ".conv1.weight", (s.cin, 1, 3, 3), "convk"))
        t.append((s.prefix + ".pointwise.weight", (s.cout, s.cin, 1, 1), "convk"))
        if s.bn:
            t.append((s.bn, (s.cout,), "bn"))
    return t


class Layout:
    """Offsets of every parameter / buffer inside the flat arenas."""

    def __init__(self, n_input: int = 16, n_classes: int = 3, table=None):
        self.n_input, self.n_classes = n_input, n_classes
        self.table = table if table is not None else _param_table(n_input, n_classes)
        self.params: "OrderedDict[str, ParamInfo]" = OrderedDict()
        self.buffers: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()   # running_mean / running_var: name -> (offset, C)
        self.nbt: "OrderedDict[str, int]" = OrderedDict()                   # num_batches_tracked: name -> index
        self.state_keys: List[str] = []
        off = boff = 0
        for name, shape, kind in self.table:
            if kind == "bn":
                c = shape[0]
                for suffix, init in ((".weight", "ones"), (".bias", "zeros")):
                    self.params[name + suffix] = ParamInfo(name + suffix, (c,), off, init)
                    off += c
                self.buffers[name + ".running_mean"] = (boff, c); boff += c
                self.buffers[name + ".running_var"] = (boff, c); boff += c
                self.nbt[name + ".num_batches_tracked"] = len(self.nbt)
                self.state_keys += [name + s for s in (".weight", ".bias", ".running_mean", ".running_var", ".num_batches_tracked")]
            else:
                n = int(math.prod(shape))
                self.params[name] = ParamInfo(name, shape, off, kind)
                off += n
                self.state_keys.append(name)
                if kind == "conv_bias":
                    bname = name[:-len("weight")] + "bias"
                    self.params[bname] = ParamInfo(bname, (shape[0],), off, "bias:" + name)
                    off += shape[0]
                    self.state_keys.append(bname)
        self.n_params = off
        self.n_buffers = boff

    def offsets(self) -> List[int]:
        """Tensor boundaries of the parameter arena (LAMB trust ratios, gradient buckets)."""
        return [p.offset for p in self.params.values()] + [self.n_params]


def init_arena(layout: Layout, arena: torch.Tensor, seed: int | None = 333) -> None:
    """Fill a CPU fp32 arena with the reference's seed-333 initialisation, bit for bit.

    The reference draws every conv's default init (kaiming_uniform_(a=sqrt(5)), bias U(+-1/sqrt(fan_in))) at
    construction, then re-draws all Xception convs with kaiming_normal_ at the end of Xception.__init__
    (deeplab_xception.py:188-189,244-252) and each ASPP conv right after it is built (:296,304-312); the decoder and the
    other 1x1 convs keep the default draw because their __init_weight is never called.
    """
    assert arena.device.type == "cpu" and arena.dtype == torch.float32 and arena.numel() == layout.n_params
    if seed is not None:
        torch.manual_seed(seed)
    xcep: List[torch.Tensor] = []
    flushed = False

    def view(p: ParamInfo) -> torch.Tensor:
        return arena[p.offset:p.offset + int(math.prod(p.shape))].view(p.shape)

    for p in layout.params.values():
        v = view(p)
        if not flushed and not p.name.startswith("xception_features."):
            for w in xcep:
                torch.nn.init.kaiming_normal_(w)
            flushed = True
        if p.init == "ones":
            v.fill_(1.0)
        elif p.init == "zeros":
            v.zero_()
        elif p.init.startswith("bias:"):
            w = layout.params[p.init[5:]]
            fan_in = w.shape[1] * w.shape[2] * w.shape[3]
            torch.nn.init.uniform_(v, -1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        else:
            torch.nn.init.kaiming_uniform_(v, a=math.sqrt(5))
            if p.init == "convk":
                xcep.append(v)
            elif p.init == "convk_now":
                torch.nn.init.kaiming_normal_(v)
