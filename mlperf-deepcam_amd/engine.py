"""Explicit forward / backward program of the DeepCAM train step over preallocated channels-last buffers.

No tracing compiler and no autograd graph: the network is a static list of kernel launches (``fwd`` program) and a
hand-written reverse list (``bwd`` program) over buffers whose shapes are fixed at construction, so the same call
sequence can be replayed eagerly or captured once into a hipGraph.  Every arithmetic step is a call into
libdeepcam_hip.so (mlperf-deepcam_amd/lib.py); torch is used for device memory and streams only.

Data layout in HBM
  * activations / activation gradients: NHWC, dtype T (bf16 or fp32), concat inputs are channel slices of one buffer
  * parameters, gradients, Adam/LAMB moments: three flat fp32 arenas in the reference's parameter order
  * GEMM weights: packed per step from the fp32 master into T, once in forward ([tap][cout][cin]) and once in
    data-gradient ([tap][cin][cout]) orientation
  * BatchNorm: fp32 per-tile partial statistics slabs, fp32 scale/shift/mean/invstd vectors, fp32 running buffers

Reference: DeepLabv3_plus.forward (architecture/deeplab_xception.py:441-465) and everything it calls; the backward is
what ``loss.backward()`` (train_hdf5_ddp.py:363) would produce for it.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Callable, Dict, List, Optional

import torch

from . import lib as L
from . import spec as S

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Act:
    """NHWC activation view: element (n,h,w,c) at base[((n*H+h)*W+w)*ld + off + c]."""

    def __init__(self, eng: "Engine", name: str, N: int, H: int, W: int, Cc: int, parent: "Act" = None, off: int = 0,
                 dtype=None):
        self.eng, self.name, self.N, self.H, self.W, self.C = eng, name, N, H, W, Cc
        self.parent, self.off = parent, off
        if parent is None:
            # pixel stride rounded up to 64 elements: every pixel row of a bf16 tensor then starts on a 128-byte boundary (728 -> 768), so a
            # 128-byte K piece of a row is ONE L2 line for the LDS-DMA loaders of the GEMM kernels (scripts/fill_bench.hip: 15.3 TB/s from
            # line-aligned rows, 12.6 from 64-byte boundaries (736, the stride up to round 4), 10.9 from 728); the pad channels are never
            # read or written.  DC_ACT_ALIGN=32: the previous stride
            al = eng.act_align
            self.ld = Cc if Cc < 64 else (Cc + al - 1) // al * al
            self.buf = torch.empty((N, H, W, self.ld), dtype=dtype or eng.dtype, device=eng.device)
            eng.act_bytes += self.buf.numel() * self.buf.element_size()
            eng.saved[name] = self.buf
            eng.saved_channels[name] = Cc      # (the buffer's last dimension is ld >= Cc; the pad channels hold whatever the allocator left)
        else:
            assert (parent.N, parent.H, parent.W) == (N, H, W) and off + Cc <= parent.C
            self.buf, self.ld = parent.buf, parent.ld
            self.off = parent.off + off
        self._grad: Optional[Act] = None
        self.grad_init = False      # set while the backward program is built: has some op written the gradient yet?
        self.grad_takes = 0         # consumers that have registered their contribution to the gradient so far (execution order)

    @property
    def M(self) -> int:
        return self.N * self.H * self.W

    @property
    def ptr(self) -> C.c_void_p:
        return C.c_void_p(self.buf.data_ptr() + self.off * self.buf.element_size())

    def slice(self, name: str, off: int, Cc: int) -> "Act":
        return Act(self.eng, name, self.N, self.H, self.W, Cc, parent=self, off=off)

    def view(self) -> torch.Tensor:
        return self.buf.view(self.N, self.H, self.W, self.ld)[..., self.off:self.off + self.C]

    @property
    def grad(self) -> "Act":
        if self._grad is None:
            if self.parent is not None:
                self._grad = Act(self.eng, "d" + self.name, self.N, self.H, self.W, self.C, parent=self.parent.grad,
                                 off=self.off - self.parent.off)
            else:
                self._grad = Act(self.eng, "d" + self.name, self.N, self.H, self.W, self.C, dtype=self.buf.dtype)
        return self._grad

    def take_grad_mode(self) -> int:
        """0 = this op is the first writer of the gradient, 1 = accumulate (read-modify-write).
        Called while the backward program is resolved, in execution order."""
        a, init = self, False
        while a is not None:
            init = init or a.grad_init
            a = a.parent
        self.grad_init = True
        self.grad_takes += 1
        return 1 if init else 0


class LazyAct:
    """A BatchNorm(+ReLU) output that is never stored: its only consumer is a depthwise conv, which applies
    act(y*scale + shift) while loading the raw conv output y.  Only the GRADIENT w.r.t. this activation is a real buffer."""

    def __init__(self, y: Act, scale: torch.Tensor, shift: torch.Tensor, relu: bool, name: str):
        self.y, self.scale, self.shift, self.relu, self.name = y, scale, shift, relu, name
        self.N, self.H, self.W, self.C = y.N, y.H, y.W, y.C
        self._grad: Optional[Act] = None
        self.grad_init = False

    @property
    def M(self) -> int:
        return self.y.M

    @property
    def grad(self) -> Act:
        if self._grad is None:
            self._grad = Act(self.y.eng, "d" + self.name, self.N, self.H, self.W, self.C, dtype=self.y.buf.dtype)
        return self._grad

    def take_grad_mode(self) -> int:
        mode = 1 if self.grad_init else 0
        self.grad_init = True
        return mode


class Engine:
    def __init__(self, batch: int, height: int, width: int, dtype=torch.bfloat16, device=None, n_input: int = 16,
                 n_classes: int = 3, seed: Optional[int] = 333, share_from: "Engine" = None, layout: "S.Layout" = None,
                 builder: Callable[["Engine"], None] = None):
        """layout + builder: a sub-network as an engine of its own (tests build ONE Xception Block from the op builders below
        and run it against the reference's Block vectors); the default is the whole DeepLabV3+ network."""
        if not torch.cuda.is_available():
            raise L.DeepcamHipError("mlperf_deepcam_amd.Engine needs a HIP device: there is no CPU path")
        L.load()
        assert n_classes == 3, "the fused loss / head kernels are built for the 3 DeepCAM classes"
        assert height % 16 == 0 and width % 16 == 0, "input extents must be multiples of the output stride (16)"
        self.B, self.H, self.W = batch, height, width
        self.dtype, self.dt = dtype, L.dtype_code(dtype)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.n_input, self.n_classes = n_input, n_classes
        self.layout = layout if layout is not None else S.Layout(n_input, n_classes)
        self._builder = builder
        lay = self.layout
        self.act_bytes = 0
        self.act_align = int(os.environ.get("DC_ACT_ALIGN", "64"))      # pixel stride of the activation tensors in elements (Act)
        assert self.act_align in (32, 64, 128), "DC_ACT_ALIGN: 32, 64 or 128"
        # name -> tensor of everything forward leaves behind for backward (activations, BatchNorm coefficient vectors): lets a
        # test put two engines on ONE linearisation point (tests/test_model_gpu.py::test_backward_parity_at_shared_activations)
        self.saved: Dict[str, torch.Tensor] = {}
        self.saved_channels: Dict[str, int] = {}

        # ---- flat arenas (shared between engines of different batch shape, e.g. train B=2 and validation B=1) ----
        if share_from is not None:
            assert share_from.layout.n_params == lay.n_params and share_from.device == self.device
            self.params, self.grads, self.buffers, self.nbt = share_from.params, share_from.grads, share_from.buffers, share_from.nbt
            self.version = share_from.version
            self.grads_live = share_from.grads_live
        else:
            host = torch.empty(lay.n_params, dtype=torch.float32)
            if layout is None:
                S.init_arena(lay, host, seed)
            else:
                host.zero_()                  # a custom layout's owner fills the arena (engine.params) itself
            self.params = host.to(self.device)
            self.grads = torch.zeros(lay.n_params, dtype=torch.float32, device=self.device)
            hb = torch.zeros(lay.n_buffers, dtype=torch.float32)
            for name, (off, c) in lay.buffers.items():
                if name.endswith("running_var"):
                    hb[off:off + c] = 1.0
            self.buffers = hb.to(self.device)
            self.nbt = torch.zeros(len(lay.nbt), dtype=torch.int64, device=self.device)
            self.version = [0]          # bumped whenever the master weights change (optimizer step, load_state_dict)
            # [True] between a loss.backward() through the module and the next optimizer.zero_grad() / step(): a second backward in that
            # window OVERWRITES the gradients (the engine never accumulates), which nn._NetFn.backward then says out loud
            self.grads_live = [False]
        self.packed_version = -1
        self._keep: List[torch.Tensor] = []
        # Fusing the BatchNorm(+ReLU) into the consuming depthwise conv is implemented and tested, but measured slower at B=8
        # (55.0 vs 53.8 ms/step: the 9-tap stencil becomes VALU-heavy), so it is off by default.
        self.fuse_bn_reduce = os.environ.get("DC_FUSE_BN_REDUCE", "1") != "0"   # BN backward sums taken in the consumer dw data-gradient kernel
        self.fuse_bn_into_dw = os.environ.get("DC_FUSE_BN_DW", "1") != "0"   # BN(+ReLU) applied in the consumer depthwise kernel's LDS tile
        self.mask_from_y = os.environ.get("DC_MASK_FROM_Y", "1") != "0"
        # BatchNorm backward sums taken in the epilogue of the consuming dense conv's data gradient (dc_conv_dgrad_bnstats /
        # dc_head_bwd_bnstats) instead of a dc_bn_bwd_reduce pass over the gradient and the BatchNorm input
        # (time-neutral at local batch 8 and 4, where it saves 2.8 GB of reads per step; round 4, one job at local batch 2: 13.99 -> 13.95 ms)
        self.fuse_bn_conv = os.environ.get("DC_FUSE_BN_CONV", "1") != "0"
        self.fuse_bn_head = os.environ.get("DC_FUSE_BN_HEAD", "1") != "0"
        # the BatchNorm + ReLU in front of the classifier head (upsample.deconv3.1/.2) is applied by the head itself while it loads
        # (dc_head_fwd_loss_bnin / dc_head_bwd_bnin): the 256-channel 384 x 576 activation is never stored (at local batch 8: a 0.35 ms
        # dc_bn_apply pass and 1.8 GB of traffic less per step).  bf16 only (the fused head kernel).
        self.fuse_bn_into_head = dtype == torch.bfloat16 and os.environ.get("DC_FUSE_BN_INTO_HEAD", "1") != "0"
        # the head's data gradient (906 MB at local batch 8, read once) is not stored: the head's kernel takes the BatchNorm sums in a first
        # pass and writes the BatchNorm input's gradient itself in a second one (dc_head_bwd_bnin_apply)
        self.fuse_head_apply = os.environ.get("DC_FUSE_HEAD_APPLY", "1") != "0"
        self.inline_last_wgrad = os.environ.get("DC_INLINE_LAST_WGRAD", "1") != "0"
        self.pack_side = os.environ.get("DC_PACK_SIDE", "1") != "0"       # the batch's layout pass beside the weight repack
        self._layout_forked = False
        self.aspp_side = os.environ.get("DC_ASPP_SIDE", "1") != "0"       # forward: the small ASPP branches beside the grouped atrous launch
        # depthwise weight gradient taken inside the depthwise data gradient (dc_dwconv_dgrad_bnstats_wgrad) where the layer's input is a
        # never-stored BatchNorm output: the separate dc_dwconv_wgrad launch (and its second read of dy and y) disappears.  On the tiled
        # kernel the fusion held 232 registers (two workgroups per CU instead of three) and paid from local batch 8 only; on the persistent
        # pipelined kernel (csrc/dwpipe.hip, one workgroup per CU by design) it pays everywhere: one job in round 4, local batch 2
        # 14.36 -> 13.99 ms, batch 4 21.92 -> 21.42 ms
        self.fuse_dw_wgrad = os.environ.get("DC_FUSE_DW_WGRAD", "1") != "0"
        # the backward sums of a block-output BatchNorm (relu(bn(y) + residual)) taken by the next block's first depthwise data gradient,
        # the last writer of that output's gradient (dc_dwconv_dgrad_wgrad_bnres) instead of a dc_bn_bwd_reduce pass over three tensors
        self.fuse_bn_res = os.environ.get("DC_FUSE_BN_RES", "1") != "0"
        self.fuse_bn_bwd_fin = os.environ.get("DC_FUSE_BN_BWD_FIN", "1") != "0"    # dc_bn_bwd_finalize inside dc_bn_bwd_apply for short slabs
        # dc_bn_finalize inside the kernel that consumes the coefficients (dc_dwconv_fwd_fin, dc_bn_apply_fin) for short slabs of small tensors
        self.fuse_bn_fwd_fin = os.environ.get("DC_FUSE_BN_FWD_FIN", "1") != "0"
        self.bn_fin_max_m = int(os.environ.get("DC_BN_FIN_MAX_M", "16384"))          # ... up to this many pixels (each workgroup repeats the sum)
        # timing experiments only: "fwd" / "bwd" / "both" leave out the BatchNorm finalize launches (results are then garbage): what the
        # 77 + 78 tiny kernels and the dispatch gaps around them cost the chain
        self._debug_skip_finalize = os.environ.get("DC_DEBUG_SKIP_BN_FINALIZE", "")
        # BatchNorm backward apply + pointwise data gradient + pointwise weight gradient of the entry flow's thin layers in one pass (dc_pw_bn_bwd)
        self.fuse_pw_bn_bwd = os.environ.get("DC_FUSE_PW_BN_BWD", "1") != "0"
        self.fuse_head_wgrad = os.environ.get("DC_FUSE_HEAD_WGRAD", "1") != "0"
        self.fuse_bn_src_dw = os.environ.get("DC_FUSE_BN_SRC_DW", "1") != "0"       # a stored BatchNorm output's last depthwise reader takes its backward sums
        # BatchNorm sums as ONE fp64 row per layer that the producing launch adds to (dc_conv_sum_row_kn; include/deepcam_hip.h) instead of a row
        # per workgroup: the consumer of the coefficients reads two numbers per channel and runs the finalize itself at any tensor size, so the
        # 6 us finalize launch leaves the dependent chain.  The rows live in one arena that a single memset zeroes at the start of a train step.
        self.bn_sum_rows = os.environ.get("DC_BN_SUM_ROWS", "1") != "0"
        self._sum_arena = None
        self._sum_used = 0
        self.conv_kn = os.environ.get("DC_CONV_KN", "1") != "0"                    # pointwise layers pass both packed weight images (dc_conv_fwd_kn / dc_conv_dgrad_kn)
        self.fuse_sep_fwd = os.environ.get("DC_FUSE_SEP_FWD", "1") != "0"           # depthwise + pointwise forward of the thin layers in one kernel
        # layers per grouped weight-gradient launch (dc_conv_wgrad_group).  Measured with both streams (scripts/ab_step.py): local
        # batch 2: 16.07 -> 15.42 ms/step with groups of 3, batch 4: 24.50 -> 24.11, batch 8: 40.31 -> 40.10 (round 3; 1 / 2 / 3 / 4 layers:
        # 40.31 / 40.28 / 40.10 / 40.18) -- the fp32 split slabs (256 KiB per workgroup whatever the batch) are a third of a 728-channel
        # weight gradient's time at batch 2, and a group of three writes and re-reads a third of them.
        wg = os.environ.get("DC_WGRAD_GROUP", "auto")
        # Round 5: the pointwise layers run on 256 x 384 tiles (csrc/wgrad384.hip), which take up to 16 layers per launch; groups of 3 / 4 / 6 / 12:
        # local batch 8 35.35 / 35.20 / 35.02 / 34.95 ms, batch 2 13.27 / - / 13.09 / 13.14, batch 4 20.51 / - / 20.40 / 20.34 (one job each,
        # profiles/r05_ab_wgrad384.txt); the 256 x 256 kernel keeps at most four.  Twelve: equal in time to six, and the 728-channel layers are cut
        # into 2 splits instead of 5 (4.7 MB of fp32 slab per layer written and read back by the fold instead of 11.8 MB)
        self.wgrad_group = 12 if wg == "auto" else max(1, min(16, int(wg)))
        self._wg_recs: List[dict] = []
        # Weight-gradient partial sums stay in per-layer slabs and are folded by dc_fold_slabs (csrc/fold.hip): one fold per dense
        # weight-gradient launch, which also takes the rows that the depthwise data-gradient kernels since the previous fold have left.
        # _fold_seq lists, in execution order, ("dw", FoldEntry, name, ready-list) and ("dense", rec); _plan_folds() turns it into
        # static per-launch tables once the backward program is resolved.
        self._fold_seq: List[tuple] = []
        self._final_fold = None                       # (FoldEntry array, count, names): rows produced after the last dense launch
        self.aspp_group = os.environ.get("DC_ASPP_GROUP", "1") != "0"      # atrous ASPP branches in one forward launch
        self._deferred_names: List[str] = []
        self._ready_now: List[str] = []

        # ---- program containers -----------------------------------------------------------------------------
        self.fwd_train: List[Callable[[], None]] = []
        self.fwd_eval: List[Callable[[], None]] = []
        self.bwd: List[Callable[[], None]] = []      # appended in forward order, executed reversed
        self.pack_ops: List[Callable[[], None]] = []
        self._pack_entries: List[L.PackEntry] = []
        self.grad_ready: List[List[str]] = []        # per bwd entry: parameter names whose gradient is final after it
        self.ws_bytes = 0
        self._ws_users: List[Callable[[], int]] = []
        self.x_in: Optional[torch.Tensor] = None     # caller's NCHW fp32 batch (set per call)
        self.x0: Optional[Act] = None                # NHWC stem input (when the stem runs on the MFMA kernels)
        self.x_static = torch.empty((batch, n_input, height, width), dtype=torch.float32, device=self.device)
        self.logits = torch.empty((batch, n_classes, height, width), dtype=torch.float32, device=self.device)
        self.dlogits = torch.empty_like(self.logits)
        self._build()
        self.workspace = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=self.device)
        # one device table -> one launch repacks every layer's weights (dc_pack_all)
        arr = (L.PackEntry * len(self._pack_entries))(*self._pack_entries)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        self._pack_table = raw.to(self.device)
        npk = len(self._pack_entries)
        self.pack_ops.append(lambda: L.call("dc_pack_all", self.dt, L.dptr(self._pack_table), npk, self._st()))
        # Weight gradients are off the critical path of backward (nothing downstream reads them until the optimizer) and
        # are MFMA-bound, while the chain they branch off (BN backward, depthwise data gradient) is HBM-bound: they run
        # on a side HIP stream with a private workspace and overlap it.  backward() joins the streams at the end.
        # DC_SIDE_PRIORITY=low puts the weight gradients on a lowest-priority HIP stream (dc_stream_create); measured the same
        # step time as a plain stream (41.3-41.6 ms): workgroups are not preempted, so priority only orders the dispatch.
        self._side_raw = None
        if os.environ.get("DC_SIDE_PRIORITY", "normal") == "low":
            raw = C.c_void_p()
            with torch.cuda.device(self.device):
                L.call("dc_stream_create", -1, C.byref(raw))
            self._side_raw = raw
            self.side = torch.cuda.ExternalStream(raw.value, device=self.device)
        else:
            self.side = torch.cuda.Stream(device=self.device)
        self.workspace2 = torch.empty(max(self.ws_bytes, 256), dtype=torch.uint8, device=self.device)
        self.use_side_stream = os.environ.get("DC_SIDE_STREAM", "1") != "0"
        self.on_grad_ready: Optional[Callable[[List[str]], None]] = None
        self._debug_skip_side = os.environ.get("DC_DEBUG_SKIP_SIDE", "0") == "1"
        self._debug_skip_kind = os.environ.get("DC_DEBUG_SKIP_KIND", "")     # "_conv" / "_dw": skip that kind of weight gradient (timing only)
        self.region_marks: Optional[list] = None      # bench.py sets a list to collect (name, event) at the encoder boundaries
        self.loss_args: Optional[dict] = None         # set per call by nn.TrainStep (DC_FUSE_HEAD_LOSS): loss inside the classifier's kernel

    # ------------------------------------------------------------------------------------------------ helpers
    def pptr(self, name: str) -> C.c_void_p:
        return C.c_void_p(self.params.data_ptr() + self.layout.params[name].offset * 4)

    def gptr(self, name: str) -> C.c_void_p:
        return C.c_void_p(self.grads.data_ptr() + self.layout.params[name].offset * 4)

    def bptr(self, name: str) -> C.c_void_p:
        return C.c_void_p(self.buffers.data_ptr() + self.layout.buffers[name][0] * 4)

    def param_view(self, name: str) -> torch.Tensor:
        p = self.layout.params[name]
        return self.params[p.offset:p.offset + math.prod(p.shape)].view(p.shape)

    def grad_view(self, name: str) -> torch.Tensor:
        p = self.layout.params[name]
        return self.grads[p.offset:p.offset + math.prod(p.shape)].view(p.shape)

    def buffer_view(self, name: str) -> torch.Tensor:
        if name in self.layout.nbt:
            return self.nbt[self.layout.nbt[name]]
        off, c = self.layout.buffers[name]
        return self.buffers[off:off + c]

    def _need_ws(self, nbytes: int) -> None:
        self.ws_bytes = max(self.ws_bytes, int(nbytes))

    def _wsptr(self) -> C.c_void_p:
        return C.c_void_p(self.workspace.data_ptr())

    def _on_side(self, fn: Callable[[C.c_void_p], None]) -> None:
        """Run fn(workspace_ptr) on the side stream, ordered after everything enqueued so far on the current stream."""
        if self._debug_skip_side:      # timing experiments only (DC_DEBUG_SKIP_SIDE=1): the weight gradients are NOT computed
            return
        if self._debug_skip_kind and self._debug_skip_kind in getattr(fn, "__qualname__", ""):
            return
        if not self.use_side_stream:
            fn(self._wsptr())
            return
        # (a library call, not a torch event: the fence is then part of a recorded launch list, lib.Program)
        L.call("dc_stream_fence", L.stream_ptr(), C.c_void_p(self.side.cuda_stream))
        with torch.cuda.stream(self.side):
            fn(C.c_void_p(self.workspace2.data_ptr()))

    @staticmethod
    def _st():
        return L.stream_ptr()

    def _f32(self, n: int) -> torch.Tensor:
        return torch.empty(n, dtype=torch.float32, device=self.device)

    SUM_ARENA_FLOATS = 1 << 21      # 8 MiB: a sum row is double[2][C] = 4 * C floats; the network has some 150 BatchNorms of at most 2048 channels

    def _sum_row(self, channels: int) -> torch.Tensor:
        """A sum row (double[2][channels], as a float32 view) out of the arena that _run_forward zeroes once per train step."""
        if self._sum_arena is None:
            self._sum_arena = torch.zeros(self.SUM_ARENA_FLOATS, dtype=torch.float32, device=self.device)
        n = (4 * channels + 15) // 16 * 16          # 64-byte granules
        if self._sum_used + n > self.SUM_ARENA_FLOATS:
            raise L.DeepcamHipError("the arena of BatchNorm sum rows is full (Engine.SUM_ARENA_FLOATS)")
        row = self._sum_arena[self._sum_used:self._sum_used + n]
        self._sum_used += n
        return row

    # ------------------------------------------------------------------------------------------------ op builders
    def _conv(self, x: Act, wname: str, cout: int, k: int = 1, stride: int = 1, pad: int = 0, dil: int = 1,
              transposed: bool = False, out: Act = None, stats: bool = True, bias: str = None, name: str = None,
              need_dx: bool = True, f32: bool = False, fwd_group: list = None, sole_consumer: bool = False, bn_fuse: bool = False,
              sum_row: bool = True):
        """Dense conv (implicit GEMM).  Returns (y, slab, rows).  fwd_group: a list that collects (dilation, wf, y, slab) instead
        of this layer's own forward launch (the caller then launches the members together: _dilated_group_fwd).
        sole_consumer: x is a stored BatchNorm(+ReLU) output that feeds this conv and nothing else; its BatchNorm's backward sums
        are then taken in this layer's data-gradient epilogue (dc_conv_dgrad_bnstats) instead of a pass of their own.
        bn_fuse: this is the pointwise conv of a separable conv (x, a depthwise output, has no other consumer) and y feeds one BatchNorm: where
        the library serves the shape (dc_pw_bn_bwd_rows: the entry flow's first block), that BatchNorm's backward apply, this layer's data
        gradient and its weight gradient are ONE pass over (dout, y, x) -- y's gradient is never stored (dc_pw_bn_bwd)."""
        lib = L.load()
        dt, tdtype = (L.DC_F32, torch.float32) if f32 else (self.dt, self.dtype)
        d = L.ConvDesc(dt, 3 if transposed else k, stride, pad, dil, 1 if transposed else 0, x.C, cout)
        ho, wo = C.c_int(), C.c_int()
        L.call("dc_conv_out_hw", C.byref(d), x.H, x.W, C.byref(ho), C.byref(wo))
        y = out or Act(self, name or wname, x.N, ho.value, wo.value, cout, dtype=tdtype)
        assert (y.H, y.W, y.C) == (ho.value, wo.value, cout) and y.buf.dtype == tdtype and x.buf.dtype == tdtype
        nwf, nwb = C.c_size_t(), C.c_size_t()
        L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
        wf = torch.empty(nwf.value, dtype=tdtype, device=self.device)
        wb = torch.empty(nwb.value, dtype=tdtype, device=self.device) if need_dx else None
        # (a pointwise layer that hands the library both weight images gets the 224-pixel tiles' compact slab: one row per tile, no zero rows --
        # at local batch 4 that is 62 rows, short enough for the kernels that run the BatchNorm finalize themselves)
        both = self.conv_kn and k == 1 and stride == 1 and not transposed and not f32 and need_dx and dt == L.DC_BF16
        rows = (lib.dc_conv_stat_rows_kn if both else lib.dc_conv_stat_rows)(C.byref(d), x.N, x.H, x.W) if stats else 0
        slab = self._f32(2 * rows * cout) if stats else None
        if (stats and both and sum_row and self.bn_sum_rows and fwd_group is None and bias is None and out is None
                and lib.dc_conv_sum_row_kn(C.byref(d), x.N, x.H, x.W) == 1):
            # the 224-pixel tile kernel adds its sums to ONE fp64 row (rows = -1 from here on: _bn hands it to the consumer of the coefficients)
            rows, slab = -1, self._sum_row(cout)
        y.pw_fuse = None
        if (bn_fuse and self.fuse_pw_bn_bwd and k == 1 and stride == 1 and not transposed and bias is None and not f32 and need_dx
                and out is None and x.parent is None):
            frows = lib.dc_pw_bn_bwd_rows(dt, x.C, cout, x.M)
            if frows > 0:
                y.pw_fuse = {"rows": frows, "taken": False}      # the BatchNorm's make_bwd (which runs first) takes it or leaves it
        N, H, W = x.N, x.H, x.W
        pw, gw = self.pptr(wname), self.gptr(wname)
        pb = self.pptr(bias) if bias else None
        if f32 and self.dt != L.DC_F32:
            # the fp32 island of a bf16 engine (image-pool conv) keeps its own pack call: dc_pack_all packs one dtype
            self.pack_ops.append(lambda: L.call("dc_conv_pack_weights", C.byref(d), pw, L.dptr(wf), L.dptr(wb), self._st()))
        else:
            self._pack_entries.append(L.PackEntry(pw.value, wf.data_ptr(), wb.data_ptr() if wb is not None else None, x.C, cout,
                                                  d.k * d.k, 1 if transposed else 0))
            self._keep += [wf] + ([wb] if wb is not None else [])

        # a pointwise layer hands the library BOTH packed images: the 224 x 384 tile kernel streams its weight stages from the [k][n] one
        # (csrc/igemm224.hip; dc_conv_fwd_kn / dc_conv_dgrad_kn fall back to the plain path for every shape it does not serve)
        assert both == (self.conv_kn and k == 1 and stride == 1 and not transposed and not f32 and wb is not None and dt == L.DC_BF16)

        def fwd(train: bool):
            if both:
                L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, x.ptr, x.ld, L.dptr(wf), L.dptr(wb), pb, y.ptr, y.ld,
                       L.dptr(slab) if train else None, rows, 0, self._st())
            else:
                L.call("dc_conv_fwd", C.byref(d), N, H, W, x.ptr, x.ld, L.dptr(wf), pb, y.ptr, y.ld,
                       L.dptr(slab) if train else None, 0, self._st())

        if fwd_group is not None:
            assert k == 3 and stride == 1 and pad == dil and not transposed and bias is None and not f32
            fwd_group.append((dil, wf, y, slab, d, x))
        else:
            self.fwd_train.append(lambda: fwd(True))
            self.fwd_eval.append(lambda: fwd(False))
            y.conv_fwd = {"wf": wf, "slab": slab, "rows": rows, "plain": k == 1 and stride == 1 and not transposed and bias is None and not f32 and out is None}

        def make_bwd():
            pwf = y.pw_fuse
            if pwf is not None and pwf["taken"]:
                # the BatchNorm behind this layer left its backward apply to this layer: one pass writes dx and this layer's weight-gradient rows,
                # which stay in their slab until the next dense weight-gradient launch folds them (as the depthwise rows do); the BatchNorm's
                # parameter gradients are read by that pass, so they are reported ready here, not by the BatchNorm's own step
                mode = x.take_grad_mode()
                assert mode == 0, f"{wname}: the fused backward pass writes the input gradient first"
                dx, do = x.grad, pwf["do"]
                frows = pwf["rows"]
                fslab = self._f32(frows * cout * x.C)
                ready = [wname] + pwf["names"]
                self._fold_seq.append(("dw", L.FoldEntry(fslab.data_ptr(), gw.value, L.DC_FOLD_CONV, frows, 1, cout, x.C), wname, ready))

                def bwd_fused():
                    L.call("dc_pw_bn_bwd", dt, y.M, x.C, cout, y.M, do.ptr, do.ld, y.ptr, y.ld, pwf["relu"], pwf["gam"], L.dptr(pwf["mean"]),
                           L.dptr(pwf["invstd"]), pwf["dgam"], pwf["dbet"], L.dptr(pwf["scale"]), L.dptr(pwf["shift"]), x.ptr, x.ld, L.dptr(wb),
                           dx.ptr, dx.ld, L.dptr(fslab), frows, self._st())
                return bwd_fused, ready
            dy = y.grad
            mode = x.take_grad_mode() if need_dx else 0
            dx = x.grad if need_dx else None
            ready = [wname] + ([bias] if bias else [])
            if bias:
                self._need_ws(lib.dc_colsum_workspace(y.M, cout))

            # Weight gradients of consecutive layers of ONE geometry (the 728 -> 728 pointwise convs of the middle flow) are
            # deferred and launched together (one dc_conv_wgrad_partial call): _plan_folds() fixes the roles once the program is resolved.
            bsrc = getattr(x, "bn_src", None) if (sole_consumer and need_dx and self.fuse_bn_conv) else None
            srows = lib.dc_conv_dgrad_bnstats_rows(C.byref(d), N, H, W) if (bsrc is not None and mode == 0 and x.parent is None) else 0
            if srows > 0:
                sslab = self._f32(2 * srows * x.C)
                x.fused_bwd = (sslab, srows)       # read by the BatchNorm's make_bwd, which runs after this one
                by = bsrc["y"]
            rec = {"key": (dt, d.k, stride, pad, dil, d.transposed, x.C, cout, N, H, W, x.ld, dy.ld), "d": d, "x": x, "dy": dy,
                   "gw": gw, "wname": wname, "ready": ready, "role": "solo", "launch": None, "fold": None, "bias": bool(bias)}
            self._wg_recs.append(rec)
            self._fold_seq.append(("dense", rec))

            def wgrad():
                if rec["role"] == "defer":
                    return
                # partial sums of this layer (or of the group this layer closes) into their slabs, then ONE fold of those slabs and of the
                # depthwise rows produced since the previous fold -- both behind a single fence on the weight-gradient stream
                cnt, xs, dys, slabs, splits = rec["launch"]
                ents, nent = rec["fold"]
                for i, m in enumerate(rec["members"]):      # (an activation's buffer may be swapped between steps: the input pipeline's slots)
                    xs[i] = m["x"].ptr.value
                    dys[i] = m["dy"].ptr.value

                def side(ws):
                    L.call("dc_conv_wgrad_partial", C.byref(d), N, H, W, cnt, xs, x.ld, dys, dy.ld, slabs, splits, self._st())
                    L.call("dc_fold_slabs", ents, nent, self._st())
                if not need_dx and self.inline_last_wgrad:
                    # a layer without a data gradient (the stem) ends the backward chain: nothing is left for the main stream to run beside
                    # this weight gradient, and on the side stream it would queue behind the previous layer's
                    side(self._wsptr())
                else:
                    self._on_side(side)

            def bwd():
                if bias:
                    L.call("dc_colsum", dt, y.M, cout, dy.ptr, dy.ld, self.gptr(bias), self._wsptr(), self._st())
                wgrad()
                if need_dx and srows > 0:
                    L.call("dc_conv_dgrad_bnstats", C.byref(d), N, H, W, dy.ptr, dy.ld, L.dptr(wb), dx.ptr, dx.ld, by.ptr, by.ld,
                           L.dptr(bsrc["mean"]), L.dptr(bsrc["invstd"]), L.dptr(bsrc["scale"]), L.dptr(bsrc["shift"]), bsrc["relu"],
                           L.dptr(sslab), self._st())
                elif need_dx and both:
                    L.call("dc_conv_dgrad_kn", C.byref(d), N, H, W, dy.ptr, dy.ld, L.dptr(wb), L.dptr(wf), dx.ptr, dx.ld, mode, self._st())
                elif need_dx:
                    L.call("dc_conv_dgrad", C.byref(d), N, H, W, dy.ptr, dy.ld, L.dptr(wb), dx.ptr, dx.ld, mode, self._st())
            return bwd, ready

        self.bwd.append(make_bwd)
        return y, slab, rows

    def _fork_forward(self, begin) -> None:
        """Moves the forward ops appended since `begin` (indices into fwd_train / fwd_eval) onto the side stream: one closure that orders
        the side stream behind the main stream's current position and issues them there.  The caller appends the join (a fence back)."""
        for lst, b in ((self.fwd_train, begin[0]), (self.fwd_eval, begin[1])):
            ops = lst[b:]
            del lst[b:]

            def forked(ops=ops):
                if not self.use_side_stream:          # (bench.py's one-stream passes switch it off at run time)
                    for op in ops:
                        op()
                    return
                L.call("dc_stream_fence", L.stream_ptr(), C.c_void_p(self.side.cuda_stream))
                with torch.cuda.stream(self.side):
                    for op in ops:
                        op()
            lst.append(forked)
        self._forked = True

    def _dilated_group_fwd(self, members: list) -> None:
        """One launch for the forward of several "same" dilated 3x3 convolutions of one input (the atrous ASPP branches,
        deeplab_xception.py:445-447): dc_conv_fwd_dilated_group.  Bit-identical to the per-layer launches."""
        cnt = len(members)
        d0, x = members[0][4], members[0][5]
        y0 = members[0][2]
        assert all(m[5] is x and m[2].ld == y0.ld and (m[2].H, m[2].W, m[2].C) == (y0.H, y0.W, y0.C) for m in members)
        dils = (C.c_int * cnt)(*[m[0] for m in members])
        wfs = (C.c_void_p * cnt)(*[m[1].data_ptr() for m in members])
        slabs = (C.c_void_p * cnt)(*[m[3].data_ptr() for m in members])
        N, H, W = x.N, x.H, x.W
        # few pixels under a long K loop (local batch 2: 81 tiles of 576 K steps on 256 CUs): the library cuts the launch along K when it is
        # given room for the partial tiles (dc_conv_fwd_dilated_group_ws)
        wsb = L.load().dc_conv_fwd_dilated_group_workspace(C.byref(d0), N, H, W, cnt, dils) if self.dt == L.DC_BF16 else 0
        kws = torch.empty(wsb, dtype=torch.uint8, device=self.device) if wsb > 0 else None
        if kws is not None:
            self._keep.append(kws)

        def fwd(train: bool):
            ys = (C.c_void_p * cnt)(*[m[2].ptr.value for m in members])
            if kws is not None:
                L.call("dc_conv_fwd_dilated_group_ws", C.byref(d0), N, H, W, cnt, dils, x.ptr, x.ld, wfs, ys, y0.ld, slabs if train else None,
                       L.dptr(kws), wsb, self._st())
                return
            L.call("dc_conv_fwd_dilated_group", C.byref(d0), N, H, W, cnt, dils, x.ptr, x.ld, wfs, ys, y0.ld, slabs if train else None,
                   self._st())

        self.fwd_train.append(lambda: fwd(True))
        self.fwd_eval.append(lambda: fwd(False))

    def _dw(self, x, wname: str, stride: int, dil: int, name: str) -> Act:
        """Depthwise 3x3.  x is an Act, or a LazyAct (then the preceding BatchNorm+ReLU is applied on load)."""
        lib = L.load()
        lazy = isinstance(x, LazyAct)
        src = x.y if lazy else x                       # tensor actually read
        ps = L.dptr(x.scale) if lazy else None
        psh = L.dptr(x.shift) if lazy else None
        prelu = (1 if x.relu else 0) if lazy else 0
        if lazy:
            self._keep += [x.scale, x.shift]
        Ho, Wo = (x.H - 1) // stride + 1, (x.W - 1) // stride + 1
        y = Act(self, name, x.N, Ho, Wo, x.C)
        N, H, W, Cc = x.N, x.H, x.W, x.C
        pm, gw = self.pptr(wname), self.gptr(wname)
        wpk = self._f32(9 * Cc)
        self._keep.append(wpk)             # closures below hold only the raw pointer
        pw = L.dptr(wpk)
        self._pack_entries.append(L.PackEntry(pm.value, wpk.data_ptr(), None, Cc, Cc, 9, 2))
        self._need_ws(lib.dc_dwconv_wgrad_workspace(Cc, N, H, W, stride))

        def fwd():
            L.call("dc_dwconv_fwd", self.dt, Cc, stride, dil, N, H, W, src.ptr, src.ld, pw, y.ptr, y.ld, ps, psh, prelu, self._st())

        fwd_train = fwd
        fin = getattr(x, "fin", None) if lazy else None
        if fin is not None and not fin["taken"] and lib.dc_dwconv_fwd_fin_ok(self.dt, Cc, stride, dil, N, H, W):
            fin["taken"] = True           # this layer runs the BatchNorm's finalize itself (dc_dwconv_fwd_fin)

            def fwd_train():
                L.call("dc_dwconv_fwd_fin", self.dt, Cc, stride, dil, N, H, W, src.ptr, src.ld, pw, y.ptr, y.ld, prelu, *fin["args"], self._st())
        else:
            fin = None
        self.fwd_train.append(fwd_train)
        self.fwd_eval.append(fwd)
        y.dw_fwd = {"fin": fin is not None, "src": src, "ps": ps, "psh": psh, "prelu": prelu, "taps": pw, "stride": stride, "dil": dil}     # (_sep may fuse the forward)

        def make_bwd():
            dy = y.grad
            mode = x.take_grad_mode()
            dx = x.grad
            # the producer is a never-stored BatchNorm output and this is its only consumer: take that BatchNorm's backward
            # sums (sum g, sum g*xhat) on the way out of the data-gradient kernel instead of re-reading dx and y for them
            srows = lib.dc_dwconv_dgrad_bnstats_rows(self.dt, Cc, stride, dil, N, H, W) if (lazy and self.fuse_bn_reduce and mode == 0) else 0
            wrows = 0
            if srows == 0 and self.fuse_dw_wgrad:
                # stored input (or a lazy one whose statistics are taken elsewhere): data gradient + weight-gradient rows in one kernel
                wrows = lib.dc_dwconv_dgrad_wgrad_rows(self.dt, Cc, stride, dil, N, H, W)
            # BatchNorm sums into ONE fp64 row (the persistent kernel's workgroups add to it): the BatchNorm's apply then runs the finalize itself
            sum_ok = self.bn_sum_rows and lib.dc_dwconv_dgrad_sum_row_ok(self.dt, Cc, stride, dil, N, H, W) == 1
            stats_sum = False
            if srows > 0:
                if self.fuse_dw_wgrad:
                    wrows = lib.dc_dwconv_dgrad_wgrad_rows(self.dt, Cc, stride, dil, N, H, W)
                stats_sum = sum_ok and wrows > 0
                sslab = self._sum_row(Cc) if stats_sum else self._f32(2 * srows * Cc)
                x.fused_bwd = (sslab, -1 if stats_sum else srows)
                mean_p, invstd_p = L.dptr(x.mean), L.dptr(x.invstd)
            # the stored input is a block output relu(bn(y) + residual) and this data gradient completes d(x): that BatchNorm's sums ride along
            res = getattr(x, "bn_res", None) if (not lazy and wrows > 0 and srows == 0 and self.fuse_bn_res) else None
            if res is not None and getattr(x, "fused_bwd", 1) is None and x.parent is None and \
                    lib.dc_dwconv_dgrad_wgrad_bnres_rows(self.dt, Cc, stride, dil, N, H, W) == wrows:
                rslab2 = self._sum_row(Cc) if sum_ok else self._f32(2 * wrows * Cc)
                x.fused_bwd = (rslab2, -1 if sum_ok else wrows, x.grad_takes)       # (this layer's take_grad_mode call above was the last one so far)
            else:
                res = None
            # the BatchNorm's make_bwd runs after this one: if it finds that ANOTHER consumer wrote the gradient later (this layer was not the
            # last writer after all), it cancels the fusion and this layer falls back to the plain fused data + weight gradient
            # the stored input is a BatchNorm(+ReLU) output whose mask is recomputed from the BatchNorm's input (bn_src) and that several layers read
            # (bn2: block 1's first depthwise layer and its shortcut conv): the depthwise layer, writing the gradient LAST, takes that BatchNorm's
            # backward sums on the way (dc_dwconv_dgrad_bnstats_wgrad_add) instead of a dc_bn_bwd_reduce pass over the 226 MB gradient and input
            bsrc = getattr(x, "bn_src", None) if (not lazy and res is None and wrows > 0 and srows == 0 and stride == 1 and self.fuse_bn_src_dw) else None
            if bsrc is not None and getattr(x, "fused_bwd", 1) is None and x.parent is None:
                bslab2 = self._f32(2 * wrows * Cc)
                x.fused_bwd = (bslab2, wrows, x.grad_takes)
            else:
                bsrc = None
            fuse_state = {"res": res, "bsrc": bsrc}
            if res is not None or bsrc is not None:
                x.fused_bwd_cancel = lambda: fuse_state.update(res=None, bsrc=None)
            ready = [wname]
            if wrows > 0:
                # the rows stay in this layer's own slab until the next dense weight-gradient launch folds them (dc_fold_slabs); the
                # gradient is reported ready there (_plan_folds moves the name)
                wslab = self._f32(wrows * 9 * Cc)
                self._fold_seq.append(("dw", L.FoldEntry(wslab.data_ptr(), gw.value, L.DC_FOLD_DW, wrows, 9, Cc, 1), wname, ready))

            def dw_wgrad(ws):
                L.call("dc_dwconv_wgrad", self.dt, Cc, stride, dil, N, H, W, src.ptr, src.ld, dy.ptr, dy.ld, ws, gw, ps, psh, prelu, self._st())

            def bwd():
                res = fuse_state["res"]
                if res is not None:
                    L.call("dc_dwconv_dgrad_wgrad_bnres_sum" if sum_ok else "dc_dwconv_dgrad_wgrad_bnres", self.dt, Cc, stride, dil, N, H, W, dy.ptr, dy.ld, pw, dx.ptr if mode else None, dx.ld,
                           dx.ptr, dx.ld, src.ptr, src.ld, L.dptr(wslab), res["y"].ptr, res["y"].ld, L.dptr(res["mean"]), L.dptr(res["invstd"]),
                           res["relu"], L.dptr(rslab2), self._st())
                    return
                bs = fuse_state["bsrc"]
                if bs is not None:
                    L.call("dc_dwconv_dgrad_bnstats_wgrad_add", self.dt, Cc, stride, dil, N, H, W, dy.ptr, dy.ld, pw, dx.ptr if mode else None, dx.ld,
                           dx.ptr, dx.ld, bs["y"].ptr, bs["y"].ld, L.dptr(bs["mean"]), L.dptr(bs["invstd"]), L.dptr(bs["scale"]), L.dptr(bs["shift"]),
                           bs["relu"], L.dptr(bslab2), L.dptr(wslab), self._st())
                    return
                if wrows > 0 and srows == 0:
                    L.call("dc_dwconv_dgrad_wgrad", self.dt, Cc, stride, dil, N, H, W, dy.ptr, dy.ld, pw, dx.ptr if mode else None, dx.ld,
                           dx.ptr, dx.ld, src.ptr, src.ld, ps, psh, prelu, L.dptr(wslab), self._st())
                    return
                if wrows > 0:
                    L.call("dc_dwconv_dgrad_bnstats_wgrad_sum" if stats_sum else "dc_dwconv_dgrad_bnstats_wgrad", self.dt, Cc, stride, dil, N, H, W, dy.ptr, dy.ld, pw, dx.ptr, dx.ld, src.ptr, src.ld,
                           mean_p, invstd_p, ps, psh, prelu, L.dptr(sslab), L.dptr(wslab), self._st())
                    return
                self._on_side(dw_wgrad)
                if srows > 0:
                    L.call("dc_dwconv_dgrad_bnstats", self.dt, Cc, stride, dil, N, H, W, dy.ptr, dy.ld, pw, dx.ptr, dx.ld, src.ptr, src.ld,
                           mean_p, invstd_p, ps, psh, prelu, L.dptr(sslab), self._st())
                else:
                    L.call("dc_dwconv_dgrad", self.dt, Cc, stride, dil, N, H, W, dy.ptr, dy.ld, pw, dx.ptr if mode else None, dx.ld,
                           dx.ptr, dx.ld, self._st())
            return bwd, ready

        self.bwd.append(make_bwd)
        return y

    def _bn(self, y: Act, slab: torch.Tensor, rows: int, bname: str, relu: bool, residual: Act = None, out: Act = None,
            name: str = None, lazy: bool = False):
        """BatchNorm (+residual) (+ReLU).  lazy=True: do not store the result; return a LazyAct for a depthwise consumer."""
        lib = L.load()
        Cc, M = y.C, y.M
        assert not (lazy and (residual is not None or out is not None))
        bdt = L.dtype_code(y.buf.dtype)
        if lazy:
            o = None
        else:
            o = out or Act(self, name or bname, y.N, y.H, y.W, Cc, dtype=y.buf.dtype)
            assert o.buf.dtype == y.buf.dtype
        scale, shift, mean, invstd = (self._f32(Cc) for _ in range(4))
        for suffix, t in ((".scale", scale), (".shift", shift), (".save_mean", mean), (".save_invstd", invstd)):
            self.saved[bname + suffix] = t
        gam, bet = self.pptr(bname + ".weight"), self.pptr(bname + ".bias")
        rm, rv = self.bptr(bname + ".running_mean"), self.bptr(bname + ".running_var")
        nbt = C.c_void_p(self.nbt.data_ptr() + 8 * self.layout.nbt[bname + ".num_batches_tracked"])
        relu_i = 1 if relu else 0
        rptr = (lambda: residual.ptr) if residual is not None else (lambda: None)
        rld = residual.ld if residual is not None else 0

        # a short slab of a small tensor (the 54 rows the 128-pixel tiles leave at local batch 2 on the 48 x 72 layers): the consumer of the
        # coefficients sums it itself, every workgroup for its own channels (same bits): one launch and one dependent boundary less per
        # BatchNorm.  Lazy: the depthwise layer that reads through this BatchNorm takes it (_dw sets fin["taken"])
        fin = None
        if (self.fuse_bn_fwd_fin and not self._debug_skip_finalize and
                (rows == -1 or (rows <= lib.dc_bn_bwd_apply_fin_max_rows() and M <= self.bn_fin_max_m))):      # (a sum row: at any size)
            fin = {"taken": False, "args": (M, L.dptr(slab), rows, gam, bet, rm, rv, nbt, BN_MOMENTUM, BN_EPS, L.dptr(scale), L.dptr(shift),
                                            L.dptr(mean), L.dptr(invstd))}

        def fwd_train():
            if fin is not None and (fin["taken"] or not lazy):
                if not lazy:
                    L.call("dc_bn_apply_fin", bdt, M, Cc, M, y.ptr, y.ld, *fin["args"][1:], rptr(), rld, relu_i, o.ptr, o.ld, self._st())
                return
            if self._debug_skip_finalize == "async":
                # (timing experiment with realistic data: the finalize runs unordered on a stream of its own, the chain uses the previous
                # step's coefficients)
                if not hasattr(self, "_dbg_stream"):
                    self._dbg_stream = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(self._dbg_stream):
                    L.call("dc_bn_finalize", Cc, M, L.dptr(slab), rows, gam, bet, rm, rv, nbt, BN_MOMENTUM, BN_EPS, L.dptr(scale),
                           L.dptr(shift), L.dptr(mean), L.dptr(invstd), self._st())
            elif self._debug_skip_finalize not in ("fwd", "both"):
                L.call("dc_bn_finalize", Cc, M, L.dptr(slab), rows, gam, bet, rm, rv, nbt, BN_MOMENTUM, BN_EPS, L.dptr(scale),
                       L.dptr(shift), L.dptr(mean), L.dptr(invstd), self._st())
            if not lazy:
                L.call("dc_bn_apply", bdt, M, Cc, y.ptr, y.ld, L.dptr(scale), L.dptr(shift), rptr(), rld, relu_i, o.ptr, o.ld, self._st())

        def fwd_eval():
            L.call("dc_bn_eval_coeffs", Cc, gam, bet, rm, rv, BN_EPS, L.dptr(scale), L.dptr(shift), self._st())
            if not lazy:
                L.call("dc_bn_apply", bdt, M, Cc, y.ptr, y.ld, L.dptr(scale), L.dptr(shift), rptr(), rld, relu_i, o.ptr, o.ld, self._st())

        lz = LazyAct(y, scale, shift, relu, (name or bname) + ".lazy") if lazy else None
        if lz is not None:
            lz.mean, lz.invstd, lz.fused_bwd = mean, invstd, None
            lz.fin = fin
        elif residual is None and out is None and (self.mask_from_y or not relu):
            # a stored BatchNorm(+ReLU) output whose ReLU mask is recomputed from y: a sole dense-conv consumer may take this
            # BatchNorm's backward sums in its data-gradient epilogue (_conv(..., sole_consumer=True), the classifier head)
            o.bn_src = {"y": y, "scale": scale, "shift": shift, "mean": mean, "invstd": invstd, "relu": relu_i}
            o.fused_bwd = None
        elif not lazy and residual is not None and out is None:
            # a block output relu(bn(y) + residual): the depthwise conv that reads it first (the next block's) is the last writer of its
            # gradient and may take this BatchNorm's backward sums on the way (_dw: dc_dwconv_dgrad_wgrad_bnres)
            o.bn_res = {"y": y, "mean": mean, "invstd": invstd, "relu": relu_i}
            o.fused_bwd = None

        self.fwd_train.append(fwd_train)
        self.fwd_eval.append(fwd_eval)
        brows = lib.dc_bn_stat_rows(M)
        # the reduce pass (where no producer takes the sums on its way) adds to a sum row as well -- on tensors of 256 channels and more: same-address
        # atomics go at one per 17 ns, so the pass is cut into 256 row blocks in that form, too few workgroups for a narrow tensor to stream at rate
        # (128 channels x 1.77 M pixels: 151 -> 175 us, more than the fold and the finalize launch it saves; scripts/sum_row_contention.py)
        bsum = self.bn_sum_rows and Cc >= 256
        if bsum:
            brows, bslab = -1, self._sum_row(Cc)
        else:
            bslab = self._f32(2 * brows * Cc)

        def make_bwd():
            apply_by = getattr(lz, "apply_by", None) if lazy else None      # the consumer writes dy itself (the head, two passes)
            do = None if apply_by is not None else (lz.grad if lazy else o.grad)
            assert y.take_grad_mode() == 0, "a conv output feeds exactly one BatchNorm"
            g_out = None
            if residual is not None:
                assert residual.take_grad_mode() == 0, f"{bname}: residual gradient must be first written here"
                g_out = residual.grad
            dgam, dbet = self.gptr(bname + ".weight"), self.gptr(bname + ".bias")
            # ReLU mask: recomputed from y with the forward scale / shift whenever there is no residual (one tensor read less
            # in both backward kernels, which run at the HBM roofline); from the stored output when a residual was added
            from_y = lazy or (relu and residual is None and self.mask_from_y)
            mrelu = (2 if relu else 0) if from_y else relu_i
            optr = (lambda: None) if from_y else (lambda: o.ptr)
            old_ = 0 if from_y else o.ld
            # the producing pointwise conv applies this BatchNorm's backward itself, in one pass with its own two gradients (_conv: bn_fuse)
            pwf = getattr(y, "pw_fuse", None) if (apply_by is None and residual is None and mrelu in (0, 2)) else None
            if pwf is not None:
                pwf.update(taken=True, do=do, gam=gam, mean=mean, invstd=invstd, dgam=dgam, dbet=dbet, scale=scale, shift=shift, relu=mrelu,
                           names=[bname + ".weight", bname + ".bias"])
            dy = None if pwf is not None else y.grad

            # set by the consumer (a depthwise conv for a lazy output, a dense conv / the head for a stored one): its make_bwd ran first
            fused = lz.fused_bwd if lazy else getattr(o, "fused_bwd", None)
            if fused is not None and len(fused) > 2 and o.grad_takes != fused[2]:
                # the consumer that was to take the sums is not the last writer of this gradient (its sums would miss the later addends):
                # no fusion for this BatchNorm -- the consumer runs its plain kernel, the sums come from dc_bn_bwd_reduce below
                cancel = getattr(o, "fused_bwd_cancel", None)
                if cancel is None:
                    raise L.DeepcamHipError(f"{bname}: a consumer took this BatchNorm's backward sums but is not the last writer of the gradient")
                cancel()
                fused = None
            rslab, rrows = (fused[0], fused[1]) if fused is not None else (bslab, brows)
            # a short slab (the 42 rows the persistent depthwise data gradient leaves on the 728-channel layers) of a SMALL tensor: the apply
            # kernel sums it itself (dc_bn_bwd_apply_fin, same bits): one launch and one dependent boundary less per BatchNorm of the middle
            # flow.  Every 32-row block repeats the sum for its channels, so it pays only while the grid is small: local batch 2 12.47 ->
            # 12.38 ms, batch 4 19.80 -> 19.83, batch 8 33.47 -> 33.88 (2 592 blocks re-reading 86 KB each).  With 64 rows per block in this form
            # (option bn_fin_mul_bwd = 2) batch 4 gains too (19.02 -> 18.93) and batch 8 is level at best: on up to 16 384 pixels (DC_BN_FIN_MAX_M)
            fin_in_apply = (self.fuse_bn_bwd_fin and apply_by is None and pwf is None
                            and (rrows == -1 or (rrows <= lib.dc_bn_bwd_apply_fin_max_rows() and M <= self.bn_fin_max_m))      # (a sum row: at any size)
                            and os.environ.get("DC_DEBUG_SKIP_BN_FINALIZE", "") not in ("bwd", "both"))     # (the timing switch is read later)

            def bwd():
                if fused is None:
                    L.call("dc_bn_bwd_reduce_sum" if bsum else "dc_bn_bwd_reduce", bdt, M, Cc, do.ptr, do.ld, y.ptr, y.ld, optr(), old_, mrelu, L.dptr(mean),
                           L.dptr(invstd), L.dptr(bslab), L.dptr(scale), L.dptr(shift), self._st())
                if fin_in_apply:
                    L.call("dc_bn_bwd_apply_fin", bdt, M, Cc, M, do.ptr, do.ld, y.ptr, y.ld, optr(), old_, mrelu, gam, L.dptr(mean),
                           L.dptr(invstd), L.dptr(rslab), rrows, dgam, dbet, dy.ptr, dy.ld, g_out.ptr if g_out is not None else None,
                           g_out.ld if g_out is not None else 0, L.dptr(scale), L.dptr(shift), self._st())
                    return
                if self._debug_skip_finalize not in ("bwd", "both"):
                    L.call("dc_bn_bwd_finalize", Cc, L.dptr(rslab), rrows, dgam, dbet, self._st())
                if pwf is not None:
                    return
                if apply_by is not None:
                    apply_by(gam, dgam, dbet, dy)
                    return
                L.call("dc_bn_bwd_apply", bdt, M, Cc, M, do.ptr, do.ld, y.ptr, y.ld, optr(), old_, mrelu, gam, L.dptr(mean),
                       L.dptr(invstd), dgam, dbet, dy.ptr, dy.ld, g_out.ptr if g_out is not None else None,
                       g_out.ld if g_out is not None else 0, L.dptr(scale), L.dptr(shift), self._st())
            return bwd, ([] if pwf is not None else [bname + ".weight", bname + ".bias"])

        self.bwd.append(make_bwd)
        return lz if lazy else o

    def _sep(self, x, s: S.SepSpec, residual: Act = None, relu_override: Optional[bool] = None, lazy: bool = False):
        """depthwise 3x3 -> pointwise 1x1 [-> BN (+residual) (+ReLU)].  lazy: the BN output feeds only the next
        depthwise conv and is fused into it instead of being stored."""
        d = self._dw(x, s.prefix + ".conv1.weight", s.stride, s.dil, s.prefix + ".dw")
        # (where dc_sepconv_fwd may take the pair below, the statistics stay a row slab: that operator writes rows)
        sep_cand = self.fuse_sep_fwd and L.load().dc_sepconv_fwd_rows(self.dt, d.C, s.cout, s.stride, s.dil, d.N, d.H, d.W) > 0
        y, slab, rows = self._conv(d, s.prefix + ".pointwise.weight", s.cout, stats=bool(s.bn), name=s.prefix + ".pw",
                                   bn_fuse=bool(s.bn) and residual is None, sum_row=not sep_cand)
        # the entry flow's thin layers: depthwise + pointwise forward as ONE operator (dc_sepconv_fwd: d is written once and not read back);
        # the two forward entries the builders above appended are replaced, everything backward stays as it is
        dwf, cvf = getattr(d, "dw_fwd", None), getattr(y, "conv_fwd", None)
        if (self.fuse_sep_fwd and dwf is not None and not dwf["fin"] and cvf is not None and cvf["plain"] and d.parent is None and
                L.load().dc_sepconv_fwd_rows(self.dt, d.C, s.cout, dwf["stride"], dwf["dil"], d.N, d.H, d.W) > 0 and
                (slab is None or L.load().dc_sepconv_fwd_rows(self.dt, d.C, s.cout, 1, 1, d.N, d.H, d.W) <= rows)):
            src, wf_t = dwf["src"], cvf["wf"]
            if slab is not None:
                # the kernel leaves one statistics row per workgroup: the BatchNorm behind it is told so (the slab _conv sized for one row per 128
                # pixels is simply longer than needed), and its finalize needs no two-stage fold of 13 824 rows that are mostly zeros
                rows = L.load().dc_sepconv_fwd_rows(self.dt, d.C, s.cout, 1, 1, d.N, d.H, d.W)

            def sep_fwd(train: bool):
                L.call("dc_sepconv_fwd", self.dt, d.C, s.cout, d.N, d.H, d.W, src.ptr, src.ld, dwf["ps"], dwf["psh"], dwf["prelu"], dwf["taps"],
                       d.ptr, d.ld, L.dptr(wf_t), y.ptr, y.ld, L.dptr(slab) if (train and slab is not None) else None, rows if slab is not None else 0,
                       self._st())
            del self.fwd_train[-2:]
            del self.fwd_eval[-2:]
            self.fwd_train.append(lambda: sep_fwd(True))
            self.fwd_eval.append(lambda: sep_fwd(False))
        if not s.bn:
            return y
        relu = s.relu_after if relu_override is None else relu_override
        return self._bn(y, slab, rows, s.bn, relu, residual=residual, name=s.bn + ".out", lazy=lazy and self.fuse_bn_into_dw)

    def _xblock(self, blk: S.BlockSpec, z: Act, relu_out: Optional[bool] = None) -> Act:
        """One Xception Block (deeplab_xception.py:69-122) on an input that is already ReLU'd (the reference's in-place leading
        ReLU also reaches the shortcut): rep(z) + skip(z), materialised through the NEXT block's in-place ReLU when relu_out
        (blocks 1-19; block20's output is consumed as it is by conv3, :229-230)."""
        X = "xception_features."
        ro = blk.relu_out if relu_out is None else relu_out
        t = z
        nsep = len(blk.seps)
        for i, s in enumerate(blk.seps):
            last = i == nsep - 1
            if last and not blk.skip:
                # identity shortcut: out = relu(bn(pw(dw(t))) + z)   (x += skip, then the next block's in-place ReLU)
                t = self._sep(t, s, residual=z, relu_override=ro)
            else:
                # a BatchNorm between two separable convs is consumed by the next depthwise conv only
                t = self._sep(t, s, lazy=bool(s.bn) and not last)
        if blk.skip:
            ys, slab, rows = self._conv(z, f"{X}{blk.name}.skip.weight", blk.cout, stride=blk.stride, name=blk.name + ".skip")
            t = self._bn(ys, slab, rows, f"{X}{blk.name}.skipbn", ro, residual=t, name=blk.name + ".out")
        return t

    # ------------------------------------------------------------------------------------------------ network
    def _build(self) -> None:
        if self._builder is not None:
            self._enc_fwd_end = self._enc_bwd_makers = 0
            self._builder(self)
            self._resolve_backward()
            return
        self._build_network()
        self._resolve_backward()

    def _build_network(self) -> None:
        lib = L.load()
        B, H, W = self.B, self.H, self.W
        X = "xception_features."
        # ---- stem.  16 input channels: one layout pass (NCHW fp32 -> NHWC T) and the 3x3/s2 conv runs on the MFMA kernels
        #      (forward and weight gradient; the input needs no data gradient).  Other channel counts (--channels subsets)
        #      use the dedicated direct-convolution kernels that read NCHW in place.
        w1 = X + "conv1.weight"
        kpv = 8 if self.dtype == torch.bfloat16 else 4
        if self.n_input % kpv == 0:
            x0 = Act(self, "x_nhwc", B, H, W, self.n_input)

            self.x0 = x0
            self._x0_own = x0.buf

            def layout_fwd():
                if self._layout_forked:          # issued beside the weight repack (_run_forward): join the side stream here
                    self._layout_forked = False
                    L.call("dc_stream_fence", C.c_void_p(self.side.cuda_stream), L.stream_ptr())
                    return
                if self.x_in is not None:        # NCHW fp32 batch: the one layout pass; else x0.buf already IS the NHWC batch
                    L.call("dc_nchw_to_nhwc", self.dt, B, self.n_input, H, W, L.dptr(self.x_in), x0.ptr, x0.ld, self._st())

            self._layout_op = layout_fwd

            self.fwd_train.append(layout_fwd)
            self.fwd_eval.append(layout_fwd)
            self.bwd.append(lambda: ((lambda: None), []))
            c1, sslab, srows = self._conv(x0, w1, 32, k=3, stride=2, pad=1, name="stem", need_dx=False)
        else:
            c1 = Act(self, "stem", B, H // 2, W // 2, 32)
            srows = lib.dc_stem_stat_rows(B, H, W)
            sslab = self._f32(2 * srows * 32)
            self._need_ws(lib.dc_stem_wgrad_workspace(B, self.n_input, H, W))

            def stem_fwd():
                L.call("dc_stem_fwd", self.dt, B, self.n_input, H, W, L.dptr(self.x_in), self.pptr(w1), c1.ptr, c1.ld, L.dptr(sslab), self._st())

            self.fwd_train.append(stem_fwd)
            self.fwd_eval.append(stem_fwd)

            def stem_bwd_make():
                dy = c1.grad

                def bwd():
                    L.call("dc_stem_wgrad", self.dt, B, self.n_input, H, W, L.dptr(self.x_in), dy.ptr, dy.ld, self._wsptr(), self.gptr(w1), self._st())
                return bwd, [w1]

            self.bwd.append(stem_bwd_make)
        x = self._bn(c1, sslab, srows, X + "bn1", True)
        y, slab, rows = self._conv(x, X + "conv2.weight", 64, k=3, pad=1)
        x = self._bn(y, slab, rows, X + "bn2", True)

        # ---- Xception blocks.  Every block input `z` is already ReLU'd (the reference's in-place leading ReLU).
        low = None
        for blk in S.blocks():
            x = self._xblock(blk, x)
            if blk.name == "block1":
                low = x            # low_level_feat, after block2's in-place ReLU has hit it
        for i, s in enumerate(S.EXIT_SEPS):
            x = self._sep(x, s, lazy=i + 1 < len(S.EXIT_SEPS))
        e = x
        h16, w16 = e.H, e.W
        # end of the Xception encoder (deeplab_xception.py:195-242) in the forward program / its share of the backward makers:
        # bench.py times the region on its own (north_star quotes a roofline fraction for the encoder)
        self._enc_fwd_end = len(self.fwd_train)
        self._enc_bwd_makers = len(self.bwd)

        # ---- ASPP: five branches write channel slices of one buffer (torch.cat is free)
        cat1 = Act(self, "aspp_cat", B, h16, w16, 1280)
        small_begin = (len(self.fwd_train), len(self.fwd_eval))     # the image-pool and 1 x 1 branches: see _fork_forward below
        # image-pool branch first, so that in the backward program it is the LAST contributor to d(e)
        pooled = Act(self, "gap", B, 1, 1, 2048, dtype=torch.float32)     # this branch runs in fp32 (B values per channel)
        HW = h16 * w16

        def pool_fwd():
            L.call("dc_avgpool_fwd", self.dt, B, HW, 2048, e.ptr, e.ld, pooled.ptr, self._st())

        self.fwd_train.append(pool_fwd)
        self.fwd_eval.append(pool_fwd)

        def pool_bwd_make():
            dg = pooled.grad
            mode = e.take_grad_mode()
            assert mode == 1, "avg-pool backward accumulates: another branch must have written d(e) first"
            de = e.grad

            def bwd():
                L.call("dc_avgpool_bwd_add", self.dt, B, HW, 2048, dg.ptr, de.ptr, de.ld, self._st())
            return bwd, []

        self.bwd.append(pool_bwd_make)
        yg, slab, rows = self._conv(pooled, "global_avg_pool.1.weight", 256, name="gap.conv", f32=True)
        ag = self._bn(yg, slab, rows, "global_avg_pool.2", True)
        bslice = cat1.slice("aspp5", 1024, 256)

        def bc_fwd():
            L.call("dc_broadcast_hw", self.dt, B, HW, 256, ag.ptr, bslice.ptr, bslice.ld, self._st())

        self.fwd_train.append(bc_fwd)
        self.fwd_eval.append(bc_fwd)

        def bc_bwd_make():
            ds = bslice.grad
            assert ag.take_grad_mode() == 0
            dag = ag.grad

            def bwd():
                L.call("dc_sum_hw", self.dt, B, HW, 256, ds.ptr, ds.ld, dag.ptr, self._st())
            return bwd, []

        self.bwd.append(bc_bwd_make)
        # The atrous branches share ONE forward launch (dc_conv_fwd_dilated_group): each is 108 tiles of the 256-tile kernel at local
        # batch 8 and 27 at batch 2, on 256 CUs.  (DC_ASPP_GROUP=0: one launch per branch.)
        group = [] if self.aspp_group else None
        pending = []
        for i, rate in enumerate(S.ASPP_RATES, start=1):
            k, pad = (1, 0) if rate == 1 else (3, rate)
            y, slab, rows = self._conv(e, f"aspp{i}.atrous_convolution.weight", 256, k=k, pad=pad, dil=rate, name=f"aspp{i}.conv",
                                       fwd_group=group if rate != 1 else None)
            if group is not None and rate != 1:
                pending.append((y, slab, rows, i))
            else:
                self._bn(y, slab, rows, f"aspp{i}.bn", True, out=cat1.slice(f"aspp{i}", 256 * (i - 1), 256))
            if rate == 1 and group is not None and self.aspp_side:
                # The image-pool branch (five launches on B x 2048 values), the 1 x 1 branch and the decoder's low-level projection (it reads
                # block1's output, ready long before) are ~200 us of small kernels; the grouped atrous launch behind them is 324 tiles on
                # 256 CUs, i.e. a second round with 188 CUs idle.  Forward only: the small branches go to the (otherwise idle) side stream
                # and run in that shadow; the projection conv joins them.
                cat2 = Act(self, "dec_cat", B, H // 4, W // 4, 304)
                yl, slab, rows = self._conv(low, "conv2.weight", 48, name="lowproj")
                self._bn(yl, slab, rows, "bn2", True, out=cat2.slice("low48", 256, 48))
                self._fork_forward(small_begin)
        if group:
            self._dilated_group_fwd(group)
            for y, slab, rows, i in pending:
                self._bn(y, slab, rows, f"aspp{i}.bn", True, out=cat1.slice(f"aspp{i}", 256 * (i - 1), 256))
        if getattr(self, "_forked", False):
            join = lambda: L.call("dc_stream_fence", C.c_void_p(self.side.cuda_stream), L.stream_ptr())   # noqa: E731
            self.fwd_train.append(join)
            self.fwd_eval.append(join)
        y, slab, rows = self._conv(cat1, "conv1.weight", 256, name="proj")
        p = self._bn(y, slab, rows, "bn1", True)

        # ---- decoder
        if not getattr(self, "_forked", False):
            cat2 = Act(self, "dec_cat", B, H // 4, W // 4, 304)
            yl, slab, rows = self._conv(low, "conv2.weight", 48, name="lowproj")
            self._bn(yl, slab, rows, "bn2", True, out=cat2.slice("low48", 256, 48))
        U = "upsample."
        y, slab, rows = self._conv(p, U + "deconv1.0.weight", 256, transposed=True, name="deconv1", sole_consumer=True)
        a = self._bn(y, slab, rows, U + "deconv1.1", True)
        y, slab, rows = self._conv(a, U + "deconv2.0.weight", 256, transposed=True, name="deconv2", sole_consumer=True)
        self._bn(y, slab, rows, U + "deconv2.1", True, out=cat2.slice("up256", 0, 256))
        y, slab, rows = self._conv(cat2, U + "conv1.0.weight", 256, k=3, pad=1, name="dec.conv0")
        a = self._bn(y, slab, rows, U + "conv1.1", True)
        y, slab, rows = self._conv(a, U + "conv1.3.weight", 256, k=3, pad=1, name="dec.conv3", sole_consumer=True)
        a = self._bn(y, slab, rows, U + "conv1.4", True)
        y, _, _ = self._conv(a, U + "conv1.6.weight", 256, stats=False, bias=U + "conv1.6.bias", name="dec.conv6", sole_consumer=True)
        y, slab, rows = self._conv(y, U + "deconv3.0.weight", 256, transposed=True, name="deconv3")
        a = self._bn(y, slab, rows, U + "deconv3.1", True, lazy=self.fuse_bn_into_head)
        lazy_in = isinstance(a, LazyAct)

        # ---- classifier head -> NCHW fp32 logits (GEMM + sub-pixel tap combination, own workspace: P must survive to bwd? no,
        #      it is recomputed from dlogits; the workspace only has to be private to the head)
        wl = U + "last_deconv.0.weight"
        hws = torch.empty(lib.dc_head_workspace(self.dt, B, 256, a.H, a.W) + 256, dtype=torch.uint8, device=self.device)
        self._keep.append(hws)
        hptr = C.c_void_p((hws.data_ptr() + 255) // 256 * 256)

        def head_fwd():
            if lazy_in:
                L.call("dc_head_fwd_bnin", self.dt, B, 256, a.H, a.W, a.y.ptr, a.y.ld, L.dptr(a.scale), L.dptr(a.shift), int(a.relu), self.pptr(wl),
                       L.dptr(self.logits), hptr, self._st())
                return
            L.call("dc_head_fwd", self.dt, B, 256, a.H, a.W, a.ptr, a.ld, self.pptr(wl), L.dptr(self.logits), hptr, self._st())

        def head_fwd_train():
            # a fused train step (nn.TrainStep) hands the loss arguments over: the classifier's kernel then computes the weighted
            # cross-entropy, its gradient, the argmax and the IoU counts on the logits it has in registers (dc_head_fwd_loss)
            la = self.loss_args
            if la is None:
                return head_fwd()
            if lazy_in:
                L.call("dc_head_fwd_loss_bnin", self.dt, B, 256, a.H, a.W, a.y.ptr, a.y.ld, L.dptr(a.scale), L.dptr(a.shift), int(a.relu),
                       self.pptr(wl), L.dptr(self.logits) if la["store_logits"] else None, hptr, L.dptr(la["labels"]), la["labels"].element_size(),
                       L.dptr(la["weight"]), la["grad_scale"], L.dptr(la["loss_sum"]), L.dptr(self.dlogits), L.dptr(la["pred"]),
                       L.dptr(la["counts"]), self._st())
                return
            L.call("dc_head_fwd_loss", self.dt, B, 256, a.H, a.W, a.ptr, a.ld, self.pptr(wl), L.dptr(self.logits) if la["store_logits"] else None,
                   hptr, L.dptr(la["labels"]), la["labels"].element_size(), L.dptr(la["weight"]), la["grad_scale"], L.dptr(la["loss_sum"]),
                   L.dptr(self.dlogits), L.dptr(la["pred"]), L.dptr(la["counts"]), self._st())

        self.fwd_train.append(head_fwd_train)
        self.fwd_eval.append(head_fwd)

        def head_bwd_make():
            assert a.take_grad_mode() == 0
            bsrc = getattr(a, "bn_src", None) if self.fuse_bn_head else None
            two_pass = lazy_in and self.fuse_bn_head and self.fuse_head_apply and self.dt == L.DC_BF16
            da = None if two_pass else a.grad
            if lazy_in:
                bsrc = None
                sslab = None
                if self.fuse_bn_head:
                    srows = (a.M + 127) // 128
                    sslab = self._f32(2 * srows * 256)
                    a.fused_bwd = (sslab, srows)
                if two_pass:
                    def head_apply(gam, dgam, dbet, dy):       # called by the BatchNorm's backward behind its dc_bn_bwd_finalize
                        L.call("dc_head_bwd_bnin_apply", self.dt, B, 256, a.H, a.W, a.y.ptr, a.y.ld, L.dptr(a.scale), L.dptr(a.shift), int(a.relu),
                               gam, L.dptr(a.mean), L.dptr(a.invstd), dgam, dbet, a.M, dy.ptr, dy.ld, hptr, self._st())
                    a.apply_by = head_apply
            elif bsrc is not None:
                srows = (a.M + 127) // 128
                sslab = self._f32(2 * srows * 256)
                a.fused_bwd = (sslab, srows)
                by = bsrc["y"]

            def bwd():
                if lazy_in:
                    # gathered gradient image + data gradient (+ the BatchNorm's sums) on the chain, the weight gradient (0.35 ms at local
                    # batch 8) behind it on the weight-gradient stream like every other one
                    def call(parts):
                        L.call("dc_head_bwd_bnin", self.dt, B, 256, a.H, a.W, a.y.ptr, a.y.ld, L.dptr(a.scale), L.dptr(a.shift), int(a.relu),
                               L.dptr(self.dlogits), self.pptr(wl), da.ptr if da is not None else None, da.ld if da is not None else 0,
                               self.gptr(wl), hptr, L.dptr(a.mean), L.dptr(a.invstd),
                               L.dptr(sslab) if sslab is not None else None, parts, self._st())
                    if two_pass and self.fuse_head_wgrad:
                        # the statistics pass (dx is not stored) has dP and y of every pixel in registers: the weight gradient rides on it
                        # instead of re-reading the 906 MB BatchNorm input in a pass of its own
                        call(3)
                        return
                    call(1)
                    self._on_side(lambda ws_: call(2))
                    return
                if bsrc is not None:
                    L.call("dc_head_bwd_bnstats", self.dt, B, 256, a.H, a.W, a.ptr, a.ld, L.dptr(self.dlogits), self.pptr(wl), da.ptr, da.ld,
                           self.gptr(wl), hptr, by.ptr, by.ld, L.dptr(bsrc["mean"]), L.dptr(bsrc["invstd"]), L.dptr(bsrc["scale"]),
                           L.dptr(bsrc["shift"]), bsrc["relu"], L.dptr(sslab), self._st())
                    return
                L.call("dc_head_bwd", self.dt, B, 256, a.H, a.W, a.ptr, a.ld, L.dptr(self.dlogits), self.pptr(wl), da.ptr, da.ld,
                       self.gptr(wl), hptr, self._st())
            return bwd, [wl]

        self.bwd.append(head_bwd_make)

    def _resolve_backward(self) -> None:
        """Resolve the backward program in reverse order (fixes write/accumulate modes of every gradient)."""
        makers = self.bwd
        self.bwd = []
        self.grad_ready = []
        for mk in reversed(makers):
            fn, ready = mk()
            self.bwd.append(fn)
            self.grad_ready.append(ready)
        self._plan_folds()
        seen = [n for r in self.grad_ready for n in r] + self._deferred_names
        assert sorted(seen) == sorted(self.layout.params), "every parameter must receive its gradient exactly once"

    def _plan_folds(self) -> None:
        """Fixes how every dense weight gradient is launched and folded (the backward program is static, so this runs once).
        Runs of consecutive dense layers with one geometry (in backward order: the three 728 -> 728 pointwise convs of a middle-flow
        Block) become groups of up to `wgrad_group` layers: all but the last member defer, the last one launches the group
        (dc_conv_wgrad_partial) and reports every member's gradient as ready.  Every launch is followed, on the same stream, by one
        dc_fold_slabs over its own slabs plus the rows left by the depthwise layers whose data-gradient kernels were enqueued since the
        previous fold (their names move to this op's ready list: on_grad_ready feeds the all-reduce buckets)."""
        lib = L.load()
        G = self.wgrad_group
        recs = self._wg_recs            # make_bwd ran in backward order, so this list is in execution order
        i = 0
        while i < len(recs):
            j = i
            r0 = recs[i]
            groupable = G > 1 and not r0["bias"] and self.dt == L.DC_BF16 and r0["key"][0] == L.DC_BF16
            while groupable and j + 1 < len(recs) and j + 1 - i < G and recs[j + 1]["key"] == r0["key"] and not recs[j + 1]["bias"]:
                j += 1
            d, x = r0["d"], r0["x"]
            splits, sbytes = C.c_int(), C.c_size_t()
            while j > i and lib.dc_conv_wgrad_plan(C.byref(d), x.N, x.H, x.W, j + 1 - i, C.byref(splits), C.byref(sbytes)) != 0:
                j -= 1                                  # the grouped launch serves fewer layers of this geometry (or none: one launch per layer)
            members = recs[i:j + 1]
            cnt = len(members)
            L.call("dc_conv_wgrad_plan", C.byref(d), x.N, x.H, x.W, cnt, C.byref(splits), C.byref(sbytes))
            slabs = [torch.empty(max(sbytes.value, 16), dtype=torch.uint8, device=self.device) for _ in members]
            self._keep += slabs
            last = members[-1]
            for m in members[:-1]:
                m["role"] = "defer"
                m["ready"].remove(m["wname"])
                last["ready"].append(m["wname"])
            last["role"] = "flush" if cnt > 1 else "solo"
            last["members"] = members
            last["launch"] = (cnt, (C.c_void_p * cnt)(*[m["x"].ptr.value for m in members]),
                              (C.c_void_p * cnt)(*[m["dy"].ptr.value for m in members]),
                              (C.c_void_p * cnt)(*[t.data_ptr() for t in slabs]), splits.value)
            kind = L.DC_FOLD_CONVT if d.transposed else L.DC_FOLD_CONV
            last["own_entries"] = [L.FoldEntry(t.data_ptr(), m["gw"].value, kind, splits.value, d.k * d.k, d.cout, d.cin)
                                   for m, t in zip(members, slabs)]
            i = j + 1
        pending: List[tuple] = []
        for item in self._fold_seq:
            if item[0] == "dw":
                pending.append(item)
                continue
            rec = item[1]
            if rec["role"] == "defer":
                continue
            ents = [it[1] for it in pending] + rec["own_entries"]
            rec["fold"] = ((L.FoldEntry * len(ents))(*ents), len(ents))
            for _, _, name, ready in pending:
                ready.remove(name)
                rec["ready"].append(name)
            pending = []
        if pending:       # depthwise layers behind the last dense launch of the program (the entry flow's first separable convs)
            ents = [it[1] for it in pending]
            for _, _, name, ready in pending:
                ready.remove(name)
                self._deferred_names.append(name)
            self._final_fold = ((L.FoldEntry * len(ents))(*ents), len(ents), [it[2] for it in pending])

    # ------------------------------------------------------------------------------------------------ execution
    def pack_weights(self) -> None:
        for op in self.pack_ops:
            op()
        self.packed_version = self.version[0]

    def mark_weights_changed(self) -> None:
        self.version[0] += 1

    def forward(self, x_nchw: torch.Tensor, train: bool = True) -> torch.Tensor:
        """logits[B,3,H,W] (fp32, NCHW) = net(x); returns the engine-owned logits buffer.
        x is the reference's NCHW fp32 batch [B,16,H,W], or -- from the input pipeline (data.py) -- an NHWC batch
        [B,H,W,16] already in the activation dtype, which is then used in place (no layout pass)."""
        if (self.x0 is not None and x_nchw.dim() == 4 and tuple(x_nchw.shape) == (self.B, self.H, self.W, self.n_input)
                and x_nchw.dtype == self.dtype and x_nchw.is_contiguous() and x_nchw.device == self.device):
            self.x0.buf = x_nchw
            self.x_in = None
            return self._run_forward(train)
        if self.x0 is not None:
            self.x0.buf = self._x0_own
        if tuple(x_nchw.shape) != (self.B, self.n_input, self.H, self.W):
            raise L.DeepcamHipError(f"engine built for input {(self.B, self.n_input, self.H, self.W)}, got {tuple(x_nchw.shape)}")
        if train and self.B * (self.H // 16) * (self.W // 16) < 1:
            raise ValueError("empty batch")
        if train and self.B < 2:
            # the image-pool BatchNorm sees B values per channel (deeplab_xception.py:425-428,449)
            raise ValueError("Expected more than 1 value per channel when training, got input size torch.Size([1, 256, 1, 1])")
        if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or x_nchw.device != self.device:
            x_nchw = x_nchw.to(device=self.device, dtype=torch.float32).contiguous()
        self.x_in = x_nchw
        return self._run_forward(train)

    def _run_forward(self, train: bool) -> torch.Tensor:
        if train and self.B < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size torch.Size([1, 256, 1, 1])")
        if self.packed_version != self.version[0]:
            if (self.pack_side and self.use_side_stream and self.x_in is not None and getattr(self, "_layout_op", None) is not None):
                # the layout pass of the batch (NCHW fp32 -> NHWC) does not need the weights: it runs on the side stream beside the repack
                # (both are short memory-bound kernels that leave the other room); layout_fwd joins
                L.call("dc_stream_fence", L.stream_ptr(), C.c_void_p(self.side.cuda_stream))
                with torch.cuda.stream(self.side):
                    self._layout_op()
                self._layout_forked = True
            self.pack_weights()
        marks = self.region_marks if train else None      # measurement hook: events at the encoder's boundaries
        if marks is not None:
            marks.append(("fwd_begin", torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))))
        if train and self._sum_used:
            L.call("dc_memset_async", L.dptr(self._sum_arena), 0, 4 * self._sum_used, self._st())      # every BatchNorm sum row of the step
        for i, op in enumerate(self.fwd_train if train else self.fwd_eval):
            if marks is not None and i == self._enc_fwd_end:
                marks.append(("fwd_enc_end", torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))))
            op()
        return self.logits

    def backward(self) -> None:
        """Consumes self.dlogits (NCHW fp32); leaves every parameter gradient in self.grads."""
        cb = self.on_grad_ready
        marks = self.region_marks
        enc_begin = len(self.bwd) - self._enc_bwd_makers      # the backward program is the reversed maker list
        for i, (op, ready) in enumerate(zip(self.bwd, self.grad_ready)):
            if marks is not None and i == enc_begin:
                marks.append(("bwd_enc_begin", torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))))
            op()
            if self._ready_now:           # gradients whose (deferred) kernels this op has just submitted
                ready, self._ready_now = ready + self._ready_now, []
            if cb is not None and ready:
                cb(ready)
        if self._final_fold is not None:
            ents, nent, names = self._final_fold
            self._on_side(lambda ws: L.call("dc_fold_slabs", ents, nent, self._st()))
            self._ready_now += names
        if self._ready_now:
            ready, self._ready_now = self._ready_now, []
            if cb is not None:
                cb(ready)
        if self.use_side_stream:
            L.call("dc_stream_fence", C.c_void_p(self.side.cuda_stream), L.stream_ptr())
        if marks is not None:
            marks.append(("bwd_end", torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))))
