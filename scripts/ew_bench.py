"""True durations of the HBM-bound kernels at one layer shape (default: the middle-flow unit, B=8: M=27648 pixels, C=728).

Each entry point is launched `reps` times back to back between two events (no per-call event overhead), cycling over `nbuf`
buffer sets so that the 40 MB tensors are not simply served from L2.  A device-to-device copy of the same tensor is the practical
ceiling for a 1-read-1-write kernel on this box.

    python scripts/ew_bench.py [C] [H] [W] [N]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L  # noqa: E402

Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 728
H = int(sys.argv[2]) if len(sys.argv) > 2 else 48
W = int(sys.argv[3]) if len(sys.argv) > 3 else 72
N = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda", 0)
dt, dtc = torch.bfloat16, L.DC_BF16
ld = Cc if Cc < 64 else (Cc + 31) // 32 * 32
M = N * H * W
NBUF, REPS = 4, 40
lib = L.load()
if os.environ.get("DW_TPB"):
    L.call("dc_set_option", b"dw_wgrad_tpb", int(os.environ["DW_TPB"]))
if os.environ.get("DW_CG"):
    L.call("dc_set_option", b"dw_cg", int(os.environ["DW_CG"]))
if os.environ.get("BN_CGW"):
    L.call("dc_set_option", b"bn_cgw", int(os.environ["BN_CGW"]))
if os.environ.get("BN_ROWS"):
    L.call("dc_set_option", b"bn_rows", int(os.environ["BN_ROWS"]))
st = L.stream_ptr()


def act():
    return [torch.randn(N, H, W, ld, device=dev).to(dt) for _ in range(NBUF)]


x, y, dy, dx, out = act(), act(), act(), act(), act()
wm = torch.randn(Cc * 9, device=dev) * 0.2
wp = torch.empty(9 * Cc, device=dev)
L.call("dc_dwconv_pack_weights", Cc, L.dptr(wm), L.dptr(wp), st)
ws = torch.empty(lib.dc_dwconv_wgrad_workspace(Cc, N, H, W, 1) // 4 + 64, device=dev)
gw = torch.empty(Cc * 9, device=dev)
scale, shift, mean, invstd, gam, dgam, dbet = [torch.rand(Cc, device=dev) + 0.5 for _ in range(7)]
rows = lib.dc_bn_stat_rows(M)
slab = torch.empty(2 * rows * Cc, device=dev)


def bench(name, fn, bytes_moved):
    for i in range(3):
        fn(i % NBUF)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(REPS):
        fn(i % NBUF)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REPS
    print(f"{name:34s} {us:8.1f} us   {bytes_moved / us / 1e6:6.2f} TB/s  ({bytes_moved / 1e6:.0f} MB algorithmic)")


T = M * ld * 2  # bytes of one activation tensor
print(f"shape N={N} H={H} W={W} C={Cc} ld={ld}: M={M}, {T / 1e6:.1f} MB per tensor")
bench("copy (torch, 1R+1W)", lambda i: out[i].copy_(x[i]), 2 * T)
bench("dc_dwconv_fwd", lambda i: L.call("dc_dwconv_fwd", dtc, Cc, 1, 1, N, H, W, L.dptr(x[i]), ld, L.dptr(wp), L.dptr(y[i]), ld,
                                        None, None, 0, st), 2 * T)
bench("dc_dwconv_dgrad", lambda i: L.call("dc_dwconv_dgrad", dtc, Cc, 1, 1, N, H, W, L.dptr(dy[i]), ld, L.dptr(wp), None, 0,
                                          L.dptr(dx[i]), ld, st), 2 * T)
bench("dc_dwconv_dgrad (+addend)", lambda i: L.call("dc_dwconv_dgrad", dtc, Cc, 1, 1, N, H, W, L.dptr(dy[i]), ld, L.dptr(wp),
                                                     L.dptr(dx[i]), ld, L.dptr(dx[i]), ld, st), 3 * T)
bench("dc_dwconv_wgrad", lambda i: L.call("dc_dwconv_wgrad", dtc, Cc, 1, 1, N, H, W, L.dptr(x[i]), ld, L.dptr(dy[i]), ld, L.dptr(ws),
                                          L.dptr(gw), None, None, 0, st), 2 * T)
bench("dc_bn_stats", lambda i: L.call("dc_bn_stats", dtc, M, Cc, L.dptr(x[i]), ld, L.dptr(slab), st), T)
bench("dc_bn_apply (relu)", lambda i: L.call("dc_bn_apply", dtc, M, Cc, L.dptr(y[i]), ld, L.dptr(scale), L.dptr(shift), None, 0, 1,
                                             L.dptr(out[i]), ld, st), 2 * T)
bench("dc_bn_apply (+res)", lambda i: L.call("dc_bn_apply", dtc, M, Cc, L.dptr(y[i]), ld, L.dptr(scale), L.dptr(shift), L.dptr(x[i]), ld,
                                             0, L.dptr(out[i]), ld, st), 3 * T)
bench("dc_bn_bwd_reduce (mask from y)", lambda i: L.call("dc_bn_bwd_reduce", dtc, M, Cc, L.dptr(dy[i]), ld, L.dptr(y[i]), ld, None, 0, 2,
                                                          L.dptr(mean), L.dptr(invstd), L.dptr(slab), L.dptr(scale), L.dptr(shift), st), 2 * T)
bench("dc_bn_bwd_finalize", lambda i: L.call("dc_bn_bwd_finalize", Cc, L.dptr(slab), rows, L.dptr(dgam), L.dptr(dbet), st), 2 * rows * Cc * 4)
bench("dc_bn_bwd_apply (mask from y)", lambda i: L.call("dc_bn_bwd_apply", dtc, M, Cc, M, L.dptr(dy[i]), ld, L.dptr(y[i]), ld, None, 0, 2,
                                                         L.dptr(gam), L.dptr(mean), L.dptr(invstd), L.dptr(dgam), L.dptr(dbet),
                                                         L.dptr(dx[i]), ld, None, 0, L.dptr(scale), L.dptr(shift), st), 3 * T)
