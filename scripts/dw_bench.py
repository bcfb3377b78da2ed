"""Depthwise 3x3 entry points on the middle-flow shape (728 channels, 48 x 72, local batch 8) and two others: the persistent pipelined
kernel (dwpipe.hip) against the tiled one (option dw_pipe = 0), back to back on distinct buffers.   python scripts/dw_bench.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = L.DC_BF16; lib = L.load(); st = L.stream_ptr(); P = L.dptr
NB, REPS = 4, 40
for (Cc, H, W, N, dil) in [(728, 48, 72, 8, 1), (256, 192, 288, 8, 1), (128, 384, 576, 8, 1), (1536, 48, 72, 8, 2)]:
    ld = (Cc + 63) // 64 * 64; M = N * H * W; T = M * ld * 2
    act = lambda: [torch.randn(N, H, W, ld, device=dev).to(torch.bfloat16) for _ in range(NB)]
    x, y, dy, dx, add = act(), act(), act(), act(), act()
    wp = torch.randn(9 * Cc, device=dev) * 0.2
    sc, sh, mean, invstd = [torch.rand(Cc, device=dev) + 0.5 for _ in range(4)]
    print(f"C={Cc} {H}x{W} N={N} dil={dil}: {T / 1e6:.1f} MB per tensor")
    for mode in (0, 1, 2):
        L.call("dc_set_option", b"dw_pipe", mode)
        rows = lib.dc_dwconv_dgrad_bnstats_rows(dt, Cc, 1, dil, N, H, W); wrows = lib.dc_dwconv_dgrad_wgrad_rows(dt, Cc, 1, dil, N, H, W)
        slab = torch.empty(2 * rows * Cc, device=dev); wslab = torch.empty(max(wrows, 1) * 9 * Cc, device=dev)
        fns = {
            "fwd": (lambda i: L.call("dc_dwconv_fwd", dt, Cc, 1, dil, N, H, W, P(x[i]), ld, P(wp), P(y[i]), ld, None, None, 0, st), 2 * T),
            "fwd (BN+ReLU on load)": (lambda i: L.call("dc_dwconv_fwd", dt, Cc, 1, dil, N, H, W, P(x[i]), ld, P(wp), P(y[i]), ld, P(sc), P(sh), 1, st), 2 * T),
            "dgrad": (lambda i: L.call("dc_dwconv_dgrad", dt, Cc, 1, dil, N, H, W, P(dy[i]), ld, P(wp), None, 0, P(dx[i]), ld, st), 2 * T),
            "dgrad + BN sums": (lambda i: L.call("dc_dwconv_dgrad_bnstats", dt, Cc, 1, dil, N, H, W, P(dy[i]), ld, P(wp), P(dx[i]), ld, P(y[i]), ld,
                                                 P(mean), P(invstd), P(sc), P(sh), 1, P(slab), st), 3 * T),
            "dgrad + BN sums + wgrad": (lambda i: L.call("dc_dwconv_dgrad_bnstats_wgrad", dt, Cc, 1, dil, N, H, W, P(dy[i]), ld, P(wp), P(dx[i]), ld,
                                                         P(y[i]), ld, P(mean), P(invstd), P(sc), P(sh), 1, P(slab), P(wslab), st), 3 * T),
            "dgrad + addend + wgrad": (lambda i: L.call("dc_dwconv_dgrad_wgrad", dt, Cc, 1, dil, N, H, W, P(dy[i]), ld, P(wp), P(add[i]), ld, P(dx[i]), ld,
                                                        P(x[i]), ld, None, None, 0, P(wslab), st), 4 * T),
        }
        if wrows <= 0:
            fns = {k: v for k, v in fns.items() if "wgrad" not in k}
        for name, (fn, nbytes) in fns.items():
            for i in range(3): fn(i % NB)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(REPS): fn(i % NB)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / REPS
            print(f"  {['tiled    ', 'pipelined', 'pipe+fwd '][mode]} {name:28s} {us:8.1f} us  {nbytes / us / 1e6:5.2f} TB/s  (rows {rows})", flush=True)
    L.call("dc_set_option", b"dw_pipe", 1)
    del x, y, dy, dx, add
