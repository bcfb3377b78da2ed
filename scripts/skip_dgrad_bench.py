"""Data gradient of the entry flow's 1x1 stride-2 shortcut convs (three of four sub-pixel phases have no taps): time per call, first writer and
accumulate mode.  python scripts/skip_dgrad_bench.py [N]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = lambda t: C.c_void_p(t.data_ptr())
for (cin, cout, H, W, st) in ((64, 128, 384, 576, 2), (64, 128, 192, 288, 1), (128, 256, 192, 288, 2), (256, 728, 96, 144, 2)):
    d = L.ConvDesc(L.DC_BF16, 1, st, 0, 1, 0, cin, cout)
    Ho, Wo = H // st, W // st
    ldx, ldy = (cin + 63) // 64 * 64, (cout + 63) // 64 * 64
    dy = torch.randn(N, Ho, Wo, ldy, device=dev).to(dt)
    dx = torch.zeros(N, H, W, ldx, device=dev, dtype=dt)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
    nwf, nwb = C.c_size_t(), C.c_size_t(); L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf, wb = torch.zeros(nwf.value, dtype=dt, device=dev), torch.zeros(nwb.value, dtype=dt, device=dev)
    L.call("dc_conv_pack_weights", C.byref(d), P(w), P(wf), P(wb), L.stream_ptr())
    res = []
    for acc in (0, 1):
        fn = lambda: L.call("dc_conv_dgrad", C.byref(d), N, H, W, P(dy), ldy, P(wb), P(dx), ldx, acc, L.stream_ptr())
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    mb = lambda t: t.numel() * 2 / 1e6
    print(f"{cout} -> {cin} stride {st} at {N} x {H} x {W}: first writer {res[0]:6.1f} us, accumulate {res[1]:6.1f} us   (dy {mb(dy):.0f} MB, dx {mb(dx):.0f} MB)", flush=True)
