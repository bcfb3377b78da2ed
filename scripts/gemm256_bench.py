"""A/B of the two implicit-GEMM tile shapes (128 x 128 four-wave kernel vs 256 x 256 eight-wave kernel) on the network's
forward / data-gradient shapes, bit-compared on the way.  python scripts/gemm256_bench.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
#          k  s  p   d  tr cin   cout  N  H    W
shapes = [(1, 1, 0,  1, 0, 728,  728,  8, 48,  72), (1, 1, 0, 1, 0, 728, 1024, 8, 48, 72), (1, 1, 0, 1, 0, 1536, 2048, 8, 48, 72),
          (3, 1, 6,  6, 0, 2048, 256,  8, 48,  72), (1, 1, 0, 1, 0, 1280, 256, 8, 48, 72),
          (3, 1, 1,  1, 0, 256,  256,  8, 192, 288), (3, 1, 1, 1, 0, 304, 256, 8, 192, 288), (1, 1, 0, 1, 0, 256, 256, 8, 192, 288),
          (3, 2, 1,  1, 1, 256,  256,  8, 96,  144), (3, 2, 1, 1, 1, 256, 256, 8, 192, 288), (3, 2, 1, 1, 1, 256, 256, 8, 48, 72), (1, 1, 0, 1, 0, 2048, 256, 8, 48, 72), (1, 1, 0, 1, 0, 728, 728, 8, 96, 144), (1, 1, 0, 1, 0, 128, 128, 8, 384, 576)]
lib = L.load()
for (k, s, p, d, tr, cin, cout, N, H, W) in shapes:
    desc = L.ConvDesc(L.DC_BF16, k, s, p, d, tr, cin, cout)
    kk = 9 if tr else k * k
    Ho, Wo = C.c_int(), C.c_int(); L.call("dc_conv_out_hw", C.byref(desc), H, W, C.byref(Ho), C.byref(Wo)); Ho, Wo = Ho.value, Wo.value
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    wf = (torch.randn(kk * cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    outs, res = [], []
    for mode in (0, 2):
        L.call("dc_set_option", b"igemm256", mode)
        y = torch.zeros(N, Ho, Wo, r32(cout), device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
        once = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(wf), None, L.dptr(y), r32(cout), L.dptr(slab), 0, L.stream_ptr())
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): once()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append(f"{us:8.1f} us {2.0 * N * Ho * Wo * cin * cout * kk / (4 if tr else 1) / us / 1e6:7.1f} TF")
        outs.append((y[..., :cout].clone(), slab.clone()))
    same_y = torch.equal(outs[0][0], outs[1][0])
    dslab = (outs[0][1] - outs[1][1]).abs().max().item() / (outs[0][1].abs().max().item() + 1e-30)
    print(f"k{k}s{s}d{d}{'T' if tr else ' '} {cin:4d}->{cout:4d} @{H}x{W}: 128-tile {res[0]} | 256-tile {res[1]} | y bit-equal {same_y}, slab rel diff {dslab:.1e}")
L.call("dc_set_option", b"igemm256", 0)
