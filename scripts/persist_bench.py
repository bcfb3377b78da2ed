"""A/B of the 256 x 256 implicit-GEMM kernel, one tile per workgroup (igemm256.hip) against the persistent form (igemm256p.hip, a
workgroup per CU walking its tiles), on the network's multi-round forward shapes; outputs bit-compared on the way.
python scripts/persist_bench.py [workgroups ...]      (-1: the library's own choice)"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
WGS = [int(v) for v in sys.argv[1:]] or [256]
#          k  s  p   d  tr cin   cout  N  H    W
shapes = [(1, 1, 0,  1, 0, 728,  728,  8, 48,  72), (1, 1, 0, 1, 0, 1536, 2048, 8, 48, 72), (3, 1, 6,  6, 0, 2048, 256,  8, 48,  72),
          (3, 1, 1,  1, 0, 256,  256,  8, 192, 288), (3, 1, 1, 1, 0, 304, 256, 8, 192, 288), (1, 1, 0, 1, 0, 256, 256, 8, 192, 288),
          (3, 2, 1,  1, 1, 256,  256,  8, 96,  144), (3, 2, 1, 1, 1, 256, 256, 8, 192, 288), (1, 1, 0, 1, 0, 728, 728, 8, 96, 144),
          (1, 1, 0, 1, 0, 128, 256, 8, 192, 288), (3, 1, 1, 1, 0, 128, 128, 8, 384, 576)]
lib = L.load()
L.call("dc_set_option", b"igemm256", 2)
L.call("dc_set_option", b"pw384", 0)
L.call("dc_set_option", b"igemm256p_min", 1)
for (k, s, p, d, tr, cin, cout, N, H, W) in shapes:
    desc = L.ConvDesc(L.DC_BF16, k, s, p, d, tr, cin, cout)
    kk = 9 if tr else k * k
    Ho, Wo = C.c_int(), C.c_int(); L.call("dc_conv_out_hw", C.byref(desc), H, W, C.byref(Ho), C.byref(Wo)); Ho, Wo = Ho.value, Wo.value
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    wf = (torch.randn(kk * cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    outs, res = [], []
    for mode in [0] + WGS:
        L.call("dc_set_option", b"igemm256p", 1 if mode else 0)
        if mode: L.call("dc_set_option", b"igemm256p_wgs", max(mode, 0))
        y = torch.zeros(N, Ho, Wo, r32(cout), device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
        once = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(wf), None, L.dptr(y), r32(cout), L.dptr(slab), 0, L.stream_ptr())
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): once()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append(f"{'1/wg' if not mode else 'k128B' if mode == -2 else 'auto' if mode < 0 else 'p%d' % mode} {us:7.1f} us {2.0 * N * Ho * Wo * cin * cout * kk / (4 if tr else 1) / us / 1e6:6.1f} TF")
        outs.append((y[..., :cout].clone(), slab.clone()))
    same = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
    dslab = max((outs[0][1] - o[1]).abs().max().item() for o in outs[1:]) / (outs[0][1].abs().max().item() + 1e-30)
    tiles = ((cout + 255) // 256) * ((N * Ho * Wo // (4 if tr else 1) + 255) // 256) * (4 if tr else 1)
    print(f"k{k}s{s}d{d}{'T' if tr else ' '} {cin:4d}->{cout:4d} @{H}x{W} {tiles:5d} tiles: " + " | ".join(res) + f" | bit-equal {same}, slab rel diff {dslab:.1e}", flush=True)
