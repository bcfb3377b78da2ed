"""Mixed tile plan (whole 256-tile rounds + the rest on the 128-tile kernel) against the single 256-tile launch on the layers
whose last round is mostly empty; forward and data gradient, bit-compared.  python scripts/gemm_mix_bench.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
#          k  s  p  d  tr cin   cout  N  H    W
shapes = [(1, 1, 0, 1, 0, 728,  728,  8, 48,  72), (1, 1, 0, 1, 0, 1536, 2048, 8, 48, 72), (1, 1, 0, 1, 0, 728, 728, 8, 96, 144),
          (1, 1, 0, 1, 0, 728, 1024, 8, 48, 72), (1, 1, 0, 1, 0, 1024, 1536, 8, 48, 72), (1, 1, 0, 1, 0, 1536, 1536, 8, 48, 72)]
lib = L.load()
for (k, s, p, d, tr, cin, cout, N, H, W) in shapes:
    desc = L.ConvDesc(L.DC_BF16, k, s, p, d, tr, cin, cout)
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    wf = (torch.randn(cout * r32(cin), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    outs, res = [], []
    for mix, pct in ((0, 40), (1, 40), (1, 60), (1, 80)):
        L.call("dc_set_option", b"igemm_mix", mix); L.call("dc_set_option", b"igemm_mix_tail", pct)
        y = torch.zeros(N, H, W, r32(cout), device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
        once = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(wf), None, L.dptr(y), r32(cout), L.dptr(slab), 0, L.stream_ptr())
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): once()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append(f"{us:7.1f} us {2.0 * N * H * W * cin * cout / us / 1e6:6.1f} TF")
        outs.append((y[..., :cout].clone(), slab.clone()))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    print(f"{cin:4d}->{cout:4d} @{H}x{W}: single {res[0]} | mix<=40% {res[1]} | <=60% {res[2]} | <=80% {res[3]} | bit-equal {same}")
L.call("dc_set_option", b"igemm_mix", 1); L.call("dc_set_option", b"igemm_mix_tail", 40)
