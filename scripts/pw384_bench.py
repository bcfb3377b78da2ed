"""Pointwise-conv GEMMs on the one-wave-per-SIMD kernel (csrc/igemm384.hip) against the planner's previous choice (256 x 256 eight-wave
or 128 x 128 four-wave tiles): time per call, outputs bit-compared, BatchNorm slabs compared.  python scripts/pw384_bench.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
#          cin   cout  N  H    W
shapes = [(728, 728, 2, 48, 72), (728, 728, 4, 48, 72)] if len(sys.argv) > 1 and sys.argv[1] == "small" else [(728, 728, 8, 48, 72), (728, 728, 4, 48, 72), (728, 728, 2, 48, 72), (728, 1024, 8, 48, 72), (1024, 1536, 8, 48, 72),
          (1536, 1536, 8, 48, 72), (1536, 2048, 8, 48, 72), (256, 728, 8, 96, 144), (728, 728, 8, 96, 144), (256, 256, 8, 192, 288),
          (1280, 256, 8, 48, 72), (2048, 256, 8, 48, 72), (128, 128, 8, 384, 576), (728, 728, 3, 47, 71)]
lib = L.load()
for (cin, cout, N, H, W) in shapes:
    desc = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    wf = (torch.randn(cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    outs, res = [], []
    for mode in (0, 2, 3, 4, 5):
        L.call("dc_set_option", b"pw384", mode)
        y = torch.zeros(N, H, W, r32(cout), device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
        once = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(wf), None, L.dptr(y), r32(cout), L.dptr(slab), 0, L.stream_ptr())
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): once()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append(f"{us:7.1f} us {2.0 * N * H * W * cin * cout / us / 1e6:6.0f} TF")
        outs.append((y[..., :cout].clone(), slab.clone()))
    eq = [torch.equal(outs[0][0], o[0]) for o in outs[1:]]
    ds = [(outs[0][1] - o[1]).abs().max().item() / (outs[0][1].abs().max().item() + 1e-30) for o in outs[1:]]
    print(f"{cin:4d}->{cout:4d} M={N*H*W:7d}: planner(old) {res[0]} | 256x384 {res[1]} | 128x384 {res[2]} | 256x384 K64 {res[3]} | 128x192 {res[4]} | y bit-equal {eq}, slab rel diff {max(ds):.1e}", flush=True)
L.call("dc_set_option", b"pw384", 1)
