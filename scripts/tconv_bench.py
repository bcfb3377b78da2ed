"""Transposed-conv forward / strided data gradient on the 256-tile kernel: tile order A/B (igemm256_phase_fast) in one job.
python scripts/tconv_bench.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
lib = L.load()
for (H, W) in [(192, 288), (96, 144), (48, 72)]:
    N, cin, cout = 8, 256, 256
    desc = L.ConvDesc(L.DC_BF16, 3, 2, 1, 1, 1, cin, cout)
    x = torch.randn(N, H, W, cin, device=dev).to(dt)
    wf = (torch.randn(9 * cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    res, outs = [], []
    for pf in (0, 1, 0, 1):
        L.call("dc_set_option", b"igemm256_phase_fast", pf)
        y = torch.zeros(N, 2 * H, 2 * W, cout, device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
        once = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), cin, L.dptr(wf), None, L.dptr(y), cout, L.dptr(slab), 0, L.stream_ptr())
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): once()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3); outs.append((y.clone(), slab.clone()))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    print(f"convT 256->256 @{H}x{W}: phase-major {res[0]:.1f} / {res[2]:.1f} us, phase-fastest {res[1]:.1f} / {res[3]:.1f} us, same bits {same}")
L.call("dc_set_option", b"igemm256_phase_fast", 1)
