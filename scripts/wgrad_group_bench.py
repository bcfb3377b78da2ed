"""Grouped vs per-layer weight gradient of the 728 -> 728 pointwise layers (B=8, 48x72): `layers` plain dc_conv_wgrad calls
against one dc_conv_wgrad_group call, distinct operand tensors per layer (as in the step).
python scripts/wgrad_group_bench.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
lib = L.load()
for (cin, cout, N, H, W) in [(728, 728, 8, 48, 72), (256, 256, 8, 192, 288)]:
    desc = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    NL = 12
    xs = [torch.randn(N, H, W, r32(cin), device=dev).to(dt) for _ in range(NL)]
    dys = [torch.randn(N, H, W, r32(cout), device=dev).to(dt) for _ in range(NL)]
    gws = [torch.zeros(cout * cin, device=dev) for _ in range(NL)]
    for G in (1, 2, 3, 4):
        wsb = lib.dc_conv_wgrad_group_workspace(C.byref(desc), N, H, W, G); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        def once():
            for i in range(0, NL, G):
                L.call("dc_conv_wgrad_group", C.byref(desc), N, H, W, G, pa(xs[i:i + G]), r32(cin), pa(dys[i:i + G]), r32(cout), L.dptr(ws), wsb,
                       pa(gws[i:i + G]), L.stream_ptr())
        for _ in range(2): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): once()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 / NL * 1e3
        print(f"{cin}->{cout} @{H}x{W} group {G}: {us:7.1f} us per layer  {2.0 * N * H * W * cin * cout / us / 1e6:7.1f} TF")
