"""A/B of the one-pass thin-conv weight gradient against the tiled GEMM kernels on the two stem layers (full size, B=8)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
lib = L.load()
for (k, s, p, d, tr, cin, cout, N, H, W) in [(3, 2, 1, 1, 0, 16, 32, 8, 768, 1152), (3, 1, 1, 1, 0, 32, 64, 8, 384, 576)]:
    desc = L.ConvDesc(L.DC_BF16, k, s, p, d, tr, cin, cout)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(N, H, W, cin, device=dev).to(dt); dy = torch.randn(N, Ho, Wo, cout, device=dev).to(dt)
    wsb = lib.dc_conv_wgrad_workspace(C.byref(desc), N, H, W); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    outs, res = [], []
    for mode in (0, 1):
        L.call("dc_set_option", b"thin_wgrad", mode)
        gw = torch.zeros(cout * cin * 9, device=dev)
        once = lambda: L.call("dc_conv_wgrad", C.byref(desc), N, H, W, L.dptr(x), cin, L.dptr(dy), cout, L.dptr(ws), wsb, L.dptr(gw), L.stream_ptr())
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): once()
        e1.record(); torch.cuda.synchronize()
        res.append(f"{e0.elapsed_time(e1) / 10 * 1e3:8.1f} us"); outs.append(gw.clone())
    rel = (outs[0] - outs[1]).abs().max().item() / (outs[0].abs().max().item() + 1e-30)
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    print(f"k3 s{s} {cin}->{cout} @{H}x{W}: tiled {res[0]} | one-pass {res[1]} | max rel diff {rel:.1e} | x+dy = {mb:.0f} MB")
L.call("dc_set_option", b"thin_wgrad", 1)
