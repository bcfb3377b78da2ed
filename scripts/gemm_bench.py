"""Micro-benchmark of the igemm / wgrad entry points on chosen shapes (interleaved rounds in one process)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
shapes = [(1,1,0,1,0,728,728,8,48,72), (1,1,0,1,0,768,768,8,48,72), (1,1,0,1,0,736,736,8,48,72), (1,1,0,1,0,704,704,8,48,72),
          (1,1,0,1,0,1024,1024,8,48,72), (1,1,0,1,0,256,256,8,192,288), (3,1,1,1,0,256,256,8,192,288)]
def run(kind, k,s,p,d,tr,cin,cout,N,H,W, reps=20):
    desc = L.ConvDesc(L.DC_BF16, k,s,p,d,tr,cin,cout)
    x = torch.randn(N,H,W,cin, device=dev).to(dt); kk = k*k
    wf = torch.randn(kk*cout*((cin+63)//64*64), device=dev).to(dt) * 0.05
    y = torch.empty(N,H,W,cout, device=dev, dtype=dt)
    rows = L.load().dc_conv_stat_rows(C.byref(desc), N,H,W); slab = torch.empty(2*rows*cout, device=dev)
    wsb = L.load().dc_conv_wgrad_workspace(C.byref(desc), N,H,W); ws = torch.empty(max(wsb,16), dtype=torch.uint8, device=dev)
    gw = torch.empty(cout*cin*kk, device=dev)
    st = L.stream_ptr()
    def once():
        if kind == "fwd": L.call("dc_conv_fwd", C.byref(desc), N,H,W, L.dptr(x), cin, L.dptr(wf), None, L.dptr(y), cout, L.dptr(slab), 0, st)
        else: L.call("dc_conv_wgrad", C.byref(desc), N,H,W, L.dptr(x), cin, L.dptr(y), cout, L.dptr(ws), wsb, L.dptr(gw), st)
    for _ in range(3): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): once()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * N*H*W*cin*cout*kk
    return us, fl / us / 1e6
for kind in ("fwd", "wgrad"):
    for sh in shapes:
        res = []
        for mode in ((2, 3, 4) if kind == "fwd" else (2,)):
            L.load().dc_set_option(b"igemm_mode", mode)
            us, tf = run(kind, *sh); res.append(f"mode{mode}: {us:7.1f} us {tf:6.1f} TF")
        print(kind, sh, " | ".join(res))
