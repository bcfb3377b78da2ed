import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
cin = cout = int(sys.argv[1]); N, H, W = 8, 48, 72; ld = (cin + 31)//32*32
desc = L.ConvDesc(L.DC_BF16, 1,1,0,1,0,cin,cout)
x = torch.randn(N,H,W,ld, device=dev).to(dt); wf = torch.randn(cout*((cin+63)//64*64), device=dev).to(dt) * 0.05
y = torch.empty(N,H,W,ld, device=dev, dtype=dt)
rows = L.load().dc_conv_stat_rows(C.byref(desc), N,H,W); slab = torch.empty(2*rows*cout, device=dev)
for _ in range(5): L.call("dc_conv_fwd", C.byref(desc), N,H,W, L.dptr(x), ld, L.dptr(wf), None, L.dptr(y), ld, L.dptr(slab), 0, L.stream_ptr())
torch.cuda.synchronize()
