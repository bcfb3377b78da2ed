"""Where a tile's time goes inside the pipelined depthwise kernel (diagnostic build: make -C mlperf-deepcam_amd/csrc dwstamps):
cycles per phase (wait for the tile + barrier | LDS-DMA issue | BatchNorm transform | stencil + stores), summed per wave over its tiles.
    DEEPCAM_HIP_LIB=mlperf-deepcam_amd/libdeepcam_hip_dwstamps.so python scripts/dw_stamps.py"""
import ctypes as C, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = L.DC_BF16; lib = L.load(); st = L.stream_ptr(); P = L.dptr
lib.dc_debug_dwp_stamps.argtypes = [C.c_void_p]; lib.dc_debug_dwp_stamps.restype = C.c_int
Cc, H, W, N, dil = 728, 48, 72, 8, 1
ld = (Cc + 63) // 64 * 64
act = lambda: torch.randn(N, H, W, ld, device=dev).to(torch.bfloat16)
x, y, dy, dx, add = act(), act(), act(), act(), act()
wp = torch.randn(9 * Cc, device=dev) * 0.2
sc, sh, mean, invstd = [torch.rand(Cc, device=dev) + 0.5 for _ in range(4)]
rows = lib.dc_dwconv_dgrad_bnstats_rows(dt, Cc, 1, dil, N, H, W)
slab = torch.empty(2 * rows * Cc, device=dev); wslab = torch.empty(rows * 9 * Cc, device=dev)
fns = {"fwd": lambda: L.call("dc_dwconv_fwd", dt, Cc, 1, dil, N, H, W, P(x), ld, P(wp), P(y), ld, None, None, 0, st),
       "fwd xform": lambda: L.call("dc_dwconv_fwd", dt, Cc, 1, dil, N, H, W, P(x), ld, P(wp), P(y), ld, P(sc), P(sh), 1, st),
       "dgrad stats wgrad": lambda: L.call("dc_dwconv_dgrad_bnstats_wgrad", dt, Cc, 1, dil, N, H, W, P(dy), ld, P(wp), P(dx), ld, P(y), ld, P(mean),
                                           P(invstd), P(sc), P(sh), 1, P(slab), P(wslab), st)}
buf = np.zeros(256 * 8 * 4, dtype=np.uint64)
for name, fn in fns.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    assert lib.dc_debug_dwp_stamps(buf.ctypes.data) == 0
    a = buf.reshape(256, 8, 4).astype(np.float64)[:252]
    tot = a.sum(-1)
    print(f"{name:20s} cycles per wave (mean over 252 x 8 waves): wait+barrier {a[..., 0].mean():8.0f}  dma issue {a[..., 1].mean():8.0f}  "
          f"transform {a[..., 2].mean():8.0f}  stencil+stores {a[..., 3].mean():8.0f}  total {tot.mean():8.0f} (min {tot.min():.0f} max {tot.max():.0f})")
