"""Where a launch of the 224 x 384 pointwise kernel spends its time: per-workgroup stamps of a diagnostic build.
    make -C mlperf-deepcam_amd/csrc stamps224 && DEEPCAM_HIP_LIB=mlperf-deepcam_amd/libdeepcam_hip_stamps224.so python scripts/pw224_stamps.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
import numpy as np
dev = torch.device("cuda", 0); dt = torch.bfloat16
r64 = lambda c: (c + 63) // 64 * 64
lib = L.load()
assert hasattr(lib, "dc_debug_pw224_stamps"), "load the stamps224 build through DEEPCAM_HIP_LIB"
L.call("dc_set_option", b"pw224", 2)
for (cin, cout, N, H, W) in [(728, 728, 8, 48, 72), (728, 728, 8, 96, 144), (1536, 1536, 8, 24, 36)]:
    d = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    w = torch.randn(cout, cin, 1, 1, device=dev) * cin ** -0.5
    wf = torch.empty(nwf.value, dtype=dt, device=dev); wb = torch.empty(nwb.value, dtype=dt, device=dev)
    L.call("dc_conv_pack_weights", C.byref(d), L.dptr(w), L.dptr(wf), L.dptr(wb), L.stream_ptr())
    NSET = 6
    xs = [torch.randn(N, H, W, r64(cin), device=dev).to(dt) for _ in range(NSET)]
    ys = [torch.zeros(N, H, W, r64(cout), device=dev, dtype=dt) for _ in range(NSET)]
    rows = lib.dc_conv_stat_rows(C.byref(d), N, H, W)
    slab = torch.zeros(2 * rows * cout, device=dev)
    for i in range(12):
        L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, L.dptr(xs[i % NSET]), r64(cin), L.dptr(wf), L.dptr(wb), None, L.dptr(ys[i % NSET]), r64(cout), L.dptr(slab), 0, 0, L.stream_ptr())
    torch.cuda.synchronize()
    tiles = ((cout + 383) // 384) * ((N * H * W + 223) // 224)
    nb = min(tiles, 1024)
    buf = np.zeros((nb, 8), dtype=np.uint64)
    rc = lib.dc_debug_pw224_stamps(buf.ctypes.data_as(C.c_void_p), nb)
    assert rc == 0
    rt = buf[:, :4].astype(np.int64); ck = buf[:, 4:].astype(np.int64)
    t0 = rt[:, 0].min()
    us = (rt - t0) / 100.0
    d01, d12, d23 = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    clk = (ck[:, 2] - ck[:, 1]) / np.maximum(rt[:, 2] - rt[:, 1], 1) * 100.0     # MHz inside the loop
    nsteps = (cin + 31) // 32
    q = lambda a: f"{np.median(a):6.2f} (min {a.min():6.2f} max {a.max():6.2f})"
    print(f"{cin}->{cout} M={N*H*W} tiles {tiles} (first {nb}): entry spread {us[:, 0].max():5.2f} us | prologue {q(d01)} | K loop {q(d12)} = {np.median(ck[:, 2] - ck[:, 1]) / nsteps:6.0f} cycles/step "
          f"at {np.median(clk):5.0f} MHz | epilogue + drain {q(d23)} | last exit {us[:, 3].max():6.2f} us", flush=True)
