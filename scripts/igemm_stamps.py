"""Phase timeline of the implicit-GEMM kernel from in-kernel stamps (diagnostic build of the library, see DC_STAMPS in igemm.hip).

    make -C mlperf-deepcam_amd/csrc stamps        # -> libdeepcam_hip_stamps.so
    DEEPCAM_HIP_LIB=$PWD/mlperf-deepcam_amd/libdeepcam_hip_stamps.so python scripts/igemm_stamps.py [cin cout k H W N]
"""
import ctypes as C, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
a = [int(v) for v in sys.argv[1:]] + [None] * 6
cin, cout, k, H, W, N = a[0] or 728, a[1] or 728, a[2] or 1, a[3] or 48, a[4] or 72, a[5] or 8
dev = torch.device("cuda", 0); dt = torch.bfloat16
ld, ldo = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
desc = L.ConvDesc(L.DC_BF16, k, 1, k // 2, 1, 0, cin, cout)
x = torch.randn(N, H, W, ld, device=dev).to(dt); wf = (torch.randn(k * k * cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
y = torch.empty(N, H, W, ldo, device=dev, dtype=dt)
lib = L.load()
rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W); slab = torch.empty(2 * rows * cout, device=dev)
nblk = ((cout + 127) // 128) * ((N * H * W + 127) // 128)
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
run = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), ld, L.dptr(wf), None, L.dptr(y), ldo, L.dptr(slab), 0, L.stream_ptr())
for _ in range(20): run()
torch.cuda.synchronize()
assert lib.dc_debug_stamp_buf(C.c_void_p(buf.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(nblk, 8).astype(np.float64)
print(f"{cin}->{cout} k{k} @{H}x{W} N={N}: {nblk} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (stamped launch)")
t0 = s[:, 0].min()
start_us, end_us = (s[:, 0] - t0) / 100.0, (s[:, 7] - t0) / 100.0            # s_memrealtime: 100 MHz
life = end_us - start_us
clk = (s[:, 6] - s[:, 1]) / np.maximum(life, 1e-9) / 1e3                       # GHz seen by the workgroup
print(f"workgroup lifetime: mean {life.mean():.2f} us, p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f}; last end {end_us.max():.1f} us; clock ~{np.median(clk):.2f} GHz")
names = ["setup (tap table, row indices)", "first stage landed", "K loop", "acc -> LDS C tile", "stores + statistics"]
for i, nme in enumerate(names):
    d = (s[:, i + 2] - s[:, i + 1])
    print(f"  {nme:32s} mean {d.mean():9.0f} cyc  ({100 * d.mean() / (s[:, 6] - s[:, 1]).mean():5.1f} %)   p90 {np.percentile(d, 90):9.0f}")
hist, edges = np.histogram(start_us, bins=12)
print("start-time histogram (us):", " ".join(f"{edges[i]:.0f}:{hist[i]}" for i in range(len(hist))))

# ---- the 256 x 256 eight-wave kernel: per-segment cycle sums of the K loop (wave 0 of each group)
if hasattr(lib, "dc_debug_stamp_buf256"):
    L.call("dc_set_option", b"igemm256", 2)
    nblk2 = ((cout + 255) // 256) * ((N * H * W + 255) // 256)
    buf2 = torch.zeros(nblk2 * 16, dtype=torch.int64, device=dev)
    for _ in range(5): run()
    torch.cuda.synchronize()
    assert lib.dc_debug_stamp_buf256(C.c_void_p(buf2.data_ptr())) == 0
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    s2 = buf2.cpu().numpy().reshape(nblk2, 2, 8).astype(np.float64)
    print(f"256-tile kernel: {nblk2} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (stamped launch)")
    segn = ["M0: 4 ds_read + 2 LDS-DMA + 16 MFMA issue", "wait vmcnt(6) lgkmcnt(0)", "barrier", "M1: 8 ds_read + 2 LDS-DMA + 16 MFMA issue", "-", "-"]
    for gidx in (0, 1):
        st_ = s2[:, gidx, 7].mean()
        print(f"  group {gidx}: {st_:.0f} stages, loop {s2[:, gidx, 6].mean() / st_:.0f} cycles per stage")
        for k in range(4):
            print(f"      {segn[k]:52s} {s2[:, gidx, k].mean() / st_:7.0f} cyc")
        tot, pro, loop = s2[:, gidx, 5].mean(), s2[:, gidx, 4].mean(), s2[:, gidx, 6].mean()
        print(f"      whole workgroup {tot:.0f} cyc = prologue {pro:.0f} + K loop {loop:.0f} + epilogue {tot - pro - loop:.0f}")
    L.call("dc_set_option", b"igemm256", 1)
