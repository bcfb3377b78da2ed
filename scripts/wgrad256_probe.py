"""What each part of the 256 x 256 weight-gradient kernel's K loop costs.  Diagnostic builds of wgrad256.hip (-DDC_WG256_PROBE=mask, one
library per mask, built by hand as in the header of this file's history: hipcc -DDC_WG256_PROBE=m -c experiments/wgrad256_variants.hip (make experiments), linked with the other
objects into libdeepcam_hip_wgprobe<m>.so) drop the LDS-DMA issues (1), the transposing LDS fragment reads (2) and / or the slab stores (8).
Results are garbage by construction; only the times mean something.  Times include the slab reduction (dc_conv_wgrad = kernel + reduce).
    python scripts/wgrad256_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import ctypes as C, os, sys, torch
sys.path.insert(0, %r)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
lib = L.load()
L.call("dc_set_option", b"wgrad256", 2)
out = []
for (k, p, cin, cout, N, H, W) in [(1, 0, 728, 728, 8, 48, 72), (3, 1, 256, 256, 8, 192, 288), (1, 0, 1536, 2048, 8, 48, 72)]:
    desc = L.ConvDesc(L.DC_BF16, k, 1, p, 1, 0, cin, cout)
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    dy = torch.randn(N, H, W, r32(cout), device=dev).to(dt)
    wsb = lib.dc_conv_wgrad_workspace(C.byref(desc), N, H, W); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    gw = torch.zeros(cout * cin * k * k, device=dev)
    once = lambda: L.call("dc_conv_wgrad", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(dy), r32(cout), L.dptr(ws), wsb, L.dptr(gw), L.stream_ptr())
    for _ in range(3): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): once()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20 * 1e3)
print("US", *out)
''' % ROOT
names = {0: "everything", 8: "no slab stores", 1: "no LDS-DMA", 2: "no LDS reads", 3: "MFMA + epilogue only"}
print("columns: 728->728 1x1 M=27648 (9 tiles x 21 splits of 42 stages) | 256->256 3x3 M=442368 (9 x 21 of 658 stages) | 1536->2048 1x1 M=27648 (48 x 4 of 216 stages)")
for m in (0, 8, 1, 2, 3):
    libp = os.path.join(ROOT, "mlperf-deepcam_amd", "libdeepcam_hip.so" if m == 0 else f"libdeepcam_hip_wgprobe{m}.so")
    p = subprocess.run([sys.executable, "-c", WORKER], env=dict(os.environ, DEEPCAM_HIP_LIB=libp), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("US")]
    if not line:
        print(f"probe {m}: FAILED {p.stderr[-300:]}")
        continue
    us = [float(v) for v in line[0].split()[1:]]
    print(f"probe {m:2d} {names[m]:22s} " + " | ".join(f"{v:7.1f} us" for v in us), flush=True)
