"""What each part of the pointwise 384-tile kernel's K loop costs.  The diagnostic builds (make -C mlperf-deepcam_amd/csrc probes: one
library per compile-time mask) drop the LDS-DMA issues (1), the LDS fragment reads (2) and / or the epilogue stores (8).  Results are garbage by construction; only the times mean something.      python scripts/pw384_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import ctypes as C, os, sys, torch
sys.path.insert(0, %r)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
lib = L.load()
out = []
for (cin, cout, N, H, W, mode) in [(728, 728, 2, 48, 72, 2), (728, 728, 8, 48, 72, 2), (2912, 728, 8, 48, 72, 2), (728, 728, 4, 48, 72, 3)]:
    desc = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    wf = (torch.randn(cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    y = torch.zeros(N, H, W, r32(cout), device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
    L.call("dc_set_option", b"pw384", mode)
    once = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(wf), None, L.dptr(y), r32(cout), L.dptr(slab), 0, L.stream_ptr())
    for _ in range(3): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): once()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20 * 1e3)
print("US", *out)
''' % ROOT
names = {0: "everything", 8: "no stores", 1: "no LDS-DMA", 2: "no LDS reads", 3: "MFMA + epilogue only", 9: "no LDS-DMA, no stores",
         11: "MFMA only"}
print("columns: 728->728 M=6912 (54 tiles of 256x384, 23 K steps) | M=27648 (216 tiles) | 2912->728 M=27648 (216 tiles, 91 K steps) | "
      "728->728 M=13824 on 128x384 tiles (216 tiles)")
for m in (0, 8, 1, 9, 2, 3, 11):
    env = dict(os.environ, DEEPCAM_HIP_LIB=os.path.join(ROOT, "mlperf-deepcam_amd", f"libdeepcam_hip_probe{m}.so"))
    p = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("US")]
    if not line:
        print(f"probe {m}: FAILED {p.stderr[-300:]}")
        continue
    us = [float(v) for v in line[0].split()[1:]]
    print(f"probe {m:2d} {names[m]:26s} " + " | ".join(f"{v:7.1f} us" for v in us), flush=True)
