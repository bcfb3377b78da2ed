"""The grouped 728 -> 728 weight-gradient launch (three layers, 189 workgroups of 123 stages: 23 launches and 4.8 ms of a local-batch-8 step)
under the probe builds of wgrad256.hip (-DDC_WG256_PROBE=mask: 1 no LDS-DMA, 2 no LDS fragment reads, 8 no slab stores): the kernel alone
(dc_conv_wgrad_partial, no fold), operands cycling over distinct buffers as in the step.   python scripts/wgrad_group_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import ctypes as C, os, sys, torch
sys.path.insert(0, %r)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
lib = L.load()
out = []
for (cin, cout, N, H, W, G) in [(728, 728, 8, 48, 72, 3), (728, 728, 8, 48, 72, 1)]:
    ld = (cin + 31) // 32 * 32
    desc = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    NL = 12
    xs = [torch.randn(N, H, W, ld, device=dev).to(dt) for _ in range(NL)]
    dys = [torch.randn(N, H, W, ld, device=dev).to(dt) for _ in range(NL)]
    splits, sb = C.c_int(), C.c_size_t()
    L.call("dc_conv_wgrad_plan", C.byref(desc), N, H, W, G, C.byref(splits), C.byref(sb))
    slabs = [torch.empty(sb.value // 4, device=dev) for _ in range(G)]
    pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    def once():
        for i in range(0, NL, G):
            L.call("dc_conv_wgrad_partial", C.byref(desc), N, H, W, G, pa(xs[i:i + G]), ld, pa(dys[i:i + G]), ld, pa(slabs), splits.value, L.stream_ptr())
    for _ in range(2): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): once()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 10 / (NL // G) * 1e3)
    out.append(splits.value)
print("US", *out)
''' % ROOT
names = {0: "everything", 8: "no slab stores", 1: "no LDS-DMA", 2: "no LDS reads", 3: "MFMA + epilogue only", 4: "operands from L2", 6: "from L2, no LDS reads"}
for m in (0, 4, 6, 8, 1, 2, 3):
    libp = os.path.join(ROOT, "mlperf-deepcam_amd", "libdeepcam_hip.so" if m == 0 else f"libdeepcam_hip_wgprobe{m}.so")
    p = subprocess.run([sys.executable, "-c", WORKER], env=dict(os.environ, DEEPCAM_HIP_LIB=libp), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("US")]
    if not line:
        print(f"probe {m}: FAILED {p.stderr[-300:]}"); continue
    v = line[0].split()[1:]
    print(f"probe {m:2d} {names[m]:22s} group of 3: {float(v[0]):7.1f} us per launch ({v[1]} splits)   single layer: {float(v[2]):7.1f} us ({v[3]} splits)", flush=True)
