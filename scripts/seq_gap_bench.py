"""What a dependent launch boundary costs between the REAL kernels of a middle-flow unit (pointwise GEMM, BatchNorm finalize, depthwise),
un-profiled: event time per iteration of a sequence against the sum of its members timed alone (each alone-loop already contains one
boundary per kernel), plus the host time the enqueue loop took.    python scripts/seq_gap_bench.py"""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = L.DC_BF16; lib = L.load()
use_own = os.environ.get("SEQ_STREAM", "default") == "own"
own = torch.cuda.Stream() if use_own else None
if own is not None: torch.cuda.set_stream(own)
st = L.stream_ptr(); P = L.dptr
Cc, H, W, N = 728, 48, 72, 8
ld = (Cc + 31) // 32 * 32; M = N * H * W
NB = 3
xs = [torch.randn(N, H, W, ld, device=dev).to(torch.bfloat16) for _ in range(NB)]
ds = [torch.empty_like(x) for x in xs]
ys = [torch.empty_like(x) for x in xs]
desc = L.ConvDesc(dt, 1, 1, 0, 1, 0, Cc, Cc)
wf = (torch.randn(Cc * ((Cc + 63) // 64 * 64), device=dev) * 0.05).to(torch.bfloat16)
rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
slab = torch.zeros(2 * rows * Cc, device=dev)
wp = torch.randn(9 * Cc, device=dev) * 0.2
gam, bet, rm, rv = [torch.rand(Cc, device=dev) + 0.5 for _ in range(4)]
nbt = torch.zeros(1, dtype=torch.int64, device=dev)
scale, shift, mean, invstd = [torch.empty(Cc, device=dev) for _ in range(4)]
def gemm(i): L.call("dc_conv_fwd", C.byref(desc), N, H, W, P(ds[i]), ld, P(wf), None, P(ys[i]), ld, P(slab), 0, st)
def fin(i): L.call("dc_bn_finalize", Cc, M, P(slab), rows, P(gam), P(bet), P(rm), P(rv), P(nbt), 0.1, 1e-5, P(scale), P(shift), P(mean), P(invstd), st)
def dw(i): L.call("dc_dwconv_fwd", dt, Cc, 1, 1, N, H, W, P(ys[i]), ld, P(wp), P(ds[(i + 1) % NB]), ld, P(scale), P(shift), 1, st)
def app(i): L.call("dc_bn_apply", dt, M, Cc, P(ys[i]), ld, P(scale), P(shift), None, 0, 1, P(xs[i]), ld, st)
def bench(seq, reps=100):
    for i in range(6):
        for f in seq: f(i % NB)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9; host = 0
    for _ in range(3):
        e0.record(); t0 = time.perf_counter()
        for i in range(reps):
            for f in seq: f(i % NB)
        t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        if us < best: best, host = us, (t1 - t0) / reps * 1e6
    return best, host
K = {"gemm": gemm, "fin": fin, "dw": dw, "app": app}
alone = {}
print(f"stream: {'own' if use_own else 'default'}")
for k, f in K.items():
    alone[k] = bench([f]); print(f"{k:6s} alone {alone[k][0]:7.2f} us per call (host {alone[k][1]:5.1f} us per iteration)", flush=True)
for seq in (["gemm", "dw"], ["gemm", "fin"], ["fin", "dw"], ["gemm", "fin", "dw"], ["app", "dw"], ["gemm", "fin", "app", "dw"], ["gemm", "app"]):
    t, h = bench([K[k] for k in seq]); s = sum(alone[k][0] for k in seq)
    print(f"{'+'.join(seq):22s} {t:7.2f} us   sum of alone {s:7.2f}   extra {t - s:6.2f}   host {h:5.1f}", flush=True)
