"""dc_bn_finalize + dc_dwconv_fwd (two launches) against dc_dwconv_fwd_bnfin (the finalize inside the depthwise kernel), back to back on one
stream as in the forward pass (so the dispatch boundaries count).  Default: the middle-flow unit at local batch 8.
    python scripts/bnfin_bench.py [C] [H] [W] [N]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 728
H = int(sys.argv[2]) if len(sys.argv) > 2 else 48
W = int(sys.argv[3]) if len(sys.argv) > 3 else 72
N = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda", 0); dt = L.DC_BF16; lib = L.load(); st = L.stream_ptr()
ld = (Cc + 31) // 32 * 32; M = N * H * W; rows = (M + 127) // 128
NB = 4
xs = [torch.randn(N, H, W, ld, device=dev).to(torch.bfloat16) for _ in range(NB)]
ys = [torch.empty_like(x) for x in xs]
wp = torch.randn(9 * Cc, device=dev) * 0.2
slab = torch.rand(2, rows, Cc, device=dev) * 128 + 1.0; slab[1] += slab[0] ** 2 / 64
gam, bet, rm, rv = [torch.rand(Cc, device=dev) + 0.5 for _ in range(4)]
nbt = torch.zeros(1, dtype=torch.int64, device=dev)
scale, shift, mean, invstd = [torch.empty(Cc, device=dev) for _ in range(4)]
words = lib.dc_dwconv_fwd_bnfin_sync_words(dt, Cc, 1, 1)
sync = torch.zeros(words, dtype=torch.int32, device=dev)
bn = L.BnFin(M, slab.data_ptr(), rows, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5, scale.data_ptr(),
             shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), sync.data_ptr(), 0)
P = L.dptr
def two(i):
    L.call("dc_bn_finalize", Cc, M, P(slab), rows, P(gam), P(bet), P(rm), P(rv), P(nbt), 0.1, 1e-5, P(scale), P(shift), P(mean), P(invstd), st)
    L.call("dc_dwconv_fwd", dt, Cc, 1, 1, N, H, W, P(xs[i]), ld, P(wp), P(ys[i]), ld, P(scale), P(shift), 1, st)
def one(i):
    L.call("dc_dwconv_fwd_bnfin", dt, Cc, 1, 1, N, H, W, P(xs[i]), ld, P(wp), P(ys[i]), ld, C.byref(bn), 1, st)
    bn.epoch += 1
def plain(i):
    L.call("dc_dwconv_fwd", dt, Cc, 1, 1, N, H, W, P(xs[i]), ld, P(wp), P(ys[i]), ld, P(scale), P(shift), 1, st)
def bench(name, fn, reps=60):
    for i in range(4): fn(i % NB)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i % NB)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:52s} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us per layer", flush=True)
print(f"C={Cc} {H}x{W} N={N}: {rows} slab rows, {words - 1} channel block(s)")
bench("dc_dwconv_fwd alone (coefficients given)", plain)
bench("dc_bn_finalize + dc_dwconv_fwd", two)
bench("dc_dwconv_fwd_bnfin", one)
for v in (1,):
    L.call("dc_set_option", b"dw_fin_fallback", v)
    bench("dc_dwconv_fwd_bnfin, every wait 'run out' (fallback)", one, reps=8)
L.call("dc_set_option", b"dw_fin_fallback", 0)
print("waits that ran out:", int(sync[-1]))
