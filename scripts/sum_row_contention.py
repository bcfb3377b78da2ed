"""How fast do fp64 atomics onto few addresses go?  dc_bn_bwd_reduce (a slab row per block) against dc_bn_bwd_reduce_sum (every block adds its 2 x C sums
to ONE row) on tensors whose reduce pass has up to 2 048 blocks per channel block: the atomic form's extra time per launch = what the memory side
charges for ~rows same-address adds.   python scripts/sum_row_contention.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = L.DC_BF16; lib = L.load(); st = L.stream_ptr(); P = L.dptr
for (M, Cc) in [(8 * 384 * 576, 128), (8 * 192 * 288, 256), (8 * 96 * 144, 128), (8 * 96 * 144, 728), (8 * 48 * 72, 728), (8 * 48 * 72, 64)]:
    NB = 3
    g = [torch.randn(M, Cc, device=dev).to(torch.bfloat16) for _ in range(NB)]
    y = [torch.randn(M, Cc, device=dev).to(torch.bfloat16) for _ in range(NB)]
    mean, invstd, sc, sh = [torch.rand(Cc, device=dev) + 0.5 for _ in range(4)]
    rows = lib.dc_bn_stat_rows(M)
    slab = torch.empty(2 * rows * Cc, device=dev); srow = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
    res = {}
    for name, fn, sl in (("rows", "dc_bn_bwd_reduce", slab), ("sum row", "dc_bn_bwd_reduce_sum", srow)):
        call = lambda i: L.call(fn, dt, M, Cc, P(g[i]), Cc, P(y[i]), Cc, None, 0, 2, P(mean), P(invstd), P(sl), P(sc), P(sh), st)
        for i in range(NB): call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        REPS = 30
        e0.record()
        for r in range(REPS): call(r % NB)
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / REPS * 1e3
    print(f"M={M:8d} C={Cc:4d} rows={rows:5d}: row slab {res['rows']:7.1f} us   sum row {res['sum row']:7.1f} us   ({2 * M * Cc * 2 / res['rows'] / 1e6:.2f} -> {2 * M * Cc * 2 / res['sum row'] / 1e6:.2f} TB/s)")
