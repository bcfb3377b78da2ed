"""pw224 with a row slab against pw224 adding to a sum row (124 same-address fp64 adds at the end of the launch), 728 -> 728 on 27 648 pixels and the
local-batch-4 / -2 shapes, rotating operand sets.   python scripts/sum_row_pw224.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); lib = L.load(); st = L.stream_ptr(); P = L.dptr
for (cin, cout, N, H, W) in [(728, 728, 8, 48, 72), (728, 728, 4, 48, 72), (728, 728, 2, 48, 72), (1536, 1536, 8, 48, 72)]:
    d = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    M = N * H * W; ldi = (cin + 63) // 64 * 64; ldo = (cout + 63) // 64 * 64
    NB = 4
    x = [torch.randn(M, ldi, device=dev).to(torch.bfloat16) for _ in range(NB)]
    y = [torch.empty(M, ldo, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    w = torch.randn(cout, cin, 1, 1, device=dev) * cin ** -0.5
    wf = torch.empty(nwf.value, dtype=torch.bfloat16, device=dev); wb = torch.empty(nwb.value, dtype=torch.bfloat16, device=dev)
    L.call("dc_conv_pack_weights", C.byref(d), P(w), P(wf), P(wb), st)
    rows = lib.dc_conv_stat_rows_kn(C.byref(d), N, H, W)
    assert lib.dc_conv_sum_row_kn(C.byref(d), N, H, W) == 1
    slab = torch.empty(2 * rows * cout, device=dev); srow = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    out = {}
    for name, sl, r in (("no sums", None, 0), ("row slab", slab, rows), ("sum row", srow, -1)):
        call = lambda i: L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, P(x[i]), ldi, P(wf), P(wb), None, P(y[i]), ldo, P(sl) if sl is not None else None, r, 0, st)
        for i in range(NB): call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        REPS = 40
        e0.record()
        for k in range(REPS): call(k % NB)
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / REPS * 1e3
    print(f"{cin}->{cout} M={M} ({rows} rows): " + "  ".join(f"{k} {v:6.1f} us" for k, v in out.items()))
