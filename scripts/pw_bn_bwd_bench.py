"""dc_pw_bn_bwd (BatchNorm backward apply + pointwise data gradient + weight gradient in one pass, csrc/pwbwd.hip) against the three passes it replaces,
at the entry flow's shapes.  python scripts/pw_bn_bwd_bench.py [N]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = L.load(); S = L.stream_ptr; P = lambda t: C.c_void_p(t.data_ptr())
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (cin, cout, H, W) in ((128, 128, 384, 576), (64, 128, 384, 576), (256, 256, 192, 288), (128, 256, 192, 288)):
    M = N * H * W
    y = torch.randn(M, cout, device=dev).to(dt); do = torch.randn(M, cout, device=dev).to(dt); x = torch.randn(M, cin, device=dev).to(dt)
    dy = torch.empty(M, cout, device=dev, dtype=dt); dx = torch.empty(M, cin, device=dev, dtype=dt); dx2 = torch.empty_like(dx)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
    d = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    nwf, nwb = C.c_size_t(), C.c_size_t(); L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf, wb = torch.zeros(nwf.value, dtype=dt, device=dev), torch.zeros(nwb.value, dtype=dt, device=dev)
    L.call("dc_conv_pack_weights", C.byref(d), P(w), P(wf), P(wb), S())
    gamma, mean, invstd = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
    sc, sh = gamma * invstd, torch.randn(cout, device=dev) * 0.3
    dg, db = torch.randn(cout, device=dev) * M * 0.01, torch.randn(cout, device=dev) * M * 0.01
    gw = torch.empty(cout, cin, 1, 1, device=dev); gw2 = torch.empty_like(gw)
    splits, sbytes = C.c_int(), C.c_size_t()
    L.call("dc_conv_wgrad_plan", C.byref(d), N, H, W, 1, C.byref(splits), C.byref(sbytes))
    slab = torch.empty(sbytes.value // 4, device=dev)
    pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    ent = (L.FoldEntry * 1)(L.FoldEntry(slab.data_ptr(), gw.data_ptr(), L.DC_FOLD_CONV, splits.value, 1, cout, cin))
    t_apply = timed(lambda: L.call("dc_bn_bwd_apply", L.DC_BF16, M, cout, M, P(do), cout, P(y), cout, None, 0, 2, P(gamma), P(mean), P(invstd), P(dg), P(db),
                                   P(dy), cout, None, 0, P(sc), P(sh), S()))
    t_dgrad = timed(lambda: L.call("dc_conv_dgrad", C.byref(d), N, H, W, P(dy), cout, P(wb), P(dx), cin, 0, S()))
    def wg():
        L.call("dc_conv_wgrad_partial", C.byref(d), N, H, W, 1, pa([x]), cin, pa([dy]), cout, pa([slab]), splits.value, S())
        L.call("dc_fold_slabs", ent, 1, S())
    t_wgrad = timed(wg)
    rows = lib.dc_pw_bn_bwd_rows(L.DC_BF16, cin, cout, M)
    slab2 = torch.empty(rows * cout * cin, device=dev)
    ent2 = (L.FoldEntry * 1)(L.FoldEntry(slab2.data_ptr(), gw2.data_ptr(), L.DC_FOLD_CONV, rows, 1, cout, cin))
    def fused():
        L.call("dc_pw_bn_bwd", L.DC_BF16, M, cin, cout, M, P(do), cout, P(y), cout, 2, P(gamma), P(mean), P(invstd), P(dg), P(db), P(sc), P(sh),
               P(x), cin, P(wb), P(dx2), cin, P(slab2), rows, S())
        L.call("dc_fold_slabs", ent2, 1, S())
    t_fused = timed(fused)
    t_k = timed(lambda: L.call("dc_pw_bn_bwd", L.DC_BF16, M, cin, cout, M, P(do), cout, P(y), cout, 2, P(gamma), P(mean), P(invstd), P(dg), P(db), P(sc), P(sh),
               P(x), cin, P(wb), P(dx2), cin, P(slab2), rows, S()))
    byts = M * (2 * cout + 2 * cin) * 2
    print(f"{cin:3d} -> {cout} at {N} x {H} x {W} (M = {M}): apply {t_apply:6.1f} + data gradient {t_dgrad:6.1f} + weight gradient (+ fold, {splits.value} splits) {t_wgrad:6.1f} = "
          f"{t_apply + t_dgrad + t_wgrad:6.1f} us | one pass {t_k:6.1f} (+ fold of {rows} rows: {t_fused:6.1f}) us = {byts / t_k / 1e6:5.2f} TB/s of (dout, y, x, dx) | "
          f"dx bit-equal {torch.equal(dx, dx2)}, dW rel diff {((gw - gw2).abs().max() / gw.abs().max()).item():.1e}", flush=True)
