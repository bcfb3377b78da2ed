"""3 x 3 "same" convolutions: weight gradient on 256 x 384 tiles of the [tap][ci] axis (wgrad384.hip, TAPS) against the planner's previous
kernels (wgrad256.hip / wgrad.hip), dc_conv_wgrad_partial + dc_fold_slabs per layer.   python scripts/wgrad384_taps_bench.py [B]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CONFIGS = [("old kernels", {"wgrad384": 3}), ("384 taps", {}), ("384 taps slots 216", {"wgrad384_slots": 216}), ("384 taps slots 252", {"wgrad384_slots": 252}),
           ("384 taps slots 144", {"wgrad384_slots": 144})]
SHAPES = [(3, 1, 256, 256, 192, 288, 0), (3, 1, 304, 256, 192, 288, 0), (3, 6, 2048, 256, 48, 72, 0), (3, 18, 2048, 256, 48, 72, 0),
          (3, 1, 256, 256, 192, 288, 1), (3, 1, 256, 256, 96, 144, 1), (3, 1, 256, 256, 48, 72, 1)]
if len(sys.argv) > 2 and sys.argv[2] == "tconv": SHAPES = [s for s in SHAPES if s[6]]
for (k, dil, cin, cout, H, W, tr) in SHAPES:
    N = B
    pad = dil * (k - 1) // 2
    desc = L.ConvDesc(L.DC_BF16, k, 2 if tr else 1, 1 if tr else pad, dil, tr, cin, cout)
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    dy = torch.randn(N, (2 if tr else 1) * H, (2 if tr else 1) * W, r32(cout), device=dev).to(dt)
    gw = torch.zeros(cout * cin * k * k, device=dev)
    ref = None
    for label, opts in CONFIGS:
        L.call("dc_reset_options")
        for kk, v in opts.items(): L.call("dc_set_option", kk.encode(), v)
        splits, sbytes = C.c_int(), C.c_size_t()
        L.call("dc_conv_wgrad_plan", C.byref(desc), N, H, W, 1, C.byref(splits), C.byref(sbytes))
        slab = torch.empty(sbytes.value // 4, device=dev)
        pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        ent = (L.FoldEntry * 1)(L.FoldEntry(slab.data_ptr(), gw.data_ptr(), L.DC_FOLD_CONVT if tr else L.DC_FOLD_CONV, splits.value, k * k, cout, cin))
        def once():
            L.call("dc_conv_wgrad_partial", C.byref(desc), N, H, W, 1, pa([x]), r32(cin), pa([dy]), r32(cout), pa([slab]), splits.value, L.stream_ptr())
            L.call("dc_fold_slabs", ent, 1, L.stream_ptr())
        for _ in range(2): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): once()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        got = gw.clone()
        if ref is None: ref = got
        err = float((got - ref).abs().max() / ref.abs().max())
        print(f"B={B} k{k} d{dil}{'T' if tr else ''} {cin}->{cout} @{H}x{W} {label:20s} splits {splits.value:3d}: {us:8.1f} us  {2.0 * N * H * W * cin * cout * k * k / us / 1e6:7.1f} TF  rel.diff vs first {err:.1e}", flush=True)
L.call("dc_reset_options")
