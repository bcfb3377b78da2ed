"""256 x 384 (wgrad384.hip) vs 256 x 256 (wgrad256.hip) weight gradient of the pointwise layers, `NL` layers with distinct operand tensors (as
in the step), through dc_conv_wgrad_partial (slabs only) + dc_fold_slabs, per group size / split plan.
python scripts/wgrad384_bench.py [B]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NL = 12
CONFIGS = [  # label, options
    ("256x256 g3", {"wgrad384": 0}, 3),
    ("384 g3", {"wgrad384": 1}, 3),
    ("384 g4", {"wgrad384": 1}, 4),
    ("384 g3 slots216", {"wgrad384": 1, "wgrad384_slots": 216}, 3),
    ("384 g3 slots252", {"wgrad384": 1, "wgrad384_slots": 252}, 3),
    ("384 g6", {"wgrad384": 1}, 6),
    ("384 g12", {"wgrad384": 1}, 12),
    ("384 g12 nosplit", {"wgrad384": 1, "wgrad384_min_stages": 100000}, 12),
    ("384 g6 nosplit", {"wgrad384": 1, "wgrad384_min_stages": 100000}, 6),
    ("384 g12 slots144", {"wgrad384": 1, "wgrad384_slots": 144}, 12),
]
for (cin, cout, H, W) in [(728, 728, 48, 72), (1536, 1536, 48, 72)]:
    N = B
    desc = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    xs = [torch.randn(N, H, W, r32(cin), device=dev).to(dt) for _ in range(NL)]
    dys = [torch.randn(N, H, W, r32(cout), device=dev).to(dt) for _ in range(NL)]
    gws = [torch.zeros(cout * cin, device=dev) for _ in range(NL)]
    ref = None
    for label, opts, G in CONFIGS:
        L.call("dc_reset_options")
        for k, v in opts.items(): L.call("dc_set_option", k.encode(), v)
        splits, sbytes = C.c_int(), C.c_size_t()
        if lib.dc_conv_wgrad_plan(C.byref(desc), N, H, W, G, C.byref(splits), C.byref(sbytes)) != 0:
            print(f"{label}: not served"); continue
        slabs = [torch.empty(sbytes.value // 4, device=dev) for _ in range(NL)]
        pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        ents = [L.FoldEntry(slabs[l].data_ptr(), gws[l].data_ptr(), L.DC_FOLD_CONV, splits.value, 1, cout, cin) for l in range(NL)]
        def once(fold=True):
            for i in range(0, NL, G):
                L.call("dc_conv_wgrad_partial", C.byref(desc), N, H, W, G, pa(xs[i:i + G]), r32(cin), pa(dys[i:i + G]), r32(cout), pa(slabs[i:i + G]), splits.value, L.stream_ptr())
                if fold: L.call("dc_fold_slabs", (L.FoldEntry * G)(*ents[i:i + G]), G, L.stream_ptr())
        res = []
        for fold in (True, False):
            for _ in range(2): once(fold)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6): once(fold)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 6 / NL * 1e3)
        once(True); torch.cuda.synchronize()
        got = torch.stack([g.clone() for g in gws])
        if ref is None: ref = got
        err = float((got - ref).abs().max() / ref.abs().max())
        tiles = ((cin + 383) // 384 if opts.get("wgrad384", 1) else (cin + 255) // 256) * ((cout + 255) // 256)
        print(f"B={B} {cin}->{cout} {label:22s} splits {splits.value:2d} wgs {tiles * splits.value * G:4d}: {res[0]:7.1f} us/layer with fold, {res[1]:7.1f} without  "
              f"{2.0 * N * H * W * cin * cout / res[0] / 1e6:7.1f} TF  rel.diff vs first {err:.1e}", flush=True)
L.call("dc_reset_options")
