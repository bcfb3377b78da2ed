"""The 224 x 384 pointwise kernel (csrc/igemm224.hip: weights from the [k][n] packing, three-deep rings) against the 256 x 384 kernel with
64-deep stages (igemm384.hip) and the planner's choice without either: time per call over ROTATING operand sets (so that neither pixels
nor outputs stay in the Infinity Cache between calls) and over one set, outputs bit-compared, BatchNorm slabs compared.
    python scripts/pw224_bench.py [small]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r64 = lambda c: (c + 63) // 64 * 64
shapes = [(728, 728, 8, 48, 72), (728, 728, 4, 48, 72), (728, 728, 2, 48, 72), (728, 1024, 8, 48, 72), (728, 1024, 4, 48, 72), (1024, 1536, 8, 24, 36), (1536, 1536, 8, 24, 36), (1536, 2048, 8, 24, 36),
          (256, 728, 8, 96, 144), (728, 728, 8, 96, 144), (728, 728, 3, 47, 71)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    shapes = shapes[:3]
lib = L.load()
NSET = 6
for (cin, cout, N, H, W) in shapes:
    d = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, cin, cout)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    w = (torch.randn(cout, cin, 1, 1, device=dev) * cin ** -0.5)
    wf = torch.empty(nwf.value, dtype=dt, device=dev); wb = torch.empty(nwb.value, dtype=dt, device=dev)
    L.call("dc_conv_pack_weights", C.byref(d), L.dptr(w), L.dptr(wf), L.dptr(wb), L.stream_ptr())
    xs = [torch.randn(N, H, W, r64(cin), device=dev).to(dt) for _ in range(NSET)]
    ys = [torch.zeros(N, H, W, r64(cout), device=dev, dtype=dt) for _ in range(NSET)]
    rows = lib.dc_conv_stat_rows(C.byref(d), N, H, W)
    slab = torch.zeros(2 * rows * cout, device=dev)
    outs, res = [], []
    # (label, options, entry point)
    for label, opts, kn in (("planner w/o 384/224", {"pw384": 0, "pw224": 0}, False), ("256x384 K64", {"pw384": 4, "pw224": 0}, False), ("224x384 kn", {"pw384": 1, "pw224": 2}, True), ("224x192 kn", {"pw384": 1, "pw224": 3}, True), ("planner", {"pw384": 1, "pw224": 1}, True)):
        for k, v in opts.items():
            L.call("dc_set_option", k.encode(), v)
        def once(i):
            x, y = xs[i % NSET], ys[i % NSET]
            if kn:
                L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, L.dptr(x), r64(cin), L.dptr(wf), L.dptr(wb), None, L.dptr(y), r64(cout), L.dptr(slab), 0, 0, L.stream_ptr())
            else:
                L.call("dc_conv_fwd", C.byref(d), N, H, W, L.dptr(x), r64(cin), L.dptr(wf), None, L.dptr(y), r64(cout), L.dptr(slab), 0, L.stream_ptr())
        t = []
        for rot in (True, False):
            for i in range(6): once(i if rot else 0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30): once(i if rot else 0)
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / 30 * 1e3)
        once(0); torch.cuda.synchronize()
        res.append(f"{label}: {t[0]:6.1f} us rotating ({2.0 * N * H * W * cin * cout / t[0] / 1e6:5.0f} TF) {t[1]:6.1f} us same set")
        outs.append((ys[0][..., :cout].clone(), slab.clone()))
    eq = [torch.equal(outs[0][0], o[0]) for o in outs[1:]]
    s0 = outs[0][1].view(2, rows, cout).sum(1)
    ds = [((s0 - o[1].view(2, rows, cout).sum(1)).abs().max() / (s0.abs().max() + 1e-30)).item() for o in outs[1:]]
    print(f"{cin:4d}->{cout:4d} M={N*H*W:7d}: " + " | ".join(res) + f" | y bit-equal {eq}, column sums rel diff {max(ds):.1e}", flush=True)
L.call("dc_set_option", b"pw384", 1); L.call("dc_set_option", b"pw224", 1)
