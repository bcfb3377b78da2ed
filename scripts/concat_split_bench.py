"""Would splitting the decoder's 304-channel concat convolution (3x3, [256 ASPP | 48 low-level] -> 256, 192x288 at local batch 8) into a 256-channel
and a 48-channel part pay?  Times the layer's three passes as they run now against the two-part form on the same buffers (views of one
304-channel NHWC tensor).  python scripts/concat_split_bench.py [B]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0); dt = torch.bfloat16
N, H, W, CO = B, 192, 288, 256
st = L.stream_ptr()


def packed(cin):
    d = L.ConvDesc(L.DC_BF16, 3, 1, 1, 1, 0, cin, CO)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    w = torch.randn(CO, cin, 3, 3, device=dev) * 0.02
    wf, wb = torch.empty(nwf.value, dtype=dt, device=dev), torch.empty(nwb.value, dtype=dt, device=dev)
    L.call("dc_conv_pack_weights", C.byref(d), L.dptr(w), L.dptr(wf), L.dptr(wb), st)
    return d, wf, wb


x = torch.randn(N, H, W, 304, device=dev).to(dt)
dx = torch.empty_like(x)
y = torch.empty(N, H, W, CO, device=dev, dtype=dt)
dy = torch.randn(N, H, W, CO, device=dev).to(dt)
esz = 2


def view(t, off):
    return C.c_void_p(t.data_ptr() + off * esz)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


lib = L.load()
res = {}
for name, cin, off in (("304", 304, 0), ("256", 256, 0), ("48", 48, 256)):
    d, wf, wb = packed(cin)
    rows = lib.dc_conv_stat_rows(C.byref(d), N, H, W)
    slab = torch.empty(2 * rows * CO, device=dev)
    wsb = lib.dc_conv_wgrad_workspace(C.byref(d), N, H, W)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    gw = torch.empty(CO * cin * 9, device=dev)
    acc = 1 if name == "48" else 0
    res[name] = (
        timeit(lambda: L.call("dc_conv_fwd", C.byref(d), N, H, W, view(x, off), 304, L.dptr(wf), None, L.dptr(y), CO, None if acc else L.dptr(slab), acc, st)),
        timeit(lambda: L.call("dc_conv_dgrad", C.byref(d), N, H, W, L.dptr(dy), CO, L.dptr(wb), view(dx, off), 304, 0, st)),
        timeit(lambda: L.call("dc_conv_wgrad", C.byref(d), N, H, W, view(x, off), 304, L.dptr(dy), CO, L.dptr(ws), wsb, L.dptr(gw), st)),
    )
    print(f"cin {name:>3}: forward{' (accumulating)' if acc else ''} {res[name][0]:7.1f} us   data gradient {res[name][1]:7.1f} us   weight gradient {res[name][2]:7.1f} us", flush=True)
a, b, c = res["304"], res["256"], res["48"]
print(f"one layer: {sum(a):.0f} us;  two parts: {sum(b) + sum(c):.0f} us  (forward {a[0]:.0f} -> {b[0] + c[0]:.0f}, data gradient {a[1]:.0f} -> {b[1] + c[1]:.0f}, weight gradient {a[2]:.0f} -> {b[2] + c[2]:.0f})")
