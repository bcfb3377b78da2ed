"""The classifier head's calls at the benchmark size (local batch 8: 256 channels at 384 x 576), alone on the GPU, under tile-planner switches.
    python scripts/head_bench.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = L.DC_BF16
N, Cin, H, W = int(os.environ.get("HB_N", "8")), 256, 384, 576
lib = L.load(); st = L.stream_ptr()
M = N * H * W
y = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
a = torch.empty_like(y); dx = torch.empty_like(y)
w = (torch.randn(Cin, 3, 3, 3, device=dev) * 0.05)
gw = torch.empty_like(w)
scale, shift, mean, invstd = [torch.rand(Cin, device=dev) + 0.5 for _ in range(4)]
logits = torch.empty(N, 3, 2 * H, 2 * W, device=dev); dl = torch.randn_like(logits) * 1e-3
labels = torch.randint(0, 3, (N, 2 * H, 2 * W), device=dev); cw = torch.tensor([0.9, 2.6, 1.7], device=dev)
ls = torch.zeros(1, dtype=torch.float64, device=dev); pred = torch.empty(N, 2 * H, 2 * W, dtype=torch.int64, device=dev); cnt = torch.zeros(9, dtype=torch.int64, device=dev)
ws = torch.empty(lib.dc_head_workspace(dt, N, Cin, H, W) + 256, dtype=torch.uint8, device=dev); wsp = C.c_void_p((ws.data_ptr() + 255) // 256 * 256)
rows = (M + 127) // 128; slab = torch.empty(2 * rows * Cin, device=dev)
P = L.dptr
def bench(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"  {name:58s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
for opts in ("", "head_dgrad_fused=0"):
    for o in filter(None, opts.split(",")):
        k, v = o.split("="); L.call("dc_set_option", k.encode(), int(v))
    print(f"[{opts or 'defaults'}]")
    bench("dc_bn_apply (the pass the head now does itself)", lambda: L.call("dc_bn_apply", dt, M, Cin, P(y), Cin, P(scale), P(shift), None, 0, 1, P(a), Cin, st))
    bench("dc_head_fwd_loss (stored input)", lambda: L.call("dc_head_fwd_loss", dt, N, Cin, H, W, P(a), Cin, P(w), None, wsp, P(labels), 8, P(cw), 1e-7, P(ls), P(dl), P(pred), P(cnt), st))
    bench("dc_head_fwd_loss_bnin", lambda: L.call("dc_head_fwd_loss_bnin", dt, N, Cin, H, W, P(y), Cin, P(scale), P(shift), 1, P(w), None, wsp, P(labels), 8, P(cw), 1e-7, P(ls), P(dl), P(pred), P(cnt), st))
    bench("dc_head_bwd_bnstats (stored input)", lambda: L.call("dc_head_bwd_bnstats", dt, N, Cin, H, W, P(a), Cin, P(dl), P(w), P(dx), Cin, P(gw), wsp, P(y), Cin, P(mean), P(invstd), P(scale), P(shift), 1, P(slab), st))
    bench("dc_head_bwd_bnin (with BatchNorm sums)", lambda: L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, P(y), Cin, P(scale), P(shift), 1, P(dl), P(w), P(dx), Cin, P(gw), wsp, P(mean), P(invstd), P(slab), 3, st))
    bench("dc_head_bwd_bnin, chain part only (parts = 1)", lambda: L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, P(y), Cin, P(scale), P(shift), 1, P(dl), P(w), P(dx), Cin, P(gw), wsp, P(mean), P(invstd), P(slab), 1, st))
    bench("dc_head_bwd_bnin (without)", lambda: L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, P(y), Cin, P(scale), P(shift), 1, P(dl), P(w), P(dx), Cin, P(gw), wsp, P(mean), P(invstd), None, 3, st))
L.call("dc_set_option", b"head_dgrad_fused", 1)
