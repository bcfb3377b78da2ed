import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
shapes = [(3, 2, 1, 1, 1, 256, 128, 8, 192, 288), (3, 1, 1, 1, 0, 128, 128, 8, 384, 576), (1, 1, 0, 1, 0, 256, 128, 8, 192, 288), (3, 2, 1, 1, 1, 256, 256, 8, 96, 144),
          (3, 1, 1, 1, 0, 256, 128, 8, 192, 288), (1, 1, 0, 1, 0, 128, 48, 8, 192, 288)]
lib = L.load()
L.call("dc_set_option", b"igemm256p_min", 1)
for (k, s, p, d, tr, cin, cout, N, H, W) in shapes:
    desc = L.ConvDesc(L.DC_BF16, k, s, p, d, tr, cin, cout)
    kk = 9 if tr else k * k
    Ho, Wo = C.c_int(), C.c_int(); L.call("dc_conv_out_hw", C.byref(desc), H, W, C.byref(Ho), C.byref(Wo)); Ho, Wo = Ho.value, Wo.value
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    wf = (torch.randn(kk * cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    res = []
    for name, opts in (("auto", [(b"igemm256", 1), (b"igemm256p", 1)]), ("128-tile", [(b"igemm256", 0)]), ("256 1/wg", [(b"igemm256", 2), (b"igemm256p", 0)]), ("256 persistent", [(b"igemm256", 2), (b"igemm256p", 1)])):
        for o, v in opts: L.call("dc_set_option", o, v)
        y = torch.zeros(N, Ho, Wo, r32(cout), device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
        once = lambda: L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(wf), None, L.dptr(y), r32(cout), L.dptr(slab), 0, L.stream_ptr())
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): once()
        e1.record(); torch.cuda.synchronize()
        res.append(f"{name} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
    print(f"k{k}s{s}{'T' if tr else ' '} {cin:4d}->{cout:4d} @{H}x{W}: " + " | ".join(res), flush=True)
