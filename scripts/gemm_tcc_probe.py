"""Program for the TCC counter pass of VERDICT r02 item 2: a few forward launches of the implicit-GEMM kernels on three layer shapes
(the 728 -> 728 middle-flow pointwise conv on both its kernels, 1536 -> 2048, the decoder's 304 -> 256 3x3 at 192 x 288), nothing else.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d out -o run --output-format csv -- python3 scripts/gemm_tcc_probe.py
    python scripts/pmc_tcc.py out profiles/r03_pmc_tcc.json
"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import lib as L
dev = torch.device("cuda", 0); dt = torch.bfloat16
r32 = lambda c: (c + 31) // 32 * 32
lib = L.load()
#          tag                 pw384  k  pad cin   cout  N  H    W
shapes = [("pw728_256x384",    1,     1, 0,  728,  728,  8, 48,  72),
          ("pw728_256x256",    0,     1, 0,  728,  728,  8, 48,  72),
          ("pw1536_2048",      0,     1, 0,  1536, 2048, 8, 48,  72),
          ("dense3x3_304_256", 0,     3, 1,  304,  256,  8, 192, 288)]
for tag, pw, k, pad, cin, cout, N, H, W in shapes:
    L.call("dc_set_option", b"pw384", pw)
    desc = L.ConvDesc(L.DC_BF16, k, 1, pad, 1, 0, cin, cout)
    x = torch.randn(N, H, W, r32(cin), device=dev).to(dt)
    wf = (torch.randn(k * k * cout * ((cin + 63) // 64 * 64), device=dev) * 0.05).to(dt)
    rows = lib.dc_conv_stat_rows(C.byref(desc), N, H, W)
    y = torch.zeros(N, H, W, r32(cout), device=dev, dtype=dt); slab = torch.zeros(2 * rows * cout, device=dev)
    for _ in range(6):
        L.call("dc_conv_fwd", C.byref(desc), N, H, W, L.dptr(x), r32(cin), L.dptr(wf), None, L.dptr(y), r32(cout), L.dptr(slab), 0, L.stream_ptr())
    torch.cuda.synchronize()
    print(tag, "done", flush=True)
