"""Per-kernel parity: every C-ABI entry point of libdeepcam_hip.so against plain PyTorch fp32 ops on the CPU.

All calls go through the C ABI (ctypes, raw device pointers).  fp32 storage is compared tightly; bf16 storage is
compared against the same fp32 reference evaluated on bf16-rounded inputs, with a tolerance of a few bf16 ulps of the
output scale (the MFMA accumulates in fp32, only the stored result is rounded).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mlperf_deepcam_amd import lib as L  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]


def dev():
    return torch.device("cuda", 0)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """Round to the storage dtype and come back to fp32 (what the kernel actually sees)."""
    return t.to(dtype).float()


def to_nhwc(t_nchw, dtype, ld=None, off=0):
    """NCHW fp32 CPU -> device NHWC view of width C inside a [N,H,W,ld] buffer poisoned with NaN."""
    n, c, h, w = t_nchw.shape
    ld = ld or c
    buf = torch.full((n, h, w, ld), float("nan"), dtype=dtype, device=dev())
    buf[..., off:off + c] = t_nchw.permute(0, 2, 3, 1).to(dtype).to(dev())
    view = buf[..., off:off + c]
    return buf, view


def from_nhwc(view):
    return view.float().cpu().permute(0, 3, 1, 2).contiguous()


def empty_nhwc(n, h, w, c, dtype, ld=None, off=0, fill=float("nan")):
    ld = ld or c
    buf = torch.full((n, h, w, ld), fill, dtype=dtype, device=dev())
    return buf, buf[..., off:off + c]


def vptr(view):
    return C.c_void_p(view.data_ptr())


def tol(dtype, ref, f32=2e-4, bf16=2.5e-2):
    scale = float(ref.abs().max()) + 1e-12
    return (f32 if dtype == torch.float32 else bf16) * scale


def assert_close(got, ref, dtype, **kw):
    t = tol(dtype, ref, **kw)
    err = float((got - ref).abs().max())
    assert err <= t, f"max abs err {err:.3e} > tol {t:.3e} (scale {float(ref.abs().max()):.3e})"


def desc(dtype, k, stride, pad, dil, transposed, cin, cout):
    return L.ConvDesc(L.dtype_code(dtype), k, stride, pad, dil, transposed, cin, cout)


def S():
    return L.stream_ptr()


CONV_CASES = [
    # name, k, stride, pad, dil, transposed, cin, cout, N, H, W
    ("pw728", 1, 1, 0, 1, 0, 728, 728, 2, 12, 10),
    ("pw_s2", 1, 2, 0, 1, 0, 64, 128, 2, 16, 12),
    ("pw_small_n", 1, 1, 0, 1, 0, 128, 48, 1, 9, 7),
    ("dense3x3", 3, 1, 1, 1, 0, 304, 256, 2, 10, 12),
    ("atrous6", 3, 1, 6, 6, 0, 64, 256, 2, 16, 20),
    ("atrous18", 3, 1, 18, 18, 0, 32, 40, 1, 48, 72),
    ("convT", 3, 2, 1, 1, 1, 256, 256, 2, 8, 6),
    ("convT_small", 3, 2, 1, 1, 1, 24, 40, 1, 5, 7),
    ("gemm_rows2", 1, 1, 0, 1, 0, 2048, 256, 2, 1, 1),      # <= 16 pixels: the GEMV path of the image-pool branch
    ("gemm_rows16", 1, 1, 0, 1, 0, 264, 72, 16, 1, 1),
    ("gemm_rows17", 1, 1, 0, 1, 0, 264, 72, 17, 1, 1),      # one more pixel: back on the tiled kernel
    # the two thin stem convolutions at widths that are multiples of 32 output pixels: one-pass weight gradient (thinconv.hip)
    ("stem_thin", 3, 2, 1, 1, 0, 16, 32, 2, 10, 128),
    ("conv2_thin", 3, 1, 1, 1, 0, 32, 64, 2, 7, 96),
    ("stem_thin_19_rows", 3, 2, 1, 1, 0, 16, 32, 1, 38, 64),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_256_tile_kernel(case):
    """The eight-wave 256 x 256 tile kernel (forced on every eligible call) passes the same checks as the 128 x 128 one."""
    L.call("dc_set_option", b"igemm256", 2)
    try:
        test_conv_fwd_dgrad_wgrad(case, torch.bfloat16)
    finally:
        L.call("dc_set_option", b"igemm256", 1)


PW_CASES = [c for c in CONV_CASES if c[1] == 1 and c[2] == 1 and c[6] >= 64 and c[8] * c[9] * c[10] > 16]


@pytest.mark.parametrize("mode", [2, 3, 5], ids=["256x384", "128x384", "128x192"])
@pytest.mark.parametrize("case", PW_CASES, ids=[c[0] for c in PW_CASES])
def test_conv_pointwise_384_tile_kernel(case, mode):
    """The two-waves-per-SIMD 256 x 384 / 128 x 384 pointwise kernel (csrc/igemm384.hip), forced on every eligible forward and
    data-gradient call, passes the same checks against F.conv2d + autograd as the other tile shapes."""
    L.call("dc_set_option", b"pw384", mode)
    try:
        test_conv_fwd_dgrad_wgrad(case, torch.bfloat16)
    finally:
        L.call("dc_set_option", b"pw384", 1)


@pytest.mark.parametrize("shape", [(728, 728, 3, 19, 17), (728, 728, 2, 48, 72), (136, 392, 2, 21, 13), (1536, 776, 1, 24, 20)],
                         ids=["ragged969", "middle_flow_b2", "one_and_a_bit_tiles", "long_k_three_tiles"])
def test_conv_pointwise_384_tile_kernel_same_bits(shape):
    """Same MFMA instruction and K order as the 256 x 256 / 128 x 128 kernels: outputs bit-equal (forward with bias, accumulate mode,
    data gradient through padded views), BatchNorm slabs equal up to the order of the additions; pad channels untouched."""
    cin, cout, N, H, W = shape
    dtype = torch.bfloat16
    d = desc(dtype, 1, 1, 0, 1, 0, cin, cout)
    x = q(rnd(N, cin, H, W, seed=1), dtype)
    w = rnd(cout, cin, 1, 1, seed=2, scale=cin ** -0.5)
    bias = rnd(cout, seed=4).to(dev())
    gy = q(rnd(N, cout, H, W, seed=3), dtype)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.empty(nwf.value, dtype=dtype, device=dev())
    wb = torch.empty(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w.to(dev())), vptr(wf), vptr(wb), S())
    _, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
    _, gyv = to_nhwc(gy, dtype)
    rows = L.load().dc_conv_stat_rows(C.byref(d), N, H, W)
    got = []
    try:
        for mode in (0, 2, 3, 5):
            L.call("dc_set_option", b"pw384", mode)
            ybuf, yv = empty_nhwc(N, H, W, cout, dtype, ld=cout + 24, off=16)
            slab = torch.full((2, rows, cout), float("nan"), device=dev())
            L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), None, vptr(yv), cout + 24, vptr(slab), 0, S())
            y0 = from_nhwc(yv).clone()
            L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), vptr(bias), vptr(yv), cout + 24, None, 1, S())
            _, gxv = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8, off=0)
            L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(gxv), cin + 8, 0, S())
            torch.cuda.synchronize()
            assert torch.isnan(ybuf[..., :16].float()).all() and torch.isnan(ybuf[..., 16 + cout:].float()).all()
            got.append((y0, from_nhwc(yv), slab.cpu(), from_nhwc(gxv)))
    finally:
        L.call("dc_set_option", b"pw384", 1)
    yref = F.conv2d(x, q(w, dtype))
    assert_close(got[0][0], yref, dtype)
    for other in got[1:]:
        assert torch.equal(got[0][0], other[0]) and torch.equal(got[0][1], other[1]) and torch.equal(got[0][3], other[3])
        assert not torch.isnan(other[2]).any()
        np.testing.assert_allclose(got[0][2].numpy(), other[2].numpy(), rtol=2e-5, atol=1e-4)


@pytest.mark.parametrize("shape", [(728, 728, 3, 19, 17), (728, 728, 2, 48, 72), (136, 392, 2, 21, 13), (1536, 776, 1, 24, 20), (128, 384, 1, 16, 14),
                                   (728, 728, 8, 48, 72)],
                         ids=["ragged969", "middle_flow_b2", "one_and_a_bit_tiles", "long_k_three_tiles", "four_steps_one_tile", "middle_flow_b8"])
@pytest.mark.parametrize("tile", [2, 3], ids=["224x384", "224x192"])
def test_conv_pointwise_224_tile_kernel_same_bits(shape, tile):
    """dc_conv_fwd_kn / dc_conv_dgrad_kn on the 224 x 384 kernel (csrc/igemm224.hip: weight stages from the [k][n] packing through transposing
    LDS reads, three-deep rings), forced on every eligible call, against the plain entry points without it: same MFMA instruction and K order,
    so outputs are bit-equal (forward, forward with bias in accumulate mode, data gradient through padded views); the BatchNorm slab is the
    compact one of dc_conv_stat_rows_kn (one row per tile) or, with slab_rows = 0, the dc_conv_stat_rows layout with zero rows behind the tiles'
    own; its column sums agree up to the order of the additions; pad channels and the slab rows of other layers stay untouched.  Shapes: ragged pixel / channel / K tiles, Cin = 128 (the shortest K loop: four steps, the
    prologue's stages and the no-issue tail meet), three channel tiles, the middle flow at local batch 2 and 8."""
    cin, cout, N, H, W = shape
    dtype = torch.bfloat16
    d = desc(dtype, 1, 1, 0, 1, 0, cin, cout)
    x = q(rnd(N, cin, H, W, seed=1), dtype)
    w = rnd(cout, cin, 1, 1, seed=2, scale=cin ** -0.5)
    bias = rnd(cout, seed=4).to(dev())
    gy = q(rnd(N, cout, H, W, seed=3), dtype)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.empty(nwf.value, dtype=dtype, device=dev())
    wb = torch.empty(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w.to(dev())), vptr(wf), vptr(wb), S())
    _, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
    _, gyv = to_nhwc(gy, dtype)
    rows = L.load().dc_conv_stat_rows(C.byref(d), N, H, W)
    M = N * H * W
    ntm = (M + 223) // 224
    got = []
    try:
        for kn in (False, True):
            L.call("dc_set_option", b"pw384", 0 if not kn else 1)
            L.call("dc_set_option", b"pw224", tile if kn else 0)
            # the compact slab of the 224-pixel tiles: one row per tile (dc_conv_stat_rows_kn under the forced planner), none of them zeros
            krows = L.load().dc_conv_stat_rows_kn(C.byref(d), N, H, W) if kn else rows
            assert krows == (ntm if kn else rows)
            ybuf, yv = empty_nhwc(N, H, W, cout, dtype, ld=cout + 24, off=16)
            slab = torch.full((3 * krows, cout), float("nan"), device=dev())     # [2 * krows ..]: a neighbour's rows, must stay NaN
            _, gxv = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8, off=0)
            if kn:
                L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), vptr(wb), None, vptr(yv), cout + 24, vptr(slab), krows, 0, S())
                y0 = from_nhwc(yv).clone()
                L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), vptr(wb), vptr(bias), vptr(yv), cout + 24, None, 0, 1, S())
                L.call("dc_conv_dgrad_kn", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(wf), vptr(gxv), cin + 8, 0, S())
                # slab_rows = 0: the dc_conv_stat_rows layout, zero rows behind the tiles' own
                wide = torch.full((2, rows, cout), float("nan"), device=dev())
                _, y2 = empty_nhwc(N, H, W, cout, dtype)
                L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), vptr(wb), None, vptr(y2), cout, vptr(wide), 0, 0, S())
                # a slab shorter than the launch writes is refused
                with pytest.raises(L.DeepcamHipError):
                    L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), vptr(wb), None, vptr(y2), cout, vptr(wide), ntm - 1, 0, S()) if ntm > 1 else L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin + 16, None, vptr(wb), None, vptr(y2), cout, vptr(wide), 0, 0, S())
            else:
                L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), None, vptr(yv), cout + 24, vptr(slab), 0, S())
                y0 = from_nhwc(yv).clone()
                L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), vptr(bias), vptr(yv), cout + 24, None, 1, S())
                L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(gxv), cin + 8, 0, S())
            torch.cuda.synchronize()
            assert torch.isnan(ybuf[..., :16].float()).all() and torch.isnan(ybuf[..., 16 + cout:].float()).all()
            assert torch.isnan(slab[2 * krows:]).all() and not torch.isnan(slab[:2 * krows]).any()
            got.append((y0, from_nhwc(yv), slab[:2 * krows].view(2, krows, cout).cpu(), from_nhwc(gxv)))
            if kn:
                assert torch.equal(from_nhwc(y2), y0)
                w_ = wide.cpu()
                assert (w_[:, ntm:] == 0).all() and torch.equal(w_[:, :ntm], got[-1][2])
    finally:
        L.call("dc_set_option", b"pw384", 1)
        L.call("dc_set_option", b"pw224", 1)
    assert_close(got[0][0], F.conv2d(x, q(w, dtype)), dtype)
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1]) and torch.equal(got[0][3], got[1][3])
    slab = got[1][2]
    assert slab.shape[1] == ntm and (slab[1] > 0).all()                              # one row per tile
    np.testing.assert_allclose(got[0][2].double().sum(1).numpy(), slab.double().sum(1).numpy(), rtol=2e-5, atol=1e-3)


PERSIST_CASES = [
    # name, k, stride, pad, dil, transposed, cin, cout, N, H, W, workgroups
    ("pw_ragged", 1, 1, 0, 1, 0, 728, 728, 2, 40, 52, 8),          # ragged K (728 = 22.75 chunks), ragged channel and pixel tiles
    ("pw_two_rounds", 1, 1, 0, 1, 0, 256, 512, 2, 48, 64, 16),     # 48 tiles on 16 workgroups: three tiles each
    ("dense3x3", 3, 1, 1, 1, 0, 72, 264, 2, 32, 48, 8),
    ("atrous6", 3, 1, 6, 6, 0, 64, 256, 1, 48, 32, 8),
    ("convT_phases", 3, 2, 1, 1, 1, 256, 256, 2, 16, 32, 8),       # 1/2/2/4 taps per output parity class, tiles of different K
    ("stride2", 3, 2, 1, 1, 0, 128, 256, 2, 32, 48, 8),
    ("one_tile_each", 1, 1, 0, 1, 0, 128, 256, 1, 32, 64, 8),      # 8 tiles on 8 workgroups: no tile switch at all
    ("pw_stride2", 1, 2, 0, 1, 0, 128, 256, 2, 32, 48, 8),         # its data gradient has three sub-pixel phases without a tap (zeros)
]


@pytest.mark.parametrize("case", PERSIST_CASES, ids=[c[0] for c in PERSIST_CASES])
def test_conv_persistent_tiles_same_bits(case):
    """The persistent 256-tile kernel (igemm256p.hip: a workgroup walks several tiles, the operand ring never drains) against the
    one-tile-per-workgroup kernel: outputs and data gradients bit for bit, BatchNorm sums up to the order of the additions, nothing
    written outside the tensor's own channels."""
    name, k, stride, pad, dil, tr, cin, cout, N, H, W, wgs = case
    dtype = torch.bfloat16
    d = desc(dtype, k, stride, pad, dil, tr, cin, cout)
    kk = 3 if tr else k
    wshape = (cin, cout, kk, kk) if tr else (cout, cin, kk, kk)
    x = q(rnd(N, cin, H, W, seed=1), dtype)
    w = rnd(*wshape, seed=2, scale=(cin * kk * kk) ** -0.5)
    Ho, Wo = C.c_int(), C.c_int()
    L.call("dc_conv_out_hw", C.byref(d), H, W, C.byref(Ho), C.byref(Wo))
    Ho, Wo = Ho.value, Wo.value
    gy = q(rnd(N, cout, Ho, Wo, seed=3), dtype)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.empty(nwf.value, dtype=dtype, device=dev())
    wb = torch.empty(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w.to(dev())), vptr(wf), vptr(wb), S())
    _, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
    _, gyv = to_nhwc(gy, dtype)
    rows = L.load().dc_conv_stat_rows(C.byref(d), N, H, W)
    got = []
    try:
        L.call("dc_set_option", b"igemm256", 2)
        L.call("dc_set_option", b"igemm256p_min", 1)
        L.call("dc_set_option", b"igemm256p_wgs", wgs)
        for persistent in (0, 1):
            L.call("dc_set_option", b"igemm256p", persistent)
            ybuf, yv = empty_nhwc(N, Ho, Wo, cout, dtype, ld=cout + 24, off=16)
            slab = torch.full((2, rows, cout), float("nan"), device=dev())
            L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), None, vptr(yv), cout + 24, vptr(slab), 0, S())
            ybuf2, yv2 = empty_nhwc(N, Ho, Wo, cout, dtype, ld=cout + 24, off=16)          # without statistics: the other instantiation
            L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), None, vptr(yv2), cout + 24, None, 0, S())
            gbuf, gxv = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8, off=0)
            L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(gxv), cin + 8, 0, S())
            torch.cuda.synchronize()
            for b in (ybuf, ybuf2):
                assert torch.isnan(b[..., :16].float()).all() and torch.isnan(b[..., 16 + cout:].float()).all()
            assert torch.isnan(gbuf[..., cin:].float()).all()
            got.append((from_nhwc(yv), from_nhwc(yv2), from_nhwc(gxv), slab.double().sum(1).cpu()))
    finally:
        L.call("dc_set_option", b"igemm256", 1)
        L.call("dc_set_option", b"igemm256p", 1)
        L.call("dc_set_option", b"igemm256p_min", 257)          # the library's defaults
        L.call("dc_set_option", b"igemm256p_wgs", 0)
    assert_close(got[0][0], conv_ref(x, q(w, dtype), None, k, stride, pad, dil, tr), dtype)
    for v in (1,):
        for i in range(3):
            assert torch.equal(got[0][i], got[v][i]), f"variant {v}: output {i} differs"
        a, b = got[0][3], got[v][3]
        assert not torch.isnan(b).any()
        assert (a - b).abs().max().item() <= 2e-5 * (a.abs().max().item() + 1e-12) + 1e-6


THIN_TILE_CASES = [
    # cin, cout, N, H, W
    (32, 64, 2, 7, 96), (32, 64, 1, 21, 150), (32, 64, 3, 64, 192), (32, 64, 1, 1, 3),
]


@pytest.mark.parametrize("case", THIN_TILE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_thin_conv_lds_tiled_kernel_same_bits(case):
    """The LDS-tiled kernel of the stride-1 thin layers (thinconv.hip thin_tile_kernel: 32 -> 64 forward, 64 -> 32 data gradient; the halo
    of a 2 x 64 pixel tile staged once, the nine taps as address offsets) against the gather-form kernel it replaces: outputs and data
    gradients bit for bit (ragged tiles, one-row images, several tiles per workgroup), BatchNorm sums up to the order of the additions."""
    cin, cout, N, H, W = case
    dtype = torch.bfloat16
    d = desc(dtype, 3, 1, 1, 1, 0, cin, cout)
    x = q(rnd(N, cin, H, W, seed=1), dtype)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(cin * 9) ** -0.5)
    gy = q(rnd(N, cout, H, W, seed=3), dtype)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.empty(nwf.value, dtype=dtype, device=dev())
    wb = torch.empty(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w.to(dev())), vptr(wf), vptr(wb), S())
    _, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
    _, gyv = to_nhwc(gy, dtype)
    rows = L.load().dc_conv_stat_rows(C.byref(d), N, H, W)
    got = []
    try:
        for tiled in (0, 1):
            L.call("dc_set_option", b"thin_tile", tiled)
            ybuf, yv = empty_nhwc(N, H, W, cout, dtype, ld=cout + 24, off=16)
            slab = torch.full((2, rows, cout), float("nan"), device=dev())
            L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), None, vptr(yv), cout + 24, vptr(slab), 0, S())
            gbuf, gxv = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8, off=0)
            L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(gxv), cin + 8, 0, S())
            torch.cuda.synchronize()
            assert torch.isnan(ybuf[..., :16].float()).all() and torch.isnan(ybuf[..., 16 + cout:].float()).all()
            assert torch.isnan(gbuf[..., cin:].float()).all()
            got.append((from_nhwc(yv), from_nhwc(gxv), slab.double().sum(1).cpu()))
    finally:
        L.call("dc_set_option", b"thin_tile", 1)
    assert_close(got[0][0], conv_ref(x, q(w, dtype), None, 3, 1, 1, 1, 0), dtype)
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    a, b = got[0][2], got[1][2]
    assert not torch.isnan(b).any()
    assert (a - b).abs().max().item() <= 2e-5 * (a.abs().max().item() + 1e-12) + 1e-6


BNSTAT_CASES = [
    # name, k, stride, pad, dil, transposed, cin, cout, N, H, W
    ("dense3x3", 3, 1, 1, 1, 0, 256, 256, 2, 12, 20),
    ("convT", 3, 2, 1, 1, 1, 256, 256, 2, 9, 7),
    ("pw_bias_layer", 1, 1, 0, 1, 0, 256, 256, 2, 19, 13),
    ("head_like_k32", 1, 1, 0, 1, 0, 256, 32, 2, 24, 36),
    ("ragged_c", 3, 1, 1, 1, 0, 72, 40, 1, 11, 9),
]


@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("tile", [0, 2], ids=["tile128", "tile256"])
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", BNSTAT_CASES, ids=[c[0] for c in BNSTAT_CASES])
def test_conv_dgrad_with_fused_bn_backward_statistics(case, dtype, tile, relu):
    """dc_conv_dgrad_bnstats = dc_conv_dgrad followed by dc_bn_bwd_reduce (mask recomputed from y): same dx bits, the same
    per-channel sums (sum g, sum g*xhat) up to the order of the additions."""
    name, k, stride, pad, dil, tr, cin, cout, N, H, W = case
    if dtype == torch.float32 and tile == 2:
        pytest.skip("the 256-tile kernel is bf16 only")
    d = desc(dtype, k, stride, pad, dil, tr, cin, cout)
    kk = 3 if tr else k
    wshape = (cin, cout, kk, kk) if tr else (cout, cin, kk, kk)
    w = rnd(*wshape, seed=2, scale=(cout * kk * kk) ** -0.5)
    Ho, Wo = C.c_int(), C.c_int()
    L.call("dc_conv_out_hw", C.byref(d), H, W, C.byref(Ho), C.byref(Wo))
    Ho, Wo = Ho.value, Wo.value
    gy = q(rnd(N, cout, Ho, Wo, seed=3), dtype)
    ybn = q(rnd(N, cin, H, W, seed=5), dtype)                # the BatchNorm's INPUT, same shape as dx
    mean, invstd = rnd(cin, seed=6, scale=0.3).to(dev()), (rnd(cin, seed=7).abs() + 0.5).to(dev())
    mscale, mshift = rnd(cin, seed=8).to(dev()), rnd(cin, seed=9, scale=0.5).to(dev())
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.empty(nwf.value, dtype=dtype, device=dev())
    wb = torch.empty(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w.to(dev())), vptr(wf), vptr(wb), S())
    _, gyv = to_nhwc(gy, dtype)
    _, ybv = to_nhwc(ybn, dtype, ld=cin + 16, off=8)
    M = N * H * W
    try:
        L.call("dc_set_option", b"igemm256", tile)
        _, dx_ref = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8, off=0)
        L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(dx_ref), cin + 8, 0, S())
        rrows = L.load().dc_bn_stat_rows(M)
        rslab = torch.zeros(2, rrows, cin, device=dev())
        L.call("dc_bn_bwd_reduce", L.dtype_code(dtype), M, cin, vptr(dx_ref), cin + 8, vptr(ybv), cin + 16, None, 0, 2 if relu else 0,
               vptr(mean), vptr(invstd), vptr(rslab), vptr(mscale), vptr(mshift), S())
        rows = L.load().dc_conv_dgrad_bnstats_rows(C.byref(d), N, H, W)
        assert rows > 0
        slab = torch.full((2, rows, cin), float("nan"), device=dev())
        _, dx = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8, off=0)
        L.call("dc_conv_dgrad_bnstats", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(dx), cin + 8, vptr(ybv), cin + 16,
               vptr(mean), vptr(invstd), vptr(mscale), vptr(mshift), relu, vptr(slab), S())
        torch.cuda.synchronize()
    finally:
        L.call("dc_set_option", b"igemm256", 1)
    assert torch.equal(from_nhwc(dx), from_nhwc(dx_ref))
    assert not torch.isnan(slab).any()
    got, ref = slab.double().sum(1).cpu(), rslab.double().sum(1).cpu()
    scale = ref.abs().max().item() + 1e-12
    assert (got - ref).abs().max().item() <= 2e-5 * scale + 1e-6, ((got - ref).abs().max().item(), scale)
    with pytest.raises(L.DeepcamHipError):      # statistics and accumulate do not go together; neither does a missing slab
        L.call("dc_conv_dgrad_bnstats", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(dx), cin + 8, vptr(ybv), cin + 16,
               vptr(mean), vptr(invstd), vptr(mscale), vptr(mshift), relu, None, S())


def conv_ref(x, w, bias, k, stride, pad, dil, transposed):
    if transposed:
        return F.conv_transpose2d(x, w, bias, 2, 1, 1)
    return F.conv2d(x, w, bias, stride, pad, dil)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, dtype):
    name, k, stride, pad, dil, tr, cin, cout, N, H, W = case
    d = desc(dtype, k, stride, pad, dil, tr, cin, cout)
    kk = 3 if tr else k
    wshape = (cin, cout, kk, kk) if tr else (cout, cin, kk, kk)
    fan = cin * kk * kk
    x = q(rnd(N, cin, H, W, seed=1), dtype)
    w = rnd(*wshape, seed=2, scale=fan ** -0.5)
    wq = q(w, dtype)
    Ho, Wo = C.c_int(), C.c_int()
    L.call("dc_conv_out_hw", C.byref(d), H, W, C.byref(Ho), C.byref(Wo))
    Ho, Wo = Ho.value, Wo.value
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    yref = conv_ref(xr, wr, None, k, stride, pad, dil, tr)
    assert (yref.shape[2], yref.shape[3]) == (Ho, Wo)
    gy = q(rnd(*yref.shape, seed=3), dtype)
    gx_ref, gw_ref = torch.autograd.grad(yref, (xr, wr), gy)
    # also the weight gradient with unrounded master weights is the same function of (x, gy)

    wm = w.to(dev())
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.empty(nwf.value, dtype=dtype, device=dev())
    wb = torch.empty(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(wm), vptr(wf), vptr(wb), S())

    # forward into a channel slice of a wider buffer, with the BN partial statistics
    _, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
    ybuf, yv = empty_nhwc(N, Ho, Wo, cout, dtype, ld=cout + 24, off=16)
    rows = L.load().dc_conv_stat_rows(C.byref(d), N, H, W)
    slab = torch.full((2, rows, cout), float("nan"), device=dev())
    L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), None, vptr(yv), cout + 24, vptr(slab), 0, S())
    torch.cuda.synchronize()
    y = from_nhwc(yv)
    assert_close(y, yref.detach(), dtype)
    # untouched bytes outside the slice stay poisoned
    assert torch.isnan(ybuf[..., :16].float()).all() and torch.isnan(ybuf[..., 16 + cout:].float()).all()
    ssum = slab[0].sum(0).cpu()
    ssq = slab[1].sum(0).cpu()
    np.testing.assert_allclose(ssum.numpy(), y.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=2e-3 * float(y.abs().sum((0, 2, 3)).max()))
    np.testing.assert_allclose(ssq.numpy(), (y * y).sum((0, 2, 3)).numpy(), rtol=2e-3)

    # accumulate: y += conv(x)
    L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), None, vptr(yv), cout + 24, None, 1, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(yv), 2 * yref.detach(), dtype, bf16=4e-2)

    # bias
    bias = rnd(cout, seed=5).to(dev())
    L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(wf), vptr(bias), vptr(yv), cout + 24, None, 0, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(yv), yref.detach() + bias.cpu()[None, :, None, None], dtype)

    # data gradient
    _, gyv = to_nhwc(gy, dtype)
    _, gxv = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8, off=0)
    L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(gxv), cin + 8, 0, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(gxv), gx_ref, dtype)
    L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(gyv), cout, vptr(wb), vptr(gxv), cin + 8, 1, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(gxv), 2 * gx_ref, dtype, bf16=4e-2)

    # weight gradient (fp32, master layout)
    wsb = L.load().dc_conv_wgrad_workspace(C.byref(d), N, H, W)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev())
    gw = torch.full(wshape, float("nan"), device=dev())
    L.call("dc_conv_wgrad", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(gyv), cout, vptr(ws), wsb, vptr(gw), S())
    torch.cuda.synchronize()
    assert_close(gw.cpu(), gw_ref, dtype, f32=3e-4, bf16=1e-2)


# The shapes that carry the FLOPs of the benchmarked step, at their real extents: the cases above are tiny.
FULL_CASES = [
    # name, k, stride, pad, dil, transposed, cin, cout, N, H, W
    ("full_pw728_b8", 1, 1, 0, 1, 0, 728, 728, 8, 48, 72),            # M = 27 648: the middle-flow pointwise conv at local batch 8
    ("full_aspp_d18", 3, 1, 18, 18, 0, 2048, 256, 2, 48, 72),         # ASPP, dilation 18 (most taps of border pixels fall outside)
    ("full_aspp_d6", 3, 1, 6, 6, 0, 2048, 256, 1, 48, 72),
    ("full_dec304", 3, 1, 1, 1, 0, 304, 256, 1, 192, 288),            # decoder conv over the 256+48 concat
    ("full_convT256", 3, 2, 1, 1, 1, 256, 256, 1, 192, 288),          # the 192x288 -> 384x576 transposed conv
    ("full_pw1536_2048", 1, 1, 0, 1, 0, 1536, 2048, 2, 48, 72),
    ("full_pw128_thin", 1, 1, 0, 1, 0, 128, 128, 1, 384, 576),
]


@pytest.mark.parametrize("case", FULL_CASES, ids=[c[0] for c in FULL_CASES])
def test_conv_full_shapes(case):
    """Forward (+ BatchNorm partial sums, accumulate, bias), data gradient and weight gradient at the benchmark's real layer
    shapes against F.conv2d / autograd on the CPU, bf16 storage (the benchmarked dtype), default tile planner."""
    test_conv_fwd_dgrad_wgrad(case, torch.bfloat16)


def test_conv_full_shape_pw728_fp32():
    test_conv_fwd_dgrad_wgrad(("full_pw728_b2_f32", 1, 1, 0, 1, 0, 728, 728, 2, 48, 72), torch.float32)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 20, 28, 64, 40), (2, 48, 72, 2048, 256), (8, 48, 72, 2048, 256)], ids=["small", "aspp_full", "aspp_b8"])
def test_dilated_group_is_bit_identical_to_single_launches(shape, dtype):
    """dc_conv_fwd_dilated_group (the three atrous ASPP branches in one launch) against one dc_conv_fwd per branch: outputs and
    BatchNorm partial sums bit for bit (fp32 runs the per-branch fallback inside the entry point: same contract)."""
    N, H, W, cin, cout = shape
    if N == 8 and dtype != torch.bfloat16:
        pytest.skip("the local-batch-8 shape is there for the split-K form of more than one round (bf16 only)")
    dils = [6, 12, 18]
    x = q(rnd(N, cin, H, W, seed=1), dtype)
    _, xv = to_nhwc(x, dtype, ld=cin + 32, off=16)
    d0 = desc(dtype, 3, 1, 1, 1, 0, cin, cout)
    singles, wfs = [], []
    cat = torch.full((N, H, W, 3 * cout + 8), float("nan"), dtype=dtype, device=dev())          # members write channel slices of one buffer
    for b, dil in enumerate(dils):
        d = desc(dtype, 3, 1, dil, dil, 0, cin, cout)
        w = rnd(cout, cin, 3, 3, seed=10 + b, scale=(cin * 9) ** -0.5).to(dev())
        nwf, nwb = C.c_size_t(), C.c_size_t()
        L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
        wf = torch.empty(nwf.value, dtype=dtype, device=dev())
        L.call("dc_conv_pack_weights", C.byref(d), vptr(w), vptr(wf), None, S())
        rows = L.load().dc_conv_stat_rows(C.byref(d), N, H, W)
        ybuf, yv = empty_nhwc(N, H, W, cout, dtype, ld=3 * cout + 8, off=b * cout)
        slab = torch.full((2, rows, cout), float("nan"), device=dev())
        L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(xv), cin + 32, vptr(wf), None, vptr(yv), 3 * cout + 8, vptr(slab), 0, S())
        singles.append((yv.clone(), slab))
        wfs.append(wf)
        if b == 0 and N < 8:   # the first member against the CPU reference
            torch.cuda.synchronize()
            assert_close(from_nhwc(yv), F.conv2d(x, q(w.cpu(), dtype), None, 1, dil, dil), dtype)
    rows = singles[0][1].shape[1]
    gslabs = [torch.full((2, rows, cout), float("nan"), device=dev()) for _ in dils]
    ys = [cat[..., b * cout:(b + 1) * cout] for b in range(3)]
    L.call("dc_conv_fwd_dilated_group", C.byref(d0), N, H, W, 3, (C.c_int * 3)(*dils), vptr(xv), cin + 32,
           (C.c_void_p * 3)(*[t.data_ptr() for t in wfs]), (C.c_void_p * 3)(*[t.data_ptr() for t in ys]), 3 * cout + 8,
           (C.c_void_p * 3)(*[t.data_ptr() for t in gslabs]), S())
    torch.cuda.synchronize()
    for b in range(3):
        assert torch.equal(ys[b], singles[b][0]), f"member {b}: outputs differ"
        assert torch.equal(gslabs[b], singles[b][1]), f"member {b}: statistics differ"
    assert torch.isnan(cat[..., 3 * cout:].float()).all()
    # no statistics (eval mode): NULL slab array
    cat.fill_(float("nan"))
    L.call("dc_conv_fwd_dilated_group", C.byref(d0), N, H, W, 3, (C.c_int * 3)(*dils), vptr(xv), cin + 32,
           (C.c_void_p * 3)(*[t.data_ptr() for t in wfs]), (C.c_void_p * 3)(*[t.data_ptr() for t in ys]), 3 * cout + 8, None, S())
    torch.cuda.synchronize()
    assert all(torch.equal(ys[b], singles[b][0]) for b in range(3))
    with pytest.raises(L.DeepcamHipError):
        L.call("dc_conv_fwd_dilated_group", C.byref(d0), N, H, W, 5, (C.c_int * 5)(1, 2, 3, 4, 5), vptr(xv), cin + 32, None, None, cout, None, S())
    # split-K form (dc_conv_fwd_dilated_group_ws): offered where the launch is 81 tiles of 576 K steps; another order of the K sum, so the
    # outputs agree to a rounding of the stored value and the sums to fp32 accuracy
    wsb = L.load().dc_conv_fwd_dilated_group_workspace(C.byref(d0), N, H, W, 3, (C.c_int * 3)(*dils))
    if dtype != torch.bfloat16 or cin < 2048:
        assert wsb == 0
        return
    if N == 8:
        # 324 tiles: one and a quarter rounds of the chip.  Option igemm256_splitk = 2 cuts such a launch too, into the 2 - 4 splits that fill
        # the rounds best (3: 972 workgroups); the default leaves it whole
        assert wsb == 0
        L.call("dc_set_option", b"igemm256_splitk", 2)
        try:
            wsb = L.load().dc_conv_fwd_dilated_group_workspace(C.byref(d0), N, H, W, 3, (C.c_int * 3)(*dils))
            assert wsb == 3 * 324 * 256 * 256 * 4
            _split_k_checks(d0, N, H, W, cin, cout, dils, xv, wfs, ys, cat, singles, rows, wsb)
        finally:
            L.call("dc_set_option", b"igemm256_splitk", 1)
        return
    assert wsb == 3 * 81 * 256 * 256 * 4
    _split_k_checks(d0, N, H, W, cin, cout, dils, xv, wfs, ys, cat, singles, rows, wsb)


def _split_k_checks(d0, N, H, W, cin, cout, dils, xv, wfs, ys, cat, singles, rows, wsb):
    ws = torch.full((wsb // 4,), float("nan"), device=dev())
    for with_stats in (True, False):
        cat.fill_(float("nan"))
        gs2 = [torch.full((2, rows, cout), float("nan"), device=dev()) for _ in dils]
        L.call("dc_conv_fwd_dilated_group_ws", C.byref(d0), N, H, W, 3, (C.c_int * 3)(*dils), vptr(xv), cin + 32,
               (C.c_void_p * 3)(*[t.data_ptr() for t in wfs]), (C.c_void_p * 3)(*[t.data_ptr() for t in ys]), 3 * cout + 8,
               (C.c_void_p * 3)(*[t.data_ptr() for t in gs2]) if with_stats else None, vptr(ws), wsb, S())
        torch.cuda.synchronize()
        assert torch.isnan(cat[..., 3 * cout:].float()).all()
        for b in range(3):
            ref, got = singles[b][0].float(), ys[b].float()
            assert (got - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item(), f"member {b}"
            assert (got != ref).float().mean().item() < 0.2, f"member {b}: more than a rounding apart"
            if with_stats:
                assert torch.allclose(gs2[b], singles[b][1], rtol=2e-3, atol=2e-2), f"member {b}: statistics"
    # too small a workspace: the unsplit launch, bit for bit
    cat.fill_(float("nan"))
    L.call("dc_conv_fwd_dilated_group_ws", C.byref(d0), N, H, W, 3, (C.c_int * 3)(*dils), vptr(xv), cin + 32,
           (C.c_void_p * 3)(*[t.data_ptr() for t in wfs]), (C.c_void_p * 3)(*[t.data_ptr() for t in ys]), 3 * cout + 8, None, vptr(ws), wsb - 16, S())
    torch.cuda.synchronize()
    assert all(torch.equal(ys[b], singles[b][0]) for b in range(3))


GROUP_CASES = [
    # name, k, stride, pad, dil, transposed, cin, cout, N, H, W, layers
    ("pw728_x3", 1, 1, 0, 1, 0, 728, 728, 2, 24, 20, 3),       # the middle-flow Block: three pointwise convs, one launch
    ("pw728_x4", 1, 1, 0, 1, 0, 728, 728, 3, 17, 13, 4),       # ragged pixel count, the largest group
    ("dense3x3_x2", 3, 1, 1, 1, 0, 256, 256, 2, 10, 12, 2),
    ("thin_x2_falls_back", 1, 1, 0, 1, 0, 64, 48, 2, 9, 7, 2),  # not served by the 256-tile kernel: plain calls
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", GROUP_CASES, ids=[c[0] for c in GROUP_CASES])
def test_conv_wgrad_group(case, dtype):
    """dc_conv_wgrad_group == `layers` independent conv_backward_weight calls (fp32 falls back to the per-layer path)."""
    name, k, stride, pad, dil, tr, cin, cout, N, H, W, layers = case
    d = desc(dtype, k, stride, pad, dil, tr, cin, cout)
    wshape = (cout, cin, k, k)
    Ho, Wo = C.c_int(), C.c_int()
    L.call("dc_conv_out_hw", C.byref(d), H, W, C.byref(Ho), C.byref(Wo))
    Ho, Wo = Ho.value, Wo.value
    xs, gys, refs, keep = [], [], [], []
    for l in range(layers):
        x = q(rnd(N, cin, H, W, seed=10 + l), dtype)
        gy = q(rnd(N, cout, Ho, Wo, seed=20 + l), dtype)
        wr = torch.zeros(wshape, requires_grad=True)
        yref = F.conv2d(x, wr, None, stride, pad, dil)
        refs.append(torch.autograd.grad(yref, wr, gy)[0])
        xb, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
        gb, gyv = to_nhwc(gy, dtype)
        keep += [xb, gb]
        xs.append(xv)
        gys.append(gyv)
    lib = L.load()
    wsb = lib.dc_conv_wgrad_group_workspace(C.byref(d), N, H, W, layers)
    assert wsb >= lib.dc_conv_wgrad_workspace(C.byref(d), N, H, W)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev())
    gws = [torch.full(wshape, float("nan"), device=dev()) for _ in range(layers)]
    pa = lambda ts: (C.c_void_p * layers)(*[t.data_ptr() for t in ts])
    L.call("dc_conv_wgrad_group", C.byref(d), N, H, W, layers, pa(xs), cin + 16, pa(gys), cout, vptr(ws), wsb, pa(gws), S())
    torch.cuda.synchronize()
    for l in range(layers):
        assert_close(gws[l].cpu(), refs[l], dtype, f32=3e-4, bf16=1e-2)
    # deterministic: a second run gives the same bits
    gws2 = [torch.full(wshape, float("nan"), device=dev()) for _ in range(layers)]
    L.call("dc_conv_wgrad_group", C.byref(d), N, H, W, layers, pa(xs), cin + 16, pa(gys), cout, vptr(ws), wsb, pa(gws2), S())
    torch.cuda.synchronize()
    for l in range(layers):
        assert torch.equal(gws[l], gws2[l])
    with pytest.raises(L.DeepcamHipError):
        L.call("dc_conv_wgrad_group", C.byref(d), N, H, W, layers, pa(xs), cin + 16, pa(gys), cout, vptr(ws), 16, pa(gws), S())


WG384_CASES = [
    # name, k, dil, cin, cout, N, H, W, layers, option overrides
    ("pw728_x3_nosplit", 1, 1, 728, 728, 2, 24, 20, 3, {"wgrad384_min_stages": 4096}),     # one split per tile
    ("pw728_x4_splits", 1, 1, 728, 728, 3, 17, 13, 4, {"wgrad384_min_stages": 4, "wgrad384_slots": 256}),
    ("pw728_1024", 1, 1, 728, 1024, 2, 12, 10, 1, {}),
    ("pw1536_256", 1, 1, 1536, 256, 2, 9, 11, 2, {}),
    ("ragged_400_264", 1, 1, 400, 264, 2, 7, 9, 2, {"wgrad384_min_stages": 4}),            # partial channel tiles on both axes
    ("tiny_m", 1, 1, 384, 256, 1, 3, 5, 1, {}),                                             # fewer pixels than one stage
    ("x6_group", 1, 1, 728, 728, 2, 9, 10, 6, {}),                                          # the engine's group size
    ("3x3_256", 3, 1, 256, 256, 2, 11, 40, 1, {"wgrad384_min_stages": 4}),                  # 36 quads = 6 tiles of the [tap][ci] axis
    ("3x3_304_256", 3, 1, 304, 256, 2, 9, 35, 1, {"wgrad384_min_stages": 4}),               # 4.75 quads per tap: a partial quad, 7.5 tiles
    ("3x3_dil6_512_256", 3, 6, 512, 256, 2, 14, 33, 1, {"wgrad384_min_stages": 8}),         # atrous: taps at +-6, mostly halo at this size
    ("3x3_96_136", 3, 1, 96, 136, 1, 33, 32, 2, {"wgrad384_min_stages": 4}),                # rows exactly one stage long, grouped
]


@pytest.mark.parametrize("case", WG384_CASES, ids=[c[0] for c in WG384_CASES])
def test_conv_wgrad_384_tile_kernel(case):
    """The 256 x 384 weight-gradient kernel (wgrad384.hip: pointwise layers and stride-1 "same" 3 x 3 convolutions) forced on (option
    wgrad384 = 2) against autograd and against the other kernels (same products, other summation order over the pixel axis)."""
    name, k, dil, cin, cout, N, H, W, layers, opts = case
    dtype = torch.bfloat16
    pad = dil * (k - 1) // 2
    d = desc(dtype, k, 1, pad, dil, 0, cin, cout)
    xs, gys, refs, keep = [], [], [], []
    for l in range(layers):
        x = q(rnd(N, cin, H, W, seed=30 + l), dtype)
        gy = q(rnd(N, cout, H, W, seed=40 + l), dtype)
        wr = torch.zeros((cout, cin, k, k), requires_grad=True)
        refs.append(torch.autograd.grad(F.conv2d(x, wr, None, 1, pad, dil), wr, gy)[0])
        xb, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
        gb, gyv = to_nhwc(gy, dtype, ld=cout + 8)
        keep += [xb, gb]
        xs.append(xv)
        gys.append(gyv)
    lib = L.load()
    pa = lambda ts: (C.c_void_p * layers)(*[t.data_ptr() for t in ts])
    outs = {}
    for mode in (2, 0):
        L.call("dc_set_option", b"wgrad384", mode)
        for kk, v in opts.items():
            L.call("dc_set_option", kk.encode(), v)
        wsb = lib.dc_conv_wgrad_group_workspace(C.byref(d), N, H, W, layers)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev())
        gws = [torch.full((cout, cin, k, k), float("nan"), device=dev()) for _ in range(layers)]
        L.call("dc_conv_wgrad_group", C.byref(d), N, H, W, layers, pa(xs), cin + 16, pa(gys), cout + 8, vptr(ws), wsb, pa(gws), S())
        torch.cuda.synchronize()
        outs[mode] = gws
    for l in range(layers):
        assert_close(outs[2][l].cpu(), refs[l], dtype, f32=3e-4, bf16=1e-2)
        assert_close(outs[2][l].cpu(), outs[0][l].cpu(), dtype, f32=1e-5, bf16=1e-5)


TCONV384_CASES = [
    # name, cin, cout, N, H, W, option overrides
    ("256_256", 256, 256, 2, 9, 40, {"wgrad384_min_stages": 4}),
    ("ragged_264_136", 264, 136, 1, 7, 33, {"wgrad384_min_stages": 4}),          # partial tiles on both axes, odd row length
    ("one_row_stage", 256, 64, 2, 5, 32, {"wgrad384_min_stages": 4}),
]


@pytest.mark.parametrize("case", TCONV384_CASES, ids=[c[0] for c in TCONV384_CASES])
def test_conv_transpose_wgrad_384_tile_kernel(case):
    """ConvTranspose2d(k 3, stride 2, pad 1, output_padding 1) weight gradient on the 256 x 384 kernel (wgrad384.hip, MODE 2: x is the linear
    operand, dy is gathered at stride 2) against autograd and against the 128 x 128 kernel."""
    name, cin, cout, N, H, W, opts = case
    dtype = torch.bfloat16
    d = desc(dtype, 3, 2, 1, 1, 1, cin, cout)
    x = q(rnd(N, cin, H, W, seed=50), dtype)
    gy = q(rnd(N, cout, 2 * H, 2 * W, seed=51), dtype)
    wr = torch.zeros((cin, cout, 3, 3), requires_grad=True)
    ref = torch.autograd.grad(F.conv_transpose2d(x, wr, None, 2, 1, 1), wr, gy)[0]
    xb, xv = to_nhwc(x, dtype, ld=cin + 16, off=8)
    gb, gyv = to_nhwc(gy, dtype, ld=cout + 8)
    lib = L.load()
    outs = {}
    for mode in (2, 0):
        L.call("dc_set_option", b"wgrad384", mode)
        for kk, v in opts.items():
            L.call("dc_set_option", kk.encode(), v)
        wsb = lib.dc_conv_wgrad_workspace(C.byref(d), N, H, W)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev())
        gw = torch.full((cin, cout, 3, 3), float("nan"), device=dev())
        L.call("dc_conv_wgrad", C.byref(d), N, H, W, vptr(xv), cin + 16, vptr(gyv), cout + 8, vptr(ws), wsb, vptr(gw), S())
        torch.cuda.synchronize()
        outs[mode] = gw
    assert_close(outs[2].cpu(), ref, dtype, f32=3e-4, bf16=1e-2)
    assert_close(outs[2].cpu(), outs[0].cpu(), dtype, f32=1e-5, bf16=1e-5)


FOLD_CASES = [
    # name, k, stride, pad, dil, transposed, cin, cout, N, H, W, layers
    ("pw728_x3", 1, 1, 0, 1, 0, 728, 728, 2, 24, 20, 3),
    ("dense3x3", 3, 1, 1, 1, 0, 256, 256, 2, 10, 12, 1),
    ("tconv", 3, 2, 1, 1, 1, 64, 48, 2, 7, 9, 1),              # ConvTranspose2d: master layout [cin][cout][3][3]
    ("thin_128tile", 1, 1, 0, 1, 0, 64, 48, 2, 9, 7, 1),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", FOLD_CASES, ids=[c[0] for c in FOLD_CASES])
def test_conv_wgrad_partial_and_fold(case, dtype):
    """dc_conv_wgrad_partial + dc_fold_slabs (slabs of several layers and a depthwise layer's rows in ONE fold launch) give the bits of
    the reducing calls dc_conv_wgrad / dc_conv_wgrad_group / dc_dwconv_wgrad_reduce."""
    name, k, stride, pad, dil, tr, cin, cout, N, H, W, layers = case
    if dtype == torch.float32 and layers > 1:
        pytest.skip("the grouped launch is bf16 only")
    d = desc(dtype, k, stride, pad, dil, tr, cin, cout)
    wshape = (cin, cout, 3, 3) if tr else (cout, cin, k, k)
    Ho, Wo = C.c_int(), C.c_int()
    L.call("dc_conv_out_hw", C.byref(d), H, W, C.byref(Ho), C.byref(Wo))
    Ho, Wo = Ho.value, Wo.value
    xs, gys, keep = [], [], []
    for l in range(layers):
        xb, xv = to_nhwc(q(rnd(N, cin, H, W, seed=10 + l), dtype), dtype, ld=cin + 16, off=8)
        gb, gyv = to_nhwc(q(rnd(N, cout, Ho, Wo, seed=20 + l), dtype), dtype)
        keep += [xb, gb]
        xs.append(xv)
        gys.append(gyv)
    lib = L.load()
    pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    # reference: the reducing calls
    ref = [torch.full(wshape, float("nan"), device=dev()) for _ in range(layers)]
    wsb = lib.dc_conv_wgrad_group_workspace(C.byref(d), N, H, W, layers)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev())
    L.call("dc_conv_wgrad_group", C.byref(d), N, H, W, layers, pa(xs), cin + 16, pa(gys), cout, vptr(ws), wsb, pa(ref), S())
    # a depthwise layer's rows ride along in the same fold
    Cd, rows = 72, 37
    wslab = torch.randn(rows, 9, Cd, device=dev())
    dref = torch.full((Cd, 1, 3, 3), float("nan"), device=dev())
    L.call("dc_dwconv_wgrad_reduce", Cd, rows, vptr(wslab), vptr(dref), S())
    # slab-only form
    splits, sbytes = C.c_int(), C.c_size_t()
    L.call("dc_conv_wgrad_plan", C.byref(d), N, H, W, layers, C.byref(splits), C.byref(sbytes))
    assert sbytes.value == splits.value * wshape[0] * wshape[1] * wshape[2] * wshape[3] * 4
    slabs = [torch.full((sbytes.value // 4,), float("nan"), device=dev()) for _ in range(layers)]
    got = [torch.full(wshape, float("nan"), device=dev()) for _ in range(layers)]
    dgot = torch.full((Cd, 1, 3, 3), float("nan"), device=dev())
    L.call("dc_conv_wgrad_partial", C.byref(d), N, H, W, layers, pa(xs), cin + 16, pa(gys), cout, pa(slabs), splits.value, S())
    kind = L.DC_FOLD_CONVT if tr else L.DC_FOLD_CONV
    ents = [L.FoldEntry(wslab.data_ptr(), dgot.data_ptr(), L.DC_FOLD_DW, rows, 9, Cd, 1)]
    ents += [L.FoldEntry(slabs[l].data_ptr(), got[l].data_ptr(), kind, splits.value, wshape[2] * wshape[3], cout, cin) for l in range(layers)]
    L.call("dc_fold_slabs", (L.FoldEntry * len(ents))(*ents), len(ents), S())
    torch.cuda.synchronize()
    for l in range(layers):
        assert torch.equal(got[l], ref[l]), name
    assert torch.equal(dgot, dref)
    with pytest.raises(L.DeepcamHipError, match="split plan"):
        L.call("dc_conv_wgrad_partial", C.byref(d), N, H, W, layers, pa(xs), cin + 16, pa(gys), cout, pa(slabs), splits.value + 1, S())


def test_fold_more_entries_than_one_launch_holds():
    """dc_fold_slabs cuts long entry lists into launches of 24; 60 small layers of three kinds against a torch sum."""
    torch.manual_seed(5)
    ents, checks, keep = [], [], []
    for i in range(60):
        if i % 3 == 2:
            Cd, rows = 8 * (1 + i % 5), 3 + i
            slab = torch.randn(rows, 9, Cd, device=dev())
            out = torch.full((Cd, 9), float("nan"), device=dev())
            ents.append(L.FoldEntry(slab.data_ptr(), out.data_ptr(), L.DC_FOLD_DW, rows, 9, Cd, 1))
            checks.append((out, slab.double().sum(0).t().float(), 1e-6))
        else:
            co, ci, taps, splits = 8 + i, 4 * (1 + i % 7), (1, 9)[i % 2], 1 + i % 11
            slab = torch.randn(splits, taps, co, ci, device=dev())
            trn = i % 3 == 1
            out = torch.full((ci, co, taps) if trn else (co, ci, taps), float("nan"), device=dev())
            ents.append(L.FoldEntry(slab.data_ptr(), out.data_ptr(), L.DC_FOLD_CONVT if trn else L.DC_FOLD_CONV, splits, taps, co, ci))
            r = slab.double().sum(0)                          # [taps][co][ci]
            checks.append((out, (r.permute(2, 1, 0) if trn else r.permute(1, 2, 0)).float(), 1e-5))
        keep.append(slab)
    L.call("dc_fold_slabs", (L.FoldEntry * len(ents))(*ents), len(ents), S())
    torch.cuda.synchronize()
    for out, want, tol in checks:
        np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=tol, atol=tol)


def test_conv_rejects_bad_arguments():
    d = desc(torch.bfloat16, 5, 1, 0, 1, 0, 64, 64)
    x = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=dev())
    with pytest.raises(L.DeepcamHipError, match="kernel size"):
        L.call("dc_conv_fwd", C.byref(d), 1, 4, 4, vptr(x), 64, vptr(x), None, vptr(x), 64, None, 0, S())
    d = desc(torch.bfloat16, 1, 1, 0, 1, 0, 60, 64)   # 60 is not a multiple of 8
    with pytest.raises(L.DeepcamHipError, match="channel count"):
        L.call("dc_conv_fwd", C.byref(d), 1, 4, 4, vptr(x), 64, vptr(x), None, vptr(x), 64, None, 0, S())
    d = desc(torch.bfloat16, 1, 2, 0, 1, 0, 64, 64)
    with pytest.raises(L.DeepcamHipError, match="odd extent"):
        L.call("dc_conv_dgrad", C.byref(d), 1, 5, 5, vptr(x), 64, vptr(x), vptr(x), 64, 0, S())


PACK_CASES = [(1, 0, 728, 728), (3, 0, 304, 256), (3, 0, 16, 32), (3, 1, 256, 256), (1, 0, 2048, 256), (3, 0, 2048, 64), (1, 0, 48, 130),
              (1, 0, 128, 48), (1, 0, 1024, 1536), (1, 0, 72, 8)]   # the last three: 16-byte pointwise path with partial tiles


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_pack_all_matches_per_layer_pack(dtype):
    """dc_pack_all (one launch, LDS tile transposes) writes exactly what dc_conv_pack_weights writes, layer by layer,
    including layers for which only one of the two operands is wanted; the depthwise entry gives [9][C]."""
    table, keep, expect = [], [], []
    for i, (k, tr, cin, cout) in enumerate(PACK_CASES):
        d = L.ConvDesc(L.dtype_code(dtype), k, 2 if tr else 1, k // 2, 1, tr, cin, cout)
        wm = rnd(*((cin, cout, k, k) if tr else (cout, cin, k, k)), seed=20 + i).to(dev())
        nwf, nwb = C.c_size_t(), C.c_size_t()
        L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
        ref_f = torch.zeros(nwf.value, dtype=dtype, device=dev())
        ref_b = torch.zeros(nwb.value, dtype=dtype, device=dev())
        L.call("dc_conv_pack_weights", C.byref(d), vptr(wm), vptr(ref_f), vptr(ref_b), S())
        got_f, got_b = torch.zeros_like(ref_f), torch.zeros_like(ref_b)
        want_b = i % 3 != 2            # every third layer asks for the forward operand only
        table.append(L.PackEntry(wm.data_ptr(), got_f.data_ptr(), got_b.data_ptr() if want_b else None, cin, cout, k * k, 1 if tr else 0))
        keep += [wm, got_f, got_b]
        expect.append((ref_f, got_f, ref_b if want_b else None, got_b))
    Cd = 728
    wdm = rnd(Cd, 1, 3, 3, seed=40).to(dev())
    ref_d, got_d = torch.empty(9 * Cd, device=dev()), torch.zeros(9 * Cd, device=dev())
    L.call("dc_dwconv_pack_weights", Cd, vptr(wdm), vptr(ref_d), S())
    table.append(L.PackEntry(wdm.data_ptr(), got_d.data_ptr(), None, Cd, Cd, 9, 2))
    raw = bytes(bytearray(b"".join(bytes(e) for e in table)))
    tdev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev())
    L.call("dc_pack_all", L.dtype_code(dtype), vptr(tdev), len(table), S())
    torch.cuda.synchronize()
    for (ref_f, got_f, ref_b, got_b), case in zip(expect, PACK_CASES):
        k, tr, cin, cout = case
        ldf, ldb = got_f.numel() // (k * k * cout), got_b.numel() // (k * k * cin)      # the library's padded row strides (dc_conv_packed_elems)
        # compare the payload (the row padding is never read and may hold anything)
        f_ref, f_got = ref_f.view(k * k, cout, ldf)[:, :, :cin], got_f.view(k * k, cout, ldf)[:, :, :cin]
        assert torch.equal(f_ref, f_got), case
        if ref_b is not None:
            assert torch.equal(ref_b.view(k * k, cin, ldb)[:, :, :cout], got_b.view(k * k, cin, ldb)[:, :, :cout]), case
        else:
            assert not got_b.any()
    assert torch.equal(ref_d, got_d)


DW_CASES = [("s1", 728, 1, 1, 2, 12, 10), ("s2", 128, 2, 1, 2, 16, 12), ("d2", 1024, 1, 2, 1, 10, 14), ("odd", 64, 2, 1, 1, 9, 11),
            # stride-1 tiled path with thin layers (16 / 8 channel groups -> 16 / 32 pixel wide tiles), several tiles, ragged edges
            ("thin128", 128, 1, 1, 1, 17, 37), ("thin64", 64, 1, 1, 2, 9, 40), ("thin32d2", 32, 1, 2, 1, 11, 35),
            # stride-2 tiled path: several tiles, ragged edges, odd extents, all three channel-group widths
            ("s2_728", 728, 2, 1, 1, 22, 34), ("s2_256", 256, 2, 1, 2, 13, 41), ("s2_64", 64, 2, 1, 1, 18, 70),
            # more stride-2 tiles than data-gradient workgroups: every workgroup walks two tiles with its sums in registers
            ("s2_walk", 64, 2, 1, 3, 264, 420),
            # extents that are multiples of 8 with >= 128 channels: the persistent pipelined kernel (dwpipe.hip) in bf16
            ("pipe728", 728, 1, 1, 2, 16, 24), ("pipe1024d2", 1024, 1, 2, 1, 16, 8), ("pipe128", 128, 1, 1, 3, 24, 40)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", DW_CASES, ids=[c[0] for c in DW_CASES])
def test_depthwise(case, dtype):
    _, Cc, stride, dil, N, H, W = case
    dt = L.dtype_code(dtype)
    x = q(rnd(N, Cc, H, W, seed=1), dtype)
    w = rnd(Cc, 1, 3, 3, seed=2, scale=1 / 3)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yref = F.conv2d(F.pad(xr, (dil, dil, dil, dil)), wr, None, stride, 0, dil, groups=Cc)
    Ho, Wo = yref.shape[2:]
    gy = q(rnd(*yref.shape, seed=3), dtype)
    gx_ref, gw_ref = torch.autograd.grad(yref, (xr, wr), gy)
    wm = w.to(dev())
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())
    _, xv = to_nhwc(x, dtype, ld=Cc + 8)
    _, yv = empty_nhwc(N, Ho, Wo, Cc, dtype)
    L.call("dc_dwconv_fwd", dt, Cc, stride, dil, N, H, W, vptr(xv), Cc + 8, vptr(wd), vptr(yv), Cc, None, None, 0, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(yv), yref.detach(), dtype, bf16=1e-2)
    _, gyv = to_nhwc(gy, dtype)
    add = q(rnd(N, Cc, H, W, seed=4), dtype)
    _, addv = to_nhwc(add, dtype)
    _, gxv = empty_nhwc(N, H, W, Cc, dtype)
    L.call("dc_dwconv_dgrad", dt, Cc, stride, dil, N, H, W, vptr(gyv), Cc, vptr(wd), None, 0, vptr(gxv), Cc, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(gxv), gx_ref, dtype, bf16=1e-2)
    L.call("dc_dwconv_dgrad", dt, Cc, stride, dil, N, H, W, vptr(gyv), Cc, vptr(wd), vptr(addv), Cc, vptr(gxv), Cc, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(gxv), gx_ref + add, dtype, bf16=1e-2)
    wsb = L.load().dc_dwconv_wgrad_workspace(Cc, N, H, W, stride)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    gw = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
    L.call("dc_dwconv_wgrad", dt, Cc, stride, dil, N, H, W, vptr(xv), Cc + 8, vptr(gyv), Cc, vptr(ws), vptr(gw), None, None, 0, S())
    torch.cuda.synchronize()
    assert_close(gw.cpu(), gw_ref, dtype, f32=2e-4, bf16=2e-3)
    # data gradient (onto an addend) and weight gradient in ONE kernel (the tiled stride-1 kernels and the stride-2 data-gradient kernel)
    rows = L.load().dc_dwconv_dgrad_wgrad_rows(dt, Cc, stride, dil, N, H, W)
    assert rows > 0
    if rows > 0:
        _, gxv2 = empty_nhwc(N, H, W, Cc, dtype)
        wslab = torch.full((rows, 9, Cc), float("nan"), device=dev())
        gw2 = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
        L.call("dc_dwconv_dgrad_wgrad", dt, Cc, stride, dil, N, H, W, vptr(gyv), Cc, vptr(wd), vptr(addv), Cc, vptr(gxv2), Cc, vptr(xv), Cc + 8,
               None, None, 0, vptr(wslab), S())
        L.call("dc_dwconv_wgrad_reduce", Cc, rows, vptr(wslab), vptr(gw2), S())
        torch.cuda.synchronize()
        assert torch.equal(from_nhwc(gxv2), from_nhwc(gxv))
        assert_close(gw2.cpu(), gw_ref, dtype, f32=2e-4, bf16=2e-3)
        np.testing.assert_allclose(gw2.cpu().numpy(), gw.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(gw.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("case", DW_CASES, ids=lambda c: c[0])
def test_depthwise_dgrad_with_fused_weight_gradient(case, dtype, relu):
    """dc_dwconv_dgrad_bnstats_wgrad + dc_dwconv_wgrad_reduce = dc_dwconv_dgrad_bnstats followed by dc_dwconv_wgrad with the BatchNorm
    prologue: the same dx bits and BatchNorm sums, the same weight gradient up to the order of the additions (and both against autograd
    on the CPU through the materialised BatchNorm output)."""
    _, Cc, stride, dil, N, H, W = case
    dt = L.dtype_code(dtype)
    rows = L.load().dc_dwconv_dgrad_wgrad_rows(dt, Cc, stride, dil, N, H, W)
    assert rows > 0
    y = q(rnd(N, Cc, H, W, seed=1), dtype)                       # raw conv output = BatchNorm input
    scale, shift = rnd(Cc, seed=5).abs() + 0.5, rnd(Cc, seed=6, scale=0.3)
    mean, invstd = rnd(Cc, seed=7, scale=0.2), rnd(Cc, seed=8).abs() + 0.5
    w = rnd(Cc, 1, 3, 3, seed=2, scale=1 / 3)
    xhat = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        xhat = xhat.clamp_min(0)
    xhat = q(xhat, dtype)
    wr = w.clone().requires_grad_(True)
    yref = F.conv2d(F.pad(xhat, (dil, dil, dil, dil)), wr, None, stride, 0, dil, groups=Cc)
    gy = q(rnd(*yref.shape, seed=3), dtype)
    (gw_ref,) = torch.autograd.grad(yref, (wr,), gy)
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(w.to(dev())), vptr(wd), S())
    _, yv = to_nhwc(y, dtype, ld=Cc + 8)
    _, gyv = to_nhwc(gy, dtype)
    dmean, dinv, dsc, dsh = (t.to(dev()) for t in (mean, invstd, scale, shift))
    srows = L.load().dc_dwconv_dgrad_bnstats_rows(dt, Cc, stride, dil, N, H, W)
    # separate calls
    _, dx_ref = empty_nhwc(N, H, W, Cc, dtype)
    slab_ref = torch.full((2, srows, Cc), float("nan"), device=dev())
    L.call("dc_dwconv_dgrad_bnstats", dt, Cc, stride, dil, N, H, W, vptr(gyv), Cc, vptr(wd), vptr(dx_ref), Cc, vptr(yv), Cc + 8, vptr(dmean),
           vptr(dinv), vptr(dsc), vptr(dsh), relu, vptr(slab_ref), S())
    ws = torch.empty(L.load().dc_dwconv_wgrad_workspace(Cc, N, H, W, stride), dtype=torch.uint8, device=dev())
    gw_sep = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
    L.call("dc_dwconv_wgrad", dt, Cc, stride, dil, N, H, W, vptr(yv), Cc + 8, vptr(gyv), Cc, vptr(ws), vptr(gw_sep), vptr(dsc), vptr(dsh), relu, S())
    # one call
    _, dx = empty_nhwc(N, H, W, Cc, dtype)
    slab = torch.full((2, srows, Cc), float("nan"), device=dev())
    wslab = torch.full((rows, 9, Cc), float("nan"), device=dev())
    gw = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
    L.call("dc_dwconv_dgrad_bnstats_wgrad", dt, Cc, stride, dil, N, H, W, vptr(gyv), Cc, vptr(wd), vptr(dx), Cc, vptr(yv), Cc + 8, vptr(dmean),
           vptr(dinv), vptr(dsc), vptr(dsh), relu, vptr(slab), vptr(wslab), S())
    L.call("dc_dwconv_wgrad_reduce", Cc, rows, vptr(wslab), vptr(gw), S())
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(dx), from_nhwc(dx_ref)) and torch.equal(slab, slab_ref)
    assert not torch.isnan(wslab).any()
    assert_close(gw.cpu(), gw_ref, dtype, f32=2e-4, bf16=2e-3)
    np.testing.assert_allclose(gw.cpu().numpy(), gw_sep.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(gw_sep.abs().max()))


FIN_CASES = [("mid_b2", torch.bfloat16, 728, 1, 2, 48, 72, 54), ("exit_d2", torch.bfloat16, 1536, 2, 1, 20, 24, 7),
             ("ragged", torch.bfloat16, 256, 1, 3, 11, 13, 64), ("one_row", torch.bfloat16, 128, 1, 1, 8, 8, 1),
             ("f32", torch.float32, 64, 1, 2, 18, 14, 9)]


@pytest.mark.parametrize("case", FIN_CASES, ids=[c[0] for c in FIN_CASES])
def test_batchnorm_finalize_inside_its_consumer(case):
    """A short slab of BatchNorm partial sums (at most dc_bn_bwd_apply_fin_max_rows() rows) is summed by the kernel that consumes the
    coefficients, every workgroup for its own channels: dc_dwconv_fwd_fin against dc_bn_finalize + dc_dwconv_fwd, dc_bn_apply_fin against
    dc_bn_finalize + dc_bn_apply -- outputs, stored vectors and running statistics bit for bit (nn.BatchNorm2d in training mode inside a Block,
    deeplab_xception.py:83-119)."""
    _, dtype, Cc, dil, N, H, W, rows = case
    dt = L.dtype_code(dtype)
    lib = L.load()
    assert rows <= lib.dc_bn_bwd_apply_fin_max_rows() and lib.dc_dwconv_fwd_fin_ok(dt, Cc, 1, dil, N, H, W) == 1
    M = N * H * W
    y = q(rnd(N, Cc, H, W, seed=21) * 1.5 + 0.25, dtype)
    res = q(rnd(N, Cc, H, W, seed=22), dtype)
    ld = (Cc + 63) // 64 * 64
    _, yv = to_nhwc(y, dtype, ld=ld)
    _, rv_ = to_nhwc(res, dtype)
    # partial sums as a producer's epilogue leaves them: `rows` rows that add up to the tensor's column sums
    yf = from_nhwc(yv).permute(0, 2, 3, 1).reshape(M, Cc).float().to(dev())
    bounds = np.linspace(0, M, rows + 1).astype(int)
    slab = torch.stack([torch.stack([yf[a:b].sum(0) for a, b in zip(bounds[:-1], bounds[1:])]),
                        torch.stack([(yf[a:b] ** 2).sum(0) for a, b in zip(bounds[:-1], bounds[1:])])]).contiguous()
    gamma, beta = (torch.rand(Cc) + 0.5).to(dev()), rnd(Cc, seed=23, scale=0.3).to(dev())
    wm = rnd(Cc, 1, 3, 3, seed=24, scale=1 / 3).to(dev())
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())

    def state():
        st = {"rm": rnd(Cc, seed=25).to(dev()), "rv": (torch.rand(Cc, generator=torch.Generator().manual_seed(26)) + 0.5).to(dev()),
              "nbt": torch.full((1,), 5, dtype=torch.int64, device=dev())}
        for k in ("scale", "shift", "mean", "invstd"):
            st[k] = torch.full((Cc,), float("nan"), device=dev())
        return st

    def fin_args(st):
        return (vptr(slab), rows, vptr(gamma), vptr(beta), vptr(st["rm"]), vptr(st["rv"]), vptr(st["nbt"]), 0.1, 1e-5, vptr(st["scale"]),
                vptr(st["shift"]), vptr(st["mean"]), vptr(st["invstd"]))

    def same_state(a, b):
        for k in a:
            assert torch.equal(a[k], b[k]), k
        assert int(a["nbt"]) == 6

    for prelu in (1, 0):
        ref, got = state(), state()
        L.call("dc_bn_finalize", Cc, M, *fin_args(ref), S())
        _, d_ref = empty_nhwc(N, H, W, Cc, dtype)
        L.call("dc_dwconv_fwd", dt, Cc, 1, dil, N, H, W, vptr(yv), ld, vptr(wd), vptr(d_ref), Cc, vptr(ref["scale"]), vptr(ref["shift"]), prelu, S())
        db, d_got = empty_nhwc(N, H, W, Cc, dtype, ld=Cc + 16, off=8)
        L.call("dc_dwconv_fwd_fin", dt, Cc, 1, dil, N, H, W, vptr(yv), ld, vptr(wd), vptr(d_got), Cc + 16, prelu, M, *fin_args(got), S())
        torch.cuda.synchronize()
        assert torch.equal(d_got.float(), d_ref.float()) and torch.isnan(db[..., :8].float()).all() and torch.isnan(db[..., 8 + Cc:].float()).all()
        same_state(ref, got)
        for use_res in (True, False):
            got = state()
            _, o_ref = empty_nhwc(N, H, W, Cc, dtype)
            L.call("dc_bn_apply", dt, M, Cc, vptr(yv), ld, vptr(ref["scale"]), vptr(ref["shift"]), vptr(rv_) if use_res else None, Cc, prelu,
                   vptr(o_ref), Cc, S())
            ob, o_got = empty_nhwc(N, H, W, Cc, dtype, ld=Cc + 16, off=8)
            L.call("dc_bn_apply_fin", dt, M, Cc, M, vptr(yv), ld, *fin_args(got), vptr(rv_) if use_res else None, Cc, prelu, vptr(o_got),
                   Cc + 16, S())
            torch.cuda.synchronize()
            assert torch.equal(o_got.float(), o_ref.float()) and torch.isnan(ob[..., :8].float()).all() and torch.isnan(ob[..., 8 + Cc:].float()).all()
            same_state(ref, got)
    # with the row-block kernel switched off (option bn_apply_rows = 0) dc_bn_apply_fin is the finalize launch followed by the grid-stride apply
    # instead of an error: same outputs, same state
    L.call("dc_set_option", b"bn_apply_rows", 0)
    try:
        got = state()
        ob, o_fb = empty_nhwc(N, H, W, Cc, dtype, ld=Cc + 16, off=8)
        L.call("dc_bn_apply_fin", dt, M, Cc, M, vptr(yv), ld, *fin_args(got), None, Cc, prelu, vptr(o_fb), Cc + 16, S())
        torch.cuda.synchronize()
    finally:
        L.call("dc_set_option", b"bn_apply_rows", 1)
    assert torch.equal(o_fb.float(), o_got.float()) and torch.isnan(ob[..., :8].float()).all()
    same_state(ref, got)
    # against torch on the real statistics
    o = F.batch_norm(from_nhwc(yv).float().cpu(), None, None, gamma.cpu(), beta.cpu(), True, 0.1, 1e-5)
    assert_close(from_nhwc(o_got), o, dtype, bf16=2e-2)
    # a slab that is not 16-byte aligned is refused (the kernels take it with 16-byte loads)
    with pytest.raises(L.DeepcamHipError):
        L.call("dc_bn_apply_fin", dt, M, Cc, M, vptr(yv), ld, C.c_void_p(slab.data_ptr() + 4), rows, *fin_args(state())[2:], None, 0, 0, vptr(o_got),
               Cc + 16, S())
    # a slab longer than the kernels take is refused
    with pytest.raises(L.DeepcamHipError):
        L.call("dc_bn_apply_fin", dt, M, Cc, M, vptr(yv), ld, vptr(slab), 65, *fin_args(state())[2:], None, 0, 0, vptr(o_got), Cc + 16, S())


@pytest.mark.parametrize("shape", [(728, 728, 8, 48, 72, 1, 2), (728, 728, 4, 48, 72, 1, 2), (1536, 1536, 3, 20, 24, 2, 2), (256, 728, 2, 33, 21, 1, 2),
                                   (728, 728, 2, 48, 72, 1, 1)],
                         ids=["middle_flow_b8", "middle_flow_b4", "exit_flow_d2", "ragged", "middle_flow_b2_128x192_tiles"])
def test_batchnorm_sum_row(shape):
    """BatchNorm sums as ONE fp64 row (dc_conv_sum_row_kn): dc_conv_fwd_kn with slab_rows = -1 adds every tile's channel sums to a zeroed
    double[2][Cout] -- exactly the fp64 column sums of the row slab the same launch writes with slab_rows = dc_conv_stat_rows_kn, in whatever order
    the atomics arrive, and twice that after a second launch -- and dc_bn_finalize, dc_dwconv_fwd_fin and dc_bn_apply_fin take the row with
    rows = -1: coefficients, stored vectors, running statistics and outputs bit for bit those of the row-slab path
    (nn.BatchNorm2d in training mode behind SeparableConv2d_same.pointwise, deeplab_xception.py:62-66,104-119)."""
    cin, cout, N, H, W, dil, force = shape      # force = 2: the 224-pixel tiles wherever eligible; 1: the planner (here: igemm192.hip's 128 x 192 tiles)
    dtype = torch.bfloat16
    dt = L.dtype_code(dtype)
    lib = L.load()
    d = desc(dtype, 1, 1, 0, 1, 0, cin, cout)
    M = N * H * W
    x = q(rnd(N, cin, H, W, seed=31) + 0.3, dtype)
    w = rnd(cout, cin, 1, 1, seed=32, scale=cin ** -0.5)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.empty(nwf.value, dtype=dtype, device=dev())
    wb = torch.empty(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w.to(dev())), vptr(wf), vptr(wb), S())
    _, xv = to_nhwc(x, dtype)
    ld = (cout + 63) // 64 * 64
    L.call("dc_set_option", b"pw224", force)
    try:
        assert lib.dc_conv_sum_row_kn(C.byref(d), N, H, W) == 1
        rows = lib.dc_conv_stat_rows_kn(C.byref(d), N, H, W)
        assert rows == ((M + 223) // 224 if force == 2 else (M + 127) // 128)
        _, yv = empty_nhwc(N, H, W, cout, dtype, ld=ld)
        slab = torch.full((2, rows, cout), float("nan"), device=dev())
        L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin, vptr(wf), vptr(wb), None, vptr(yv), ld, vptr(slab), rows, 0, S())
        y_rows = yv.clone()
        srow = torch.zeros(2 * cout + 8, dtype=torch.float64, device=dev())
        srow[2 * cout:] = float("nan")                          # a neighbour's row: must stay as it is
        L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin, vptr(wf), vptr(wb), None, vptr(yv), ld, vptr(srow), -1, 0, S())
        torch.cuda.synchronize()
        assert torch.equal(yv, y_rows)
        want = slab.double().sum(1).reshape(-1)
        assert torch.equal(srow[:2 * cout], want) and torch.isnan(srow[2 * cout:]).all()
        twice = srow.clone()
        L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin, vptr(wf), vptr(wb), None, vptr(yv), ld, vptr(twice), -1, 0, S())
        torch.cuda.synchronize()
        assert torch.equal(twice[:2 * cout], 2 * want)
        # a launch the 224-pixel tiles do not serve refuses the sum row instead of writing rows into it
        L.call("dc_set_option", b"pw224", 0)
        L.call("dc_set_option", b"pw192", 0)
        assert lib.dc_conv_sum_row_kn(C.byref(d), N, H, W) == 0
        with pytest.raises(L.DeepcamHipError):
            L.call("dc_conv_fwd_kn", C.byref(d), N, H, W, vptr(xv), cin, vptr(wf), vptr(wb), None, vptr(yv), ld, vptr(srow), -1, 0, S())
    finally:
        L.call("dc_set_option", b"pw224", 1)
        L.call("dc_set_option", b"pw192", 1)
    gamma, beta = (torch.rand(cout) + 0.5).to(dev()), rnd(cout, seed=33, scale=0.3).to(dev())
    wm = rnd(cout, 1, 3, 3, seed=34, scale=1 / 3).to(dev())
    wd = torch.empty(9 * cout, device=dev())
    L.call("dc_dwconv_pack_weights", cout, vptr(wm), vptr(wd), S())
    res = q(rnd(N, cout, H, W, seed=35), dtype)
    _, rv_ = to_nhwc(res, dtype)

    def state():
        st = {"rm": rnd(cout, seed=36).to(dev()), "rv": (torch.rand(cout, generator=torch.Generator().manual_seed(37)) + 0.5).to(dev()),
              "nbt": torch.full((1,), 5, dtype=torch.int64, device=dev())}
        for k in ("scale", "shift", "mean", "invstd"):
            st[k] = torch.full((cout,), float("nan"), device=dev())
        return st

    def fin_args(st, sl, r):
        return (vptr(sl), r, vptr(gamma), vptr(beta), vptr(st["rm"]), vptr(st["rv"]), vptr(st["nbt"]), 0.1, 1e-5, vptr(st["scale"]),
                vptr(st["shift"]), vptr(st["mean"]), vptr(st["invstd"]))

    def same_state(a, b):
        for k in a:
            assert torch.equal(a[k], b[k]), k
        assert int(a["nbt"]) == 6

    ref, got = state(), state()
    L.call("dc_bn_finalize", cout, M, *fin_args(ref, slab, rows), S())
    L.call("dc_bn_finalize", cout, M, *fin_args(got, srow, -1), S())
    torch.cuda.synchronize()
    same_state(ref, got)
    assert lib.dc_dwconv_fwd_fin_ok(dt, cout, 1, dil, N, H, W) == 1
    for prelu in (1, 0):
        got = state()
        _, d_ref = empty_nhwc(N, H, W, cout, dtype)
        L.call("dc_dwconv_fwd", dt, cout, 1, dil, N, H, W, vptr(yv), ld, vptr(wd), vptr(d_ref), cout, vptr(ref["scale"]), vptr(ref["shift"]), prelu, S())
        db, d_got = empty_nhwc(N, H, W, cout, dtype, ld=cout + 16, off=8)
        L.call("dc_dwconv_fwd_fin", dt, cout, 1, dil, N, H, W, vptr(yv), ld, vptr(wd), vptr(d_got), cout + 16, prelu, M, *fin_args(got, srow, -1), S())
        torch.cuda.synchronize()
        assert torch.equal(d_got.float(), d_ref.float()) and torch.isnan(db[..., :8].float()).all() and torch.isnan(db[..., 8 + cout:].float()).all()
        same_state(ref, got)
        for use_res in (True, False):
            got = state()
            _, o_ref = empty_nhwc(N, H, W, cout, dtype)
            L.call("dc_bn_apply", dt, M, cout, vptr(yv), ld, vptr(ref["scale"]), vptr(ref["shift"]), vptr(rv_) if use_res else None, cout, prelu,
                   vptr(o_ref), cout, S())
            ob, o_got = empty_nhwc(N, H, W, cout, dtype, ld=cout + 16, off=8)
            L.call("dc_bn_apply_fin", dt, M, cout, M, vptr(yv), ld, *fin_args(got, srow, -1), vptr(rv_) if use_res else None, cout, prelu, vptr(o_got),
                   cout + 16, S())
            torch.cuda.synchronize()
            assert torch.equal(o_got.float(), o_ref.float()) and torch.isnan(ob[..., :8].float()).all() and torch.isnan(ob[..., 8 + cout:].float()).all()
            same_state(ref, got)
    # the statistics are the tensor's (against torch)
    o = F.batch_norm(from_nhwc(yv).float().cpu(), None, None, gamma.cpu(), beta.cpu(), True, 0.1, 1e-5)
    o = (o + res.float()).clamp_min(0) if False else o
    _, o_plain = empty_nhwc(N, H, W, cout, dtype)
    L.call("dc_bn_apply", dt, M, cout, vptr(yv), ld, vptr(ref["scale"]), vptr(ref["shift"]), None, 0, 0, vptr(o_plain), cout, S())
    assert_close(from_nhwc(o_plain), o, dtype, bf16=2e-2)


@pytest.mark.parametrize("tpb", [3, 50])
def test_depthwise_wgrad_several_tiles_per_workgroup(tpb):
    """The weight-gradient planner gives a workgroup several tiles only on large layers; force it on a small one."""
    L.call("dc_set_option", b"dw_wgrad_tpb", tpb)
    try:
        test_depthwise(("s1", 728, 1, 1, 2, 12, 10), torch.bfloat16)
        test_depthwise(("thin64", 64, 1, 1, 2, 9, 40), torch.float32)
    finally:
        L.call("dc_set_option", b"dw_wgrad_tpb", 0)


def test_depthwise_tiled_matches_register_window_path():
    """A/B switch: both stride-1 implementations give the same forward / data gradient bit for bit (same fp32 tap order)."""
    Cc, N, H, W = 728, 2, 20, 19
    x = q(rnd(N, Cc, H, W, seed=11), torch.bfloat16)
    wm = rnd(Cc, 1, 3, 3, seed=12, scale=1 / 3).to(dev())
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())
    _, xv = to_nhwc(x, torch.bfloat16)
    outs = []
    for mode in (1, 0):
        L.call("dc_set_option", b"dw_tile", mode)
        _, yv = empty_nhwc(N, H, W, Cc, torch.bfloat16)
        _, gv = empty_nhwc(N, H, W, Cc, torch.bfloat16)
        L.call("dc_dwconv_fwd", L.DC_BF16, Cc, 1, 1, N, H, W, vptr(xv), Cc, vptr(wd), vptr(yv), Cc, None, None, 0, S())
        L.call("dc_dwconv_dgrad", L.DC_BF16, Cc, 1, 1, N, H, W, vptr(xv), Cc, vptr(wd), None, 0, vptr(gv), Cc, S())
        torch.cuda.synchronize()
        outs.append((yv.clone(), gv.clone()))
    L.call("dc_set_option", b"dw_tile", 1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


PIPE_CASES = [("728", 728, 1, 2, 16, 24), ("728_b8", 728, 1, 8, 48, 72), ("1024d2", 1024, 2, 2, 16, 16), ("1536", 1536, 1, 1, 8, 8),
              ("128", 128, 1, 1, 40, 64), ("256_one_tile_each", 256, 1, 1, 8, 16), ("64_half_block", 64, 1, 2, 40, 48)]


@pytest.mark.parametrize("case", PIPE_CASES, ids=[c[0] for c in PIPE_CASES])
def test_depthwise_pipelined_kernel_matches_tiled(case):
    """dwpipe.hip (persistent, LDS-DMA ring, sums kept in registers across a workgroup's tiles) against dwtile.hip (option dw_pipe = 0) on
    every entry point it serves: outputs bit for bit (same taps, same order per pixel); BatchNorm sums and weight gradients after their
    finalize / fold to the rounding of a different summation order (one slab row per workgroup instead of one per tile)."""
    _, Cc, dil, N, H, W = case
    dt = L.DC_BF16
    lib = L.load()
    x = q(rnd(N, Cc, H, W, seed=1), torch.bfloat16)
    gy = q(rnd(N, Cc, H, W, seed=3), torch.bfloat16)
    add = q(rnd(N, Cc, H, W, seed=4), torch.bfloat16)
    wm = rnd(Cc, 1, 3, 3, seed=2, scale=1 / 3).to(dev())
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())
    mean, invstd = rnd(Cc, seed=8, scale=0.2).to(dev()), (torch.rand(Cc) + 0.5).to(dev())
    sc, sh = (torch.rand(Cc) + 0.5).to(dev()), rnd(Cc, seed=9, scale=0.3).to(dev())
    ld = (Cc + 31) // 32 * 32
    _, xv = to_nhwc(x, torch.bfloat16, ld=ld)
    _, gyv = to_nhwc(gy, torch.bfloat16, ld=ld)
    _, addv = to_nhwc(add, torch.bfloat16)
    res = []
    for mode in (2, 0):                                  # 2: the forward pass on the pipelined kernel too (default 1: data gradients only)
        L.call("dc_set_option", b"dw_pipe", mode)
        out = {}
        for prelu, tag in ((None, "fwd"), (1, "fwd_bn_relu"), (0, "fwd_bn")):
            _, yv = empty_nhwc(N, H, W, Cc, torch.bfloat16)
            L.call("dc_dwconv_fwd", dt, Cc, 1, dil, N, H, W, vptr(xv), ld, vptr(wd), vptr(yv), Cc, vptr(sc) if prelu is not None else None,
                   vptr(sh) if prelu is not None else None, prelu or 0, S())
            out[tag] = yv
        _, g1 = empty_nhwc(N, H, W, Cc, torch.bfloat16)
        L.call("dc_dwconv_dgrad", dt, Cc, 1, dil, N, H, W, vptr(gyv), ld, vptr(wd), None, 0, vptr(g1), Cc, S())
        _, g2 = empty_nhwc(N, H, W, Cc, torch.bfloat16)
        L.call("dc_dwconv_dgrad", dt, Cc, 1, dil, N, H, W, vptr(gyv), ld, vptr(wd), vptr(addv), Cc, vptr(g2), Cc, S())
        out["dgrad"], out["dgrad_add"] = g1, g2
        for relu in (1, 0):
            rows = lib.dc_dwconv_dgrad_bnstats_rows(dt, Cc, 1, dil, N, H, W)
            wrows = lib.dc_dwconv_dgrad_wgrad_rows(dt, Cc, 1, dil, N, H, W)
            assert rows > 0 and wrows > 0
            _, g3 = empty_nhwc(N, H, W, Cc, torch.bfloat16)
            slab = torch.full((2, rows, Cc), float("nan"), device=dev())
            L.call("dc_dwconv_dgrad_bnstats", dt, Cc, 1, dil, N, H, W, vptr(gyv), ld, vptr(wd), vptr(g3), Cc, vptr(xv), ld, vptr(mean), vptr(invstd),
                   vptr(sc), vptr(sh), relu, vptr(slab), S())
            dgam, dbet = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
            L.call("dc_bn_bwd_finalize", Cc, vptr(slab), rows, vptr(dgam), vptr(dbet), S())
            _, g4 = empty_nhwc(N, H, W, Cc, torch.bfloat16)
            slab2 = torch.full((2, rows, Cc), float("nan"), device=dev())
            wslab = torch.full((wrows, 9, Cc), float("nan"), device=dev())
            gw = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
            L.call("dc_dwconv_dgrad_bnstats_wgrad", dt, Cc, 1, dil, N, H, W, vptr(gyv), ld, vptr(wd), vptr(g4), Cc, vptr(xv), ld, vptr(mean),
                   vptr(invstd), vptr(sc), vptr(sh), relu, vptr(slab2), vptr(wslab), S())
            L.call("dc_dwconv_wgrad_reduce", Cc, wrows, vptr(wslab), vptr(gw), S())
            torch.cuda.synchronize()
            assert torch.equal(slab, slab2) and torch.equal(g3, g4) and torch.equal(g3, g1)
            out[f"stats{relu}"] = (dgam, dbet, gw)
            # the same with an addend (this layer is the LAST of several writers of the gradient): dx = the accumulated gradient, the sums are
            # those of it (against dc_bn_bwd_reduce over that dx), the weight gradient is unchanged
            _, g6 = empty_nhwc(N, H, W, Cc, torch.bfloat16)
            slab3 = torch.full((2, rows, Cc), float("nan"), device=dev())
            wslab3 = torch.full((wrows, 9, Cc), float("nan"), device=dev())
            gw3 = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
            L.call("dc_dwconv_dgrad_bnstats_wgrad_add", dt, Cc, 1, dil, N, H, W, vptr(gyv), ld, vptr(wd), vptr(addv), Cc, vptr(g6), Cc, vptr(xv), ld,
                   vptr(mean), vptr(invstd), vptr(sc), vptr(sh), relu, vptr(slab3), vptr(wslab3), S())
            L.call("dc_dwconv_wgrad_reduce", Cc, wrows, vptr(wslab3), vptr(gw3), S())
            M_ = N * H * W
            rrows = lib.dc_bn_stat_rows(M_)
            rslab = torch.full((2, rrows, Cc), float("nan"), device=dev())
            L.call("dc_bn_bwd_reduce", dt, M_, Cc, vptr(g2), Cc, vptr(xv), ld, None, 0, 2 if relu else 0, vptr(mean), vptr(invstd), vptr(rslab), vptr(sc),
                   vptr(sh), S())
            torch.cuda.synchronize()
            assert torch.equal(g6, g2)
            want, got = rslab.double().sum(1), slab3.double().sum(1)
            assert torch.allclose(got, want, rtol=1e-4, atol=1e-4 * float(want.abs().max())), (relu, (got - want).abs().max().item())
            np.testing.assert_allclose(gw3.cpu().numpy(), gw.cpu().numpy(), rtol=3e-5, atol=3e-5 * float(gw.abs().max()))
        # stored (or lazily affine) layer input, addend, no statistics
        for aff in (0, 1):
            _, g5 = empty_nhwc(N, H, W, Cc, torch.bfloat16)
            wslab = torch.full((wrows, 9, Cc), float("nan"), device=dev())
            gw = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
            L.call("dc_dwconv_dgrad_wgrad", dt, Cc, 1, dil, N, H, W, vptr(gyv), ld, vptr(wd), vptr(addv), Cc, vptr(g5), Cc, vptr(xv), ld,
                   vptr(sc) if aff else None, vptr(sh) if aff else None, aff, vptr(wslab), S())
            L.call("dc_dwconv_wgrad_reduce", Cc, wrows, vptr(wslab), vptr(gw), S())
            torch.cuda.synchronize()
            assert torch.equal(g5, g2)
            out[f"wg_add{aff}"] = gw
        res.append(out)
    L.call("dc_set_option", b"dw_pipe", 1)
    a, b = res
    for k in ("fwd", "fwd_bn_relu", "fwd_bn", "dgrad", "dgrad_add"):
        assert torch.equal(a[k], b[k]), k
    for relu in (1, 0):
        for u, v in zip(a[f"stats{relu}"], b[f"stats{relu}"]):
            np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=3e-5, atol=3e-5 * float(v.abs().max()))
    for aff in (0, 1):
        u, v = a[f"wg_add{aff}"], b[f"wg_add{aff}"]
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=3e-5, atol=3e-5 * float(v.abs().max()))


@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("with_addend", [True, False])
@pytest.mark.parametrize("case", [c for c in PIPE_CASES if c[2] == 1], ids=lambda c: c[0])
def test_depthwise_dgrad_takes_the_residual_batchnorm_sums(case, with_addend, relu):
    """dc_dwconv_dgrad_wgrad_bnres = dc_dwconv_dgrad_wgrad (same dx and weight-gradient rows, bit for bit) + the sums dc_bn_bwd_reduce
    (mask from the stored output) would take from that dx, the BatchNorm input and the stored block output."""
    _, Cc, dil, N, H, W = case
    dt, lib = L.DC_BF16, L.load()
    rows = lib.dc_dwconv_dgrad_wgrad_bnres_rows(dt, Cc, 1, dil, N, H, W)
    assert rows > 0 and rows == lib.dc_dwconv_dgrad_wgrad_rows(dt, Cc, 1, dil, N, H, W)
    xo = q(rnd(N, Cc, H, W, seed=1), torch.bfloat16)            # stored block output (its sign is the ReLU mask)
    ybn = q(rnd(N, Cc, H, W, seed=7), torch.bfloat16)           # input of the BatchNorm that produced it
    gy = q(rnd(N, Cc, H, W, seed=3), torch.bfloat16)
    add = q(rnd(N, Cc, H, W, seed=4), torch.bfloat16)
    wm = rnd(Cc, 1, 3, 3, seed=2, scale=1 / 3).to(dev())
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())
    mean, invstd = rnd(Cc, seed=8, scale=0.2).to(dev()), (torch.rand(Cc) + 0.5).to(dev())
    ld = (Cc + 31) // 32 * 32
    _, xv = to_nhwc(xo, torch.bfloat16, ld=ld)
    _, yv = to_nhwc(ybn, torch.bfloat16, ld=ld)
    _, gyv = to_nhwc(gy, torch.bfloat16)
    res = []
    for fused in (True, False):
        _, dx = to_nhwc(add, torch.bfloat16)                     # dx starts as the other consumers' contribution
        ap, al = (vptr(dx), Cc) if with_addend else (None, 0)
        wslab = torch.full((rows, 9, Cc), float("nan"), device=dev())
        slab = torch.full((2, rows, Cc), float("nan"), device=dev())
        if fused:
            L.call("dc_dwconv_dgrad_wgrad_bnres", dt, Cc, 1, dil, N, H, W, vptr(gyv), Cc, vptr(wd), ap, al, vptr(dx), Cc, vptr(xv), ld, vptr(wslab),
                   vptr(yv), ld, vptr(mean), vptr(invstd), relu, vptr(slab), S())
            srows = rows
        else:
            L.call("dc_dwconv_dgrad_wgrad", dt, Cc, 1, dil, N, H, W, vptr(gyv), Cc, vptr(wd), ap, al, vptr(dx), Cc, vptr(xv), ld, None, None, 0,
                   vptr(wslab), S())
            M = N * H * W
            srows = lib.dc_bn_stat_rows(M)
            slab = torch.empty((2, srows, Cc), device=dev())
            L.call("dc_bn_bwd_reduce", dt, M, Cc, vptr(dx), Cc, vptr(yv), ld, vptr(xv), ld, relu, vptr(mean), vptr(invstd), vptr(slab), None, None, S())
        dgam, dbet = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
        L.call("dc_bn_bwd_finalize", Cc, vptr(slab), srows, vptr(dgam), vptr(dbet), S())
        torch.cuda.synchronize()
        res.append((dx.clone(), wslab.clone(), dgam, dbet))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for u, v in zip(res[0][2:], res[1][2:]):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=3e-5, atol=3e-5 * float(v.abs().max()))


@pytest.mark.parametrize("case", [c for c in PIPE_CASES if c[0] in ("728", "728_b8", "1024d2", "64_half_block")], ids=lambda c: c[0])
def test_depthwise_dgrad_batchnorm_sums_into_a_sum_row(case):
    """The BatchNorm-backward sums of the persistent depthwise data gradient as ONE fp64 row (dc_dwconv_dgrad_sum_row_ok):
    dc_dwconv_dgrad_bnstats_wgrad_sum and dc_dwconv_dgrad_wgrad_bnres_sum leave dx and the weight-gradient rows of the row-slab forms, bit for bit,
    and a sum row that IS the fp64 column sum of their slab rows; dc_bn_bwd_finalize and dc_bn_bwd_apply_fin take it with rows = -1 and
    return the bits of the row-slab path (the BatchNorm backward inside a Block, deeplab_xception.py:104-119 under train_hdf5_ddp.py:363)."""
    _, Cc, dil, N, H, W = case
    dt, lib, dtype = L.DC_BF16, L.load(), torch.bfloat16
    assert lib.dc_dwconv_dgrad_sum_row_ok(dt, Cc, 1, dil, N, H, W) == 1
    rows = lib.dc_dwconv_dgrad_wgrad_rows(dt, Cc, 1, dil, N, H, W)
    assert rows > 0 and rows == lib.dc_dwconv_dgrad_bnstats_rows(dt, Cc, 1, dil, N, H, W)
    M = N * H * W
    ybn = q(rnd(N, Cc, H, W, seed=7), dtype)
    gy = q(rnd(N, Cc, H, W, seed=3), dtype)
    wm = rnd(Cc, 1, 3, 3, seed=2, scale=1 / 3).to(dev())
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())
    mean, invstd = rnd(Cc, seed=8, scale=0.2).to(dev()), (torch.rand(Cc) + 0.5).to(dev())
    msc, msh = (torch.rand(Cc) + 0.5).to(dev()), rnd(Cc, seed=9, scale=0.2).to(dev())
    gamma = (torch.rand(Cc) + 0.5).to(dev())
    ld = (Cc + 31) // 32 * 32
    _, yv = to_nhwc(ybn, dtype, ld=ld)
    _, gyv = to_nhwc(gy, dtype)

    def run(sum_row):
        _, dx = empty_nhwc(N, H, W, Cc, dtype)
        wslab = torch.full((rows, 9, Cc), float("nan"), device=dev())
        if sum_row:
            slab = torch.zeros(2 * Cc + 4, dtype=torch.float64, device=dev())
            slab[2 * Cc:] = float("nan")
        else:
            slab = torch.full((2, rows, Cc), float("nan"), device=dev())
        L.call("dc_dwconv_dgrad_bnstats_wgrad_sum" if sum_row else "dc_dwconv_dgrad_bnstats_wgrad", dt, Cc, 1, dil, N, H, W, vptr(gyv), Cc, vptr(wd),
               vptr(dx), Cc, vptr(yv), ld, vptr(mean), vptr(invstd), vptr(msc), vptr(msh), 1, vptr(slab), vptr(wslab), S())
        torch.cuda.synchronize()
        return dx, wslab, slab

    dx_r, ws_r, slab_r = run(False)
    dx_s, ws_s, slab_s = run(True)
    assert torch.equal(dx_r, dx_s) and torch.equal(ws_r, ws_s)
    assert torch.equal(slab_s[:2 * Cc], slab_r.double().sum(1).reshape(-1)) and torch.isnan(slab_s[2 * Cc:]).all()
    # the consumers: finalize alone, and the apply that runs it itself
    outs = []
    for slab, r in ((slab_r, rows), (slab_s, -1)):
        dgam, dbet = torch.full((Cc,), float("nan"), device=dev()), torch.full((Cc,), float("nan"), device=dev())
        L.call("dc_bn_bwd_finalize", Cc, vptr(slab), r, vptr(dgam), vptr(dbet), S())
        _, dyv = empty_nhwc(N, H, W, Cc, dtype)
        L.call("dc_bn_bwd_apply", dt, M, Cc, M, vptr(dx_r), Cc, vptr(yv), ld, None, 0, 2, vptr(gamma), vptr(mean), vptr(invstd), vptr(dgam), vptr(dbet),
               vptr(dyv), Cc, None, 0, vptr(msc), vptr(msh), S())
        dgam2, dbet2 = torch.full((Cc,), float("nan"), device=dev()), torch.full((Cc,), float("nan"), device=dev())
        _, dyv2 = empty_nhwc(N, H, W, Cc, dtype)
        if r == -1 or r <= lib.dc_bn_bwd_apply_fin_max_rows():
            L.call("dc_bn_bwd_apply_fin", dt, M, Cc, M, vptr(dx_r), Cc, vptr(yv), ld, None, 0, 2, vptr(gamma), vptr(mean), vptr(invstd), vptr(slab), r,
                   vptr(dgam2), vptr(dbet2), vptr(dyv2), Cc, None, 0, vptr(msc), vptr(msh), S())
            torch.cuda.synchronize()
            assert torch.equal(dgam2, dgam) and torch.equal(dbet2, dbet) and torch.equal(dyv2.float(), dyv.float())
        torch.cuda.synchronize()
        outs.append((dgam, dbet, dyv))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u.float(), v.float())
    if dil != 1:
        return
    # the residual form: sums of the BatchNorm whose output (plus a residual, through a ReLU) is this layer's stored input
    xo = q(rnd(N, Cc, H, W, seed=1), dtype)
    add = q(rnd(N, Cc, H, W, seed=4), dtype)
    _, xv = to_nhwc(xo, dtype, ld=ld)
    res = []
    for sum_row in (False, True):
        _, dx = to_nhwc(add, dtype)
        wslab = torch.full((rows, 9, Cc), float("nan"), device=dev())
        slab = torch.zeros(2 * Cc, dtype=torch.float64, device=dev()) if sum_row else torch.full((2, rows, Cc), float("nan"), device=dev())
        L.call("dc_dwconv_dgrad_wgrad_bnres_sum" if sum_row else "dc_dwconv_dgrad_wgrad_bnres", dt, Cc, 1, dil, N, H, W, vptr(gyv), Cc, vptr(wd),
               vptr(dx), Cc, vptr(dx), Cc, vptr(xv), ld, vptr(wslab), vptr(yv), ld, vptr(mean), vptr(invstd), 1, vptr(slab), S())
        torch.cuda.synchronize()
        res.append((dx, wslab, slab))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[1][2], res[0][2].double().sum(1).reshape(-1))


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [c for c in DW_CASES if c[0] in ("s1", "d2", "thin64", "s2_728", "s2_64", "odd")], ids=lambda c: c[0])
@pytest.mark.parametrize("relu", [1, 0])
def test_depthwise_dgrad_with_fused_bn_backward_statistics(case, dtype, relu):
    """dc_dwconv_dgrad_bnstats = dc_dwconv_dgrad (same dx, bit for bit) + the per-channel sums dc_bn_bwd_reduce would produce
    from that dx and the BatchNorm input (mask recomputed from y), finished by dc_bn_bwd_finalize."""
    _, Cc, stride, dil, N, H, W = case
    dt = L.dtype_code(dtype)
    rows = L.load().dc_dwconv_dgrad_bnstats_rows(dt, Cc, stride, dil, N, H, W)
    if rows == 0:
        pytest.skip("shape not served by the tiled kernels")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    gy = q(rnd(N, Cc, Ho, Wo, seed=3), dtype)
    ybn = q(rnd(N, Cc, H, W, seed=7), dtype)
    wm = rnd(Cc, 1, 3, 3, seed=2, scale=1 / 3).to(dev())
    wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())
    mean, invstd = rnd(Cc, seed=8, scale=0.2).to(dev()), (torch.rand(Cc) + 0.5).to(dev())
    sc, sh = (torch.rand(Cc) + 0.5).to(dev()), rnd(Cc, seed=9, scale=0.3).to(dev())
    _, gyv = to_nhwc(gy, dtype)
    _, yv = to_nhwc(ybn, dtype, ld=Cc + 8)
    _, gx_a = empty_nhwc(N, H, W, Cc, dtype)
    _, gx_b = empty_nhwc(N, H, W, Cc, dtype)
    slab = torch.full((2, rows, Cc), float("nan"), device=dev())
    L.call("dc_dwconv_dgrad_bnstats", dt, Cc, stride, dil, N, H, W, vptr(gyv), Cc, vptr(wd), vptr(gx_a), Cc, vptr(yv), Cc + 8,
           vptr(mean), vptr(invstd), vptr(sc), vptr(sh), relu, vptr(slab), S())
    L.call("dc_dwconv_dgrad", dt, Cc, stride, dil, N, H, W, vptr(gyv), Cc, vptr(wd), None, 0, vptr(gx_b), Cc, S())
    torch.cuda.synchronize()
    assert torch.equal(gx_a, gx_b)
    dgam, dbet = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    L.call("dc_bn_bwd_finalize", Cc, vptr(slab), rows, vptr(dgam), vptr(dbet), S())
    # reference sums from the stored dx and y, as dc_bn_bwd_reduce defines them
    M = N * H * W
    rows2 = L.load().dc_bn_stat_rows(M)
    slab2 = torch.empty((2, rows2, Cc), device=dev())
    dgam2, dbet2 = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    L.call("dc_bn_bwd_reduce", dt, M, Cc, vptr(gx_b), Cc, vptr(yv), Cc + 8, None, 0, 2 if relu else 0, vptr(mean), vptr(invstd), vptr(slab2),
           vptr(sc), vptr(sh), S())
    L.call("dc_bn_bwd_finalize", Cc, vptr(slab2), rows2, vptr(dgam2), vptr(dbet2), S())
    torch.cuda.synchronize()
    tol = dict(rtol=2e-5, atol=2e-4 * max(1.0, float(dbet2.abs().max())))
    assert torch.allclose(dbet, dbet2, **tol) and torch.allclose(dgam, dgam2, **tol)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", DW_CASES, ids=[c[0] for c in DW_CASES])
@pytest.mark.parametrize("prelu", [1, 0])
def test_depthwise_with_fused_bn_prologue(case, dtype, prelu):
    """dw(act(y*scale+shift)) with the BatchNorm(+ReLU) applied on load == dw of the materialised activation; and the BN
    backward that recomputes the ReLU mask from y (relu=2) == the one that reads the stored activation (relu=1)."""
    _, Cc, stride, dil, N, H, W = case
    dt = L.dtype_code(dtype)
    y = q(rnd(N, Cc, H, W, seed=1), dtype)
    scale = torch.rand(Cc, generator=torch.Generator().manual_seed(2)) + 0.5
    shift = rnd(Cc, seed=3, scale=0.3)
    a = y * scale[None, :, None, None] + shift[None, :, None, None]
    if prelu:
        a = F.relu(a)
    w = rnd(Cc, 1, 3, 3, seed=4, scale=1 / 3)
    ar, wr = a.clone().requires_grad_(True), w.clone().requires_grad_(True)
    oref = F.conv2d(F.pad(ar, (dil, dil, dil, dil)), wr, None, stride, 0, dil, groups=Cc)
    go = q(rnd(*oref.shape, seed=5), dtype)
    _, gw_ref = torch.autograd.grad(oref, (ar, wr), go)
    wm = w.to(dev()); wd = torch.empty(9 * Cc, device=dev())
    L.call("dc_dwconv_pack_weights", Cc, vptr(wm), vptr(wd), S())
    sc, sh = scale.to(dev()), shift.to(dev())
    _, yv = to_nhwc(y, dtype)
    Ho, Wo = oref.shape[2:]
    _, ov = empty_nhwc(N, Ho, Wo, Cc, dtype)
    L.call("dc_dwconv_fwd", dt, Cc, stride, dil, N, H, W, vptr(yv), Cc, vptr(wd), vptr(ov), Cc, vptr(sc), vptr(sh), prelu, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(ov), oref.detach(), dtype, bf16=1e-2)
    _, gov = to_nhwc(go, dtype)
    wsb = L.load().dc_dwconv_wgrad_workspace(Cc, N, H, W, stride)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    gw = torch.full((Cc, 1, 3, 3), float("nan"), device=dev())
    L.call("dc_dwconv_wgrad", dt, Cc, stride, dil, N, H, W, vptr(yv), Cc, vptr(gov), Cc, vptr(ws), vptr(gw), vptr(sc), vptr(sh), prelu, S())
    torch.cuda.synchronize()
    assert_close(gw.cpu(), gw_ref, dtype, f32=2e-4, bf16=3e-3)
    if prelu:
        # BN backward: mask recomputed from y vs mask read from the stored activation must give identical results
        M = N * H * W
        _, av = to_nhwc(a, dtype)
        da = q(rnd(N, Cc, H, W, seed=6), dtype)
        _, dav = to_nhwc(da, dtype)
        mean, inv = rnd(Cc, seed=7, scale=0.1).to(dev()), (torch.rand(Cc) + 0.5).to(dev())
        rows = L.load().dc_bn_stat_rows(M)
        res = []
        for relu_mode, outp in ((1, vptr(av)), (2, None)):
            slab = torch.empty(2, rows, Cc, device=dev())
            L.call("dc_bn_bwd_reduce", dt, M, Cc, vptr(dav), Cc, vptr(yv), Cc, outp, Cc, relu_mode, vptr(mean), vptr(inv), vptr(slab), vptr(sc), vptr(sh), S())
            dg, db = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
            L.call("dc_bn_bwd_finalize", Cc, vptr(slab), rows, vptr(dg), vptr(db), S())
            _, dyv = empty_nhwc(N, H, W, Cc, dtype)
            L.call("dc_bn_bwd_apply", dt, M, Cc, M, vptr(dav), Cc, vptr(yv), Cc, outp, Cc, relu_mode, vptr(sc), vptr(mean), vptr(inv), vptr(dg), vptr(db),
                   vptr(dyv), Cc, None, 0, vptr(sc), vptr(sh), S())
            torch.cuda.synchronize()
            res.append((dg.cpu(), db.cpu(), from_nhwc(dyv)))
        if dtype == torch.float32:
            for u, v in zip(res[0], res[1]):
                assert torch.equal(u, v)
        else:
            # bf16: the stored activation was rounded, so elements with 0 < a < 2^-133 do not exist; masks agree except where
            # y*scale+shift rounds to exactly zero in bf16 (measure zero for random data)
            for u, v in zip(res[0], res[1]):
                assert_close(u, v, dtype, bf16=1e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 728, 12, 10), (2, 32, 40, 33), (2, 256, 1, 1), (2, 64, 37, 29), (2, 128, 19, 23), (3, 24, 17, 9)],
                         ids=["c728", "c32", "pool", "c64", "c128", "c24"])
@pytest.mark.parametrize("relu,use_res", [(1, 0), (1, 1), (0, 1), (0, 0)])
def test_batchnorm_train_fwd_bwd(shape, dtype, relu, use_res):
    N, Cc, H, W = shape
    dt = L.dtype_code(dtype)
    M = N * H * W
    y = q(rnd(N, Cc, H, W, seed=1) * 2 + 0.5, dtype)
    res = q(rnd(N, Cc, H, W, seed=2), dtype) if use_res else None
    gamma = torch.rand(Cc, generator=torch.Generator().manual_seed(3)) + 0.5
    beta = rnd(Cc, seed=4, scale=0.3)
    rm0, rv0 = rnd(Cc, seed=5), torch.rand(Cc, generator=torch.Generator().manual_seed(6)) + 0.5
    # reference
    yr = y.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    o = F.batch_norm(yr, rm, rv, gr, br, True, 0.1, 1e-5)
    if use_res:
        rr = res.clone().requires_grad_(True)
        o = o + rr
    if relu:
        o = F.relu(o)
    go = q(rnd(N, Cc, H, W, seed=7), dtype)
    grads = torch.autograd.grad(o, (yr, gr, br) + ((rr,) if use_res else ()), go)
    # device
    _, yv = to_nhwc(y, dtype, ld=Cc + 8)
    rows = L.load().dc_bn_stat_rows(M)
    slab = torch.empty(2, rows, Cc, device=dev())
    L.call("dc_bn_stats", dt, M, Cc, vptr(yv), Cc + 8, vptr(slab), S())
    dv = lambda t: t.clone().to(dev())  # noqa: E731
    g_d, b_d, rm_d, rv_d = dv(gamma), dv(beta), dv(rm0), dv(rv0)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev())
    scale, shift, smean, sinv = (torch.empty(Cc, device=dev()) for _ in range(4))
    L.call("dc_bn_finalize", Cc, M, vptr(slab), rows, vptr(g_d), vptr(b_d), vptr(rm_d), vptr(rv_d), vptr(nbt), 0.1, 1e-5,
           vptr(scale), vptr(shift), vptr(smean), vptr(sinv), S())
    resv = to_nhwc(res, dtype)[1] if use_res else None
    _, ov = empty_nhwc(N, H, W, Cc, dtype, ld=Cc + 16, off=8)
    L.call("dc_bn_apply", dt, M, Cc, vptr(yv), Cc + 8, vptr(scale), vptr(shift), vptr(resv) if use_res else None, Cc, relu,
           vptr(ov), Cc + 16, S())
    # the row-layout kernel (default) and the grid-stride kernel do the same arithmetic: same bits, pad channels untouched
    ob2, ov2 = empty_nhwc(N, H, W, Cc, dtype, ld=Cc + 16, off=8)
    L.call("dc_set_option", b"bn_apply_rows", 0)
    L.call("dc_bn_apply", dt, M, Cc, vptr(yv), Cc + 8, vptr(scale), vptr(shift), vptr(resv) if use_res else None, Cc, relu,
           vptr(ov2), Cc + 16, S())
    L.call("dc_set_option", b"bn_apply_rows", 1)
    torch.cuda.synchronize()
    assert torch.equal(ov.float(), ov2.float()) and torch.isnan(ob2[..., :8].float()).all() and torch.isnan(ob2[..., 8 + Cc:].float()).all()
    assert int(nbt) == 1
    np.testing.assert_allclose(rm_d.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    assert_close(from_nhwc(ov), o.detach(), dtype, bf16=1e-2)
    # backward
    _, gov = to_nhwc(go, dtype)
    slab2 = torch.empty(2, rows, Cc, device=dev())
    L.call("dc_bn_bwd_reduce", dt, M, Cc, vptr(gov), Cc, vptr(yv), Cc + 8, vptr(ov), Cc + 16, relu, vptr(smean), vptr(sinv), vptr(slab2), None, None, S())
    dgamma, dbeta = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    L.call("dc_bn_bwd_finalize", Cc, vptr(slab2), rows, vptr(dgamma), vptr(dbeta), S())
    # the same pass into a sum row (dc_bn_bwd_reduce_sum): the fp64 column sums of those rows, and the same parameter gradients from it
    srow = torch.zeros(2 * Cc, dtype=torch.float64, device=dev())
    L.call("dc_bn_bwd_reduce_sum", dt, M, Cc, vptr(gov), Cc, vptr(yv), Cc + 8, vptr(ov), Cc + 16, relu, vptr(smean), vptr(sinv), vptr(srow), None, None, S())
    dgamma_s, dbeta_s = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    L.call("dc_bn_bwd_finalize", Cc, vptr(srow), -1, vptr(dgamma_s), vptr(dbeta_s), S())
    torch.cuda.synchronize()
    assert torch.equal(srow, slab2.double().sum(1).reshape(-1)) and torch.equal(dgamma_s, dgamma) and torch.equal(dbeta_s, dbeta)
    _, dyv = empty_nhwc(N, H, W, Cc, dtype)
    _, gv = empty_nhwc(N, H, W, Cc, dtype)
    L.call("dc_bn_bwd_apply", dt, M, Cc, M, vptr(gov), Cc, vptr(yv), Cc + 8, vptr(ov), Cc + 16, relu, vptr(g_d), vptr(smean),
           vptr(sinv), vptr(dgamma), vptr(dbeta), vptr(dyv), Cc, vptr(gv), Cc, None, None, S())
    torch.cuda.synchronize()
    # the finalize of a short slab inside the apply kernel (dc_bn_bwd_apply_fin): same bits as the two calls
    if rows <= L.load().dc_bn_bwd_apply_fin_max_rows():
        dgamma2, dbeta2 = torch.full((Cc,), float("nan"), device=dev()), torch.full((Cc,), float("nan"), device=dev())
        _, dyv2 = empty_nhwc(N, H, W, Cc, dtype)
        _, gv2 = empty_nhwc(N, H, W, Cc, dtype)
        L.call("dc_bn_bwd_apply_fin", dt, M, Cc, M, vptr(gov), Cc, vptr(yv), Cc + 8, vptr(ov), Cc + 16, relu, vptr(g_d), vptr(smean),
               vptr(sinv), vptr(slab2), rows, vptr(dgamma2), vptr(dbeta2), vptr(dyv2), Cc, vptr(gv2), Cc, None, None, S())
        torch.cuda.synchronize()
        assert torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta)
        assert torch.equal(dyv2.float(), dyv.float()) and torch.equal(gv2.float(), gv.float())
    with pytest.raises(L.DeepcamHipError):
        L.call("dc_bn_bwd_apply_fin", dt, M, Cc, M, vptr(gov), Cc, vptr(yv), Cc + 8, vptr(ov), Cc + 16, relu, vptr(g_d), vptr(smean),
               vptr(sinv), vptr(slab2), 65, vptr(dgamma), vptr(dbeta), vptr(dyv), Cc, vptr(gv), Cc, None, None, S())
    # with bf16 storage the ReLU mask is taken from the ROUNDED output; compare against a reference using that mask
    if M > 2:  # (a 2-sample BN has |xhat| == 1: dgamma/dy are cancellation-dominated, checked loosely below)
        assert_close(dgamma.cpu(), grads[1], dtype, f32=5e-4, bf16=2e-2)
        assert_close(from_nhwc(dyv), grads[0], dtype, f32=1e-3, bf16=4e-2)
    assert_close(dbeta.cpu(), grads[2], dtype, f32=5e-4, bf16=2e-2)
    if use_res:
        assert_close(from_nhwc(gv), grads[3], dtype, bf16=1e-2)


@pytest.mark.parametrize("rows,Cc", [(13824, 32), (3456, 256), (1025, 64), (4611, 24), (5000, 728), (1024, 32), (16385, 8)],
                         ids=["r13824c32", "r3456c256", "r1025c64_fallback", "r4611c24_ragged", "r5000c728", "r1024_single", "r16385_fallback"])
def test_finalize_of_large_slabs_in_two_stages(rows, Cc):
    """Slabs above 1024 rows (one row per 128-pixel tile of the 192 x 288 and 384 x 576 layers) are folded in two stages, the first leaving its fp64
    results in the slab (bn_fin.h); forward and backward finalize against fp64 column sums of the same slab."""
    g = torch.Generator().manual_seed(rows + Cc)
    M = rows * 128
    sx = torch.randn(rows, Cc, generator=g) * 20 + 64          # sum x per tile
    sq = sx * sx / 128 + torch.rand(rows, Cc, generator=g) * 300   # sum x^2 per tile (var > 0)
    slab = torch.stack([sx, sq]).contiguous().to(dev())
    ref_s, ref_q = sx.double().sum(0), sq.double().sum(0)
    gam, bet = (torch.rand(Cc, generator=g) + 0.5).to(dev()), torch.randn(Cc, generator=g).to(dev())
    scale, shift, smean, sinv = (torch.empty(Cc, device=dev()) for _ in range(4))
    slab_b = slab.clone()
    L.call("dc_bn_finalize", Cc, M, vptr(slab), rows, vptr(gam), vptr(bet), None, None, None, 0.1, 1e-5, vptr(scale), vptr(shift),
           vptr(smean), vptr(sinv), S())
    dg, db = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    L.call("dc_bn_bwd_finalize", Cc, vptr(slab_b), rows, vptr(dg), vptr(db), S())
    torch.cuda.synchronize()
    mean = ref_s / M
    var = (ref_q / M - mean * mean).clamp_min(0)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    np.testing.assert_allclose(smean.cpu().numpy(), mean.float().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sinv.cpu().numpy(), inv.float().numpy(), rtol=2e-6)
    np.testing.assert_allclose(scale.cpu().numpy(), (gam.cpu().double() * inv).float().numpy(), rtol=2e-6)
    np.testing.assert_allclose(db.cpu().numpy(), ref_s.float().numpy(), rtol=1e-6)          # "dbeta" = column sum of slab 0
    np.testing.assert_allclose(dg.cpu().numpy(), ref_q.float().numpy(), rtol=1e-6)


def test_bn_single_value_per_channel_raises():
    slab = torch.zeros(2, 1, 8, device=dev())
    v = torch.ones(8, device=dev())
    with pytest.raises(L.DeepcamHipError, match="Expected more than 1 value per channel"):
        L.call("dc_bn_finalize", 8, 1, vptr(slab), 1, vptr(v), vptr(v), None, None, None, 0.1, 1e-5, vptr(v), vptr(v), None, None, S())


def test_bn_eval_coeffs():
    Cc = 48
    g, b, rm = rnd(Cc, seed=1), rnd(Cc, seed=2), rnd(Cc, seed=3)
    rv = torch.rand(Cc) + 0.1
    sc, sh = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    gd, bd, rmd, rvd = g.to(dev()), b.to(dev()), rm.to(dev()), rv.to(dev())     # keep the device copies alive
    L.call("dc_bn_eval_coeffs", Cc, vptr(gd), vptr(bd), vptr(rmd), vptr(rvd), 1e-5, vptr(sc), vptr(sh), S())
    torch.cuda.synchronize()
    x = rnd(3, Cc, 2, 2, seed=4)
    ref = F.batch_norm(x, rm, rv, g, b, False, 0.1, 1e-5)
    got = x * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None]
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_stem(dtype):
    N, Cin, H, W = 2, 16, 20, 28
    dt = L.dtype_code(dtype)
    x = torch.rand(N, Cin, H, W, generator=torch.Generator().manual_seed(1))
    w = rnd(32, Cin, 3, 3, seed=2, scale=(2.0 / (Cin * 9)) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yref = F.conv2d(xr, wr, None, 2, 1)
    Ho, Wo = yref.shape[2:]
    xd, wd = x.to(dev()), w.to(dev())
    _, yv = empty_nhwc(N, Ho, Wo, 32, dtype)
    rows = L.load().dc_stem_stat_rows(N, H, W)
    slab = torch.empty(2, rows, 32, device=dev())
    L.call("dc_stem_fwd", dt, N, Cin, H, W, vptr(xd), vptr(wd), vptr(yv), 32, vptr(slab), S())
    torch.cuda.synchronize()
    y = from_nhwc(yv)
    assert_close(y, yref.detach(), dtype, bf16=1e-2)
    np.testing.assert_allclose(slab[0].sum(0).cpu().numpy(), y.sum((0, 2, 3)).numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(slab[1].sum(0).cpu().numpy(), (y * y).sum((0, 2, 3)).numpy(), rtol=1e-3)
    gy = q(rnd(*yref.shape, seed=3), dtype)
    (gw_ref,) = torch.autograd.grad(yref, (wr,), gy)
    _, gyv = to_nhwc(gy, dtype)
    wsb = L.load().dc_stem_wgrad_workspace(N, Cin, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    gw = torch.full((32, Cin, 3, 3), float("nan"), device=dev())
    L.call("dc_stem_wgrad", dt, N, Cin, H, W, vptr(xd), vptr(gyv), 32, vptr(ws), vptr(gw), S())
    torch.cuda.synchronize()
    assert_close(gw.cpu(), gw_ref, dtype, f32=2e-4, bf16=2e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_head(dtype):
    N, Cin, H, W = 2, 256, 6, 9
    dt = L.dtype_code(dtype)
    x = q(rnd(N, Cin, H, W, seed=1), dtype)
    w = rnd(Cin, 3, 3, 3, seed=2, scale=0.05)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, None, 2, 1, 1)
    gl = rnd(*ref.shape, seed=3)
    gx_ref, gw_ref = torch.autograd.grad(ref, (xr, wr), gl)
    _, xv = to_nhwc(x, dtype, ld=Cin + 8)
    wd = w.to(dev())
    wsb = L.load().dc_head_workspace(dt, N, Cin, H, W)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev())
    wsp = C.c_void_p((ws.data_ptr() + 255) // 256 * 256)
    out = torch.full((N, 3, 2 * H, 2 * W), float("nan"), device=dev())
    L.call("dc_head_fwd", dt, N, Cin, H, W, vptr(xv), Cin + 8, vptr(wd), vptr(out), wsp, S())
    torch.cuda.synchronize()
    # fp32 accumulate and fp32 store; in bf16 mode the WEIGHTS of the GEMM are bf16-rounded
    ref_used = F.conv_transpose2d(x, q(w, dtype), None, 2, 1, 1) if dtype == torch.bfloat16 else ref.detach()
    assert_close(out.cpu(), ref_used, torch.float32, f32=2e-4)
    gld = gl.to(dev())
    _, gxv = empty_nhwc(N, H, W, Cin, dtype)
    gw = torch.full((Cin, 3, 3, 3), float("nan"), device=dev())
    L.call("dc_head_bwd", dt, N, Cin, H, W, vptr(xv), Cin + 8, vptr(gld), vptr(wd), vptr(gxv), Cin, vptr(gw), wsp, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(gxv), gx_ref, dtype, bf16=1.5e-2)
    assert_close(gw.cpu(), gw_ref, dtype, f32=2e-4, bf16=1e-2)
    # the same call with the BatchNorm-backward sums of the layer in front of the head taken in the data gradient's epilogue
    M = N * H * W
    ybn = q(rnd(N, Cin, H, W, seed=5), dtype)
    _, ybv = to_nhwc(ybn, dtype)
    mean, invstd = rnd(Cin, seed=6, scale=0.3).to(dev()), (rnd(Cin, seed=7).abs() + 0.5).to(dev())
    mscale, mshift = rnd(Cin, seed=8).to(dev()), rnd(Cin, seed=9, scale=0.5).to(dev())
    rows = (M + 127) // 128
    slab = torch.full((2, rows, Cin), float("nan"), device=dev())
    _, gxv2 = empty_nhwc(N, H, W, Cin, dtype)
    gw2 = torch.full((Cin, 3, 3, 3), float("nan"), device=dev())
    L.call("dc_head_bwd_bnstats", dt, N, Cin, H, W, vptr(xv), Cin + 8, vptr(gld), vptr(wd), vptr(gxv2), Cin, vptr(gw2), wsp, vptr(ybv), Cin,
           vptr(mean), vptr(invstd), vptr(mscale), vptr(mshift), 1, vptr(slab), S())
    rrows = L.load().dc_bn_stat_rows(M)
    rslab = torch.zeros(2, rrows, Cin, device=dev())
    L.call("dc_bn_bwd_reduce", dt, M, Cin, vptr(gxv), Cin, vptr(ybv), Cin, None, 0, 2, vptr(mean), vptr(invstd), vptr(rslab), vptr(mscale),
           vptr(mshift), S())
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(gxv2), from_nhwc(gxv)) and torch.equal(gw2, gw)
    got, ref2 = slab.double().sum(1).cpu(), rslab.double().sum(1).cpu()
    assert (got - ref2).abs().max().item() <= 2e-5 * (ref2.abs().max().item() + 1e-12) + 1e-6
    if dtype == torch.bfloat16:
        # the streaming data-gradient kernel of the head (one K step, no LDS) against the tiled GEMM kernels: the same MFMA per element
        try:
            L.call("dc_set_option", b"head_dgrad_fused", 0)
            slab3 = torch.full((2, rows, Cin), float("nan"), device=dev())
            _, gxv3 = empty_nhwc(N, H, W, Cin, dtype)
            gw3 = torch.full((Cin, 3, 3, 3), float("nan"), device=dev())
            L.call("dc_head_bwd_bnstats", dt, N, Cin, H, W, vptr(xv), Cin + 8, vptr(gld), vptr(wd), vptr(gxv3), Cin, vptr(gw3), wsp, vptr(ybv), Cin,
                   vptr(mean), vptr(invstd), vptr(mscale), vptr(mshift), 1, vptr(slab3), S())
            torch.cuda.synchronize()
        finally:
            L.call("dc_set_option", b"head_dgrad_fused", 1)
        assert torch.equal(from_nhwc(gxv3), from_nhwc(gxv2)) and torch.equal(gw3, gw2)
        a3, a2 = slab3.double().sum(1).cpu(), slab.double().sum(1).cpu()
        assert (a3 - a2).abs().max().item() <= 2e-5 * (a2.abs().max().item() + 1e-12) + 1e-6


@pytest.mark.parametrize("shape", [(2, 20, 70), (1, 9, 229), (1, 8, 32)], ids=["3x3tiles", "8tiles_wide", "one_exact_tile"])
def test_head_fused_forward_matches_gemm_plus_combine(shape):
    """bf16 classifier head: the fused products+combination kernel equals the GEMM + combine kernels bit for bit (same fp32
    products, same addition order) over several tiles, ragged edges and more tiles than one workgroup walks."""
    N, H, W = shape
    Cin = 256
    x = q(rnd(N, Cin, H, W, seed=5), torch.bfloat16)
    wd = rnd(Cin, 3, 3, 3, seed=6, scale=0.05).to(dev())
    _, xv = to_nhwc(x, torch.bfloat16)
    wsb = L.load().dc_head_workspace(L.DC_BF16, N, Cin, H, W)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev())
    wsp = C.c_void_p((ws.data_ptr() + 255) // 256 * 256)
    outs = []
    for mode in (1, 0):
        L.call("dc_set_option", b"head_fused", mode)
        out = torch.full((N, 3, 2 * H, 2 * W), float("nan"), device=dev())
        L.call("dc_head_fwd", L.DC_BF16, N, Cin, H, W, vptr(xv), Cin, vptr(wd), vptr(out), wsp, S())
        torch.cuda.synchronize()
        outs.append(out.clone())
    L.call("dc_set_option", b"head_fused", 1)
    assert torch.equal(outs[0], outs[1])
    ref = F.conv_transpose2d(x, q(wd.cpu(), torch.bfloat16), None, 2, 1, 1)
    assert_close(outs[0].cpu(), ref, torch.float32, f32=2e-4)


@pytest.mark.parametrize("lab_dtype", [torch.int64, torch.uint8])
@pytest.mark.parametrize("shape", [(2, 20, 70), (1, 9, 229), (1, 8, 32)], ids=["3x3tiles", "8tiles_wide", "one_exact_tile"])
def test_head_forward_with_fused_loss(shape, lab_dtype):
    """dc_head_fwd_loss = dc_head_fwd followed by dc_wce_fused: logits, logit gradient, predictions and IoU counts bit for bit, the
    fp64 loss sum up to the order of its atomics; also with the logits not stored at all, and with an out-of-range label."""
    N, H, W = shape
    Cin, dtype = 256, torch.bfloat16
    dt = L.dtype_code(dtype)
    x = q(rnd(N, Cin, H, W, seed=1), dtype)
    w = rnd(Cin, 3, 3, 3, seed=2, scale=0.05)
    _, xv = to_nhwc(x, dtype)
    wd = w.to(dev())
    ws = torch.empty(L.load().dc_head_workspace(dt, N, Cin, H, W) + 256, dtype=torch.uint8, device=dev())
    wsp = C.c_void_p((ws.data_ptr() + 255) // 256 * 256)
    g = torch.Generator().manual_seed(7)
    labels = torch.randint(0, 3, (N, 2 * H, 2 * W), generator=g).to(lab_dtype).to(dev())
    cw = torch.tensor([0.9, 2.6, 1.7], device=dev())
    scale = 1.0 / labels.numel()
    ref_logits = torch.full((N, 3, 2 * H, 2 * W), float("nan"), device=dev())
    L.call("dc_head_fwd", dt, N, Cin, H, W, vptr(xv), Cin, vptr(wd), vptr(ref_logits), wsp, S())
    ref_ls = torch.zeros(1, dtype=torch.float64, device=dev())
    ref_dl, ref_pred = torch.full_like(ref_logits, float("nan")), torch.full((N, 2 * H, 2 * W), -1, dtype=torch.int64, device=dev())
    ref_cnt = torch.zeros(9, dtype=torch.int64, device=dev())
    L.call("dc_wce_fused", N, 2 * H, 2 * W, vptr(ref_logits), vptr(labels), labels.element_size(), vptr(cw), scale, vptr(ref_ls), vptr(ref_dl),
           vptr(ref_pred), vptr(ref_cnt), S())
    for store in (True, False):
        logits = torch.full_like(ref_logits, float("nan"))
        ls = torch.zeros(1, dtype=torch.float64, device=dev())
        dl, pred = torch.full_like(ref_logits, float("nan")), torch.full_like(ref_pred, -1)
        cnt = torch.zeros(9, dtype=torch.int64, device=dev())
        L.call("dc_head_fwd_loss", dt, N, Cin, H, W, vptr(xv), Cin, vptr(wd), vptr(logits) if store else None, wsp, vptr(labels),
               labels.element_size(), vptr(cw), scale, vptr(ls), vptr(dl), vptr(pred), vptr(cnt), S())
        torch.cuda.synchronize()
        if store:
            assert torch.equal(logits, ref_logits)
        else:
            assert torch.isnan(logits).all()
        assert torch.equal(dl, ref_dl) and torch.equal(pred, ref_pred) and torch.equal(cnt, ref_cnt)
        assert float(ls) == pytest.approx(float(ref_ls), rel=1e-12)
    bad = labels.clone()
    bad[0, 1, 2] = 7
    ls = torch.zeros(1, dtype=torch.float64, device=dev())
    dl = torch.zeros_like(ref_logits)
    L.call("dc_head_fwd_loss", dt, N, Cin, H, W, vptr(xv), Cin, vptr(wd), None, wsp, vptr(bad), bad.element_size(), vptr(cw), scale, vptr(ls),
           vptr(dl), None, None, S())
    torch.cuda.synchronize()
    assert torch.isnan(ls).all() and torch.isnan(dl[0, :, 1, 2]).all() and int(torch.isnan(dl).sum()) == 3


BNFIN_CASES = [
    # name, C, dil, N, H, W, rows            (rows: partial-sum rows of the slab, as the pointwise conv in front would leave them)
    ("middle_flow_like", 728, 1, 2, 24, 36, 14),      # three channel blocks, the last one ragged (728 = 2 * 256 + 216)
    ("one_tile", 64, 1, 1, 8, 8, 3),                  # ONE pixel tile per channel block: a single leader does every finalize block
    ("dilated", 256, 2, 1, 16, 24, 70),               # more rows than the 64 row lanes of a finalize block
    ("thin", 128, 1, 2, 40, 24, 600),                 # 16-group workgroups; a slab longer than one eight-row sweep per lane
]


@pytest.mark.parametrize("relu", [1, 0], ids=["relu", "affine"])
@pytest.mark.parametrize("shape", [(2, 20, 70), (1, 9, 229), (2, 8, 32)], ids=["3x3tiles", "8tiles_wide", "exact_tiles"])
def test_head_on_unstored_batchnorm_output(shape, relu):
    """dc_head_fwd_bnin / dc_head_fwd_loss_bnin / dc_head_bwd_bnin (the head reads the raw output y of the convolution in front of the
    BatchNorm and forms act(y * scale + shift) itself) against dc_bn_apply followed by the plain head calls: logits, loss, logit
    gradient, predictions, counts, data gradient and BatchNorm-backward sums bit for bit; the weight gradient (another kernel: the
    register-staged 128-tile one) to rounding."""
    N, H, W = shape
    Cin, dtype = 256, torch.bfloat16
    dt = L.dtype_code(dtype)
    M = N * H * W
    y = q(rnd(N, Cin, H, W, seed=1), dtype)
    w = rnd(Cin, 3, 3, 3, seed=2, scale=0.05)
    _, yv = to_nhwc(y, dtype, ld=Cin + 16, off=8)
    wd = w.to(dev())
    scale, shift = (rnd(Cin, seed=3).abs() + 0.3).to(dev()), rnd(Cin, seed=4, scale=0.4).to(dev())
    mean, invstd = rnd(Cin, seed=6, scale=0.3).to(dev()), (rnd(Cin, seed=7).abs() + 0.5).to(dev())
    ws = torch.empty(L.load().dc_head_workspace(dt, N, Cin, H, W) + 256, dtype=torch.uint8, device=dev())
    wsp = C.c_void_p((ws.data_ptr() + 255) // 256 * 256)
    g = torch.Generator().manual_seed(7)
    labels = torch.randint(0, 3, (N, 2 * H, 2 * W), generator=g).to(dev())
    cw = torch.tensor([0.9, 2.6, 1.7], device=dev())
    gs = 1.0 / labels.numel()
    # reference: the stored activation, then the plain calls
    _, av = empty_nhwc(N, H, W, Cin, dtype)
    L.call("dc_bn_apply", dt, M, Cin, vptr(yv), Cin + 16, vptr(scale), vptr(shift), None, 0, relu, vptr(av), Cin, S())
    ref_logits = torch.full((N, 3, 2 * H, 2 * W), float("nan"), device=dev())
    ref_ls = torch.zeros(1, dtype=torch.float64, device=dev())
    ref_dl, ref_pred = torch.full_like(ref_logits, float("nan")), torch.full((N, 2 * H, 2 * W), -1, dtype=torch.int64, device=dev())
    ref_cnt = torch.zeros(9, dtype=torch.int64, device=dev())
    L.call("dc_head_fwd_loss", dt, N, Cin, H, W, vptr(av), Cin, vptr(wd), vptr(ref_logits), wsp, vptr(labels), 8, vptr(cw), gs, vptr(ref_ls),
           vptr(ref_dl), vptr(ref_pred), vptr(ref_cnt), S())
    rows = (M + 127) // 128
    ref_slab = torch.full((2, rows, Cin), float("nan"), device=dev())
    _, ref_dx = empty_nhwc(N, H, W, Cin, dtype)
    ref_gw = torch.full((Cin, 3, 3, 3), float("nan"), device=dev())
    L.call("dc_head_bwd_bnstats", dt, N, Cin, H, W, vptr(av), Cin, vptr(ref_dl), vptr(wd), vptr(ref_dx), Cin, vptr(ref_gw), wsp, vptr(yv), Cin + 16,
           vptr(mean), vptr(invstd), vptr(scale), vptr(shift), relu, vptr(ref_slab), S())
    torch.cuda.synchronize()
    # the activation never stored
    logits = torch.full_like(ref_logits, float("nan"))
    L.call("dc_head_fwd_bnin", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(wd), vptr(logits), wsp, S())
    torch.cuda.synchronize()
    assert torch.equal(logits, ref_logits)
    for store in (True, False):
        logits = torch.full_like(ref_logits, float("nan"))
        ls = torch.zeros(1, dtype=torch.float64, device=dev())
        dl, pred = torch.full_like(ref_logits, float("nan")), torch.full_like(ref_pred, -1)
        cnt = torch.zeros(9, dtype=torch.int64, device=dev())
        L.call("dc_head_fwd_loss_bnin", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(wd),
               vptr(logits) if store else None, wsp, vptr(labels), 8, vptr(cw), gs, vptr(ls), vptr(dl), vptr(pred), vptr(cnt), S())
        torch.cuda.synchronize()
        assert torch.equal(logits, ref_logits) if store else torch.isnan(logits).all()
        assert torch.equal(dl, ref_dl) and torch.equal(pred, ref_pred) and torch.equal(cnt, ref_cnt)
        assert float(ls) == pytest.approx(float(ref_ls), rel=1e-12)
    for with_sums in (True, False):
        slab = torch.full((2, rows, Cin), float("nan"), device=dev())
        _, dx = empty_nhwc(N, H, W, Cin, dtype)
        gw = torch.full((Cin, 3, 3, 3), float("nan"), device=dev())
        for parts in ((3,) if with_sums else (1, 2)):          # in one call, or chain part and weight-gradient part one after the other
            L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(ref_dl), vptr(wd), vptr(dx), Cin,
                   vptr(gw), wsp, vptr(mean), vptr(invstd), vptr(slab) if with_sums else None, parts, S())
        torch.cuda.synchronize()
        assert torch.equal(from_nhwc(dx), from_nhwc(ref_dx))
        assert torch.equal(slab, ref_slab) if with_sums else torch.isnan(slab).all()
        assert not torch.isnan(gw).any()
        assert (gw - ref_gw).abs().max().item() <= 1e-5 * ref_gw.abs().max().item()
        print(f"[head weight gradient, register-staged vs LDS-DMA kernel] bit-equal: {torch.equal(gw, ref_gw)}")
    # two passes, the data gradient never stored: sums with dx = NULL, dc_bn_bwd_finalize, then dc_head_bwd_bnin_apply writes the BatchNorm
    # input's gradient -- against the stored dx + dc_bn_bwd_apply (mask recomputed from y), bit for bit
    gamma = (rnd(Cin, seed=11).abs() + 0.4).to(dev())
    dg, db = torch.empty(Cin, device=dev()), torch.empty(Cin, device=dev())
    slab_c = ref_slab.clone()               # (the finalize consumes its slab)
    L.call("dc_bn_bwd_finalize", Cin, vptr(slab_c), rows, vptr(dg), vptr(db), S())
    _, ref_dy = empty_nhwc(N, H, W, Cin, dtype, ld=Cin + 8)
    L.call("dc_bn_bwd_apply", dt, M, Cin, M, vptr(ref_dx), Cin, vptr(yv), Cin + 16, None, 0, 2 if relu else 0, vptr(gamma), vptr(mean), vptr(invstd),
           vptr(dg), vptr(db), vptr(ref_dy), Cin + 8, None, 0, vptr(scale), vptr(shift), S())
    slab = torch.full((2, rows, Cin), float("nan"), device=dev())
    gw = torch.full((Cin, 3, 3, 3), float("nan"), device=dev())
    L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(ref_dl), vptr(wd), None, 0,
           vptr(gw), wsp, vptr(mean), vptr(invstd), vptr(slab), 1, S())
    torch.cuda.synchronize()
    assert torch.equal(slab, ref_slab)
    dg2, db2 = torch.empty(Cin, device=dev()), torch.empty(Cin, device=dev())
    L.call("dc_bn_bwd_finalize", Cin, vptr(slab), rows, vptr(dg2), vptr(db2), S())
    _, dy = empty_nhwc(N, H, W, Cin, dtype, ld=Cin + 8)
    L.call("dc_head_bwd_bnin_apply", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(gamma), vptr(mean), vptr(invstd),
           vptr(dg2), vptr(db2), M, vptr(dy), Cin + 8, wsp, S())
    L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(ref_dl), vptr(wd), None, 0,
           vptr(gw), wsp, vptr(mean), vptr(invstd), None, 2, S())                      # the weight-gradient part still runs without a dx
    torch.cuda.synchronize()
    assert torch.equal(dg, dg2) and torch.equal(db, db2)
    assert torch.equal(from_nhwc(dy), from_nhwc(ref_dy))
    assert (gw - ref_gw).abs().max().item() <= 1e-5 * ref_gw.abs().max().item()
    with pytest.raises(L.DeepcamHipError, match="dx may be NULL only"):
        L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(ref_dl), vptr(wd), None, 0,
               vptr(gw), wsp, vptr(mean), vptr(invstd), None, 1, S())
    # both parts in ONE call without a dx: the weight gradient rides on the statistics pass (no second read of the BatchNorm input); same sums bit
    # for bit, the weight gradient to the rounding of another pixel order; with the switch off: the two passes one after the other
    for switch in (1, 0):
        L.call("dc_set_option", b"head_wgrad_fused", switch)
        slab3 = torch.full((2, rows, Cin), float("nan"), device=dev())
        gw3 = torch.full((Cin, 3, 3, 3), float("nan"), device=dev())
        L.call("dc_head_bwd_bnin", dt, N, Cin, H, W, vptr(yv), Cin + 16, vptr(scale), vptr(shift), relu, vptr(ref_dl), vptr(wd), None, 0,
               vptr(gw3), wsp, vptr(mean), vptr(invstd), vptr(slab3), 3, S())
        torch.cuda.synchronize()
        assert torch.equal(slab3, ref_slab), switch
        assert not torch.isnan(gw3).any()
        assert (gw3 - ref_gw).abs().max().item() <= 1e-5 * ref_gw.abs().max().item(), switch
    L.call("dc_set_option", b"head_wgrad_fused", 1)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_nchw_to_nhwc(dtype):
    x = rnd(2, 16, 9, 13, seed=1)
    xd = x.to(dev())
    _, ov = empty_nhwc(2, 9, 13, 16, dtype, ld=24, off=8)
    L.call("dc_nchw_to_nhwc", L.dtype_code(dtype), 2, 16, 9, 13, vptr(xd), vptr(ov), 24, S())
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(ov), q(x, dtype))


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("H,W", [(6, 5), (13, 11), (48, 72)], ids=["hw30", "hw143_ragged", "hw3456_real"])
def test_pool_branch_helpers(dtype, H, W):
    N, Cc = 2, 2048
    dt = L.dtype_code(dtype)
    HW = H * W
    x = q(rnd(N, Cc, H, W, seed=1), dtype)
    _, xv = to_nhwc(x, dtype)
    pooled = torch.empty(N, Cc, dtype=torch.float32, device=dev())
    L.call("dc_avgpool_fwd", dt, N, HW, Cc, vptr(xv), Cc, vptr(pooled), S())
    torch.cuda.synchronize()
    assert_close(pooled.float().cpu(), x.mean((2, 3)), dtype, bf16=1e-2)
    _, bv = empty_nhwc(N, H, W, Cc, dtype, ld=Cc + 8)
    L.call("dc_broadcast_hw", dt, N, HW, Cc, vptr(pooled), vptr(bv), Cc + 8, S())
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(bv), q(pooled.cpu(), dtype)[:, :, None, None].expand(N, Cc, H, W))
    summed = torch.empty(N, Cc, dtype=torch.float32, device=dev())
    L.call("dc_sum_hw", dt, N, HW, Cc, vptr(xv), Cc, vptr(summed), S())
    torch.cuda.synchronize()
    assert_close(summed.float().cpu(), x.sum((2, 3)), dtype, bf16=1e-2)
    acc0 = q(rnd(N, Cc, H, W, seed=2), dtype)
    _, av = to_nhwc(acc0, dtype)
    L.call("dc_avgpool_bwd_add", dt, N, HW, Cc, vptr(pooled), vptr(av), Cc, S())
    torch.cuda.synchronize()
    assert_close(from_nhwc(av), acc0 + pooled.float().cpu()[:, :, None, None] / HW, dtype, bf16=1e-2)
    _, cv = empty_nhwc(N, H, W, Cc, dtype, ld=Cc + 8, off=8)
    L.call("dc_copy_view", dt, N * HW, Cc, vptr(xv), Cc, vptr(cv), Cc + 8, S())
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(cv), x)
    # column sum (bias gradient)
    wsb = L.load().dc_colsum_workspace(N * HW, Cc)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    out = torch.empty(Cc, device=dev())
    L.call("dc_colsum", dt, N * HW, Cc, vptr(xv), Cc, vptr(out), vptr(ws), S())
    torch.cuda.synchronize()
    assert_close(out.cpu(), x.sum((0, 2, 3)), torch.float32, f32=1e-4)


@pytest.mark.parametrize("kind,wd", [("Adam", 1e-6), ("AdamW", 1e-2)])
def test_adam_matches_torch_optim(kind, wd):
    n = 100003
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.1)
    pr = torch.nn.Parameter(p0.clone())
    opt = (torch.optim.Adam if kind == "Adam" else torch.optim.AdamW)([pr], lr=1e-3, eps=1e-8, weight_decay=wd)
    npad = (n + 3) // 4 * 4
    p = torch.zeros(npad, device=dev()); p[:n] = p0.to(dev())
    gd = torch.zeros(npad, device=dev())
    m, v = torch.zeros(npad, device=dev()), torch.zeros(npad, device=dev())
    lr = torch.tensor([1e-3], device=dev())
    step = torch.zeros(1, dtype=torch.int32, device=dev())
    for s in range(1, 4):
        gs = g * s
        pr.grad = gs.clone()
        opt.step()
        gd[:n] = (gs * 2).to(dev())      # fed with grad_scale = 0.5
        step.fill_(s)
        L.call("dc_adam_step", L.DC_ADAM if kind == "Adam" else L.DC_ADAMW, n, vptr(p), vptr(gd), vptr(m), vptr(v), vptr(lr),
               0.9, 0.999, 1e-8, wd, vptr(step), 0.5, S())
        torch.cuda.synchronize()
        np.testing.assert_allclose(p[:n].cpu().numpy(), pr.detach().numpy(), rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("wd", [1e-2, 0.0], ids=["wd1e-2", "wd0_no_trust_ratio"])
def test_lamb_matches_oracle(wd):
    """wd == 0: apex FusedLAMB (use_nvlamb=False) skips the trust ratio.  The gradient arena is only read (header contract)."""
    from oracle.optim import OracleOptimizer
    sizes = [1000, 7, 4096, 33, 10001]   # one tensor of exactly one chunk, one of 2.4 chunks
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=3.0)
    p0[offs[1]:offs[2]] = 0.0        # a zero-norm tensor -> trust ratio 1
    cpu_p = [p0[offs[i]:offs[i + 1]].clone() for i in range(len(sizes))]
    ref = OracleOptimizer(cpu_p, "LAMB", lr=1e-2, eps=1e-6, weight_decay=wd)
    p = p0.clone().to(dev())
    m, v = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    lr = torch.tensor([1e-2], device=dev())
    step = torch.zeros(1, dtype=torch.int32, device=dev())
    ws = torch.full((L.load().dc_lamb_workspace_words(len(sizes), n),), float("nan"), device=dev())
    od = torch.from_numpy(offs).to(dev())
    for s in range(1, 4):
        gs = g / s
        ref.step([gs[offs[i]:offs[i + 1]] for i in range(len(sizes))])
        gd = gs.clone().to(dev())
        step.fill_(s)
        L.call("dc_lamb_step", len(sizes), vptr(od), n, vptr(p), vptr(gd), vptr(m), vptr(v), vptr(lr), 0.9, 0.999, 1e-6, wd, vptr(step),
               1.0, 1.0, vptr(ws), S())
        torch.cuda.synchronize()
        np.testing.assert_allclose(p.cpu().numpy(), torch.cat(cpu_p).numpy(), rtol=2e-5, atol=2e-6)
        assert torch.equal(gd.cpu(), gs), "dc_lamb_step must leave the gradient arena untouched"


def test_lamb_is_bit_reproducible():
    """No atomics in the norms: the same step from the same state gives the same bits (many chunks per tensor, odd sizes)."""
    sizes = [300000, 7, 4096, 1234567, 10001]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    p0, g = rnd(n, seed=1).to(dev()), rnd(n, seed=2, scale=3.0).to(dev())
    od = torch.from_numpy(offs).to(dev())
    lr = torch.tensor([1e-2], device=dev())
    step = torch.ones(1, dtype=torch.int32, device=dev())
    outs = []
    for _ in range(3):
        p, gd = p0.clone(), g.clone()
        m, v = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
        ws = torch.full((L.load().dc_lamb_workspace_words(len(sizes), n),), float("nan"), device=dev())
        L.call("dc_lamb_step", len(sizes), vptr(od), n, vptr(p), vptr(gd), vptr(m), vptr(v), vptr(lr), 0.9, 0.999, 1e-6, 1e-2, vptr(step),
               1.0, 1.0, vptr(ws), S())
        torch.cuda.synchronize()
        assert torch.isfinite(p).all()
        outs.append(p)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


PW_BN_BWD_CASES = [("b1_sep2", 128, 128, 2, 192, 192, 2), ("b1_sep1", 64, 128, 2, 192, 192, 2), ("ragged_nomask", 128, 128, 1, 257, 257, 0),
                   ("ragged64", 64, 128, 1, 257, 259, 2), ("b2_sep2", 256, 256, 2, 192, 192, 2), ("b2_sep1", 128, 256, 2, 192, 192, 2),
                   ("ragged256_nomask", 256, 256, 1, 257, 257, 0), ("ragged128_256", 128, 256, 1, 259, 257, 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,cin,cout,N,H,W,relu", PW_BN_BWD_CASES, ids=[c[0] for c in PW_BN_BWD_CASES])
def test_pointwise_bn_backward_in_one_pass(name, cin, cout, N, H, W, relu):
    """dc_pw_bn_bwd (BatchNorm backward apply + pointwise data gradient + pointwise weight gradient over ONE read of dout, y, x; dy never stored)
    against the three passes it replaces: dx bit for bit (same dy bits, same MFMA K order), the weight gradient against fp64 sums of the stored dy."""
    dtype, dt = torch.bfloat16, L.DC_BF16
    lib = L.load()
    M = N * H * W
    rows = lib.dc_pw_bn_bwd_rows(dt, cin, cout, M)
    assert rows > 0 and lib.dc_pw_bn_bwd_rows(dt, 96, cout, M) == 0 and lib.dc_pw_bn_bwd_rows(dt, cin, cout, 1000) == 0
    g = torch.Generator().manual_seed(5)
    ydev = (torch.randn(N, H, W, cout + 8, generator=g) * 1.5).to(dtype).to(dev())          # pad channels hold finite garbage
    dodev = (torch.randn(N, H, W, cout, generator=g) * 0.7).to(dtype).to(dev())
    xdev = torch.randn(N, H, W, cin + 16, generator=g).to(dtype).to(dev())
    yv, xv = ydev[..., :cout], xdev[..., 8:8 + cin]
    w = rnd(cout, cin, 1, 1, seed=6, scale=0.1).to(dev())
    d = L.ConvDesc(dt, 1, 1, 0, 1, 0, cin, cout)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf, wb = torch.zeros(nwf.value, dtype=dtype, device=dev()), torch.zeros(nwb.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w), vptr(wf), vptr(wb), S())
    gamma = (rnd(cout, seed=7).abs() + 0.4).to(dev())
    mean, invstd = rnd(cout, seed=8, scale=0.2).to(dev()), (torch.rand(cout, generator=g) + 0.5).to(dev())
    sc, sh = (gamma * invstd).contiguous(), rnd(cout, seed=9, scale=0.3).to(dev())
    dg, db = (rnd(cout, seed=10) * M * 0.01).to(dev()), (rnd(cout, seed=11) * M * 0.01).to(dev())
    # the three passes
    _, dyv = empty_nhwc(N, H, W, cout, dtype)
    L.call("dc_bn_bwd_apply", dt, M, cout, M, vptr(dodev), cout, vptr(yv), cout + 8, None, 0, relu, vptr(gamma), vptr(mean), vptr(invstd), vptr(dg),
           vptr(db), vptr(dyv), cout, None, 0, vptr(sc), vptr(sh), S())
    _, dx_ref = empty_nhwc(N, H, W, cin, dtype)
    L.call("dc_conv_dgrad", C.byref(d), N, H, W, vptr(dyv), cout, vptr(wb), vptr(dx_ref), cin, 0, S())
    torch.cuda.synchronize()
    gw_ref = (dyv.reshape(M, cout).double().t() @ xv.reshape(M, cin).double())
    # one pass
    dxb, dx = empty_nhwc(N, H, W, cin, dtype, ld=cin + 8)
    slab = torch.full((rows, cout, cin), float("nan"), device=dev())
    gw = torch.full((cout, cin, 1, 1), float("nan"), device=dev())
    L.call("dc_pw_bn_bwd", dt, M, cin, cout, M, vptr(dodev), cout, vptr(yv), cout + 8, relu, vptr(gamma), vptr(mean), vptr(invstd), vptr(dg), vptr(db),
           vptr(sc), vptr(sh), vptr(xv), cin + 16, vptr(wb), vptr(dx), cin + 8, vptr(slab), rows, S())
    ents = [L.FoldEntry(slab.data_ptr(), gw.data_ptr(), L.DC_FOLD_CONV, rows, 1, cout, cin)]
    L.call("dc_fold_slabs", (L.FoldEntry * 1)(*ents), 1, S())
    torch.cuda.synchronize()
    assert torch.isnan(dxb[..., cin:].float()).all()                          # pad channels untouched
    assert torch.equal(dx.float(), dx_ref.float()), name
    err = (gw.reshape(cout, cin).double() - gw_ref).abs().max().item()
    assert err <= 2e-5 * gw_ref.abs().max().item() + 1e-3, (name, err, gw_ref.abs().max().item())


SEPFWD_CASES = [("b1_sep2_lazy", 128, 2, 192, 192, 1, 1), ("b1_sep1_stored", 64, 2, 192, 192, 0, 0), ("affine_lazy64", 64, 1, 264, 256, 1, 0),
                ("one_row_of_tiles", 128, 1, 8, 8208, 0, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,cin,N,H,W,lazy,relu", SEPFWD_CASES, ids=[c[0] for c in SEPFWD_CASES])
def test_separable_conv_forward_as_one_operator(name, cin, N, H, W, lazy, relu):
    """dc_sepconv_fwd (depthwise 3x3 + pointwise 1x1 of the entry flow's thin layers in one kernel, the producer's BatchNorm applied on load)
    against dc_dwconv_fwd followed by dc_conv_fwd: the depthwise output and the pointwise output bit for bit, the BatchNorm sums after their
    finalize-style column sums to fp32 accuracy; rows the kernel does not own are zeros; pad channels of the outputs untouched."""
    dtype, dt, cout = torch.bfloat16, L.DC_BF16, 128
    lib = L.load()
    rows = lib.dc_sepconv_fwd_rows(dt, cin, cout, 1, 1, N, H, W)
    assert rows > 0 and lib.dc_sepconv_fwd_rows(dt, cin, cout, 2, 1, N, H, W) == 0 and lib.dc_sepconv_fwd_rows(dt, 96, cout, 1, 1, N, H, W) == 0
    assert lib.dc_sepconv_fwd_rows(dt, cin, cout, 1, 1, N, H + 4, W) == 0
    g = torch.Generator().manual_seed(3)
    xdev = (torch.randn(N, H, W, cin + 8, generator=g)).to(dtype).to(dev())
    xv = xdev[..., :cin]
    wm = rnd(cin, 1, 3, 3, seed=2, scale=1 / 3).to(dev())
    wd = torch.empty(9 * cin, device=dev())
    L.call("dc_dwconv_pack_weights", cin, vptr(wm), vptr(wd), S())
    w = rnd(cout, cin, 1, 1, seed=6, scale=cin ** -0.5).to(dev())
    d = L.ConvDesc(dt, 1, 1, 0, 1, 0, cin, cout)
    nwf, nwb = C.c_size_t(), C.c_size_t()
    L.call("dc_conv_packed_elems", C.byref(d), C.byref(nwf), C.byref(nwb))
    wf = torch.zeros(nwf.value, dtype=dtype, device=dev())
    L.call("dc_conv_pack_weights", C.byref(d), vptr(w), vptr(wf), None, S())
    sc = (rnd(cin, seed=7).abs() + 0.5).to(dev()) if lazy else None
    sh = rnd(cin, seed=8, scale=0.3).to(dev()) if lazy else None
    # two operators
    _, d_ref = empty_nhwc(N, H, W, cin, dtype)
    L.call("dc_dwconv_fwd", dt, cin, 1, 1, N, H, W, vptr(xv), cin + 8, vptr(wd), vptr(d_ref), cin, vptr(sc) if lazy else None, vptr(sh) if lazy else None,
           relu, S())
    _, y_ref = empty_nhwc(N, H, W, cout, dtype)
    rows_ref = lib.dc_conv_stat_rows(C.byref(d), N, H, W)
    slab_ref = torch.full((2, rows_ref, cout), float("nan"), device=dev())
    L.call("dc_conv_fwd", C.byref(d), N, H, W, vptr(d_ref), cin, vptr(wf), None, vptr(y_ref), cout, vptr(slab_ref), 0, S())
    # one operator
    db, dv = empty_nhwc(N, H, W, cin, dtype, ld=cin + 16)
    yb, yv = empty_nhwc(N, H, W, cout, dtype, ld=cout + 8)
    slab = torch.full((2, rows_ref, cout), float("nan"), device=dev())
    L.call("dc_sepconv_fwd", dt, cin, cout, N, H, W, vptr(xv), cin + 8, vptr(sc) if lazy else None, vptr(sh) if lazy else None, relu, vptr(wd),
           vptr(dv), cin + 16, vptr(wf), vptr(yv), cout + 8, vptr(slab), rows_ref, S())
    torch.cuda.synchronize()
    assert torch.equal(dv.float(), d_ref.float()), name
    assert torch.equal(yv.float(), y_ref.float()), name
    assert torch.isnan(db[..., cin:].float()).all() and torch.isnan(yb[..., cout:].float()).all()
    assert not torch.isnan(slab).any() and (slab[:, rows:] == 0).all()
    got, want = slab.double().sum(1), slab_ref.double().sum(1)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-3), (name, (got - want).abs().max().item())
    # eval mode: no statistics
    yv.fill_(0)
    L.call("dc_sepconv_fwd", dt, cin, cout, N, H, W, vptr(xv), cin + 8, vptr(sc) if lazy else None, vptr(sh) if lazy else None, relu, vptr(wd),
           vptr(dv), cin + 16, vptr(wf), vptr(yv), cout + 8, None, 0, S())
    torch.cuda.synchronize()
    assert torch.equal(yv.float(), y_ref.float())
