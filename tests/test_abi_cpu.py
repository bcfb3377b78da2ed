"""CPU-side checks of the boundary: the C-ABI library builds, loads and exports every symbol of include/deepcam_hip.h."""
import os
import re

import pytest

from mlperf_deepcam_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "deepcam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_loads():
    if not os.path.exists(L.LIB_PATH):
        L.build()
    lib = L.load()
    assert lib.dc_version() >= 1


def test_every_declared_symbol_is_exported_and_bound():
    declared = _header_functions()
    assert len(declared) >= 40
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in deepcam_hip.h but not exported"
    assert sorted(L.EXPORTS) == declared, "ctypes table and header disagree"


def test_descriptor_struct_matches_header_layout():
    import ctypes as C
    assert C.sizeof(L.ConvDesc) == 8 * 4
    assert [f[0] for f in L.ConvDesc._fields_] == ["dtype", "k", "stride", "pad", "dil", "transposed", "cin", "cout"]


def test_host_side_geometry_without_gpu():
    """dc_conv_out_hw / workspace queries are pure host code: usable without a device."""
    import ctypes as C
    lib = L.load()
    ho, wo = C.c_int(), C.c_int()
    d = L.ConvDesc(L.DC_BF16, 3, 2, 1, 1, 0, 16, 32)
    assert lib.dc_conv_out_hw(C.byref(d), 768, 1152, C.byref(ho), C.byref(wo)) == 0 and (ho.value, wo.value) == (384, 576)
    d = L.ConvDesc(L.DC_BF16, 3, 1, 18, 18, 0, 2048, 256)
    assert lib.dc_conv_out_hw(C.byref(d), 48, 72, C.byref(ho), C.byref(wo)) == 0 and (ho.value, wo.value) == (48, 72)
    d = L.ConvDesc(L.DC_BF16, 3, 2, 1, 1, 1, 256, 3)
    assert lib.dc_conv_out_hw(C.byref(d), 384, 576, C.byref(ho), C.byref(wo)) == 0 and (ho.value, wo.value) == (768, 1152)
    d = L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, 728, 728)
    assert lib.dc_conv_wgrad_workspace(C.byref(d), 2, 48, 72) >= 728 * 728 * 4
    assert lib.dc_conv_stat_rows(C.byref(d), 2, 48, 72) == 54
    # the compact slab of the 224-pixel tiles (a host-side plan, no device needed): one row per tile where igemm224.hip serves the layer --
    # 224 x 384 tiles at local batch 8, 224 x 192 at 4 --, the 128-pixel rows where another kernel does (local batch 2: 128 x 192 tiles)
    assert lib.dc_conv_stat_rows_kn(C.byref(d), 8, 48, 72) == 124 and lib.dc_conv_stat_rows(C.byref(d), 8, 48, 72) == 216
    assert lib.dc_conv_stat_rows_kn(C.byref(d), 4, 48, 72) == 62
    assert lib.dc_conv_stat_rows_kn(C.byref(d), 2, 48, 72) == 54
    d32 = L.ConvDesc(L.DC_F32, 1, 1, 0, 1, 0, 728, 728)
    assert lib.dc_conv_stat_rows_kn(C.byref(d32), 8, 48, 72) == 216
    # where a launch adds its BatchNorm sums to ONE fp64 row (a host-side plan as well): the pointwise tile kernels igemm224.hip / igemm192.hip
    # (the middle flow at every local batch), not fp32, not a layer another kernel serves (2048 -> 256 fills a third of a 384-wide tile), not the
    # stem; the depthwise data gradient: where the persistent kernel runs (stride 1, at least 64 channels, extents multiples of 8)
    assert [lib.dc_conv_sum_row_kn(C.byref(d), n, 48, 72) for n in (8, 4, 2)] == [1, 1, 1]
    assert lib.dc_conv_sum_row_kn(C.byref(d32), 8, 48, 72) == 0
    assert lib.dc_conv_sum_row_kn(C.byref(L.ConvDesc(L.DC_BF16, 1, 1, 0, 1, 0, 2048, 256)), 8, 48, 72) == 0
    assert lib.dc_conv_sum_row_kn(C.byref(L.ConvDesc(L.DC_BF16, 3, 2, 1, 1, 0, 16, 32)), 8, 768, 1152) == 0
    assert lib.dc_dwconv_dgrad_sum_row_ok(L.DC_BF16, 728, 1, 1, 8, 48, 72) == 1 and lib.dc_dwconv_dgrad_sum_row_ok(L.DC_BF16, 1536, 1, 2, 2, 48, 72) == 1
    assert lib.dc_dwconv_dgrad_sum_row_ok(L.DC_BF16, 728, 2, 1, 8, 96, 144) == 0 and lib.dc_dwconv_dgrad_sum_row_ok(L.DC_F32, 728, 1, 1, 8, 48, 72) == 0
    assert lib.dc_dwconv_dgrad_sum_row_ok(L.DC_BF16, 32, 1, 1, 8, 48, 72) == 0 and lib.dc_dwconv_dgrad_sum_row_ok(L.DC_BF16, 728, 1, 1, 8, 50, 72) == 0
    # a failed call leaves a message behind
    d = L.ConvDesc(L.DC_BF16, 1, 2, 0, 1, 0, 64, 64)
    assert lib.dc_conv_stat_rows(C.byref(d), 1, 5, 5) == 1
    with pytest.raises(L.DeepcamHipError):
        L.call("dc_conv_dgrad", C.byref(d), 1, 5, 5, None, 64, None, None, 64, 0, None)


def test_torch_custom_ops_are_registered():
    """north_star: the host calls the HIP path "through custom ops": nn.py reaches the step's operators as torch.ops.deepcam.*
    (mlperf-deepcam_amd/ops.py), each of them a dispatcher entry over the C ABI."""
    import torch
    from mlperf_deepcam_amd import ops  # noqa: F401
    for name in ("net_forward", "net_backward", "wce_fused", "confusion_counts", "optimizer_step"):
        op = getattr(torch.ops.deepcam, name)
        assert "deepcam::" + name in str(op.default._schema)
    with pytest.raises(L.DeepcamHipError):
        torch.ops.deepcam.optimizer_step(12345)              # an unknown handle fails loudly
    with pytest.raises(L.DeepcamHipError):
        torch.ops.deepcam.net_forward(torch.zeros(1, 16, 16, 16), 999, False)


def test_option_defaults_table_is_accepted():
    """dc_reset_options applies the library's table of switch defaults through dc_set_option itself: every name in it must be a switch the
    library knows (no GPU needed: the switches are host-side state)."""
    from mlperf_deepcam_amd import lib as L
    assert L.load().dc_reset_options() == 0, L.last_error()
    assert L.load().dc_set_option(b"no_such_switch", 1) != 0
