"""ONE Xception Block of the HIP engine against the reference's own Block (architecture/deeplab_xception.py:69-122): forward,
input gradient, every parameter gradient and the BatchNorm running statistics, for the five rep-list shapes the network uses
(block1 / block2 / block3 / middle flow / block20).  The known-answer vectors (tests/golden/block_kat.npz) were produced by the
reference's Block class at small widths; the engine side is built from the PRODUCT's op builders (Engine._xblock, the very code
the whole network uses) over a parameter layout of just that block.

What this pins that the whole-model tests only imply: the in-place ReLU quirk (the shortcut sees relu(inp)), the
rep-list order of grow_first / is_last / stride-2 blocks, the BN-less tail separable conv, and the residual add before the next
block's ReLU -- per tensor, in fp32, to 1e-4."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mlperf_deepcam_amd import spec as S  # noqa: E402
from mlperf_deepcam_amd.engine import Act, Engine  # noqa: E402

DEV = torch.device("cuda", 0)

# name -> (kwargs of the reference Block as tests/golden/make_golden.py::g3_blocks built it, spec._block arguments)
CASES = {
    "b1": dict(cin=8, cout=16, reps=2, stride=2, start_with_relu=False, grow_first=True, is_last=False),
    "b2": dict(cin=16, cout=24, reps=2, stride=2, start_with_relu=True, grow_first=True, is_last=False),
    "b3": dict(cin=16, cout=24, reps=2, stride=2, start_with_relu=True, grow_first=True, is_last=True),
    "mid": dict(cin=24, cout=24, reps=3, stride=1, start_with_relu=True, grow_first=True, is_last=False),
    "b20": dict(cin=24, cout=32, reps=2, stride=1, start_with_relu=True, grow_first=False, is_last=True),
}


def _nhwc(t_nchw):
    return t_nchw.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("case", list(CASES))
def test_block_matches_reference_block(golden_dir, case):
    z = np.load(os.path.join(golden_dir, "block_kat.npz"))
    kw = CASES[case]
    blk = S._block("blk", kw["cin"], kw["cout"], kw["reps"], stride=kw["stride"], start_with_relu=kw["start_with_relu"],
                   grow_first=kw["grow_first"], is_last=kw["is_last"])
    lay = S.Layout(table=S.block_param_table(blk))
    x = torch.from_numpy(z[case + "_x"])                    # [2, cin, 12, 16]
    N, _, H, W = x.shape
    made = {}

    def builder(eng):
        xin = Act(eng, "xin", N, H, W, kw["cin"])
        made["xin"] = xin
        made["out"] = eng._xblock(blk, xin, relu_out=False)     # the block's own output, before the NEXT block's in-place ReLU

    eng = Engine(N, 16, 16, torch.float32, layout=lay, builder=builder)
    prefix = "xception_features.blk."
    # the reference block's parameters and BatchNorm buffers
    for name, p in lay.params.items():
        v = torch.from_numpy(z[f"{case}_sd_{name[len(prefix):]}"]).reshape(-1)
        eng.params[p.offset:p.offset + v.numel()].copy_(v)
    eng.mark_weights_changed()
    # the engine's convention: a block input is already ReLU'd (the reference's leading in-place ReLU mutates it, and the
    # shortcut reads the mutated tensor).  block1 has no leading ReLU and sees the raw tensor.
    xin_engine = torch.relu(x) if kw["start_with_relu"] else x
    if kw["start_with_relu"]:
        np.testing.assert_array_equal(z[case + "_xin_after"], xin_engine.numpy())       # the quirk, as captured from the reference
    made["xin"].buf.copy_(_nhwc(xin_engine))
    eng.pack_weights()
    for op in eng.fwd_train:
        op()
    torch.cuda.synchronize()
    out = made["out"]
    y = out.view().permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(y.numpy(), z[case + "_y"], rtol=1e-4, atol=1e-4)

    # backward from the reference's upstream gradient
    out.grad.buf.zero_()
    out.grad.view().copy_(_nhwc(torch.from_numpy(z[case + "_go"])).to(DEV))
    eng.backward()                  # the launch list, the folds of the weight-gradient slabs and the join with the weight-gradient stream
    torch.cuda.synchronize()
    gx = made["xin"].grad.view().permute(0, 3, 1, 2).cpu()
    if kw["start_with_relu"]:
        gx = gx * (x > 0)                                    # d/dx of the leading ReLU the engine's caller owns
    scale = float(np.abs(z[case + "_gx"]).max())
    np.testing.assert_allclose(gx.numpy(), z[case + "_gx"], rtol=1e-3, atol=2e-4 * scale)
    for name, p in lay.params.items():
        ref = z[f"{case}_grad_{name[len(prefix):]}"]
        got = eng.grads[p.offset:p.offset + math.prod(p.shape)].view(p.shape).cpu().numpy()
        s = float(np.abs(ref).max()) + 1e-12
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=3e-4 * s, err_msg=name)
    # BatchNorm running statistics after the one training-mode forward: what the reference's state_dict holds
    for name, (off, c) in lay.buffers.items():
        ref = z[f"{case}_sd_{name[len(prefix):]}"]
        np.testing.assert_allclose(eng.buffers[off:off + c].cpu().numpy(), ref, rtol=1e-4, atol=1e-5, err_msg=name)
    for name, idx in lay.nbt.items():
        assert int(eng.nbt[idx]) == int(z[f"{case}_sd_{name[len(prefix):]}"]) == 1
