"""Fused weighted-CE / argmax / IoU kernel against the oracle and the golden vectors captured from the reference."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mlperf_deepcam_amd import lib as L  # noqa: E402
from oracle import loss_metric as olm  # noqa: E402  (checker only)

CLASS_W = olm.class_weights(-0.125)


def _run(logit, target, want_grad=True):
    dev = torch.device("cuda", 0)
    B, _, H, W = logit.shape
    lg = logit.contiguous().to(dev)
    tg = target.contiguous().to(dev)
    cw = torch.tensor(CLASS_W, dtype=torch.float32, device=dev)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    grad = torch.full_like(lg, float("nan")) if want_grad else None
    pred = torch.full((B, H, W), -1, dtype=torch.int64, device=dev)
    counts = torch.zeros(9, dtype=torch.int64, device=dev)
    n = B * H * W
    L.call("dc_wce_fused", B, H, W, C.c_void_p(lg.data_ptr()), C.c_void_p(tg.data_ptr()), tg.element_size(),
           C.c_void_p(cw.data_ptr()), 1.0 / n, C.c_void_p(loss.data_ptr()), L.dptr(grad), C.c_void_p(pred.data_ptr()),
           C.c_void_p(counts.data_ptr()), L.stream_ptr())
    torch.cuda.synchronize()
    return float(loss) / n, (grad.cpu() if want_grad else None), pred.cpu().numpy(), counts.cpu().numpy()


@pytest.mark.parametrize("case", ["rand", "absent", "ties", "flat"])
def test_loss_kernel_golden(golden_dir, case):
    z = np.load(os.path.join(golden_dir, "loss_kat.npz"))
    logit = torch.from_numpy(z[case + "_logit"])
    target = torch.from_numpy(z[case + "_target"])          # uint8 / int32 / int64 depending on the case
    loss, grad, pred, counts = _run(logit, target)
    assert loss == pytest.approx(float(z[case + "_loss"]), rel=1e-5)       # fp32 tolerance of north_star is 1e-3
    np.testing.assert_allclose(grad.numpy(), z[case + "_grad"], rtol=2e-4, atol=1e-8)
    np.testing.assert_array_equal(pred, z[case + "_pred"])                 # bit exact, incl. first-index tie-break
    np.testing.assert_array_equal(counts[0:3], z[case + "_tp"])
    np.testing.assert_array_equal(counts[3:6], z[case + "_fp"])
    np.testing.assert_array_equal(counts[6:9], z[case + "_fn"])
    assert olm.iou_from_counts(counts[0:3], counts[3:6], counts[6:9]) == pytest.approx(float(z[case + "_iou"]), rel=1e-6)


def test_loss_kernel_full_size_properties():
    """768x1152, B=2: compare with the oracle on the same seeded logits; counts must partition the pixels."""
    g = torch.Generator().manual_seed(5)
    logit = torch.randn(2, 3, 768, 1152, generator=g) * 2
    target = torch.randint(0, 3, (2, 768, 1152), generator=g)
    loss, grad, pred, counts = _run(logit, target)
    assert loss == pytest.approx(float(olm.fp_loss(logit, target, CLASS_W)), rel=1e-5)
    np.testing.assert_array_equal(pred, olm.argmax_first(logit))
    tp, fp, fn = olm.confusion_counts(pred, target.numpy())
    np.testing.assert_array_equal(counts, np.concatenate([tp, fp, fn]))
    n = target.numel()
    assert counts[0:3].sum() + counts[3:6].sum() == n and counts[3:6].sum() == counts[6:9].sum()
    # gradient of a mean of per-pixel terms: every pixel's three class gradients sum to zero
    assert float(grad.sum(1).abs().max()) < 1e-9
    ref = olm.fp_loss_grad(logit, target, CLASS_W)
    np.testing.assert_allclose(grad.numpy(), ref.numpy(), rtol=2e-4, atol=1e-12)


def test_confusion_counts_entry_point():
    g = torch.Generator().manual_seed(6)
    pred = torch.randint(0, 3, (3, 40, 50), generator=g)
    gt = torch.randint(0, 3, (3, 40, 50), generator=g).to(torch.int32)
    dev = torch.device("cuda", 0)
    counts = torch.zeros(9, dtype=torch.int64, device=dev)
    pd, gd = pred.to(dev), gt.to(dev)
    L.call("dc_confusion_counts", pred.numel(), C.c_void_p(pd.data_ptr()), C.c_void_p(gd.data_ptr()), 4,
           C.c_void_p(counts.data_ptr()), L.stream_ptr())
    torch.cuda.synchronize()
    tp, fp, fn = olm.confusion_counts(pred.numpy(), gt.numpy())
    np.testing.assert_array_equal(counts.cpu().numpy(), np.concatenate([tp, fp, fn]))


@pytest.mark.parametrize("bad,dtype", [(3, torch.int64), (-1, torch.int64), (2 ** 32, torch.int64), (255, torch.uint8), (7, torch.int32)])
def test_out_of_range_label_poisons_the_loss(bad, dtype):
    """The reference's nn.CrossEntropyLoss raises for a label outside [0,3) (utils/losses.py:36).  A kernel cannot raise and the
    ABI never synchronises: the loss and that pixel's gradient come back NaN instead of the pixel being dropped silently."""
    g = torch.Generator().manual_seed(3)
    logit = torch.randn(2, 3, 16, 24, generator=g)
    target = torch.randint(0, 3, (2, 16, 24), generator=g).to(dtype)
    loss, grad, _, _ = _run(logit, target)
    assert np.isfinite(loss) and torch.isfinite(grad).all()
    target[1, 5, 7] = bad
    loss, grad, _, _ = _run(logit, target)
    assert np.isnan(loss)
    assert torch.isnan(grad[1, :, 5, 7]).all() and int(torch.isnan(grad).sum()) == 3
