"""Whole-network parity of the HIP engine: against the CPU oracle on the same seeded inputs (small size) and against
the golden numbers captured from the reference itself (tests/golden/model_small.json, model_full.json)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mlperf_deepcam_amd import nn as dnn  # noqa: E402
from mlperf_deepcam_amd.engine import Engine  # noqa: E402
from oracle import loss_metric as olm  # noqa: E402  (checker only)
from oracle import model as omodel  # noqa: E402
from oracle import optim as ooptim  # noqa: E402
from util_inputs import make_inputs, sample_index  # noqa: E402

CW = olm.class_weights(-0.125)
DEV = torch.device("cuda", 0)


def _oracle_step(x, y):
    sd = omodel.init_state(333)
    keys = omodel.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    out = omodel.forward(sd, x, training=True)
    loss = olm.fp_loss(out, y, CW)
    loss.backward()
    return sd, keys, out.detach(), float(loss.detach())


def _rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def oracle_small():
    x, y = make_inputs(2, 64, 96)
    return (x, y) + _oracle_step(x, y)


# Gradient tolerances: with B=2 the 77 train-mode BatchNorms make the backward pass ill-conditioned -- against an fp64
# evaluation of the oracle, PyTorch's own CPU fp32 gradients are off by 1-3 % (relative L2, per tensor) at this size, and
# the fp32 HIP engine shows the same figures (scripts/debug_grads.py; DESIGN.md "numerics").  So fp32-vs-fp32 per-tensor
# agreement is bounded by ~2x that noise; the loss itself is well conditioned and is held to north_star's 1e-3.
@pytest.mark.parametrize("dtype,ltol,gtol", [(torch.float32, 2e-5, 8e-2), (torch.bfloat16, 5e-3, 10.0)], ids=["f32", "bf16"])
def test_forward_backward_vs_oracle(oracle_small, dtype, ltol, gtol):
    x, y, sd, keys, out_ref, loss_ref = oracle_small
    eng = Engine(2, 64, 96, dtype, seed=333)
    # identical initial weights, bit for bit
    for k in keys:
        assert torch.equal(eng.param_view(k).cpu(), sd[k].detach()), k
    logits = eng.forward(x.to(DEV), train=True)
    s = dnn.wce_fused(logits, y.to(DEV), CW, dlogits=eng.dlogits)
    eng.backward()
    torch.cuda.synchronize()
    loss = float(s.item()) / y.numel()
    assert loss == pytest.approx(loss_ref, rel=ltol)                     # north_star: 1e-3 relative
    lg = logits.cpu()
    if dtype == torch.float32:
        np.testing.assert_allclose(lg.numpy(), out_ref.numpy(), rtol=2e-3, atol=2e-3)
        # label argmax: identical except where the top-2 logits are closer than fp32 re-association noise
        a, b = olm.argmax_first(lg), olm.argmax_first(out_ref)
        top2 = out_ref.sort(1, descending=True)[0]
        margin = (top2[:, 0] - top2[:, 1]).numpy()
        assert np.all((a == b) | (margin < 5e-3))
    # (bf16: single logits are NOT comparable.  The randomly initialised 77-BatchNorm network amplifies a relative input
    #  perturbation of 1e-6 into a 3e-4 change of the logits and a 3 % change of the gradients even in fp32
    #  (scripts/sensitivity.py, profiles/sensitivity_r01.txt), so bf16's 4e-3 rounding decorrelates them; the loss, which
    #  north_star pins, is insensitive: 2e-7.)
    # gradients of every one of the 301 parameter tensors
    worst = ("", 0.0)
    for k in keys:
        g, r = eng.grad_view(k).cpu(), sd[k].grad
        e = _rel_l2(g, r)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < gtol, f"worst gradient {worst}"
    allg = torch.cat([eng.grad_view(k).cpu().flatten() for k in keys])
    allr = torch.cat([sd[k].grad.flatten() for k in keys])
    total = _rel_l2(allg, allr)
    print(f"[{dtype}] loss {loss:.7f} vs {loss_ref:.7f}; worst per-tensor grad err {worst}; whole-arena grad err {total:.3e}")
    assert total < (3e-2 if dtype == torch.float32 else 2.0)
    assert torch.isfinite(allg).all()
    # BatchNorm running statistics were updated exactly once
    for k in ("xception_features.bn1", "xception_features.block4.rep.2", "global_avg_pool.2", "upsample.deconv3.1"):
        np.testing.assert_allclose(eng.buffer_view(k + ".running_mean").cpu().numpy(), sd[k + ".running_mean"].numpy(),
                                   rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=6e-3 if dtype == torch.bfloat16 else 1e-5)
        assert int(eng.buffer_view(k + ".num_batches_tracked")) == 1


def test_golden_small_three_adam_steps(golden_dir):
    """fp32 engine vs numbers produced by the reference model + torch.optim.Adam (64x96, B=2)."""
    g = json.load(open(os.path.join(golden_dir, "model_small.json")))
    x, y = make_inputs(2, g["H"], g["W"])
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.float32, seed=333)
    net.materialize(2, g["H"], g["W"])
    opt = dnn.make_optimizer("Adam", net, 1e-3, 1e-8, 1e-6)
    step = dnn.TrainStep(net, opt, CW, 2, g["H"], g["W"], with_metrics=True)
    xd, yd = x.to(DEV), y.to(DEV)
    ref = g["adam_wd1e-6"]["steps"]
    idx = np.array(g["adam_wd1e-6"]["sample_index"])
    for s in range(3):
        step(xd, yd)
        torch.cuda.synchronize()
        # Step 0 is a pure function of identical weights: tight.  Later steps follow Adam updates (step 1 = lr*sign(g)) of a
        # network that amplifies 1e-6 perturbations 300x per forward pass (profiles/sensitivity_r01.txt): two fp32
        # implementations drift apart at the 1e-3 .. 1e-2 level by step 2 at this tiny size (the CPU oracle itself differs from
        # the reference by 5e-4 there, tests/test_oracle_golden.py).
        assert step.loss() == pytest.approx(ref[s]["loss"], rel=(2e-5, 2e-3, 1e-2)[s]), f"step {s}"
        assert step.iou() == pytest.approx(ref[s]["iou"], rel=(2e-3, 2e-2, 4e-2)[s]), f"step {s}"
        if s == 0:
            samples = step.eng.logits.flatten()[torch.from_numpy(idx).to(DEV)].cpu().numpy()
            np.testing.assert_allclose(samples, np.array(ref[0]["logit_samples"]), rtol=2e-3, atol=2e-3)
            for k, d in ref[0]["grad_digest"].items():
                assert float(step.eng.grad_view(k).double().abs().sum()) == pytest.approx(d["abs"], rel=5e-3), k


def test_golden_small_bf16_loss_within_north_star(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "model_small.json")))
    x, y = make_inputs(2, g["H"], g["W"])
    eng = Engine(2, g["H"], g["W"], torch.bfloat16, seed=333)
    logits = eng.forward(x.to(DEV), train=True)
    s = dnn.wce_fused(logits, y.to(DEV), CW)
    torch.cuda.synchronize()
    # 64x96 is 144x fewer pixels than the benchmark size: the deepest BatchNorms see 48 values per channel and bf16
    # rounding of their inputs does not average out; north_star's 1e-3 is asserted at 768x1152 below, 5e-3 here.
    got = float(s.item()) / y.numel()
    print(f"[bf16 64x96] loss {got:.7f} vs reference {g['adam_wd1e-6']['steps'][0]['loss']:.7f}")
    assert got == pytest.approx(g["adam_wd1e-6"]["steps"][0]["loss"], rel=5e-3)


def test_eval_mode_and_batch1_rule(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "model_small.json")))
    x, y = make_inputs(1, g["H"], g["W"], seed=g["eval_b1"]["seed"])
    eng = Engine(1, g["H"], g["W"], torch.float32, seed=333)
    out = eng.forward(x.to(DEV), train=False)
    torch.cuda.synchronize()
    idx = sample_index(out.numel())
    np.testing.assert_allclose(out.flatten().cpu().numpy()[idx], np.array(g["eval_b1"]["logit_samples"]), rtol=2e-3, atol=2e-4)
    loss = float(dnn.wce_fused(out, y.to(DEV), CW).item()) / y.numel()
    assert loss == pytest.approx(g["eval_b1"]["loss"], rel=1e-4)
    pred = torch.max(out, 1)[1]
    assert float(dnn.compute_score(pred, y.to(DEV), num_classes=3, device_id=0)) == pytest.approx(g["eval_b1"]["iou"], rel=1e-3)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):       # reference behaviour
        eng.forward(x.to(DEV), train=True)


def test_module_surface_runs_the_reference_loop():
    """The reference's own step sequence (train_hdf5_ddp.py:352-364), through autograd, equals the fused TrainStep."""
    x, y = make_inputs(2, 32, 48)
    xd, yd = x.to(DEV), y.to(DEV)
    res = []
    for fused in (False, True):
        torch.manual_seed(333)
        net = dnn.DeepLabv3_plus(n_input=16, n_classes=3, os=16, pretrained=False, rank=1, dtype=torch.float32)
        net.to(DEV)
        net.materialize(2, 32, 48)
        net.train()
        opt = dnn.make_optimizer("AdamW", net, 1e-3, 1e-8, 1e-2)
        sched = dnn.get_lr_schedule(1e-3, {"type": "multistep", "milestones": "2 4", "decay_rate": "0.1"}, opt, last_step=0)
        losses = []
        step = dnn.TrainStep(net, opt, CW, 2, 32, 48) if fused else None
        for _ in range(3):
            if fused:
                step(xd, yd)
                losses.append(step.loss())
            else:
                outputs = net.forward(xd)
                loss = dnn.fp_loss(outputs, yd, weight=CW, fpw_1=2.6, fpw_2=1.7)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss.item()))
            sched.step()
        torch.cuda.synchronize()
        res.append((losses, net.engine.params.clone()))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-6)
    assert torch.equal(res[0][1], res[1][1])
    sd = net.state_dict()
    assert len(sd) == 532 and sd["xception_features.bn1.num_batches_tracked"].dtype == torch.int64


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-3)], ids=["f32", "bf16"])
def test_golden_full_size_step0(golden_dir, dtype, tol):
    """768x1152, B=2 (BASELINE configs[1]/[2] shapes): loss and IoU of the first step against the reference's own run."""
    path = os.path.join(golden_dir, "model_full.json")
    g = json.load(open(path))
    x, y = make_inputs(2, 768, 1152)
    eng = Engine(2, 768, 1152, dtype, seed=333)
    logits = eng.forward(x.to(DEV), train=True)
    counts = torch.zeros(9, dtype=torch.int64, device=DEV)
    s = dnn.wce_fused(logits, y.to(DEV), CW, dlogits=eng.dlogits, counts=counts)
    eng.backward()
    torch.cuda.synchronize()
    ref = g["adam_wd1e-6"]["steps"][0]
    got = float(s.item()) / y.numel()
    print(f"[{dtype} 768x1152] loss {got:.7f} vs reference {ref['loss']:.7f} (rel {abs(got - ref['loss']) / ref['loss']:.2e}); "
          f"iou {dnn.iou_from_counts(counts.cpu().tolist()):.6f} vs {ref['iou']:.6f}")
    assert got == pytest.approx(ref["loss"], rel=tol)
    assert dnn.iou_from_counts(counts.cpu().tolist()) == pytest.approx(ref["iou"], rel=1e-3 if dtype == torch.float32 else 5e-3)
    if dtype == torch.float32:
        idx = torch.tensor(g["adam_wd1e-6"]["sample_index"], device=DEV)
        np.testing.assert_allclose(logits.flatten()[idx].cpu().numpy(), np.array(ref["logit_samples"]), rtol=2e-3, atol=2e-3)
    for k, d in ref["grad_digest"].items():
        got = float(eng.grad_view(k).double().abs().sum())
        assert got == pytest.approx(d["abs"], rel=5e-3 if dtype == torch.float32 else 0.5), k
    assert torch.isfinite(eng.grads).all()


@pytest.mark.parametrize("optname", ["LAMB", "AdamW"])
def test_train_step_is_bit_reproducible(optname):
    """Two independently built models, the same batch, three steps (weight gradients on the side stream): no kernel that feeds
    the update uses an order-dependent reduction, so parameters, BatchNorm buffers and IoU agree bit for bit.  (The network amplifies a 1e-7
    perturbation into 1e-3 of the loss within ten steps, so anything less would show up as diverging loss curves.)"""
    x, y = make_inputs(4, 96, 160)
    dev = torch.device("cuda", 0)

    def run():
        net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
        net.materialize(4, 96, 160)
        net.train()
        opt = dnn.make_optimizer(optname, net, 1e-3, 1e-8, 1e-2)
        step = dnn.TrainStep(net, opt, olm.class_weights(-0.125), 4, 96, 160, with_metrics=True)
        out = []
        for _ in range(3):
            step(x.to(dev), y.to(dev))
            torch.cuda.synchronize()
            out.append((step.loss(), step.iou(), net.engine.params.clone(), net.engine.buffers.clone()))
        return out

    a, b = run(), run()
    for (la, ia, pa, ba), (lb, ib, pb, bb) in zip(a, b):
        assert abs(la - lb) <= 1e-12 * abs(la) and ia == ib      # the reported loss is a double-precision atomic sum (1e-16 jitter)
        assert torch.equal(pa, pb) and torch.equal(ba, bb)


def test_steps_enqueued_ahead_match_synchronised_steps():
    """The host may enqueue steps far ahead of the GPU (bench.py does): the optimizer's device scalars (lr, step count) must be
    the ones of THEIR step.  Six steps enqueued back to back == six steps with a device synchronisation after each."""
    B, H, W = 4, 192, 288
    x, y = make_inputs(B, H, W)
    dev = torch.device("cuda", 0)
    xd, yd = x.to(dev), y.to(dev)

    def run(sync_each):
        net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
        net.materialize(B, H, W)
        net.train()
        opt = dnn.make_optimizer("LAMB", net, 1e-3, 1e-8, 1e-2)
        step = dnn.TrainStep(net, opt, olm.class_weights(-0.125), B, H, W)
        for _ in range(6):
            step(xd, yd)
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return net.engine.params.clone(), opt.m.clone(), opt.v.clone()

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)
