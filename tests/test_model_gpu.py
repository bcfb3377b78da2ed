"""Whole-network parity of the HIP engine: against the CPU oracle on the same seeded inputs (small size) and against
the golden numbers captured from the reference itself (tests/golden/model_small.json, model_full.json)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mlperf_deepcam_amd import nn as dnn  # noqa: E402
from mlperf_deepcam_amd import lib as L  # noqa: E402
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


# Gradient tolerances.  With B=2 the 77 train-mode BatchNorms make the backward pass ill-conditioned: against an fp64 evaluation
# of the oracle, PyTorch's own CPU fp32 gradients are off by 1-3 % (relative L2, per tensor) at this size and the fp32 HIP engine
# shows the same figures (scripts/debug_grads.py), so fp32-vs-fp32 per-tensor agreement is bounded by ~2x that noise.
# bf16 gradients are NOT compared with the oracle's: the measurement behind that decision is scripts/grad_check.py
# (profiles/r02_grad_check.txt) -- at 768x1152 the fp32 engine's gradient agrees with central differences of its own loss, while
# the bf16 gradient's projection on the fp32 encoder gradient is 0.16 (cosine over the whole arena 0.56, decoder 0.995): the
# randomly initialised network amplifies bf16's 4e-3 rounding of the activations beyond the radius in which the gradient is
# constant.  What IS checked for bf16: the loss (north_star's quantity); backward kernel by kernel (test_kernels_gpu.py, incl. the
# full layer shapes); the whole backward program against the fp32 engine at ONE shared linearisation point
# (test_backward_parity_at_shared_activations); and directional derivatives where the problem is well conditioned
# (test_directional_derivatives_full_size).
@pytest.mark.parametrize("dtype,ltol", [(torch.float32, 2e-5), (torch.bfloat16, 5e-3)], ids=["f32", "bf16"])
def test_forward_backward_vs_oracle(oracle_small, dtype, ltol):
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
    allg = torch.cat([eng.grad_view(k).cpu().flatten() for k in keys])
    allr = torch.cat([sd[k].grad.flatten() for k in keys])
    assert torch.isfinite(allg).all()
    if dtype == torch.float32:
        np.testing.assert_allclose(lg.numpy(), out_ref.numpy(), rtol=2e-3, atol=2e-3)
        # label argmax: identical except where the top-2 logits are closer than fp32 re-association noise
        a, b = olm.argmax_first(lg), olm.argmax_first(out_ref)
        top2 = out_ref.sort(1, descending=True)[0]
        margin = (top2[:, 0] - top2[:, 1]).numpy()
        assert np.all((a == b) | (margin < 5e-3))
        # gradients of every one of the 301 parameter tensors
        worst = ("", 0.0)
        for k in keys:
            e = _rel_l2(eng.grad_view(k).cpu(), sd[k].grad)
            if e > worst[1]:
                worst = (k, e)
        total = _rel_l2(allg, allr)
        print(f"[fp32] loss {loss:.7f} vs {loss_ref:.7f}; worst per-tensor grad err {worst}; whole-arena grad err {total:.3e}")
        assert worst[1] < 8e-2, f"worst gradient {worst}"
        assert total < 3e-2
    else:
        # a descent direction of the right magnitude, no more (measured at this size: cosine 0.1786, norm ratio 0.8462; the run is
        # deterministic, so the bounds sit just outside the measured values: ADVICE r02)
        cos = float((allg.double() @ allr.double()) / (allg.double().norm() * allr.double().norm()))
        ratio = float(allg.double().norm() / allr.double().norm())
        print(f"[bf16] loss {loss:.7f} vs {loss_ref:.7f}; gradient cosine with the oracle {cos:.4f}, norm ratio {ratio:.4f}")
        assert cos > 0.15 and 0.78 < ratio < 0.95
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
        # abs-sums: fp32 measured within 2.3e-3 of the reference, bf16 within 0.11 (profiles/r02_grad_check.txt); an all-zero or
        # mis-scaled tensor fails either bound (a sign flip does not: see test_backward_parity_at_shared_activations for that)
        assert got == pytest.approx(d["abs"], rel=5e-3 if dtype == torch.float32 else 0.2), k
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


def test_hip_graph_replays_the_eager_steps():
    """TrainStep.enable_graph() (bench.py --graph; slower than eager launches since the weight gradients moved to a second stream, kept as an
    option): the captured step -- both streams, the library's stream fences and memsets included -- replays the eager steps bit for bit."""
    dev = torch.device("cuda", 0)
    B, H, W = 4, 96, 160
    batches = [make_inputs(B, H, W, seed=s) for s in (5, 6, 7)]
    z = (torch.zeros(B, 16, H, W, device=dev), torch.zeros(B, H, W, dtype=torch.int64, device=dev))

    def run(graph):
        net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
        net.materialize(B, H, W)
        net.train()
        opt = dnn.make_optimizer("LAMB", net, 1e-3, 1e-8, 1e-2)
        step = dnn.TrainStep(net, opt, olm.class_weights(-0.125), B, H, W, with_metrics=True)
        step(*z)
        if graph:
            net.engine.x_static.zero_()
            step.enable_graph()             # its warm-up is one more launch with the scalars of the step before; the capture pass executes nothing
        else:
            step.launch(*z)
        out = []
        for x, y in batches:
            step(x.to(dev), y.to(dev))
            torch.cuda.synchronize()
            out.append((step.loss(), step.iou(), net.engine.params.clone(), net.engine.buffers.clone()))
        return out

    a, b = run(False), run(True)
    for (la, ia, pa, ba), (lb, ib, pb, bb) in zip(a, b):
        assert abs(la - lb) <= 1e-12 * abs(la) and ia == ib
        assert torch.equal(pa, pb) and torch.equal(ba, bb)


@pytest.mark.parametrize("optname", ["LAMB", "AdamW"])
def test_recorded_launch_list_replays_the_eager_steps(optname):
    """TrainStep.enable_program(): the step recorded once as a C-side launch list (lib.Program / dc_program_*: ~700 library calls with their
    arguments, stream fences and the optimizer included) and replayed by ONE call per step, against the same steps issued call by call
    from Python: parameters, BatchNorm buffers, loss and IoU bit for bit, with different batches from step to step (the batch enters through
    the static buffers) and the lr changing between steps (it enters through the optimizer's device scalars)."""
    dev = torch.device("cuda", 0)
    B, H, W = 4, 96, 160
    batches = [make_inputs(B, H, W, seed=s) for s in (1, 2, 3, 4)]

    def run(program):
        net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
        net.materialize(B, H, W)
        net.train()
        opt = dnn.make_optimizer(optname, net, 1e-3, 1e-8, 1e-2)
        step = dnn.TrainStep(net, opt, olm.class_weights(-0.125), B, H, W, with_metrics=True)
        # the recording (and its warm-up) is two real steps on the static buffers: give the eager run the same two steps
        z = (torch.zeros(B, 16, H, W, device=dev), torch.zeros(B, H, W, dtype=torch.int64, device=dev))
        if program:
            net.engine.x_static.zero_()
            step.enable_program()
            names = step._program.names()
            assert len(names) > 300 and "dc_stream_fence" in names and "dc_pack_all" in names
            assert ("dc_lamb_step" if optname == "LAMB" else "dc_adam_step") in names
        else:
            step(*z)
            step(*z)
        out = []
        for i, (x, y) in enumerate(batches):
            opt.param_groups[0]["lr"] = 1e-3 * (1 + i)
            step(x.to(dev), y.to(dev))
            torch.cuda.synchronize()
            out.append((step.loss(), step.iou(), net.engine.params.clone(), net.engine.buffers.clone()))
        return out

    a, b = run(False), run(True)
    for (la, ia, pa, ba), (lb, ib, pb, bb) in zip(a, b):
        assert abs(la - lb) <= 1e-12 * abs(la) and ia == ib
        assert torch.equal(pa, pb) and torch.equal(ba, bb)


def test_program_slots_and_errors():
    """dc_program_* by hand: a slot bound before each run, the failing call reported by index and name, unrecordable names refused."""
    dev = torch.device("cuda", 0)
    a, b = torch.full((64,), 7, dtype=torch.uint8, device=dev), torch.full((64,), 7, dtype=torch.uint8, device=dev)
    prog = L.Program()
    st = L.stream_ptr()
    prog.append("dc_memset_async", L.Slot("buf", L.dptr(a)), 1, 64, st)
    prog.append("dc_stream_fence", st, st)
    assert len(prog) == 2 and prog.names() == ["dc_memset_async", "dc_stream_fence"]
    prog.run()
    prog.bind("buf", L.dptr(b))
    prog.run()
    torch.cuda.synchronize()
    assert int(a.sum()) == 64 and int(b.sum()) == 64
    with pytest.raises(L.DeepcamHipError, match="no slot named"):
        prog.bind("nope", 0)
    with pytest.raises(L.DeepcamHipError, match="cannot be recorded"):
        prog.append("dc_last_error")
    bad = L.Program()
    bad.append("dc_memset_async", None, 0, 16, st)          # null destination: the entry point refuses, the program reports which call
    with pytest.raises(L.DeepcamHipError, match=r"program call 0 \(dc_memset_async\)"):
        bad.run()


def test_backward_without_zero_grad_accumulates():
    """torch semantics through the module surface: a loss.backward() that arrives without an optimizer.zero_grad() /
    step() since the previous one ADDS to p.grad (the engine's backward overwrites, so the earlier gradients are set aside and added back);
    zero_grad() or step() ends the window.  Checked on two micro-batches: grad after (backward A, backward B) == grad A + grad B, the same
    after a zero_grad() in between is grad B alone, and the reference's own order is unaffected."""
    B, H, W = 2, 64, 96
    dev = torch.device("cuda", 0)
    xa, ya = make_inputs(B, H, W)
    xb = torch.flip(xa, dims=(0,)) * 0.5 + 0.25
    yb = torch.flip(ya, dims=(0,))
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.float32, seed=333)
    net.materialize(B, H, W)
    net.train()
    opt = dnn.make_optimizer("AdamW", net, 1e-3, 1e-8, 1e-2)
    cw = olm.class_weights(-0.125)

    def grad_of(x, y, zero):
        if zero:
            opt.zero_grad()
        dnn.fp_loss(net(x.to(dev)), y.to(dev), cw).backward()
        torch.cuda.synchronize()
        return net.engine.grads.clone()

    ga = grad_of(xa, ya, True)
    gb = grad_of(xb, yb, True)
    assert not torch.equal(ga, gb)
    g1 = grad_of(xa, ya, True)
    gab = grad_of(xb, yb, False)                 # no zero_grad: accumulates
    assert torch.equal(g1, ga)
    # (BatchNorm running statistics move between the calls, the batch statistics that enter the gradient do not: same bits)
    torch.testing.assert_close(gab, ga + gb, rtol=0, atol=0)
    gabb = grad_of(xb, yb, False)                # a third micro-batch in the same window
    torch.testing.assert_close(gabb, (ga + gb) + gb, rtol=0, atol=0)
    assert torch.equal(grad_of(xb, yb, True), gb)      # zero_grad ends the window
    p = dict(net.named_parameters())["upsample.last_deconv.0.weight"]
    assert p.grad is not None and torch.equal(p.grad.flatten(), net.engine.grad_view("upsample.last_deconv.0.weight").flatten())
    opt.step()
    after_step = grad_of(xa, ya, False)                # ... and so does step(): the first backward behind it stands alone
    assert torch.equal(after_step, grad_of(xa, ya, True)) and not torch.equal(after_step, ga)


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


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs beyond B=2 step 0, and well-conditioned backward checks
# ------------------------------------------------------------------------------------------------------------------------------
def _full_steps(B, dtype, nsteps, optname="Adam", wd=1e-6):
    x, y = make_inputs(B, 768, 1152)
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=dtype, seed=333)
    net.materialize(B, 768, 1152)
    opt = dnn.make_optimizer(optname, net, 1e-3, 1e-8, wd)
    step = dnn.TrainStep(net, opt, CW, B, 768, 1152, with_metrics=True)
    xd, yd = x.to(DEV), y.to(DEV)
    out = []
    for _ in range(nsteps):
        step(xd, yd)
        torch.cuda.synchronize()
        out.append((step.loss(), step.iou(), [int((step.pred == j).sum()) for j in range(3)]))
    return out, net


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("B,fixture", [(2, "model_full.json"), (4, "model_full_b4.json"), (8, "model_full_b8.json"), (8, "model_full_b8_adamw.json")],
                         ids=["b2", "b4", "b8", "b8-adamw"])
def test_golden_full_size_two_adam_steps(golden_dir, B, fixture, dtype):
    """768x1152 at local batch 2 (configs[1]), 4 (configs[2]) and 8 (configs[4] per GPU: the benched shape, where the batch-8-only kernel
    choices are the defaults): loss, IoU and argmax histogram of TWO Adam steps against the reference's own run.  Step 0 is a pure function of identical weights.  Step 1 follows an Adam update of lr*sign(g) on every
    weight: measured (scripts/grad_check.py, profiles/r02_grad_check.txt) fp32 1.6e-4 / 1.8e-4 off the reference at B=2 / 4 --
    inside north_star's 1e-3 -- and bf16 2.3e-3 / 7.6e-4 / 6.3e-3 at B = 2 / 4 / 8 (its gradient is noisier, see above), held to the envelope per
    fixture below."""
    g = json.load(open(os.path.join(golden_dir, fixture)))
    # [b8-adamw]: AdamW with weight decay 1e-2 (train_hdf5_ddp.py:215-216), the decoupled-decay half of the benched LAMB path, at the benched shape
    key, optname, wd = ("adamw_wd1e-2", "AdamW", 1e-2) if "adamw_wd1e-2" in g else ("adam_wd1e-6", "Adam", 1e-6)
    ref = g[key]["steps"]
    got, net = _full_steps(B, dtype, 2, optname, wd)
    f32 = dtype == torch.float32
    # bf16 behind ONE update: the measured envelope per fixture instead of a blanket 1e-2 (profiles/r06_third_step.txt: B=2 2.3e-3, B=4 7.6e-4,
    # B=8 6.3e-3, B=8 AdamW 6.3e-3; the runs are deterministic, the bounds sit 30 - 60 % outside)
    bf16_step1 = {"model_full.json": 4e-3, "model_full_b4.json": 2e-3, "model_full_b8.json": 8e-3, "model_full_b8_adamw.json": 8e-3}[fixture]
    ltol = ((2e-5, 1e-3) if f32 else (1e-3, bf16_step1))
    itol = ((1e-3, 5e-3) if f32 else (5e-3, 1e-2))
    for s in range(2):
        loss, iou, hist = got[s]
        print(f"[B={B} {dtype} step {s}] loss {loss:.7f} vs {ref[s]['loss']:.7f} (rel {abs(loss - ref[s]['loss']) / ref[s]['loss']:.2e}); "
              f"iou {iou:.6f} vs {ref[s]['iou']:.6f}; argmax histogram {hist} vs {ref[s]['pred_hist']}")
        assert loss == pytest.approx(ref[s]["loss"], rel=ltol[s]), f"step {s}"
        assert iou == pytest.approx(ref[s]["iou"], rel=itol[s]), f"step {s}"
        assert sum(hist) == B * 768 * 1152
        # argmax histogram: every class count within 2 % (fp32: 0.2 %) of the image away from the reference's
        tol_px = (0.002 if f32 else 0.02) * B * 768 * 1152
        assert all(abs(a - b) <= tol_px for a, b in zip(hist, ref[s]["pred_hist"])), (hist, ref[s]["pred_hist"])
    if f32:
        dg = g[key]["final_state_digest"]
        sd = net.state_dict()
        assert int(sd["xception_features.bn1.num_batches_tracked"]) == 2
        for k in ("xception_features.bn1.running_mean", "xception_features.bn1.running_var", "global_avg_pool.2.running_var"):
            assert float(sd[k].double().abs().sum()) == pytest.approx(dg[k]["abs"], rel=2e-3), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_golden_full_size_third_adam_step(golden_dir, dtype):
    """the benched shape (B=8, 768x1152) followed for THREE Adam steps of the reference (tests/golden/
    model_full_b8_3steps.json, made by make_golden.py --only full_b8_3steps; its steps 0 and 1 are model_full_b8.json's).  north_star's
    "loss curve within 1e-3 of the reference" is asserted for the fp32 engine on all three steps; the bf16 engine is held to the envelope
    measured per step (step 0: 1e-3 = north_star; behind updates: a bf16 gradient sends the chaotic network down another trajectory,
    DESIGN section 4) -- the bounds sit just outside the measured values instead of a blanket 1e-2."""
    g = json.load(open(os.path.join(golden_dir, "model_full_b8_3steps.json")))
    ref = g["adam_wd1e-6"]["steps"]
    base = json.load(open(os.path.join(golden_dir, "model_full_b8.json")))["adam_wd1e-6"]["steps"]
    assert [r["loss"] for r in ref[:2]] == [r["loss"] for r in base]              # the same reference run, one step further
    got, net = _full_steps(8, dtype, 3, "Adam", 1e-6)
    f32 = dtype == torch.float32
    # measured (profiles/r06_third_step.txt): fp32 3.5e-8 / 1.2e-4 / 4.2e-4; bf16 2.7e-5 / 6.3e-3 / 1.3e-2
    ltol = (2e-5, 1e-3, 1e-3) if f32 else (1e-3, 8e-3, 1.6e-2)
    for s in range(3):
        loss, iou, hist = got[s]
        rel = abs(loss - ref[s]["loss"]) / ref[s]["loss"]
        print(f"[B=8 {dtype} step {s}] loss {loss:.7f} vs {ref[s]['loss']:.7f} (rel {rel:.2e}); iou {iou:.6f} vs {ref[s]['iou']:.6f}; "
              f"argmax histogram {hist} vs {ref[s]['pred_hist']}")
        assert rel <= ltol[s], f"step {s}: {rel:.3e} > {ltol[s]}"
        assert iou == pytest.approx(ref[s]["iou"], rel=5e-3 if f32 else 1.5e-2), f"step {s}"
        tol_px = (0.002 if f32 else 0.02) * 8 * 768 * 1152
        assert all(abs(a - b) <= tol_px for a, b in zip(hist, ref[s]["pred_hist"])), (hist, ref[s]["pred_hist"])
    if f32:
        sd = net.state_dict()
        assert int(sd["xception_features.bn1.num_batches_tracked"]) == 3
        dg = g["adam_wd1e-6"]["final_state_digest"]
        for k in ("xception_features.bn1.running_mean", "xception_features.bn1.running_var", "global_avg_pool.2.running_var"):
            assert float(sd[k].double().abs().sum()) == pytest.approx(dg[k]["abs"], rel=2e-3), k


def test_loss_curve_fp32_engine_vs_oracle_24_steps():
    """north_star's "loss curve within 1e-3" as a test instead of a log file: 24 LAMB steps -- one validation interval
    of scripts/convergence_pair.sh -- of the fp32 engine and of the CPU oracle on the SAME four learnable synthetic samples at 64 x 96, from
    the same seed-333 weights.  Compared: every step's loss while the two runs are the same trajectory, and the mean over the interval
    (what profiles/r0*_convergence_curves.txt tabulates).  At this size the network amplifies fp32 re-association noise by ~300 x per
    forward pass (profiles/sensitivity_r01.txt): the two fp32 implementations are ONE trajectory for two steps (1.5e-7, 4.9e-4) and two
    trajectories of the same curve from there on (1e-3 .. 2.3e-2 per step, 5.8e-3 on the interval mean, the same final loss to 1.5 %) --
    at full size the fp32 engine holds 1e-3 against the reference for three steps (test_golden_full_size_third_adam_step).  The bounds
    are that measured envelope."""
    from mlperf_deepcam_amd.data import SyntheticHWC
    B, H, W, STEPS = 4, 64, 96, 24
    ds = SyntheticHWC(B, H, W, 16, learnable=True)
    data, lab = np.empty((B, H, W, 16), np.float32), np.empty((B, H, W), np.int64)
    for i in range(B):
        ds.read_into(i, data[i], lab[i])
    # the reference's host-side normalisation (cam_hdf5_dataset.py:117-122), then CHW
    x = torch.from_numpy(((data - ds.data_shift) * ds.data_scale).transpose(0, 3, 1, 2).copy())
    y = torch.from_numpy(lab)
    sd = omodel.init_state(333)
    keys = omodel.param_keys(sd)
    params = [sd[k].requires_grad_(True) for k in keys]
    opt = ooptim.OracleOptimizer([p.detach() for p in params], "LAMB", lr=2e-3, eps=1e-8, weight_decay=1e-2)
    ref = []
    for _ in range(STEPS):
        for p in params:
            p.grad = None
        loss = olm.fp_loss(omodel.forward(sd, x, training=True), y, CW)
        loss.backward()
        opt.step([p.grad for p in params])
        ref.append(float(loss.detach()))
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.float32, seed=333)
    net.materialize(B, H, W)
    step = dnn.TrainStep(net, dnn.make_optimizer("LAMB", net, 2e-3, 1e-8, 1e-2), CW, B, H, W)
    got = []
    for _ in range(STEPS):
        step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        got.append(step.loss())
    rel = [abs(a - b) / b for a, b in zip(got, ref)]
    mean_rel = abs(sum(got) / STEPS - sum(ref) / STEPS) / (sum(ref) / STEPS)
    print("[curve fp32 vs oracle] per-step rel: " + " ".join(f"{r:.1e}" for r in rel) + f"; interval mean rel {mean_rel:.2e}; "
          f"loss {ref[0]:.4f} -> {ref[-1]:.4f} (oracle) {got[-1]:.4f} (engine)")
    # measured (profiles/r06_third_step.txt): 1.5e-7, 4.9e-4, then 1e-3 .. 2.3e-2 per step; interval mean 5.8e-3; 2.011 -> 0.107 / 0.108
    assert ref[-1] < 0.1 * ref[0] and got[-1] < 0.1 * got[0]                      # both learn the synthetic task
    assert rel[0] <= 2e-5 and rel[1] <= 1e-3                                      # one trajectory: north_star's bound holds
    assert max(rel) <= 4e-2 and mean_rel <= 1e-2                                  # two fp32 trajectories of a chaotic map: the same curve at 1e-2
    assert got[-1] == pytest.approx(ref[-1], rel=5e-2)


def test_bench_configuration_b8_lamb_full_size():
    """BASELINE configs[4]'s per-GPU shape, the one bench.py times: local batch 8, bf16, LAMB, 768x1152.  Finite, bit-reproducible
    across two independently built models, and the bf16 loss curve against the fp32 engine's: 1e-3 at step 0 (same weights);
    after LAMB updates the two precisions part -- measured 7.2e-3 at step 1 and 1.5e-2 at step 2 -- and both keep descending."""
    runs = [_full_steps(8, torch.bfloat16, 3, "LAMB", 1e-2) for _ in range(2)]
    (a, neta), (b, netb) = runs
    assert torch.equal(neta.engine.params, netb.engine.params) and torch.equal(neta.engine.buffers, netb.engine.buffers)
    assert [v[1:] for v in a] == [v[1:] for v in b]                      # IoU and histograms bit-equal (the loss is a fp64 atomic sum)
    assert torch.isfinite(neta.engine.params).all() and torch.isfinite(neta.engine.grads).all()
    del netb, runs
    torch.cuda.empty_cache()
    f, _ = _full_steps(8, torch.float32, 3, "LAMB", 1e-2)
    for s, tol in enumerate((1e-3, 2e-2, 4e-2)):
        print(f"[B=8 LAMB step {s}] bf16 {a[s][0]:.7f} fp32 {f[s][0]:.7f} rel {abs(a[s][0] - f[s][0]) / f[s][0]:.2e}")
        assert a[s][0] == pytest.approx(f[s][0], rel=tol), f"step {s}"
    assert a[0][0] > a[1][0] > a[2][0] and f[0][0] > f[1][0] > f[2][0]  # the curve descends in both precisions


def test_lamb_step_fp32_engine_vs_oracle_b8():
    """The bench configuration's optimizer against the oracle at a size the oracle affords (B=8, 64x96): loss of step 0 to 2e-5,
    and the loss AFTER one LAMB update (which depends on every gradient through the global norm and the trust ratios)."""
    B, H, W = 8, 64, 96
    x, y = make_inputs(B, H, W)
    sd = omodel.init_state(333)
    keys = omodel.param_keys(sd)
    params = [sd[k].requires_grad_(True) for k in keys]
    opt = ooptim.OracleOptimizer([p.detach() for p in params], "LAMB", lr=1e-3, eps=1e-8, weight_decay=1e-2)
    ref = []
    for _ in range(2):
        for p in params:
            p.grad = None
        loss = olm.fp_loss(omodel.forward(sd, x, training=True), y, CW)
        loss.backward()
        opt.step([p.grad for p in params])
        ref.append(float(loss.detach()))
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.float32, seed=333)
    net.materialize(B, H, W)
    step = dnn.TrainStep(net, dnn.make_optimizer("LAMB", net, 1e-3, 1e-8, 1e-2), CW, B, H, W)
    got = []
    for _ in range(2):
        step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        got.append(step.loss())
    print(f"[LAMB B=8 64x96] engine {got} oracle {ref}")
    assert got[0] == pytest.approx(ref[0], rel=2e-5)
    assert got[1] == pytest.approx(ref[1], rel=3e-4)                      # measured 3.3e-5


def _shared_point_engines(B, H, W):
    """A bf16 and an fp32 engine on ONE linearisation point: identical effective weights (masters rounded to bf16), the bf16
    engine's forward state (every stored activation, every BatchNorm coefficient vector) copied into the fp32 engine, and the same
    d(loss)/d(logits).  Backward is then the same LINEAR map in both, evaluated in bf16 and in fp32 arithmetic."""
    x, y = make_inputs(B, H, W)
    # (the bf16 engine stores the head's input here as the fp32 engine does, so that every activation of the point can be copied)
    old = os.environ.get("DC_FUSE_BN_INTO_HEAD")
    os.environ["DC_FUSE_BN_INTO_HEAD"] = "0"
    try:
        e16 = Engine(B, H, W, torch.bfloat16, seed=333)
    finally:
        if old is None:
            del os.environ["DC_FUSE_BN_INTO_HEAD"]
        else:
            os.environ["DC_FUSE_BN_INTO_HEAD"] = old
    e32 = Engine(B, H, W, torch.float32, seed=333)
    for e in (e16, e32):
        e.params.copy_(e.params.to(torch.bfloat16).float())
        e.mark_weights_changed()
    for e in (e16, e32):
        lg = e.forward(x.to(DEV), train=True)
        dnn.wce_fused(lg, y.to(DEV), CW, dlogits=e.dlogits)
    torch.cuda.synchronize()
    # (the bf16 engine never stores the output gradient of the entry flow's thin pointwise convs: dc_pw_bn_bwd forms it in registers)
    extra = set(e32.saved) - set(e16.saved)
    assert set(e16.saved) <= set(e32.saved) and all(k.startswith(("dxception_features.block1.", "dxception_features.block2.")) and k.endswith(".pw")
                                                    for k in extra), extra
    for k, t in e16.saved.items():
        assert e32.saved[k].shape == t.shape, k
        e32.saved[k].copy_(t)
    e32.dlogits.copy_(e16.dlogits)
    e16.backward()
    e32.backward()
    torch.cuda.synchronize()
    return e16, e32


def _stage_of(name):
    """Stored activation / coefficient name -> the part of the network it belongs to."""
    if name.startswith("xception_features."):
        rest = name[len("xception_features."):]
        if rest.startswith("block"):
            b = int(rest[5:].split(".")[0])
            return "entry" if b <= 3 else ("middle" if b <= 19 else "exit")
        return "entry" if rest[:5] in ("conv1", "conv2", "bn1.", "bn2.") or rest.startswith(("conv1", "conv2", "bn1", "bn2")) else "exit"
    if name.startswith(("aspp", "global_avg_pool", "conv1", "bn1")):
        return "aspp"
    return "decoder"


def test_forward_activations_by_stage_bf16_vs_fp32_full_size():
    """The bf16 FORWARD kernels stage by stage: both engines run on bf16-rounded weights and the same input, each on its
    own; every stored activation of the bf16 engine is compared with the fp32 engine's.  The error grows along the network (every
    layer rounds its output to bf16 and the BatchNorms renormalise), so the bound is per stage; a wrong kernel in one stage shows as
    a jump there rather than as a loss within north_star's 1e-3."""
    B, H, W = 2, 768, 1152
    x, _ = make_inputs(B, H, W)
    worst = {}
    junk = torch.full((1 << 30,), float("nan"), device=DEV)      # whatever the allocator hands out next is NaN: pad channels must not matter
    del junk
    e32 = Engine(B, H, W, torch.float32, seed=333)
    e32.params.copy_(e32.params.to(torch.bfloat16).float())
    e32.mark_weights_changed()
    e32.forward(x.to(DEV), train=True)
    torch.cuda.synchronize()
    valid = lambda e, k, t: t[..., :e.saved_channels[k]] if k in e.saved_channels else t      # never the pad channels of a pixel row
    ref = {k: valid(e32, k, t).float().cpu() for k, t in e32.saved.items()}
    del e32
    torch.cuda.empty_cache()
    e16 = Engine(B, H, W, torch.bfloat16, seed=333)
    e16.params.copy_(e16.params.to(torch.bfloat16).float())
    e16.mark_weights_changed()
    e16.forward(x.to(DEV), train=True)
    torch.cuda.synchronize()
    # (the bf16 engine never stores the head's input, the BatchNorm + ReLU output of upsample.deconv3: Engine.fuse_bn_into_head)
    assert {k for k in set(ref) ^ set(e16.saved) if not k.startswith("d")} <= {"upsample.deconv3.1"}
    for k, t in e16.saved.items():
        if k not in ref:
            continue
        # forward activations only (the gradient buffers are registered under the same dictionary and hold nothing yet); the
        # per-channel coefficient vectors are covered through the activations they produce
        if t.numel() < 4096 or not k.startswith(("xception_features.", "aspp", "global_avg_pool", "conv1", "bn1", "conv2", "bn2", "last_conv", "upsample")):
            continue
        err = _rel_l2(valid(e16, k, t).float().cpu(), ref[k])
        st = _stage_of(k)
        if err > worst.get(st, ("", 0.0))[1]:
            worst[st] = (k, err)
    print("[bf16 vs fp32 forward, stored activations, worst relative L2 per stage]", worst)
    assert set(worst) >= {"entry", "middle", "exit", "aspp", "decoder"}, sorted(worst)
    # measured (worst tensor per stage): entry 2.3e-2 (block3), middle 0.19 (block19), exit 0.27 (bn5), ASPP 0.36, decoder 0.46 -- the
    # network amplifies every rounding (profiles/sensitivity_r01.txt), so only the entry flow is a sharp check; behind it the bounds
    # separate "bf16 noise, amplified" from a kernel that computes something else (error of order 1 from the first layer it touches)
    bound = {"entry": 4e-2, "middle": 0.3, "exit": 0.4, "aspp": 0.5, "decoder": 0.6}
    for st, (k, err) in worst.items():
        assert err < bound[st], (st, k, err)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (2, 768, 1152)], ids=["small", "full"])
def test_backward_parity_at_shared_activations(B, H, W):
    """Model-level check of the bf16 BACKWARD kernels (data gradients, weight gradients, depthwise, BatchNorm backward, head) that
    does not suffer from the network's ill-conditioning: both engines differentiate at the same stored activations, so the only
    difference is bf16 rounding of the gradient activations along the way.  Per tensor and over the whole arena."""
    e16, e32 = _shared_point_engines(B, H, W)
    errs = []
    for k in e32.layout.params:
        errs.append((_rel_l2(e16.grad_view(k), e32.grad_view(k)), k))
    errs.sort(reverse=True)
    total = _rel_l2(e16.grads, e32.grads)
    cos = float((e16.grads.double() @ e32.grads.double()) / (e16.grads.double().norm() * e32.grads.double().norm()))
    med = errs[len(errs) // 2][0]
    print(f"[shared point {B}x{H}x{W}] whole-arena rel L2 {total:.3e}, cosine {cos:.6f}, median per-tensor {med:.3e}, worst {errs[:4]}")
    # measured: whole arena 1.3e-2 / 1.6e-2 (full / small), cosine 0.99992 / 0.99987, worst tensor 2.6e-2 / 2.0e-2
    assert total < 3e-2 and cos > 0.9995
    assert med < 2.5e-2
    assert errs[0][0] < 6e-2, errs[:4]


def test_directional_derivatives_full_size():
    """<grad, d> against central differences of the fp32 engine's loss, 768x1152, B=2 (the check VERDICT r01 asks for, where it is
    well conditioned).  Directions are the fp32 gradient restricted to a parameter group, normalised.  Measured
    (profiles/r02_grad_check.txt): decoder group -- fp32 5e-6, bf16 0.5 % off the finite difference; all parameters -- fp32
    converges to its finite difference as eps -> 0 (1.8 % at 1e-3: the loss is that non-linear along its own gradient), bf16's
    projection is 0.56 of it (no assertion: see the comment at the top of this file)."""
    B, H, W = 2, 768, 1152
    x, y = make_inputs(B, H, W)
    xd, yd = x.to(DEV), y.to(DEV)

    def loss_of(eng, backward):
        s = torch.zeros(1, dtype=torch.float64, device=DEV)
        lg = eng.forward(xd, train=True)
        dnn.wce_fused(lg, yd, CW, dlogits=eng.dlogits if backward else None, loss_sum=s)
        if backward:
            eng.backward()
        torch.cuda.synchronize()
        return float(s.item()) / yd.numel()

    e32 = Engine(B, H, W, torch.float32, seed=333)
    loss_of(e32, True)
    g32 = e32.grads.clone()
    e16 = Engine(B, H, W, torch.bfloat16, seed=333)
    loss_of(e16, True)
    g16 = e16.grads.clone()
    del e16
    lay = e32.layout
    import math

    def direction(sel):
        d = torch.zeros_like(g32)
        for n, p in lay.params.items():
            if sel(n):
                k = math.prod(p.shape)
                d[p.offset:p.offset + k] = g32[p.offset:p.offset + k]
        return d / float(d.double().norm())

    p0 = e32.params.clone()

    def fd(d, eps):
        vals = []
        for sgn in (1.0, -1.0):
            e32.params.copy_(p0 + sgn * eps * d)
            e32.mark_weights_changed()
            vals.append(loss_of(e32, False))
        e32.params.copy_(p0)
        e32.mark_weights_changed()
        return (vals[0] - vals[1]) / (2 * eps)

    d_dec = direction(lambda n: n.startswith("upsample.") or n.startswith("conv2") or n.startswith("bn2"))
    f = fd(d_dec, 3e-3)
    a32, a16 = float(g32.double() @ d_dec.double()), float(g16.double() @ d_dec.double())
    print(f"[decoder direction] FD {f:.6e}  <g32,d> {a32:.6e}  <g16,d> {a16:.6e}")
    assert a32 == pytest.approx(f, rel=1e-3)
    assert a16 == pytest.approx(f, rel=1.5e-2)                             # measured 5.3e-3
    d_all = direction(lambda n: True)
    f = fd(d_all, 2e-4)
    a32, a16 = float(g32.double() @ d_all.double()), float(g16.double() @ d_all.double())
    print(f"[whole-gradient direction] FD {f:.6e}  <g32,d> {a32:.6e}  <g16,d> {a16:.6e} (ratio {a16 / a32:.3f})")
    assert a32 == pytest.approx(f, rel=1.2e-2)                             # measured 3.9e-3 at eps 2e-4
    assert a16 > 0.25 * a32                      # a descent direction; measured 0.56
