"""Measurements behind the tolerances of tests/test_model_gpu.py (run on the GPU box; prints, asserts nothing).

  1. full-size B=2: bf16 engine gradients against fp32 engine gradients (whole arena, per tensor) and the golden digests
  2. directional derivatives: <g, d> of both engines against central differences of the fp32 engine's loss, eps sweep
  3. step-1 / step-2 losses after Adam (B=2, B=4) and LAMB (B=8), fp32 and bf16, against the golden where it exists

    python scripts/grad_check.py [--size 768 1152] [--skip_steps]
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlperf_deepcam_amd import nn as dnn  # noqa: E402
from mlperf_deepcam_amd.engine import Engine  # noqa: E402
from util_inputs import make_inputs  # noqa: E402

DEV = torch.device("cuda", 0)
CW = dnn.class_weights(-0.125)


def loss_and_grads(eng, x, y, backward=True):
    s = torch.zeros(1, dtype=torch.float64, device=DEV)
    lg = eng.forward(x, train=True)
    dnn.wce_fused(lg, y, CW, dlogits=eng.dlogits if backward else None, loss_sum=s)
    if backward:
        eng.backward()
    torch.cuda.synchronize()
    return float(s.item()) / y.numel()


def directions(eng, g32):
    """Unit directions: the fp32 gradient restricted to parameter groups (large signal), and random directions scaled per tensor
    by the weights' RMS (relative perturbation)."""
    lay = eng.layout
    out = {}
    groups = {"all": lambda n: True, "encoder": lambda n: n.startswith("xception_features."),
              "middle_pw": lambda n: "block" in n and "pointwise" in n, "depthwise": lambda n: n.endswith("conv1.weight") and "rep" in n,
              "bn": lambda n: ".bn" in n or n.endswith(".bias") or (n.count(".") >= 2 and n.split(".")[-2].isdigit() and n.endswith("weight") and len(lay.params[n].shape) == 1),
              "aspp": lambda n: n.startswith("aspp") or n.startswith("global_avg_pool") or n in ("conv1.weight", "bn1.weight", "bn1.bias"),
              "decoder": lambda n: n.startswith("upsample.") or n.startswith("conv2") or n.startswith("bn2")}
    for gname, sel in groups.items():
        d = torch.zeros_like(g32)
        for n, p in lay.params.items():
            if sel(n):
                k = math.prod(p.shape)
                d[p.offset:p.offset + k] = g32[p.offset:p.offset + k]
        nrm = float(d.double().norm())
        if nrm > 0:
            out["grad:" + gname] = d / nrm
    gen = torch.Generator(device="cpu").manual_seed(7)
    for r in range(3):
        d = torch.zeros(lay.n_params)
        for n, p in lay.params.items():
            k = math.prod(p.shape)
            w = eng.params[p.offset:p.offset + k]
            rms = float(w.double().pow(2).mean().sqrt()) or 1.0
            d[p.offset:p.offset + k] = torch.randn(k, generator=gen) * rms
        d = d.to(DEV)
        out[f"rand{r}"] = d / float(d.double().norm())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=[768, 1152])
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--skip_steps", action="store_true")
    ap.add_argument("--skip_dirs", action="store_true")
    a = ap.parse_args()
    H, W = a.size
    B = a.batch
    x, y = make_inputs(B, H, W)
    x, y = x.to(DEV), y.to(DEV)

    e32 = Engine(B, H, W, torch.float32, seed=333)
    l32 = loss_and_grads(e32, x, y)
    g32 = e32.grads.clone()
    e16 = Engine(B, H, W, torch.bfloat16, seed=333)
    l16 = loss_and_grads(e16, x, y)
    g16 = e16.grads.clone()
    print(f"== B={B} {H}x{W}: loss fp32 {l32:.8f}  bf16 {l16:.8f}  rel {abs(l16 - l32) / l32:.2e}")
    d = (g16.double() - g32.double())
    print(f"whole-arena grad: rel L2 {float(d.norm() / g32.double().norm()):.3e}  cosine {float((g16.double() @ g32.double()) / (g16.double().norm() * g32.double().norm())):.5f}"
          f"  |g32| {float(g32.double().norm()):.4e} |g16| {float(g16.double().norm()):.4e}")
    errs = []
    for n, p in e32.layout.params.items():
        k = math.prod(p.shape)
        a32, a16 = g32[p.offset:p.offset + k].double(), g16[p.offset:p.offset + k].double()
        errs.append((float((a16 - a32).norm() / (a32.norm() + 1e-30)), float(a16.abs().sum() / (a32.abs().sum() + 1e-30)), n))
    errs.sort(reverse=True)
    print("per-tensor rel L2 (bf16 vs fp32 engine): worst 8:")
    for e, r, n in errs[:8]:
        print(f"   {e:.3e}  abs-sum ratio {r:.4f}  {n}")
    es = sorted(e for e, _, _ in errs)
    print(f"   median {es[len(es) // 2]:.3e}  90% {es[int(len(es) * .9)]:.3e}; abs-sum ratio range {min(r for _, r, _ in errs):.4f} .. {max(r for _, r, _ in errs):.4f}")
    gp = os.path.join(ROOT, "tests", "golden", "model_full.json" if B == 2 else "model_full_b4.json")
    if (H, W) == (768, 1152) and os.path.exists(gp) and B in (2, 4):
        ref = json.load(open(gp))["adam_wd1e-6"]["steps"][0]
        print(f"golden loss {ref['loss']:.8f}: fp32 rel {abs(l32 - ref['loss']) / ref['loss']:.2e}  bf16 rel {abs(l16 - ref['loss']) / ref['loss']:.2e}")
        for k, dg in ref["grad_digest"].items():
            a32, a16 = float(e32.grad_view(k).double().abs().sum()), float(e16.grad_view(k).double().abs().sum())
            print(f"   abs-sum vs golden: fp32 {a32 / dg['abs'] - 1:+.2e}  bf16 {a16 / dg['abs'] - 1:+.2e}   {k}")

    if not a.skip_dirs:
        dirs = directions(e32, g32)
        p0 = e32.params.clone()
        print("directional derivatives: name, <g32,d>, <g16,d>, central differences of the fp32 loss at eps = ...")
        for name, dvec in dirs.items():
            a32, a16 = float(g32.double() @ dvec.double()), float(g16.double() @ dvec.double())
            fds = []
            for eps in (1e-1, 3e-2, 1e-2, 3e-3, 1e-3):
                vals = []
                for sgn in (+1, -1):
                    e32.params.copy_(p0 + sgn * eps * dvec)
                    e32.mark_weights_changed()
                    vals.append(loss_and_grads(e32, x, y, backward=False))
                fds.append((eps, (vals[0] - vals[1]) / (2 * eps), vals[0] - vals[1]))
            e32.params.copy_(p0)
            e32.mark_weights_changed()
            print(f"  {name:16s} g32 {a32:+.5e}  g16 {a16:+.5e}  (g16/g32 {a16 / a32 if a32 else float('nan'):.4f})  FD: " +
                  "  ".join(f"{eps:g}:{fd:+.5e}(dL {dl:+.1e})" for eps, fd, dl in fds))

    if not a.skip_steps:
        del e32, e16
        torch.cuda.empty_cache()
        for (bb, optn, wd, n) in ((2, "Adam", 1e-6, 3), (4, "Adam", 1e-6, 2), (8, "LAMB", 1e-2, 3)):
            xx, yy = make_inputs(bb, H, W)
            xx, yy = xx.to(DEV), yy.to(DEV)
            res = {}
            for dt in (torch.float32, torch.bfloat16):
                net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=dt, seed=333)
                net.materialize(bb, H, W)
                opt = dnn.make_optimizer(optn, net, 1e-3, 1e-8, wd)
                st = dnn.TrainStep(net, opt, CW, bb, H, W, with_metrics=True)
                ls = []
                for _ in range(n):
                    st(xx, yy)
                    torch.cuda.synchronize()
                    ls.append((st.loss(), st.iou()))
                res[dt] = ls
                del net, opt, st
                torch.cuda.empty_cache()
            gp = os.path.join(ROOT, "tests", "golden", {2: "model_full.json", 4: "model_full_b4.json"}.get(bb, "none"))
            gold = json.load(open(gp))["adam_wd1e-6"]["steps"] if ((H, W) == (768, 1152) and os.path.exists(gp)) else []
            for s in range(n):
                l32, i32 = res[torch.float32][s]
                l16, i16 = res[torch.bfloat16][s]
                gtxt = f" golden {gold[s]['loss']:.8f} (fp32 rel {abs(l32 - gold[s]['loss']) / gold[s]['loss']:.2e}, bf16 rel {abs(l16 - gold[s]['loss']) / gold[s]['loss']:.2e}) iou {gold[s]['iou']:.6f}" if s < len(gold) else ""
                print(f"B={bb} {optn} step {s}: fp32 {l32:.8f} iou {i32:.6f} | bf16 {l16:.8f} iou {i16:.6f} | bf16 vs fp32 rel {abs(l16 - l32) / l32:.2e}{gtxt}")


if __name__ == "__main__":
    main()
