"""How well conditioned is the seed-333 network?  fp32 engine on x and on x perturbed by relative eps noise."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlperf_deepcam_amd import nn as dnn
from mlperf_deepcam_amd.engine import Engine
from util_inputs import make_inputs
H, W = int(sys.argv[1]), int(sys.argv[2]); DEV = torch.device("cuda", 0); CW = dnn.class_weights(-0.125)
x, y = make_inputs(2, H, W); xd, yd = x.to(DEV), y.to(DEV)
eng = Engine(2, H, W, torch.float32, seed=333)
def run(xin):
    lg = eng.forward(xin, train=True).clone()
    s = dnn.wce_fused(lg, yd, CW, dlogits=eng.dlogits); eng.backward(); torch.cuda.synchronize()
    return lg, eng.grads.clone(), float(s.item()) / y.numel()
rel = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
l0, g0, s0 = run(xd)
for eps in (1e-6, 1e-5, 1e-4, 1e-3, 4e-3):
    g = torch.Generator(device=DEV).manual_seed(1)
    xp = xd * (1 + eps * torch.randn(xd.shape, generator=g, device=DEV))
    l1, g1, s1 = run(xp)
    print(f"eps {eps:.0e}: logits rel-L2 {rel(l1, l0):.3e}  grads rel-L2 {rel(g1, g0):.3e}  loss rel {abs(s1-s0)/s0:.2e}  amplification {rel(l1,l0)/eps:.1f}x")
