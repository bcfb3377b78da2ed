import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from mlperf_deepcam_amd import nn as dnn
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
junk = torch.full((3 << 30,), float("nan"), device=dev); del junk      # poison what the allocator hands out next
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333); net.materialize(B, 768, 1152); net.train()
opt = dnn.make_optimizer("LAMB", net, 1e-3, 1e-8, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(), B, 768, 1152, with_metrics=True)
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 16, 768, 1152, generator=g).to(dev); y = torch.randint(0, 3, (B, 768, 1152), generator=g).to(dev)
eng = net.engine
for s in range(int(os.environ.get("NH_STEPS", "3"))):
    step(x, y); torch.cuda.synchronize()
    bad = [n for n in eng.layout.params if not torch.isfinite(eng.grad_view(n)).all()]
    badp = [n for n in eng.layout.params if not torch.isfinite(eng.param_view(n)).all()]
    print("step", s, "loss", step.loss(), "non-finite grads:", bad[:12], len(bad), "non-finite params:", badp[:6], len(badp))
