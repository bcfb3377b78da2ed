"""Run-to-run determinism of the train step at the benchmark shape: the same batch from the same initial state twice; gradients
after the first backward, then parameters after each optimizer step, compared bit for bit.
python scripts/determinism.py [B] [optimizer] [steps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import nn as dnn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
optname = sys.argv[2] if len(sys.argv) > 2 else "LAMB"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H, W = 768, 1152
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, 16, H, W, generator=g).to(dev)
y = torch.randint(0, 3, (B, H, W), generator=g).to(dev)

def run():
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
    net.materialize(B, H, W); net.train()
    opt = dnn.make_optimizer(optname, net, 1e-3, 1e-8, 1e-2)
    step = dnn.TrainStep(net, opt, dnn.class_weights(-0.125), B, H, W, with_metrics=False)
    out = []
    for s in range(steps):
        step(x, y)
        torch.cuda.synchronize()
        out.append((net.engine.grads.clone(), net.engine.params.clone(), step.loss()))
    del step, opt, net
    torch.cuda.empty_cache()
    return out

a, b = run(), run()
for s in range(steps):
    ga, pa, la = a[s]; gb, pb, lb = b[s]
    ng = int((ga != gb).sum()); npar = int((pa != pb).sum())
    print(f"step {s}: loss {la:.9f} / {lb:.9f}; gradient elements that differ {ng} of {ga.numel()}"
          f" (max abs diff {float((ga - gb).abs().max()):.3e}); parameters that differ {npar}")
    if ng and s == 0:
        from mlperf_deepcam_amd import spec
        eng_layout = spec.Layout(16, 3)
        import math
        bad = []
        for name, p in eng_layout.params.items():
            n = math.prod(p.shape)
            d = int((ga[p.offset:p.offset + n] != gb[p.offset:p.offset + n]).sum())
            if d: bad.append((name, d, n))
        print("  first-step tensors with differing gradients:", bad[:12], "... total", len(bad))
