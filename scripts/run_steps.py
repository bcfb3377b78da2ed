"""Plain train steps (no event brackets, no roofline passes) for kernel traces:  rocprofv3 --kernel-trace ... -- python3 scripts/run_steps.py [B] [steps]
Honours the DC_* switches (also the timing-only DC_DEBUG_* ones that bench.py refuses)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import nn as dnn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333); net.materialize(B, 768, 1152); net.train()
opt = dnn.make_optimizer("LAMB", net, 1e-3, 1e-8, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(), B, 768, 1152)
g = torch.Generator().manual_seed(1); dev = torch.device("cuda", 0)
x = torch.rand(B, 16, 768, 1152, generator=g).to(dev); y = torch.randint(0, 3, (B, 768, 1152), generator=g).to(dev)
for _ in range(3): step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step(x, y)
torch.cuda.synchronize(); print("MS", (time.perf_counter() - t0) / steps * 1e3, step.loss())
