"""Upper bound of what the forward BatchNorm finalize launches cost, with REALISTIC data: a few normal steps first (so that every BatchNorm
has real coefficients), then DC_DEBUG_SKIP_BN_FINALIZE-style modes switched on the live engine ("async": the finalize runs unordered on a stream
of its own and the chain uses the previous step's coefficients; "fwd"/"bwd"/"both": not launched at all, the chain keeps the last coefficients).
(Set from the environment at start-up those modes never leave the all-zero fixed point: zero coefficients -> zero activations -> zero sums.)

    python scripts/async_fin_probe.py [B] [steps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import nn as dnn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333); net.materialize(B, 768, 1152); net.train()
opt = dnn.make_optimizer("LAMB", net, 1e-3, 1e-8, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(), B, 768, 1152)
g = torch.Generator().manual_seed(1); dev = torch.device("cuda", 0)
x = torch.rand(B, 16, 768, 1152, generator=g).to(dev); y = torch.randint(0, 3, (B, 768, 1152), generator=g).to(dev)
eng = step.eng


def run(mode):
    eng._debug_skip_finalize = mode
    for _ in range(3): step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step(x, y)
    torch.cuda.synchronize()
    print(f"[{mode or 'normal'}] {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step  loss {step.loss():.6f}", flush=True)


for mode in ("", "fwd", "", "bwd", "", "both", ""):
    run(mode)
