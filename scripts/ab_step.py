"""A/B of whole-step time under tuning switches: every configuration runs in its own process (the switches are read at load).

    python scripts/ab_step.py --batch 8 --steps 15 "" "DEEPCAM_HIP_OPTIONS=wgrad384=0" "DC_WGRAD_GROUP=3 DEEPCAM_HIP_OPTIONS=wgrad384_slots=128"

Prints ms/step per configuration (min and median over three interleaved rounds, since boxes drift by a few percent)."""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, torch
ROOT = %r
if os.environ.get("AB_ROOT"):          # another tree's package + library (scripts/snapshot_tree.sh)
    ROOT = os.path.join(ROOT, os.environ["AB_ROOT"])
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlperf_deepcam_amd import nn as dnn
B, steps, optn, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
dtype = torch.bfloat16 if dt == "bf16" else torch.float32
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=dtype, seed=333); net.materialize(B, 768, 1152); net.train()
opt = dnn.make_optimizer(optn, net, 1e-3, 1e-8, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(), B, 768, 1152)
g = torch.Generator().manual_seed(1); dev = torch.device("cuda", 0)
x = torch.rand(B, 16, 768, 1152, generator=g).to(dev); y = torch.randint(0, 3, (B, 768, 1152), generator=g).to(dev)
import contextlib
ctx = torch.cuda.stream(torch.cuda.Stream(device=dev)) if os.environ.get("AB_OWN_STREAM", "0") == "1" else contextlib.nullcontext()   # A/B: a created stream instead of the default one
with ctx:
    for _ in range(4): step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step(x, y)
    torch.cuda.synchronize(); print("MS", (time.perf_counter() - t0) / steps * 1e3, step.loss())
''' % (ROOT,)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=15)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--optimizer", default="LAMB")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("configs", nargs="+")
a = ap.parse_args()
res = {c: [] for c in a.configs}
for r in range(a.rounds):
    for c in a.configs:
        env = dict(os.environ)
        for kv in c.split():
            k, _, v = kv.partition("=")
            env[k] = v
        p = subprocess.run([sys.executable, "-c", WORKER, str(a.batch), str(a.steps), a.optimizer, a.dtype], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("MS")]
        if not line:
            print("FAILED", c, p.stderr[-500:]); continue
        res[c].append(float(line[0].split()[1]))
        print(f"round {r} [{c or 'default'}] {res[c][-1]:.3f} ms  loss {line[0].split()[2]}", flush=True)
for c, v in res.items():
    if v:
        print(f"== B={a.batch} [{c or 'default'}]  min {min(v):.3f}  median {statistics.median(v):.3f} ms/step")
