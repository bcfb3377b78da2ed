"""Which kernel is not run-to-run deterministic?  One forward, then the backward program twice on the same dlogits: every
activation-gradient buffer and the parameter-gradient arena are compared bit for bit, in creation (= program) order.
python scripts/find_nondeterminism.py [B] [H] [W]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import nn as dnn, engine as E
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1152
dev = torch.device("cuda", 0)
acts = []
_init = E.Act.__init__
def init(self, *a, **k):
    _init(self, *a, **k)
    if self.parent is None: acts.append(self)
E.Act.__init__ = init
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, 16, H, W, generator=g).to(dev)
y = torch.randint(0, 3, (B, H, W), generator=g).to(dev)
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
net.materialize(B, H, W); net.train()
opt = dnn.make_optimizer("AdamW", net, 1e-3, 1e-8, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(-0.125), B, H, W, with_metrics=False)
eng = step.eng
def fwd():
    step.loss_sum.zero_()
    logits = eng.forward(x, train=True)
    dnn.wce_fused(logits, y, step.weight, dlogits=eng.dlogits, pred=None, counts=None, loss_sum=step.loss_sum)
    torch.cuda.synchronize()
def snap():
    torch.cuda.synchronize()
    return [(a.name, a.buf.clone()) for a in acts], eng.grads.clone(), eng.dlogits.clone(), eng.logits.clone()
fwd(); f1 = snap()
fwd(); f2 = snap()
print("forward twice: logits equal", torch.equal(f1[3], f2[3]), "dlogits equal", torch.equal(f1[2], f2[2]))
bad = [n for (n, a), (_, b) in zip(f1[0], f2[0]) if not n.startswith("d") and not torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32), b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32))]
print("forward activations that differ (NaN pads compared as bits):", bad[:10])
eng.backward(); b1 = snap()
eng.backward(); b2 = snap()
first = None
for (n, a), (_, b) in zip(b1[0], b2[0]):
    va = a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32); vb = b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32)
    if not torch.equal(va, vb):
        d = int((va != vb).sum())
        print(f"  differs: {n}  ({d} of {va.numel()} elements)")
        first = first or n
import math
lay = eng.layout
badp = []
for name, p in lay.params.items():
    n = math.prod(p.shape)
    d = int((b1[1][p.offset:p.offset + n] != b2[1][p.offset:p.offset + n]).sum())
    if d: badp.append((name, d, n))
print("parameter gradients that differ:", len(badp), "of", len(lay.params), badp[-8:])
