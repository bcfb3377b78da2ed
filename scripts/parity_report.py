"""bf16 engine vs fp32 engine (itself pinned to the reference) on the same seeded batch: loss, logits, argmax, gradients."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlperf_deepcam_amd import nn as dnn
from mlperf_deepcam_amd.engine import Engine
from util_inputs import make_inputs
H, W = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
DEV = torch.device("cuda", 0); CW = dnn.class_weights(-0.125)
x, y = make_inputs(B, H, W); xd, yd = x.to(DEV), y.to(DEV)
res = {}
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    eng = Engine(B, H, W, dt, seed=333)
    lg = eng.forward(xd, train=True)
    counts = torch.zeros(9, dtype=torch.int64, device=DEV)
    pred = torch.empty((B, H, W), dtype=torch.int64, device=DEV)
    s = dnn.wce_fused(lg, yd, CW, dlogits=eng.dlogits, pred=pred, counts=counts)
    eng.backward(); torch.cuda.synchronize()
    res[name] = dict(loss=float(s.item()) / y.numel(), iou=dnn.iou_from_counts(counts.cpu().tolist()), logits=lg.clone(), pred=pred.clone(),
                     grads=eng.grads.clone(), layout=eng.layout)
    del eng; torch.cuda.empty_cache()
a, b = res["f32"], res["bf16"]
rel = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
print(f"size {B}x{H}x{W}: loss f32 {a['loss']:.7f} bf16 {b['loss']:.7f} rel {abs(a['loss']-b['loss'])/a['loss']:.2e}")
print(f"iou f32 {a['iou']:.6f} bf16 {b['iou']:.6f}; argmax agreement {float((a['pred']==b['pred']).float().mean()):.5f}")
print(f"logits rel-L2 {rel(b['logits'], a['logits']):.3e}; whole-arena grad rel-L2 {rel(b['grads'], a['grads']):.3e}")
import math
errs = []
for n, p in a["layout"].params.items():
    k = math.prod(p.shape); errs.append((rel(b["grads"][p.offset:p.offset+k], a["grads"][p.offset:p.offset+k]), n))
errs.sort(reverse=True)
print("worst grads:", [(n, f"{e:.2e}") for e, n in errs[:6]]); print("median grad err %.2e" % errs[len(errs)//2][0])
