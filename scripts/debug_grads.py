import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mlperf_deepcam_amd import nn as dnn
from mlperf_deepcam_amd.engine import Engine
from oracle import loss_metric as olm, model as omodel
from util_inputs import make_inputs
H, W = int(sys.argv[1]), int(sys.argv[2]); dtype = torch.float32 if sys.argv[3] == "f32" else torch.bfloat16
CW = olm.class_weights(-0.125); DEV = torch.device("cuda", 0)
x, y = make_inputs(2, H, W)
sd = omodel.init_state(333); keys = omodel.param_keys(sd)
for k in keys: sd[k].requires_grad_(True)
out = omodel.forward(sd, x, training=True); loss = olm.fp_loss(out, y, CW); loss.backward()
# second oracle run in float64 to see the reference's own noise level
sd64 = {k: (v.detach().double() if v.is_floating_point() else v.clone()) for k, v in omodel.init_state(333).items()}
for k in keys: sd64[k].requires_grad_(True)
out64 = omodel.forward(sd64, x.double(), training=True)
w = torch.tensor(CW, dtype=torch.float64)
lse = torch.logsumexp(out64, 1); picked = torch.gather(out64, 1, y.unsqueeze(1)).squeeze(1)
loss64 = (w[y] * (lse - picked)).mean(); loss64.backward()
eng = Engine(2, H, W, dtype, seed=333)
lg = eng.forward(x.to(DEV), train=True)
s = dnn.wce_fused(lg, y.to(DEV), CW, dlogits=eng.dlogits); eng.backward(); torch.cuda.synchronize()
print("loss hip", float(s.item()) / y.numel(), "oracle32", float(loss), "oracle64", float(loss64))
def rel(a, b): a, b = a.double().flatten(), b.double().flatten(); return float((a - b).norm() / (b.norm() + 1e-30))
print("logits rel: hip-vs-64 %.2e  cpu32-vs-64 %.2e" % (rel(lg.cpu(), out64), rel(out, out64)))
rows = []
for k in keys:
    rows.append((k, rel(eng.grad_view(k).cpu(), sd64[k].grad), rel(sd[k].grad, sd64[k].grad)))
for k, a, b in rows:
    flag = " <<<" if a > 5 * max(b, 1e-6) and a > 1e-3 else ""
    print(f"{k:60s} hip {a:.2e}  cpu32 {b:.2e}{flag}")
