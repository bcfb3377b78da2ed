"""Per-call HIP-event timing of one train step, aggregated by entry point and shape (for finding what to optimise)."""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlperf_deepcam_amd import lib as L, nn as dnn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
H, W = 768, 1152
if os.environ.get('DC_WGB') is not None: L.load().dc_set_option(b'wgrad_target_blocks', int(os.environ['DC_WGB']))
if os.environ.get('DC_MODE') is not None: L.load().dc_set_option(b'igemm_mode', int(os.environ['DC_MODE']))
dev = torch.device("cuda", 0)
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=dt, seed=333); net.materialize(B, H, W)
opt = dnn.make_optimizer("AdamW", net, 1e-3, 1e-8, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(), B, H, W)
g = torch.Generator().manual_seed(1); x = torch.rand(B, 16, H, W, generator=g).to(dev); y = torch.randint(0, 3, (B, H, W), generator=g).to(dev)
for _ in range(2): step(x, y)
torch.cuda.synchronize()
orig = L.call; recs = []
def call(name, *args):
    key = name
    fl = 0.0
    if name in ("dc_conv_fwd", "dc_conv_dgrad", "dc_conv_wgrad"):
        d = args[0]._obj; N, Hi, Wi = args[1], args[2], args[3]
        k = 3 if d.transposed else d.k
        if d.transposed: macs = N * Hi * Wi * d.cin * d.cout * 9
        else:
            Ho = (Hi + 2 * d.pad - d.dil * (k - 1) - 1) // d.stride + 1; Wo = (Wi + 2 * d.pad - d.dil * (k - 1) - 1) // d.stride + 1
            macs = N * Ho * Wo * d.cin * d.cout * k * k
        fl = 2.0 * macs
        key = f"{name} k{k}s{d.stride}d{d.dil}{'T' if d.transposed else ''} {d.cin}->{d.cout} @{Hi}x{Wi}"
    elif name == "dc_conv_wgrad_partial":
        d = args[0]._obj; N, Hi, Wi, cnt = args[1], args[2], args[3], args[4]
        k = 3 if d.transposed else d.k
        if d.transposed: macs = N * Hi * Wi * d.cin * d.cout * 9
        else:
            Ho = (Hi + 2 * d.pad - d.dil * (k - 1) - 1) // d.stride + 1; Wo = (Wi + 2 * d.pad - d.dil * (k - 1) - 1) // d.stride + 1
            macs = N * Ho * Wo * d.cin * d.cout * k * k
        fl = 2.0 * macs * cnt
        key = f"{name} x{cnt} splits{args[10]} k{k}s{d.stride}d{d.dil}{'T' if d.transposed else ''} {d.cin}->{d.cout} @{Hi}x{Wi}"
    elif name.startswith("dc_dwconv") and name != "dc_dwconv_pack_weights":
        key = f"{name} C{args[1]} s{args[2]} d{args[3]} @{args[5]}x{args[6]}"
    elif name in ("dc_bn_apply", "dc_bn_bwd_reduce", "dc_bn_bwd_apply"):
        key = f"{name} M{args[1]} C{args[2]}"
    elif name in ("dc_bn_finalize",):
        key = f"{name} C{args[0]} rows{args[3]}"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *args); e1.record(); recs.append((key, name, fl, e0, e1))
L.call = call
step(x, y); torch.cuda.synchronize(); L.call = orig
agg = collections.OrderedDict(); byname = collections.Counter(); tot = 0.0
for key, name, fl, e0, e1 in recs:
    ms = e0.elapsed_time(e1); a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl; byname[name] += ms; tot += ms
print(f"B={B} {dt}: sum of per-call times {tot:.2f} ms over {len(recs)} calls")
for n, ms in byname.most_common(): print(f"  {n:24s} {ms:8.3f} ms  {100*ms/tot:5.1f}%")
print("--- top shapes")
for key, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("DC_TOP", "45"))]:
    tf = f"{fl/ms/1e9:7.1f} TF/s" if fl else ""
    print(f"{ms:8.3f} ms  x{cnt:<3d} {ms/cnt*1e3:8.1f} us/call  {tf:14s} {key}")
