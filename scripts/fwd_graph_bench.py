"""Forward pass only: eager launches against a captured hipGraph (one stream, ~230 kernels).  python scripts/fwd_graph_bench.py [B]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import nn as dnn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333); net.materialize(B, 768, 1152); net.train()
eng = net.engine
x = torch.rand(B, 16, 768, 1152, device=dev)
eng.x_static.copy_(x)
def fwd():
    eng.forward(eng.x_static, train=True)
for _ in range(3): fwd()
torch.cuda.synchronize()
def timeit(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) / n * 1e3
print(f"B={B} eager forward: {timeit(fwd)[0]:.3f} ms GPU")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    fwd()
torch.cuda.synchronize()
print(f"B={B} graph forward: {timeit(g.replay)[0]:.3f} ms GPU")
print(f"B={B} eager forward again: {timeit(fwd)[0]:.3f} ms GPU")
