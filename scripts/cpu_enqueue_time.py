"""Host time to enqueue one train step (about 830 launches through ctypes) against the GPU time of the step: the host must stay ahead.
python scripts/cpu_enqueue_time.py [B]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import nn as dnn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 768, 1152
dev = torch.device("cuda", 0)
x = torch.rand(B, 16, H, W, device=dev); y = torch.randint(0, 3, (B, H, W), device=dev)
net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=torch.bfloat16, seed=333)
net.materialize(B, H, W); net.train()
opt = dnn.make_optimizer("LAMB", net, 1e-3, 1e-8, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(-0.125), B, H, W)
for _ in range(3): step(x, y)
torch.cuda.synchronize()
# enqueue-only time: launch() without the scalar hand-shake, GPU idle at the start so nothing blocks the host
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step.launch(x, y); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step(x, y)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / 10
print(f"host enqueue {min(ts) * 1e3:.1f} ms (median {sorted(ts)[2] * 1e3:.1f}) per step; step {gpu * 1e3:.1f} ms; threads {torch.get_num_threads()}, cpus {os.cpu_count()}")
# the same step replayed from the C-side launch list (TrainStep.enable_program -> dc_program_run): one library call per step
step.enable_program()
torch.cuda.synchronize()
tp = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step._program.run(); tp.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step(x, y)
torch.cuda.synchronize()
gpu2 = (time.perf_counter() - t0) / 10
print(f"launch-list replay: host {min(tp) * 1e3:.2f} ms (median {sorted(tp)[2] * 1e3:.2f}) per step for {len(step._program)} recorded calls; step {gpu2 * 1e3:.1f} ms (includes the copy of the batch into the static buffers)")
