"""PCIe-inclusive rate of the train step fed by data.InputPipeline (pinned staging -> copy-stream DMA -> normalise kernel), next to
the HBM-resident rate of the same step.  The dataset is a memcpy source with the CAM5 on-disk layout (HWC fp32, 56.6 MB/sample),
i.e. what a page-cached file read costs the host.  python scripts/pipeline_rate.py [workers]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mlperf_deepcam_amd import nn as dnn, data as D

B, H, W = 8, 768, 1152
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 4


class MemcpySource(D.SyntheticHWC):
    def __init__(self, n):
        super().__init__(n, H, W)
        rs = np.random.RandomState(0)
        self._d = rs.random_sample((H, W, 16)).astype(np.float32) * 100.0
        self._l = rs.randint(0, 3, size=(H, W)).astype(np.int64)

    def read_into(self, i, data_out, label_out):
        np.copyto(data_out, self._d)
        np.copyto(label_out, self._l)
        return self.files[i]


torch.manual_seed(333)
dev = torch.device("cuda", 0)
net = dnn.DeepLabv3_plus(n_input=16, n_classes=3, os=16, pretrained=False, rank=0, dtype=torch.bfloat16)
net.to(dev)
net.materialize(B, H, W)
opt = dnn.make_optimizer("LAMB", net, 1e-3, 1e-6, 1e-2)
step = dnn.TrainStep(net, opt, dnn.class_weights(-0.125), B, H, W, with_metrics=False)
net.train()
nb = 16
pipe = D.InputPipeline(MemcpySource(nb * B), B, dtype=torch.bfloat16, device=dev, workers=workers)
t0 = None
for i, (x, y, _) in enumerate(pipe):
    if i == 4:
        torch.cuda.synchronize(); t0 = time.time()
    step(x, y)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"pipeline-fed (PCIe-inclusive): {(nb - 4) * B / dt:.1f} samples/s ({dt / (nb - 4) * 1e3:.1f} ms/step), reader workers = {workers}")
