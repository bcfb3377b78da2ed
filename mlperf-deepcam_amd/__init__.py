"""MI355X-native DeepCAM train step: host side (Python) over libdeepcam_hip.so (hand-written gfx950 kernels).

Import as ``mlperf_deepcam_amd``.  Layout:
  csrc/            HIP kernels + the C ABI (include/deepcam_hip.h)
  lib.py           ctypes binding of the C ABI (fails loudly if the library is missing)
  spec.py          layer table of DeepLabV3+/Xception and parameter-arena layout
  engine.py        explicit forward / backward program over preallocated NHWC buffers
  nn.py            reference-compatible call surface: DeepLabv3_plus, fp_loss, compute_score, optimizers, schedules
  dist.py          comm wire-up + RCCL gradient all-reduce overlapped with backward
  train.py         train_hdf5_ddp.py-compatible driver (CLI, checkpoints, MLLOG lines)
"""
__version__ = "0.1.0"
