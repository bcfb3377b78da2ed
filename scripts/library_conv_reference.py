"""What PyTorch's own operators (MIOpen behind them) reach on three layers of the step, channels_last bf16, as context for this repository's
kernels (the product path calls none of them): the decoder's 3x3 256 -> 256 convolution at 192 x 288, the middle flow's depthwise 3x3 on 728
channels at 48 x 72, and a training-mode BatchNorm + ReLU on the same tensor; local batch 8.   python scripts/library_conv_reference.py"""
import time, torch, torch.nn.functional as F
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True


def bench(name, fn, flop=0.0, nbytes=0.0, reps=30):
    t0 = time.time()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    warm = time.time() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    extra = (f"  {flop / us / 1e6:7.1f} TFLOP/s" if flop else "") + (f"  {nbytes / us / 1e6:5.2f} TB/s" if nbytes else "")
    print(f"{name:58s} {us:9.1f} us{extra}   (first three calls {warm:.1f} s)", flush=True)


N = 8
x = torch.randn(N, 256, 192, 288, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
bench("conv 3x3 256 -> 256, 192 x 288, forward", lambda: F.conv2d(x, w, None, 1, 1), flop=2.0 * N * 192 * 288 * 256 * 256 * 9)
xd = torch.randn(N, 728, 48, 72, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
wd = (torch.randn(728, 1, 3, 3, device=dev) * 0.3).to(torch.bfloat16)
bench("depthwise 3x3, 728 channels, 48 x 72, forward", lambda: F.conv2d(xd, wd, None, 1, 1, 1, 728), nbytes=2.0 * xd.numel() * 2)
g, b = torch.ones(728, device=dev), torch.zeros(728, device=dev)
rm, rv = torch.zeros(728, device=dev), torch.ones(728, device=dev)
bench("BatchNorm2d (training) + ReLU, 728 channels, 48 x 72", lambda: F.relu(F.batch_norm(xd, rm, rv, g, b, True, 0.1, 1e-5)), nbytes=2.0 * xd.numel() * 2)
