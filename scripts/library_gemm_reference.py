"""What the vendor GEMM library reaches on the pointwise shapes of the middle flow (context for the roofline fractions of pw384 / pw192; the
product path does not call it): torch.matmul in bf16 (hipBLASLt / rocBLAS behind it) on [M x 768] x [768 x 768], 728 channels padded to 768 as
the engine stores them, M = 27 648 / 13 824 / 6 912 pixels (local batch 8 / 4 / 2).   python scripts/library_gemm_reference.py"""
import torch
dev = torch.device("cuda", 0)
for M in (27648, 13824, 6912):
    for K, N in ((768, 768),):
        a = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(8)]       # rotating operands: nothing stays in L2 / MALL by accident
        b = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(8)]
        out = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(8)]
        for i in range(16):
            torch.matmul(a[i % 8], b[i % 8].t(), out=out[i % 8])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for i in range(reps):
            torch.matmul(a[i % 8], b[i % 8].t(), out=out[i % 8])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"M={M:6d} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s (padded)  {2.0 * M * 728 * 728 / us / 1e6:7.1f} TFLOP/s (728 channels)", flush=True)

# the same layers' weight gradients, dW[co][ci] = sum over pixels dy[p][co] * x[p][ci]: one layer, and twelve layers as one batched call (the
# grouping wgrad384 launches)
for M in (27648, 6912):
    for L in (1, 12):
        dy = [torch.randn(L, M, 768, device=dev).to(torch.bfloat16) for _ in range(3)]
        x = [torch.randn(L, M, 768, device=dev).to(torch.bfloat16) for _ in range(3)]
        out = [torch.empty(L, 768, 768, device=dev, dtype=torch.bfloat16) for _ in range(3)]
        f = lambda i: torch.bmm(dy[i % 3].transpose(1, 2), x[i % 3], out=out[i % 3])
        for i in range(6):
            f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 60
        e0.record()
        for i in range(reps):
            f(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"weight gradient, {L:2d} layer(s), {M} pixels: {us:8.1f} us  {2.0 * L * M * 768 * 768 / us / 1e6:7.1f} TFLOP/s (padded)", flush=True)
