"""Worker for tests/test_dist_gpu.py::test_ddp_wrapper_runs_the_reference_loop: 2 ranks (gloo, both on cuda:0).

Runs INTEGRATION.md's Level-1 sequence verbatim (``net = DDP(net)``; ``loss.backward()``; ``optimizer.step()``: the reference's
train_hdf5_ddp.py:227,352-364) for three steps on rank-specific batches, then the fused TrainStep path on the same batches,
and on rank 0 a single-process reference that averages the two ranks' gradients by hand.  All three must give the same
weights bit for bit (world size 2: the 1/2 scaling is exact wherever it is applied), and p.grad after backward must be the
AVERAGED gradient, as under apex / torch DDP."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlperf_deepcam_amd import dist as comm, nn as dnn
from mlperf_deepcam_amd.dist import DistributedDataParallel as DDP
from util_inputs import make_inputs

B, H, W, STEPS, OPT = 2, 64, 96, 3, "LAMB"       # LAMB: its gradient clipping tells a summed gradient from an averaged one


def batch(rank, step, dev):
    x, y = make_inputs(B, H, W, seed=1234 + rank + 10 * step)
    return x.to(dev), y.to(dev)


def main():
    comm.init("env", backend="gloo")
    rank, world = comm.get_rank(), comm.get_size()
    torch.cuda.set_device(0); dev = torch.device("cuda", 0); cw = dnn.class_weights()

    # ---- path 1: the reference's loop, through autograd and the DDP wrapper
    torch.manual_seed(333 + rank)                 # ranks start from different weights: DDP must broadcast rank 0's
    net = dnn.DeepLabv3_plus(n_input=16, n_classes=3, os=16, pretrained=False, rank=rank, _print=False, dtype=torch.float32)
    net.to(dev)
    net.materialize(B, H, W)
    optimizer = dnn.make_optimizer(OPT, net, 1e-3, 1e-8, 1e-2)
    net = DDP(net)
    net.train()
    grads_seen = []
    for step in range(STEPS):
        inputs, label = batch(rank, step, dev)
        outputs = net.forward(inputs)
        loss = dnn.fp_loss(outputs, label, weight=cw, fpw_1=2.6, fpw_2=1.7)
        optimizer.zero_grad()
        loss.backward()
        grads_seen.append(torch.cat([p.grad.flatten() for p in net.module.parameters()]).clone())
        optimizer.step()
    torch.cuda.synchronize()
    p_l1 = net.module.engine.params.clone()
    assert optimizer.grad_scale == 1.0

    # ---- path 2: the fused step with the reducer attached
    torch.manual_seed(333 + rank)
    net2 = dnn.DeepLabv3_plus(n_input=16, n_classes=3, os=16, pretrained=False, rank=rank, _print=False, dtype=torch.float32)
    net2.materialize(B, H, W)
    opt2 = dnn.make_optimizer(OPT, net2, 1e-3, 1e-8, 1e-2)
    ddp2 = DDP(net2)
    ts = dnn.TrainStep(net2, opt2, cw, B, H, W)
    ts.attach_reducer(ddp2.reducer)
    assert opt2.grad_scale == 0.5
    for step in range(STEPS):
        ts(*batch(rank, step, dev))
    torch.cuda.synchronize()
    p_ts = net2.engine.params.clone()
    assert torch.equal(p_l1, p_ts), "DDP autograd path and fused TrainStep path diverged"

    # every rank holds the same weights
    both = [torch.empty_like(p_l1) for _ in range(world)]
    dist.all_gather(both, p_l1)
    assert torch.equal(both[0], both[1]), "ranks diverged"

    # ---- path 3 (rank 0): single process, gradients of both ranks' batches averaged by hand
    if rank == 0:
        torch.manual_seed(333)
        ref = dnn.DeepLabv3_plus(n_input=16, n_classes=3, os=16, pretrained=False, rank=1, _print=False, dtype=torch.float32)
        ref.materialize(B, H, W)
        ropt = dnn.make_optimizer(OPT, ref, 1e-3, 1e-8, 1e-2)
        eng = ref.engine
        for step in range(STEPS):
            gs = []
            for r in range(world):
                x, y = batch(r, step, dev)
                lg = eng.forward(x, train=True)
                dnn.wce_fused(lg, y, cw, dlogits=eng.dlogits)
                eng.backward()
                torch.cuda.synchronize()
                gs.append(eng.grads.clone())
            avg = (gs[0] + gs[1]) * 0.5
            assert torch.equal(grads_seen[step], avg), f"step {step}: p.grad after loss.backward() is not the averaged gradient"
            eng.grads.copy_(avg)
            ropt.step()
        torch.cuda.synchronize()
        assert torch.equal(eng.params, p_l1), "DDP result differs from the hand-averaged single-process reference"
        print(f"DDP_WORKER ok: {STEPS} steps, {len(ddp2.reducer.buckets)} buckets, |w| {float(p_l1.double().norm()):.6f}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
