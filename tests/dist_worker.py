"""Worker for tests/test_dist_gpu.py: 2 ranks (gloo, both on cuda:0) run forward/backward on different batches through the
engine + GradReducer (side-stream all-reduce hand-off); rank 0 checks the reduced gradients against the sum of the two
single-process gradients."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlperf_deepcam_amd import dist as ddist, nn as dnn
from mlperf_deepcam_amd.engine import Engine
from util_inputs import make_inputs

def grads_for(eng, seed, dev, cw):
    x, y = make_inputs(2, 64, 96, seed=seed)
    lg = eng.forward(x.to(dev), train=True)
    dnn.wce_fused(lg, y.to(dev), cw, dlogits=eng.dlogits)
    eng.backward()

def main():
    ddist.init("env", backend=os.environ.get("DC_TEST_BACKEND", "gloo"))
    rank, world = ddist.get_rank(), ddist.get_size()
    torch.cuda.set_device(0); dev = torch.device("cuda", 0); cw = dnn.class_weights()
    eng = Engine(2, 64, 96, torch.float32, seed=333 + rank)          # different init per rank: broadcast must fix it
    red = ddist.GradReducer(eng, world, bucket_mb=16.0)
    red.broadcast_parameters()
    grads_for(eng, 1234 + rank, dev, cw)
    red.finish(); torch.cuda.synchronize()
    reduced = eng.grads.clone()
    psum = float(eng.params.double().sum())
    if rank == 0:
        ref = Engine(2, 64, 96, torch.float32, seed=333)
        assert abs(float(ref.params.double().sum()) - psum) < 1e-6, "parameters were not broadcast from rank 0"
        total = torch.zeros_like(ref.grads)
        for r in range(world):
            # fresh BatchNorm buffers do not matter for gradients; same weights, rank r's batch
            grads_for(ref, 1234 + r, dev, cw); torch.cuda.synchronize()
            total += ref.grads
        err = float((reduced.double() - total.double()).norm() / total.double().norm())
        how = f" collective {red.collective} transport {red.comm.info()['transport']}" if red.collective == "library" else ""
        print(f"DIST_WORKER rel_err {err:.3e} launched {len(red.buckets)} buckets{how}", flush=True)
        assert err < 1e-6, err
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
