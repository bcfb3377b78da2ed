"""N>1 on the GPU path: two processes (gloo, sharing cuda:0) through Engine + GradReducer, and bench.py's N=2 code path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, port, timeout=300, nproc=2, extra_env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    env = dict(os.environ, DC_DIST_BACKEND="gloo", **(extra_env or {}))
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_reduced_gradients_equal_sum_of_rank_gradients():
    r = _torchrun([os.path.join(ROOT, "tests", "dist_worker.py")], 29621)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_WORKER rel_err" in r.stdout


def test_bench_two_ranks_prints_one_json_line():
    import json
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--local_batch_size", "2",
                   "--height", "128", "--width", "192"], 29622)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 4 and out["value"] > 0


def test_bench_four_ranks_on_one_gpu():
    """The widest rehearsal of `bench.py --gpus N` a one-GPU box allows (its process guard admits six processes on the card, and
    this test runner and the launcher are two of them; the N = 8 case itself is covered on the CPU by tests/test_dist_cpu.py): four
    ranks over gloo, 64 x 96 inputs, bucketed all-reduce overlapped with backward, barriers, max-over-ranks timing, ONE JSON line
    from rank 0."""
    import json
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--local_batch_size", "2",
                   "--height", "64", "--width", "96", "--no_cpu_baseline"], 29627, timeout=600, nproc=4)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp4" and out["value"] > 0
    assert out["step_ms"]["min"] <= out["step_ms"]["median"] <= out["step_ms"]["max"]


def test_bench_two_ranks_bf16_gradient_payload():
    """DC_GRAD_PAYLOAD=bf16: the buckets travel as bf16 (dc_grad_pack_bf16 / dc_grad_unpack_bf16 around the collective)."""
    import json
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--local_batch_size", "2",
                   "--height", "64", "--width", "96", "--no_cpu_baseline"], 29628, extra_env={"DC_GRAD_PAYLOAD": "bf16"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["switches"].get("DC_GRAD_PAYLOAD") == "bf16" and out["value"] > 0 and out["loss_last_step"] < 20


def test_gradient_payload_kernels_round_trip():
    import torch
    from mlperf_deepcam_amd import lib as L
    dev = torch.device("cuda", 0)
    for n in (8, 1000, 4096 + 5, 1 << 20):
        g = torch.randn(n + 8, device=dev)[:n] * 3.0
        g = g.clone()
        send = torch.empty(n, dtype=torch.bfloat16, device=dev)
        L.call("dc_grad_pack_bf16", n, L.dptr(g), L.dptr(send), L.stream_ptr())
        assert torch.equal(send, g.to(torch.bfloat16))                      # round-to-nearest-even, as torch's cast
        back = torch.zeros(n, device=dev)
        L.call("dc_grad_unpack_bf16", n, L.dptr(send), L.dptr(back), L.stream_ptr())
        assert torch.equal(back, send.float())


def test_ddp_wrapper_runs_the_reference_loop():
    """INTEGRATION.md Level 1 at world size 2: DDP(net); loss.backward(); optimizer.step() == fused TrainStep == hand-averaged
    single-process gradients, three LAMB steps (train_hdf5_ddp.py:227,359-364)."""
    r = _torchrun([os.path.join(ROOT, "tests", "ddp_worker.py")], 29623)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DDP_WORKER ok" in r.stdout


def test_train_driver_two_ranks_end_to_end(tmp_path):
    """python -m mlperf_deepcam_amd.train at world size 2 on synthetic data: rank 0 alone writes the checkpoint and the log,
    global_batch_size counts both ranks, the validation sums are all-reduced, both ranks finish."""
    import json
    import torch
    out = str(tmp_path / "run2")
    r = _torchrun(["-m", "mlperf_deepcam_amd.train", "--wireup_method", "env", "--run_tag", "w2", "--output_dir", out,
                   "--synthetic_samples", "12", "--local_batch_size", "2", "--height", "64", "--width", "96", "--logging_frequency", "1",
                   "--validation_frequency", "2", "--save_frequency", "2", "--optimizer", "LAMB", "--weight_decay", "1e-2",
                   "--amp_opt_level", "O1", "--max_epochs", "1", "--max_steps", "2", "--training_visualization_frequency", "0",
                   "--validation_visualization_frequency", "0"], 29624)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ev = [json.loads(l[len(":::MLLOG "):]) for l in r.stdout.splitlines() if l.startswith(":::MLLOG ")]
    by = {}
    for e in ev:
        by.setdefault(e["key"], []).append(e)
    assert by["global_batch_size"][0]["value"] == 4
    assert by["train_samples"][0]["value"] == 12 and len(by["train_loss"]) == 2          # rank 0 only: one line per step
    assert len(by["eval_accuracy"]) == 1 and 0.0 <= by["eval_accuracy"][0]["value"] <= 1.0
    assert all(0 < e["value"] < 20 for e in by["train_loss"]) and len(by["run_stop"]) == 1
    ck = os.path.join(out, "model_step_2.cpt")
    assert os.path.exists(ck)
    c = torch.load(ck, map_location="cpu", weights_only=False)
    assert c["step"] == 2 and len(c["model"]) == 532 and all(k.startswith("module.") for k in c["model"])
    assert sorted(os.listdir(os.path.join(out, "logs"))) == ["w2.log"]


def test_rccl_backend_single_rank():
    """The reducer's call pattern on the REAL backend (RCCL: in-place all_reduce of arena slices launched from the side stream,
    async work objects waited on the compute stream, broadcast of the arenas), with the one rank a one-GPU box allows: the
    collectives are identities, the stream hand-off and the API usage are what is exercised."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29625", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, DC_TEST_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_WORKER rel_err" in r.stdout


def test_bench_single_rank_through_torchrun_rccl():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, RCCL), at N = 1."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29626", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--local_batch_size", "2", "--height", "128", "--width", "192", "--no_cpu_baseline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0


def test_library_collective_two_ranks_callback_transport():
    """dc_grad_allreduce_enqueue / _wait (csrc/comm.cpp) in the reducer's place: same worker, same check (reduced gradients == sum of the two
    ranks' gradients), the reductions travel through the library's entry points with the callback transport (gloo between two ranks on one GPU)."""
    r = _torchrun([os.path.join(ROOT, "tests", "dist_worker.py")], 29629, extra_env={"DC_GRAD_COLLECTIVE": "lib", "DC_GRAD_COLLECTIVE_TRANSPORT": "callback"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_WORKER rel_err" in r.stdout


def test_library_collective_rccl_single_rank():
    """The library's RCCL transport brought up with the one rank a one-GPU box allows: ncclGetUniqueId / ncclCommInitRank resolved at run time and
    the reducer wired to it.  (At world size 1 the reducer has nothing to reduce and never reaches ncclAllReduce: that call is exercised by
    test_library_rccl_allreduce_on_a_device_buffer below.)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29630", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, DC_TEST_BACKEND="nccl", DC_GRAD_COLLECTIVE="lib"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_WORKER rel_err" in r.stdout and "transport rccl" in r.stdout


def test_bench_two_ranks_replayed_from_the_launch_list():
    """bench.py --program at world size 2: the recorded launch list holds the bucket all-reduces (dc_grad_allreduce_enqueue), their stream
    fences and the wait in front of the optimizer, so the multi-GPU step is ONE C call per step (callback transport here: gloo on one GPU)."""
    import json
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--local_batch_size", "2", "--program",
                   "--height", "64", "--width", "96", "--no_cpu_baseline"], 29631,
                  extra_env={"DC_GRAD_COLLECTIVE": "lib", "DC_GRAD_COLLECTIVE_TRANSPORT": "callback"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["launch_list_replay"] is True and out["comm"]["collective"] == "library (callback)" and out["value"] > 0
    assert out["loss_last_step"] < 20


def test_library_rccl_allreduce_on_a_device_buffer():
    """the hand-declared RCCL ABI of csrc/comm.cpp, driven directly (no GradReducer, which skips the collective at world size 1):
    dc_comm_create with a unique id at world 1 brings up a real ncclComm_t; dc_grad_allreduce_enqueue runs ncclAllReduce(SUM, in place) with the
    library's dtype codes (fp32 = 7, bf16 = 9) on the library's communication stream BEHIND a kernel that fills the buffer on a non-default
    compute stream, and dc_grad_allreduce_wait orders the compute stream behind it.  The sum over one rank is the buffer itself, so: values
    unchanged, the fill visible to the collective (ev_in ordering), a kernel enqueued after the wait sees the result (ev_out ordering),
    info()['enqueued'] advanced.  Runs in a child process: a communicator's teardown must not meet the test runner's other GPU state."""
    code = r"""
import ctypes as C, sys, torch
sys.path.insert(0, ROOT_DIR)
from mlperf_deepcam_amd import lib as L, dist as ddist
comm = ddist.LibraryComm.rccl()
assert comm.info()["transport"] == "rccl" and comm.info()["world"] == 1
compute = torch.cuda.Stream()
n = 1 << 22
for dt, code in ((torch.float32, L.DC_F32), (torch.bfloat16, L.DC_BF16)):
    buf = torch.zeros(n, dtype=dt, device="cuda")
    out = torch.zeros(n, dtype=dt, device="cuda")
    ref = (torch.arange(n, device="cuda") % 251).to(dt)
    torch.cuda.synchronize()
    before = comm.info()["enqueued"]
    with torch.cuda.stream(compute):
        for _ in range(20):                       # keep the compute stream busy so that an unordered collective would read zeros
            buf.copy_(ref * 0)
        buf.copy_(ref)
        L.call("dc_grad_allreduce_enqueue", comm.h, C.c_void_p(buf.data_ptr()), C.c_size_t(n), code, C.c_void_p(compute.cuda_stream))
        L.call("dc_grad_allreduce_wait", comm.h, C.c_void_p(compute.cuda_stream))
        out.copy_(buf)                            # behind the wait: sees the reduced buffer
    compute.synchronize()
    torch.cuda.synchronize()
    assert comm.info()["enqueued"] == before + 1
    assert torch.equal(buf, ref) and torch.equal(out, ref), dt
# an empty range is counted and skipped
L.call("dc_grad_allreduce_enqueue", comm.h, None, C.c_size_t(0), L.DC_F32, C.c_void_p(compute.cuda_stream))
comm.close()
print("RCCL_DIRECT ok")
""".replace("ROOT_DIR", repr(ROOT))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_DIRECT ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
