#!/usr/bin/env python3
"""DeepCAM train-step benchmark on MI355X: samples/s of forward + weighted CE + backward + gradient all-reduce + optimizer
on synthetic 768x1152x16 batches resident in HBM, plus the MFMA-roofline fraction of the dominant kernel and a CPU
baseline (the oracle restatement of the reference step on the node's host cores).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; weak scaling (local batch fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_SAMPLE = 1963.97e9      # conv fwd + dgrad + wgrad, 2*MAC, BASELINE.md section 2
ENCODER_FLOP_PER_SAMPLE = 934.9e9   # the Xception encoder's share, forward + backward
PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}
CLASS_FREQ = (0.986267818390377, 0.0004578708870701058, 0.01327431072255291)


def synthetic_batch(B, H, W, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 16, H, W, generator=g)
    y = torch.multinomial(torch.tensor(CLASS_FREQ), B * H * W, replacement=True, generator=g).view(B, H, W)
    return x.to(device), y.to(device)


def cpu_baseline(H, W, local_batch=2, timed_steps=3, step_budget_s=60.0):
    """SURVEY 8d / BASELINE.md section 3: the oracle's train step (forward, fp_loss, zero_grad, backward, Adam step) on the node's
    host cores at the reference's canonical local batch (2, run_training_dgx2.sh:70), fp32, full size, 1 warm-up + 3 timed steps,
    forward / backward / optimizer split.  Bounded: if the warm-up step alone takes longer than step_budget_s only ONE step is
    timed, and the sample string says so."""
    from oracle import loss_metric as olm, model as omodel, optim as ooptim     # timed as the baseline, never shipped
    cw = olm.class_weights(-0.125)
    threads = torch.get_num_threads()
    sd = omodel.init_state(333)
    keys = omodel.param_keys(sd)
    params = [sd[k].requires_grad_(True) for k in keys]
    opt = ooptim.OracleOptimizer([p.detach() for p in params], "Adam", lr=1e-3, eps=1e-8, weight_decay=1e-6)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(local_batch, 16, H, W, generator=g)
    y = torch.randint(0, 3, (local_batch, H, W), generator=g)

    def one():
        t0 = time.perf_counter()
        loss = olm.fp_loss(omodel.forward(sd, x, training=True), y, cw)
        t1 = time.perf_counter()
        for p in params:
            p.grad = None
        loss.backward()
        t2 = time.perf_counter()
        opt.step([p.grad for p in params])
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    warm = sum(one())
    n = timed_steps if warm <= step_budget_s else 1
    runs = [one() for _ in range(n)]
    fwd, bwd, opt_s = (sum(r[i] for r in runs) / n for i in range(3))
    total = fwd + bwd + opt_s
    return {"value": local_batch / total, "unit": "samples/s", "cores": threads, "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"oracle train step (fwd + fp_loss + zero_grad + bwd + Adam), B={local_batch} {H}x{W} fp32, 1 warm-up "
                      f"({warm:.1f} s) + {n} timed step(s), mean {total:.2f} s/step",
            "seconds_per_step": round(total, 3), "forward_s": round(fwd, 3), "backward_s": round(bwd, 3), "optimizer_s": round(opt_s, 3)}


def torch_gpu_baseline(H, W, local_batch, dev, mode="nhwc", timed_steps=5):
    """--torch_gpu_baseline: the oracle's train step (the network as a chain of torch operators: F.conv2d, F.batch_norm, F.conv_transpose2d ...,
    autograd, torch.optim.Adam) on THIS GPU, eager, bf16 autocast over fp32 weights, channels_last -- what the reference's own stack
    (PyTorch operators over MIOpen / rocBLAS; the reference runs apex O1 mixed precision, train_hdf5_ddp.py:222-224) delivers on the hardware the
    headline number is quoted on.  A reported baseline like cpu_baseline, at the same local batch as the timed step; never the thing measured.
    mode: "nchw" = as the reference runs it (contiguous NCHW tensors, no algorithm search); "nhwc" = channels_last tensors; "_tuned" = with
    torch.backends.cudnn.benchmark (MIOpen's search for the fastest algorithm per convolution, minutes of warm-up)."""
    from oracle import loss_metric as olm, model as omodel      # timed as a baseline, never shipped
    cw = olm.class_weights(-0.125)
    sd = omodel.init_state(333)
    fmt = torch.contiguous_format if mode.startswith("nchw") else torch.channels_last
    torch.backends.cudnn.benchmark = mode.endswith("_tuned")
    sd = {k: (v.to(dev).contiguous(memory_format=fmt) if v.dim() == 4 else v.to(dev)) for k, v in sd.items()}
    keys = omodel.param_keys(sd)
    params = [sd[k].requires_grad_(True) for k in keys]
    opt = torch.optim.Adam(params, lr=1e-3, eps=1e-8, weight_decay=1e-6)
    x, y = synthetic_batch(local_batch, H, W, 1234, dev)
    x = x.contiguous(memory_format=fmt)

    def one():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = omodel.forward(sd, x, training=True)
        loss = olm.fp_loss(logits.float(), y, cw)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    # MIOpen compiles (and, "_tuned", searches) kernels for ~60 convolution shapes in the first steps: minutes without output otherwise
    import sys, threading
    done = threading.Event()

    def heartbeat():
        t = time.perf_counter()
        while not done.wait(60.0):
            print(f"torch_gpu_baseline [{mode}]: warming up, {time.perf_counter() - t:.0f} s", file=sys.stderr, flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    t0 = time.perf_counter()
    try:
        for _ in range(2):
            one()
        torch.cuda.synchronize(dev)
    finally:
        done.set()
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        loss = one()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / timed_steps
    out = {"value": local_batch / dt, "unit": "samples/s", "ms_per_step": round(dt * 1e3, 2), "kind": "port",
           "sample": f"oracle train step through torch operators on the GPU (eager, bf16 autocast, {mode}, torch.optim.Adam with wd 1e-6 -- the headline step's optimizer may differ (LAMB, wd 1e-2): a cheaper update on this side, so the ratio is conservative), B={local_batch} {H}x{W}, "
                     f"2 warm-up steps ({warm:.1f} s) + {timed_steps} timed", "loss_last_step": round(float(loss.detach()), 6),
           "peak_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}
    del params, sd, opt
    torch.cuda.empty_cache()
    return out


class KernelTimer:
    """HIP-event timing of every launch of the two MFMA kernel families on the stream they run on:
       igemm  dc::igemm256_kernel / dc::igemm256p_kernel / dc::pw224_kernel / dc::pw384_kernel / dc::pw192_kernel / dc::igemm_kernel<T> / dc::tiny_gemm_kernel<T>
              (dc_conv_fwd + dc_conv_dgrad: dense conv forward and data gradient; the library's planner picks the tile shape per layer)
       wgrad  dc::wgrad384_kernel / dc::wgrad_dma_kernel<T>  (dc_conv_wgrad_partial: dense conv weight gradient, split-K
              partial sums; a grouped call is ONE launch of the kernel for up to sixteen layers and counts as one) + dc::fold_kernel (dc_fold_slabs: the
              fixed-order sum of those slabs, which also folds the depthwise layers' rows; its time counts, it adds no launch or flop)
       Not in either family (neither its time nor its flop): dc::pw_bn_bwd_kernel (dc_pw_bn_bwd), the HBM-bound pass that does the BatchNorm backward
       apply, the data gradient and the weight gradient of the entry flow's two thin pointwise layers at once."""

    FAMILY = {"dc_conv_fwd": "igemm", "dc_conv_dgrad": "igemm", "dc_conv_fwd_kn": "igemm", "dc_conv_dgrad_kn": "igemm", "dc_conv_dgrad_bnstats": "igemm", "dc_conv_wgrad": "wgrad",
              "dc_conv_wgrad_group": "wgrad", "dc_conv_wgrad_partial": "wgrad"}

    def __init__(self, lib_module):
        self.L = lib_module
        self.events = {"igemm": [], "wgrad": []}
        self.fold_events = []
        self.flops = {"igemm": 0.0, "wgrad": 0.0}
        self.bytes = {"igemm": 0.0, "wgrad": 0.0}        # algorithmic: input + output + weights, each touched once
        self._orig = lib_module.call

    def __enter__(self):
        timer = self

        def call(name, *args):
            fam = timer.FAMILY.get(name)
            if name == "dc_fold_slabs":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                timer._orig(name, *args)
                e1.record()
                timer.fold_events.append((e0, e1))
                return
            if fam is None:
                return timer._orig(name, *args)
            d = args[0]._obj
            N, Hi, Wi = args[1], args[2], args[3]
            layers = args[4] if name in ("dc_conv_wgrad_group", "dc_conv_wgrad_partial") else 1   # one launch serves `layers` layers of one geometry
            k = 3 if d.transposed else d.k
            if d.transposed:
                macs = N * Hi * Wi * d.cin * d.cout * 9
                Ho, Wo = 2 * Hi, 2 * Wi
            else:
                Ho = (Hi + 2 * d.pad - d.dil * (k - 1) - 1) // d.stride + 1
                Wo = (Wi + 2 * d.pad - d.dil * (k - 1) - 1) // d.stride + 1
                macs = N * Ho * Wo * d.cin * d.cout * k * k
            esz = 2 if d.dtype == timer.L.DC_BF16 else 4
            wbytes = k * k * d.cin * d.cout * (4 if fam == "wgrad" else esz)        # the weight gradient leaves as fp32
            timer.bytes[fam] += layers * ((N * Hi * Wi * d.cin + N * Ho * Wo * d.cout) * esz + wbytes)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            timer._orig(name, *args)
            e1.record()
            timer.events[fam].append((e0, e1))
            timer.flops[fam] += 2.0 * macs * layers

        self.L.call = call
        return self

    def __exit__(self, *exc):
        self.L.call = self._orig

    def result(self):
        torch.cuda.synchronize()
        out = {}
        for fam, evs in self.events.items():
            t = sum(a.elapsed_time(b) for a, b in evs) * 1e-3
            if fam == "wgrad":
                t += sum(a.elapsed_time(b) for a, b in self.fold_events) * 1e-3
            out[fam] = (self.flops[fam], t, len(evs), self.bytes[fam])
        return out


def also_config(dnn, B, H, W, dtype_name, optimizer, steps, warmup, dev):
    """BASELINE.json configs[1] (local batch 2, fp32) and configs[2] (local batch 4, bf16) timed in the same process as the
    headline configuration, the same way (W warm-up steps, K steps between two synchronisations, nothing skipped), each with its whole-step and
    encoder-region fraction of the dtype's MFMA peak.  Single GPU only; runs after the headline's timed region and roofline passes."""
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=dtype, seed=333)
    net.materialize(B, H, W)
    net.train()
    wd = 1e-2 if optimizer != "Adam" else 1e-6
    opt = dnn.make_optimizer(optimizer, net, 1e-3, 1e-8, wd)
    step = dnn.TrainStep(net, opt, dnn.class_weights(-0.125), B, H, W, with_metrics=False)
    x, y = synthetic_batch(B, H, W, 1234, dev)
    for _ in range(warmup):
        step(x, y)
    import gc
    gc.collect()
    gc.disable()
    try:
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            step(x, y)
        marks[steps].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    loss = step.loss()
    sps = steps * B / dt
    peak = PEAK[dtype_name]
    scale = (H * W) / (768 * 1152)
    # the encoder region on its own, one stream (as the headline's third roofline pass)
    eng = net.engine
    side = eng.use_side_stream
    eng.use_side_stream = False
    eng.region_marks = []
    nroof = min(steps, 3)
    for _ in range(nroof):
        step(x, y)
    torch.cuda.synchronize()
    ev = eng.region_marks
    eng.region_marks = None
    eng.use_side_stream = side
    per = [dict(ev[i:i + 4]) for i in range(0, len(ev) - len(ev) % 4, 4)]
    t_fwd = sum(m["fwd_begin"].elapsed_time(m["fwd_enc_end"]) for m in per) / len(per)
    t_bwd = sum(m["bwd_enc_begin"].elapsed_time(m["bwd_end"]) for m in per) / len(per)
    enc_flop = ENCODER_FLOP_PER_SAMPLE * B * scale
    out = {"config": {"workload": f"DeepLabV3+/Xception train step (fwd + weighted CE + bwd + {optimizer}), {H}x{W}x16, local_batch={B}, "
                                  f"{dtype_name} activations / fp32 master weights, random-init seed 333",
                      "local_batch": B, "optimizer": optimizer},
           "dtype": dtype_name, "value": round(sps, 3), "unit": "samples/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "step_ms_median": round(per_step[len(per_step) // 2], 3),
           "loss_last_step": round(loss, 6),
           "whole_step_frac": round(sps * FLOP_PER_SAMPLE * scale / peak, 4),
           "encoder_region": {"fwd_ms": round(t_fwd, 3), "bwd_ms": round(t_bwd, 3),
                              "achieved": round(enc_flop / ((t_fwd + t_bwd) * 1e-3) / 1e12, 2),
                              "frac": round(enc_flop / ((t_fwd + t_bwd) * 1e-3) / peak, 4)}}
    del step, net, opt, eng
    gc.collect()                  # (an engine sits in reference cycles: its launch closures point back at it)
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--local_batch_size", type=int, default=8, help="per-GPU batch; 8 = BASELINE configs[4] (global 64 on 8 GPUs)")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--optimizer", choices=["Adam", "AdamW", "LAMB"], default="LAMB")
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--width", type=int, default=1152)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_also", action="store_true", help="skip the two other single-GPU configurations of BASELINE.json (local batch 2 fp32 Adam, "
                    "local batch 4 bf16) that a default N=1 run times after the headline and attaches as `also`")
    ap.add_argument("--torch_gpu_baseline", choices=["nchw", "nchw_tuned", "nhwc", "nhwc_tuned"], default=None, help="also time the oracle's step through PyTorch's own operators on this GPU "
                    "(eager, bf16 autocast, MIOpen / rocBLAS): the reference's stack on the same hardware, reported as `torch_rocm_baseline`")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (default: eager launches, which "
                    "measured faster once weight gradients moved to a side stream: 55.3 vs 59.5 ms at B=8)")
    ap.add_argument("--program", action="store_true", help="replay the step from the C-side launch list (TrainStep.enable_program: dc_program_run, "
                    "one call per step, no Python between launches).  The GPU work is the same list of launches.  With more than one rank it "
                    "needs the library's own collective (DC_GRAD_COLLECTIVE=lib: dc_grad_allreduce_enqueue / _wait over RCCL)")
    a = ap.parse_args()

    # timing-experiment switches that make backward skip work must never produce a bench line
    skipping = {k: v for k, v in os.environ.items() if k.startswith("DC_DEBUG_") and v not in ("", "0")}
    if skipping:
        raise SystemExit(f"bench.py refuses to run with work-skipping debug switches set: {skipping}")
    switches = {k: v for k, v in sorted(os.environ.items()) if k == "DEEPCAM_HIP_OPTIONS" or k == "DEEPCAM_HIP_LIB" or k.startswith("DC_")}

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DC_DIST_BACKEND", "nccl")      # "nccl" is RCCL; gloo only for single-GPU testing of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mlperf_deepcam_amd import lib as L, nn as dnn
    from mlperf_deepcam_amd import dist as ddist

    B, H, W = a.local_batch_size, a.height, a.width
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    net = dnn.DeepLabv3_plus(16, 3, os=16, _print=False, dtype=dtype, seed=333)
    net.materialize(B, H, W)
    net.train()
    wd = 1e-2 if a.optimizer != "Adam" else 1e-6
    opt = dnn.make_optimizer(a.optimizer, net, 1e-3, 1e-8, wd)
    step = dnn.TrainStep(net, opt, dnn.class_weights(-0.125), B, H, W, with_metrics=False)
    reducer = None
    if world > 1:
        reducer = ddist.GradReducer(net.engine, world)
        reducer.broadcast_parameters()
        step.attach_reducer(reducer)
    x, y = synthetic_batch(B, H, W, 1234 + rank, dev)
    # N > 1: what the collective backend saw and how long the step waited for it.  The wait of reducer.finish() (the compute stream
    # stalls until the outstanding bucket all-reduces are done, then the optimizer runs) is bracketed by two events on that stream in
    # every timed step: with no kernel between them their distance IS the exposed communication time.
    comm_events = []
    if reducer is not None:
        inner_finish = step.after_backward

        def timed_finish():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            inner_finish()
            e1.record()
            comm_events.append((e0, e1))
        step.after_backward = timed_finish
    graphed = False
    if world == 1 and a.graph:
        step(x, y)
        step.enable_graph()
        graphed = True
    programmed = False
    if a.program and (graphed or not (world == 1 or reducer.collective == "library")):
        raise SystemExit("--program cannot be honoured: " + ("--graph replays the step already" if graphed else
                         "with more than one rank the launch list needs the library's collective (DC_GRAD_COLLECTIVE=lib)"))
    if a.program:
        step(x, y)
        step.enable_program()
        programmed = True

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(x, y)
    # the host enqueues ~660 launches per step through ctypes, which leaves garbage behind; as `timeit` does, the collector is off inside the
    # timed region (nothing of the step is skipped).  It is not what makes one step of twenty 1 - 6 ms longer at local batch 2 on some boxes
    # (the lines read the same with it on): that is the host between launches, and `--program` takes the host out of the loop
    import gc
    gc.collect()
    gc.disable()
    try:
        barrier()
        comm_events.clear()
        # per-step HIP events on the launch stream (SURVEY 8d asks for the median of per-step times beside the mean): recording an event
        # costs no synchronisation, the timed region stays K steps between two barriers
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        t0 = time.perf_counter()
        for i in range(a.steps):
            marks[i].record()
            step(x, y)
        marks[a.steps].record()
        barrier()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    step_ms = {"median": round(per_step[len(per_step) // 2] if len(per_step) % 2 else
                               0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2]), 3),
               "min": round(per_step[0], 3), "max": round(per_step[-1], 3),
               "what": "HIP-event time of each of the K timed steps on rank 0's launch stream"}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = step.loss()
    sps = a.steps * B * world / dt
    comm = None
    if reducer is not None:
        waits = sorted(e0.elapsed_time(e1) for e0, e1 in comm_events[:a.steps])
        step.after_backward = inner_finish               # the roofline passes below run without the brackets
        comm = {"backend": dist.get_backend(), "world": dist.get_world_size(), "payload": reducer.payload,
                "collective": reducer.collective if reducer.collective == "torch" else f"library ({reducer.comm.info()['transport']})",
                "buckets": len(reducer.buckets), "bytes_per_step_per_rank": int(net.engine.layout.n_params * (2 if reducer.payload == "bf16" else 4)),
                "buckets_launched_per_step": reducer.launched_last if hasattr(reducer, "launched_last") else None,
                "exposed_wait_ms": {"median": round(waits[len(waits) // 2], 3), "max": round(waits[-1], 3)} if waits else None,
                "what": "exposed_wait_ms = rank 0's compute stream stalled in reducer.finish() (bucket all-reduces still running when "
                        "backward had finished), HIP events around the wait, per timed step"}

    # ---- roofline of the dominant kernel family, timed per launch with HIP events on the launch stream ----------------
    roof = None
    # every rank runs these extra steps (they contain the gradient collectives); only rank 0 brackets its launches with events
    step.graphed = False                         # per-launch events need eager launches (same kernels, same streams)
    step.programmed = False
    nroof = min(a.steps, 3)
    def timed_pass():
        if rank == 0:
            with KernelTimer(L) as kt:
                for _ in range(nroof):
                    step(x, y)
                return kt.result()
        for _ in range(nroof):
            step(x, y)
        return None

    fams = timed_pass()                          # as in the timed region: weight gradients overlap on the side stream
    barrier()
    eng = net.engine
    side = eng.use_side_stream
    eng.use_side_stream = False                  # second pass: every kernel alone on the GPU (standalone kernel efficiency)
    fams_alone = timed_pass()
    # third pass, same serial mode: the Xception encoder region on its own (forward up to the ASPP, backward from the point where
    # the gradient re-enters the encoder), ALL its kernels: convs, depthwise, BatchNorm.  north_star quotes its roofline target
    # (40 % of the bf16 MFMA peak) on this region.
    enc = None
    if rank == 0:
        eng.region_marks = []
    for _ in range(nroof):
        step(x, y)
    if rank == 0:
        torch.cuda.synchronize()
        ev = eng.region_marks
        eng.region_marks = None
        per = [dict(ev[i:i + 4]) for i in range(0, len(ev) - len(ev) % 4, 4)]
        t_fwd = sum(m["fwd_begin"].elapsed_time(m["fwd_enc_end"]) for m in per) / len(per)
        t_bwd = sum(m["bwd_enc_begin"].elapsed_time(m["bwd_end"]) for m in per) / len(per)
        enc = (t_fwd, t_bwd)
    eng.use_side_stream = side
    step.graphed = graphed
    step.programmed = programmed
    barrier()
    if rank == 0:
        peak = PEAK[a.dtype]
        names = {"igemm": f"dc::igemm256_kernel + dc::igemm256p_kernel + dc::pw224_kernel + dc::pw384_kernel + dc::pw192_kernel + dc::igemm_kernel<{a.dtype}> (dense conv "
                          "forward + data gradient: gather-form implicit GEMM; 256x256 (one tile per workgroup or persistent) / 224x384 / 256x384 / 128x192 "
                          "eight-wave or 128x128 four-wave tile per layer)",
                 "wgrad": f"dc::wgrad384_kernel + dc::wgrad_dma_kernel<{a.dtype}> + dc::fold_kernel (dense conv weight gradient: "
                          "split-K partial sums on 256x384 tiles of the [tap][ci] axis (pointwise and stride-1 3x3 layers), 128x128 tiles (the rest), "
                          "then the fixed-order fold of the slabs)"}
        dom = max(fams, key=lambda f: fams[f][1])
        def entry(f, src=None):
            fl, secs, n, nbytes = (src or fams)[f]
            ach = fl / secs if secs > 0 else 0.0
            return {"kernel": names[f], "achieved": round(ach / 1e12, 2), "frac": round(ach / peak, 4), "launches_timed": n,
                    "avg_launch_us": round(secs / max(n, 1) * 1e6, 2), "ms_per_step": round(secs / min(a.steps, 3) * 1e3, 3),
                    "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 2), "algorithmic_bytes_per_launch": round(nbytes / max(n, 1))}
        e = entry(dom)
        # HBM bytes per launch come from PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs of this very command) that
        # cannot be taken from inside the process; the newest committed measurement of this configuration is attached and named.
        traffic, traffic_source = None, None
        try:
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
                with open(path) as f:
                    rec = json.load(f)
                cfg = rec.get("config", {"local_batch": 8, "dtype": "bf16", "height": 768, "width": 1152})
                if (cfg.get("local_batch"), cfg.get("dtype"), cfg.get("height"), cfg.get("width")) == (B, a.dtype, H, W):
                    traffic = round(rec["kernels"][dom]["hbm_bytes_per_launch"])
                    traffic_source = (f"profiles/{os.path.basename(path)}: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                      "this command (FETCH doubled per the gfx950 guide); not measured inside this run")
                    break
        except Exception:
            traffic, traffic_source = None, None
        roof = {"bound": "mfma", "achieved": e["achieved"], "peak": peak / 1e12, "unit": "TFLOP/s", "frac": e["frac"], "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": e["kernel"], "launches_timed": e["launches_timed"], "avg_launch_us": e["avg_launch_us"],
                "ms_per_step": e["ms_per_step"], "algorithmic_gflop_per_launch": e["algorithmic_gflop_per_launch"],
                "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"], "other_mfma_kernel": entry([f for f in fams if f != dom][0]),
                "note": "achieved/frac are measured as in the timed region, i.e. while weight-gradient kernels run concurrently on the "
                        "side stream; 'standalone' repeats the measurement with one kernel on the GPU at a time",
                "standalone": {f: {k: v for k, v in entry(f, fams_alone).items() if k != "kernel"} for f in fams_alone},
                "whole_step_frac": round(sps / world * FLOP_PER_SAMPLE * (H * W) / (768 * 1152) / peak, 4)}
        if enc is not None:
            enc_flop = ENCODER_FLOP_PER_SAMPLE * B * (H * W) / (768 * 1152)
            roof["encoder_region"] = {
                "what": "Xception encoder forward + backward, every kernel of the region (convs, depthwise, BatchNorm), one stream",
                "fwd_ms": round(enc[0], 3), "bwd_ms": round(enc[1], 3), "gflop_per_sample": ENCODER_FLOP_PER_SAMPLE / 1e9,
                "achieved": round(enc_flop / ((enc[0] + enc[1]) * 1e-3) / 1e12, 2), "frac": round(enc_flop / ((enc[0] + enc[1]) * 1e-3) / peak, 4)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {"metric": "samples/sec (768x1152x16) train step", "value": round(sps, 3), "unit": "samples/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "step_ms": step_ms, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"DeepLabV3+/Xception train step (fwd + weighted CE + bwd + {a.optimizer}"
                                      f"{' + RCCL grad all-reduce' if world > 1 else ''}), {H}x{W}x16, local_batch={B}, "
                                      f"{a.dtype} activations / fp32 master weights, random-init seed 333",
                          "local_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "optimizer": a.optimizer,
                          "hip_graph": graphed, "launch_list_replay": programmed, "switches": switches},
               "loss_last_step": round(loss, 6), "roofline": roof}
        if comm is not None:
            out["comm"] = comm
        if world == 1:
            del step, net, opt, eng       # the engine's arenas go back to the allocator before the other configurations / baselines
            gc.collect()
            torch.cuda.empty_cache()
        if world == 1 and not a.no_also and (B, a.dtype) == (8, "bf16"):
            # BASELINE.json configs[1] and configs[2]: the other two single-GPU configurations, timed the same way in this process
            out["also"] = []
            # ... and the per-rank shape of configs[3] (8 x MI355X, global batch 16, bf16): local batch 2 in bf16
            for ab, adt, aopt in ((2, "fp32", "Adam"), (4, "bf16", a.optimizer), (2, "bf16", a.optimizer)):
                try:
                    out["also"].append(also_config(dnn, ab, H, W, adt, aopt, a.steps, a.warmup, dev))
                except Exception as e:  # never take the headline down
                    out["also"].append({"config": {"local_batch": ab}, "dtype": adt, "value": None, "error": f"{type(e).__name__}: {e}"})
        if world == 1 and a.torch_gpu_baseline:
            try:
                out["torch_rocm_baseline"] = torch_gpu_baseline(H, W, B, dev, a.torch_gpu_baseline)
                out["torch_rocm_baseline"]["this_repository_over_it"] = round(sps / out["torch_rocm_baseline"]["value"], 2)
            except Exception as e:  # a baseline must never take the GPU number down with it
                out["torch_rocm_baseline"] = {"value": None, "unit": "samples/s", "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(H, W)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        reducer.close()               # the library path's communicator (ncclComm_t, stream, events) goes back before the process group does
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
