"""train_hdf5_ddp.py-compatible driver for the MI355X engine.

Same command line as the reference (src/deepCam/train_hdf5_ddp.py:548-578, flag for flag), same step order (:345-371),
same logging-frequency reductions (:398-414), validation averaging (:423-512: mean of per-sample IoUs at batch 1),
checkpoint dictionary (:515-527: step, epoch, model with 'module.'-prefixed keys, optimizer) and ``:::MLLOG`` event keys
(utils/mlperf_log_utils.py).  Out of scope by SURVEY section 8: W&B and Basemap plots (flags accepted, ignored).
Input: data.InputPipeline over the HDF5 files (decoded with h5py, or through the HDF5 C library when h5py is not installed:
h5lite), or over synthetic HWC fields with ``--synthetic_samples N``.

Extra flags (not in the reference): --wireup_method env|single, --dtype, --synthetic_samples, --synthetic_learnable,
--height/--width, --max_steps.
"""
from __future__ import annotations

import argparse as ap
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import data as ddata
from . import dist as comm
from . import nn as dnn


class StoreDictKeyPair(ap.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        d = {}
        for kv in values.split(","):
            k, v = kv.split("=")
            d[k] = v.strip('"')
        setattr(namespace, self.dest, d)


class MLLogger:
    """Writes the mlperf_logging line format directly (the package is not a dependency here): rank 0 only, optional
    barrier before the event (mlperf_log_utils.py:92-105)."""

    def __init__(self, filename, benchmark="deepcam", org="MI355X-native"):
        self.rank = comm.get_rank()
        self.f = None
        if self.rank == 0:
            os.makedirs(os.path.dirname(filename), exist_ok=True)
            self.f = open(filename, "a")
        self.barrier()
        for k, v in (("submission_benchmark", benchmark), ("submission_org", org), ("submission_division", "closed"),
                     ("submission_status", "onprem"), ("submission_platform", "1xMI355X")):
            self.log_event(k, v)

    @staticmethod
    def barrier():
        if dist.is_available() and dist.is_initialized():
            dist.barrier()

    def _emit(self, etype, key, value, metadata, sync):
        if sync:
            self.barrier()
        if self.rank != 0:
            return
        rec = {"namespace": "", "time_ms": int(time.time() * 1000), "event_type": etype, "key": key, "value": value,
               "metadata": metadata or {}}
        line = ":::MLLOG " + json.dumps(rec)
        print(line, flush=True)
        self.f.write(line + "\n")
        self.f.flush()

    def log_start(self, key, value=None, metadata=None, sync=False):
        self._emit("INTERVAL_START", key, value, metadata, sync)

    def log_end(self, key, value=None, metadata=None, sync=False):
        self._emit("INTERVAL_END", key, value, metadata, sync)

    def log_event(self, key, value=None, metadata=None, sync=False):
        self._emit("POINT_IN_TIME", key, value, metadata, sync)


def amp_state_dict(opt_level: str = "O0"):
    """The 'amp' entry of the reference's checkpoint (train_hdf5_ddp.py:524-525 writes apex's amp.state_dict(), :238-239 feeds it
    back through amp.load_state_dict, which raises KeyError without it).  apex's format as published (apex/amp/frontend.py,
    state_dict / load_state_dict): one entry per loss scaler, 'loss_scaler%d' -> {'loss_scale': float, 'unskipped': int}.
    bf16 here needs no loss scaling, so the entry carries what a fresh apex run starts from: the dynamic scaler's initial 2**16
    for O1/O2/O3 and the static 1.0 of O0 -- a reference + apex run resuming from this file then behaves like a fresh scaler."""
    from collections import OrderedDict
    scale = 1.0 if opt_level == "O0" else 2.0 ** 16
    return OrderedDict(loss_scaler0={"loss_scale": scale, "unskipped": 0})


def check_amp_state(state) -> None:
    """Accept (and ignore) the 'amp' entry of a checkpoint: loss-scaler state has no counterpart in the bf16 / fp32 engine.
    A malformed entry is an error, a missing one is fine (reference runs without apex do not write it)."""
    if state is None:
        return
    for k, v in state.items():
        if not (k.startswith("loss_scaler") and "loss_scale" in v and "unskipped" in v):
            raise ValueError(f"checkpoint['amp'][{k!r}] is not an apex loss-scaler entry")


def build_parser():
    AP = ap.ArgumentParser()
    AP.add_argument("--wireup_method", type=str, default="nccl-openmpi",
                    choices=["nccl-openmpi", "nccl-slurm", "nccl-slurm-pmi", "mpi", "env", "single"], help="Specify what is used for wiring up the ranks")
    AP.add_argument("--wandb_certdir", type=str, default="/opt/certs", help="accepted for compatibility; W&B is out of scope")
    AP.add_argument("--run_tag", type=str, help="Unique run tag, to allow for better identification")
    AP.add_argument("--output_dir", type=str, help="Directory used for storing output. Needs to read/writeable from rank 0")
    AP.add_argument("--checkpoint", type=str, default=None, help="Checkpoint file to restart training from.")
    AP.add_argument("--data_dir_prefix", type=str, default="/", help="prefix to data dir")
    AP.add_argument("--max_inter_threads", type=int, default=1, help="Maximum number of concurrent readers")
    AP.add_argument("--max_epochs", type=int, default=30, help="Maximum number of epochs to train")
    AP.add_argument("--save_frequency", type=int, default=100, help="Frequency with which the model is saved in number of steps")
    AP.add_argument("--validation_frequency", type=int, default=100, help="Frequency with which the model is validated")
    AP.add_argument("--max_validation_steps", type=int, default=None, help="Number of validation steps to perform (invalidates a submission)")
    AP.add_argument("--logging_frequency", type=int, default=100, help="Frequency with which the training progress is logged")
    AP.add_argument("--training_visualization_frequency", type=int, default=50, help="accepted; plotting is out of scope (0 = off)")
    AP.add_argument("--validation_visualization_frequency", type=int, default=50, help="accepted; plotting is out of scope (0 = off)")
    AP.add_argument("--local_batch_size", type=int, default=1, help="Number of samples per local minibatch")
    AP.add_argument("--channels", type=int, nargs="+", default=list(range(16)), help="Channels used in input")
    AP.add_argument("--optimizer", type=str, default="Adam", choices=["Adam", "AdamW", "LAMB"], help="Optimizer to use")
    AP.add_argument("--start_lr", type=float, default=1e-3, help="Start LR")
    AP.add_argument("--adam_eps", type=float, default=1e-8, help="Adam Epsilon")
    AP.add_argument("--weight_decay", type=float, default=1e-6, help="Weight decay")
    AP.add_argument("--loss_weight_pow", type=float, default=-0.125, help="Decay factor to adjust the weights")
    AP.add_argument("--lr_warmup_steps", type=int, default=0, help="Number of steps for linear LR warmup")
    AP.add_argument("--lr_warmup_factor", type=float, default=1.0, help="Multiplier for linear LR warmup")
    AP.add_argument("--lr_schedule", action=StoreDictKeyPair)
    AP.add_argument("--target_iou", type=float, default=0.82, help="Target IoU score.")
    AP.add_argument("--model_prefix", type=str, default="model", help="Prefix for the stored model")
    AP.add_argument("--amp_opt_level", type=str, default="O0", help="O0 = fp32 activations, O1/O2 = bf16 activations (no loss scaling needed)")
    AP.add_argument("--enable_wandb", action="store_true")
    AP.add_argument("--resume_logging", action="store_true")
    # extensions
    AP.add_argument("--dtype", type=str, default=None, choices=["fp32", "bf16"], help="overrides --amp_opt_level")
    AP.add_argument("--synthetic_samples", type=int, default=0, help="train on this many synthetic samples instead of HDF5 files")
    AP.add_argument("--synthetic_learnable", action="store_true", help="synthetic labels that are a function of the fields (convergence / time-to-target runs)")
    AP.add_argument("--height", type=int, default=768)
    AP.add_argument("--width", type=int, default=1152)
    AP.add_argument("--max_steps", type=int, default=None)
    return AP


def main(pargs):
    comm.init(pargs.wireup_method)
    rank, local_rank, size = comm.get_rank(), comm.get_local_rank(), comm.get_size()
    pargs.logging_frequency = max([pargs.logging_frequency, 1])                               # train_hdf5_ddp.py:106
    log_file = os.path.normpath(os.path.join(pargs.output_dir, "logs", pargs.run_tag + ".log"))
    logger = MLLogger(log_file)
    logger.log_start(key="init_start", sync=True)
    logger.log_event(key="cache_clear")
    seed = 333
    logger.log_event(key="seed", value=seed)
    torch.manual_seed(seed)
    if not torch.cuda.is_available():
        raise RuntimeError("mlperf_deepcam_amd.train needs an MI355X: there is no CPU path")
    device = torch.device("cuda", local_rank)
    torch.cuda.manual_seed(seed)
    torch.cuda.set_device(device)
    if rank == 0:
        os.makedirs(pargs.output_dir, exist_ok=True)

    logger.log_event(key="global_batch_size", value=pargs.local_batch_size * size)
    logger.log_event(key="opt_name", value=pargs.optimizer)
    logger.log_event(key="opt_base_learning_rate", value=pargs.start_lr * pargs.lr_warmup_factor)
    logger.log_event(key="opt_learning_rate_warmup_steps", value=pargs.lr_warmup_steps)
    logger.log_event(key="opt_learning_rate_warmup_factor", value=pargs.lr_warmup_factor)
    logger.log_event(key="opt_epsilon", value=pargs.adam_eps)

    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[pargs.dtype] if pargs.dtype else \
        (torch.float32 if pargs.amp_opt_level == "O0" else torch.bfloat16)
    H, W, B = pargs.height, pargs.width, pargs.local_batch_size
    net = dnn.DeepLabv3_plus(n_input=len(pargs.channels), n_classes=3, os=16, pretrained=False, rank=rank, dtype=dtype)
    net.to(device)
    net.materialize(B, H, W)
    class_weights = dnn.class_weights(pargs.loss_weight_pow)
    optimizer = dnn.make_optimizer(pargs.optimizer, net, pargs.start_lr, pargs.adam_eps, pargs.weight_decay)
    ddp = comm.DistributedDataParallel(net)

    if pargs.checkpoint:
        checkpoint = torch.load(pargs.checkpoint, map_location=device, weights_only=False)
        start_step, start_epoch = checkpoint["step"], checkpoint["epoch"]
        optimizer.load_state_dict(checkpoint["optimizer"])
        ddp.load_state_dict(checkpoint["model"])
        check_amp_state(checkpoint.get("amp"))          # written by a reference + apex run (train_hdf5_ddp.py:524-525): nothing to restore
    else:
        start_step, start_epoch = 0, 0

    scheduler = None
    if pargs.lr_schedule:
        scheduler_after = dnn.get_lr_schedule(pargs.start_lr, pargs.lr_schedule, optimizer, last_step=start_step)
        scheduler = scheduler_after
        if pargs.lr_warmup_steps > 0:
            scheduler = dnn.GradualWarmupScheduler(optimizer, multiplier=pargs.lr_warmup_factor, total_epoch=pargs.lr_warmup_steps,
                                                   after_scheduler=scheduler_after)
    if size > 1:
        steptens = torch.tensor(np.array([start_step, start_epoch]), requires_grad=False).to(device)
        dist.broadcast(steptens, src=0)
        start_step, start_epoch = int(steptens[0]), int(steptens[1])

    # ---- data feeder (train_hdf5_ddp.py:274-306): HDF5 files when h5py and a data directory exist, else synthetic HWC fields.
    #      Both go through the same pipeline: pinned staging -> async copy stream -> fused normalise kernel -> NHWC activations.
    root_dir = os.path.join(pargs.data_dir_prefix)
    if pargs.synthetic_samples > 0:
        train_set = ddata.SyntheticHWC(pargs.synthetic_samples, H, W, channels=pargs.channels, allow_uneven_distribution=False,
                                       shuffle=True, comm_size=size, comm_rank=rank, learnable=pargs.synthetic_learnable)
        n_val = max(size, pargs.synthetic_samples // 8)
        validation_set = ddata.SyntheticHWC(n_val, H, W, channels=pargs.channels, allow_uneven_distribution=True,
                                            shuffle=(pargs.max_validation_steps is not None), comm_size=size, comm_rank=rank, seed=54321,
                                            learnable=pargs.synthetic_learnable)
    else:
        train_set = ddata.CamDataset(os.path.join(root_dir, "train"), os.path.join(root_dir, "stats.h5"), pargs.channels,
                                     allow_uneven_distribution=False, shuffle=True, preprocess=True, comm_size=size, comm_rank=rank)
        validation_set = ddata.CamDataset(os.path.join(root_dir, "validation"), os.path.join(root_dir, "stats.h5"), pargs.channels,
                                          allow_uneven_distribution=True, shuffle=(pargs.max_validation_steps is not None),
                                          preprocess=True, comm_size=size, comm_rank=rank)
        assert tuple(train_set.data_shape[:2]) == (H, W), "pass --height/--width matching the files"
    # channel counts the MFMA stem does not take go through the direct stem kernel, which reads the reference's NCHW fp32 batch
    layout = "nhwc" if net.engine.x0 is not None else "nchw"
    train_loader = ddata.InputPipeline(train_set, B, dtype=dtype, device=device, layout=layout)
    validation_loader = ddata.InputPipeline(validation_set, 1, dtype=dtype, device=device, layout=layout)
    logger.log_event(key="train_samples", value=train_set.global_size)
    val_size = validation_set.global_size if pargs.max_validation_steps is None else \
        min([validation_set.global_size, pargs.max_validation_steps * B * size])
    logger.log_event(key="eval_samples", value=val_size)
    if pargs.max_validation_steps is not None:
        logger.log_event(key="invalid_submission")

    train_step = dnn.TrainStep(net, optimizer, class_weights, B, H, W, with_metrics=True)
    if ddp.reducer is not None:
        train_step.attach_reducer(ddp.reducer)
    step, epoch = start_step, start_epoch
    current_lr = pargs.start_lr if scheduler is None else scheduler.get_last_lr()[0]
    stop_training = False
    net.train()
    logger.log_end(key="init_stop", sync=True)
    logger.log_start(key="run_start", sync=True)

    while True:
        logger.log_start(key="epoch_start", metadata={"epoch_num": epoch + 1, "step_num": step}, sync=True)
        for inputs, label, filename in train_loader:
            train_step(inputs, label)                          # forward, loss, backward (+all-reduce), optimizer.step
            step += 1
            if scheduler is not None:
                current_lr = scheduler.get_last_lr()[0]
                scheduler.step()
            if step % pargs.logging_frequency == 0:
                loss_now = train_step.loss()
                # the fused loss kernel turns a label outside [0, 3) into NaN (the reference's CrossEntropyLoss raises on it,
                # losses.py:50); one such step poisons every weight, so stop instead of training on NaNs.  The decision is
                # collective: only the rank that saw the corrupt label has a NaN at this step, and a rank that raised alone would leave
                # the others blocked in the reduction below.
                vals = torch.tensor([loss_now, train_step.iou(), 0.0 if math.isfinite(loss_now) else 1.0], dtype=torch.float32, device=device)
                if size > 1:
                    dist.all_reduce(vals, op=dist.ReduceOp.SUM)
                if float(vals[2]) > 0 or not bool(torch.isfinite(vals[0])):
                    raise RuntimeError(f"non-finite training loss at step {step} on {int(vals[2])} rank(s) (local value {loss_now}): "
                                       "corrupt label or diverged run")
                loss_avg_train, iou_avg_train = (vals[:2] / float(size)).tolist()
                md = {"epoch_num": epoch + 1, "step_num": step}
                logger.log_event(key="learning_rate", value=current_lr, metadata=md)
                logger.log_event(key="train_accuracy", value=iou_avg_train, metadata=md)
                logger.log_event(key="train_loss", value=loss_avg_train, metadata=md)
            if step % pargs.validation_frequency == 0:
                logger.log_start(key="eval_start", metadata={"epoch_num": epoch + 1})
                net.eval()
                sums = torch.zeros(3, dtype=torch.float64, device=device)            # count, loss, iou
                with torch.no_grad():
                    step_val = 0
                    for inputs_val, label_val, _ in validation_loader:
                        outputs_val = net.engine_for(shape=(1, len(pargs.channels), H, W)).forward(inputs_val, train=False)
                        counts = torch.zeros(9, dtype=torch.int64, device=device)
                        ls = dnn.wce_fused(outputs_val, label_val, class_weights, counts=counts)
                        sums[0] += 1.0
                        sums[1] += ls[0] / label_val.numel()
                        sums[2] += dnn.iou_from_counts(counts.cpu().tolist())          # per-sample IoU, then averaged
                        step_val += 1
                        if pargs.max_validation_steps is not None and step_val > pargs.max_validation_steps:
                            break
                if size > 1:
                    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
                loss_avg_val, iou_avg_val = float(sums[1] / sums[0]), float(sums[2] / sums[0])
                md = {"epoch_num": epoch + 1, "step_num": step}
                logger.log_event(key="eval_accuracy", value=iou_avg_val, metadata=md)
                logger.log_event(key="eval_loss", value=loss_avg_val, metadata=md)
                if iou_avg_val >= pargs.target_iou:
                    logger.log_event(key="target_accuracy_reached", value=pargs.target_iou, metadata=md)
                    stop_training = True
                net.train()
                logger.log_end(key="eval_stop", metadata={"epoch_num": epoch + 1})
            if pargs.save_frequency > 0 and step % pargs.save_frequency == 0:
                md = {"epoch_num": epoch + 1, "step_num": step}
                logger.log_start(key="save_start", metadata=md, sync=True)
                # never write a checkpoint of poisoned weights (the loss check above runs every logging_frequency steps only): same
                # collective decision, on this step's loss
                bad = torch.tensor([0.0 if math.isfinite(train_step.loss()) else 1.0], dtype=torch.float32, device=device)
                if size > 1:
                    dist.all_reduce(bad, op=dist.ReduceOp.SUM)
                if float(bad[0]) > 0:
                    raise RuntimeError(f"non-finite training loss at step {step} on {int(bad[0])} rank(s): checkpoint not written")
                if rank == 0:
                    checkpoint = {"step": step, "epoch": epoch, "model": ddp.state_dict(), "optimizer": optimizer.state_dict(),
                                  "amp": amp_state_dict(pargs.amp_opt_level)}
                    torch.save(checkpoint, os.path.join(pargs.output_dir, pargs.model_prefix + "_step_" + str(step) + ".cpt"))
                logger.log_end(key="save_stop", metadata=md, sync=True)
            if pargs.max_steps is not None and step >= pargs.max_steps:
                stop_training = True
            if stop_training:
                break
        logger.log_end(key="epoch_stop", metadata={"epoch_num": epoch + 1, "step_num": step}, sync=True)
        epoch += 1
        if epoch >= pargs.max_epochs or stop_training:
            break
    logger.log_end(key="run_stop", sync=True, metadata={"status": "success"})
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main(build_parser().parse_args())
