"""Input pipeline: files -> pinned host buffers -> HBM (async copy stream) -> normalise + channel-select -> NHWC activations.

SURVEY section 8f row 1, BASELINE configs[4] ("NHWC HDF5 -> pinned-host -> HBM async pipeline").  Reference:
data/cam_hdf5_dataset.py (file listing, shuffle, sharding :36-65,71-83; shapes :86-90; min/max statistics :93-102; sample
read + normalisation :115-131) and the DataLoader set-up at train_hdf5_ddp.py:277-306.

What is different by design: the CAM5 files store (768, 1152, 16) fp32 fields, i.e. channels-LAST.  The reference transposes
to CHW and normalises in numpy on the host (:126-129) and then pays a layout conversion again on the device.  Here the raw
HWC sample is copied as it lies in the file into a pinned staging buffer, DMA'd to HBM on a copy stream, and ONE kernel
(dc_input_normalize_hwc) selects the channels, applies scale*(x-shift) and writes the NHWC activation tensor the stem
consumes: the training step then contains no layout pass at all.  Reading is double-buffered behind the compute stream.

h5py is not available in every image: CamDataset decodes with h5py when it is installed and otherwise through `h5lite`, a ctypes
binding of the HDF5 C library (contiguous fp32 payloads are pread() straight into the pinned staging buffer, several files in
parallel).  The tests write small CAM5-shaped files with h5lite and read them back through the whole pipeline.
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L


# ---------------------------------------------------------------------------------------------------------------------
# sharding (host logic, no device needed)
# ---------------------------------------------------------------------------------------------------------------------
def shard_files(all_files: Sequence[str], comm_size: int, comm_rank: int, allow_uneven_distribution: bool, shuffle: bool,
                seed: int = 12345) -> Tuple[List[str], int]:
    """(this rank's files, global_size), exactly as CamDataset.init_reader (cam_hdf5_dataset.py:38-65): sorted list,
    optionally shuffled ONCE with RandomState(seed) (never reshuffled per epoch), contiguous slice per rank; even mode drops
    the remainder and reports global_size = comm_size * local, uneven mode gives the remainder to the last rank."""
    files = sorted(all_files)
    if shuffle:
        np.random.RandomState(seed).shuffle(files)
    global_size = len(files)
    n = global_size // comm_size
    start = comm_rank * n
    if allow_uneven_distribution:
        end = start + n if comm_rank != comm_size - 1 else global_size
        mine = files[start:end]
    else:
        mine = files[start:start + n]
        global_size = comm_size * len(mine)
    return list(mine), global_size


def _h5_backend():
    """h5py when it is installed, else the ctypes binding of libhdf5 (h5lite); RuntimeError when neither is usable."""
    try:
        import h5py
        return "h5py", h5py
    except ImportError:
        from . import h5lite
        if h5lite.available():
            return "h5lite", h5lite
    raise RuntimeError("CamDataset needs h5py or an HDF5 C library (libhdf5.so; set DEEPCAM_HDF5_LIB) to decode the CAM5 files; "
                       "neither was found (data.SyntheticHWC runs without)")


class CamDataset:
    """HDF5-backed source (cam_hdf5_dataset.py:36-131).  read_into(i, data, label) fills data float32 [H, W, Cfile] exactly as
    stored and label int64 [H, W]; channel selection and normalisation happen on the device."""

    def __init__(self, source: str, statsfile: str, channels: Sequence[int], allow_uneven_distribution: bool = False,
                 shuffle: bool = False, preprocess: bool = True, comm_size: int = 1, comm_rank: int = 0, seed: int = 12345):
        self.backend, self._h5 = _h5_backend()
        self.channels = list(channels)
        all_files = [os.path.join(source, x) for x in os.listdir(source) if x.endswith(".h5")]
        self.files, self.global_size = shard_files(all_files, comm_size, comm_rank, allow_uneven_distribution, shuffle, seed)
        self.local_size = len(self.files)
        if self.backend == "h5py":
            with self._h5.File(self.files[0], "r") as f:
                self.data_shape = tuple(f["climate"]["data"].shape)
                self.label_shape = tuple(f["climate"]["labels_0"].shape)
            with self._h5.File(statsfile, "r") as f:
                minval, maxval = f["climate"]["minval"][...], f["climate"]["maxval"][...]
        else:
            with self._h5.File(self.files[0]) as f:
                self.data_shape, self.label_shape = f.shape("climate/data"), f.shape("climate/labels_0")
                self._label_dtype = f.dtype("climate/labels_0").newbyteorder("=")
            with self._h5.File(statsfile) as f:
                minval, maxval = f.read("climate/minval"), f.read("climate/maxval")
        shift = np.asarray(minval)[self.channels]                                   # :96-98
        scale = 1.0 / (np.asarray(maxval)[self.channels] - shift)
        if not preprocess:                                                          # :100-102
            shift, scale = np.zeros_like(shift), np.ones_like(scale)
        self.data_shift = np.asarray(shift, np.float32)
        self.data_scale = np.asarray(scale, np.float32)
        if comm_rank == 0:
            print("Initialized dataset with ", self.global_size, " samples.")

    def __len__(self):
        return self.local_size

    def read_into(self, i: int, data_out: np.ndarray, label_out: np.ndarray) -> str:
        if self.backend == "h5py":
            with self._h5.File(self.files[i], "r") as f:
                f["climate/data"].read_direct(data_out)
                label_out[...] = f["climate/labels_0"][...]
        else:
            with self._h5.File(self.files[i]) as f:
                f.read_direct("climate/data", data_out)               # contiguous fp32: pread straight into the pinned buffer
                if self._label_dtype == label_out.dtype:
                    f.read_direct("climate/labels_0", label_out)
                else:
                    label_out[...] = f.read("climate/labels_0")
        return self.files[i]


class SyntheticHWC:
    """Deterministic stand-in with the dataset's on-disk layout: HWC float32 fields with per-channel offset/range (so that the
    normalisation does real work) and labels drawn with the reference's class frequencies.  Same sharding rule."""

    CLASS_FREQ = (0.986267818390377, 0.0004578708870701058, 0.01327431072255291)

    def __init__(self, global_size: int, H: int, W: int, cfile: int = 16, channels: Sequence[int] = tuple(range(16)),
                 allow_uneven_distribution: bool = False, shuffle: bool = False, comm_size: int = 1, comm_rank: int = 0,
                 seed: int = 12345, learnable: bool = False):
        # learnable: the labels are a function of the fields (block-constant structures in channels 0 and 1 decide the class), so
        # a run has something to converge on: used for the time-to-target demonstration of the driver's stop rule
        # (train_hdf5_ddp.py:505-507).  Otherwise labels are independent draws with the reference's class frequencies.
        self.learnable = learnable
        names = [f"data-synthetic-{i:06d}.h5" for i in range(global_size)]
        self.files, self.global_size = shard_files(names, comm_size, comm_rank, allow_uneven_distribution, shuffle, seed)
        self.local_size = len(self.files)
        self.H, self.W, self.cfile = H, W, cfile
        self.channels = list(channels)
        self.data_shape, self.label_shape = (H, W, cfile), (H, W)
        lo = np.linspace(-50.0, 200.0, cfile).astype(np.float32)
        hi = lo + np.linspace(1.0, 300.0, cfile).astype(np.float32)
        self._lo, self._hi = lo, hi
        self.data_shift = lo[self.channels].copy()
        self.data_scale = (1.0 / (hi[self.channels] - lo[self.channels])).astype(np.float32)

    def __len__(self):
        return self.local_size

    def read_into(self, i: int, data_out: np.ndarray, label_out: np.ndarray) -> str:
        idx = int(self.files[i].split("-")[-1].split(".")[0])
        rs = np.random.RandomState(1000003 + idx)
        u = rs.random_sample((self.H, self.W, self.cfile)).astype(np.float32)
        if self.learnable:
            bh, bw = (self.H + 7) // 8, (self.W + 7) // 8
            blk = rs.random_sample((bh, bw, 2)).astype(np.float32)
            field = np.repeat(np.repeat(blk, 8, axis=0), 8, axis=1)[:self.H, :self.W]
            u[..., :2] = 0.9 * field + 0.1 * u[..., :2]
            lab = np.zeros((self.H, self.W), np.int64)
            lab[field[..., 0] > 0.75] = 2
            lab[(field[..., 0] > 0.75) & (field[..., 1] > 0.8)] = 1
            data_out[...] = self._lo + u * (self._hi - self._lo)
            label_out[...] = lab
            return self.files[i]
        data_out[...] = self._lo + u * (self._hi - self._lo)
        label_out[...] = rs.choice(3, size=(self.H, self.W), p=np.array(self.CLASS_FREQ) / sum(self.CLASS_FREQ))
        return self.files[i]


# ---------------------------------------------------------------------------------------------------------------------
# device half
# ---------------------------------------------------------------------------------------------------------------------
class _Slot:
    def __init__(self, B, H, W, cfile, C_, dtype, device, nchw=False):
        self.data_host = torch.empty((B, H, W, cfile), dtype=torch.float32).pin_memory()
        self.label_host = torch.empty((B, H, W), dtype=torch.int64).pin_memory()
        self.data_dev = torch.empty((B, H, W, cfile), dtype=torch.float32, device=device)
        # NHWC activations, consumed in place by the MFMA stem; or the reference's NCHW fp32 batch for the direct stem kernel
        self.x = torch.empty((B, C_, H, W), dtype=torch.float32, device=device) if nchw else \
            torch.empty((B, H, W, C_), dtype=dtype, device=device)
        self.label = torch.empty((B, H, W), dtype=torch.int64, device=device)
        self.ready = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.names: List[str] = []


class InputPipeline:
    """Iterable over (x_nhwc [B,H,W,C] in the activation dtype, labels int64 [B,H,W], filenames).

    A reader thread fills pinned staging buffers (numpy/h5py release the GIL while copying); the main thread enqueues, on a
    dedicated copy stream, the H2D DMA and the normalise kernel of batch i+1 while the compute stream trains on batch i.
    `depth` slots (>= 2) bound the memory; a slot is recycled only after the compute stream has passed the point where its
    tensors were last used (the consumer calls release(), or simply asks for the next batch)."""

    def __init__(self, dataset, batch_size: int, dtype=torch.bfloat16, device=None, depth: int = 3, drop_last: bool = True,
                 workers: int = 4, layout: str = "nhwc"):
        if not torch.cuda.is_available():
            raise L.DeepcamHipError("InputPipeline needs a HIP device")
        self.ds, self.B, self.dtype = dataset, batch_size, dtype
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.dt = L.dtype_code(dtype)
        H, W, cfile = dataset.data_shape
        self.H, self.W, self.cfile = H, W, cfile
        self.C = len(dataset.channels)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # layout "nchw": batches leave as the reference's NCHW fp32 tensors (any channel count).  An engine whose channel count
        # the MFMA stem does not take (--channels subsets that are not a multiple of 8 in bf16 / 4 in fp32) reads those.
        if layout not in ("nhwc", "nchw"):
            raise ValueError(f"layout must be 'nhwc' or 'nchw', got {layout!r}")
        self.nchw = layout == "nchw"
        self.slots = [_Slot(batch_size, H, W, cfile, self.C, dtype, self.device, self.nchw) for _ in range(max(2, depth))]
        self.shift = torch.from_numpy(np.asarray(dataset.data_shift, np.float32)).to(self.device)
        self.scale = torch.from_numpy(np.asarray(dataset.data_scale, np.float32)).to(self.device)
        ident = list(dataset.channels) == list(range(cfile))
        self.channels = None if ident else torch.tensor(list(dataset.channels), dtype=torch.int32, device=self.device)
        self.nbatches = len(dataset) // batch_size if drop_last else -(-len(dataset) // batch_size)
        self._prev: Optional[_Slot] = None
        self._reader_thread: Optional[threading.Thread] = None
        # one 56.6 MB sample is ~10 ms of host copy/decoding: a single reader caps the pipeline near 100 samples/s, below the
        # train step's rate, so the samples of a batch are read by `workers` threads (numpy / h5py release the GIL while copying)
        self.workers = max(1, int(workers))

    def __len__(self):
        return self.nbatches

    def _reader(self, free_q: "queue.Queue", full_q: "queue.Queue", stop: "threading.Event"):
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(self.workers) if self.workers > 1 else None
        try:
            for b in range(self.nbatches):
                slot = free_q.get()
                if slot is None or stop.is_set():
                    return
                slot.consumed.synchronize()                       # host wait: the GPU is done with this slot's tensors
                dn, ln = slot.data_host.numpy(), slot.label_host.numpy()
                if pool is None:
                    slot.names = [self.ds.read_into(b * self.B + j, dn[j], ln[j]) for j in range(self.B)]
                else:
                    slot.names = list(pool.map(lambda j: self.ds.read_into(b * self.B + j, dn[j], ln[j]), range(self.B)))
                full_q.put(slot)
            full_q.put(None)
        except BaseException as e:  # surface reader failures in the consumer
            full_q.put(e)
        finally:
            if pool is not None:
                pool.shutdown(wait=False)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, List[str]]]:
        # An iteration abandoned half way (train.py breaks out of the validation loop at --max_validation_steps) leaves its
        # reader possibly still writing a slot's pinned buffer: it has been told to stop (finally: below); wait for it before
        # any slot is handed out again, and forget the slot the abandoned iteration had out.
        if self._reader_thread is not None:
            # (the abandoned generator's `finally` normally did this already; repeat it in case the generator object is still alive)
            self._reader_stop.set()
            self._reader_free_q.put(None)
            self._reader_thread.join()
            self._reader_thread = None
        self._prev = None
        free_q: "queue.Queue" = queue.Queue()
        full_q: "queue.Queue" = queue.Queue()
        stop = threading.Event()
        # in-flight H2D copies / normalise kernels of an abandoned iteration are on the copy stream: order the new ones and
        # the slots' reuse behind them
        torch.cuda.current_stream().wait_stream(self.copy_stream)
        for s in self.slots:
            s.consumed.record(torch.cuda.current_stream())
            free_q.put(s)
        t = threading.Thread(target=self._reader, args=(free_q, full_q, stop), daemon=True)
        self._reader_thread, self._reader_stop, self._reader_free_q = t, stop, free_q
        t.start()
        staged: List[_Slot] = []

        def stage_next() -> bool:
            item = full_q.get()
            if item is None:
                return False
            if isinstance(item, BaseException):
                raise item
            with torch.cuda.stream(self.copy_stream):
                item.data_dev.copy_(item.data_host, non_blocking=True)
                item.label.copy_(item.label_host, non_blocking=True)
                if self.nchw:
                    L.call("dc_input_normalize_hwc_to_nchw", self.B, self.H * self.W, self.cfile, self.C, L.dptr(self.channels),
                           L.dptr(item.data_dev), L.dptr(self.shift), L.dptr(self.scale), L.dptr(item.x),
                           C.c_void_p(self.copy_stream.cuda_stream))
                else:
                    L.call("dc_input_normalize_hwc", self.dt, self.B * self.H * self.W, self.cfile, self.C, L.dptr(self.channels),
                           L.dptr(item.data_dev), L.dptr(self.shift), L.dptr(self.scale), L.dptr(item.x), self.C,
                           C.c_void_p(self.copy_stream.cuda_stream))
                item.ready.record(self.copy_stream)
            staged.append(item)
            return True

        try:
            more = stage_next()
            while staged:
                # the consumer is back for another batch, so it is done with the previous one: recycle that slot FIRST (the
                # reader may be waiting for it), then put the next batch in flight, then hand out the current one
                if self._prev is not None:
                    self._release(self._prev, free_q)
                    self._prev = None
                cur = staged.pop(0)
                if more:
                    more = stage_next()                                  # batch i+1 is in flight while batch i trains
                torch.cuda.current_stream().wait_event(cur.ready)
                self._prev = cur
                yield cur.x, cur.label, cur.names
            if self._prev is not None:
                self._release(self._prev, free_q)
                self._prev = None
        finally:
            if self._reader_thread is t:          # (a stale generator finalised late must not disturb its successor)
                self._prev = None
            stop.set()
            free_q.put(None)

    @staticmethod
    def _release(slot: _Slot, free_q: "queue.Queue") -> None:
        slot.consumed.record(torch.cuda.current_stream())                # everything that used the slot is enqueued before this
        free_q.put(slot)
