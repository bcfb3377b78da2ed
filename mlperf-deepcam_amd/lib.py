"""ctypes binding of libdeepcam_hip.so (the C ABI declared in include/deepcam_hip.h).

There is NO fallback: if the shared library is missing, ``load()`` raises.  Tensors cross the boundary as raw
device pointers (``tensor.data_ptr()``) plus sizes; the HIP stream is torch's current stream handle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEEPCAM_HIP_LIB") or os.path.join(_HERE, "libdeepcam_hip.so")   # env: an alternative build (A/B runs)
CSRC = os.path.join(_HERE, "csrc")

DC_F32, DC_BF16 = 0, 1
DC_ADAM, DC_ADAMW, DC_LAMB = 0, 1, 2


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("dil", C.c_int),
                ("transposed", C.c_int), ("cin", C.c_int), ("cout", C.c_int)]


class PackEntry(C.Structure):
    _fields_ = [("master", C.c_void_p), ("wf", C.c_void_p), ("wb", C.c_void_p), ("cin", C.c_int), ("cout", C.c_int),
                ("taps", C.c_int), ("kind", C.c_int)]


class FoldEntry(C.Structure):
    """dc_fold_entry (include/deepcam_hip.h): one layer's partial slabs and where their fixed-order sum goes."""
    _fields_ = [("slab", C.c_void_p), ("grad", C.c_void_p), ("kind", C.c_int), ("splits", C.c_int), ("taps", C.c_int),
                ("co", C.c_int), ("ci", C.c_int)]


DC_FOLD_CONV, DC_FOLD_CONVT, DC_FOLD_DW = 0, 1, 2


P, I, L, F, SZ = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
CD = C.POINTER(ConvDesc)

# name -> (restype, argtypes).  Order and meaning of arguments: include/deepcam_hip.h.
_SIGS = {
    "dc_last_error": (C.c_char_p, []),
    "dc_version": (I, []),
    "dc_set_option": (I, [C.c_char_p, I]),
    "dc_reset_options": (I, []),
    "dc_stream_create": (I, [I, P]),
    "dc_stream_destroy": (I, [P]),
    "dc_stream_priority_range": (I, [P, P]),
    "dc_stream_fence": (I, [P, P]),
    "dc_memset_async": (I, [P, I, SZ, P]),
    "dc_program_create": (I, [P]),
    "dc_program_destroy": (I, [P]),
    "dc_program_append": (I, [P, C.c_char_p, I, P, P]),
    "dc_program_bind": (I, [P, I, C.c_longlong]),
    "dc_program_len": (I, [P]),
    "dc_program_op_name": (C.c_char_p, [P, I]),
    "dc_program_run": (I, [P, P]),
    "dc_conv_out_hw": (I, [CD, I, I, C.POINTER(I), C.POINTER(I)]),
    "dc_conv_packed_elems": (I, [CD, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "dc_conv_pack_weights": (I, [CD, P, P, P, P]),
    "dc_pack_all": (I, [I, P, I, P]),
    "dc_conv_stat_rows": (I, [CD, I, I, I]),
    "dc_conv_fwd": (I, [CD, I, I, I, P, I, P, P, P, I, P, I, P]),
    "dc_conv_fwd_kn": (I, [CD, I, I, I, P, I, P, P, P, P, I, P, I, I, P]),
    "dc_conv_stat_rows_kn": (I, [CD, I, I, I]),
    "dc_conv_sum_row_kn": (I, [CD, I, I, I]),
    "dc_conv_dgrad_kn": (I, [CD, I, I, I, P, I, P, P, P, I, I, P]),
    "dc_conv_fwd_f32out": (I, [CD, I, I, I, P, I, P, P, I, P]),
    "dc_conv_fwd_dilated_group": (I, [CD, I, I, I, I, P, P, I, P, P, I, P, P]),
    "dc_conv_fwd_dilated_group_workspace": (SZ, [CD, I, I, I, I, P]),
    "dc_conv_fwd_dilated_group_ws": (I, [CD, I, I, I, I, P, P, I, P, P, I, P, P, SZ, P]),
    "dc_conv_dgrad": (I, [CD, I, I, I, P, I, P, P, I, I, P]),
    "dc_conv_dgrad_bnstats_rows": (I, [CD, I, I, I]),
    "dc_conv_dgrad_bnstats": (I, [CD, I, I, I, P, I, P, P, I, P, I, P, P, P, P, I, P, P]),
    "dc_conv_wgrad_workspace": (SZ, [CD, I, I, I]),
    "dc_conv_wgrad": (I, [CD, I, I, I, P, I, P, I, P, SZ, P, P]),
    "dc_conv_wgrad_group_workspace": (SZ, [CD, I, I, I, I]),
    "dc_conv_wgrad_group": (I, [CD, I, I, I, I, P, I, P, I, P, SZ, P, P]),
    "dc_conv_wgrad_plan": (I, [CD, I, I, I, I, C.POINTER(I), C.POINTER(C.c_size_t)]),
    "dc_conv_wgrad_partial": (I, [CD, I, I, I, I, P, I, P, I, P, I, P]),
    "dc_fold_slabs": (I, [P, I, P]),
    "dc_colsum": (I, [I, L, I, P, I, P, P, P]),
    "dc_colsum_workspace": (SZ, [L, I]),
    "dc_dwconv_pack_weights": (I, [I, P, P, P]),
    "dc_dwconv_fwd": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, P, I, P]),
    "dc_dwconv_fwd_fin_ok": (I, [I, I, I, I, I, I, I]),
    "dc_dwconv_fwd_fin": (I, [I, I, I, I, I, I, I, P, I, P, P, I, I, L, P, I, P, P, P, P, P, F, F, P, P, P, P, P]),
    "dc_dwconv_dgrad": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P]),
    "dc_dwconv_dgrad_bnstats_rows": (I, [I, I, I, I, I, I, I]),
    "dc_dwconv_dgrad_bnstats": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P, P, P, P, I, P, P]),
    "dc_dwconv_dgrad_wgrad_rows": (I, [I, I, I, I, I, I, I]),
    "dc_dwconv_dgrad_bnstats_wgrad": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P, P, P, P, I, P, P, P]),
    "dc_dwconv_dgrad_bnstats_wgrad_add": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P, I, P, P, P, P, I, P, P, P]),
    "dc_dwconv_dgrad_wgrad": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P, I, P, P, I, P, P]),
    "dc_dwconv_dgrad_wgrad_bnres_rows": (I, [I, I, I, I, I, I, I]),
    "dc_dwconv_dgrad_wgrad_bnres": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P, I, P, P, I, P, P, I, P, P]),
    "dc_dwconv_dgrad_sum_row_ok": (I, [I, I, I, I, I, I, I]),
    "dc_dwconv_dgrad_bnstats_wgrad_sum": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P, P, P, P, I, P, P, P]),
    "dc_dwconv_dgrad_wgrad_bnres_sum": (I, [I, I, I, I, I, I, I, P, I, P, P, I, P, I, P, I, P, P, I, P, P, I, P, P]),
    "dc_dwconv_wgrad_reduce": (I, [I, I, P, P, P]),
    "dc_dwconv_wgrad_workspace": (SZ, [I, I, I, I, I]),
    "dc_dwconv_wgrad": (I, [I, I, I, I, I, I, I, P, I, P, I, P, P, P, P, I, P]),
    "dc_bn_stat_rows": (I, [L]),
    "dc_bn_stats": (I, [I, L, I, P, I, P, P]),
    "dc_bn_finalize": (I, [I, L, P, I, P, P, P, P, P, F, F, P, P, P, P, P]),
    "dc_bn_eval_coeffs": (I, [I, P, P, P, P, F, P, P, P]),
    "dc_bn_apply": (I, [I, L, I, P, I, P, P, P, I, I, P, I, P]),
    "dc_bn_apply_fin": (I, [I, L, I, L, P, I, P, I, P, P, P, P, P, F, F, P, P, P, P, P, I, I, P, I, P]),
    "dc_bn_bwd_reduce": (I, [I, L, I, P, I, P, I, P, I, I, P, P, P, P, P, P]),
    "dc_bn_bwd_reduce_sum": (I, [I, L, I, P, I, P, I, P, I, I, P, P, P, P, P, P]),
    "dc_bn_bwd_finalize": (I, [I, P, I, P, P, P]),
    "dc_bn_bwd_apply": (I, [I, L, I, L, P, I, P, I, P, I, I, P, P, P, P, P, P, I, P, I, P, P, P]),
    "dc_bn_bwd_apply_fin_max_rows": (I, []),
    "dc_sepconv_fwd_rows": (I, [I, I, I, I, I, I, I, I]),
    "dc_sepconv_fwd": (I, [I, I, I, I, I, I, P, I, P, P, I, P, P, I, P, P, I, P, I, P]),
    "dc_pw_bn_bwd_rows": (I, [I, I, I, L]),
    "dc_pw_bn_bwd": (I, [I, L, I, I, L, P, I, P, I, I, P, P, P, P, P, P, P, P, I, P, P, I, P, I, P]),
    "dc_bn_bwd_apply_fin": (I, [I, L, I, L, P, I, P, I, P, I, I, P, P, P, P, I, P, P, P, I, P, I, P, P, P]),
    "dc_stem_stat_rows": (I, [I, I, I]),
    "dc_stem_fwd": (I, [I, I, I, I, I, P, P, P, I, P, P]),
    "dc_stem_wgrad_workspace": (SZ, [I, I, I, I]),
    "dc_stem_wgrad": (I, [I, I, I, I, I, P, P, I, P, P, P]),
    "dc_head_workspace": (SZ, [I, I, I, I, I]),
    "dc_head_fwd": (I, [I, I, I, I, I, P, I, P, P, P, P]),
    "dc_head_fwd_loss": (I, [I, I, I, I, I, P, I, P, P, P, P, I, P, F, P, P, P, P, P]),
    "dc_head_bwd": (I, [I, I, I, I, I, P, I, P, P, P, I, P, P, P]),
    "dc_head_bwd_bnstats": (I, [I, I, I, I, I, P, I, P, P, P, I, P, P, P, I, P, P, P, P, I, P, P]),
    "dc_head_fwd_bnin": (I, [I, I, I, I, I, P, I, P, P, I, P, P, P, P]),
    "dc_head_fwd_loss_bnin": (I, [I, I, I, I, I, P, I, P, P, I, P, P, P, P, I, P, F, P, P, P, P, P]),
    "dc_head_bwd_bnin": (I, [I, I, I, I, I, P, I, P, P, I, P, P, P, I, P, P, P, P, P, I, P]),
    "dc_head_bwd_bnin_apply": (I, [I, I, I, I, I, P, I, P, P, I, P, P, P, P, P, L, P, I, P, P]),
    "dc_nchw_to_nhwc": (I, [I, I, I, I, I, P, P, I, P]),
    "dc_input_normalize_hwc": (I, [I, L, I, I, P, P, P, P, P, I, P]),
    "dc_input_normalize_hwc_to_nchw": (I, [I, L, I, I, P, P, P, P, P, P]),
    "dc_wce_fused": (I, [I, I, I, P, P, I, P, F, P, P, P, P, P]),
    "dc_confusion_counts": (I, [L, P, P, I, P, P]),
    "dc_avgpool_fwd": (I, [I, I, I, I, P, I, P, P]),
    "dc_avgpool_bwd_add": (I, [I, I, I, I, P, P, I, P]),
    "dc_broadcast_hw": (I, [I, I, I, I, P, P, I, P]),
    "dc_sum_hw": (I, [I, I, I, I, P, I, P, P]),
    "dc_copy_view": (I, [I, L, I, P, I, P, I, P]),
    "dc_adam_step": (I, [I, L, P, P, P, P, P, F, F, F, F, P, F, P]),
    "dc_lamb_workspace_words": (SZ, [I, C.c_long]),
    "dc_lamb_step": (I, [I, P, L, P, P, P, P, P, F, F, F, F, P, F, F, P, P]),
    "dc_grad_pack_bf16": (I, [L, P, P, P]),
    "dc_grad_unpack_bf16": (I, [L, P, P, P]),
    "dc_comm_unique_id": (I, [P]),
    "dc_comm_create": (I, [P, I, I, C.POINTER(P)]),
    "dc_comm_adopt": (I, [P, I, I, C.POINTER(P)]),
    "dc_comm_create_callback": (I, [P, P, I, I, I, C.POINTER(P)]),
    "dc_comm_destroy": (I, [P]),
    "dc_comm_info": (I, [P, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(L)]),
    "dc_grad_allreduce_enqueue": (I, [P, P, SZ, I, P]),
    "dc_grad_allreduce_wait": (I, [P, P]),
}
EXPORTS = sorted(_SIGS)

_lib: Optional[C.CDLL] = None


class DeepcamHipError(RuntimeError):
    pass


def build(verbose: bool = False, force: bool = False) -> str:
    """Compile csrc/ into libdeepcam_hip.so for gfx950 (hipcc cross-compiles without a GPU).  force: rebuild every object
    (`make -B`), so that a tree that already carries object files still proves that the sources compile."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"] + (["-B"] if force else []), capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise DeepcamHipError("building libdeepcam_hip.so failed")
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeepcamHipError(
            f"{LIB_PATH} not found: the HIP library is the product, there is no fallback path. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C mlperf-deepcam_amd/csrc`.")
    # torch first: it brings its own HIP runtime, and the library must bind to THAT one (streams and device pointers are torch's).
    # Loaded the other way round, the process ends up with two runtimes and the library's one reports "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # DEEPCAM_HIP_OPTIONS="igemm256=0,dw_tile=1": tuning switches of dc_set_option for A/B runs of whole programs
    for item in filter(None, os.environ.get("DEEPCAM_HIP_OPTIONS", "").split(",")):
        key, _, val = item.partition("=")
        if lib.dc_set_option(key.strip().encode(), int(val)) != 0:
            raise DeepcamHipError(f"DEEPCAM_HIP_OPTIONS: {lib.dc_last_error().decode()}")
    return lib


def last_error() -> str:
    return load().dc_last_error().decode()


def check(rc: int) -> None:
    if rc != 0:
        raise DeepcamHipError(last_error())


_recorder: Optional["Program"] = None     # set by Program.recording(): every call() of THAT thread is appended to it (and executed as usual)
_recorder_thread = 0                       # (an input pipeline's reader threads issue library calls of their own: not part of the step's list)


def call(name: str, *args):
    """Call an int-returning entry point and raise on failure."""
    if _recorder is not None and threading.get_ident() == _recorder_thread:
        _recorder.append(name, *args)
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise DeepcamHipError(f"{name}: {last_error()}")


class Slot:
    """A Program argument whose value is bound before each run (Program.bind): a batch pointer that changes from step to step."""

    def __init__(self, name: str, value):
        self.name, self.value = name, value


def _word(ctype, v) -> int:
    """The 8-byte word dc_program_append stores for one argument (include/deepcam_hip.h)."""
    import struct
    if ctype is F:
        return struct.unpack("q", struct.pack("d", float(v)))[0]
    if ctype in (I, L, SZ, C.c_longlong):
        return int(v.value if hasattr(v, "value") else v)
    # pointer-like: None, int, c_void_p, byref(obj), a ctypes array / structure (its address)
    if v is None:
        return 0
    if isinstance(v, int):
        return v
    if isinstance(v, C.c_void_p):
        return v.value or 0
    if hasattr(v, "_obj"):                      # ctypes.byref(obj)
        return C.addressof(v._obj)
    if isinstance(v, (C.Array, C.Structure)):
        return C.addressof(v)
    if isinstance(v, C._Pointer):
        return C.cast(v, C.c_void_p).value or 0
    raise DeepcamHipError(f"Program: cannot record an argument of type {type(v).__name__}")


class Program:
    """A recorded launch list (dc_program_*): `with prog.recording(): step()` appends every library call of the block -- which also runs
    as usual --, `prog.run()` issues the list again from C.  The recorder keeps every argument object alive (descriptors, fold tables and
    pointer arrays are recorded by address), and the buffers the recorded device pointers refer to belong to the caller."""

    def __init__(self):
        self._h = C.c_void_p()
        check(load().dc_program_create(C.byref(self._h)))
        self._keep = []
        self._slots = {}

    def __del__(self):
        try:
            if self._h:
                load().dc_program_destroy(self._h)
        except Exception:
            pass

    def __len__(self) -> int:
        return load().dc_program_len(self._h)

    def append(self, name: str, *args) -> None:
        res, argtypes = _SIGS[name]
        if res is not I or len(argtypes) != len(args):
            raise DeepcamHipError(f"Program: {name} cannot be recorded with {len(args)} arguments")
        n = len(args)
        words, slots = (C.c_longlong * max(n, 1))(), (C.c_int * max(n, 1))()
        bound = []
        for k, (t, v) in enumerate(zip(argtypes, args)):
            slots[k] = -1
            if isinstance(v, Slot):
                slots[k] = self._slots.setdefault(v.name, len(self._slots))
                bound.append(v)
                v = v.value
            words[k] = _word(t, v)
        self._keep.append(args)
        check(load().dc_program_append(self._h, name.encode(), n, words, slots))
        for v in bound:                      # the value the slot was recorded with stays bound until bind() replaces it
            self.bind(v.name, v.value)

    def bind(self, slot: str, value, _check: bool = True) -> None:
        if slot not in self._slots:
            if _check:
                raise DeepcamHipError(f"Program: no slot named {slot!r}")
            return
        check(load().dc_program_bind(self._h, self._slots[slot], _word(P, value)))

    def recording(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            global _recorder, _recorder_thread
            if _recorder is not None:
                raise DeepcamHipError("Program: a recording is already in progress")
            _recorder, _recorder_thread = self, threading.get_ident()
            try:
                yield self
            finally:
                _recorder = None
        return ctx()

    def run(self) -> None:
        failed = C.c_int(-1)
        rc = load().dc_program_run(self._h, C.byref(failed))
        if rc != 0:
            nm = load().dc_program_op_name(self._h, failed.value)
            raise DeepcamHipError(f"program call {failed.value} ({nm.decode() if nm else '?'}): {last_error()}")

    def names(self):
        lib = load()
        return [lib.dc_program_op_name(self._h, k).decode() for k in range(len(self))]


def dtype_code(torch_dtype) -> int:
    import torch
    if torch_dtype == torch.float32:
        return DC_F32
    if torch_dtype == torch.bfloat16:
        return DC_BF16
    raise DeepcamHipError(f"unsupported activation dtype {torch_dtype}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t) -> C.c_void_p:
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())
