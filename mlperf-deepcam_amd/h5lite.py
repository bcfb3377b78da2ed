"""Minimal HDF5 access through the HDF5 C library (ctypes), for images that have libhdf5 but not h5py.

The reference reads its samples with h5py (`data/cam_hdf5_dataset.py:86-131`): groups, contiguous n-d datasets of native
float / integer types, whole-dataset reads.  That is all this module covers:

    with File(path) as f:                 # "r" (default) | "w"
        f.shape("climate/data")           # (768, 1152, 16)
        f.dtype("climate/data")           # numpy dtype
        f.read("climate/minval")          # -> new ndarray
        f.read_direct("climate/data", out)            # into a caller buffer (e.g. pinned staging memory), no temporaries
        f.write("climate/mean", array)    # "w" mode: creates intermediate groups

`read_direct` of a contiguous, unfiltered dataset whose file type equals the buffer type bypasses H5Dread: the dataset's byte
offset is taken from H5Dget_offset and the payload is pread() straight into the buffer, without the library lock -- several
reader threads then stream different files in parallel (libhdf5 itself is entered by one thread at a time).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
import threading
from typing import Optional, Tuple

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE = 0
H5D_CONTIGUOUS = 1
HADDR_UNDEF = (1 << 64) - 1

_lock = threading.RLock()
_lib: Optional[C.CDLL] = None


class H5Error(RuntimeError):
    pass


def _candidates():
    env = os.environ.get("DEEPCAM_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*",
                "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def available() -> bool:
    try:
        _load()
        return True
    except H5Error:
        return False


def _load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for cand in _candidates():
        try:
            lib = C.CDLL(cand)
            lib.H5open.restype = herr_t
            if lib.H5open() < 0:
                continue
        except OSError as e:
            last = e
            continue
        sig = {
            "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
            "H5Fclose": (herr_t, [hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dget_offset": (C.c_uint64, [hid_t]),
            "H5Dget_create_plist": (hid_t, [hid_t]), "H5Pget_layout": (C.c_int, [hid_t]), "H5Pget_nfilters": (C.c_int, [hid_t]),
            "H5Pclose": (herr_t, [hid_t]), "H5Pcreate": (hid_t, [hid_t]), "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tget_sign": (C.c_int, [hid_t]), "H5Tget_order": (C.c_int, [hid_t]), "H5Tclose": (herr_t, [hid_t]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]), "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        }
        try:
            for name, (res, args) in sig.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
        except AttributeError as e:
            last = e
            continue
        lib.H5Eset_auto2(0, None, None)          # errors are reported through return codes -> H5Error, not printed
        _lib = lib
        return lib
    raise H5Error(f"no usable HDF5 C library found (set DEEPCAM_HDF5_LIB); last error: {last}")


def _gid(name: str) -> int:
    return hid_t.in_dll(_load(), name).value


_NATIVE = {"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int8": "H5T_NATIVE_INT8_g", "uint8": "H5T_NATIVE_UINT8_g",
           "int16": "H5T_NATIVE_INT16_g", "uint16": "H5T_NATIVE_UINT16_g", "int32": "H5T_NATIVE_INT32_g", "uint32": "H5T_NATIVE_UINT32_g",
           "int64": "H5T_NATIVE_INT64_g", "uint64": "H5T_NATIVE_UINT64_g"}


def _native_type(dt: np.dtype) -> int:
    key = np.dtype(dt).name
    if key not in _NATIVE:
        raise H5Error(f"unsupported dtype {dt}")
    return _gid(_NATIVE[key])


class File:
    def __init__(self, path: str, mode: str = "r"):
        lib = _load()
        self.path, self.mode = path, mode
        with _lock:
            if mode == "r":
                self.fid = lib.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
            elif mode == "w":
                self.fid = lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
            else:
                raise ValueError("mode must be 'r' or 'w'")
        if self.fid < 0:
            raise H5Error(f"cannot open {path} ({mode})")
        self._raw_fd: Optional[int] = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if getattr(self, "fid", -1) >= 0:
            with _lock:
                _load().H5Fclose(self.fid)
            self.fid = -1
        if self._raw_fd is not None:
            os.close(self._raw_fd)
            self._raw_fd = None

    def __contains__(self, name: str) -> bool:
        lib = _load()
        with _lock:
            parts = name.strip("/").split("/")
            for i in range(1, len(parts) + 1):
                if lib.H5Lexists(self.fid, "/".join(parts[:i]).encode(), H5P_DEFAULT) <= 0:
                    return False
        return True

    # -- metadata -----------------------------------------------------------------------------------------------------
    def _open(self, name: str) -> int:
        did = _load().H5Dopen2(self.fid, name.encode(), H5P_DEFAULT)
        if did < 0:
            raise H5Error(f"{self.path}: no dataset {name!r}")
        return did

    def _info(self, did: int) -> Tuple[Tuple[int, ...], np.dtype]:
        lib = _load()
        sid = lib.H5Dget_space(did)
        nd = lib.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(sid, dims, None)
        lib.H5Sclose(sid)
        tid = lib.H5Dget_type(did)
        cls, size, sign, order = lib.H5Tget_class(tid), lib.H5Tget_size(tid), lib.H5Tget_sign(tid), lib.H5Tget_order(tid)
        lib.H5Tclose(tid)
        if cls == H5T_FLOAT:
            dt = np.dtype(f"f{size}")
        elif cls == H5T_INTEGER:
            dt = np.dtype(f"{'u' if sign == H5T_SGN_NONE else 'i'}{size}")
        else:
            raise H5Error(f"{self.path}: dataset type class {cls} is not supported (float / integer only)")
        dt = dt.newbyteorder(">" if order == 1 else "<")
        return tuple(int(d) for d in dims[:nd]), dt

    def shape(self, name: str) -> Tuple[int, ...]:
        with _lock:
            did = self._open(name)
            try:
                return self._info(did)[0]
            finally:
                _load().H5Dclose(did)

    def dtype(self, name: str) -> np.dtype:
        with _lock:
            did = self._open(name)
            try:
                return self._info(did)[1]
            finally:
                _load().H5Dclose(did)

    # -- reading ------------------------------------------------------------------------------------------------------
    def read_direct(self, name: str, out: np.ndarray) -> None:
        """Whole dataset into `out` (C-contiguous, same shape; the library converts the element type if it differs)."""
        lib = _load()
        if not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise H5Error("read_direct needs a writable C-contiguous buffer")
        with _lock:
            did = self._open(name)
            try:
                shape, fdt = self._info(did)
                if tuple(out.shape) != shape:
                    raise H5Error(f"{self.path}:{name}: dataset shape {shape} != buffer shape {tuple(out.shape)}")
                raw_off = None
                if fdt == out.dtype and fdt.isnative:
                    pl = lib.H5Dget_create_plist(did)
                    plain = lib.H5Pget_layout(pl) == H5D_CONTIGUOUS and lib.H5Pget_nfilters(pl) == 0
                    lib.H5Pclose(pl)
                    off = lib.H5Dget_offset(did)
                    if plain and off != HADDR_UNDEF:
                        raw_off = int(off)
                if raw_off is None:
                    rc = lib.H5Dread(did, _native_type(out.dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
                    if rc < 0:
                        raise H5Error(f"{self.path}: reading {name!r} failed")
                    return
                if self._raw_fd is None:
                    self._raw_fd = os.open(self.path, os.O_RDONLY)
                fd = self._raw_fd
            finally:
                lib.H5Dclose(did)
        # contiguous native payload: plain pread outside the library lock
        view = memoryview(out).cast("B")
        done, total = 0, out.nbytes
        while done < total:
            n = os.preadv(fd, [view[done:]], raw_off + done)
            if n <= 0:
                raise H5Error(f"{self.path}: short read of {name!r}")
            done += n

    def read(self, name: str, dtype=None) -> np.ndarray:
        shape, fdt = self.shape(name), self.dtype(name)
        out = np.empty(shape, dtype=np.dtype(dtype) if dtype is not None else fdt.newbyteorder("="))
        self.read_direct(name, out)
        return out

    # -- writing (dataset preparation, tests) -----------------------------------------------------------------------
    def write(self, name: str, array) -> None:
        if self.mode != "w":
            raise H5Error("file is not open for writing")
        lib = _load()
        a = np.asarray(array)
        if a.ndim:
            a = np.ascontiguousarray(a)
        if a.dtype.kind == "f" and a.dtype.itemsize not in (4, 8):
            a = a.astype(np.float32)
        with _lock:
            tid = _native_type(a.dtype)
            if a.ndim == 0:
                sid = lib.H5Screate(0)      # H5S_SCALAR
            else:
                dims = (hsize_t * a.ndim)(*a.shape)
                sid = lib.H5Screate_simple(a.ndim, dims, None)
            lcpl = lib.H5Pcreate(_gid("H5P_CLS_LINK_CREATE_ID_g"))
            lib.H5Pset_create_intermediate_group(lcpl, 1)
            did = lib.H5Dcreate2(self.fid, name.encode(), tid, sid, lcpl, H5P_DEFAULT, H5P_DEFAULT)
            try:
                if did < 0:
                    raise H5Error(f"{self.path}: cannot create {name!r}")
                if lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) < 0:
                    raise H5Error(f"{self.path}: writing {name!r} failed")
            finally:
                if did >= 0:
                    lib.H5Dclose(did)
                lib.H5Pclose(lcpl)
                lib.H5Sclose(sid)
