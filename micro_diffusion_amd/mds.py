"""ctypes binding of libmicrodit_io.so (C ABI in include/microdit_io.h): memory-mapped reader of the uncompressed MDS
shards that hold the precomputed latents (reference: `streaming.StreamingDataset` as used by
micro_diffusion/datasets/latents_loader.py:8-70).  Host-only code, built in-tree with g++; no fallback reader exists —
a missing library raises."""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from ctypes import POINTER, c_char_p, c_int, c_int32, c_int64, c_void_p, byref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "io", "mds_reader.cpp")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
_HEADER = os.path.join(_INCLUDE, "microdit_io.h")
LIB_PATH = os.path.join(_HERE, "libmicrodit_io.so")
_HASH_PATH = os.path.join(_HERE, ".libmicrodit_io.hash")
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall"]

OK, BAD_ARG, NOT_FOUND, BAD_FORMAT, UNSUPPORTED, SIZE_MISMATCH = 0, -1, -2, -3, -4, -5


class MDSError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[md_io {code}] {message}")
        self.code = code


def _source_hash() -> str:
    h = hashlib.sha256()
    for f in (_SRC, _HEADER):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(CXX_FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """Compile csrc/io/mds_reader.cpp into libmicrodit_io.so in-tree (idempotent)."""
    want = _source_hash()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(_HASH_PATH):
        with open(_HASH_PATH) as fh:
            if fh.read().strip() == want:
                return LIB_PATH
    cxx = os.environ.get("CXX", "g++")
    r = subprocess.run([cxx, *CXX_FLAGS, "-I", _INCLUDE, _SRC, "-o", LIB_PATH], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"{cxx} failed on mds_reader.cpp:\n{r.stdout.decode(errors='replace')}")
    with open(_HASH_PATH, "w") as fh:
        fh.write(want)
    return LIB_PATH


_SIGS = {
    "md_io_abi_version": (c_int32, []),
    "md_mds_open": (c_int, [c_char_p, POINTER(c_void_p)]),
    "md_mds_close": (None, [c_void_p]),
    "md_mds_last_error": (c_char_p, [c_void_p]),
    "md_mds_num_samples": (c_int64, [c_void_p]),
    "md_mds_num_shards": (c_int32, [c_void_p]),
    "md_mds_num_columns": (c_int32, [c_void_p]),
    "md_mds_column_name": (c_char_p, [c_void_p, c_int32]),
    "md_mds_column_encoding": (c_char_p, [c_void_p, c_int32]),
    "md_mds_column_index": (c_int32, [c_void_p, c_char_p]),
    "md_mds_sample_size": (c_int, [c_void_p, c_int64, c_int32, POINTER(c_int64)]),
    "md_mds_read_sample": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, POINTER(c_int64)]),
    "md_mds_read_batch": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int64, c_int32]),
}


def exported_symbols():
    return sorted(_SIGS)


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        stale = True
        if os.path.exists(LIB_PATH) and os.path.exists(_HASH_PATH):
            with open(_HASH_PATH) as fh:
                stale = fh.read().strip() != _source_hash()
        if stale:
            build()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
        if _lib.md_io_abi_version() != 1:
            raise RuntimeError("libmicrodit_io.so ABI version mismatch; rebuild")
    return _lib


class MDSDir:
    """One local MDS directory (= `streaming.Stream(remote=None, local=dir)`, latents_loader.py:89)."""

    def __init__(self, local: str):
        self._L = lib()
        self._h = c_void_p()
        rc = self._L.md_mds_open(os.fsencode(local), byref(self._h))
        if rc != OK:
            raise MDSError(rc, (self._L.md_mds_last_error(None) or b"").decode(errors="replace"))
        self.local = local
        self.num_samples = int(self._L.md_mds_num_samples(self._h))
        self.num_shards = int(self._L.md_mds_num_shards(self._h))
        nc = self._L.md_mds_num_columns(self._h)
        self.column_names = [self._L.md_mds_column_name(self._h, c).decode() for c in range(nc)]
        self.column_encodings = [self._L.md_mds_column_encoding(self._h, c).decode() for c in range(nc)]

    def close(self):
        if getattr(self, "_h", None):
            self._L.md_mds_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library may already be gone
            pass

    def __len__(self):
        return self.num_samples

    def _check(self, rc: int):
        if rc != OK:
            raise MDSError(rc, (self._L.md_mds_last_error(self._h) or b"").decode(errors="replace"))

    def column(self, name: str) -> int:
        c = self._L.md_mds_column_index(self._h, name.encode())
        if c < 0:
            raise KeyError(f"{self.local}: no column {name!r} (have {self.column_names})")
        return c

    def sample_size(self, sample: int, column: int) -> int:
        n = c_int64()
        self._check(self._L.md_mds_sample_size(self._h, sample, column, byref(n)))
        return n.value

    def read_value(self, sample: int, column: int) -> bytes:
        """One column value as bytes (the per-sample API of StreamingDataset.__getitem__)."""
        n = self.sample_size(sample, column)
        buf = ctypes.create_string_buffer(max(n, 1))
        got = c_int64()
        self._check(self._L.md_mds_read_sample(self._h, sample, column, buf, n, byref(got)))
        return buf.raw[:got.value]

    def read_batch(self, samples: np.ndarray, column: int, dst_ptr: int, row_bytes: int, row_stride: int, n_threads: int = 4):
        """Gather `column` of `samples` (int64 array) into host memory at dst_ptr, one row of row_bytes per sample."""
        samples = np.ascontiguousarray(samples, dtype=np.int64)
        self._check(self._L.md_mds_read_batch(self._h, samples.ctypes.data, len(samples), column, dst_ptr, row_bytes,
                                              row_stride, n_threads))
