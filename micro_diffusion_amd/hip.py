"""ctypes binding of libmicrodit_hip.so (the C ABI declared in include/microdit_hip.h).

This is the only place Python touches the HIP kernels.  There is deliberately NO fallback: if the shared
library is missing or a launch fails, a RuntimeError is raised (the product path never routes through the
CPU oracle or through PyTorch ops).
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
import sys
from ctypes import c_float, c_int, c_int32, c_int64, c_void_p, POINTER, Structure, byref

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(_HERE, "libmicrodit_hip.so")
_HASH_PATH = os.path.join(_HERE, ".libmicrodit_hip.hash")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-result"]

# enums (mirror include/microdit_hip.h)
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU = 0, 1, 2, 3
EPI_STORE_BF16, EPI_RESIDUAL, EPI_STORE_F32, EPI_ACCUM_F32, EPI_ATOMIC_F32, EPI_DACT = 0, 1, 2, 3, 4, 5


def _sources():
    return sorted(f for f in os.listdir(_CSRC) if f.endswith(".hip"))


def _source_hash() -> str:
    h = hashlib.sha256()
    files = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(_INCLUDE, "microdit_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip under csrc/ for gfx950 and link libmicrodit_hip.so in-tree (idempotent)."""
    want = _source_hash()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(_HASH_PATH):
        with open(_HASH_PATH) as fh:
            if fh.read().strip() == want:
                return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = os.path.join(_CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in _sources():
        obj = os.path.join(objdir, src[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, *HIPCC_FLAGS, "-I", _INCLUDE, "-c", os.path.join(_CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out.strip():
            sys.stderr.write(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    with open(_HASH_PATH, "w") as fh:
        fh.write(want)
    return LIB_PATH


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p), ("bias", c_void_p),
        ("res", c_void_p), ("gate", c_void_p), ("aux", c_void_p),
        ("M", c_int64), ("N", c_int64), ("K", c_int64),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64), ("ldc2", c_int64), ("ldr", c_int64),
        ("ldg", c_int64), ("ldaux", c_int64),
        ("sA", c_int64), ("sB", c_int64), ("sC", c_int64), ("sC2", c_int64), ("sBias", c_int64),
        ("sAux", c_int64),
        ("rows_per_sample", c_int64),
        ("batch", c_int32), ("ksplit", c_int32), ("a_kcontig", c_int32), ("b_kcontig", c_int32),
        ("mode", c_int32), ("act", c_int32), ("alpha", c_float),
    ]


_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) the in-tree shared library; raise loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MicroDiT HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU / PyTorch fallback for the training path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
        if _lib.md_abi_version() != 1:
            raise RuntimeError("libmicrodit_hip.so ABI version mismatch; rebuild")
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code} "
                           f"({'bad argument' if code == -1 else 'hipError_t'})")


# name -> argtypes; every function returns int and takes the stream last.
_SIGS = {}


def _sig(name, *argtypes):
    _SIGS[name] = list(argtypes)


P, I64, I32, F32 = c_void_p, c_int64, c_int32, c_float

_sig("md_gemm_bf16", POINTER(GemmArgs), P)
_sig("md_debug_tr_probe", P, P, P)
_sig("md_debug_mfma_probe", P, P, P, P)


def _declare(l: ctypes.CDLL) -> None:
    l.md_abi_version.restype = c_int
    l.md_abi_version.argtypes = []
    for name, argtypes in _SIGS.items():
        fn = getattr(l, name)  # AttributeError here = header/library mismatch, fail loudly
        fn.restype = c_int
        fn.argtypes = argtypes


def exported_symbols():
    """Names the header declares (used by the CPU-side ABI test)."""
    return ["md_abi_version", *_SIGS.keys()]


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def gemm(A, B, C, M, N, K, *, lda, ldb, ldc, a_kcontig=True, b_kcontig=True, mode=EPI_STORE_BF16,
         act=ACT_NONE, alpha=1.0, bias=None, res=None, ldr=0, gate=None, ldg=0, rows_per_sample=0,
         aux=None, ldaux=0, C2=None, ldc2=0, batch=1, sA=0, sB=0, sC=0, sC2=0, sBias=0, sAux=0, ksplit=1,
         stream=None):
    """Raw-pointer GEMM launch.  A/B/C/... are ints (device addresses) or torch tensors."""
    def ptr(x):
        if x is None:
            return None
        return x if isinstance(x, int) else x.data_ptr()
    a = GemmArgs(ptr(A), ptr(B), ptr(C), ptr(C2), ptr(bias), ptr(res), ptr(gate), ptr(aux),
                 M, N, K, lda, ldb, ldc, ldc2, ldr, ldg, ldaux, sA, sB, sC, sC2, sBias, sAux,
                 rows_per_sample, batch, ksplit, int(a_kcontig), int(b_kcontig), mode, act, alpha)
    check(lib().md_gemm_bf16(byref(a), stream if stream is not None else stream_ptr()), "md_gemm_bf16")
