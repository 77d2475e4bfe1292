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
               "-Wno-unused-result", "-Wno-inline-asm"]   # (inline-asm: gemm_w4's literal v[128:255] clobbers are "reserved" by design)

# enums (mirror include/microdit_hip.h)
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU = 0, 1, 2, 3
EPI_STORE_BF16, EPI_RESIDUAL, EPI_STORE_F32, EPI_ACCUM_F32, EPI_ATOMIC_F32, EPI_DACT, EPI_SWIGLU_BWD = 0, 1, 2, 3, 4, 5, 6
GEMM_AUTO, GEMM_REG128, GEMM_DMA128, GEMM_PACED256, GEMM_PP256, GEMM_W4 = 0, 1, 2, 3, 4, 5
GEMM_VARIANT_NAMES = {"auto": 0, "reg128": 1, "dma128": 2, "paced256": 3, "pp256": 4, "w4": 5}
# md_attn_args.bwd_split: backward kernel selector (0 = the library's rule; the others force a kernel: tests, A/B runs)
ATTN_BWD_AUTO, ATTN_BWD_FUSED_1PHASE, ATTN_BWD_FUSED_2PHASE, ATTN_BWD_FUSED_2PHASE_SPLIT, ATTN_BWD_STREAM_PAIR = 0, 2, 3, 4, 5   # (1: removed)


def _sources():
    return sorted(f for f in os.listdir(_CSRC) if f.endswith(".hip"))


def _gemm_source_hash() -> str:
    """Hash of what determines the GEMM kernels alone (gemm*.hip, their headers and generated includes, md_common.h, the public header,
    the compiler flags): the key of profiles/*_gemm_traffic.json, so that an edit of attention.hip does not void a GEMM measurement."""
    h = hashlib.sha256()
    files = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC))
             if f.endswith((".hip", ".h", ".inc")) and (f.startswith("gemm") or f == "md_common.h")]
    files.append(os.path.join(_INCLUDE, "microdit_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


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
    """Compile every .hip under csrc/ for gfx950 and link libmicrodit_hip.so in-tree (idempotent).  Serialised across
    processes by a file lock: with one process per GPU every rank may find the library stale at the same moment."""
    import fcntl
    os.makedirs(os.path.join(_CSRC, "build"), exist_ok=True)
    with open(os.path.join(_CSRC, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
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
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(_HASH_PATH, "w") as fh:
        fh.write(want)
    return LIB_PATH


class GemmProblem(Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("lda", c_int64), ("ldb", c_int64), ("M", c_int64), ("N", c_int64), ("c_off", c_int64)]


GEMM_MAX_PROBLEMS = 8
NUM_CU = 256            # MI355X; md_gemm_args.cu_limit is counted against it


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p), ("bias", c_void_p),
        ("res", c_void_p), ("gate", c_void_p), ("aux", c_void_p),
        ("M", c_int64), ("N", c_int64), ("K", c_int64),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64), ("ldc2", c_int64), ("ldr", c_int64),
        ("ldg", c_int64), ("ldaux", c_int64),
        ("sA", c_int64), ("sB", c_int64), ("sC", c_int64), ("sC2", c_int64), ("sBias", c_int64),
        ("sAux", c_int64), ("sSplit", c_int64),
        ("rows_per_sample", c_int64),
        ("batch", c_int32), ("ksplit", c_int32), ("a_kcontig", c_int32), ("b_kcontig", c_int32),
        ("mode", c_int32), ("act", c_int32), ("alpha", c_float), ("variant", c_int32), ("raster_group_n", c_int32),
        ("timeline", c_void_p), ("chosen_variant", c_void_p), ("A_list", c_void_p), ("B_list", c_void_p), ("list_segments", c_int32),
        ("problems", c_void_p), ("n_problems", c_int32), ("cu_limit", c_int32),
        ("tail_ws", c_void_p), ("tail_ws_bytes", c_int64), ("tail_mode", c_int32), ("tail_used", c_void_p),
        ("dact_cached", c_int32),
    ]


_lib = None
ABI_VERSION = 6        # MD_ABI_VERSION of include/microdit_hip.h this binding was written against


def lib() -> ctypes.CDLL:
    """Load (once) the in-tree shared library; raise loudly when it is absent."""
    global _lib
    if _lib is None and os.environ.get("MICRODIT_LIB"):       # experiments only: load an explicitly named build
        _lib = ctypes.CDLL(os.environ["MICRODIT_LIB"])
        _declare(_lib)
        return _lib
    if _lib is None:
        stale = True
        if os.path.exists(LIB_PATH) and os.path.exists(_HASH_PATH):
            with open(_HASH_PATH) as fh:
                stale = fh.read().strip() != _source_hash()
        if stale and (os.path.exists("/opt/rocm/bin/hipcc") or os.environ.get("HIPCC")):
            build()                      # sources changed since the last build: never run a stale library
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MicroDiT HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU / PyTorch fallback for the training path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
        if _lib.md_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libmicrodit_hip.so reports ABI version {_lib.md_abi_version()}, this binding is written for "
                               f"{ABI_VERSION}; rebuild")
    return _lib


NOT_ELIGIBLE = -2      # MD_NOT_ELIGIBLE: the forced kernel variant does not cover the problem (nothing was launched)


def check(code: int, what: str) -> None:
    if code != 0:
        why = {-1: "bad argument", NOT_ELIGIBLE: "forced kernel variant not eligible for this problem"}.get(code, "hipError_t")
        raise RuntimeError(f"{what} failed with code {code} ({why})")


# name -> argtypes; every function returns int and takes the stream last.
_SIGS = {}


def _sig(name, *argtypes):
    _SIGS[name] = list(argtypes)


P, I64, I32, F32 = c_void_p, c_int64, c_int32, c_float

class LnArgs(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("shift", c_void_p), ("scale", c_void_p), ("pos", c_void_p),
                ("out", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
                ("rows", c_int64), ("C", c_int64), ("ldx", c_int64), ("ldo", c_int64), ("ldmod", c_int64),
                ("rows_per_sample", c_int64), ("pos_rows", c_int64), ("eps", c_float), ("act", c_int32)]


class LnBwdArgs(Structure):
    _fields_ = [("dz", c_void_p), ("dx", c_void_p), ("dscale", c_void_p), ("dshift", c_void_p), ("dw", c_void_p),
                ("lddz", c_int64), ("lddx", c_int64), ("ldg", c_int64), ("rows_per_block", c_int64),
                ("accumulate", c_int32), ("dscale_is_output", c_int32)]


class AttnArgs(Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p), ("lse", c_void_p),
                ("d_o", c_void_p), ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p), ("delta", c_void_p),
                ("B", c_int64), ("H", c_int64), ("Sq", c_int64), ("Skv", c_int64),
                ("ldq", c_int64), ("ldk", c_int64), ("ldv", c_int64), ("ldo", c_int64),
                ("sq", c_int64), ("sk", c_int64), ("sv", c_int64), ("so", c_int64),
                ("lddq", c_int64), ("lddk", c_int64), ("lddv", c_int64), ("lddo", c_int64),
                ("sdq", c_int64), ("sdk", c_int64), ("sdv", c_int64), ("sdo", c_int64),
                ("scale", c_float), ("hd", c_int32), ("bwd_split", c_int32),
                ("hsq", c_int64), ("hsk", c_int64), ("hsv", c_int64), ("hso", c_int64),      # ABI 6: per-tensor head strides
                ("hsdq", c_int64), ("hsdk", c_int64), ("hsdv", c_int64), ("hsdo", c_int64)]  # (0 = hd: heads packed in the row)


class AdamWArgs(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("shadow", c_void_p),
                ("sumsq", c_void_p), ("g_bf16", c_void_p), ("ema", c_void_p), ("n", c_int64),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float),
                ("bias_corr1", c_float), ("bias_corr2", c_float), ("max_norm", c_float), ("grad_scale", c_float),
                ("ema_smoothing", c_float), ("zero_grad", c_int32), ("ema_mode", c_int32)]


SUMSQ_PARTIALS = 1024    # MD_SUMSQ_PARTIALS


_sig("md_gemm_bf16", POINTER(GemmArgs), P)
_sig("md_splitk_reduce_flat", P, P, I64, I64, I32, I32, P)
_sig("md_splitk_reduce", P, P, I64, I64, I64, I64, I32, I32, I32, P)
_sig("md_ln_fwd", POINTER(LnArgs), P)
_sig("md_ln_bwd", POINTER(LnArgs), POINTER(LnBwdArgs), P)
_sig("md_qkln_fwd", P, I64, I64, I64, I64, I32, I64, P, F32, P)
_sig("md_qkln_bwd", P, I64, I64, P, I64, I64, I64, I64, I32, I64, I64, P, P)
_sig("md_qkln_fwd_hm", P, I64, I64, I64, I64, I32, I64, P, I64, I64, I32, P, F32, P)
_sig("md_qkln_bwd_hm", P, I64, P, I64, P, I64, I64, I64, I64, I64, I32, I64, I32, P, P)
_sig("md_attn_fwd", POINTER(AttnArgs), P)
_sig("md_attn_bwd", POINTER(AttnArgs), P)
_sig("md_swiglu_fwd", P, I64, P, I64, I64, I64, P)
_sig("md_swiglu_bwd", P, I64, P, I64, P, I64, I64, I64, P)
_sig("md_gate_bwd", P, P, P, I64, P, P, I64, I64, I64, I64, I64, P)
_sig("md_act_fwd", P, P, I64, I32, P)
_sig("md_act_bwd", P, P, P, I64, I32, P)
_sig("md_colsum", P, I32, I64, P, I64, I64, P)
_sig("md_cast_f32_bf16", P, P, I64, P, P)
_sig("md_cast_f32_bf16_clear", P, P, I64, P)
_sig("md_cast_rows_bf16", P, I32, P, I64, I64, P, I64, P)
_sig("md_mean_tokens", P, P, I64, I64, I64, P)
_sig("md_mean_tokens_bwd", P, P, I64, I64, I64, P)
_sig("md_add_bf16", P, P, P, I64, P)
_sig("md_fill_zero", P, I64, P)
_sig("md_get_mask", P, I64, I64, I64, P, P, P, P)
_sig("md_gather_rows", P, I64, P, P, I64, I64, I64, P)
_sig("md_scatter_rows", P, I64, P, P, I64, I64, I64, P)
_sig("md_moe_route", P, P, I64, I64, I64, I32, I32, P, P, P, P)
_sig("md_moe_combine", P, P, P, P, P, I64, P, P, I64, I64, I32, I32, I64, P)
_sig("md_moe_combine_bwd", P, P, P, P, P, P, I64, I64, P)
_sig("md_moe_dispatch_bwd", P, P, P, P, I64, P, P, I64, I64, I64, I32, I32, I64, P)
_sig("md_edm_prepare", P, P, P, P, P, P, P, I64, I64, F32, F32, F32, P)
_sig("md_edm_prepare_f16", P, P, P, P, P, P, P, P, I64, I64, F32, F32, F32, P)
_sig("md_patchify", P, P, P, I64, I32, I32, I32, I32, P)
_sig("md_timestep_embed", P, P, I64, I32, P)
_sig("md_unpatchify", P, P, I64, P, P, I64, I32, I32, I32, I32, P)
_sig("md_edm_loss", P, P, P, P, P, P, P, P, I64, I64, I32, I32, I32, I32, F32, P)
_sig("md_edm_loss_train", P, P, P, P, P, P, P, P, F32, P, F32, I64, I64, I32, I32, I32, I32, F32, P)
_sig("md_edm_sampler_input", P, P, I64, F32, F32, I32, P)
_sig("md_edm_heun_update", P, P, P, P, P, I64, F32, I32, ctypes.c_double, ctypes.c_double, ctypes.c_double, F32, I32, P)
_sig("md_sumsq", P, I32, I64, P, P)
_sig("md_sumsq_finish", P, I64, P, P)
_sig("md_checksum_u16", P, I64, P, P)
_sig("md_adamw_step", POINTER(AdamWArgs), P)
_sig("md_adamw_step_ranges", POINTER(AdamWArgs), P, P, I32, P)


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
         aux=None, ldaux=0, C2=None, ldc2=0, batch=1, sA=0, sB=0, sC=0, sC2=0, sBias=0, sAux=0, sSplit=0, ksplit=1,
         variant=GEMM_AUTO, raster_group_n=0, timeline=None, stream=None, expect=0, A_list=None, B_list=None, list_segments=0, chosen=None, cu_limit=0,
         tail_ws=None, tail_mode=0, tail_used=None, dact_cached=0):
    """Raw-pointer GEMM launch.  A/B/C/... are ints (device addresses) or torch tensors.  tail_ws: a torch tensor used as the
    workspace of the whole-rounds + split-K-tail form (md_gemm_args.tail_ws); tail_used: list that receives the split it ran with (0 = not used)."""
    def ptr(x):
        if x is None:
            return None
        return x if isinstance(x, int) else x.data_ptr()
    a = GemmArgs(ptr(A), ptr(B), ptr(C), ptr(C2), ptr(bias), ptr(res), ptr(gate), ptr(aux),
                 M, N, K, lda, ldb, ldc, ldc2, ldr, ldg, ldaux, sA, sB, sC, sC2, sBias, sAux, sSplit,
                 rows_per_sample, batch, ksplit, int(a_kcontig), int(b_kcontig), mode, act, alpha, variant, raster_group_n,
                 ptr(timeline), None, ptr(A_list), ptr(B_list), list_segments, None, 0, cu_limit,
                 ptr(tail_ws), (tail_ws.numel() * tail_ws.element_size()) if tail_ws is not None else 0, tail_mode, None, dact_cached)
    ch, tu = ctypes.c_int32(-1), ctypes.c_int32(-1)
    a.chosen_variant = ctypes.addressof(ch)
    a.tail_used = ctypes.addressof(tu)
    rc = lib().md_gemm_bf16(byref(a), stream if stream is not None else stream_ptr())
    if chosen is not None:
        chosen.append(ch.value)
    if tail_used is not None:
        tail_used.append(tu.value)
    if expect is None:
        return rc
    check(rc, "md_gemm_bf16")
