"""ctypes binding of libmicrodit_comm.so (C ABI in include/microdit_comm.h): the data-parallel gradient exchange straight on RCCL.

`trainer.GradSync(transport="native")` (or MD_COMM=native) issues its reduce-scatter / all-gather / all-reduce buckets through
this communicator instead of `torch.distributed`; torch.distributed is then only the bootstrap side channel for the 128-byte
unique id.  Host-only code built in-tree with g++ against the HIP runtime; RCCL is bound at run time (the image PyTorch has mapped).
Reference being replaced: Composer's FSDP gradient reduction (configs/res_256_pretrain.yaml:117-118) and the NCCL calls under it."""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from ctypes import POINTER, byref, c_char_p, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "comm", "md_comm.cpp")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
_HEADER = os.path.join(_INCLUDE, "microdit_comm.h")
LIB_PATH = os.path.join(_HERE, "libmicrodit_comm.so")
_HASH_PATH = os.path.join(_HERE, ".libmicrodit_comm.hash")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__"]

BF16, F32 = 0, 1
UNIQUE_ID_BYTES = 128
ABI_VERSION = 2


def _source_hash() -> str:
    h = hashlib.sha256()
    for f in (_SRC, _HEADER):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(CXX_FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """Compile csrc/comm/md_comm.cpp into libmicrodit_comm.so in-tree (idempotent; host code only: g++ + the HIP runtime).
    With MD_COMM=native all N ranks of a node reach this together: an exclusive file lock serialises them (as hip.build does) and
    the library is linked to a temporary name and renamed into place, so no rank can dlopen a half-written file."""
    import fcntl
    if not force and _up_to_date():          # nothing to do: no lock file is touched (a read-only install works)
        return LIB_PATH
    with open(os.path.join(_HERE, ".libmicrodit_comm.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _up_to_date() -> bool:
    if not (os.path.exists(LIB_PATH) and os.path.exists(_HASH_PATH)):
        return False
    with open(_HASH_PATH) as fh:
        return fh.read().strip() == _source_hash()


def _build_locked(force: bool) -> str:
    want = _source_hash()
    if not force and _up_to_date():
        return LIB_PATH
    cxx = os.environ.get("CXX", "g++")
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    cmd = [cxx, *CXX_FLAGS, "-I", _INCLUDE, "-I", os.path.join(ROCM, "include"), _SRC, "-o", tmp,
           "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-ldl", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError(f"{cxx} failed on md_comm.cpp:\n{r.stdout.decode(errors='replace')}")
    os.replace(tmp, LIB_PATH)
    with open(_HASH_PATH, "w") as fh:
        fh.write(want)
    return LIB_PATH


_SIGS = {
    "md_comm_abi_version": (c_int32, []),
    "md_comm_last_error": (c_char_p, []),
    "md_comm_unique_id": (c_int32, [c_void_p]),
    "md_comm_init": (c_int32, [POINTER(c_void_p), c_void_p, c_int32, c_int32, c_int32]),
    "md_comm_destroy": (c_int32, [c_void_p]),
    "md_comm_rank": (c_int32, [c_void_p]),
    "md_comm_world": (c_int32, [c_void_p]),
    "md_comm_allreduce_bucket": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, POINTER(c_int64)]),
    "md_comm_reduce_scatter_bucket": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, POINTER(c_int64)]),
    "md_comm_all_gather_bucket": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, POINTER(c_int64)]),
    "md_comm_wait": (c_int32, [c_void_p, c_int64, c_void_p]),
    "md_comm_query": (c_int32, [c_void_p, c_int64]),
    "md_comm_synchronize": (c_int32, [c_void_p]),
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
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.md_comm_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libmicrodit_comm.so reports ABI version {L.md_comm_abi_version()}, this binding is written for {ABI_VERSION}")
        _lib = L
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        why = {-1: "bad argument", -2: "librccl.so could not be loaded", -3: (lib().md_comm_last_error() or b"").decode(errors="replace")}.get(code, "?")
        raise RuntimeError(f"{what} failed with code {code} ({why})")


class Ticket:
    """Handle of one asynchronous collective: wait() makes the CURRENT torch stream wait for it on the device (the same contract
    as the Work object torch.distributed returns for async_op=True under NCCL)."""

    def __init__(self, comm: "Comm", ticket: int):
        self.comm, self.ticket = comm, ticket

    def wait(self) -> None:
        import torch
        check(lib().md_comm_wait(self.comm.handle, self.ticket, torch.cuda.current_stream().cuda_stream), "md_comm_wait")

    def is_completed(self) -> bool:
        """Host-side poll (torch.distributed's Work.is_completed): has the collective finished on the device?"""
        r = lib().md_comm_query(self.comm.handle, self.ticket)
        if r < 0:
            check(r, "md_comm_query")
        return r == 1


class Comm:
    """One RCCL communicator per process (= per GPU), collectives on its own high-priority stream."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        assert len(unique_id) == UNIQUE_ID_BYTES
        self.handle = c_void_p()
        buf = ctypes.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        check(lib().md_comm_init(byref(self.handle), buf, rank, world, device), "md_comm_init")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        check(lib().md_comm_unique_id(buf), "md_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group=None) -> "Comm":
        """Bootstrap over an existing torch.distributed group (any backend): rank 0's unique id is broadcast as an object."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world, torch.cuda.current_device())

    @staticmethod
    def _dt(t) -> int:
        import torch
        if t.dtype == torch.bfloat16:
            return BF16
        if t.dtype == torch.float32:
            return F32
        raise TypeError(f"md_comm moves bf16 or fp32 buffers, got {t.dtype}")

    def _after(self):
        import torch
        return torch.cuda.current_stream().cuda_stream

    def all_reduce(self, buf) -> Ticket:
        t = c_int64(0)
        check(lib().md_comm_allreduce_bucket(self.handle, buf.data_ptr(), buf.numel(), self._dt(buf), self._after(), byref(t)), "md_comm_allreduce_bucket")
        return Ticket(self, t.value)

    def reduce_scatter(self, out, buf) -> Ticket:
        assert buf.numel() == out.numel() * self.world and out.dtype == buf.dtype
        t = c_int64(0)
        check(lib().md_comm_reduce_scatter_bucket(self.handle, buf.data_ptr(), out.data_ptr(), out.numel(), self._dt(buf), self._after(), byref(t)),
              "md_comm_reduce_scatter_bucket")
        return Ticket(self, t.value)

    def all_gather(self, out, mine) -> Ticket:
        assert out.numel() == mine.numel() * self.world and out.dtype == mine.dtype
        t = c_int64(0)
        check(lib().md_comm_all_gather_bucket(self.handle, mine.data_ptr(), out.data_ptr(), mine.numel(), self._dt(mine), self._after(), byref(t)),
              "md_comm_all_gather_bucket")
        return Ticket(self, t.value)

    def synchronize(self) -> None:
        check(lib().md_comm_synchronize(self.handle), "md_comm_synchronize")

    def destroy(self) -> None:
        if self.handle:
            lib().md_comm_destroy(self.handle)
            self.handle = c_void_p()
