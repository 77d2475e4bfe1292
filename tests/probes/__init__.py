"""Hardware-semantics probes (tests only): tests/probes/hw_probes.hip -> tests/probes/libmd_probes.so, bound with ctypes.
Kept out of the product library and its header; `__graft_entry__.build()` compiles it so it travels to the GPU box."""
import ctypes
import hashlib
import os
import subprocess
from ctypes import c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "hw_probes.hip")
LIB_PATH = os.path.join(_HERE, "libmd_probes.so")
_HASH_PATH = os.path.join(_HERE, ".libmd_probes.hash")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def build(force: bool = False) -> str:
    import fcntl
    h0 = hashlib.sha256(open(_SRC, "rb").read() + " ".join(FLAGS).encode()).hexdigest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(_HASH_PATH) and open(_HASH_PATH).read().strip() == h0:
        return LIB_PATH                          # up to date: no lock file is touched
    with open(os.path.join(_HERE, ".libmd_probes.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            h = hashlib.sha256(open(_SRC, "rb").read() + " ".join(FLAGS).encode()).hexdigest()
            if not force and os.path.exists(LIB_PATH) and os.path.exists(_HASH_PATH) and open(_HASH_PATH).read().strip() == h:
                return LIB_PATH
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            r = subprocess.run([hipcc, *FLAGS, _SRC, "-o", tmp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on hw_probes.hip:\n" + r.stdout.decode(errors="replace"))
            os.replace(tmp, LIB_PATH)
            with open(_HASH_PATH, "w") as fh:
                fh.write(h)
            return LIB_PATH
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        P = c_void_p
        for name, args in (("mdp_tr_probe", [P, P, P]), ("mdp_mfma_probe", [P, P, P, P]),
                           ("mdp_vmcnt_order_probe", [P, c_int64, P, P, c_int32, P]), ("mdp_cu_hog", [c_int32, c_int64, P, P])):
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = c_int32, args
    return _lib
