"""Loader of the HIP shared library (nav-gym_amd/nav_gym_amd/libnavsim_hip.so) through ctypes.

The product path has NO CPU fallback: if the library is missing or no GPU is visible, every
entry point raises.  (The CPU oracle lives in oracle/ and is never imported from here.)
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NAVSIM_LIB", os.path.join(_HERE, "libnavsim_hip.so"))   # override: diagnostic / A-B builds
_LIB = None


class NavsimError(RuntimeError):
    pass


def build_library(verbose=False):
    """Compiles csrc/navsim_kernels.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    script = os.path.join(os.path.dirname(_HERE), "csrc", "build.sh")
    out = subprocess.run(["bash", script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise NavsimError("hipcc build failed:\n" + out.stdout)
    return LIB_PATH


def source_hash():
    """First 16 hex digits of the SHA-256 over the library's sources (csrc/*.hip, *.hpp and include/navsim.h, in
    name order).  Profiles record it (profiles/summarize.py) and bench.py only quotes counter figures of a profile
    taken from the same sources."""
    import glob
    import hashlib
    csrc = os.path.join(os.path.dirname(_HERE), "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")))
    files.append(os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "navsim.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load():
    """Returns the ctypes handle with argtypes attached.  Raises NavsimError if the .so is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise NavsimError(
                "HIP extension %s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or nav-gym_amd/csrc/build.sh (there is no CPU fallback)" % LIB_PATH)
        # torch first: it ships its own libamdhip64; loading ours afterwards binds the library to the
        # SAME HIP runtime instance (one device context, shared streams).  The other order leaves two
        # runtimes in the process and ours reports "no ROCm-capable device".
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        abi.declare(L, "")
        L.navsim_abi_version.restype = C.c_int
        L.navsim_error_string.restype = C.c_char_p
        L.navsim_error_string.argtypes = [C.c_int]
        L.navsim_step_kernel_name.restype = C.c_char_p
        L.navsim_last_hip_error.restype = C.c_char_p
        L.navsim_build_dt_workspace_bytes.restype = C.c_size_t
        L.navsim_build_dt_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        for n in ("navsim_sizeof_config", "navsim_sizeof_state", "navsim_sizeof_step_io"):
            getattr(L, n).restype = C.c_size_t
        L.navsim_debug_math.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        if L.navsim_abi_version() != abi.ABI_VERSION:
            raise NavsimError("ABI version mismatch between abi.py and %s" % LIB_PATH)
        if L.navsim_sizeof_config() != C.sizeof(abi.NavsimConfig):
            raise NavsimError("navsim_config layout mismatch")
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = load().navsim_error_string(rc).decode()
        if rc == abi.E_LAUNCH:
            msg += " [HIP: %s]" % load().navsim_last_hip_error().decode()
        raise NavsimError("%s: %s (%d)" % (what, msg, rc))


def default_config(**kw):
    cfg = abi.NavsimConfig()
    check(load().navsim_default_config(C.byref(cfg)), "navsim_default_config")
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise NavsimError("no MI355X visible: the batched NavGym step has no CPU fallback")
    return torch
