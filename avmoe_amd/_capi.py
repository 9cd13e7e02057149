"""ctypes binding of include/avmoe.h (libavmoe_hip.so).  Loading fails loudly: there is no CPU or
PyTorch fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVMOE_LIB", os.path.join(_HERE, "lib", "libavmoe_hip.so"))   # AVMOE_LIB: A/B builds (dev)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "avmoe.h")

F32, BF16 = 0, 1
K_MAJOR, MN_MAJOR = 0, 1


class AvmoeError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("M", "N", "K", "nb1", "nb2", "dtype", "out_dtype", "a_layout",
                                         "b_layout", "accumulate", "ksplit", "tile", "fp32_planes")] + \
               [("alpha", C.c_float)] + \
               [(n, C.c_int64) for n in ("lda", "ldb", "sA1", "sA2", "sB1", "sB2", "sCi", "sCj", "sC1", "sC2",
                                         "sRS1", "sRS2", "sDi", "sD1", "sD2")]


_lib = None


def lib():
    """The loaded library.  Built on demand if the .so is absent (hipcc needed); never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        from . import build as _b
        _b.build()
    if not os.path.isfile(LIB_PATH):
        raise AvmoeError(f"{LIB_PATH} is missing: build it with `python -m avmoe_amd.build` "
                         "(the adapter path has no fallback)")
    import torch  # noqa: F401  -- first: the library must bind to the HIP runtime torch has loaded (one runtime per process)
    L = C.CDLL(LIB_PATH)
    L.avmoe_abi_version.restype = C.c_int
    L.avmoe_last_error.restype = C.c_char_p
    L.avmoe_gemm_workspace_bytes.restype = C.c_size_t
    L.avmoe_gemm_workspace_bytes.argtypes = [C.POINTER(GemmDesc)]
    L.avmoe_gemm.restype = C.c_int
    L.avmoe_gemm.argtypes = [C.POINTER(GemmDesc)] + [C.c_void_p] * 7
    L.avmoe_prof_enable.argtypes = [C.c_int]
    L.avmoe_prof_enable.restype = None
    L.avmoe_prof_reset.restype = None
    L.avmoe_prof_report.restype = C.c_size_t
    L.avmoe_prof_report.argtypes = [C.c_char_p, C.c_size_t]
    if hasattr(L, "avmoe_test_hooks"):          # (ABI 11; a development A/B may load an older library through AVMOE_LIB)
        L.avmoe_test_hooks.restype = C.c_uint32
        L.avmoe_test_hooks.argtypes = [C.c_uint32, C.c_int32]
    from . import _capi_moe
    _capi_moe.declare(L)
    _lib = L
    return L


HOOK_TOKPAIR2_FORCE, HOOK_DPAIR_FORCE, HOOK_HOP1S_FORCE, HOOK_KFS_FORCE, HOOK_KFS_OFF = 1, 2, 4, 8, 16
HOOK_ALL_FORCE = 1 | 2 | 4 | 8          # every size threshold lifted: small shapes through the benchmarked kernels
# what this process last set (force mask, nxn chunk); like the library's own initial values, seeded from the environment once
_hooks = [(1 if os.environ.get("AVMOE_TOKPAIR2_FORCE") is not None else 0) | (2 if os.environ.get("AVMOE_DPAIR_FORCE") is not None else 0)
          | (4 if os.environ.get("AVMOE_HOP1S_FORCE") is not None else 0) | (8 if os.environ.get("AVMOE_KFS_FORCE") is not None else 0)
          | (16 if os.environ.get("AVMOE_KFS_OFF") is not None else 0), max(0, int(os.environ.get("AVMOE_NXN_CHUNK", "0") or 0))]


class test_hooks:
    """Context manager around avmoe_test_hooks (include/avmoe.h): lifts the streaming kernels' size thresholds / sets the N x N
    block's frames per chunk for the calls inside it, and restores what was set before.  Tests and bench.py's parity leg only."""

    def __init__(self, force_mask: int = 0, nxn_chunk: int = 0):
        self.new = (int(force_mask), int(nxn_chunk))

    def __enter__(self):
        self.old = tuple(_hooks)
        _hooks[:] = self.new
        lib().avmoe_test_hooks(*self.new)
        return self

    def __exit__(self, *exc):
        _hooks[:] = self.old
        lib().avmoe_test_hooks(*self.old)
        return False


def check(status: int, what: str = "avmoe"):
    if status != 0:
        msg = lib().avmoe_last_error()
        raise AvmoeError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def exported_symbols():
    """Every function include/avmoe.h declares (parsed from the header) -- the CPU test checks that
    the built library exports each of them."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(avmoe_[a-z0-9_]+)\s*\(", txt)))


def prof_report():
    """Per-kernel-family timing collected while avmoe_prof_enable(1) was on (list of dicts)."""
    import json
    L = lib()
    n = L.avmoe_prof_report(None, 0)
    buf = C.create_string_buffer(n + 1)
    L.avmoe_prof_report(buf, n + 1)
    return json.loads(buf.value.decode())
