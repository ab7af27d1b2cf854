"""ctypes binding of libcolbert_hip.so (include/colbert_hip.h).  There is no CPU fallback: if the
library is missing, or there is no GPU, calls raise -- loudly."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# COLBERT_HIP_LIB: another build of the same library (e.g. the tuning build `make ABLATIONS=1`)
LIB_PATH = os.environ.get("COLBERT_HIP_LIB") or os.path.join(CSRC, "libcolbert_hip.so")
HEADER = os.path.normpath(os.path.join(_HERE, "..", "include", "colbert_hip.h"))


class ColBERTError(RuntimeError):
    code = -1


class DimensionMismatch(ColBERTError):
    code = 1


class DomainError(ColBERTError):
    code = 2


class BoundsError(ColBERTError):
    code = 3


class ArgumentError(ColBERTError):
    code = 4


class HipError(ColBERTError):
    code = 10


class Unsupported(ColBERTError):
    code = 11


class OutOfMemory(ColBERTError):
    code = 12


_ERRORS = {c.code: c for c in (DimensionMismatch, DomainError, BoundsError, ArgumentError, HipError,
                               Unsupported, OutOfMemory)}


def build(force: bool = False, jobs: int = 3) -> str:
    """Compile libcolbert_hip.so for gfx950 with hipcc (colbert.jl_amd/csrc/Makefile)."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def declared_symbols() -> list[str]:
    """Every function name include/colbert_hip.h declares."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clb_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ColBERTError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7 (same SONAME as the
        # system one this library links against).  Whichever is loaded first serves both; torch cannot
        # initialise on top of the system copy, so when torch is installed it is imported first.
        if os.environ.get("COLBERT_HIP_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        l = C.CDLL(LIB_PATH)
        l.clb_version.restype = C.c_char_p
        l.clb_last_error.restype = C.c_char_p
        l.clb_searcher_device_bytes.restype = C.c_int64
        l.clb_packed_topk_bytes.restype = C.c_int64
        l.clb_comm_unique_id_bytes.restype = C.c_int64
        l.clb_kmeans_shard_block_bytes.restype = C.c_int64
        _lib = l
    return _lib


def measure_copy_rate(device: int = 0, nbytes: int = 1 << 30, reps: int = 5) -> float:
    """GB/s (read + written) of a device-to-device copy of `nbytes` on `device`, now (clb_measure_copy_rate)."""
    out = C.c_double(0.0)
    check(lib().clb_measure_copy_rate(C.c_int(device), C.c_int64(nbytes), C.c_int(reps), C.byref(out)))
    return float(out.value)


def measure_read_rate(device: int = 0, nbytes: int = 1 << 30, reps: int = 5) -> float:
    """GB/s of a read-only sweep over `nbytes` on `device`, now (clb_measure_read_rate)."""
    out = C.c_double(0.0)
    check(lib().clb_measure_read_rate(C.c_int(device), C.c_int64(nbytes), C.c_int(reps), C.byref(out)))
    return float(out.value)


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().clb_last_error().decode(errors="replace")
        raise _ERRORS.get(rc, ColBERTError)(msg or f"libcolbert_hip error {rc}")


def fptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def colmajor(a, dtype) -> np.ndarray:
    """A numpy array laid out like the Julia Array of the same shape (column-major, dense)."""
    return np.asfortranarray(np.asarray(a, dtype=dtype))


i64 = C.c_int64
