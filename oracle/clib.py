"""ctypes binding of oracle/libselftok_oracle.so (built by `make -C oracle`)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libselftok_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE])
        _LIB = C.CDLL(path)
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def l2norm16(z: np.ndarray) -> np.ndarray:
    z, zp = _f32(z)
    assert z.shape[-1] == 16
    out = np.empty_like(z)
    lib().selftok_oracle_l2norm16(zp, out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(z.size // 16))
    return out


def vq_encode(z: np.ndarray, codebook: np.ndarray, normalize: bool = True):
    """z [n,16] (pre-norm), codebook [C,16] -> (ids int64 [n], best fp32 [n])."""
    z, zp = _f32(z)
    cb, cbp = _f32(codebook)
    n = z.size // 16
    ids = np.empty(n, dtype=np.int64)
    best = np.empty(n, dtype=np.float32)
    lib().selftok_oracle_vq_encode(zp, cbp, ids.ctypes.data_as(C.POINTER(C.c_int64)),
                                   best.ctypes.data_as(C.POINTER(C.c_float)),
                                   C.c_int64(n), C.c_int64(cb.shape[0]), C.c_int(1 if normalize else 0))
    return ids, best


def vq_encode_mt(z: np.ndarray, codebook: np.ndarray, threads: int = 0, normalize: bool = True):
    """vq_encode over row chunks on a thread pool (rows are independent; the C call releases the GIL): the same scalar
    arithmetic, only wall time differs -- lets the full-size GPU parity tests compare every row in seconds."""
    from concurrent.futures import ThreadPoolExecutor
    z = np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 16)
    n = z.shape[0]
    if threads <= 0:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                threads = min(threads, max(1, int(int(q) / int(p))))
        except Exception:
            pass
    threads = max(1, min(threads, 64, (n + 255) // 256))
    if threads == 1:
        return vq_encode(z, codebook, normalize)
    bounds = np.linspace(0, n, threads + 1).astype(np.int64)
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda i: vq_encode(z[bounds[i]:bounds[i + 1]], codebook, normalize), range(threads)))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def vq_scores(x: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    x, xp = _f32(x)
    cb, cbp = _f32(codebook)
    n = x.size // 16
    out = np.empty((n, cb.shape[0]), dtype=np.float32)
    lib().selftok_oracle_vq_scores(xp, cbp, out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(n), C.c_int64(cb.shape[0]))
    return out


def code_gather(ids: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    cb, cbp = _f32(codebook)
    out = np.empty(ids.shape + (16,), dtype=np.float32)
    lib().selftok_oracle_code_gather(ids.ctypes.data_as(C.POINTER(C.c_int64)), cbp,
                                     out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(ids.size))
    return out


def linspace(start: float, end: float, steps: int) -> np.ndarray:
    out = np.empty(steps, dtype=np.float32)
    lib().selftok_oracle_linspace(C.c_float(start), C.c_float(end), C.c_int(steps), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def diti_index(t: int, stages, k_per_stage, K: int) -> int:
    st = (C.c_int * (len(stages) + 1))(0, *stages)
    kp = (C.c_int * len(k_per_stage))(*k_per_stage)
    f = lib().selftok_oracle_diti_index
    f.restype = C.c_int64
    return int(f(C.c_int64(int(t)), st, kp, C.c_int(len(k_per_stage)), C.c_int(K)))
