"""ctypes wrapper of oracle/libbz_oracle.so (oracle/c/bz_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_lib_omp = None
G_KINDS = {"zero": 0, "l1": 1, "nonneg": 2, "l1box": 3, "indbox": 4}
D_KINDS = {"zero": 0, "free": 1, "box": 2}


def _open(name):
    lib = C.CDLL(os.path.join(_HERE, name))
    dp = C.POINTER(C.c_double)
    lib.bzo_panoc_run.restype = C.c_int
    lib.bzo_panoc_run.argtypes = [C.c_int64, dp, dp, C.c_int, C.c_double, dp, C.c_double, C.c_double,
                                  C.c_int, C.c_double, C.c_double, dp, dp, dp, C.c_int64, C.c_int,
                                  C.c_double, dp, dp, dp, dp]
    return lib


def load(omp=False):
    """omp=True: the all-cores build (same loops under `omp parallel for`; set OMP_NUM_THREADS before the
    first call).  Timed only — its reductions round differently, it is never the checker."""
    global _lib, _lib_omp
    if omp:
        if _lib_omp is None:
            _lib_omp = _open("libbz_oracle_omp.so")
        return _lib_omp
    if _lib is None:
        _lib = _open("libbz_oracle.so")
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def panoc_run(q, b, mu, y, x0, iters, *, g="l1", lam=0.0, g_u=None, g_lo=0.0, g_hi=0.0, D="box", D_lo=-1.0,
              D_hi=1.0, M=5, minimum_gamma=1e-7, want_trace=False, omp=False):
    n = x0.shape[0]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (q, b, g_u, mu, y, x0)]
    q, b, g_u, mu, y, x0 = arrs
    x, z = np.empty(n), np.empty(n)
    stats = np.zeros(8)
    trace = np.zeros((iters, 4)) if want_trace else None
    rc = load(omp).bzo_panoc_run(n, _p(q), _p(b), G_KINDS[g], lam, _p(g_u), g_lo, g_hi, D_KINDS[D], D_lo, D_hi,
                              _p(mu), _p(y), _p(x0), iters, M, minimum_gamma, _p(x), _p(z), _p(stats),
                              _p(trace) if want_trace else None)
    if rc != 0:
        raise RuntimeError(f"bzo_panoc_run failed: {rc}")
    names = ("gamma", "tau", "f_x", "g_z", "stop_norm", "n_grad", "n_backtracks", "n_halvings")
    return x, z, dict(zip(names, stats.tolist())), trace
