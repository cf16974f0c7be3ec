"""ctypes binding of libbazinga_hip.so (include/bazinga_hip.h).

There is no CPU fallback: if the library is missing this module raises, and if no
GPU is visible every compute entry point returns BZ_ERR_HIP (raised as RuntimeError).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbazinga_hip.so")

BZ_OK = 0
BZ_ERR_ARG, BZ_ERR_HIP, BZ_ERR_UNSUPPORTED, BZ_ERR_STATE, BZ_ERR_COMM, BZ_ERR_MU, BZ_ERR_CALLBACK = -1, -2, -3, -4, -5, -6, -7
BZ_F64, BZ_F32 = 0, 1
BZ_F_ZERO, BZ_F_DIAG_QUADRATIC, BZ_F_STENCIL5, BZ_F_LEAST_SQUARES, BZ_F_QUADRATIC = 0, 1, 2, 3, 4
BZ_G_ZERO, BZ_G_NORM_L1, BZ_G_NORM_L1_NONNEG, BZ_G_NORM_L1_BOX, BZ_G_IND_BOX, BZ_G_NORM_L0_BOX = 0, 1, 2, 3, 4, 5
BZ_G_NORM_LP_NONNEG, BZ_G_NORM_LP_BOX = 6, 7
BZ_C_IDENTITY, BZ_C_DENSE_AFFINE = 0, 1
BZ_D_ZERO, BZ_D_FREE, BZ_D_BOX = 0, 1, 2
BZ_D_VC_PAIRS, BZ_D_CC_PAIRS, BZ_D_EITHEROR_PAIRS, BZ_D_XOR_PAIRS = 3, 4, 5, 6
BZ_F_CALLBACK, BZ_G_CALLBACK, BZ_C_CALLBACK, BZ_D_CALLBACK = 5, 8, 2, 7
# host-callback oracle protocol (include/bazinga_hip.h)
F_GRADIENT_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
G_PROX_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int64)
C_EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64)
C_JTPROD_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64)
D_PROJ_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
NUM_KERNEL_CATEGORIES = 16
KERNEL_CATEGORIES = ("k_axpy_dot", "k_fused_sep", "al_gradient", "fb_step",
                     "lbfgs_update", "collect", "all_gather", "misc", "k_dot", "gemv", "k_twoloop_persist", "k_gemv_t_mfma",
                     "k_fused_iterates", "k_stencil_fb", "k_stencil_update", "x_d")


class CtxOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32),
                ("flags", C.c_int32), ("comm_id", C.c_void_p)]


BZ_CTX_RUNTIME_TUNING = 1
BZ_CTX_SHARED_DEVICE = 2


class ProfileRec(C.Structure):
    _fields_ = [("timed_launches", C.c_int64), ("timed_ms", C.c_double), ("timed_bytes", C.c_double),
                ("launches", C.c_int64), ("bytes", C.c_double), ("form", C.c_char * 96)]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("f_kind", C.c_int32), ("g_kind", C.c_int32), ("c_kind", C.c_int32),
        ("D_kind", C.c_int32), ("slack", C.c_int32),
        ("n", C.c_int64), ("ny", C.c_int64),
        ("f_q", C.c_void_p), ("f_b", C.c_void_p), ("f_grid_nx", C.c_int64), ("f_grid_ny", C.c_int64),
        ("f_A", C.c_void_p), ("f_rows", C.c_int64),
        ("g_lambda", C.c_double), ("g_p", C.c_double), ("g_u", C.c_void_p), ("g_lo", C.c_double), ("g_hi", C.c_double),
        ("g_lo_vec", C.c_void_p), ("g_hi_vec", C.c_void_p),
        ("c_A", C.c_void_p), ("c_b", C.c_void_p),
        ("D_lo", C.c_double), ("D_hi", C.c_double), ("D_lo_vec", C.c_void_p), ("D_hi_vec", C.c_void_p),
        ("cb_user", C.c_void_p), ("cb_f_gradient", F_GRADIENT_FN), ("cb_g_prox", G_PROX_FN),
        ("cb_c_eval", C_EVAL_FN), ("cb_c_jtprod", C_JTPROD_FN), ("cb_D_proj", D_PROJ_FN),
    ]


class PanocOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("maxit", C.c_int64), ("freq", C.c_int32), ("verbose", C.c_int32),
                ("minimum_gamma", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("max_backtracks", C.c_int32), ("lbfgs_memory", C.c_int32), ("fuse", C.c_int32),
                ("persist", C.c_int32), ("lbfgs_compact", C.c_int32), ("affine_refresh", C.c_int32),
                ("directions", C.c_int32), ("reserved", C.c_int32), ("broyden_theta_bar", C.c_double),
                ("gamma", C.c_double), ("Lf", C.c_double), ("adaptive", C.c_int32), ("reserved2", C.c_int32)]


BZ_DIR_LBFGS, BZ_DIR_ANDERSON, BZ_DIR_BROYDEN = 0, 1, 2


class PanocStats(C.Structure):
    _fields_ = [("iters", C.c_int64), ("f_z", C.c_double), ("g_z", C.c_double), ("al_z", C.c_double),
                ("gamma", C.c_double), ("tau", C.c_double), ("stop_norm", C.c_double),
                ("n_grad", C.c_int64), ("n_prox", C.c_int64), ("n_backtracks", C.c_int64),
                ("n_gamma_halvings", C.c_int64), ("n_fused_iters", C.c_int64),
                ("n_lbfgs_skips", C.c_int64), ("elapsed_s", C.c_double), ("status", C.c_int32),
                ("persist_fallbacks", C.c_int32), ("n_affine_images", C.c_int64),
                ("n_gated_launches", C.c_int64), ("n_gate_aborts", C.c_int64), ("n_gate_fallbacks", C.c_int64),
                ("n_dense_onepass", C.c_int64), ("n_dense_fallbacks", C.c_int64)]


class AlpsOpts(C.Structure):
    _fields_ = [("tol_prim", C.c_double), ("tol_dual", C.c_double), ("inner_tol", C.c_double),
                ("maxit", C.c_int64), ("theta_penalty", C.c_double), ("kappa_penalty", C.c_double),
                ("kappa_tol", C.c_double), ("subsolver_maxit", C.c_int64), ("verbose", C.c_int32),
                ("warm_start", C.c_int32)]


class AlpsStats(C.Structure):
    _fields_ = [("tot_it", C.c_int64), ("tot_inner_it", C.c_int64), ("elapsed_s", C.c_double),
                ("status", C.c_int32), ("reserved", C.c_int32), ("inner_tol", C.c_double),
                ("norm_res_prim", C.c_double), ("objective", C.c_double)]


_P = C.POINTER
_vp = C.c_void_p

# name -> (restype, argtypes): every function include/bazinga_hip.h declares
SIGNATURES = {
    "bz_comm_unique_id": (C.c_int, [_vp]),
    "bz_ctx_create": (C.c_int, [_P(CtxOpts), _P(_vp)]),
    "bz_ctx_destroy": (None, [_vp]),
    "bz_ctx_synchronize": (C.c_int, [_vp]),
    "bz_ctx_comm_nranks": (C.c_int, [_vp, _P(C.c_int32)]),
    "bz_ctx_p2p_export": (C.c_int, [_vp, _vp]),
    "bz_ctx_p2p_connect": (C.c_int, [_vp, _vp, _P(C.c_int32)]),
    "bz_last_error": (C.c_char_p, []),
    "bz_version": (C.c_char_p, []),
    "bz_device_info": (C.c_int, [_vp, C.c_char_p, _P(C.c_int32), _P(C.c_int64)]),
    "bz_problem_create": (C.c_int, [_vp, _P(ProblemDesc), _P(_vp)]),
    "bz_problem_destroy": (None, [_vp]),
    "bz_panoc_default_opts": (None, [_P(PanocOpts)]),
    "bz_problem_set_multipliers": (C.c_int, [_vp, _vp, _vp]),
    "bz_panoc_solve": (C.c_int, [_vp, _P(PanocOpts), _vp, _vp, _P(PanocStats)]),
    "bz_panoc_begin": (C.c_int, [_vp, _P(PanocOpts), _vp]),
    "bz_problem_halo_export": (C.c_int, [_vp, _vp]),
    "bz_problem_halo_connect": (C.c_int, [_vp, _vp, _vp]),
    "bz_problem_allreduce_export": (C.c_int, [_vp, _vp]),
    "bz_problem_allreduce_connect": (C.c_int, [_vp, _vp]),
    "bz_panoc_step": (C.c_int, [_vp]),
    "bz_panoc_steps": (C.c_int, [_vp, C.c_int64]),
    "bz_panoc_finish": (C.c_int, [_vp, _vp, _P(PanocStats)]),
    "bz_panoc_scalars": (C.c_int, [_vp, _P(C.c_double)]),
    "bz_panoc_vector": (C.c_int, [_vp, C.c_int32, _vp]),
    "bz_alps_default_opts": (None, [_P(AlpsOpts), C.c_int32]),
    "bz_alps_solve": (C.c_int, [_vp, _P(AlpsOpts), _P(PanocOpts), _vp, _vp, _vp, _vp, _vp, _vp, _P(AlpsStats)]),
    "bz_als_solve": (C.c_int, [_vp, _P(AlpsOpts), _P(PanocOpts), _vp, _vp, _vp, _vp, _vp, _vp, _P(AlpsStats)]),
    "bz_eval_al_gradient": (C.c_int, [_vp, _vp, _vp, _P(C.c_double)]),
    "bz_eval_prox": (C.c_int, [_vp, _vp, C.c_double, _vp, _P(C.c_double)]),
    "bz_eval_lbfgs": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "bz_profile_enable": (C.c_int, [_vp, C.c_int32]),
    "bz_profile_get": (C.c_int, [_vp, C.c_int32, _P(C.c_int64), _P(C.c_double)]),
    "bz_profile_reset": (C.c_int, [_vp]),
    "bz_profile_get2": (C.c_int, [_vp, C.c_int32, _P(ProfileRec)]),
    "bz_runtime_tuning": (C.c_int, []),
    "bz_callback_abort": (None, []),
}

_lib = None


class BazingaHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbazinga_hip error {code}: {msg}")
        self.code = code


def load():
    """Load the shared library (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C bazinga.jl_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != BZ_OK:
        msg = load().bz_last_error().decode("utf-8", "replace")
        if rc == BZ_ERR_MU:
            raise ValueError(msg)       # Julia: error("parameters `mu` must be positive")
        raise BazingaHipError(rc, msg)
