"""The f/g/c/D oracle types of Bazinga.alps that this build lowers to the device.

The reference's oracles are duck-typed Julia structs (README.md:17-20,
src/Bazinga.jl:7-16).  A device cannot run arbitrary closures, so the host side
pattern-matches the structured types below and lowers them to a C descriptor
(`bz_problem_desc`, include/bazinga_hip.h).

Anything else is a GENERIC oracle (demo/rosenbrock.jl:39-80 — BASELINE config 1): an object with
the reference's protocol

    f.gradient(dfx, x) -> f(x)        gradient!(dfx, f, x)       src/Bazinga.jl:16
    g.prox(z, x, gamma) -> g(z)       prox!(z, g, x, gamma)      src/Bazinga.jl:15
    c.eval(cx, x), c.jtprod(jtv, x, v)                           src/Bazinga.jl:11-12
    D.proj(s, v)                      proj!(s, D, v)             src/Bazinga.jl:14

is handed to the library as host callbacks (BZ_*_CALLBACK): the host evaluates the oracles, the
device keeps the L-BFGS / line-search vector work.  When one of the four is generic all four travel
as callbacks (the structured types below carry the same protocol for that case).  An object with
neither raises ``UnsupportedOracle``; nothing ever falls back to a CPU solver.
"""
from __future__ import annotations

import numpy as np

import ctypes as C

from . import _lib as L


class UnsupportedOracle(TypeError):
    pass


# ----------------------------------------------------------------- abstract tags
class ProximableFunction:   # src/Bazinga.jl:7
    pass


class SmoothFunction:       # src/Bazinga.jl:8
    pass


class ClosedSetBase:        # src/Bazinga.jl:9  (abstract type ClosedSet)
    pass


# ------------------------------------------------------------------------- f
class Zero(ProximableFunction):
    """src/proxoperators/zero.jl:11-25; usable as f or g."""

    def gradient(self, dfx, x):
        dfx[...] = 0
        return x.dtype.type(0)

    def prox(self, z, x, gamma):
        z[...] = x
        return x.dtype.type(0)


class DiagQuadratic(ProximableFunction):
    """f(x) = sum_i x_i (0.5 q_i x_i - b_i) — diagonal special case of
    ProximalOperators.Quadratic (test/problems/test_nonconvex_qp.jl:14)."""

    def __init__(self, q, b):
        self.q = np.ascontiguousarray(q)
        self.b = np.ascontiguousarray(b)
        if self.q.shape != self.b.shape or self.q.ndim != 1:
            raise ValueError("q and b must be vectors of equal length")

    def gradient(self, dfx, x):
        qx = self.q * x
        dfx[...] = qx - self.b
        return np.sum(x * (x.dtype.type(0.5) * qx - self.b))


class LeastSquares(ProximableFunction):
    """ProximalOperators.LeastSquares(A, b): f(x) = 0.5||A x - b||^2 (test/problems/test_verbose.jl:22)."""

    def __init__(self, A, b):
        self.A = np.ascontiguousarray(A)
        self.b = np.ascontiguousarray(b)
        if self.A.ndim != 2 or self.b.shape != (self.A.shape[0],):
            raise ValueError("A must be m-by-n and b of length m")

    def gradient(self, dfx, x):
        r = self.A @ x - self.b
        dfx[...] = self.A.T @ r
        return x.dtype.type(0.5) * np.dot(r, r)


class Quadratic(ProximableFunction):
    """ProximalOperators.Quadratic(Q, q): f(x) = 0.5 x'Qx + q'x, Q dense symmetric
    (test/problems/test_nonconvex_qp.jl:14,65)."""

    def __init__(self, Q, q):
        self.Q = np.ascontiguousarray(Q)
        self.q = np.ascontiguousarray(q)
        if self.Q.ndim != 2 or self.Q.shape[0] != self.Q.shape[1] or self.q.shape != (self.Q.shape[0],):
            raise ValueError("Q must be n-by-n and q of length n")

    def gradient(self, dfx, x):
        Qx = self.Q @ x
        dfx[...] = Qx + self.q
        return x.dtype.type(0.5) * np.dot(x, Qx) + np.dot(self.q, x)


class Stencil5ptQuadratic(ProximableFunction):
    """f(x) = 0.5 x'A_h x - b'x on an nx-by-ny grid (row-major), A_h the 5-point Laplacian
    (4,-1,-1,-1,-1) with homogeneous Dirichlet halo — the structured `Quadratic` of BASELINE
    config 3; 1-D analogue in the reference: demo/obstacle.jl:97-112."""

    def __init__(self, nx, ny, b):
        self.nx, self.ny = int(nx), int(ny)
        self.b = np.ascontiguousarray(b)
        if self.b.shape != (self.nx * self.ny,):
            raise ValueError("b must have length nx*ny")


# ------------------------------------------------------------------------- g
class IndFree(ProximableFunction):
    """ProximalOperators.IndFree (test_nonconvex_qp.jl:39)."""

    def prox(self, z, x, gamma=1.0):
        z[...] = x
        return x.dtype.type(0)


class NormL1(ProximableFunction):
    """ProximalOperators.NormL1(lambda) (test_verbose.jl:23, demo/basispursuit.jl:63)."""

    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        self.lam = float(lam)

    def prox(self, z, x, gamma):
        gl = x.dtype.type(gamma * self.lam)
        z[...] = x + np.where(x <= -gl, gl, np.where(x >= gl, -gl, -x))
        return x.dtype.type(self.lam) * np.sum(np.abs(z))


class NormL1Nonneg(ProximableFunction):
    """src/proxoperators/normL1Nonneg.jl:9-42"""

    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("λ must be nonnegative")
        self.lam = float(lam)

    def prox(self, z, x, gamma):
        gl = x.dtype.type(gamma * self.lam)
        z[...] = np.where(x >= gl, x - gl, 0)
        return x.dtype.type(self.lam) * np.sum(z)


class NormL1Box(ProximableFunction):
    """src/proxoperators/normL1Box.jl:11-39"""

    def __init__(self, lam=1.0, *, u):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        self.u = np.ascontiguousarray(u)
        if np.any(self.u < 0):
            raise ValueError("vector u must have nonnegative entries")
        self.lam = float(lam)

    def prox(self, z, x, gamma):
        gl = x.dtype.type(gamma * self.lam)
        z[...] = np.maximum(0, np.minimum(x - gl, self.u))
        return x.dtype.type(self.lam) * np.sum(z)


class NormL0Box(ProximableFunction):
    """src/proxoperators/normL0Box.jl:12-58: lambda*nnz(x) + indicator of [0, u]."""

    def __init__(self, lam=1.0, *, u):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        self.u = np.ascontiguousarray(u)
        if np.any(self.u < 0):
            raise ValueError("vector u must have nonnegative entries")
        self.lam = float(lam)


class NormLpPowerNonneg(ProximableFunction):
    """src/proxoperators/normLpNonneg.jl:14-40: alpha * sum x^p on x >= 0, 0 < p < 1."""

    def __init__(self, p, *, alpha=1.0):
        if p <= 0:
            raise ValueError("p must be positive")
        if p >= 1:
            raise ValueError("p must be smaller than one")
        if alpha < 0:
            raise ValueError("alpha must be nonnegative")
        self.p, self.alpha = float(p), float(alpha)


class NormLpPowerBox(ProximableFunction):
    """src/proxoperators/normLpBox.jl:11-45: alpha * sum x^p on 0 <= x <= u, 0 < p < 1."""

    def __init__(self, p, alpha=1.0, *, u):
        if p <= 0:
            raise ValueError("p must be positive")
        if p >= 1:
            raise ValueError("p must be smaller than one")
        if alpha < 0:
            raise ValueError("alpha must be nonnegative")
        self.u = np.ascontiguousarray(u)
        if np.any(self.u < 0):
            raise ValueError("vector u must have nonnegative entries")
        self.p, self.alpha = float(p), float(alpha)


class IndBox(ProximableFunction):
    """ProximalOperators.IndBox(lb, ub) (test_nonconvex_qp.jl:15); scalar or vector bounds."""

    def __init__(self, lb, ub):
        self.lb = lb
        self.ub = ub
        if np.any(np.asarray(lb) > np.asarray(ub)):
            raise ValueError("bounds must satisfy lb <= ub")

    def prox(self, z, x, gamma=1.0):
        z[...] = np.where(x < self.lb, self.lb, np.where(x > self.ub, self.ub, x))
        return x.dtype.type(0)


# ------------------------------------------------------------------------- c
class IdentityFunction(SmoothFunction):
    """test/definitions/identityFunction.jl:3-13"""

    def eval(self, cx, x):
        cx[...] = x

    def jtprod(self, jtv, x, v):
        jtv[...] = v


class DenseAffine(SmoothFunction):
    """c(x) = A x - b with a dense row-major A[ny][n]: `ConstraintBasisPursuit`
    (demo/basispursuit.jl:38-49); eval! = A*x - b, jtprod! = A'*v."""

    def __init__(self, A, b):
        self.A = np.ascontiguousarray(A)
        self.b = np.ascontiguousarray(b)
        if self.A.ndim != 2 or self.b.shape != (self.A.shape[0],):
            raise ValueError("A must be ny-by-n and b of length ny")

    def eval(self, cx, x):
        cx[...] = self.A @ x - self.b

    def jtprod(self, jtv, x, v):
        jtv[...] = self.A.T @ v


# ------------------------------------------------------------------------- D
class ZeroSet(ClosedSetBase):
    """src/projections/zeroSet.jl:8-20"""

    def proj(self, s, v):
        s[...] = 0


class FreeSet(ClosedSetBase):
    """src/projections/freeSet.jl:8-20"""

    def proj(self, s, v):
        s[...] = v


class IndicatorSet(ClosedSetBase):
    """src/projections/indicatorSet.jl:4-11"""

    def __init__(self, f):
        self.f = f

    def proj(self, s, v):
        self.f.prox(s, v, 1.0)


def ClosedSet(f):
    """src/Bazinga.jl:18"""
    return IndicatorSet(f)


class PairwiseSet(ClosedSetBase):
    """D = product of a 2-element set over the ADJACENT pairs (c(x)[2j], c(x)[2j+1]) — how demo/mpvca.jl:
    105-106,147-148 and demo/eitheror.jl:79-88,123-130 build their sets from the package's 2-element
    projections.  kind: "vc" (project_onto_VC_set!, vanishingConstraints.jl:27-46), "cc"
    (project_onto_CC_set!, complementarityConstraints.jl:8-20), "eitheror" / "xor"
    (orConstraints.jl:7-17 / 24-36)."""
    KINDS = ("vc", "cc", "eitheror", "xor")

    def __init__(self, kind, layout="adjacent"):
        """layout="adjacent": pairs (c(x)[2j], c(x)[2j+1]); layout="split": pairs (c(x)[i], c(x)[i+N]), N = ny/2, the way
        demo/obstacle.jl:151-168 (SetObstacleRed) lays its complementarity pairs out.  The device kernels work on
        adjacent pairs (a pair is one 16-byte pack); the split layout is the same problem under the interleaving
        permutation, which the host binding applies to every vector at the boundary (device.Problem)."""
        if kind not in self.KINDS:
            raise ValueError(f"kind must be one of {self.KINDS}")
        if layout not in ("adjacent", "split"):
            raise ValueError("layout must be 'adjacent' or 'split'")
        self.kind = kind
        self.layout = layout

    def proj(self, s, v):
        """the 2-element projections over adjacent pairs (host protocol: only used when another oracle of the
        problem is generic)"""
        if self.layout == "split":
            N = v.shape[0] // 2
            x1, x2 = v[:N], v[N:]
        else:
            x1, x2 = v[0::2], v[1::2]
        z1, z2 = x1.copy(), x2.copy()
        if self.kind == "vc":          # vanishingConstraints.jl:27-46
            a = x1 <= 0
            b = ~a & (x2 >= 0)
            c = ~a & ~b & (x1 + x2 > 0)
            z1[...] = np.where(b | c, x1, 0)
            z2[...] = np.where(a | b | ~c, x2, 0)
            z2[c] = 0
        elif self.kind == "cc":        # complementarityConstraints.jl:8-20
            both = (x1 > 0) & (x2 > 0)
            z1[...] = np.where(both, np.where(x2 > x1, 0, x1), np.maximum(x1, 0))
            z2[...] = np.where(both, np.where(x2 > x1, x2, 0), np.maximum(x2, 0))
        elif self.kind == "eitheror":  # orConstraints.jl:7-17
            both = (x1 < 0) & (x2 < 0)
            z1[...] = np.where(both & (x1 > x2), 0, x1)
            z2[...] = np.where(both & ~(x1 > x2), 0, x2)
        else:                          # xor, orConstraints.jl:24-36
            same = x1 * x2 > 0
            up = x1 > x2
            z1[...] = np.where(same, np.where(up, np.maximum(x1, 0), np.minimum(x1, 0)), x1)
            z2[...] = np.where(same, np.where(up, np.minimum(x2, 0), np.maximum(x2, 0)), x2)
        if self.layout == "split":
            s[:N], s[N:] = z1, z2
        else:
            s[0::2] = z1
            s[1::2] = z2


def VanishingConstraintPairs():
    return PairwiseSet("vc")


def ComplementarityPairs():
    return PairwiseSet("cc")


def EitherOrPairs():
    return PairwiseSet("eitheror")


def XorPairs():
    return PairwiseSet("xor")


# ------------------------------------------------------------------ lowering
def split_permutation(n):
    """perm with x_adjacent = x_split[perm]: position 2j holds element j, position 2j+1 element j + N (N = n/2)"""
    N = n // 2
    perm = np.empty(n, np.int64)
    perm[0::2] = np.arange(N)
    perm[1::2] = np.arange(N) + N
    return perm


def _vec(a, dtype, n, name):
    v = np.ascontiguousarray(a, dtype=dtype)
    if v.shape != (n,):
        raise ValueError(f"{name} must have length {n}")
    return v


_LOWERED_F = lambda f: isinstance(f, (Zero, DiagQuadratic, LeastSquares, Quadratic, Stencil5ptQuadratic))
_LOWERED_G = lambda g: isinstance(g, (Zero, IndFree, NormL1, NormL1Nonneg, NormL1Box, NormL0Box, NormLpPowerNonneg,
                                      NormLpPowerBox, IndBox))
_LOWERED_C = lambda c: isinstance(c, (IdentityFunction, DenseAffine))
_LOWERED_D = lambda D: isinstance(D, (ZeroSet, FreeSet, PairwiseSet)) or \
    (isinstance(D, IndicatorSet) and isinstance(D.f, (IndBox, IndFree)))


class CallbackError(RuntimeError):
    """An oracle callback raised: the original exception is the __cause__."""


def lower_generic(f, g, c, D, n, ny, dtype, slack=False):
    """Generic oracles -> host callbacks (BZ_*_CALLBACK).  Returns (desc, keep): `keep` holds the ctypes thunks and
    must live as long as the problem; keep[-1] is the list exceptions raised inside callbacks are parked in."""
    if slack:
        raise UnsupportedOracle("generic (callback) oracles are not available in the slack (ALS) form")
    for obj, names, what in ((f, ("gradient",), "f"), (g, ("prox",), "g"), (c, ("eval", "jtprod"), "c"), (D, ("proj",), "D")):
        for nm in names:
            if not callable(getattr(obj, nm, None)):
                raise UnsupportedOracle(f"{what} of type {type(obj).__name__} is neither a lowered oracle type nor a generic "
                                        f"one (no `{nm}` method)")
    dt = np.dtype(dtype)
    errors = []

    ctype = C.c_double if dt == np.float64 else C.c_float

    def arr(ptr, cnt):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(cnt,))

    def guard(fn, default):
        def wrapped(*a):
            if errors:                      # an earlier callback of this library call failed: evaluate nothing more
                L.load().bz_callback_abort()
                return default
            try:
                return fn(*a)
            except BaseException as e:      # noqa: BLE001  (nothing may unwind through the C frames)
                errors.append(e)
                # the library call in progress ends with BZ_ERR_CALLBACK as soon as this callback has returned
                # (device.Problem._call then raises CallbackError from the parked exception)
                L.load().bz_callback_abort()
                return default
        return wrapped

    def f_gradient(_u, px, pdfx, nn):
        return float(f.gradient(arr(pdfx, nn), arr(px, nn)))

    def g_prox(_u, px, gamma, pz, nn):
        return float(g.prox(arr(pz, nn), arr(px, nn), dt.type(gamma)))

    def c_eval(_u, px, pcx, nn, nny):
        c.eval(arr(pcx, nny), arr(px, nn))

    def c_jtprod(_u, px, pv, pjtv, nn, nny):
        c.jtprod(arr(pjtv, nn), arr(px, nn), arr(pv, nny))

    def d_proj(_u, pv, ps, nny):
        D.proj(arr(ps, nny), arr(pv, nny))

    d = L.ProblemDesc()
    d.dtype = L.BZ_F64 if dt == np.float64 else L.BZ_F32
    d.n, d.ny, d.slack = n, ny, 0
    d.f_kind, d.g_kind, d.c_kind, d.D_kind = L.BZ_F_CALLBACK, L.BZ_G_CALLBACK, L.BZ_C_CALLBACK, L.BZ_D_CALLBACK
    nan = float("nan")
    thunks = [L.F_GRADIENT_FN(guard(f_gradient, nan)), L.G_PROX_FN(guard(g_prox, nan)), L.C_EVAL_FN(guard(c_eval, None)),
              L.C_JTPROD_FN(guard(c_jtprod, None)), L.D_PROJ_FN(guard(d_proj, None))]
    d.cb_f_gradient, d.cb_g_prox, d.cb_c_eval, d.cb_c_jtprod, d.cb_D_proj = thunks
    return d, [f, g, c, D, thunks, errors]


def lower(f, g, c, D, n, ny, dtype, slack=False):
    """(f, g, c, D) -> (ProblemDesc, keepalive list).  Raises UnsupportedOracle.
    slack=True: the ALS form on xs = [x; s] (src/utilities/auglagfunslack.jl)."""
    dtype = np.dtype(dtype)
    if dtype in (np.float64, np.float32) and not (_LOWERED_F(f) and _LOWERED_G(g) and _LOWERED_C(c) and _LOWERED_D(D)):
        return lower_generic(f, g, c, D, n, ny, dtype, slack)
    if dtype == np.float64:
        code = L.BZ_F64
    elif dtype == np.float32:
        code = L.BZ_F32
    else:
        raise UnsupportedOracle(f"eltype {dtype} is not lowered (Float64/Float32 only)")
    d = L.ProblemDesc()
    keep = []
    d.dtype, d.n, d.ny = code, n, ny
    d.slack = 1 if slack else 0
    # split-layout pairwise set: every per-element parameter vector goes to the device in the interleaved
    # (adjacent-pair) order; device.Problem permutes the state vectors at the boundary with the same map
    perm = split_permutation(n) if (isinstance(D, PairwiseSet) and D.layout == "split") else None
    if perm is not None and not isinstance(f, (Zero, DiagQuadratic)):
        raise UnsupportedOracle("split-layout pairwise sets are lowered with an element-wise f (Zero, DiagQuadratic)")

    def ptr(a):
        if perm is not None and a.ndim == 1 and a.shape[0] == n:
            a = np.ascontiguousarray(a[perm])
        keep.append(a)
        return a.ctypes.data

    # f
    if isinstance(f, Zero):
        d.f_kind = L.BZ_F_ZERO
    elif isinstance(f, DiagQuadratic):
        d.f_kind = L.BZ_F_DIAG_QUADRATIC
        d.f_q = ptr(_vec(f.q, dtype, n, "q"))
        d.f_b = ptr(_vec(f.b, dtype, n, "b"))
    elif isinstance(f, LeastSquares):
        d.f_kind = L.BZ_F_LEAST_SQUARES
        if f.A.shape[1] != n:
            raise ValueError(f"A must have {n} columns")
        d.f_A = ptr(np.ascontiguousarray(f.A, dtype=dtype))
        d.f_rows = f.A.shape[0]
        d.f_b = ptr(_vec(f.b, dtype, f.A.shape[0], "b"))
    elif isinstance(f, Quadratic):
        d.f_kind = L.BZ_F_QUADRATIC
        if f.Q.shape != (n, n):
            raise ValueError(f"Q must be {n}-by-{n}")
        d.f_A = ptr(np.ascontiguousarray(f.Q, dtype=dtype))
        d.f_rows = n
        d.f_b = ptr(_vec(f.q, dtype, n, "q"))
    elif isinstance(f, Stencil5ptQuadratic):
        d.f_kind = L.BZ_F_STENCIL5
        d.f_grid_nx, d.f_grid_ny = f.nx, f.ny
        d.f_b = ptr(_vec(f.b, dtype, n, "b"))
    else:
        raise UnsupportedOracle(f"f of type {type(f).__name__} is not lowered to the device")
    # g
    if isinstance(g, (Zero, IndFree)):
        d.g_kind = L.BZ_G_ZERO
    elif isinstance(g, NormL1):
        d.g_kind, d.g_lambda = L.BZ_G_NORM_L1, g.lam
    elif isinstance(g, NormL1Nonneg):
        d.g_kind, d.g_lambda = L.BZ_G_NORM_L1_NONNEG, g.lam
    elif isinstance(g, NormL1Box):
        d.g_kind, d.g_lambda = L.BZ_G_NORM_L1_BOX, g.lam
        d.g_u = ptr(_vec(g.u, dtype, n, "u"))
    elif isinstance(g, NormL0Box):
        d.g_kind, d.g_lambda = L.BZ_G_NORM_L0_BOX, g.lam
        d.g_u = ptr(_vec(g.u, dtype, n, "u"))
    elif isinstance(g, NormLpPowerNonneg):
        d.g_kind, d.g_lambda, d.g_p = L.BZ_G_NORM_LP_NONNEG, g.alpha, g.p
    elif isinstance(g, NormLpPowerBox):
        d.g_kind, d.g_lambda, d.g_p = L.BZ_G_NORM_LP_BOX, g.alpha, g.p
        d.g_u = ptr(_vec(g.u, dtype, n, "u"))
    elif isinstance(g, IndBox):
        d.g_kind = L.BZ_G_IND_BOX
        if np.ndim(g.lb) == 0:
            d.g_lo = float(g.lb)
        else:
            d.g_lo_vec = ptr(_vec(g.lb, dtype, n, "lb"))
        if np.ndim(g.ub) == 0:
            d.g_hi = float(g.ub)
        else:
            d.g_hi_vec = ptr(_vec(g.ub, dtype, n, "ub"))
    else:
        raise UnsupportedOracle(f"g of type {type(g).__name__} is not lowered to the device")
    # c
    if isinstance(c, IdentityFunction):
        d.c_kind = L.BZ_C_IDENTITY
        if ny != n:
            raise ValueError("IdentityFunction requires length(y0) == length(x0)")
    elif isinstance(c, DenseAffine):
        d.c_kind = L.BZ_C_DENSE_AFFINE
        if c.A.shape != (ny, n):
            raise ValueError(f"A must be {ny}-by-{n}")
        A = np.ascontiguousarray(c.A, dtype=dtype)
        d.c_A = ptr(A)
        d.c_b = ptr(_vec(c.b, dtype, ny, "b"))
    else:
        raise UnsupportedOracle(f"c of type {type(c).__name__} is not lowered to the device")
    # D
    if isinstance(D, ZeroSet):
        d.D_kind = L.BZ_D_ZERO
    elif isinstance(D, FreeSet):
        d.D_kind = L.BZ_D_FREE
    elif isinstance(D, IndicatorSet) and isinstance(D.f, IndBox):
        d.D_kind = L.BZ_D_BOX
        if np.ndim(D.f.lb) == 0:
            d.D_lo = float(D.f.lb)
        else:
            d.D_lo_vec = ptr(_vec(D.f.lb, dtype, ny, "lb"))
        if np.ndim(D.f.ub) == 0:
            d.D_hi = float(D.f.ub)
        else:
            d.D_hi_vec = ptr(_vec(D.f.ub, dtype, ny, "ub"))
    elif isinstance(D, IndicatorSet) and isinstance(D.f, IndFree):
        d.D_kind = L.BZ_D_FREE
    elif isinstance(D, PairwiseSet):
        d.D_kind = {"vc": L.BZ_D_VC_PAIRS, "cc": L.BZ_D_CC_PAIRS, "eitheror": L.BZ_D_EITHEROR_PAIRS,
                    "xor": L.BZ_D_XOR_PAIRS}[D.kind]
        if ny % 2:
            raise ValueError("pairwise sets need an even number of constraints")
    else:
        raise UnsupportedOracle(f"D of type {type(D).__name__} is not lowered to the device")
    return d, keep
