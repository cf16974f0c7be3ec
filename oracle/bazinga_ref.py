"""CPU restatement (numpy) of the Bazinga.alps -> PANOCplus hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product path is the HIP library
behind ``include/bazinga_hip.h``.

PARITY STATUS
-------------
* In-tree reference code (Bazinga.jl) is restated line by line, keeping the
  operation order of the Julia broadcasts:
    - ``alps``                     src/algorithms/alps.jl:7-117
    - ``AugLagFun``                src/utilities/auglagfun.jl:11-101
    - ``NonsmoothCostFun``         src/utilities/nonsmoothcostfun.jl:1-22
    - safeguards                   src/utilities/safeguards.jl:2-18
    - ``ZeroSet/FreeSet/IndicatorSet``  src/projections/{zeroSet,freeSet,indicatorSet}.jl
    - either-or projection         src/projections/orConstraints.jl:7-17
    - ``Zero/NormL1Nonneg/NormL1Box``   src/proxoperators/{zero,normL1Nonneg,normL1Box}.jl
    - ``IdentityFunction``         test/definitions/identityFunction.jl:3-13
    - rosenbrock oracles           demo/rosenbrock.jl:39-80
    - basis-pursuit constraint     demo/basispursuit.jl:38-49
* The inner solver (PANOCplus, LBFGS, IterativeAlgorithm, f_model,
  backtrack_stepsize!, lower_bound_smoothness_constant) lives in the
  third-party package ProximalAlgorithms.jl, which Bazinga's Project.toml lists
  WITHOUT a version bound and which is absent from /root/reference
  (call sites: src/algorithms/alps.jl:5,64,66).  It is restated here from the
  published algorithm (De Marchi & Themelis, "Proximal gradient algorithms
  under local Lipschitz gradient continuity", JOTA 2022, Alg. PANOC+) and the
  package's public behaviour (0.5.x series API as used by Bazinga).  The same
  holds for the ProximalOperators.jl functions used by the reference's tests
  (LeastSquares, Quadratic, NormL1, IndBox, IndFree, Zero).
* Pinned by the reference's own tests (tests/test_oracle_kat.py):
    - lasso KAT                    test/problems/test_verbose.jl:7-13,29,42-44
    - nonconvex QP property        test/problems/test_nonconvex_qp.jl:9-36,56-105
    - rosenbrock minimiser (0,0)   demo/rosenbrock.jl:85-136,186
  These pin the SOLUTION.  Per-iterate values of PANOCplus are pinned by
  nothing the reference holds, and Julia is not available in the build
  container: ITERATE-LEVEL PARITY IS "PARITY UNPINNED" (the restatement below
  defines it).

All vectors are 1-D numpy arrays of dtype T (float64 or float32); scalars are
kept in T as Julia would.
"""
from __future__ import annotations

import math
import time
import warnings
from dataclasses import dataclass, field

import numpy as np

# ---------------------------------------------------------------------------
# Reductions over the decision / constraint vectors go through REDUCER so that a test
# can run the SAME restatement on a shard of x and fold the partial scalars across
# ranks (tests/test_sharded_cpu.py, world_size 2 over gloo): the multi-GPU design
# shards x and exchanges scalars only.
# ---------------------------------------------------------------------------
class LocalReducer:
    """Single-process reductions (numpy pairwise sum / BLAS dot)."""

    def sum(self, v):
        return np.sum(v)

    def dot(self, a, b):
        return np.dot(a, b)

    def max(self, v):
        return np.max(v) if v.size else v.dtype.type(0)

    def any(self, m):
        return bool(np.any(m))


REDUCER = LocalReducer()


def set_reducer(r):
    global REDUCER
    REDUCER = r if r is not None else LocalReducer()


def _sum(v):
    return REDUCER.sum(v)


def _dot(a, b):
    return REDUCER.dot(a, b)


def _max(v):
    return REDUCER.max(v)


# ---------------------------------------------------------------------------
# oracle protocol (README.md:17-20, src/Bazinga.jl:11-16)
#   f(x) -> value ; f.gradient(out, x) -> value          [gradient!]
#   g.prox(z, x, gamma) -> g(z)                            [prox!]
#   c.eval(cx, x) ; c.jtprod(jtv, x, v)                    [eval!, jtprod!]
#   D.proj(s, v)                                           [proj!]
# ---------------------------------------------------------------------------


# ----------------------------- D: closed sets ------------------------------
class ZeroSet:
    """src/projections/zeroSet.jl:8-20"""

    def proj(self, y, x):
        y[...] = 0
        return None


class FreeSet:
    """src/projections/freeSet.jl:8-20"""

    def proj(self, y, x):
        y[...] = x
        return None


class IndicatorSet:
    """src/projections/indicatorSet.jl:4-11 : proj! forwards to prox!(z, f.f, x)
    (no gamma).  ``ClosedSet(f) = IndicatorSet(f)`` (src/Bazinga.jl:18)."""

    def __init__(self, f):
        self.f = f

    def proj(self, z, x):
        self.f.prox(z, x, 1.0)
        return None


def ClosedSet(f):
    return IndicatorSet(f)


def project_onto_EITHEROR_set(z, x):
    """src/projections/orConstraints.jl:7-17"""
    z[...] = x
    if x[0] < 0 and x[1] < 0:
        if x[0] > x[1]:
            z[0] = 0
        else:
            z[1] = 0
    return None


def project_onto_XOR_set(z, x):
    """src/projections/orConstraints.jl:24-36"""
    z[...] = x
    if x[0] * x[1] > 0:
        if x[0] > x[1]:
            z[0] = max(0, x[0])
            z[1] = min(0, x[1])
        else:
            z[0] = min(0, x[0])
            z[1] = max(0, x[1])
    return None


def project_onto_VC_set(z, x):
    """src/projections/vanishingConstraints.jl:27-46"""
    z[...] = 0
    if x[0] <= 0:
        z[1] = x[1]
    else:
        if x[1] >= 0:
            z[...] = x
        else:
            if x[0] + x[1] > 0:
                z[0] = x[0]
            elif x[0] + x[1] < 0:
                z[1] = x[1]
            else:               # set-valued case
                z[1] = x[1]
    return None


def project_onto_CC_set(z, x):
    """src/projections/complementarityConstraints.jl:8-20"""
    if x[0] > 0 and x[1] > 0:
        z[...] = x
        if x[1] > x[0]:
            z[0] = 0
        else:
            z[1] = 0
    else:
        z[...] = np.maximum(0, x)
    return None


class PairwiseSet:
    """D as the demos build it from the 2-element projections: the same set over every ADJACENT pair
    (demo/mpvca.jl:105-106,147-148 ; demo/eitheror.jl:79-88,123-130)."""
    _P = {"vc": project_onto_VC_set, "cc": project_onto_CC_set, "eitheror": project_onto_EITHEROR_set,
          "xor": project_onto_XOR_set}

    def __init__(self, kind, layout="adjacent"):
        self.kind = kind
        self.layout = layout
        self._proj = self._P[kind]

    def proj(self, z, x):
        if self.layout == "split":
            # demo/obstacle.jl:151-168 (SetObstacleRed): the pairs are (x[i], x[i + N]), i = 1..N
            N = x.shape[0] // 2
            for i in range(N):
                zz = np.empty(2, x.dtype)
                self._proj(zz, np.array([x[i], x[i + N]], x.dtype))
                z[i], z[i + N] = zz[0], zz[1]
            return None
        for j in range(0, x.shape[0], 2):
            self._proj(z[j:j + 2], x[j:j + 2])
        return None


class SetRosenbrock:
    """demo/rosenbrock.jl:76-80"""

    def proj(self, z, cx):
        project_onto_EITHEROR_set(z, cx)
        return None


# ------------------------- g: proximable functions -------------------------
class Zero:
    """src/proxoperators/zero.jl:11-25 (also ProximalOperators.Zero)."""

    def __call__(self, x):
        return 0.0

    def gradient(self, dfx, x):
        dfx[...] = 0.0
        return 0.0

    def prox(self, y, x, gamma):
        y[...] = x
        return 0.0


class IndFree(Zero):
    """ProximalOperators.IndFree (test_nonconvex_qp.jl:39): prox = identity, value 0."""


class NormL1:
    """ProximalOperators.NormL1(lambda) (test_verbose.jl:23, basispursuit.jl:63):
    two-sided soft threshold, returns lambda*||z||_1.  lambda may be a scalar
    or an array (per-coordinate weights)."""

    def __init__(self, lam=1.0):
        self.lam = lam

    def __call__(self, x):
        return np.sum(np.abs(self.lam * x)) if np.ndim(self.lam) else self.lam * np.sum(np.abs(x))

    def prox(self, y, x, gamma):
        T = x.dtype.type
        if np.ndim(self.lam) == 0:
            gl = T(gamma) * T(self.lam)
            # y[i] = x[i] + (x[i] <= -gl ? gl : (x[i] >= gl ? -gl : -x[i]))
            y[...] = x + np.where(x <= -gl, gl, np.where(x >= gl, -gl, -x))
            return T(self.lam) * _sum(np.abs(y))
        gl = T(gamma) * np.asarray(self.lam, dtype=x.dtype)
        y[...] = x + np.where(x <= -gl, gl, np.where(x >= gl, -gl, -x))
        return _sum(np.asarray(self.lam, dtype=x.dtype) * np.abs(y))


class NormL1Nonneg:
    """src/proxoperators/normL1Nonneg.jl:29-42"""

    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("λ must be nonnegative")
        self.lam = lam

    def __call__(self, x):
        return self.lam * np.sum(np.abs(x))

    def prox(self, y, x, gamma):
        T = x.dtype.type
        gl = T(gamma) * T(self.lam)
        m = x >= gl
        y[...] = np.where(m, x - gl, T(0))
        return T(self.lam) * _sum(y)


class NormL1Box:
    """src/proxoperators/normL1Box.jl:30-39 ; u is a nonnegative vector."""

    def __init__(self, lam=1.0, *, u):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        if np.any(np.asarray(u) < 0):
            raise ValueError("vector u must have nonnegative entries")
        self.lam = lam
        self.u = np.asarray(u)

    def __call__(self, x):
        return self.lam * np.sum(np.abs(x))

    def prox(self, y, x, gamma):
        T = x.dtype.type
        gl = T(gamma) * T(self.lam)
        y[...] = np.maximum(T(0), np.minimum(x - gl, self.u.astype(x.dtype, copy=False)))
        return T(self.lam) * _sum(y)


class NormL0Box:
    """src/proxoperators/normL0Box.jl:12-58"""

    def __init__(self, lam=1.0, *, u):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        if np.any(np.asarray(u) < 0):
            raise ValueError("vector u must have nonnegative entries")
        self.lam = lam
        self.u = np.asarray(u)

    def __call__(self, x):
        return self.lam * x.dtype.type(np.count_nonzero(x))

    def prox(self, y, x, gamma):
        T = x.dtype.type
        gl2 = T(gamma) * T(self.lam)
        u = self.u.astype(x.dtype, copy=False)
        big = x > np.sqrt(gl2)
        above = x > u
        keep_above = x * x > gl2 + (u - x) ** 2
        nz = (u != 0) & big & (~above | keep_above)
        y[...] = np.where(nz, x, T(0))
        return T(self.lam) * _sum(nz.astype(x.dtype))


def _solve_lp_quasi_norm(x, p, a, gamma, u=None):
    """Vectorised transcription of solve_lp_quasi_norm_subproblem_nonneg / _box
    (src/proxoperators/normLpNonneg.jl:44-84, normLpBox.jl:47-97): per element the same Newton
    iteration with the same start, stopping rule (|dphi| <= 1e-12, at most 1000 steps) and the same
    global-minimum tests; elements that have stopped are frozen."""
    T = x.dtype.type
    box = u is not None
    alpha = T(a) * T(gamma)
    p = T(p)
    out = np.zeros_like(x)
    ap = alpha * p
    zbar = (T(1) / (ap * (T(1) - p))) ** (T(1) / (p - T(2)))
    psi = zbar + ap * zbar ** (p - T(1))
    act = (x > 0) & (psi < x)
    if box:
        act &= (u != 0)
    if not np.any(act):
        return out
    xa = x[act]
    z = np.full_like(xa, zbar + (T(0.1) if box else T(1)))
    run = np.ones(xa.shape, bool)
    for _ in range(1000):
        dphi = z - xa + ap * z ** (p - T(1))
        run &= ~(np.abs(dphi) <= 1e-12)
        if not np.any(run):
            break
        ddphi = T(1) + ap * (p - T(1)) * z ** (p - T(2))
        z = np.where(run, z - dphi / ddphi, z)
    phi0 = T(0.5) * (xa * xa)
    phiz = T(0.5) * (z - xa) ** 2 + alpha * z ** p
    res = np.where(phi0 <= phiz, T(0), z)
    if box:
        ua = u[act]
        phiu = T(0.5) * (ua - xa) ** 2 + alpha * ua ** p
        over = (res != 0) & (z > ua)
        res = np.where(over, np.where(phiu < phi0, ua, T(0)), res)
    out[act] = res
    return out


class NormLpPowerNonneg:
    """src/proxoperators/normLpNonneg.jl:14-40"""

    def __init__(self, p, *, alpha=1.0):
        if p <= 0:
            raise ValueError("p must be positive")
        if p >= 1:
            raise ValueError("p must be smaller than one")
        if alpha < 0:
            raise ValueError("alpha must be nonnegative")
        self.p, self.alpha = p, alpha

    def __call__(self, x):
        return self.alpha * np.sum(x ** self.p)

    def prox(self, y, x, gamma):
        T = x.dtype.type
        y[...] = _solve_lp_quasi_norm(x, self.p, self.alpha, gamma)
        return T(self.alpha) * _sum(y ** T(self.p))


class NormLpPowerBox:
    """src/proxoperators/normLpBox.jl:11-45"""

    def __init__(self, p, alpha=1.0, *, u):
        if p <= 0:
            raise ValueError("p must be positive")
        if p >= 1:
            raise ValueError("p must be smaller than one")
        if alpha < 0:
            raise ValueError("alpha must be nonnegative")
        if np.any(np.asarray(u) < 0):
            raise ValueError("vector u must have nonnegative entries")
        self.p, self.alpha, self.u = p, alpha, np.asarray(u)

    def __call__(self, x):
        return self.alpha * np.sum(x ** self.p)

    def prox(self, y, x, gamma):
        T = x.dtype.type
        y[...] = _solve_lp_quasi_norm(x, self.p, self.alpha, gamma, self.u.astype(x.dtype, copy=False))
        return T(self.alpha) * _sum(y ** T(self.p))


class IndBox:
    """ProximalOperators.IndBox(lb, ub) (test_nonconvex_qp.jl:15): prox = clamp,
    value 0.  lb/ub scalar or array."""

    def __init__(self, lb, ub):
        self.lb = lb
        self.ub = ub

    def __call__(self, x):
        return 0.0 if np.all((x >= self.lb) & (x <= self.ub)) else math.inf

    def prox(self, y, x, gamma=1.0):
        # if x<lb: lb elif x>ub: ub else x
        y[...] = np.where(x < self.lb, self.lb, np.where(x > self.ub, self.ub, x))
        return x.dtype.type(0)


class NonsmoothCostRosenbrock:
    """demo/rosenbrock.jl:52-64"""

    def __init__(self, lam):
        self.lam = lam

    def prox(self, y, x, gamma):
        gl = gamma * self.lam
        y[0] = 0.0 if abs(x[0]) <= gl else np.sign(x[0]) * (abs(x[0]) - gl)
        y[1] = x[1]
        return self.lam * abs(y[0])


# --------------------------- f: smooth functions ---------------------------
class DiagQuadratic:
    """Build-defined structured special case of ProximalOperators.Quadratic
    (SURVEY.md §8(a) a11, §8(d) cfg 2):  f(x) = sum_i x_i*(0.5*q_i*x_i - b_i),
    grad_i = q_i*x_i - b_i.  Per-element operation order is the contract the
    HIP kernel mirrors bit for bit:
        qx = q*x ; grad = qx - b ; term = x*(0.5*qx - b)."""

    def __init__(self, q, b):
        self.q = np.asarray(q)
        self.b = np.asarray(b)

    def __call__(self, x):
        T = x.dtype.type
        qx = self.q * x
        return _sum(x * (T(0.5) * qx - self.b))

    def gradient(self, dfx, x):
        T = x.dtype.type
        qx = self.q * x
        dfx[...] = qx - self.b
        return _sum(x * (T(0.5) * qx - self.b))


class Stencil5ptQuadratic:
    """Build-defined (SURVEY.md §8(d) cfg 3):  f(x) = 0.5 x'A_h x - b'x on an
    nx-by-ny grid (row-major, index = i*ny + j), A_h = 5-point Laplacian
    (4,-1,-1,-1,-1) with homogeneous Dirichlet halo.
    Per-element order (the HIP kernel mirrors it):
        Ax = ((((4*x_c - x_w) - x_e) - x_n) - x_s)   (missing neighbours = 0)
        grad = Ax - b ; term = x_c*(0.5*Ax - b)."""

    def __init__(self, nx, ny, b):
        self.nx, self.ny = int(nx), int(ny)
        self.b = np.asarray(b)

    def _Ax(self, x):
        T = x.dtype.type
        X = x.reshape(self.nx, self.ny)
        P = np.zeros((self.nx + 2, self.ny + 2), dtype=x.dtype)
        P[1:-1, 1:-1] = X
        A = T(4) * X
        A = A - P[1:-1, :-2]   # west  (j-1)
        A = A - P[1:-1, 2:]    # east  (j+1)
        A = A - P[:-2, 1:-1]   # north (i-1)
        A = A - P[2:, 1:-1]    # south (i+1)
        return A.reshape(-1)

    def __call__(self, x):
        T = x.dtype.type
        Ax = self._Ax(x)
        return np.sum(x * (T(0.5) * Ax - self.b))

    def gradient(self, dfx, x):
        T = x.dtype.type
        Ax = self._Ax(x)
        dfx[...] = Ax - self.b
        return np.sum(x * (T(0.5) * Ax - self.b))


class Quadratic:
    """ProximalOperators.Quadratic(Q, q): f = 0.5 x'Qx + q'x (test_nonconvex_qp.jl:14)."""

    def __init__(self, Q, q):
        self.Q = np.asarray(Q)
        self.q = np.asarray(q)

    def __call__(self, x):
        T = x.dtype.type
        return T(0.5) * np.dot(x, self.Q @ x) + np.dot(x, self.q)

    def gradient(self, y, x):
        T = x.dtype.type
        y[...] = self.Q @ x
        fx = T(0.5) * np.dot(x, y)
        y += self.q
        return fx + np.dot(x, self.q)


class LeastSquares:
    """ProximalOperators.LeastSquares(A, b): f = 0.5||Ax-b||^2 (test_verbose.jl:22)."""

    def __init__(self, A, b, lam=1.0):
        self.A = np.asarray(A)
        self.b = np.asarray(b)
        self.lam = lam

    def __call__(self, x):
        T = x.dtype.type
        r = self.A @ x - self.b
        return T(self.lam / 2) * np.dot(r, r)

    def gradient(self, y, x):
        T = x.dtype.type
        r = self.A @ x - self.b
        y[...] = T(self.lam) * (self.A.T @ r)
        return T(self.lam / 2) * np.dot(r, r)


class SmoothCostRosenbrock:
    """demo/rosenbrock.jl:39-50"""

    def __init__(self, w):
        self.w = w

    def __call__(self, x):
        return self.w * (x[1] + 1 - (x[0] + 1) ** 2) ** 2

    def gradient(self, dfx, x):
        tmp = x[1] + 1 - (x[0] + 1) ** 2
        dfx[0] = -4 * self.w * tmp * (x[0] + 1)
        dfx[1] = 2 * self.w * tmp
        return self.w * tmp ** 2


# ------------------------------ c: constraints -----------------------------
class IdentityFunction:
    """test/definitions/identityFunction.jl:3-13"""

    def eval(self, fx, x):
        fx[...] = x
        return None

    def jtprod(self, jtv, x, v):
        jtv[...] = v
        return None


class DenseAffine:
    """demo/basispursuit.jl:38-49 (ConstraintBasisPursuit): c(x) = A x - b."""

    def __init__(self, A, b):
        self.A = np.asarray(A)
        self.b = np.asarray(b)

    def eval(self, cx, x):
        cx[...] = self.A @ x - self.b
        return None

    def jtprod(self, jtv, x, v):
        jtv[...] = self.A.T @ v
        return None


class ConstraintRosenbrock:
    """demo/rosenbrock.jl:66-74"""

    def eval(self, cx, x):
        cx[...] = [-x[0] - x[1], x[1] - x[0]]
        return None

    def jtprod(self, jtv, x, v):
        jtv[...] = [-v[0] - v[1], v[1] - v[0]]
        return None


# ------------------------- src/utilities/safeguards.jl ---------------------
def default_dual_safeguard(y, cx=None):
    """safeguards.jl:2-10"""
    y[...] = np.maximum(-1e20, np.minimum(y, 1e20))
    return None


def default_penalty_parameter(mu, cx, proj_cx, objx):
    """safeguards.jl:13-18 (Float64 literals; result stored back into mu's dtype)."""
    # each Julia statement computes in Float64 (literal promotion) and rounds into mu's eltype
    d2 = ((cx - proj_cx) ** 2).astype(np.float64, copy=False)      # (cx .- proj_cx).^2 in T
    mu[...] = (np.maximum(1.0, 0.5 * d2) / max(1.0, float(objx))).astype(mu.dtype, copy=False)
    mu[...] = (mu.astype(np.float64, copy=False) * 0.1).astype(mu.dtype, copy=False)
    mu[...] = np.maximum(1e-8, np.minimum(mu.astype(np.float64, copy=False), 1e8)).astype(mu.dtype, copy=False)
    return None


# ----------------------- src/utilities/nonsmoothcostfun.jl -----------------
class NonsmoothCostFun:
    """nonsmoothcostfun.jl:1-22"""

    def __init__(self, g):
        self.g = g
        self.gamma = 0.0
        self.gz = 0.0

    def prox(self, z, x, gamma):
        gz = self.g.prox(z, x, gamma)
        self.gamma = gamma
        self.gz = gz
        return gz


# -------------------------- src/utilities/auglagfun.jl ---------------------
class AugLagFun:
    """auglagfun.jl:11-101.  L(x) = f(x) + 1/(2mu) dist_D^2(c(x)+mu y) - mu/2 ||y||^2"""

    def __init__(self, f, c, D, mu, y, x):
        if REDUCER.any(mu <= 0):                              # :33-34
            raise ValueError("parameters `mu` must be positive")
        T = x.dtype.type
        self.f, self.c, self.D = f, c, D
        self.mu, self.y = mu, y                               # aliases, as in Julia
        self.muy = mu * y                                     # :36
        self.musqy = T(0.5) * _sum(self.muy * y)              # :37
        self.cx = np.empty_like(y)
        self.s = np.empty_like(y)
        self.yupd = np.empty_like(y)
        self.fx = T(0)
        self.dfx = np.empty_like(x)
        self.jtv = np.empty_like(x)
        self.n_eval = 0
        self.n_grad = 0

    def __call__(self, x):                                    # :58-69
        T = x.dtype.type
        self.n_eval += 1
        self.c.eval(self.cx, x)
        np.add(self.cx, self.muy, out=self.yupd)
        self.D.proj(self.s, self.yupd)
        self.yupd -= self.s
        lx = T(0.5) * _sum(self.yupd ** 2 / self.mu)
        self.yupd /= self.mu
        self.fx = self.f(x)
        lx += self.fx
        lx -= self.musqy
        return lx

    def gradient(self, dlx, x):                               # :73-86
        T = x.dtype.type
        self.n_grad += 1
        self.c.eval(self.cx, x)                               # cx
        np.add(self.cx, self.muy, out=self.yupd)              # cx + mu.*y
        self.D.proj(self.s, self.yupd)                        # s
        self.yupd -= self.s                                   # cx + mu.*y - s
        lx = T(0.5) * _sum(self.yupd ** 2 / self.mu)
        self.yupd /= self.mu                                  # yupd
        self.fx = self.f.gradient(self.dfx, x)                # fx, dfx
        lx += self.fx
        lx -= self.musqy
        self.c.jtprod(self.jtv, x, self.yupd)                 # jtv
        np.add(self.dfx, self.jtv, out=dlx)                   # dlx
        return lx


def AugLagUpdate(al: AugLagFun, mu, y):                       # :91-101
    if REDUCER.any(mu <= 0):
        raise ValueError("parameters `mu` must be positive")
    T = y.dtype.type
    al.mu[...] = mu
    al.y[...] = y
    al.muy[...] = al.mu * al.y
    al.musqy = T(0.5) * _sum(al.muy * al.y)
    return None


# ---------------------------------------------------------------------------
#  ProximalAlgorithms.jl restatement (external, unpinned — see module header)
# ---------------------------------------------------------------------------
class LBFGS:
    """``LBFGS(memory)`` direction factory.  compact=True selects the compact (Byrd-Nocedal-Schnabel)
    representation of the SAME operator — an alternate evaluation order, see LBFGSCompactOperator."""

    def __init__(self, memory=5, compact=False):
        self.memory = int(memory)
        self.compact = bool(compact)


class NoAcceleration:
    """``NoAcceleration()`` directions (demo/rosenbrock.jl:96-97): d = -res, nothing to update."""


class _NoAccelOperator:
    currmem = 0
    H = 1.0

    def mul(self, d, v):
        d[...] = v
        return d

    def update(self, s, y):
        return _dot(s, y)

    def reset(self):
        pass


class AndersonAcceleration:
    """``AndersonAcceleration(n)`` directions (demo/rosenbrock.jl:100-101).  EXTERNAL, UNPINNED, restated from the published
    method (type-II Anderson acceleration: Walker & Ni, SIAM J. Numer. Anal. 49 (2011); Fang & Saad, Numer. Linear
    Algebra Appl. 16 (2009)): over the last n pairs (s_i, y_i), every pair kept,
        H = I + (S - Y) Y^+ ,   d = H v = v + (S - Y) (Y \\ v)
    which satisfies the multisecant equations H y_i = s_i.  PARITY UNPINNED (no reference output pins it)."""

    def __init__(self, memory=5):
        self.memory = int(memory)


class _AndersonOperator:
    H = 1.0

    def __init__(self, M, x):
        self.M = M
        self.S, self.Y = [], []

    @property
    def currmem(self):
        return len(self.S)

    def update(self, s, y):
        self.S.append(s.copy())
        self.Y.append(y.copy())
        if len(self.S) > self.M:
            self.S.pop(0)
            self.Y.pop(0)
        return _dot(s, y)

    def reset(self):
        self.S, self.Y = [], []

    def mul(self, d, v):
        d[...] = v
        if self.S:
            Ym = np.stack(self.Y, axis=1)
            Sm = np.stack(self.S, axis=1)
            a = np.linalg.lstsq(Ym.astype(np.float64), v.astype(np.float64), rcond=None)[0].astype(v.dtype)
            d += (Sm - Ym) @ a
        return d


class Broyden:
    """``Broyden(; theta_bar = 0.2)`` directions (demo/rosenbrock.jl:98-99).  EXTERNAL, UNPINNED, restated from the
    published modified Broyden update of the PANOC / SuperMann papers (Themelis & Patrinos, IEEE TAC 64 (2019), §VI-A):
        delta = <H y, s> / <s, s> ;  theta = 1 if |delta| >= theta_bar else (1 - sgn(delta) theta_bar) / (1 - delta), sgn(0) = 1
        H <- H + (s - H y) (s'H) / <s, (1/theta - 1) s + H y>
    on a dense operator started (and reset) at the identity.  PARITY UNPINNED."""

    def __init__(self, theta_bar=0.2):
        self.theta_bar = float(theta_bar)


class _BroydenOperator:
    currmem = 0

    def __init__(self, theta_bar, x):
        self.theta_bar = x.dtype.type(theta_bar)
        self.n = x.shape[0]
        self.Hm = np.eye(self.n, dtype=x.dtype)
        self.H = 1.0

    def mul(self, d, v):
        d[...] = self.Hm @ v
        return d

    def reset(self):
        self.Hm = np.eye(self.n, dtype=self.Hm.dtype)

    def update(self, s, y):
        T = s.dtype.type
        Hy = self.Hm @ y
        sH = s @ self.Hm
        ss = _dot(s, s)
        hys = _dot(Hy, s)
        if not ss > 0:
            return _dot(s, y)
        delta = hys / ss
        theta = T(1)
        if abs(delta) < self.theta_bar:
            sg = T(1) if delta >= 0 else T(-1)
            theta = (T(1) - sg * self.theta_bar) / (T(1) - delta)
        denom = (T(1) / theta - T(1)) * ss + hys
        if denom == 0 or denom != denom:
            return _dot(s, y)
        self.Hm = self.Hm + np.outer((s - Hy) * (T(1) / denom), sH)
        return _dot(s, y)


class LBFGSOperator:
    """Two-loop L-BFGS operator with ring buffer of M pairs.
    update!: insert iff <s,y> > 0, H = ys/yty of the newest pair.
    reset!:  currmem = curridx = 0, H = 1.
    mul!:    d = v; loop1 newest->oldest; d *= H; loop2 oldest->newest."""

    def __init__(self, M, x):
        self.M = M
        T = x.dtype.type
        self.currmem = 0
        self.curridx = 0          # 1-based like Julia; 0 = empty
        self.s_M = [np.zeros_like(x) for _ in range(M)]
        self.y_M = [np.zeros_like(x) for _ in range(M)]
        self.ys_M = np.zeros(M, dtype=x.dtype)
        self.alphas = np.zeros(M, dtype=x.dtype)
        self.H = T(1)

    def update(self, s, y):
        ys = _dot(s, y)
        if ys > 0:
            self.curridx += 1
            if self.curridx > self.M:
                self.curridx = 1
            self.currmem += 1
            if self.currmem > self.M:
                self.currmem = self.M
            self.ys_M[self.curridx - 1] = ys
            self.s_M[self.curridx - 1][...] = s
            self.y_M[self.curridx - 1][...] = y
            yty = _dot(y, y)
            self.H = ys / yty
        return ys

    def reset(self):
        self.currmem, self.curridx = 0, 0
        self.H = self.ys_M.dtype.type(1)

    def mul(self, d, v):
        d[...] = v
        idx = self.curridx
        for _ in range(self.currmem):                       # loop1
            a = _dot(self.s_M[idx - 1], d) / self.ys_M[idx - 1]
            self.alphas[idx - 1] = a
            d -= a * self.y_M[idx - 1]
            idx -= 1
            if idx == 0:
                idx = self.M
        d *= self.H
        for _ in range(self.currmem):                       # loop2
            idx += 1
            if idx > self.M:
                idx = 1
            beta = _dot(self.y_M[idx - 1], d) / self.ys_M[idx - 1]
            d += (self.alphas[idx - 1] - beta) * self.s_M[idx - 1]
        return d


class LBFGSCompactOperator:
    """The L-BFGS operator of LBFGSOperator in its compact form (Byrd, Nocedal, Schnabel 1994):
        H = H0 I + [S, H0 Y] [[R^-T (D + H0 Y'Y) R^-1, -R^-T], [-R^-1, 0]] [S'; H0 Y']
    with pairs ordered oldest -> newest, R = triu(S'Y), D = diag(S'Y), H0 = ys/yty of the newest pair.
    Mathematically identical to the two-loop recursion (same pairs, same H0, same insert-iff-<s,y>>0 rule)
    but ALL 2m inner products with v are independent: one reduction phase per application instead of 2m
    sequential ones — the form the multi-GPU path uses.  Rounding differs from the
    two-loop at the 1e-16 level; tests/test_oracle_kat.py shows the two forms agree to that level and then
    only drift within the restatement's own rounding sensitivity.  The operation order below is the
    contract the HIP kernels (k_gram_dots, k_fused_sep<COMPACT>) and the host code mirror."""

    def __init__(self, M, x):
        self.M = M
        self.S, self.Y = [], []                 # oldest -> newest
        self.SY = np.zeros((0, 0))              # s_i . y_j  (only i <= j is used)
        self.YY = np.zeros((0, 0))
        self.H = x.dtype.type(1)
        self.dtype = x.dtype

    @property
    def currmem(self):
        return len(self.S)

    def reset(self):
        self.S, self.Y = [], []
        self.SY = np.zeros((0, 0))
        self.YY = np.zeros((0, 0))
        self.H = self.dtype.type(1)

    def update(self, s, y):
        ys = _dot(s, y)
        if ys > 0:
            if len(self.S) == self.M:           # ring full: the oldest pair is overwritten
                self.S.pop(0); self.Y.pop(0)
                self.SY = self.SY[1:, 1:]
                self.YY = self.YY[1:, 1:]
            m = len(self.S)
            sy = np.array([float(_dot(self.S[i], y)) for i in range(m)] + [float(ys)])
            yty = _dot(y, y)
            yy = np.array([float(_dot(self.Y[i], y)) for i in range(m)] + [float(yty)])
            SY = np.zeros((m + 1, m + 1)); SY[:m, :m] = self.SY; SY[:, m] = sy
            YY = np.zeros((m + 1, m + 1)); YY[:m, :m] = self.YY; YY[:, m] = yy; YY[m, :] = yy
            self.SY, self.YY = SY, YY
            self.S.append(s.copy()); self.Y.append(y.copy())
            self.H = ys / yty
        return ys

    @staticmethod
    def coefficient_matrices(SY, YY, H0):
        """M1 = R^-T (D + H0 Y'Y) R^-1 and M2 = R^-1 in float64, explicit loops (fixed order)."""
        m = SY.shape[0]
        Ri = np.zeros((m, m))
        for j in range(m):                      # back-substitution, column j of R^-1
            Ri[j, j] = 1.0 / SY[j, j]
            for i in range(j - 1, -1, -1):
                acc = 0.0
                for k in range(i + 1, j + 1):
                    acc += SY[i, k] * Ri[k, j]
                Ri[i, j] = -acc / SY[i, i]
        B = np.zeros((m, m))
        for i in range(m):
            for j in range(m):
                B[i, j] = H0 * YY[i, j] + (SY[i, i] if i == j else 0.0)
        T1 = np.zeros((m, m))                   # T1 = B R^-1
        for i in range(m):
            for j in range(m):
                acc = 0.0
                for k in range(j + 1):
                    acc += B[i, k] * Ri[k, j]
                T1[i, j] = acc
        M1 = np.zeros((m, m))                   # M1 = R^-T T1
        for i in range(m):
            for j in range(m):
                acc = 0.0
                for k in range(i + 1):
                    acc += Ri[k, i] * T1[k, j]
                M1[i, j] = acc
        return M1, Ri

    def mul(self, d, v):
        T = self.dtype.type
        m = len(self.S)
        H0 = float(self.H)
        if m == 0:
            d[...] = T(H0) * v
            return d
        p = [float(_dot(self.S[i], v)) for i in range(m)]
        w = [float(_dot(self.Y[i], v)) for i in range(m)]
        M1, M2 = self.coefficient_matrices(self.SY, self.YY, H0)
        u1, u2 = [0.0] * m, [0.0] * m
        for i in range(m):
            a = 0.0
            for j in range(m):
                a += M1[i, j] * p[j]
            b = 0.0
            for j in range(m):
                b += M2[j, i] * w[j]
            u1[i] = a - H0 * b
            c = 0.0
            for j in range(m):
                c += M2[i, j] * p[j]
            u2[i] = -c
        d[...] = T(H0) * v
        for i in range(m):
            d += T(u1[i]) * self.S[i]
        for i in range(m):
            d += T(H0 * u2[i]) * self.Y[i]
        return d


def _norm(v):
    return np.sqrt(_dot(v, v))


def f_model(f_x, grad_f_x, res, L):
    """f_x - <grad, res> + (L/2) ||res||^2"""
    nr = _norm(res)
    return f_x - _dot(grad_f_x, res) + (L / 2) * (nr * nr)


def lower_bound_smoothness_constant(f, x, grad_f_x):
    """||grad f(x+1) - grad f(x)|| / ||(x+1) - x||   (A = I)."""
    xeps = x + x.dtype.type(1)
    g = np.empty_like(x)
    f.gradient(g, xeps)
    return _norm(g - grad_f_x) / _norm(xeps - x)


@dataclass
class PANOCplusState:
    x: np.ndarray
    f_x: float
    grad_f_x: np.ndarray
    gamma: float
    y: np.ndarray
    z: np.ndarray
    g_z: float
    res: np.ndarray
    H: LBFGSOperator
    tau: float = 0.0
    x_prev: np.ndarray = None
    res_prev: np.ndarray = None
    d: np.ndarray = None
    x_d: np.ndarray = None
    f_x_d: float = 0.0
    grad_f_x_d: np.ndarray = None
    z_curr: np.ndarray = None
    grad_f_z: np.ndarray = None
    # instrumentation (not part of the upstream state)
    n_backtracks: int = 0
    n_gamma_halvings: int = 0
    last_ys: float = 0.0
    f_z: float = 0.0


class PANOCplusIteration:
    """Restatement of ProximalAlgorithms.PANOCplusIteration with A = I."""

    def __init__(self, f, g, x0, alpha=0.95, beta=0.5, Lf=None, gamma=None, adaptive=None,
                 minimum_gamma=1e-7, max_backtracks=20, directions=None):
        self.f, self.g, self.x0 = f, g, x0
        T = x0.dtype.type
        self.T = T
        self.alpha, self.beta = T(alpha), T(beta)
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else T(alpha) / T(Lf))
        self.adaptive = (self.gamma is None) if adaptive is None else adaptive
        self.minimum_gamma = T(minimum_gamma)
        self.max_backtracks = int(max_backtracks)
        self.directions = directions if directions is not None else LBFGS(5)
        self.eps = np.finfo(x0.dtype).eps

    # -- helpers --------------------------------------------------------
    def _f_model(self, st):
        return f_model(st.f_x, st.grad_f_x, st.res, self.alpha / st.gamma)

    def _backtrack_stepsize(self, st):
        """backtrack_stepsize!(gamma, f, A, g, x, f_Ax, At_grad_f_Ax, y, z, g_z, res, Az, grad_f_Az)"""
        T = self.T
        gamma = st.gamma
        f_z_upp = f_model(st.f_x, st.grad_f_x, st.res, self.alpha / gamma)
        f_z = self.f.gradient(st.grad_f_z, st.z)
        tol = T(10) * self.eps * (T(1) + abs(f_z))
        while f_z > f_z_upp + tol and gamma >= self.minimum_gamma:
            gamma = gamma / T(2)
            st.n_gamma_halvings += 1
            st.y[...] = st.x - gamma * st.grad_f_x
            st.g_z = self.g.prox(st.z, st.y, gamma)
            st.res[...] = st.x - st.z
            f_z_upp = f_model(st.f_x, st.grad_f_x, st.res, self.alpha / gamma)
            f_z = self.f.gradient(st.grad_f_z, st.z)
            tol = T(10) * self.eps * (T(1) + abs(f_z))
        if gamma < self.minimum_gamma:
            warnings.warn(f"stepsize `gamma` became too small ({gamma})")
        st.gamma = gamma
        st.f_z = f_z
        return f_z, f_z_upp

    # -- Base.iterate(iter) ----------------------------------------------
    def init(self):
        T = self.T
        x = self.x0.copy()
        grad_f_x = np.empty_like(x)
        f_x = self.f.gradient(grad_f_x, x)
        if self.gamma is None:
            gamma = self.alpha / lower_bound_smoothness_constant(self.f, x, grad_f_x)
        else:
            gamma = self.gamma
        gamma = T(gamma)
        y = x - gamma * grad_f_x
        z = np.empty_like(x)
        g_z = self.g.prox(z, y, gamma)
        st = PANOCplusState(
            x=x, f_x=f_x, grad_f_x=grad_f_x, gamma=gamma, y=y, z=z, g_z=g_z, res=x - z,
            H=(_NoAccelOperator() if isinstance(self.directions, NoAcceleration)
               else _AndersonOperator(self.directions.memory, x) if isinstance(self.directions, AndersonAcceleration)
               else _BroydenOperator(self.directions.theta_bar, x) if isinstance(self.directions, Broyden)
               else (LBFGSCompactOperator(self.directions.memory, x) if getattr(self.directions, "compact", False)
                     else LBFGSOperator(self.directions.memory, x))),
            x_prev=np.empty_like(x), res_prev=np.empty_like(x), d=np.empty_like(x),
            x_d=np.empty_like(x), grad_f_x_d=np.empty_like(x), z_curr=np.empty_like(x),
            grad_f_z=np.empty_like(x),
        )
        if self.gamma is None or self.adaptive:
            self._backtrack_stepsize(st)
        else:
            st.f_z = self.f.gradient(st.grad_f_z, st.z)
        return st

    # -- Base.iterate(iter, state) ---------------------------------------
    def step(self, st: PANOCplusState):
        T = self.T
        # store iterate and residual for metric update later on
        st.x_prev[...] = st.x
        st.res_prev[...] = st.res

        # compute FBE
        FBE_x = self._f_model(st) + st.g_z

        # compute direction:  d = H * (-res)
        st.H.mul(st.d, -st.res)

        # backtrack tau 1 -> 0
        st.tau = T(1)
        np.add(st.x, st.d, out=st.x_d)
        st.f_x_d = self.f.gradient(st.grad_f_x_d, st.x_d)

        st.x[...] = st.x_d
        st.grad_f_x[...] = st.grad_f_x_d
        st.f_x = st.f_x_d

        st.z_curr[...] = st.z

        sigma = self.beta * (T(0.5) / st.gamma) * (T(1) - self.alpha)
        tol = T(10) * self.eps * (T(1) + abs(FBE_x))
        nr = _norm(st.res)
        threshold = FBE_x - sigma * (nr * nr) + tol

        st.n_backtracks = 0
        for k in range(1, self.max_backtracks + 1):
            st.y[...] = st.x - st.gamma * st.grad_f_x
            st.g_z = self.g.prox(st.z, st.y, st.gamma)
            st.res[...] = st.x - st.z

            f_z_upp = self._f_model(st)

            if self.gamma is None or self.adaptive:
                f_z = self.f.gradient(st.grad_f_z, st.z)
                st.f_z = f_z
                tol = T(10) * self.eps * (T(1) + abs(f_z))
                if f_z > f_z_upp + tol and st.gamma >= self.minimum_gamma:
                    st.gamma = st.gamma * T(0.5)
                    st.n_gamma_halvings += 1
                    if st.gamma < self.minimum_gamma:
                        warnings.warn(f"stepsize `gamma` became too small ({st.gamma})")
                    sigma = sigma * T(2)     # (upstream updates sigma only; threshold is kept)
                    st.H.reset()
                    continue
            else:
                st.f_z = self.f.gradient(st.grad_f_z, st.z)

            FBE_x_new = f_z_upp + st.g_z
            if FBE_x_new <= threshold or k >= self.max_backtracks:
                break
            st.tau = T(0) if k >= self.max_backtracks - 1 else st.tau / T(2)
            st.n_backtracks += 1
            st.x[...] = st.tau * st.x_d + (T(1) - st.tau) * st.z_curr
            st.f_x = self.f.gradient(st.grad_f_x, st.x)

        st.last_ys = st.H.update(st.x - st.x_prev, st.res - st.res_prev)
        return st

    def stop_norm(self, st):
        return _max(np.abs(st.res / st.gamma - st.grad_f_x + st.grad_f_z))


class PANOCplus:
    """``PANOCplus(; maxit=1000, tol=1e-8, verbose=false, freq=10, kwargs...)`` —
    IterativeAlgorithm wrapper.  Calling the object with f=, g=, x0= runs

        for (k, state) in enumerate(iter):
            if k >= maxit || stop(iter, state): return (solution, k)

    where the initial state counts as k = 1; the solution is ``state.z``."""

    def __init__(self, maxit=1000, tol=1e-8, verbose=False, freq=10, trace=None, **kwargs):
        self.maxit, self.tol, self.verbose, self.freq = maxit, tol, verbose, freq
        self.kwargs = kwargs
        self.trace = trace
        self.last_state = None

    def __call__(self, *, f, g, x0):
        it = PANOCplusIteration(f, g, x0, **self.kwargs)
        st = it.init()
        k = 1
        while True:
            sn = it.stop_norm(st)
            if self.trace is not None:
                self.trace(k, st, sn)
            if k >= self.maxit or sn <= self.tol:
                if self.verbose:
                    self._display(k, st)
                self.last_state = st
                return st.z, k
            if self.verbose and k % self.freq == 0:
                self._display(k, st)
            st = it.step(st)
            k += 1

    @staticmethod
    def _display(k, st):
        print("%5d | %.3e | %.3e | %.3e" % (k, st.gamma, np.max(np.abs(st.res)) / st.gamma, st.tau))


default_subsolver = PANOCplus


# ----------------------------- src/algorithms/alps.jl ----------------------
def alps(f, g, c, D, x0, y0, *, tol=None, tol_prim=None, tol_dual=None, inner_tol=None,
         maxit=100, theta_penalty=0.8, kappa_penalty=0.5, kappa_tol=0.1, verbose=False,
         dual_safeguard=default_dual_safeguard, subsolver=default_subsolver,
         subsolver_maxit=1_000_000_000, outer_trace=None, warm_start=False):
    """alps.jl:7-117.  Returns the 10-tuple
    (x, y, tot_it, tot_inner_it, elapsed_time, status, inner_tol, norm_res_prim, s, mu).

    warm_start (NOT in the reference, False = alps.jl:64 as written): from the second subproblem on
    ``subsolver(tol=…, verbose=…, gamma=γ_prev, adaptive=true)`` with the step size the previous subproblem ended
    with — upstream's own `gamma` / `adaptive` keywords of PANOCplus, no Lipschitz estimate."""
    start_time = time.time()
    T = x0.dtype.type
    if tol is None:
        tol = T(1e-6)
    if tol_prim is None:
        tol_prim = tol
    if tol_dual is None:
        tol_dual = tol
    if inner_tol is None:
        inner_tol = float(np.cbrt(tol_dual))

    x = np.empty_like(x0)
    y = np.empty_like(y0)
    cx = np.empty_like(y0)
    s = np.empty_like(y0)
    mu = np.empty_like(y0)
    gFun = NonsmoothCostFun(g)
    gFun.prox(x, x0, np.finfo(x0.dtype).eps)                  # :38
    objx = f(x) + gFun.gz                                     # :39
    c.eval(cx, x)                                             # :40
    D.proj(s, cx)                                             # :41
    default_penalty_parameter(mu, cx, s, objx)                # :42
    y[...] = y0                                               # :43
    norm_res_prim = None
    norm_res_prim_old = None
    alFun = AugLagFun(f, c, D, mu, y, x)                      # :46
    tot_it = 0
    tot_inner_it = 0
    solved = False
    tired = tot_it >= maxit
    broken = bool(np.isnan(objx))
    if verbose:
        print(f"[ Info: initial penalty parameters μ ∈ [{mu.min()}, {mu.max()}]")
        print(f"[ Info: initial inner tolerance {inner_tol}")

    can_stop = solved or tired or broken
    gamma_prev = None
    while not can_stop:
        tot_it += 1
        dual_safeguard(y, cx)                                 # :62
        if warm_start and gamma_prev is not None:
            sub_solver = subsolver(tol=inner_tol, verbose=verbose, gamma=gamma_prev, adaptive=True)
        else:
            sub_solver = subsolver(tol=inner_tol, verbose=verbose)  # :64
        AugLagUpdate(alFun, mu, y)                            # :65
        sub_sol, sub_it = sub_solver(f=alFun, g=gFun, x0=x)   # :66
        if getattr(sub_solver, "last_state", None) is not None:
            gamma_prev = sub_solver.last_state.gamma
        x[...] = sub_sol
        objx = alFun.fx + gFun.gz                             # :68
        tot_inner_it += sub_it
        sub_solved = sub_it < subsolver_maxit                 # :70
        c.eval(cx, x)                                         # :72
        np.add(cx, alFun.muy, out=y)                          # :74
        D.proj(s, y)                                          # :75
        y -= s                                                # :80
        y /= mu                                               # :81
        norm_res_prim_old = norm_res_prim
        norm_res_prim = _max(np.abs(cx - s))                  # :84

        solved = (inner_tol <= tol_dual and sub_solved) and (norm_res_prim <= tol_prim)
        tired = tot_it >= maxit
        broken = bool(np.isnan(objx))
        can_stop = solved or tired or broken
        if outer_trace is not None:
            outer_trace(tot_it, x, y, mu, sub_it, inner_tol, norm_res_prim, objx)

        if not can_stop:
            if norm_res_prim_old is None:
                pass
            elif norm_res_prim > max(theta_penalty * norm_res_prim_old, tol_prim):
                mu *= T(kappa_penalty)                        # :97
            inner_tol = max(kappa_tol * inner_tol, tol_dual)  # :100
    elapsed_time = time.time() - start_time

    if solved:
        status = "first_order"
    elif tired:
        status = "max_iter"
    elif broken:
        status = "exception"
    else:
        status = "unknown"
    return x, y, tot_it, tot_inner_it, elapsed_time, status, inner_tol, norm_res_prim, s, mu


# ---------------------------------------------------------------------------
#  ALS: the slack-variable sibling of ALPS
#    src/utilities/auglagfunslack.jl:15-154 ; src/algorithms/als.jl:7-120
# ---------------------------------------------------------------------------
class AugLagFunSlack:
    """auglagfunslack.jl:15-114.  al_f(x,s) = f(x) + 1/(2mu)||c(x) + mu y - s||^2 - mu/2 ||y||^2 on xs = [x; s]."""

    def __init__(self, f, c, mu, y, x):
        if REDUCER.any(mu <= 0):                               # :35-37
            raise ValueError("parameters `mu` must be positive")
        T = x.dtype.type
        self.f, self.c = f, c
        self.nx, self.ny = x.shape[0], y.shape[0]
        self.mu, self.y = mu, y
        self.muy = mu * y
        self.musqy = T(0.5) * _sum(self.muy * y)
        self.cx = np.empty_like(y)
        self.yupd = np.empty_like(y)
        self.dfx = np.empty_like(x)
        self.jtv = np.empty_like(x)
        self.n_grad = 0

    def __call__(self, xs):                                    # :59-73
        T = xs.dtype.type
        if xs.shape[0] != self.nx + self.ny:
            raise ValueError("wrong length of passed argument xs")
        x, s = xs[:self.nx].copy(), xs[self.nx:].copy()
        fx = self.f(x)
        self.c.eval(self.cx, x)
        self.yupd[...] = self.cx + self.muy - s
        Fxs = T(0.5) * _sum(self.yupd ** 2 / self.mu)
        Fxs += fx
        Fxs -= self.musqy
        self.yupd[...] = self.y + (self.cx - s) / self.mu
        return Fxs

    def gradient(self, dFxs, xs):                              # :78-97
        T = xs.dtype.type
        if xs.shape[0] != self.nx + self.ny or dFxs.shape[0] != self.nx + self.ny:
            raise ValueError("wrong length of passed argument")
        self.n_grad += 1
        x, s = xs[:self.nx].copy(), xs[self.nx:].copy()
        fx = self.f.gradient(self.dfx, x)
        self.c.eval(self.cx, x)
        Fxs = T(0.5) * _sum((self.cx + self.muy - s) ** 2 / self.mu)
        Fxs += fx
        Fxs -= self.musqy
        self.yupd[...] = self.y + (self.cx - s) / self.mu
        self.c.jtprod(self.jtv, x, self.yupd)
        dFxs[:self.nx] = self.dfx + self.jtv
        dFxs[self.nx:] = -self.yupd
        return Fxs


def AugLagUpdateSlack(F: AugLagFunSlack, mu, y):              # :102-114
    if REDUCER.any(mu <= 0):
        raise ValueError("parameters `mu` must be positive")
    if y.shape[0] != F.ny:
        raise ValueError("wrong length of passed argument y")
    T = y.dtype.type
    F.mu[...] = mu
    F.y[...] = y
    F.muy[...] = F.mu * F.y
    F.musqy = T(0.5) * _sum(F.muy * F.y)
    return None


class NonsmoothCostFunSlack:
    """auglagfunslack.jl:118-154: prox of [x; s] = [prox_g(x); proj_D(s)]."""

    def __init__(self, g, D, nx, ny):
        self.g, self.D, self.nx, self.ny = g, D, nx, ny
        self.gamma = 0.0
        self.gz = 0.0

    def prox(self, z, xs, gamma):
        if xs.shape[0] != self.nx + self.ny or z.shape[0] != self.nx + self.ny:
            raise ValueError("wrong length of passed argument")
        self.gamma = gamma
        x, s = xs[:self.nx].copy(), xs[self.nx:].copy()
        zx = np.empty_like(x)
        gz = self.g.prox(zx, x, gamma)
        z[:self.nx] = zx
        self.gz = gz
        zs = np.empty_like(s)
        self.D.proj(zs, s)
        z[self.nx:] = zs
        return gz


def als(f, g, c, D, x0, y0, *, tol=None, tol_prim=None, tol_dual=None, inner_tol=None, maxit=100,
        theta_penalty=0.8, kappa_penalty=0.5, kappa_tol=0.1, verbose=False,
        dual_safeguard=default_dual_safeguard, subsolver=default_subsolver,
        subsolver_maxit=1_000_000_000, warm_start=False):
    """als.jl:7-120.  Same 10-tuple as alps.  warm_start: as in `alps` (not in the reference; bit 0 only)."""
    start_time = time.time()
    T = x0.dtype.type
    nx, ny = x0.shape[0], y0.shape[0]
    if tol is None:
        tol = T(1e-6)
    tol_prim = tol if tol_prim is None else tol_prim
    tol_dual = tol if tol_dual is None else tol_dual
    inner_tol = float(np.cbrt(tol_dual)) if inner_tol is None else inner_tol
    x = np.empty_like(x0)
    s = np.empty_like(y0)
    xSlack = np.zeros(nx + ny, dtype=x0.dtype)
    y = np.empty_like(y0)
    cx = np.empty_like(y0)
    mu = np.empty_like(y0)
    gFun = NonsmoothCostFun(g)
    gFun.prox(x, x0, np.finfo(x0.dtype).eps)                   # :41
    objx = f(x) + gFun.gz
    c.eval(cx, x)
    D.proj(s, cx)
    default_penalty_parameter(mu, cx, s, objx)
    y[...] = y0
    norm_res_prim = None
    norm_res_prim_old = None
    fSlack = AugLagFunSlack(f, c, mu, y, x)                    # :49
    gSlack = NonsmoothCostFunSlack(g, D, nx, ny)               # :50
    tot_it = 0
    tot_inner_it = 0
    solved = False
    tired = tot_it >= maxit
    broken = bool(np.isnan(objx))
    can_stop = solved or tired or broken
    while not can_stop:
        tot_it += 1
        dual_safeguard(y, cx)                                  # :66
        if warm_start and tot_it > 1 and getattr(sub_solver, "last_state", None) is not None:
            sub_solver = subsolver(tol=inner_tol, verbose=verbose, gamma=sub_solver.last_state.gamma, adaptive=True)
        else:
            sub_solver = subsolver(tol=inner_tol, verbose=verbose)
        AugLagUpdateSlack(fSlack, mu, y)                       # :69
        xSlack[:nx] = x
        xSlack[nx:] = s
        sub_sol, sub_it = sub_solver(f=fSlack, g=gSlack, x0=xSlack)   # :72
        if sub_sol.shape[0] != nx + ny:
            raise ValueError("wrong dimension of sub_sol")
        xSlack[...] = sub_sol
        x[...] = xSlack[:nx]
        s[...] = xSlack[nx:]
        objx = f(x) + gSlack.gz                                # :79
        tot_inner_it += sub_it
        sub_solved = sub_it < subsolver_maxit
        c.eval(cx, x)
        y[...] = y + (cx - s) / mu                             # :84
        norm_res_prim_old = norm_res_prim
        norm_res_prim = _max(np.abs(cx - s))                   # :87
        solved = (inner_tol <= tol_dual and sub_solved) and norm_res_prim <= tol_prim
        tired = tot_it >= maxit
        broken = bool(np.isnan(objx))
        can_stop = solved or tired or broken
        if not can_stop:
            if norm_res_prim_old is None:
                pass
            elif norm_res_prim > max(theta_penalty * norm_res_prim_old, tol_prim):
                mu *= T(kappa_penalty)
            inner_tol = max(kappa_tol * inner_tol, tol_dual)
    elapsed_time = time.time() - start_time
    status = "first_order" if solved else ("max_iter" if tired else ("exception" if broken else "unknown"))
    return x, y, tot_it, tot_inner_it, elapsed_time, status, inner_tol, norm_res_prim, s, mu
