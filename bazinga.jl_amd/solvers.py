"""Host-side mirror of the reference's solver interface for the hot path.

    alps(f, g, c, D, x0, y0; kw...)          src/algorithms/alps.jl:7-117
    PANOCplus(; kw...)(f=alFun, g=gFun, x0)  the `subsolver` seam, alps.jl:24,64-66
    AugLagFun / AugLagUpdate                 src/utilities/auglagfun.jl
    NonsmoothCostFun                         src/utilities/nonsmoothcostfun.jl
    default_dual_safeguard / default_penalty_parameter   src/utilities/safeguards.jl

Same names, argument meaning, defaults and error behaviour; the arithmetic on
n-vectors runs in libbazinga_hip.so (HIP kernels, gfx950).  Nothing here falls
back to a CPU solver: without the library or a GPU these calls raise.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib as L
from .device import Problem, default_context
from .oracles import UnsupportedOracle


class LBFGS:
    """ProximalAlgorithms.LBFGS(memory) — the only `directions` the shipped scripts select
    (demo/rosenbrock.jl:103,275)."""

    def __init__(self, memory=5, compact=None):
        self.memory = int(memory)
        # how the operator is evaluated (bz_panoc_opts.lbfgs_compact): False = two-loop recursion in the
        # reference's operation order; True = compact representation (same operator, one pass per iteration
        # on the separable path); None = compact where that one-pass kernel applies, two-loop elsewhere
        self.compact = None if compact is None else bool(compact)


class NoAcceleration:
    """ProximalAlgorithms.NoAcceleration() (demo/rosenbrock.jl:96-97): plain forward-backward direction
    d = -res; lowered as L-BFGS with an empty memory."""
    memory = 0


class AndersonAcceleration:
    """ProximalAlgorithms.AndersonAcceleration(n) (demo/rosenbrock.jl:100-101): type-II Anderson acceleration over the
    last n pairs, d = v + (S - Y) (Y \\ v); every pair is kept (no curvature test).  On the device it shares the
    compact-form kernels: d = v + sum a_i s_i - sum a_i y_i with a = (Y'Y)^-1 Y'v from the Gram products the passes
    return anyway."""

    def __init__(self, memory=5):
        if not 1 <= int(memory) <= 5:
            raise UnsupportedOracle("AndersonAcceleration(n) is lowered for 1 <= n <= 5")
        self.memory = int(memory)


class Broyden:
    """ProximalAlgorithms.Broyden(; theta_bar = 0.2) (demo/rosenbrock.jl:98-99): the modified Broyden update of the PANOC
    papers on a dense n-by-n operator — the tiny-n demos only (n <= 4096 here)."""
    memory = 0

    def __init__(self, theta_bar=0.2):
        self.theta_bar = float(theta_bar)


# ------------------------------------------------------------------ safeguards
def default_dual_safeguard(y, cx=None):
    """src/utilities/safeguards.jl:2-10"""
    np.clip(y, -1e20, 1e20, out=y)
    return None


def default_penalty_parameter(mu, cx, proj_cx, objx):
    """src/utilities/safeguards.jl:13-18"""
    d2 = ((cx - proj_cx) ** 2).astype(np.float64, copy=False)
    mu[...] = (np.maximum(1.0, 0.5 * d2) / max(1.0, float(objx))).astype(mu.dtype, copy=False)
    mu[...] = (mu.astype(np.float64, copy=False) * 0.1).astype(mu.dtype, copy=False)
    mu[...] = np.maximum(1e-8, np.minimum(mu.astype(np.float64, copy=False), 1e8)).astype(mu.dtype, copy=False)
    return None


# --------------------------------------------------------- AL functor wrappers
class NonsmoothCostFun:
    """src/utilities/nonsmoothcostfun.jl:1-22 — records gamma and g(z) of the last prox."""

    def __init__(self, g):
        self.g = g
        self.gamma = 0.0
        self.gz = 0.0


class AugLagFun:
    """src/utilities/auglagfun.jl:11-54.  Holds (f, c, D, mu, y) and the scalar caches the
    outer loop reads back (`fx`, `muy`, `musqy`).  The value/gradient evaluation itself
    (auglagfun.jl:58-86) happens on the device inside the subsolver."""

    def __init__(self, f, c, D, mu, y, x):
        if np.any(mu <= 0):
            raise ValueError("parameters `mu` must be positive")     # auglagfun.jl:33-34
        self.f, self.c, self.D = f, c, D
        self.mu, self.y = mu, y
        self.muy = mu * y
        self.musqy = x.dtype.type(0.5) * np.sum(self.muy * y)
        self.fx = x.dtype.type(0)
        self.n, self.ny, self.dtype = x.shape[0], y.shape[0], x.dtype
        self._problem = None
        self._problem_key = None

    def problem(self, g, ctx=None) -> Problem:
        key = (id(g), id(ctx))
        if self._problem is None or self._problem_key != key:
            self._problem = Problem(self.f, g, self.c, self.D, self.n, self.ny, self.dtype, ctx)
            self._problem_key = key
        return self._problem


class AugLagFunSlack:
    """src/utilities/auglagfunslack.jl:15-54: smooth part of the slack-form AL on xs = [x; s]."""

    def __init__(self, f, c, mu, y, x):
        if np.any(mu <= 0):
            raise ValueError("parameters `mu` must be positive")
        self.f, self.c = f, c
        self.nx, self.ny = x.shape[0], y.shape[0]
        self.mu, self.y = mu, y
        self.muy = mu * y
        self.musqy = x.dtype.type(0.5) * np.sum(self.muy * y)
        self.fx = x.dtype.type(0)
        self.dtype = x.dtype
        self._problem = None
        self._problem_key = None

    def problem(self, g, D, ctx=None) -> Problem:
        key = (id(g), id(D), id(ctx))
        if self._problem is None or self._problem_key != key:
            self._problem = Problem(self.f, g, self.c, D, self.nx, self.ny, self.dtype, ctx, slack=True)
            self._problem_key = key
        return self._problem


class NonsmoothCostFunSlack:
    """src/utilities/auglagfunslack.jl:118-154: prox of [x; s] = [prox_g(x); proj_D(s)]."""

    def __init__(self, g, D, nx, ny):
        self.g, self.D, self.nx, self.ny = g, D, nx, ny
        self.gamma = 0.0
        self.gz = 0.0


def AugLagUpdate(al, mu, y):
    """src/utilities/auglagfun.jl:91-101"""
    if np.any(mu <= 0):
        raise ValueError("parameters `mu` must be positive")
    al.mu[...] = mu
    al.y[...] = y
    al.muy[...] = al.mu * al.y
    al.musqy = y.dtype.type(0.5) * np.sum(al.muy * al.y)
    return None


# ----------------------------------------------------------------- inner solver
class PANOCplus:
    """Drop-in for ``ProximalAlgorithms.PANOCplus(; kwargs...)`` at the `subsolver` seam.

    ``solver = PANOCplus(tol=..., verbose=...)`` then
    ``sol, it = solver(f=alFun, g=gFun, x0=x)`` (alps.jl:64-66).  `f` must be an
    AugLagFun and `g` a NonsmoothCostFun over lowered oracle types."""

    def __init__(self, *, directions=None, maxit=1000, tol=1e-8, verbose=False, freq=10,
                 minimum_gamma=1e-7, alpha=0.95, beta=0.5, max_backtracks=20, fuse=True, persist=True,
                 affine_refresh=16, ctx=None, Lf=None, gamma=None, adaptive=None):
        self.directions = directions if directions is not None else LBFGS(5)
        if not isinstance(self.directions, (LBFGS, NoAcceleration, AndersonAcceleration, Broyden)):
            raise UnsupportedOracle("directions must be LBFGS(M), NoAcceleration(), AndersonAcceleration(n) or Broyden()")
        self.maxit, self.tol, self.verbose, self.freq = maxit, tol, verbose, freq
        self.minimum_gamma, self.alpha, self.beta = minimum_gamma, alpha, beta
        self.max_backtracks, self.fuse, self.persist, self.ctx = max_backtracks, fuse, persist, ctx
        # affine images (bz_panoc_opts.affine_refresh): dense affine c with D = ZeroSet / FreeSet — 0 off, k >= 1: a
        # pass-over-A evaluation of the trial point's gradient every k-th iteration, images in between
        self.affine_refresh = int(affine_refresh)
        # upstream's step-size keywords: Lf = nothing, gamma = Lf === nothing ? nothing : alpha / Lf,
        # adaptive = gamma === nothing
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else alpha / Lf)
        self.adaptive = (self.gamma is None) if adaptive is None else bool(adaptive)
        if self.gamma is not None and not self.gamma > 0:
            raise ValueError("gamma must be positive")
        self.stats = None

    def c_opts(self) -> L.PanocOpts:
        o = L.PanocOpts()
        L.load().bz_panoc_default_opts(C.byref(o))
        o.tol, o.maxit = float(self.tol), int(min(self.maxit, 2 ** 62))
        o.freq, o.verbose = int(min(self.freq, 2 ** 31 - 1)), int(bool(self.verbose))
        o.minimum_gamma, o.alpha, o.beta = float(self.minimum_gamma), float(self.alpha), float(self.beta)
        o.max_backtracks, o.lbfgs_memory, o.fuse = int(self.max_backtracks), self.directions.memory, int(bool(self.fuse))
        o.persist = int(bool(self.persist))
        o.affine_refresh = self.affine_refresh
        o.gamma = 0.0 if self.gamma is None else float(self.gamma)
        o.Lf = 0.0 if self.Lf is None else float(self.Lf)
        o.adaptive = int(self.adaptive)
        if isinstance(self.directions, AndersonAcceleration):
            o.directions = L.BZ_DIR_ANDERSON
        elif isinstance(self.directions, Broyden):
            o.directions, o.broyden_theta_bar = L.BZ_DIR_BROYDEN, self.directions.theta_bar
        cm = getattr(self.directions, "compact", None)
        o.lbfgs_compact = 2 if cm is None else int(bool(cm))
        return o

    def __call__(self, *, f, g, x0):
        if isinstance(f, AugLagFunSlack) and isinstance(g, NonsmoothCostFunSlack):
            prob = f.problem(g.g, g.D, self.ctx)            # ALS seam, als.jl:68-72
        elif isinstance(f, AugLagFun) and isinstance(g, NonsmoothCostFun):
            prob = f.problem(g.g, self.ctx)
        else:
            raise UnsupportedOracle("the device subsolver takes f=AugLagFun(...), g=NonsmoothCostFun(...) "
                                    "or their Slack forms")
        prob.set_multipliers(f.mu, f.y)
        z, st = prob.panoc_solve(self.c_opts(), x0)
        f.fx = x0.dtype.type(st.f_z)            # side channels read by alps.jl:68
        g.gz = x0.dtype.type(st.g_z)
        g.gamma = st.gamma
        self.stats = st
        return z, int(st.iters)


default_subsolver = PANOCplus

_STATUS = ("first_order", "max_iter", "exception", "unknown")


def als(f, g, c, D, x0, y0, **kw):
    """Bazinga.als (src/algorithms/als.jl:7-120): the slack-variable sibling of `alps` — same keywords,
    same 10-tuple; the inner solver works on xs = [x; s] with the block prox [prox_g; proj_D]
    (src/utilities/auglagfunslack.jl)."""
    return alps(f, g, c, D, x0, y0, _slack=True, **kw)


def alps(f, g, c, D, x0, y0, *, tol=None, tol_prim=None, tol_dual=None, inner_tol=None, maxit=100,
         theta_penalty=0.8, kappa_penalty=0.5, kappa_tol=0.1, verbose=False,
         dual_safeguard=default_dual_safeguard, subsolver=default_subsolver,
         subsolver_maxit=1_000_000_000, resident=None, ctx=None, problem=None, _slack=False, warm_start=False):
    """Bazinga.alps (src/algorithms/alps.jl:7-117): same keywords and defaults, same 10-tuple
    ``(x, y, tot_it, tot_inner_it, elapsed_time, status, inner_tol, norm_res_prim, s, mu)``
    (status is the Symbol's name as a string).  x0 / y0 are never mutated.

    resident=True (default when `subsolver` and `dual_safeguard` are the defaults) runs the
    whole outer loop with device-resident vectors (bz_alps_solve): only scalars cross PCIe
    between subproblems.  resident=False runs the outer loop below on the host exactly as
    alps.jl does and enters the device at the `subsolver` seam.

    problem: an already created device Problem for (f, g, c, D) to run the resident loop on — what a sharded
    Stencil5ptQuadratic needs, whose halo regions must be connected between the ranks first.

    warm_start (NOT a keyword of the reference; False = alps.jl:64 as written): from the second subproblem on the
    subsolver is built as ``subsolver(tol=…, verbose=…, gamma=γ_prev, adaptive=True)`` with the step size the previous
    subproblem ended with — no Lipschitz estimate (one AL gradient less per subproblem: two passes over a dense A)."""
    x0 = np.asarray(x0)
    y0 = np.asarray(y0)
    T = x0.dtype.type
    if tol is None:
        tol = T(1e-6)
    if problem is not None:
        resident = True
    tol_prim = tol if tol_prim is None else tol_prim
    tol_dual = tol if tol_dual is None else tol_dual
    inner_tol = float(np.cbrt(tol_dual)) if inner_tol is None else inner_tol
    if resident is None:
        resident = subsolver is default_subsolver and dual_safeguard is default_dual_safeguard

    if resident:
        sub = subsolver(tol=inner_tol, verbose=verbose)
        if not isinstance(sub, PANOCplus):
            raise UnsupportedOracle("resident=True needs a PANOCplus subsolver factory")
        prob = problem or Problem(f, g, c, D, x0.shape[0], y0.shape[0], x0.dtype, ctx or sub.ctx, slack=_slack)
        ao = L.AlpsOpts()
        L.load().bz_alps_default_opts(C.byref(ao), L.BZ_F64 if x0.dtype == np.float64 else L.BZ_F32)
        ao.tol_prim, ao.tol_dual, ao.inner_tol = float(tol_prim), float(tol_dual), float(inner_tol)
        ao.maxit, ao.theta_penalty, ao.kappa_penalty = int(maxit), float(theta_penalty), float(kappa_penalty)
        ao.kappa_tol, ao.subsolver_maxit, ao.verbose = float(kappa_tol), int(subsolver_maxit), int(bool(verbose))
        ao.warm_start = int(warm_start)
        x, y, s, mu, st = prob.alps_solve(ao, sub.c_opts(), x0, y0)
        if problem is None:
            prob.close()
        return (x, y, int(st.tot_it), int(st.tot_inner_it), st.elapsed_s, _STATUS[st.status],
                st.inner_tol, st.norm_res_prim if st.tot_it else None, s, mu)

    # ---- host outer loop, device subsolver (line numbers: src/algorithms/alps.jl)
    start_time = time.time()
    x = np.empty_like(x0)                                       # :31-35
    y = np.empty_like(y0)
    cx = np.empty_like(y0)
    s = np.empty_like(y0)
    mu = np.empty_like(y0)
    gFun = NonsmoothCostFun(g)                                  # :37
    probe = Problem(f, g, c, D, x0.shape[0], y0.shape[0], x0.dtype, ctx)
    xz, gz0 = probe.eval_prox(x0, np.finfo(x0.dtype).eps)       # :38  prox!(x, gFun, x0, eps(T))
    x[...] = xz
    gFun.gz = T(gz0)
    # f(x), c(x), proj_D(c(x)): evaluated through the device AL functor with mu = 1, y = 0
    probe.set_multipliers(np.ones_like(y0), np.zeros_like(y0))
    _, vals = probe.eval_al_gradient(x)
    objx = T(vals[1]) + gFun.gz                                 # :39
    cx[...] = _eval_c_host(c, x, cx.shape[0])                                # :40
    s[...] = _proj_host(D, cx)                                  # :41
    default_penalty_parameter(mu, cx, s, objx)                  # :42
    y[...] = y0                                                 # :43
    probe.close()
    norm_res_prim = None
    norm_res_prim_old = None
    if _slack:
        if warm_start:
            raise UnsupportedOracle("warm_start with the host outer loop of als: use resident=True")
        return _als_host_loop(f, g, c, D, x, y, cx, s, mu, gFun, objx, tol_prim, tol_dual, inner_tol, maxit,
                              theta_penalty, kappa_penalty, kappa_tol, verbose, dual_safeguard, subsolver,
                              subsolver_maxit, start_time)
    alFun = AugLagFun(f, c, D, mu, y, x)                        # :46
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
        dual_safeguard(y, cx)                                   # :62
        if warm_start and gamma_prev is not None:               # (opt-in deviation from :64)
            sub_solver = subsolver(tol=inner_tol, verbose=verbose, gamma=gamma_prev, adaptive=True)
        else:
            sub_solver = subsolver(tol=inner_tol, verbose=verbose)  # :64
        AugLagUpdate(alFun, mu, y)                              # :65
        sub_sol, sub_it = sub_solver(f=alFun, g=gFun, x0=x)     # :66
        gamma_prev = getattr(gFun, "gamma", None)
        x[...] = sub_sol
        objx = alFun.fx + gFun.gz                               # :68
        tot_inner_it += sub_it
        sub_solved = sub_it < subsolver_maxit                   # :70
        cx[...] = _eval_c_host(c, x, cx.shape[0])                            # :72
        np.add(cx, alFun.muy, out=y)                            # :74
        s[...] = _proj_host(D, y)                               # :75
        y -= s                                                  # :80
        y /= mu                                                 # :81
        norm_res_prim_old = norm_res_prim
        norm_res_prim = np.max(np.abs(cx - s))                  # :84
        solved = (inner_tol <= tol_dual and sub_solved) and (norm_res_prim <= tol_prim)
        tired = tot_it >= maxit
        broken = bool(np.isnan(objx))
        can_stop = solved or tired or broken
        if not can_stop:
            if norm_res_prim_old is None:
                pass
            elif norm_res_prim > max(theta_penalty * norm_res_prim_old, tol_prim):
                mu *= T(kappa_penalty)                          # :97
            inner_tol = max(kappa_tol * inner_tol, tol_dual)    # :100
    elapsed_time = time.time() - start_time
    status = "first_order" if solved else ("max_iter" if tired else ("exception" if broken else "unknown"))
    return x, y, tot_it, tot_inner_it, elapsed_time, status, inner_tol, norm_res_prim, s, mu


def _als_host_loop(f, g, c, D, x, y, cx, s, mu, gFun, objx, tol_prim, tol_dual, inner_tol, maxit, theta_penalty,
                   kappa_penalty, kappa_tol, verbose, dual_safeguard, subsolver, subsolver_maxit, start_time):
    """als.jl:46-118 on the host, device at the subsolver seam (line numbers: src/algorithms/als.jl)."""
    T = x.dtype.type
    nx, ny = x.shape[0], y.shape[0]
    xSlack = np.zeros(nx + ny, dtype=x.dtype)                   # :35
    norm_res_prim = None
    norm_res_prim_old = None
    fSlack = AugLagFunSlack(f, c, mu, y, x)                     # :49
    gSlack = NonsmoothCostFunSlack(g, D, nx, ny)                # :50
    tot_it = 0
    tot_inner_it = 0
    solved = False
    tired = tot_it >= maxit
    broken = bool(np.isnan(objx))
    can_stop = solved or tired or broken
    while not can_stop:
        tot_it += 1
        dual_safeguard(y, cx)                                   # :66
        sub_solver = subsolver(tol=inner_tol, verbose=verbose)  # :68
        AugLagUpdate(fSlack, mu, y)                             # :69
        xSlack[:nx] = x                                         # :70-71
        xSlack[nx:] = s
        sub_sol, sub_it = sub_solver(f=fSlack, g=gSlack, x0=xSlack)   # :72
        if sub_sol.shape[0] != nx + ny:
            raise ValueError("wrong dimension of sub_sol")
        xSlack[...] = sub_sol
        x[...] = xSlack[:nx]
        s[...] = xSlack[nx:]
        objx = fSlack.fx + gSlack.gz                            # :79  f(x) at the returned point
        tot_inner_it += sub_it
        sub_solved = sub_it < subsolver_maxit
        cx[...] = _eval_c_host(c, x, cx.shape[0])                            # :82
        y[...] = y + (cx - s) / mu                              # :84
        norm_res_prim_old = norm_res_prim
        norm_res_prim = np.max(np.abs(cx - s))                  # :87
        solved = (inner_tol <= tol_dual and sub_solved) and norm_res_prim <= tol_prim
        tired = tot_it >= maxit
        broken = bool(np.isnan(objx))
        can_stop = solved or tired or broken
        if not can_stop:
            if norm_res_prim_old is None:
                pass
            elif norm_res_prim > max(theta_penalty * norm_res_prim_old, tol_prim):
                mu *= T(kappa_penalty)                          # :100
            inner_tol = max(kappa_tol * inner_tol, tol_dual)    # :103
    elapsed_time = time.time() - start_time
    status = "first_order" if solved else ("max_iter" if tired else ("exception" if broken else "unknown"))
    return x, y, tot_it, tot_inner_it, elapsed_time, status, inner_tol, norm_res_prim, s, mu


def _eval_c_host(c, x, ny=None):
    """eval!(cx, c, x) for the host outer loop's O(ny) bookkeeping (outside the hot path)."""
    from .oracles import DenseAffine, IdentityFunction
    if isinstance(c, IdentityFunction):
        return x
    if isinstance(c, DenseAffine):
        return (c.A @ x - c.b).astype(x.dtype, copy=False)
    if callable(getattr(c, "eval", None)) and ny is not None:      # generic oracle: the reference's protocol
        cx = np.empty(ny, x.dtype)
        c.eval(cx, x)
        return cx
    raise UnsupportedOracle(f"c of type {type(c).__name__} is not lowered")


def _proj_host(D, v):
    """proj!(s, D, v) for the O(ny) outer-loop bookkeeping of the host path (the reference
    keeps this on the CPU too; the hot path never calls it)."""
    from .oracles import FreeSet, IndBox, IndFree, IndicatorSet, ZeroSet
    if isinstance(D, ZeroSet):
        return np.zeros_like(v)
    if isinstance(D, FreeSet) or (isinstance(D, IndicatorSet) and isinstance(D.f, IndFree)):
        return v.copy()
    if isinstance(D, IndicatorSet) and isinstance(D.f, IndBox):
        return np.where(v < D.f.lb, D.f.lb, np.where(v > D.f.ub, D.f.ub, v)).astype(v.dtype, copy=False)
    if callable(getattr(D, "proj", None)):                          # generic oracle / pairwise sets
        s = np.empty_like(v)
        D.proj(s, v)
        return s
    raise UnsupportedOracle(f"D of type {type(D).__name__} is not lowered")
