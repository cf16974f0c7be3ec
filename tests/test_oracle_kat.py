"""The CPU oracle against every fixed answer the reference's own tests and demos hold for the
hot path.  These pin the oracle; the GPU tests then pin the HIP path to it."""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import bazinga_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_lasso_kat_test_verbose():
    """test/problems/test_verbose.jl:7-13,19,22-25,29,37,42-44"""
    A = np.array([[1, -2, 3, -4, 5], [2, -1, 0, -1, 3], [-1, 0, 4, -3, 2], [-1, -1, -1, 1, 3]], float)
    b = np.array([1, 2, 3, 4.0])
    lam = 0.1 * np.max(np.abs(A.T @ b))
    x_star = np.array([-3.877278911564627e-01, 0, 0, 2.174149659863943e-02, 6.168435374149660e-01])
    x0, y0 = np.zeros(5), np.zeros(5)
    out = R.alps(R.LeastSquares(A, b), R.NormL1(lam), R.IdentityFunction(), R.FreeSet(), x0, y0, verbose=True)
    x, it, subit = out[0], out[2], out[3]
    assert x.dtype == np.float64
    assert np.max(np.abs(x - x_star)) <= 1e-4
    assert it < 10
    assert subit < 50
    assert out[5] == "first_order"


def test_lasso_kat_with_the_warm_started_outer_loop():
    """warm-starting gamma across subproblems is an opt-in deviation from alps.jl:64 — the reference's own
    KAT (test_verbose.jl:29,42-44) still holds with it, and from the second subproblem on no Lipschitz estimate is made."""
    A = np.array([[1, -2, 3, -4, 5], [2, -1, 0, -1, 3], [-1, 0, 4, -3, 2], [-1, -1, -1, 1, 3]], float)
    b = np.array([1, 2, 3, 4.0])
    lam = 0.1 * np.max(np.abs(A.T @ b))
    x_star = np.array([-3.877278911564627e-01, 0, 0, 2.174149659863943e-02, 6.168435374149660e-01])
    made = []

    class Sub(R.PANOCplus):
        def __init__(self, **kw):
            made.append(kw.get("gamma"))
            super().__init__(**kw)
    out = R.alps(R.LeastSquares(A, b), R.NormL1(lam), R.IdentityFunction(), R.FreeSet(), np.zeros(5), np.zeros(5),
                 warm_start=True, subsolver=Sub)
    assert np.max(np.abs(out[0] - x_star)) <= 1e-4 and out[2] < 10 and out[3] < 50 and out[5] == "first_order"
    assert made[0] is None and all(g is not None and g > 0 for g in made[1:]) and len(made) == out[2] >= 2


def test_given_step_size_keywords():
    """upstream's `gamma` / `Lf` / `adaptive`: a given step size is kept unless `adaptive` is set"""
    n = 50
    rng = np.random.default_rng(0)
    q, bb = rng.uniform(0.5, 4.0, n), rng.standard_normal(n)
    f, g = R.DiagQuadratic(q, bb), R.NormL1(0.1)
    for kw, g0, adaptive in ((dict(gamma=0.1), 0.1, False), (dict(Lf=4.0), 0.95 / 4.0, False),
                             (dict(gamma=10.0, adaptive=True), 10.0, True)):
        it = R.PANOCplusIteration(f, g, np.zeros(n), **kw)
        st = it.init()
        for _ in range(10):
            st = it.step(st)
        if adaptive:
            assert st.gamma < g0 and st.n_gamma_halvings >= 1 and 0.95 / st.gamma >= 4.0 * 0.9
        else:
            assert st.gamma == g0 and st.n_gamma_halvings == 0


@pytest.mark.parametrize("gkind", ["box", "free"])
def test_nonconvex_qp_tiny(gkind):
    """test/problems/test_nonconvex_qp.jl:8-52"""
    Q = np.diag([-0.5, 1.0])
    q = np.array([0.3, 0.5])
    g = R.IndBox(-1.0, 1.0) if gkind == "box" else R.IndFree()
    x0, y0 = np.zeros(2), np.zeros(2)
    x0b = x0.copy()
    out = R.alps(R.Quadratic(Q, q), g, R.IdentityFunction(), R.ClosedSet(R.IndBox(-1.0, 1.0)), x0, y0)
    gamma = 0.95 / 1.0
    x = out[0]
    z = np.minimum(1.0, np.maximum(-1.0, x - gamma * (Q @ x + q)))
    assert np.max(np.abs(x - z)) / gamma <= 1e-4
    assert np.array_equal(x0, x0b)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("gkind", ["box", "free"])
def test_nonconvex_qp_random(k, gkind):
    """test/problems/test_nonconvex_qp.jl:55-108 (property, re-seeded with numpy's RNG: Julia's
    MersenneTwister stream is not reproducible outside Julia)."""
    rng = np.random.default_rng(k)
    n = 100
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = 2 * rng.random(n) - 1
    Q = U @ np.diag(ev) @ U.T
    Q = 0.5 * (Q + Q.T)
    q = rng.standard_normal(n)
    gamma = 0.95 / np.max(np.abs(ev))
    g = R.IndBox(-1.0, 1.0) if gkind == "box" else R.IndFree()
    x0, y0 = np.zeros(n), np.zeros(n)
    x0b = x0.copy()
    out = R.alps(R.Quadratic(Q, q), g, R.IdentityFunction(), R.ClosedSet(R.IndBox(-1.0, 1.0)), x0, y0)
    x = out[0]
    z = np.minimum(1.0, np.maximum(-1.0, x - gamma * (Q @ x + q)))
    assert np.max(np.abs(x - z)) / gamma <= 1e-4
    assert np.array_equal(x0, x0b)


def test_rosenbrock_config1():
    """BASELINE config 1 (demo/rosenbrock.jl:85-136,186): from a grid of starts every ALPS +
    PANOCplus(LBFGS(5), minimum_gamma=1e-32) run ends at the expected minimiser (0,0)."""
    warnings.simplefilter("ignore")
    sub = lambda **kw: R.PANOCplus(directions=R.LBFGS(5), maxit=10 ** 9, freq=10 ** 9, minimum_gamma=1e-32, **kw)
    for x1 in np.arange(-5, 5.01, 1.25):
        for x2 in np.arange(-5, 5.01, 1.25):
            out = R.alps(R.SmoothCostRosenbrock(10.0), R.NonsmoothCostRosenbrock(1.0), R.ConstraintRosenbrock(),
                         R.SetRosenbrock(), np.array([x1, x2]), np.zeros(2), tol=1e-8, inner_tol=1.0,
                         subsolver=sub, subsolver_maxit=10 ** 9)
            assert out[5] == "first_order"
            assert np.max(np.abs(out[0])) <= 1e-4, (x1, x2, out[0])


@pytest.mark.parametrize("D", ["box", "free", "zero"])
def test_auglag_gradient_finite_difference(D):
    """AugLagFun.gradient! (auglagfun.jl:73-86) is the gradient of AugLagFun value (auglagfun.jl:58-69)."""
    rng = np.random.default_rng(0)
    n = 40
    q, b = rng.uniform(0.1, 10, n), rng.standard_normal(n) * 10
    Ds = {"box": R.ClosedSet(R.IndBox(-1.0, 1.0)), "free": R.FreeSet(), "zero": R.ZeroSet()}[D]
    mu, y, x = rng.uniform(0.01, 1, n), rng.standard_normal(n), rng.standard_normal(n) * 2
    al = R.AugLagFun(R.DiagQuadratic(q, b), R.IdentityFunction(), Ds, mu, y, x)
    g = np.empty(n)
    lx = al.gradient(g, x)
    assert abs(lx - al(x)) <= 1e-12 * max(1, abs(lx))
    h = 1e-6
    for i in range(0, n, 7):
        e = np.zeros(n)
        e[i] = h
        fd = (al(x + e) - al(x - e)) / (2 * h)
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(g[i]))


def test_free_set_closed_form():
    """With D = FreeSet the AL term is identically zero and x_i = soft(b_i, lambda)/q_i."""
    import bazinga_jl_amd as bz
    n = 5000
    d = bz.synth.l1_quadratic(n)
    out = R.alps(R.DiagQuadratic(d["q"], d["b"]), R.NormL1(d["lam"]), R.IdentityFunction(), R.FreeSet(),
                 np.zeros(n), np.zeros(n), tol=1e-9)
    xs = np.sign(d["b"]) * np.maximum(np.abs(d["b"]) - d["lam"], 0) / d["q"]
    assert out[5] == "first_order"
    assert np.max(np.abs(out[0] - xs)) <= 1e-7


def test_stencil_matches_dense_laplacian():
    nx, ny = 7, 5
    rng = np.random.default_rng(1)
    b = rng.standard_normal(nx * ny)
    f = R.Stencil5ptQuadratic(nx, ny, b)
    A = np.zeros((nx * ny, nx * ny))
    for i in range(nx):
        for j in range(ny):
            k = i * ny + j
            A[k, k] = 4
            for di, dj in ((0, -1), (0, 1), (-1, 0), (1, 0)):
                ii, jj = i + di, j + dj
                if 0 <= ii < nx and 0 <= jj < ny:
                    A[k, ii * ny + jj] = -1
    x = rng.standard_normal(nx * ny)
    g = np.empty_like(x)
    fx = f.gradient(g, x)
    assert np.allclose(g, A @ x - b, atol=1e-13)
    assert abs(fx - (0.5 * x @ A @ x - b @ x)) <= 1e-12
    assert abs(fx - f(x)) <= 1e-13


def test_golden_traces():
    """Committed restatement-generated traces (tests/golden/make_golden.py): guards the oracle
    itself against accidental edits."""
    import bazinga_jl_amd as bz
    with open(os.path.join(GOLD, "panoc_trace_cfg2_n64.json")) as fh:
        gold = json.load(fh)
    from tests.golden.make_golden import cfg2_trace
    now = cfg2_trace(gold["n"], gold["iters"])
    for a, b in zip(now["rows"], gold["rows"]):
        assert a["k"] == b["k"]
        assert a["gamma"] == b["gamma"]
        assert np.allclose(a["x"], b["x"], rtol=1e-12, atol=1e-14)
        assert np.allclose(a["z"], b["z"], rtol=1e-12, atol=1e-14)
    with open(os.path.join(GOLD, "panoc_trace_cfg2_n64_compact.json")) as fh:
        goldc = json.load(fh)
    nowc = cfg2_trace(goldc["n"], goldc["iters"], compact=True)
    for a, b, t in zip(nowc["rows"], goldc["rows"], gold["rows"]):
        assert a["gamma"] == b["gamma"]
        assert np.allclose(a["x"], b["x"], rtol=1e-12, atol=1e-14)
        assert np.allclose(a["z"], t["z"], rtol=1e-9, atol=1e-12)       # the two forms are the same operator
    from tests.golden.make_golden import pairs_alps
    with open(os.path.join(GOLD, "alps_pairs_n128.json")) as fh:
        goldp = json.load(fh)
    for kind, g in goldp.items():
        nowp = pairs_alps(g["n"], kind)
        assert nowp["status"] == g["status"] and nowp["tot_it"] == g["tot_it"] and nowp["tot_inner_it"] == g["tot_inner_it"]
        assert np.allclose(nowp["x"], g["x"], rtol=1e-12, atol=1e-14)


def test_als_rosenbrock_and_agrees_with_alps():
    """ALS (src/algorithms/als.jl) in the oracle: the rosenbrock demo runs both solvers and expects the
    same minimiser (0,0) (demo/rosenbrock.jl:131-136,186); on a convex problem ALS and ALPS agree."""
    warnings.simplefilter("ignore")
    sub = lambda **kw: R.PANOCplus(directions=R.LBFGS(5), maxit=10 ** 9, freq=10 ** 9, minimum_gamma=1e-32, **kw)
    for x1 in (-5.0, -1.25, 2.5):
        for x2 in (-3.75, 0.0, 5.0):
            out = R.als(R.SmoothCostRosenbrock(10.0), R.NonsmoothCostRosenbrock(1.0), R.ConstraintRosenbrock(),
                        R.SetRosenbrock(), np.array([x1, x2]), np.zeros(2), tol=1e-8, inner_tol=1.0,
                        subsolver=sub, subsolver_maxit=10 ** 9)
            assert out[5] == "first_order" and np.max(np.abs(out[0])) <= 1e-4
    import bazinga_jl_amd as bz
    n = 400
    d = bz.synth.l1_quadratic(n)
    orc = (R.DiagQuadratic(d["q"], d["b"]), R.NormL1(d["lam"]), R.IdentityFunction(), R.ClosedSet(R.IndBox(-1.0, 1.0)))
    a, b = R.als(*orc, np.zeros(n), np.zeros(n)), R.alps(*orc, np.zeros(n), np.zeros(n))
    assert a[5] == b[5] == "first_order" and np.max(np.abs(a[0] - b[0])) <= 1e-5
    # slack gradient is the gradient of the slack value
    rng = np.random.default_rng(0)
    m = 24
    mu, y, xs = rng.uniform(0.1, 1, m), rng.standard_normal(m), rng.standard_normal(2 * m)
    F = R.AugLagFunSlack(R.DiagQuadratic(rng.uniform(0.1, 2, m), rng.standard_normal(m)), R.IdentityFunction(), mu, y, xs[:m])
    g = np.empty(2 * m)
    v = F.gradient(g, xs)
    assert abs(v - F(xs)) <= 1e-12
    for i in range(0, 2 * m, 5):
        e = np.zeros(2 * m)
        e[i] = 1e-6
        assert abs((F(xs + e) - F(xs - e)) / 2e-6 - g[i]) <= 1e-6


@pytest.mark.parametrize("kind", ["vc", "cc", "eitheror", "xor"])
def test_pairwise_projections_are_nearest_points(kind):
    """The reference ships no test for its 2-element projections (vanishingConstraints.jl:27-46,
    complementarityConstraints.jl:8-20, orConstraints.jl:7-36): pin the restatement by what a projection
    is.  Each set is a union of two convex pieces; the result must lie in the set, be idempotent and be as
    close as the nearer of the two piece projections — on random points, axis points, ties and zeros."""
    def max0(v): return v if v > 0 else 0.0
    def min0(v): return v if v < 0 else 0.0
    pieces = {"vc": (lambda a, b: (0.0, b), lambda a, b: (max0(a), max0(b))),
              "cc": (lambda a, b: (0.0, max0(b)), lambda a, b: (max0(a), 0.0)),
              "eitheror": (lambda a, b: (max0(a), b), lambda a, b: (a, max0(b))),
              "xor": (lambda a, b: (max0(a), min0(b)), lambda a, b: (min0(a), max0(b)))}[kind]
    member = {"vc": lambda a, b: a >= 0 and a * b >= 0,
              "cc": lambda a, b: a >= 0 and b >= 0 and a * b == 0,
              "eitheror": lambda a, b: a >= 0 or b >= 0,
              "xor": lambda a, b: (a >= 0 and b <= 0) or (a <= 0 and b >= 0)}[kind]
    rng = np.random.default_rng(11)
    pts = [tuple(p) for p in rng.standard_normal((400, 2)) * 2]
    special = [-2.0, -1.0, -0.0, 0.0, 1.0, 2.0]
    pts += [(a, b) for a in special for b in special] + [(1.5, -1.5), (-1.5, 1.5), (0.3, -0.7), (0.7, -0.3)]
    D = R.PairwiseSet(kind)
    x = np.array(pts).reshape(-1)
    z = np.empty_like(x)
    D.proj(z, x)
    z2 = np.empty_like(x)
    D.proj(z2, z)
    assert np.array_equal(z, z2)                                   # idempotent
    for j, (a, b) in enumerate(pts):
        za, zb = z[2 * j], z[2 * j + 1]
        assert member(za, zb), (kind, a, b, za, zb)
        d = np.hypot(za - a, zb - b)
        best = min(np.hypot(p[0] - a, p[1] - b) for p in (f(a, b) for f in pieces))
        assert d <= best + 1e-15, (kind, a, b, za, zb)
