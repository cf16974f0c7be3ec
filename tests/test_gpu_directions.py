"""`directions` other than L-BFGS (SURVEY f-4; demo/rosenbrock.jl:96-103): AndersonAcceleration(n) and Broyden().

Both are EXTERNAL, UN-PINNED algorithms (ProximalAlgorithms.jl is not vendored and has no pinned version) that no shipped
script selects (rosenbrock.jl:275 picks LBFGS); the oracle restates them from the published methods (oracle/
bazinga_ref.py: AndersonAcceleration, Broyden) — PARITY UNPINNED.  What is checked: the device follows that restatement,
and the solves end where L-BFGS solves end."""
import warnings

import numpy as np
import pytest

from tests.test_gpu_parity import make_cfg2, rel

pytestmark = pytest.mark.gpu


def _trace(bz, ref, dev, orc, n, dirs_d, dirs_r, iters, mu, y, x0):
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, directions=dirs_d).c_opts(), x0)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, directions=dirs_r)
    st = it.init()
    errs = []
    for k in range(iters):
        errs.append((rel(prob.panoc_vector("x"), st.x), rel(prob.panoc_vector("z"), st.z),
                     abs(prob.panoc_scalars()["gamma"] - float(st.gamma)) / float(st.gamma)))
        if k + 1 < iters:
            prob.panoc_step()
            st = it.step(st)
    stats = prob.panoc_stats()
    prob.close()
    return errs, stats


def test_anderson_acceleration_follows_the_restatement(bz, ref):
    n = 4000
    d, dev, orc = make_cfg2(bz, ref, n)
    rng = np.random.default_rng(0)
    mu, y, x0 = np.full(n, 0.1), 0.2 * rng.standard_normal(n), np.zeros(n)
    errs, st = _trace(bz, ref, dev, orc, n, bz.AndersonAcceleration(5), ref.AndersonAcceleration(5), 14, mu, y, x0)
    for k, (ex, ez, eg) in enumerate(errs):
        # (a = (Y'Y)^-1 Y'v on the device — the Gram products its passes return — against a QR least squares in the
        # restatement: they agree to cond(Y)^2 eps)
        assert eg <= 1e-12 and ex <= 1e-6 and ez <= 1e-6, (k, ex, ez, eg)
    assert st.n_fused_iters >= 10                      # ... through the one-pass kernels (coefficients are just numbers)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = bz.alps(*dev, x0, np.zeros(n), subsolver=lambda **kw: bz.PANOCplus(directions=bz.AndersonAcceleration(5), **kw))
        o = bz.alps(*dev, x0, np.zeros(n))
    assert a[5] == o[5] == "first_order" and np.max(np.abs(a[0] - o[0])) <= 2e-5
    with pytest.raises(bz.UnsupportedOracle):
        bz.AndersonAcceleration(9)


def test_broyden_follows_the_restatement(bz, ref):
    n = 600
    d, dev, orc = make_cfg2(bz, ref, n)
    rng = np.random.default_rng(1)
    mu, y, x0 = np.full(n, 0.1), 0.2 * rng.standard_normal(n), np.zeros(n)
    errs, st = _trace(bz, ref, dev, orc, n, bz.Broyden(), ref.Broyden(), 20, mu, y, x0)
    for k, (ex, ez, eg) in enumerate(errs):
        assert eg <= 1e-12 and ex <= 1e-9 and ez <= 1e-9, (k, ex, ez, eg)
    a = bz.alps(*dev, x0, np.zeros(n), subsolver=lambda **kw: bz.PANOCplus(directions=bz.Broyden(), **kw))
    o = bz.alps(*dev, x0, np.zeros(n))
    assert a[5] == o[5] == "first_order" and np.max(np.abs(a[0] - o[0])) <= 2e-5


def test_rosenbrock_with_every_direction(bz, ref):
    """demo/rosenbrock.jl:85-136 selects the subsolver by name (noaccel / broyden / anderson / lbfgs): all of them through
    the product on the generic (callback) oracles.  L-BFGS, Broyden and NoAcceleration reach (0, 0) from every start
    tried; the Anderson restatement is checked against its oracle twin only (it stalls from some starts in the
    restatement too — un-pinned, see the module docstring)."""
    from tests.test_gpu_generic import ConstraintRosenbrock, NonsmoothCostRosenbrock, SetRosenbrock, SmoothCostRosenbrock
    warnings.simplefilter("ignore")
    f, g, c, D = SmoothCostRosenbrock(10.0), NonsmoothCostRosenbrock(1.0), ConstraintRosenbrock(), SetRosenbrock()
    prob = bz.Problem(f, g, c, D, 2, 2, np.float64)
    for name, dirs in (("lbfgs", bz.LBFGS(5)), ("broyden", bz.Broyden()), ("noaccel", bz.NoAcceleration())):
        sub = lambda **kw: bz.PANOCplus(directions=dirs, maxit=20000, minimum_gamma=1e-32, **kw)
        for x0 in ([-5.0, -2.5], [2.5, 5.0], [0.0, -2.5]):
            out = bz.alps(f, g, c, D, np.array(x0), np.zeros(2), tol=1e-8, inner_tol=1.0, subsolver=sub,
                          subsolver_maxit=10 ** 9, maxit=40, problem=prob)
            assert out[5] == "first_order" and np.max(np.abs(out[0])) <= 1e-4, (name, x0, out[5], out[0])
    sub = lambda **kw: bz.PANOCplus(directions=bz.AndersonAcceleration(5), maxit=3000, minimum_gamma=1e-32, **kw)
    rsub = lambda **kw: ref.PANOCplus(directions=ref.AndersonAcceleration(5), maxit=3000, minimum_gamma=1e-32, **kw)
    out = bz.alps(f, g, c, D, np.array([2.5, 5.0]), np.zeros(2), tol=1e-8, inner_tol=1.0, subsolver=sub, maxit=30, problem=prob)
    o = ref.alps(ref.SmoothCostRosenbrock(10.0), ref.NonsmoothCostRosenbrock(1.0), ref.ConstraintRosenbrock(),
                 ref.SetRosenbrock(), np.array([2.5, 5.0]), np.zeros(2), tol=1e-8, inner_tol=1.0, subsolver=rsub, maxit=30)
    assert out[5] == o[5]
    if o[5] == "first_order":
        assert np.max(np.abs(out[0] - o[0])) <= 1e-4
    prob.close()
