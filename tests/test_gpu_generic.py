"""Generic (user-defined) oracles through the PRODUCT entry point (missing item 1): f, g, c, D are
arbitrary host objects with the reference's protocol (gradient!/prox!/eval!/jtprod!/proj!, src/Bazinga.jl:11-16),
handed to the library as host callbacks (BZ_*_CALLBACK); the device keeps the L-BFGS / line-search vector work.
BASELINE config 1 — demo/rosenbrock.jl:39-80 (closures + 2x2 linear c + either-or D), starts on a grid, expected
minimiser (0, 0) (rosenbrock.jl:85-136,186) — runs through bz.alps, next to the oracle's run of the same problem."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class SmoothCostRosenbrock:                      # demo/rosenbrock.jl:39-50
    def __init__(self, w):
        self.w = w

    def gradient(self, dfx, x):
        tmp = x[1] + 1 - (x[0] + 1) ** 2
        dfx[0] = -4 * self.w * tmp * (x[0] + 1)
        dfx[1] = 2 * self.w * tmp
        return self.w * tmp ** 2


class NonsmoothCostRosenbrock:                   # demo/rosenbrock.jl:52-64
    def __init__(self, lam):
        self.lam = lam

    def prox(self, y, x, gamma):
        gl = gamma * self.lam
        y[0] = 0.0 if abs(x[0]) <= gl else np.sign(x[0]) * (abs(x[0]) - gl)
        y[1] = x[1]
        return self.lam * abs(y[0])


class ConstraintRosenbrock:                      # demo/rosenbrock.jl:66-74
    def eval(self, cx, x):
        cx[0] = -x[0] - x[1]
        cx[1] = x[1] - x[0]

    def jtprod(self, jtv, x, v):
        jtv[0] = -v[0] - v[1]
        jtv[1] = v[1] - v[0]


class SetRosenbrock:                             # demo/rosenbrock.jl:76-80 -> project_onto_EITHEROR_set!, orConstraints.jl:7-17
    def proj(self, z, cx):
        z[...] = cx
        if cx[0] < 0 and cx[1] < 0:
            if cx[0] > cx[1]:
                z[0] = 0
            else:
                z[1] = 0


def test_rosenbrock_config1_through_the_product(bz, ref):
    warnings.simplefilter("ignore")
    sub = lambda **kw: bz.PANOCplus(directions=bz.LBFGS(5), maxit=10 ** 9, freq=10 ** 9, minimum_gamma=1e-32, **kw)
    rsub = lambda **kw: ref.PANOCplus(directions=ref.LBFGS(5), maxit=10 ** 9, freq=10 ** 9, minimum_gamma=1e-32, **kw)
    f, g, c, D = SmoothCostRosenbrock(10.0), NonsmoothCostRosenbrock(1.0), ConstraintRosenbrock(), SetRosenbrock()
    prob = bz.Problem(f, g, c, D, 2, 2, np.float64)
    assert prob.generic
    checked = 0
    for x1 in np.arange(-5, 5.01, 2.5):
        for x2 in np.arange(-5, 5.01, 2.5):
            x0 = np.array([x1, x2])
            out = bz.alps(f, g, c, D, x0, np.zeros(2), tol=1e-8, inner_tol=1.0, subsolver=sub, subsolver_maxit=10 ** 9,
                          problem=prob)
            assert out[5] == "first_order", (x1, x2, out[5])
            assert np.max(np.abs(out[0])) <= 1e-4, (x1, x2, out[0])          # rosenbrock.jl:186: (0, 0)
            assert np.array_equal(x0, [x1, x2])                              # x0 never mutated
            if checked < 6:
                o = ref.alps(ref.SmoothCostRosenbrock(10.0), ref.NonsmoothCostRosenbrock(1.0), ref.ConstraintRosenbrock(),
                             ref.SetRosenbrock(), x0, np.zeros(2), tol=1e-8, inner_tol=1.0, subsolver=rsub,
                             subsolver_maxit=10 ** 9)
                assert o[5] == out[5] and o[2] == out[2], (x1, x2, o[2], out[2])
                assert abs(o[3] - out[3]) <= max(3, 0.1 * o[3])
                assert np.max(np.abs(o[0] - out[0])) <= 1e-6 and np.max(np.abs(o[1] - out[1])) <= 1e-5
                checked += 1
    prob.close()


def test_rosenbrock_host_outer_loop_and_subsolver_seam(bz, ref):
    """the same through the `subsolver` seam (alps.jl:64-66) with the outer loop on the host (resident=False)"""
    warnings.simplefilter("ignore")
    sub = lambda **kw: bz.PANOCplus(directions=bz.LBFGS(5), maxit=10 ** 9, freq=10 ** 9, minimum_gamma=1e-32, **kw)
    f, g, c, D = SmoothCostRosenbrock(10.0), NonsmoothCostRosenbrock(1.0), ConstraintRosenbrock(), SetRosenbrock()
    out = bz.alps(f, g, c, D, np.array([-2.5, 5.0]), np.zeros(2), tol=1e-8, inner_tol=1.0, subsolver=sub,
                  subsolver_maxit=10 ** 9, resident=False)
    assert out[5] == "first_order" and np.max(np.abs(out[0])) <= 1e-4


class Quartic:
    """a user-defined smooth cost no structured kind covers: f(x) = sum 0.25 (x_i - a_i)^4 + 0.5 x_i^2"""

    def __init__(self, a):
        self.a = a

    def gradient(self, dfx, x):
        d = x - self.a
        dfx[...] = d ** 3 + x
        return np.sum(0.25 * d ** 4 + 0.5 * x * x)


def test_generic_f_mixed_with_structured_g_c_D(bz, ref):
    """one generic oracle makes all four travel as callbacks: the structured Python types carry the host protocol.
    Checked against the oracle run on the same objects' restatements, and against first-order optimality."""
    n = 300
    rng = np.random.default_rng(1)
    a = rng.standard_normal(n) * 2
    f = Quartic(a)
    dev = (f, bz.NormL1(0.3), bz.IdentityFunction(), bz.ClosedSet(bz.IndBox(-1.0, 1.0)))

    class RefQuartic(Quartic):
        def __call__(self, x):
            d = x - self.a
            return np.sum(0.25 * d ** 4 + 0.5 * x * x)
    orc = (RefQuartic(a), ref.NormL1(0.3), ref.IdentityFunction(), ref.ClosedSet(ref.IndBox(-1.0, 1.0)))
    out = bz.alps(*dev, np.zeros(n), np.zeros(n), tol=1e-7)
    o = ref.alps(*orc, np.zeros(n), np.zeros(n), tol=1e-7)
    assert out[5] == o[5] == "first_order" and out[2] == o[2]
    assert np.max(np.abs(out[0] - o[0])) <= 1e-6
    assert np.max(np.abs(out[0])) <= 1.0 + 1e-6                        # c(x) = x in D to tol_prim


def test_callback_exception_surfaces_as_python_error(bz):
    class Bad:
        def gradient(self, dfx, x):
            raise ZeroDivisionError("boom")
    n = 4
    with pytest.raises(bz.CallbackError) as ei:
        bz.alps(Bad(), bz.NormL1(0.1), bz.IdentityFunction(), bz.FreeSet(), np.ones(n), np.zeros(n))
    assert isinstance(ei.value.__cause__, ZeroDivisionError)


def test_failed_callback_ends_the_library_call_at_once(bz):
    """The error channel of the callbacks (bz_callback_abort, include/bazinga_hip.h): a callback that raises parks its
    exception and asks the library to end the call in progress.  Nothing more is evaluated — no further callback is invoked,
    the solve does not iterate on the stale buffers the failed callback left (it used to go on until the
    objective turned NaN, calling the failing callback again and again)."""
    calls = {"f": 0, "c": 0, "after": 0}
    state = {"failed": False}

    class F:
        def gradient(self, dfx, x):
            calls["f"] += 1
            calls["after"] += state["failed"]
            dfx[...] = x - 1.0
            return 0.5 * float(np.sum((x - 1.0) ** 2))

    class Cmap:
        def eval(self, cx, x):
            calls["c"] += 1
            calls["after"] += state["failed"]
            if calls["c"] == 7:
                state["failed"] = True
                raise KeyError("seventh eval")
            cx[...] = x

        def jtprod(self, jtv, x, v):
            calls["after"] += state["failed"]
            jtv[...] = v

    n = 6
    with pytest.raises(bz.CallbackError) as ei:
        bz.alps(F(), bz.NormL1(0.1), Cmap(), bz.ClosedSet(bz.IndBox(-0.5, 0.5)), np.zeros(n), np.zeros(n))
    assert isinstance(ei.value.__cause__, KeyError)
    assert calls["c"] == 7 and calls["after"] == 0
    # the library is usable afterwards (the abort request does not outlive the call it ended)
    state["failed"] = False
    calls["c"] = 100
    out = bz.alps(F(), bz.NormL1(0.1), Cmap(), bz.ClosedSet(bz.IndBox(-0.5, 0.5)), np.zeros(n), np.zeros(n))
    assert out[5] == "first_order"


def test_object_without_the_protocol_is_rejected(bz):
    with pytest.raises(bz.UnsupportedOracle):
        bz.Problem(object(), bz.NormL1(0.1), bz.IdentityFunction(), bz.FreeSet(), 4, 4, np.float64)
