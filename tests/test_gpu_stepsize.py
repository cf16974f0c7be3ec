"""The step-size keywords of PANOCplus (`gamma`, `Lf`, `adaptive`: upstream's `Lf = nothing`, `gamma = Lf === nothing ?
nothing : alpha / Lf`, `adaptive = gamma === nothing`) and the warm-started outer loop built on them (SURVEY 8(f-1):
"warm-start γ across outer iterations as an opt-in deviation from alps.jl:64").

The oracle restates the same keywords (`oracle/bazinga_ref.py: PANOCplusIteration.__init__/init/step`,
`alps(warm_start=True)`); like every iterate-level check, parity is with that restatement (parity unpinned)."""
import numpy as np
import pytest

from tests.test_gpu_parity import RTOL_ITER, _err, make_cfg2, make_cfg3, make_cfg4, rel

pytestmark = pytest.mark.gpu


def _side_by_side(bz, ref, dev, orc, n, ny, mu, y, x0, iters, kw, dtype=np.float64, compact=None):
    prob = bz.Problem(*dev, n, ny, dtype)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=1e-30, directions=bz.LBFGS(5, compact=compact),
                                  **kw).c_opts(), x0)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, minimum_gamma=1e-30, **kw)
    st = it.init()
    rows = []
    for k in range(iters):
        sc = prob.panoc_scalars()
        rows.append((k + 1, _err(prob.panoc_vector("x"), st.x), _err(prob.panoc_vector("z"), st.z), sc["gamma"],
                     float(st.gamma)))
        if k + 1 < iters:
            prob.panoc_step()
            st = it.step(st)
    stats = prob.panoc_finish()[1]
    prob.close()
    return rows, st, stats


@pytest.mark.parametrize("form", ["default", "two-loop"])
@pytest.mark.parametrize("kw", [dict(gamma=0.02), dict(gamma=0.5, adaptive=True), dict(Lf=30.0),
                                dict(Lf=30.0, adaptive=True), dict(gamma=0.02, adaptive=False)])
def test_given_step_size_iterates_match_oracle(bz, ref, kw, form):
    """cfg 2 family, 25 states: a given gamma (or Lf) is taken as it is — no Lipschitz estimate — and halved only when
    `adaptive` says so.  gamma = 0.5 is too large for this problem (L = max q + 1/mu ≈ 20): with `adaptive` the start
    halves it exactly as the oracle does."""
    n = 20011
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    rows, st, stats = _side_by_side(bz, ref, dev, orc, n, n, mu, y, x0, 25, kw, compact=None if form == "default" else False)
    adaptive = kw.get("adaptive", False)
    g0 = kw["gamma"] if "gamma" in kw else 0.95 / kw["Lf"]
    for k, ex, ez, g_d, g_r in rows:
        assert abs(g_d - g_r) <= 1e-15 * g_r, f"gamma differs at k={k}: {g_d} {g_r}"
        if not adaptive:
            assert g_d == np.float64(g0)
        assert ex <= RTOL_ITER and ez <= RTOL_ITER, f"iterate mismatch at k={k}: {ex} {ez}"
    if adaptive and g0 == 0.5:
        assert rows[0][3] < 0.5 and stats.n_gamma_halvings == st.n_gamma_halvings >= 1
    if not adaptive:
        assert stats.n_gamma_halvings == 0


def test_given_step_size_skips_the_lipschitz_estimate(bz, ref):
    """generic start (stencil f: no one-pass start kernel): the estimate costs one AL gradient at x + 1"""
    nx, ny_ = 64, 96
    d, n, dev, orc = make_cfg3(bz, ref, nx, ny_)
    mu, y = np.full(n, 0.1), np.zeros(n)
    counts = {}
    kw = {}
    for name in ("estimate", "given"):
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=1e-30, **kw).c_opts(), d["x0"])
        st = prob.panoc_finish()[1]
        counts[name] = (st.n_grad, st.n_gamma_halvings, st.gamma)
        prob.close()
        kw = dict(gamma=st.gamma, adaptive=True)          # the step size the estimate + its halvings arrived at
    # estimate: gradients at x, x + 1 and z, one more at z per halving; given: at x and z
    assert counts["estimate"][0] == 3 + counts["estimate"][1]
    assert counts["given"] == (2, 0, counts["estimate"][2])


def test_step_size_keyword_errors(bz, ref):
    n = 100
    d, dev, orc = make_cfg2(bz, ref, n)
    with pytest.raises(ValueError):
        bz.PANOCplus(gamma=-1.0)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(np.full(n, 0.1), np.zeros(n))
    o = bz.PANOCplus().c_opts()
    o.adaptive = 7
    with pytest.raises(bz.BazingaHipError, match="adaptive"):
        prob.panoc_begin(o, np.zeros(n))
    o = bz.PANOCplus().c_opts()
    o.Lf = -2.0
    with pytest.raises(bz.BazingaHipError, match="gamma and Lf"):
        prob.panoc_begin(o, np.zeros(n))
    prob.close()


@pytest.mark.parametrize("case", ["cfg2", "cfg2-free", "cfg3", "cfg4"])
def test_warm_started_alps_matches_the_warm_started_oracle(bz, ref, case):
    """alps(warm_start=True): resident device loop, host loop with the device subsolver and the oracle with the same
    option agree on counts, solution, multipliers and penalties; and the option changes nothing about the answer (same
    solution as the reference loop to the solve's tolerance)."""
    sub = None
    if case.startswith("cfg2"):
        n = 3000
        d, dev, orc = make_cfg2(bz, ref, n, D="free" if case.endswith("free") else "box")
        x0, y0, dt = np.zeros(n), np.zeros(n), np.float64
        kw = {}
    elif case == "cfg3":
        d, n, dev, orc = make_cfg3(bz, ref, 24, 32, load=-1.0)
        x0, y0, dt = d["x0"].copy(), np.zeros(n), np.float64
        eps = float(np.finfo(float).eps)
        kw = dict(tol=1e-7)
        sub = lambda R: (lambda **k: R.PANOCplus(maxit=100000, minimum_gamma=eps, **k))
    else:
        ny, n = 20, 100
        d, dev, orc = make_cfg4(bz, ref, ny, n, np.float64, density=0.1)
        x0, y0, dt = np.zeros(n), np.zeros(ny), np.float64
        kw = {}
    kwo = dict(kw, subsolver=sub(ref)) if sub else kw
    kwd = dict(kw, subsolver=sub(bz)) if sub else kw
    o_cold = ref.alps(*orc, x0, y0, **kwo)
    o = ref.alps(*orc, x0, y0, warm_start=True, **kwo)
    a = bz.alps(*dev, x0, y0, warm_start=True, resident=True, **kwd)
    b = bz.alps(*dev, x0, y0, warm_start=True, resident=False, **kwd)
    assert o[5] == o_cold[5] == "first_order"
    for r in (a, b):
        assert r[5] == "first_order"
        assert r[2] == o[2], f"outer counts {r[2]} vs {o[2]}"
        assert abs(r[3] - o[3]) <= (0 if case.startswith("cfg2") else max(3, o[3] // 20)), f"inner counts {r[3]} vs {o[3]}"
        assert rel(r[0], o[0]) <= (1e-9 if case.startswith("cfg2") else 2e-5)
        assert rel(r[9], o[9]) <= 1e-12
    # the same answer as the reference loop (to what tol = 1e-6 / 1e-7 resolves)
    assert rel(a[0], o_cold[0]) <= 1e-4


def test_unknown_warm_start_bits_are_refused(bz, ref):
    n = 64
    d, dev, orc = make_cfg2(bz, ref, n)
    with pytest.raises(bz.BazingaHipError, match="warm_start"):
        bz.alps(*dev, np.zeros(n), np.zeros(n), warm_start=2)


def test_warm_start_saves_gradient_evaluations(bz, ref):
    """one Lipschitz estimate for the whole solve instead of one per subproblem"""
    ny, n = 20, 100
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float64, density=0.1)
    x0, y0 = np.zeros(n), np.zeros(ny)
    tot = {}
    for warm in (False, True):
        grads = []

        class Sub(bz.PANOCplus):
            def __call__(self, **k):
                r = super().__call__(**k)
                grads.append((self.gamma is not None, self.stats.n_grad, self.stats.iters))
                return r
        out = bz.alps(*dev, x0, y0, warm_start=warm, resident=False, subsolver=Sub)
        assert out[5] == "first_order"
        tot[warm] = grads
    assert [g[0] for g in tot[False]] == [False] * len(tot[False])
    assert [g[0] for g in tot[True]] == [False] + [True] * (len(tot[True]) - 1)


@pytest.mark.parametrize("form", ["default", "two-loop"])
def test_adaptive_false_without_a_step_size_still_backtracks(bz, ref, form):
    """`PANOCplus(adaptive = false)` with neither `gamma` nor `Lf`: upstream tests `iter.gamma === nothing || iter.adaptive ==
    true` at both halving sites, so the ESTIMATED step size is still backtracked (the device took `adaptive` alone
    and never halved).  cfg 2's Lipschitz estimate is 2x optimistic — every solve halves at its start — so the two readings
    differ from the first state on."""
    n = 20011
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    rows, st, stats = _side_by_side(bz, ref, dev, orc, n, n, mu, y, x0, 25, dict(adaptive=False),
                                    compact=None if form == "default" else False)
    assert st.n_gamma_halvings >= 1 and stats.n_gamma_halvings == st.n_gamma_halvings
    for k, ex, ez, g_d, g_r in rows:
        assert abs(g_d - g_r) <= 1e-15 * g_r, f"gamma differs at k={k}: {g_d} {g_r}"
        assert ex <= RTOL_ITER and ez <= RTOL_ITER, f"iterate mismatch at k={k}: {ex} {ez}"
