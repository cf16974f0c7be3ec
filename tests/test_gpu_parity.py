"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Element-wise outputs must be BIT-EXACT (same operation order, no FMA
contraction); reduced scalars agree to summation-order rounding (<= 1e-13 relative);
iterates of the full solve agree within the north-star tolerance 1e-10 relative.

Oracle provenance and its "parity unpinned" status at iterate level: oracle/bazinga_ref.py.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_ITER = 1e-10      # BASELINE.json north_star: iterates within 1e-10 relative


def make_cfg2(bz, ref, n, seed_start=0, D="box", dtype=np.float64, g="l1"):
    d = bz.synth.l1_quadratic(n, start=seed_start, dtype=dtype)
    f_d = bz.DiagQuadratic(d["q"], d["b"])
    f_r = ref.DiagQuadratic(d["q"], d["b"])
    if g == "l1":
        g_d, g_r = bz.NormL1(d["lam"]), ref.NormL1(d["lam"])
    elif g == "nonneg":
        g_d, g_r = bz.NormL1Nonneg(d["lam"]), ref.NormL1Nonneg(d["lam"])
    elif g == "l1box":
        u = np.full(n, 0.75, dtype)
        g_d, g_r = bz.NormL1Box(d["lam"], u=u), ref.NormL1Box(d["lam"], u=u)
    elif g == "l0box":
        u = np.where(np.arange(n) % 7 == 0, 0.0, 0.6).astype(dtype)
        g_d, g_r = bz.NormL0Box(0.3, u=u), ref.NormL0Box(0.3, u=u)
    elif g == "lpnonneg":
        g_d, g_r = bz.NormLpPowerNonneg(0.5, alpha=0.8), ref.NormLpPowerNonneg(dtype(0.5), alpha=dtype(0.8))
    elif g == "lpbox":
        u = np.where(np.arange(n) % 5 == 0, 0.0, 0.9).astype(dtype)
        g_d, g_r = bz.NormLpPowerBox(0.5, 0.8, u=u), ref.NormLpPowerBox(dtype(0.5), dtype(0.8), u=u)
    elif g == "indbox":
        g_d, g_r = bz.IndBox(-0.5, 0.5), ref.IndBox(dtype(-0.5), dtype(0.5))
    else:
        g_d, g_r = bz.Zero(), ref.Zero()
    if D == "box":
        D_d, D_r = bz.ClosedSet(bz.IndBox(d["lo"], d["hi"])), ref.ClosedSet(ref.IndBox(dtype(d["lo"]), dtype(d["hi"])))
    elif D == "free":
        D_d, D_r = bz.FreeSet(), ref.FreeSet()
    else:
        D_d, D_r = bz.ZeroSet(), ref.ZeroSet()
    return d, (f_d, g_d, bz.IdentityFunction(), D_d), (f_r, g_r, ref.IdentityFunction(), D_r)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.mark.parametrize("n", [1, 2, 5, 1000, 4097, 262147])
@pytest.mark.parametrize("D", ["box", "free", "zero"])
def test_al_gradient_bit_exact(bz, ref, n, D):
    """K1: gradient!(dlx, al, x) (auglagfun.jl:73-86) element-wise bit-exact, value to 1e-13."""
    d, dev, orc = make_cfg2(bz, ref, n, D=D)
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    mu = 10.0 ** rng.uniform(-3, 1, n)
    y = rng.standard_normal(n) * 3
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x)
    g_ref = np.empty(n)
    lx = al.gradient(g_ref, x)
    assert np.array_equal(g_dev, g_ref)
    assert abs(vals[0] - lx) <= 1e-13 * max(1.0, abs(lx))
    assert abs(vals[1] - al.fx) <= 1e-13 * max(1.0, abs(al.fx))
    prob.close()


@pytest.mark.parametrize("g", ["l1", "nonneg", "l1box", "l0box", "indbox", "zero"])
@pytest.mark.parametrize("n", [3, 1001, 70001])
def test_prox_bit_exact(bz, ref, n, g):
    """K3 prox!(z, g, x, gamma): soft-threshold family, element-wise bit-exact."""
    d, dev, orc = make_cfg2(bz, ref, n, g=g)
    rng = np.random.default_rng(7 * n + 1)
    x = rng.standard_normal(n) * 2
    gamma = 0.37
    prob = bz.Problem(*dev, n, n, np.float64)
    z_dev, gz_dev = prob.eval_prox(x, gamma)
    z_ref = np.empty(n)
    gz_ref = orc[1].prox(z_ref, x, gamma)
    assert np.array_equal(z_dev, z_ref)
    assert abs(gz_dev - gz_ref) <= 1e-13 * max(1.0, abs(gz_ref))
    prob.close()


@pytest.mark.parametrize("m", [0, 1, 2, 5])
def test_lbfgs_two_loop(bz, ref, m):
    """K4: d = H v through the chained axpy+dot kernels vs the oracle's two-loop."""
    n = 5003
    rng = np.random.default_rng(m)
    S = [rng.standard_normal(n) for _ in range(m)]
    Y = [s * rng.uniform(0.5, 2.0, n) for s in S]          # <s,y> > 0
    v = rng.standard_normal(n)
    d, dev, orc = make_cfg2(bz, ref, n)
    prob = bz.Problem(*dev, n, n, np.float64)
    d_dev = prob.eval_lbfgs(S, Y, v)
    H = ref.LBFGSOperator(max(1, m), v)
    for s, y in zip(S, Y):
        H.update(s, y)
    d_ref = np.empty(n)
    H.mul(d_ref, v)
    assert rel(d_dev, d_ref) <= 1e-12
    prob.close()


class LongDoubleReducer:
    """The oracle's reductions carried in extended precision: a second, equally valid rounding of the
    same restatement.  |oracle(default sums) - oracle(extended sums)| is the restatement's OWN
    rounding sensitivity (line-search / active-set dynamics amplify last-bit
    differences of the reduced scalars); no implementation can be closer to "the" iterates than that."""

    def sum(self, v):
        return v.dtype.type(np.sum(v.astype(np.longdouble)))

    def dot(self, a, b):
        return a.dtype.type(np.sum(a.astype(np.longdouble) * b.astype(np.longdouble)))

    def max(self, v):
        return np.max(v) if v.size else v.dtype.type(0)

    def any(self, m):
        return bool(np.any(m))


def _err(a, b):
    return rel(a, b) if np.any(b) else float(np.max(np.abs(a - b)))


def run_traces(bz, ref, dev, orc, n, mu, y, x0, iters, fuse=True, minimum_gamma=1e-7, dtype=np.float64,
               ny=None, compact=None, affine_refresh=16):
    """Step the device solver and the oracle side by side.  Returns rows
    (k, err_x, err_z, gamma_dev, gamma_ref, stop_dev, stop_ref, fused, self_sensitivity)."""
    ny = n if ny is None else ny
    prob = bz.Problem(*dev, n, ny, dtype)
    prob.set_multipliers(mu, y)
    sub = bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=minimum_gamma, fuse=fuse, directions=bz.LBFGS(5, compact=compact),
                       affine_refresh=affine_refresh)
    prob.panoc_begin(sub.c_opts(), x0)
    its, sts = [], []
    for red in (None, LongDoubleReducer()):
        ref.set_reducer(red)
        al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
        it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, minimum_gamma=minimum_gamma)
        its.append(it)
        sts.append(it.init())
    ref.set_reducer(None)
    rows = []
    env = 0.0
    for k in range(iters):
        st = sts[0]
        sc = prob.panoc_scalars()
        xd, zd = prob.panoc_vector("x"), prob.panoc_vector("z")
        env = max(env, _err(sts[1].x, st.x), _err(sts[1].z, st.z))
        rows.append((k + 1, _err(xd, st.x), _err(zd, st.z), sc["gamma"], float(st.gamma), sc["stop_norm"],
                     float(its[0].stop_norm(st)), sc["fused"], env))
        if k + 1 < iters:
            prob.panoc_step()
            sts[0] = its[0].step(sts[0])
            ref.set_reducer(LongDoubleReducer())
            sts[1] = its[1].step(sts[1])
            ref.set_reducer(None)
    return prob, sts[0], rows


def iter_tol(self_sens):
    """North-star tolerance 1e-10 relative, widened only where the restatement's own rounding
    sensitivity (see LongDoubleReducer) already exceeds it."""
    return max(RTOL_ITER, 100.0 * self_sens)


@pytest.mark.parametrize("n", [1000, 4097, 200003])
@pytest.mark.parametrize("D", ["box", "free"])
@pytest.mark.parametrize("form", ["default", "two-loop"])
def test_panoc_iterates_match_oracle(bz, ref, n, D, form):
    """Per-iteration parity of x and z over the first 30 PANOCplus iterations
    (two different multiplier settings: y = 0 as in outer iteration 1, and y != 0) against the oracle in the
    reference's two-loop form — for the library default (the compact representation on this separable path)
    and for the two-loop kernels."""
    d, dev, orc = make_cfg2(bz, ref, n, D=D)
    rng = np.random.default_rng(3)
    for y in (np.zeros(n), rng.standard_normal(n)):
        mu = np.full(n, 0.1)
        x0 = np.zeros(n)
        prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, x0, 30, compact=None if form == "default" else False)
        for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
            assert abs(g_d - g_r) <= 1e-13 * g_r, f"gamma differs at k={k}"
            assert ex <= RTOL_ITER and ez <= RTOL_ITER, f"iterate mismatch at k={k}: {ex} {ez}"
            assert abs(sn_d - sn_r) <= 1e-9 * max(1.0, sn_r)
        assert sum(r[7] for r in rows) >= 20       # the fused fast path actually served the iterations
        prob.close()


@pytest.mark.parametrize("g", ["l1", "nonneg", "l1box", "l0box", "indbox", "zero"])
def test_fused_equals_generic_bitwise(bz, ref, g):
    """The single-pass fused kernel and the generic kernel chain are the same arithmetic:
    iterates and scalars must be identical to the last bit."""
    n = 50001
    d, dev, orc = make_cfg2(bz, ref, n, g=g)
    rng = np.random.default_rng(11)
    mu = 10.0 ** rng.uniform(-2, 0, n)
    y = rng.standard_normal(n)
    x0 = rng.standard_normal(n) * 0.1
    out = []
    for fuse in (True, False):
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        # (the two-loop form: with the compact one p, w are summed by different kernels in the two runs)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, fuse=fuse, directions=bz.LBFGS(5, compact=False)).c_opts(), x0)
        for _ in range(25):
            prob.panoc_step()
        out.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_vector("res"),
                    prob.panoc_scalars()))
        prob.close()
    (x1, z1, r1, s1) = out[0]
    for (x2, z2, r2, s2) in out[1:]:
        assert np.array_equal(x1, x2) and np.array_equal(z1, z2) and np.array_equal(r1, r2)
        for key in ("gamma", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_H", "al_z"):
            assert s1[key] == s2[key], key
    assert s1["fused"] == 1.0 and s2["fused"] == 0.0


def test_alps_matches_oracle_small(bz, ref):
    """Whole ALPS solve (alps.jl:7-117): resident device outer loop, host outer loop with
    the device subsolver, and the oracle agree on the solution, multipliers and counts."""
    n = 3000
    d, dev, orc = make_cfg2(bz, ref, n)
    x0, y0 = np.zeros(n), np.zeros(n)
    x0b = x0.copy()
    o = ref.alps(*orc, x0, y0)
    a = bz.alps(*dev, x0, y0)
    b = bz.alps(*dev, x0, y0, resident=False)
    assert np.array_equal(x0, x0b)                      # test_nonconvex_qp.jl:36
    for r in (a, b):
        assert r[5] == o[5] == "first_order"
        assert r[2] == o[2] and r[3] == o[3]
        assert rel(r[0], o[0]) <= 1e-9
        assert np.max(np.abs(r[1] - o[1])) <= 1e-8 * max(1.0, np.max(np.abs(o[1])))
        assert rel(r[9], o[9]) <= 1e-12
    # first-order fixed point of the box-constrained l1 problem
    x = a[0]
    assert np.all(np.abs(x) <= 1 + 1e-6)


def test_alps_free_set_closed_form(bz, ref):
    """cfg 2a (SURVEY §8(c) KAT 5): with D = FreeSet the AL term vanishes and the minimiser is
    x_i = soft(b_i, lambda)/q_i."""
    n = 100003
    d, dev, orc = make_cfg2(bz, ref, n, D="free")
    out = bz.alps(*dev, np.zeros(n), np.zeros(n), tol=1e-9)
    xs = np.sign(d["b"]) * np.maximum(np.abs(d["b"]) - d["lam"], 0) / d["q"]
    assert out[5] == "first_order"
    assert np.max(np.abs(out[0] - xs)) <= 1e-7


def test_mu_must_be_positive(bz, ref):
    n = 100
    d, dev, orc = make_cfg2(bz, ref, n)
    prob = bz.Problem(*dev, n, n, np.float64)
    mu = np.ones(n)
    mu[17] = 0.0
    with pytest.raises(ValueError, match="must be positive"):      # auglagfun.jl:92-93
        prob.set_multipliers(mu, np.zeros(n))
    prob.close()


def test_float32_path(bz, ref):
    """T = Float32 (cfg 4's eltype): element-wise bit-exact vs the float32 oracle."""
    n = 40003
    d, dev, orc = make_cfg2(bz, ref, n, dtype=np.float32)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n).astype(np.float32)
    mu = (10.0 ** rng.uniform(-2, 0, n)).astype(np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    prob = bz.Problem(*dev, n, n, np.float32)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x)
    g_ref = np.empty(n, np.float32)
    lx = al.gradient(g_ref, x)
    assert np.array_equal(g_dev, g_ref)
    assert abs(vals[0] - float(lx)) <= 2e-4 * max(1.0, abs(float(lx)))
    z_dev, gz = prob.eval_prox(x, 0.3)
    z_ref = np.empty(n, np.float32)
    orc[1].prox(z_ref, x, np.float32(0.3))
    assert np.array_equal(z_dev, z_ref)
    prob.close()


def test_float32_iterate_history_form_is_bitwise_neutral(bz, ref):
    """The fp32 instantiation of the one-pass kernel (4 elements per 16-byte pack, hardware division): the
    9..11-pass form (BZ_XR=2) against the stored-pair form (BZ_XR=0) on one grid — identical bits over a run with
    rejected trial points — and both against the float32 oracle's iterates to fp32 accuracy."""
    n = 50003
    d, dev, orc = make_cfg2(bz, ref, n, dtype=np.float32)
    mu = np.full(n, 0.1, np.float32)
    y = np.zeros(n, np.float32)
    x0 = np.zeros(n, np.float32)
    runs = []
    try:
        os.environ["BZ_GFC"] = "2"
        os.environ["BZ_TRIALFUSE"] = "0"
        for xr in ("0", "2"):
            os.environ["BZ_XR"] = xr
            prob = bz.Problem(*dev, n, n, np.float32)
            prob.set_multipliers(mu, y)
            prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=float(np.finfo(np.float32).eps)).c_opts(), x0)
            prob.profile_enable(True)
            for _ in range(90):
                prob.panoc_step()
            st = prob.panoc_stats()
            runs.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars(),
                         prob.profile()["k_fused_iterates"]["launches"], st.n_fused_iters))
            prob.close()
    finally:
        os.environ.pop("BZ_XR", None)
        os.environ.pop("BZ_GFC", None)
        os.environ.pop("BZ_TRIALFUSE", None)
    a, b = runs
    assert a[3] == 0 and b[3] >= 80
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for key in ("gamma", "f_x", "g_z", "ss_res", "stop_norm"):
        assert a[2][key] == b[2][key], key
    assert a[0].dtype == np.float32 and np.all(np.isfinite(a[0]))
    # the solve makes progress in fp32: the fixed-point residual has dropped by orders of magnitude
    assert b[2]["stop_norm"] < 1e-2


@pytest.mark.parametrize("variant", ["two-loop", "compact", "nofuse", "large-compact", "large-two-loop"])
def test_allgather_plumbing_single_rank(bz, ref, variant):
    """The multi-GPU scalar paths on ONE rank — k_pack -> RCCL all-gather -> fold over ranks, and the
    self-loop of the peer-to-peer mailboxes — must reproduce the no-communicator iterates bit for bit
    (the per-rank fold is the same fold, in the same order).  'large' sizes use capped grids, so slots
    written by different kernels hold different numbers of block partials."""
    n = 600_011 if variant.startswith("large") else 30011
    d, dev, orc = make_cfg2(bz, ref, n)
    rng = np.random.default_rng(2)
    mu = np.full(n, 0.1)
    y = rng.standard_normal(n)
    # persist=False: the persistent two-loop kernel folds 256 block partials, the kernel chain 2048 — the
    # transports must be compared on the same kernels
    opts = bz.PANOCplus(tol=0.0, maxit=10 ** 9, fuse=variant != "nofuse", persist=False,
                        directions=bz.LBFGS(5, compact="compact" in variant)).c_opts()
    res = []
    for transport in ("none", "rccl", "p2p"):
        ctx = bz.Context(device=0, rank=0, nranks=1, comm_id=bz.Context.unique_id() if transport == "rccl" else None)
        if transport == "p2p":
            ctx.p2p_connect([ctx.p2p_export()], [0])
        prob = bz.Problem(*dev, n, n, np.float64, ctx)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(opts, np.zeros(n))
        for _ in range(14):
            prob.panoc_step()
        prob.panoc_steps(26)          # (the library's own loop: with the compact form the next pass is launched early, gated)
        res.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars(), prob.panoc_stats()))
        prob.close()
        ctx.close()
    if "compact" in variant:
        assert all(r[3].n_gated_launches >= 15 for r in res)      # ... behind the exchange on both transports too
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0]) and np.array_equal(res[0][1], other[1])
        assert res[0][2]["stop_norm"] == other[2]["stop_norm"] and res[0][2]["gamma"] == other[2]["gamma"]
        assert res[0][3].n_grad == other[3].n_grad and res[0][3].n_backtracks == other[3].n_backtracks


def test_full_size_properties(bz, ref):
    """BASELINE size n = 10^7 (too big for the numpy oracle to iterate): size-independent
    properties.  (1) D = FreeSet closed form x_i = soft(b_i, lambda)/q_i;  (2) with D = Box the
    returned point is a fixed point of the forward-backward map of the final AL subproblem and
    satisfies the box to tol_prim; (3) x0 is not mutated."""
    n = 10_000_000
    d = bz.synth.l1_quadratic(n)
    f, g, c = bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction()
    x0 = np.zeros(n)
    out = bz.alps(f, g, c, bz.FreeSet(), x0, np.zeros(n), tol=1e-8)
    xs = np.sign(d["b"]) * np.maximum(np.abs(d["b"]) - d["lam"], 0) / d["q"]
    assert out[5] == "first_order"
    assert np.max(np.abs(out[0] - xs)) <= 1e-6
    out = bz.alps(f, g, c, bz.ClosedSet(bz.IndBox(-1.0, 1.0)), x0, np.zeros(n))
    x, y, mu = out[0], out[1], out[9]
    assert out[5] == "first_order" and not np.any(x0)
    assert np.max(np.abs(x - np.clip(x, -1, 1))) <= 1e-6            # primal feasibility (alps.jl:84,87)
    # KKT of  min f + g  s.t. x in [-1,1]:  0 in q x - b + y + lam sign(x)
    r = d["q"] * x - d["b"] + y
    viol = np.where(x > 1e-9, np.abs(r + d["lam"]), np.where(x < -1e-9, np.abs(r - d["lam"]),
                    np.maximum(np.abs(r) - d["lam"], 0)))
    assert np.max(viol) <= 1e-4


# ------------------------------------------------------------------ cfg 3: 5-pt stencil QP, box D
def make_cfg3(bz, ref, nx, ny, load=1.0):
    d = bz.synth.obstacle_grid(nx, ny, load=load)
    n = nx * ny
    dev = (bz.Stencil5ptQuadratic(nx, ny, d["b"]), bz.Zero(), bz.IdentityFunction(),
           bz.ClosedSet(bz.IndBox(d["psi"], np.inf)))
    orc = (ref.Stencil5ptQuadratic(nx, ny, d["b"]), ref.Zero(), ref.IdentityFunction(),
           ref.ClosedSet(ref.IndBox(d["psi"], np.inf)))
    return d, n, dev, orc


@pytest.mark.parametrize("shape", [(1, 2), (2, 2), (3, 8), (17, 34), (64, 128), (2048, 2048)])
def test_stencil_al_gradient_bit_exact(bz, ref, shape):
    """K10+K1: gradient!(dlx, al, x) with the 5-point-stencil f — bit-exact up to the FULL BASELINE
    size 2048^2 (one numpy evaluation is cheap), every boundary case included."""
    nx, ny = shape
    d, n, dev, orc = make_cfg3(bz, ref, nx, ny)
    rng = np.random.default_rng(nx * 7 + ny)
    x = rng.standard_normal(n)
    mu = 10.0 ** rng.uniform(-3, 0, n)
    y = rng.standard_normal(n)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x)
    g_ref = np.empty(n)
    lx = al.gradient(g_ref, x)
    assert np.array_equal(g_dev, g_ref)
    assert abs(vals[0] - lx) <= 1e-12 * max(1.0, abs(lx))
    assert abs(vals[1] - al.fx) <= 1e-12 * max(1.0, abs(al.fx))
    prob.close()


@pytest.mark.parametrize("form", ["default", "two-loop"])
@pytest.mark.parametrize("shape,iters", [((16, 32), 40), ((200, 128), 25), ((2048, 2048), 8)])
def test_stencil_panoc_iterates_match_oracle(bz, ref, shape, iters, form):
    """cfg 3 against the two-loop oracle, in the library default (compact representation: x_d, k_stencil_fb,
    k_stencil_update_c — one reduction phase) and with the two-loop kernels (reference operation order)."""
    nx, ny = shape
    d, n, dev, orc = make_cfg3(bz, ref, nx, ny)
    mu = np.full(n, 0.1)
    y = np.zeros(n)
    prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, d["x0"].copy(), iters,
                                minimum_gamma=float(np.finfo(float).eps), compact=None if form == "default" else False)
    p2 = prob.profile2()
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert abs(g_d - g_r) <= 1e-13 * g_r
        # ill-conditioned Laplacian + active-set changes: errors of ANY two roundings grow
        # geometrically after ~10 states; the device must stay within the oracle's own sensitivity
        assert ex <= iter_tol(sens) and ez <= iter_tol(sens), f"iterate mismatch at k={k}: {ex} {ez} (self {sens})"
        if k <= 12:
            assert ex <= RTOL_ITER and ez <= RTOL_ITER
    prob.close()


def test_stencil_compact_form_is_the_default_and_runs_without_the_persistent_kernel(bz, ref):
    nx, ny = 1024, 1024
    d, n, dev, orc = make_cfg3(bz, ref, nx, ny)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(np.full(n, 0.1), np.zeros(n))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), d["x0"])
    prob.profile_enable(True)
    for _ in range(12):
        prob.panoc_step()
    p = prob.profile2()
    prob.close()
    assert p["k_twoloop_persist"]["launches"] == 0 and p["k_axpy_dot"]["launches"] == 0
    assert p["k_stencil_update"]["form"].startswith("k_stencil_update_c<") and p["k_stencil_update"]["launches"] >= 11
    assert p["x_d"]["form"].startswith("k_compact_xd<")


def test_stencil_alps_small_obstacle(bz, ref):
    """Whole ALPS on a small obstacle problem: same counts and solution as the oracle; the solution
    satisfies the obstacle KKT system  A x - b + y = 0, x >= psi, y <= 0, y (x - psi) = 0."""
    nx, ny = 24, 32
    d, n, dev, orc = make_cfg3(bz, ref, nx, ny, load=-1.0)     # load pushes onto the obstacle
    sub = lambda **kw: bz.PANOCplus(maxit=100000, minimum_gamma=float(np.finfo(float).eps), **kw)
    subr = lambda **kw: ref.PANOCplus(maxit=100000, minimum_gamma=float(np.finfo(float).eps), **kw)
    a = bz.alps(*dev, d["x0"], np.zeros(n), subsolver=sub, resident=True)
    o = ref.alps(*orc, d["x0"], np.zeros(n), subsolver=subr)
    assert a[5] == o[5] == "first_order"
    assert a[2] == o[2]
    assert abs(a[3] - o[3]) <= max(3, 0.02 * o[3])       # inner counts may differ by a late branch flip
    assert np.max(np.abs(a[0] - o[0])) <= 1e-6
    x, y = a[0], a[1]
    g = np.empty(n)
    orc[0].gradient(g, x)
    assert np.max(np.abs(g + y)) <= 1e-4
    assert np.min(x - d["psi"]) >= -1e-5 and np.max(y) <= 1e-6
    assert np.max(np.abs(y * (x - d["psi"]))) <= 1e-5
    assert np.sum(x - d["psi"] <= 1e-6) > 10                 # the obstacle is really active


# ------------------------------------------------------------------ cfg 4: basis pursuit, dense A
def make_cfg4(bz, ref, ny, n, dtype, density=0.05):
    d = bz.synth.basis_pursuit(ny, n, dtype=dtype, density=density)
    dev = (bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(d["A"], d["b"]), bz.ZeroSet())
    orc = (ref.Zero(), ref.NormL1(1.0), ref.DenseAffine(d["A"], d["b"]), ref.ZeroSet())
    return d, dev, orc


@pytest.mark.parametrize("shape", [(20, 100), (3, 8), (64, 512), (257, 1028)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_al_gradient(bz, ref, shape, dtype):
    """K9+K1: gradient!(dlx, al, x) with c(x) = A x - b (demo/basispursuit.jl:38-49), D = ZeroSet.
    GEMV sums are order-dependent: tolerance 1e-12 (fp64) / 2e-5 (fp32) relative to ||grad||_inf."""
    ny, n = shape
    d, dev, orc = make_cfg4(bz, ref, ny, n, dtype)
    rng = np.random.default_rng(ny + n)
    x = rng.standard_normal(n).astype(dtype)
    mu = (10.0 ** rng.uniform(-2, 0, ny)).astype(dtype)
    y = rng.standard_normal(ny).astype(dtype)
    prob = bz.Problem(*dev, n, ny, dtype)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x)
    g_ref = np.empty(n, dtype)
    lx = float(al.gradient(g_ref, x))
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert np.max(np.abs(g_dev.astype(np.float64) - g_ref)) <= tol * np.max(np.abs(g_ref))
    assert abs(vals[0] - lx) <= tol * max(1.0, abs(lx))
    prob.close()


def test_dense_panoc_and_alps_fp64(bz, ref):
    """The reference's own basis-pursuit demo shape (demo/basispursuit.jl:55-66: 20x100) in fp64:
    iterates follow the oracle, ALPS returns a feasible l1 solution."""
    ny, n = 20, 100
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float64, density=0.1)
    mu, y = np.full(ny, 0.1), np.zeros(ny)
    prob = bz.Problem(*dev, n, ny, np.float64)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=2.3e-16).c_opts(), np.zeros(n))
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), np.zeros(n))
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), np.zeros(n), minimum_gamma=2.3e-16)
    st = it.init()
    for k in range(15):
        assert rel(prob.panoc_vector("z"), st.z) <= 1e-9 if np.any(st.z) else True
        prob.panoc_step()
        st = it.step(st)
    prob.close()
    sub = lambda **kw: bz.PANOCplus(maxit=100000, minimum_gamma=2.3e-16, **kw)
    subr = lambda **kw: ref.PANOCplus(maxit=100000, minimum_gamma=2.3e-16, **kw)
    for resident in (True, False):
        a = bz.alps(*dev, np.zeros(n), np.zeros(ny), subsolver=sub, subsolver_maxit=100000, resident=resident)
        o = ref.alps(*orc, np.zeros(n), np.zeros(ny), subsolver=subr, subsolver_maxit=100000)
        assert a[5] == o[5] == "first_order"
        assert np.max(np.abs(d["A"] @ a[0] - d["b"])) <= 1e-5
        assert abs(np.sum(np.abs(a[0])) - np.sum(np.abs(o[0]))) <= 1e-4 * np.sum(np.abs(o[0]))
        assert np.max(np.abs(a[0] - o[0])) <= 1e-4


def test_dense_basis_pursuit_recovers_sparse_signal_fp32(bz, ref):
    """fp32 as in BASELINE config 4 (scaled down): basis pursuit recovers the planted sparse +-1 signal."""
    ny, n = 256, 1024
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float32, density=0.02)
    sub = lambda **kw: bz.PANOCplus(maxit=20000, minimum_gamma=float(np.finfo(np.float32).eps), **kw)
    a = bz.alps(*dev, np.zeros(n, np.float32), np.zeros(ny, np.float32), tol=np.float32(1e-4), subsolver=sub,
                subsolver_maxit=20000)
    assert a[0].dtype == np.float32
    assert a[5] == "first_order"
    assert np.max(np.abs(a[0] - d["xtrue"])) <= 5e-3


def test_dense_full_size_gradient_fp32(bz, ref):
    """BASELINE config 4 size (A 8192 x 65536 fp32 = 2 GiB): the device AL gradient (two GEMVs over A)
    against numpy's sgemv on the same inputs."""
    ny, n = 8192, 65536
    d = bz.synth.basis_pursuit(ny, n, dtype=np.float32)
    prob = bz.Problem(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(d["A"], d["b"]), bz.ZeroSet(), n, ny, np.float32)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(n) * 0.1).astype(np.float32)
    mu = np.full(ny, 0.1, np.float32)
    y = rng.standard_normal(ny).astype(np.float32)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    t = (d["A"].astype(np.float64) @ x.astype(np.float64) - d["b"] + mu.astype(np.float64) * y) / mu
    g_ref = d["A"].T.astype(np.float64) @ t if False else (t @ d["A"].astype(np.float64, copy=False))
    assert np.max(np.abs(g_dev - g_ref)) <= 2e-4 * np.max(np.abs(g_ref))
    prob.close()


def test_persistent_two_loop_matches_kernel_chain(bz, ref):
    """K4 in its persistent form (one launch, d register-resident, grid-barrier phases) against the
    one-kernel-per-step chain: same arithmetic, different fixed summation tree -> agreement to
    rounding, and against the oracle within the north-star tolerance.  n is chosen so that several
    register packs per thread AND the ragged tail are exercised."""
    n = 3_000_017
    d, dev, orc = make_cfg2(bz, ref, n)
    rng = np.random.default_rng(4)
    mu = np.full(n, 0.1)
    y = rng.standard_normal(n)
    out = []
    for persist in (True, False):
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, persist=persist, directions=bz.LBFGS(5, compact=False)).c_opts(), np.zeros(n))
        for _ in range(14):
            prob.panoc_step()
        out.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars(), prob.profile()))
        prob.close()
    (x1, z1, s1, _), (x2, z2, s2, _) = out
    assert rel(x1, x2) <= 1e-12 and rel(z1, z2) <= 1e-12
    assert abs(s1["stop_norm"] - s2["stop_norm"]) <= 1e-9 * max(1.0, s2["stop_norm"])
    assert s1["gamma"] == s2["gamma"] and s1["lbfgs_mem"] == s2["lbfgs_mem"] == 5.0
    # oracle after the same 15 states
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), np.zeros(n))
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), np.zeros(n))
    st = it.init()
    for _ in range(14):
        st = it.step(st)
    assert rel(x1, st.x) <= RTOL_ITER and rel(z1, st.z) <= RTOL_ITER


def test_persistent_kernel_is_used_at_benchmark_size(bz, ref):
    n = 10_000_000
    d = bz.synth.l1_quadratic(n)
    prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                      bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n, n, np.float64)
    prob.set_multipliers(np.full(n, 0.1), np.zeros(n))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, directions=bz.LBFGS(5, compact=False)).c_opts(), np.zeros(n))
    prob.profile_enable(True)
    for _ in range(8):
        prob.panoc_step()
    p = prob.profile()
    assert p["k_twoloop_persist"]["launches"] >= 6 and p["k_fused_sep"]["launches"] == 8
    prob.close()


# ------------------------------------------------------------------ solver semantics / edge cases
@pytest.mark.parametrize("tol,maxit", [(1e-2, 1000), (1e-6, 1000), (1e-12, 7), (1e30, 1000), (1e-8, 1)])
def test_subsolver_return_contract(bz, ref, tol, maxit):
    """`sol, it = PANOCplus(tol=..., maxit=...)(f=alFun, g=gFun, x0=x)` (alps.jl:64-66): same z, same
    iteration count (initial state counts as 1; stop when k >= maxit or the stopping norm <= tol), and
    the side channels alFun.fx / gFun.gz (alps.jl:68) carry f and g at the last evaluation point."""
    n = 5000
    d, dev, orc = make_cfg2(bz, ref, n)
    rng = np.random.default_rng(9)
    mu, y, x0 = np.full(n, 0.1), rng.standard_normal(n), rng.standard_normal(n) * 0.01
    alD = bz.AugLagFun(dev[0], dev[2], dev[3], mu.copy(), y.copy(), x0)
    gD = bz.NonsmoothCostFun(dev[1])
    z_d, it_d = bz.PANOCplus(tol=tol, maxit=maxit)(f=alD, g=gD, x0=x0)
    alR = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    gR = ref.NonsmoothCostFun(orc[1])
    z_r, it_r = ref.PANOCplus(tol=tol, maxit=maxit)(f=alR, g=gR, x0=x0)
    assert it_d == it_r
    assert rel(z_d, z_r) <= 1e-9
    assert abs(float(alD.fx) - float(alR.fx)) <= 1e-9 * max(1.0, abs(float(alR.fx)))
    assert abs(float(gD.gz) - float(gR.gz)) <= 1e-9 * max(1.0, abs(float(gR.gz)))
    assert gD.gamma == pytest.approx(float(gR.gamma), rel=1e-12)


def test_vector_bounds(bz, ref):
    """IndBox with per-coordinate bounds as g and as D = ClosedSet(IndBox(lo, hi)) (indicatorSet.jl:8-11)."""
    n = 20001
    rng = np.random.default_rng(12)
    d = bz.synth.l1_quadratic(n)
    lo, hi = -rng.uniform(0.1, 1.0, n), rng.uniform(0.1, 1.0, n)
    glo, ghi = -rng.uniform(0.5, 2.0, n), rng.uniform(0.5, 2.0, n)
    dev = (bz.DiagQuadratic(d["q"], d["b"]), bz.IndBox(glo, ghi), bz.IdentityFunction(), bz.ClosedSet(bz.IndBox(lo, hi)))
    orc = (ref.DiagQuadratic(d["q"], d["b"]), ref.IndBox(glo, ghi), ref.IdentityFunction(), ref.ClosedSet(ref.IndBox(lo, hi)))
    x, mu, y = rng.standard_normal(n), rng.uniform(0.01, 1, n), rng.standard_normal(n)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x)
    g_ref = np.empty(n)
    al.gradient(g_ref, x)
    assert np.array_equal(g_dev, g_ref)
    z_dev, _ = prob.eval_prox(x * 2, 0.5)
    z_ref = np.empty(n)
    orc[1].prox(z_ref, x * 2, 0.5)
    assert np.array_equal(z_dev, z_ref)
    prob.close()
    a = bz.alps(*dev, np.zeros(n), np.zeros(n))
    o = ref.alps(*orc, np.zeros(n), np.zeros(n))
    assert a[5] == o[5] == "first_order" and a[2] == o[2] and a[3] == o[3]
    assert rel(a[0], o[0]) <= 1e-9


def test_float32_solver_follows_float32_oracle(bz, ref):
    """T = Float32 end to end: same host scalar arithmetic in Float32 as the oracle; iterates agree to a
    few float32 ulps for the first states, alps converges to the same point."""
    n = 30000
    d, dev, orc = make_cfg2(bz, ref, n, dtype=np.float32)
    mu, y, x0 = np.full(n, 0.1, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, x0, 12, dtype=np.float32)
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert abs(g_d - g_r) <= 1e-5 * g_r
        assert ex <= max(2e-5, 100 * sens) and ez <= max(2e-5, 100 * sens), (k, ex, ez, sens)
    prob.close()
    a = bz.alps(*dev, x0, y, tol=np.float32(1e-4))
    o = ref.alps(*orc, x0, y, tol=np.float32(1e-4))
    assert a[0].dtype == np.float32 and a[5] == o[5] == "first_order"
    assert np.max(np.abs(a[0] - o[0])) <= 1e-3


def test_custom_dual_safeguard_uses_host_loop(bz, ref):
    """alps(...; dual_safeguard=f) (alps.jl:23,62): a user callback keeps the outer loop on the host
    and enters the device at the subsolver seam."""
    n = 2000
    d, dev, orc = make_cfg2(bz, ref, n)
    calls = []

    def guard(y, cx):
        calls.append(1)
        np.clip(y, -5.0, 5.0, out=y)

    a = bz.alps(*dev, np.zeros(n), np.zeros(n), dual_safeguard=guard)
    o = ref.alps(*orc, np.zeros(n), np.zeros(n), dual_safeguard=guard)
    assert len(calls) == a[2] + o[2] and a[5] == o[5]
    assert rel(a[0], o[0]) <= 1e-8


def test_headline_size_iterates_match_oracle(bz, ref):
    """BASELINE config 2 at FULL size (n = 10^7): the first PANOCplus states of the exact benchmark
    problem (persistent two-loop + fused kernel) against the numpy oracle, within 1e-10 relative."""
    n = 10_000_000
    d, dev, orc = make_cfg2(bz, ref, n)
    prob, st, rows = run_traces(bz, ref, dev, orc, n, np.full(n, 0.1), np.zeros(n), np.zeros(n), 7,
                                minimum_gamma=float(np.finfo(float).eps), compact=False)
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert abs(g_d - g_r) <= 1e-13 * g_r
        assert ex <= RTOL_ITER and ez <= RTOL_ITER, f"iterate mismatch at k={k}: {ex} {ez}"
    assert sum(r[7] for r in rows) >= 5
    p = prob.profile()
    prob.close()


@pytest.mark.parametrize("g", ["lpnonneg", "lpbox"])
def test_lp_power_prox(bz, ref, g):
    """src/proxoperators/normLpNonneg.jl / normLpBox.jl: scalar Newton solve per element.  pow() is not
    correctly rounded on either side, so z agrees to 1e-10 instead of bit for bit; the zero pattern
    (the discrete decisions of the global-minimum tests) must agree except within rounding of a tie."""
    n = 40001
    d, dev, orc = make_cfg2(bz, ref, n, g=g)
    rng = np.random.default_rng(21)
    x = rng.standard_normal(n) * 2
    prob = bz.Problem(*dev, n, n, np.float64)
    for gamma in (0.37, 1.9):
        z_dev, gz_dev = prob.eval_prox(x, gamma)
        z_ref = np.empty(n)
        gz_ref = orc[1].prox(z_ref, x, gamma)
        same_support = (z_dev != 0) == (z_ref != 0)
        assert np.mean(same_support) >= 0.9999
        m = same_support
        assert np.max(np.abs(z_dev[m] - z_ref[m])) <= 1e-10
        assert abs(gz_dev - gz_ref) <= 1e-6 * max(1.0, abs(gz_ref))
        assert np.all(z_dev >= 0)
    prob.close()
    # a full (nonconvex) ALPS solve ends at a stationary point of the same quality as the oracle's
    n = 3000
    d, dev, orc = make_cfg2(bz, ref, n, g=g)
    a = bz.alps(*dev, np.zeros(n), np.zeros(n))
    o = ref.alps(*orc, np.zeros(n), np.zeros(n))
    assert a[5] == o[5]
    obj = lambda x: np.sum(x * (0.5 * d["q"] * x - d["b"])) + 0.8 * np.sum(np.maximum(x, 0) ** 0.5)
    assert abs(obj(a[0]) - obj(o[0])) <= 1e-6 * max(1.0, abs(obj(o[0])))


def test_no_acceleration_direction(bz, ref):
    """directions = NoAcceleration() (demo/rosenbrock.jl:96-97): d = -res; iterates follow the oracle."""
    n = 20000
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, directions=bz.NoAcceleration()).c_opts(), x0)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, directions=ref.NoAcceleration())
    st = it.init()
    for k in range(20):
        assert rel(prob.panoc_vector("z"), st.z) <= RTOL_ITER
        prob.panoc_step()
        st = it.step(st)
    assert prob.panoc_scalars()["lbfgs_mem"] == 0.0
    prob.close()


@pytest.mark.parametrize("compact", [False, True])
def test_stencil_fast_path_equals_generic_bitwise(bz, ref, compact, monkeypatch):
    """cfg 3: the two fused stencil passes ({gradL(x_d) + FB step}, {gradL(z) + pair + stop norm [+ the compact
    form's Gram products and next p, w: k_stencil_update_c]}) are the same arithmetic and the same summation order
    as the generic kernels they replace, in both forms of the L-BFGS operator — with grad L(x_d) and res re-formed in the
    second pass (r03, the default with the compact form: 36 passes over n instead of 39) and with both read from memory."""
    nx, ny = 96, 128
    d, n, dev, orc = make_cfg3(bz, ref, nx, ny, load=-1.0)
    out = []
    for fuse, regx in ((True, "1"), (False, "1"), (True, "0"), (True, "2")):
        monkeypatch.setenv("BZ_STENCIL_REGX", regx)
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(np.full(n, 0.1), np.zeros(n))
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, fuse=fuse, minimum_gamma=2.3e-16,
                                      directions=bz.LBFGS(5, compact=compact)).c_opts(), d["x0"])
        for _ in range(30):
            prob.panoc_step()
        out.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_vector("res"), prob.panoc_scalars()))
        prob.close()
    (x1, z1, r1, s1) = out[0]
    for (x2, z2, r2, s2) in out[1:]:
        assert np.array_equal(x1, x2) and np.array_equal(z1, z2) and np.array_equal(r1, r2)
        for key in ("gamma", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_H", "al_z"):
            assert s1[key] == s2[key], key


@pytest.mark.parametrize("seed,form", [(s, "two-loop") for s in range(12)] + [(s, "compact") for s in range(12)])
def test_randomised_kinds_alps_parity(bz, ref, seed, form):
    """Seeded sweep over the lowered oracle kinds and ragged sizes: the device ALPS and the oracle ALPS take
    the same outer/inner iteration counts (up to a late branch flip) and return the same point.  Both
    evaluations of the L-BFGS operator, each against the oracle restated in the same form: the sweep walks
    through tau backtracks, gamma halvings (memory reset), skipped pairs and growing memory, i.e. every way
    the compact path can lose and regain the p, w it carries from one fused pass to the next."""
    rng = np.random.default_rng(1000 + seed)
    compact = form == "compact"
    n = int(rng.integers(1, 6000))
    q = rng.uniform(0.2, 5.0, n)
    b = rng.standard_normal(n) * 4
    fk = rng.choice(["diag", "zero"], p=[0.8, 0.2])
    gk = rng.choice(["l1", "nonneg", "l1box", "l0box", "indbox", "zero"])
    Dk = rng.choice(["box", "free", "zero"]) if fk == "diag" else "box"
    lam = float(rng.uniform(0.1, 3.0))
    u = rng.uniform(0.0, 1.5, n)
    f_d, f_r = (bz.DiagQuadratic(q, b), ref.DiagQuadratic(q, b)) if fk == "diag" else (bz.Zero(), ref.Zero())
    g_d, g_r = {"l1": (bz.NormL1(lam), ref.NormL1(lam)),
                "nonneg": (bz.NormL1Nonneg(lam), ref.NormL1Nonneg(lam)),
                "l1box": (bz.NormL1Box(lam, u=u), ref.NormL1Box(lam, u=u)),
                "l0box": (bz.NormL0Box(lam, u=u), ref.NormL0Box(lam, u=u)),
                "indbox": (bz.IndBox(-0.7, 0.9), ref.IndBox(-0.7, 0.9)),
                "zero": (bz.Zero(), ref.Zero())}[gk]
    lo, hi = -float(rng.uniform(0.2, 1.0)), float(rng.uniform(0.2, 1.0))
    D_d, D_r = {"box": (bz.ClosedSet(bz.IndBox(lo, hi)), ref.ClosedSet(ref.IndBox(lo, hi))),
                "free": (bz.FreeSet(), ref.FreeSet()), "zero": (bz.ZeroSet(), ref.ZeroSet())}[Dk]
    x0, y0 = rng.standard_normal(n) * 0.1, rng.standard_normal(n) * 0.1
    import warnings
    sub_r = lambda **kw: ref.PANOCplus(directions=ref.LBFGS(5, compact=compact), **kw)
    its_o, its_o2, its_a = [], [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o = ref.alps(f_r, g_r, ref.IdentityFunction(), D_r, x0, y0, maxit=40, subsolver=sub_r,
                     outer_trace=lambda k, x, y, mu, sub_it, *r: its_o.append(int(sub_it)))
        # the resolution of the comparison: the oracle against ITSELF with another rounding of its reductions
        # (SURVEY §7 H3).  A solve that stops at tol = 1e-6 after some hundred inner iterations moves by
        # 1e-6 .. 5e-6 in x and by a few per cent in the inner count under that perturbation alone.
        ref.set_reducer(LongDoubleReducer())
        try:
            o2 = ref.alps(f_r, g_r, ref.IdentityFunction(), D_r, x0, y0, maxit=40, subsolver=sub_r,
                          outer_trace=lambda k, x, y, mu, sub_it, *r: its_o2.append(int(sub_it)))
        finally:
            ref.set_reducer(None)
    a = bz.alps(f_d, g_d, bz.IdentityFunction(), D_d, x0, y0, maxit=40,
                subsolver=lambda **kw: bz.PANOCplus(directions=bz.LBFGS(5, compact=compact), **kw), resident=True)
    tag = f"n={n} f={fk} g={gk} D={Dk} {form}"
    # report the FIRST divergence rather than hide it.  Per-outer-iteration inner counts of the device
    # (host outer loop around the device subsolver: the same subsolves) against the oracle's, and of the oracle's
    # perturbed twin against the oracle: the index of the first outer iteration whose subsolve length differs.
    def dev_sub(**kw):
        inner = bz.PANOCplus(directions=bz.LBFGS(5, compact=compact), **kw)

        def run(*, f, g, x0):
            sol, it = inner(f=f, g=g, x0=x0)
            its_a.append(int(it))
            return sol, it
        return run
    ah = bz.alps(f_d, g_d, bz.IdentityFunction(), D_d, x0, y0, maxit=40, subsolver=dev_sub, resident=False)
    assert ah[2] == a[2] and ah[3] == a[3], tag                      # resident and host outer loops: the same solve

    def first_div(p, q):
        for i, (u, v) in enumerate(zip(p, q)):
            if u != v:
                return i + 1
        return None if len(p) == len(q) else min(len(p), len(q)) + 1
    fd_dev, fd_twin = first_div(its_a, its_o), first_div(its_o2, its_o)
    line = (f"first divergence {tag}: device at outer {fd_dev} of {len(its_o)} (inner {its_a} vs oracle {its_o}), "
            f"oracle twin at outer {fd_twin}")
    print(line)
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/first_divergence.log", "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass
    # what that log justifies: the first subsolve (same x0, mu, y; it stops long before rounding noise reaches the
    # stop norm) has the SAME length on the device and in the oracle, and so has every subsolve before the first
    # divergence; the +-30 % window on the total is needed only for the solves that do diverge — those that do not
    # must agree exactly
    if fd_twin != 1:
        assert fd_dev != 1, line
    if fd_dev is None:
        assert a[3] == o[3], line
    assert a[5] == o[5], tag
    assert a[2] == o[2] or o2[2] != o[2], tag
    # The inner count adds up the lengths of six-odd subsolves, each stopped where a noisy, non-monotone
    # stop-norm sequence first dips under the inner tolerance: over 576 seeded cases (tests/stress/stress_sweep.py)
    # device and oracle differ by up to 26 % there (the oracle's perturbed twin by up to 19 %) while agreeing on x
    # to the oracle's own resolution.
    assert abs(a[3] - o[3]) <= max(3, 0.3 * o[3]), tag
    scale = max(1.0, float(np.max(np.abs(o[0]))))
    self_x = float(np.max(np.abs(o2[0] - o[0])))
    # Box-constrained cases stop at tol = 1e-6 with active constraints and multipliers in play: over ~1000
    # seeded cases the oracle moves by up to 6e-6 against its own perturbed twin and the device by up to
    # 5.8e-6 against the oracle — one sample of the former is not a bound for the latter, so the envelope
    # is the documented one (2e-5, three times the worst seen); unconstrained-D cases keep 1e-6.
    tol = max((2e-5 if Dk == "box" else 1e-6) * scale, 4.0 * self_x)
    if gk == "l0box":       # the L0 prox is discontinuous: a tie may flip an entry (the oracle pair shows it too)
        frac_self = float(np.mean(np.abs(o2[0] - o[0]) <= 1e-4 * scale))
        assert np.mean(np.abs(a[0] - o[0]) <= max(tol, 1e-4 * scale)) >= min(0.999, frac_self - 0.002), tag
    else:
        assert np.max(np.abs(a[0] - o[0])) <= tol, tag


# ------------------------------------------------------------------ compact L-BFGS (alternate evaluation of the same operator)
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("n", [1000, 300007])
def test_compact_lbfgs_matches_compact_oracle(bz, ref, n, fuse):
    """directions = LBFGS(5, compact=True): k_gram_dots + k_fused_compact (or k_compact_xd) against the
    oracle's LBFGSCompactOperator — the same operation order, so iterates agree within the north-star
    tolerance — and against the two-loop form, which it equals up to rounding."""
    d, dev, orc = make_cfg2(bz, ref, n)
    rng = np.random.default_rng(5)
    mu, y, x0 = np.full(n, 0.1), rng.standard_normal(n), np.zeros(n)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, fuse=fuse, directions=bz.LBFGS(5, compact=True)).c_opts(), x0)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, directions=ref.LBFGS(5, compact=True))
    al2 = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    it2 = ref.PANOCplusIteration(al2, ref.NonsmoothCostFun(orc[1]), x0)
    st, st2 = it.init(), it2.init()
    for k in range(30):
        xd, zd = prob.panoc_vector("x"), prob.panoc_vector("z")
        assert rel(xd, st.x) <= RTOL_ITER and rel(zd, st.z) <= RTOL_ITER, f"compact oracle mismatch at k={k + 1}"
        assert rel(zd, st2.z) <= RTOL_ITER, f"two-loop form mismatch at k={k + 1}"
        sc = prob.panoc_scalars()
        assert sc["lbfgs_mem"] == st.H.currmem
        prob.panoc_step()
        st, st2 = it.step(st), it2.step(st2)
    prob.close()


def test_compact_lbfgs_alps(bz, ref):
    n = 4000
    d, dev, orc = make_cfg2(bz, ref, n)
    x0, y0 = np.zeros(n), np.zeros(n)
    a = bz.alps(*dev, x0, y0, subsolver=lambda **kw: bz.PANOCplus(directions=bz.LBFGS(5, compact=True), **kw), resident=True)
    o = ref.alps(*orc, x0, y0)
    assert a[5] == o[5] == "first_order" and a[2] == o[2]
    assert abs(a[3] - o[3]) <= max(3, 0.05 * o[3])
    assert rel(a[0], o[0]) <= 1e-8


def make_pairs(bz, ref, n, kind, g="zero"):
    d = bz.synth.l1_quadratic(n)
    g_d, g_r = (bz.NormL1(0.3), ref.NormL1(0.3)) if g == "l1" else (bz.Zero(), ref.Zero())
    # shift b so that the unconstrained minimiser b/q has both signs in both slots of a pair
    return ((bz.DiagQuadratic(d["q"], 0.2 * d["b"]), g_d, bz.IdentityFunction(), bz.PairwiseSet(kind)),
            (ref.DiagQuadratic(d["q"], 0.2 * d["b"]), g_r, ref.IdentityFunction(), ref.PairwiseSet(kind)))


@pytest.mark.parametrize("kind", ["vc", "cc", "eitheror", "xor"])
@pytest.mark.parametrize("n", [2, 6, 1000, 262146])
def test_pairwise_sets_al_gradient_bit_exact(bz, ref, n, kind):
    """SURVEY f-2: D built from the 2-element projections over adjacent pairs (vanishingConstraints.jl:27-46,
    complementarityConstraints.jl:8-20, orConstraints.jl:7-36).  gradient!(dlx, al, x) element-wise
    bit-exact against the oracle, including ties, zeros of either sign and points on the sets' boundaries."""
    dev, orc = make_pairs(bz, ref, n, kind)
    rng = np.random.default_rng(n + len(kind))
    x = rng.standard_normal(n)
    mu = 10.0 ** rng.uniform(-2, 1, n)
    y = rng.standard_normal(n)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5])
    k = min(n, 240)
    x[:k] = special[rng.integers(0, 6, k)]
    y[:k] = 0.0                                   # c(x) + mu*y = x exactly: the special values reach the projection
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x)
    g_ref = np.empty(n)
    lx = al.gradient(g_ref, x)
    assert np.array_equal(g_dev, g_ref)
    assert abs(vals[0] - lx) <= 1e-13 * max(1.0, abs(lx))
    prob.close()


@pytest.mark.parametrize("kind", ["vc", "cc", "eitheror", "xor"])
def test_pairwise_sets_every_special_pair_bit_exact(bz, ref, kind):
    """The device evaluates the pairwise projections as one keep-or-zero predicate per component (`proj_pair`,
    bz_kernels.h) where the reference walks a ladder of ifs (vanishingConstraints.jl:27-46,
    complementarityConstraints.jl:8-20, orConstraints.jl:7-36).  EVERY ordered pair of the values where the ladders
    branch — zeros of either sign, equal and opposite magnitudes (the ties), numbers whose product underflows (XOR tests
    x1*x2 > 0), a denormal, infinities, NaN — in both slots, through `gradient!` with mu = 1, y = 0, f = Zero (so
    dlx = x - proj_D(x): the keep-or-zero decision itself, sign of the zero included): the bits of the oracle."""
    sp = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 1e-200, -1e-200, 5e-324, -5e-324, 1e300, -1e300,
                   np.inf, -np.inf, np.nan])
    a, b = np.meshgrid(sp, sp, indexing="ij")
    x = np.stack([a.ravel(), b.ravel()], axis=1).ravel().copy()          # (x1, x2) adjacent: every ordered pair once
    n = x.size
    dev = (bz.Zero(), bz.Zero(), bz.IdentityFunction(), bz.PairwiseSet(kind))
    orc = (ref.Zero(), ref.Zero(), ref.IdentityFunction(), ref.PairwiseSet(kind))
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(np.ones(n), np.zeros(n))
    g_dev, _ = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], np.ones(n), np.zeros(n), x)
    g_ref = np.empty(n)
    with np.errstate(all="ignore"):
        al.gradient(g_ref, x)
    nan_d, nan_r = np.isnan(g_dev), np.isnan(g_ref)
    assert np.array_equal(nan_d, nan_r), np.flatnonzero(nan_d != nan_r)[:10]
    bad = np.flatnonzero((g_dev.view(np.uint64) != g_ref.view(np.uint64)) & ~nan_d)
    assert bad.size == 0, [(x[i - i % 2], x[i - i % 2 + 1], i % 2, g_dev[i], g_ref[i]) for i in bad[:8]]
    prob.close()


@pytest.mark.parametrize("kind,g", [("vc", "zero"), ("cc", "l1"), ("eitheror", "zero"), ("xor", "l1")])
def test_pairwise_sets_alps(bz, ref, kind, g):
    """Whole ALPS solves with a pairwise (nonconvex) D: same outer/inner counts and solution as the oracle,
    and the returned c(x) = x lies in the set to tol_prim."""
    n = 2000
    dev, orc = make_pairs(bz, ref, n, kind, g)
    x0, y0 = np.zeros(n), np.zeros(n)
    o = ref.alps(*orc, x0, y0)
    a = bz.alps(*dev, x0, y0)
    assert a[5] == o[5]
    assert a[2] == o[2] and abs(a[3] - o[3]) <= max(2, 0.02 * o[3])
    assert rel(a[0], o[0]) <= 1e-8
    z = np.empty(n)
    orc[3].proj(z, a[0])
    assert np.max(np.abs(z - a[0])) <= 1e-5
    with pytest.raises(Exception):
        bz.Problem(dev[0], dev[1], dev[2], bz.PairwiseSet(kind), n - 1, n - 1, np.float64)      # odd ny


@pytest.mark.parametrize("kind", ["cc", "vc", "eitheror", "xor"])
def test_pairwise_sets_split_layout(bz, ref, kind):
    """SURVEY f-2, demo/obstacle.jl:151-168 (SetObstacleRed): the pairs are (c(x)[i], c(x)[i + N]).  The host binding
    maps the split layout onto the kernels' adjacent pairs by the interleaving permutation at the boundary: the AL
    gradient is element-wise bit-exact against the oracle's split-layout projection, iterates and whole ALPS solves
    follow it."""
    n = 3000
    d = bz.synth.l1_quadratic(n)
    dev = (bz.DiagQuadratic(d["q"], 0.2 * d["b"]), bz.NormL1(0.3), bz.IdentityFunction(), bz.PairwiseSet(kind, layout="split"))
    orc = (ref.DiagQuadratic(d["q"], 0.2 * d["b"]), ref.NormL1(0.3), ref.IdentityFunction(), ref.PairwiseSet(kind, layout="split"))
    rng = np.random.default_rng(5)
    x, mu, y = rng.standard_normal(n), 10.0 ** rng.uniform(-2, 1, n), rng.standard_normal(n)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    g_dev, vals = prob.eval_al_gradient(x)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x)
    g_ref = np.empty(n)
    lx = al.gradient(g_ref, x)
    assert np.array_equal(g_dev, g_ref) and abs(vals[0] - lx) <= 1e-13 * max(1.0, abs(lx))
    prob.close()
    prob, st, rows = run_traces(bz, ref, dev, orc, n, np.full(n, 0.1), 0.3 * rng.standard_normal(n), np.zeros(n), 25)
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert ex <= iter_tol(sens) and ez <= iter_tol(sens), (k, ex, ez, sens)
    assert sum(r[7] for r in rows) >= 15          # ... through the fused one-pass kernels
    prob.close()
    a = bz.alps(*dev, np.zeros(n), np.zeros(n))
    o = ref.alps(*orc, np.zeros(n), np.zeros(n))
    assert a[5] == o[5] and a[2] == o[2] and abs(a[3] - o[3]) <= max(2, 0.05 * o[3])
    assert rel(a[0], o[0]) <= 1e-8 and rel(a[8], o[8]) <= 1e-8          # x and s = proj_D(c(x) + mu y) in the caller's layout
    z = np.empty(n)
    orc[3].proj(z, a[0])
    assert np.max(np.abs(z - a[0])) <= 1e-5


@pytest.mark.parametrize("M", [1, 3, 5])
def test_compact_lbfgs_float32_and_short_memory(bz, ref, M):
    """The compact form with T = Float32 (coefficients are formed in double on the host and applied in
    Float32) and with memories shorter than its capacity: ALPS converges to the oracle's point."""
    n = 20000
    d, dev, orc = make_cfg2(bz, ref, n, dtype=np.float32)
    x0, y0 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    a = bz.alps(*dev, x0, y0, tol=np.float32(1e-4),
                subsolver=lambda **kw: bz.PANOCplus(directions=bz.LBFGS(M, compact=True), **kw), resident=True)
    o = ref.alps(*orc, x0, y0, tol=np.float32(1e-4),
                 subsolver=lambda **kw: ref.PANOCplus(directions=ref.LBFGS(M), **kw))
    assert a[0].dtype == np.float32 and a[5] == o[5] == "first_order"
    assert np.max(np.abs(a[0] - o[0])) <= 2e-3
    assert abs(a[3] - o[3]) <= max(5, 0.2 * o[3])
    d64, dev64, orc64 = make_cfg2(bz, ref, n)
    x0, y0 = np.zeros(n), np.zeros(n)
    a = bz.alps(*dev64, x0, y0, subsolver=lambda **kw: bz.PANOCplus(directions=bz.LBFGS(M, compact=True), **kw), resident=True)
    o = ref.alps(*orc64, x0, y0, subsolver=lambda **kw: ref.PANOCplus(directions=ref.LBFGS(M, compact=True), **kw))
    assert a[5] == o[5] == "first_order" and a[2] == o[2] and abs(a[3] - o[3]) <= max(3, 0.05 * o[3])
    assert rel(a[0], o[0]) <= 1e-8


# ------------------------------------------------------------------ the device against the committed fixtures
def test_device_matches_golden_fixtures(bz, ref):
    """tests/golden/*.json (restatement-generated, committed with the script that made them): the HIP path
    reproduces the recorded PANOCplus states — both evaluations of the L-BFGS operator — and the recorded
    ALPS solves (cfg 2 and the four pairwise sets) without the oracle in the loop."""
    import json
    import os
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, compact in (("panoc_trace_cfg2_n64.json", False), ("panoc_trace_cfg2_n64_compact.json", True)):
        with open(os.path.join(gold_dir, name)) as fh:
            gold = json.load(fh)
        n = gold["n"]
        d, dev, _ = make_cfg2(bz, ref, n)
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(np.full(n, 0.1), np.sin(np.arange(n, dtype=np.float64)))
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=float(np.finfo(float).eps),
                                      directions=bz.LBFGS(5, compact=compact)).c_opts(), np.zeros(n))
        for row in gold["rows"]:
            sc = prob.panoc_scalars()
            assert int(sc["k"]) == row["k"]
            assert abs(sc["gamma"] - row["gamma"]) <= 1e-13 * row["gamma"]
            assert rel(prob.panoc_vector("x"), np.array(row["x"])) <= RTOL_ITER, (name, row["k"])
            assert rel(prob.panoc_vector("z"), np.array(row["z"])) <= RTOL_ITER, (name, row["k"])
            assert abs(sc["stop_norm"] - row["stop_norm"]) <= 1e-9 * max(1.0, row["stop_norm"])
            prob.panoc_step()
        prob.close()
    with open(os.path.join(gold_dir, "alps_cfg2_n256.json")) as fh:
        gold = json.load(fh)
    n = gold["n"]
    d, dev, _ = make_cfg2(bz, ref, n)
    a = bz.alps(*dev, np.zeros(n), np.zeros(n))
    assert a[5] == gold["status"] and a[2] == gold["tot_it"] and a[3] == gold["tot_inner_it"]
    assert rel(a[0], np.array(gold["x"])) <= 1e-9 and rel(a[9], np.array(gold["mu"])) <= 1e-12
    with open(os.path.join(gold_dir, "alps_pairs_n128.json")) as fh:
        goldp = json.load(fh)
    for kind, g in goldp.items():
        n = g["n"]
        dev, _ = make_pairs(bz, ref, n, kind, "l1")
        a = bz.alps(*dev, np.zeros(n), np.zeros(n))
        assert a[5] == g["status"] and a[2] == g["tot_it"] and abs(a[3] - g["tot_inner_it"]) <= 2, kind
        assert rel(a[0], np.array(g["x"])) <= 1e-8, kind


def test_headline_size_compact_form_matches_two_loop_oracle(bz, ref):
    """What bench.py times (n = 10^7, compact L-BFGS representation, the compile-time-specialised one-pass
    kernel with non-temporal streams) against the numpy oracle in the REFERENCE's two-loop form: the first
    states of the exact benchmark problem agree within the north-star tolerance, and the kernel that ran
    is the fused one."""
    n = 10_000_000
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    mg = float(np.finfo(float).eps)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=mg, directions=bz.LBFGS(5, compact=True)).c_opts(), x0)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, minimum_gamma=mg)
    st = it.init()
    for k in range(9):
        sc = prob.panoc_scalars()
        assert abs(sc["gamma"] - float(st.gamma)) <= 1e-13 * float(st.gamma)
        ex, ez = rel(prob.panoc_vector("x"), st.x), rel(prob.panoc_vector("z"), st.z)
        assert ex <= RTOL_ITER and ez <= RTOL_ITER, f"iterate mismatch at k={k + 1}: {ex} {ez}"
        if k < 8:
            prob.panoc_step()
            st = it.step(st)
    assert prob.panoc_stats().n_fused_iters >= 7 and int(prob.panoc_scalars()["lbfgs_mem"]) == 5
    prob.close()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("form", ["compact", "two-loop"])
def test_headline_size_30_states_both_forms(bz, ref, form):
    """THIRTY PANOCplus states of the exact benchmark problem (n = 10^7) against the numpy oracle
    in the reference's two-loop form, within the north-star tolerance 1e-10 — for what bench.py times (the compact
    representation in the one-pass kernel) and for the two-loop kernels (persistent kernel + fused pass)."""
    n = 10_000_000
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    mg = float(np.finfo(float).eps)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=mg,
                                  directions=bz.LBFGS(5, compact=form == "compact")).c_opts(), x0)
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, minimum_gamma=mg)
    st = it.init()
    worst = 0.0
    for k in range(30):
        sc = prob.panoc_scalars()
        assert abs(sc["gamma"] - float(st.gamma)) <= 1e-13 * float(st.gamma), k
        ex, ez = rel(prob.panoc_vector("x"), st.x), rel(prob.panoc_vector("z"), st.z)
        worst = max(worst, ex, ez)
        assert ex <= RTOL_ITER and ez <= RTOL_ITER, f"iterate mismatch at state {k + 1}: {ex} {ez}"
        assert abs(sc["stop_norm"] - float(it.stop_norm(st))) <= 1e-9 * max(1.0, float(it.stop_norm(st)))
        if k < 29:
            prob.panoc_step()
            st = it.step(st)
    print(f"[{form}] worst relative iterate error over 30 states at n=1e7: {worst:.3e}")
    stt = prob.panoc_stats()
    assert stt.n_fused_iters >= 27 - stt.n_backtracks
    prob.close()


def test_dual_safeguard_clamp_on_the_device(bz, ref):
    """safeguards.jl:2-10 inside bz_alps_solve: y0 far beyond +-1e20 is clamped by the device pass (k_muy with the
    safeguard) before the first subproblem, exactly as default_dual_safeguard! does on the host; the solve then
    follows the oracle started from the same y0."""
    n = 4096
    d, dev, orc = make_cfg2(bz, ref, n)
    y0 = np.zeros(n)
    y0[::7] = 3e20
    y0[1::7] = -7e22
    y0[2::7] = 1e20          # exactly the bound
    y0[3::7] = 5.0
    # one outer iteration around a one-state subsolve (maxit = 1: the initial state's z): the returned y is
    # (c(x) + mu*y_clamped - proj_D(.))/mu — an element-wise function of the CLAMPED multipliers
    sub = lambda **kw: bz.PANOCplus(maxit=1, **kw)
    rsub = lambda **kw: ref.PANOCplus(maxit=1, **kw)
    a = bz.alps(*dev, np.zeros(n), y0, maxit=1, subsolver=sub)
    o = ref.alps(*orc, np.zeros(n), y0.copy(), maxit=1, subsolver=rsub)
    assert a[2] == o[2] == 1 and a[3] == o[3] == 1 and a[5] == o[5]
    assert np.all(np.isfinite(a[1])) and np.max(np.abs(a[1])) <= 1.0001e20
    # ... and the clamp was what made them so: the oracle WITHOUT the safeguard ends elsewhere
    o_raw = ref.alps(*orc, np.zeros(n), y0.copy(), maxit=1, subsolver=rsub, dual_safeguard=lambda y, cx=None: None)
    assert np.max(np.abs(o_raw[1] - o[1])) >= 1e19
    assert np.max(np.abs(a[1] - o[1])) <= 1e-12 * np.max(np.abs(o[1]))
    assert np.max(np.abs(a[0] - o[0])) <= 1e-12 * max(1.0, np.max(np.abs(o[0])))
    assert np.array_equal(y0[::7], np.full_like(y0[::7], 3e20))          # y0 never mutated
    # the host-loop variant applies the Python safeguard: the same numbers
    b = bz.alps(*dev, np.zeros(n), y0, maxit=1, subsolver=sub, resident=False)
    assert np.max(np.abs(a[1] - b[1])) <= 1e-12 * np.max(np.abs(o[1]))


def test_gamma_below_minimum_warns_and_continues(bz, ref, capfd):
    """the `gamma < minimum_gamma` warning path (ProximalAlgorithms' @warn): with a minimum_gamma far above the step
    the problem needs, the backtracking stops at the first gamma below it, warns, and the iteration goes on with
    that gamma — same gamma and iterates as the oracle."""
    import warnings
    n = 5000
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 1e-3), np.zeros(n), np.full(n, 3.0)
    mg = 0.5
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(mu, y)
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=mg).c_opts(), x0)
    err = capfd.readouterr().err
    al = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        it = ref.PANOCplusIteration(al, ref.NonsmoothCostFun(orc[1]), x0, minimum_gamma=mg)
        st = it.init()
    assert float(st.gamma) < mg and any("too small" in str(x.message) for x in w)
    assert "stepsize `gamma` became too small" in err
    assert abs(prob.panoc_scalars()["gamma"] - float(st.gamma)) <= 1e-13 * float(st.gamma)
    for k in range(5):
        prob.panoc_step()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            st = it.step(st)
        assert abs(prob.panoc_scalars()["gamma"] - float(st.gamma)) <= 1e-13 * float(st.gamma)
        assert rel(prob.panoc_vector("z"), st.z) <= 1e-9
    prob.close()


def test_alps_termination_statuses(bz, ref):
    """alps.jl:87-90,105-113: `:max_iter` when the outer budget runs out, `:exception` when the objective turns
    NaN — same status, counts and (for max_iter) point as the oracle; resident and host outer loops alike."""
    import warnings
    n = 2000
    d, dev, orc = make_cfg2(bz, ref, n)
    x0, y0 = np.zeros(n), np.zeros(n)
    for maxit in (1, 2):
        o = ref.alps(*orc, x0, y0, maxit=maxit)
        for resident in (True, False):
            a = bz.alps(*dev, x0, y0, maxit=maxit, resident=resident)
            assert a[5] == o[5] == "max_iter" and a[2] == o[2] == maxit and a[3] == o[3]
            assert rel(a[0], o[0]) <= 1e-9 and rel(a[1], o[1]) <= 1e-8
    # NaN in the data: objective NaN after the first subproblem -> :exception (alps.jl:89,109)
    bn = d["b"].copy()
    bn[7] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o = ref.alps(ref.DiagQuadratic(d["q"], bn), orc[1], orc[2], orc[3], x0, y0, maxit=5, subsolver_maxit=50,
                     subsolver=lambda **kw: ref.PANOCplus(maxit=50, **kw))
    a = bz.alps(bz.DiagQuadratic(d["q"], bn), dev[1], dev[2], dev[3], x0, y0, maxit=5, subsolver_maxit=50,
                subsolver=lambda **kw: bz.PANOCplus(maxit=50, **kw), resident=True)
    assert o[5] == "exception" and a[5] == "exception" and a[2] == o[2]


@pytest.mark.timeout(900)
def test_config5_size_closed_form(bz, ref):
    """BASELINE config 5 size (n = 10^8, one GPU's worth of HBM: 21 vectors of 800 MB in flight): the
    size-independent closed form of the D = FreeSet variant, x_i = soft(b_i, lambda) / q_i, through the whole
    ALPS solve in both evaluations of the L-BFGS operator; x0 is not mutated."""
    n = 100_000_000
    d = bz.synth.l1_quadratic(n)
    xs = np.sign(d["b"]) * np.maximum(np.abs(d["b"]) - d["lam"], 0) / d["q"]
    x0, y0 = np.zeros(n), np.zeros(n)
    for compact in (True, False):
        out = bz.alps(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(), bz.FreeSet(), x0, y0,
                      tol=1e-8, subsolver=lambda **kw: bz.PANOCplus(directions=bz.LBFGS(5, compact=compact), **kw),
                      resident=True)
        assert out[5] == "first_order"
        assert np.max(np.abs(out[0] - xs)) <= 1e-6
        del out
    del xs
    # the config itself (D = Box[-1,1], multipliers in play): feasibility and the KKT system of
    # min f + g  s.t. x in [-1,1]:  0 in q x - b + y + lam sign(x)   (as test_full_size_properties at 10^7)
    out = bz.alps(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                  bz.ClosedSet(bz.IndBox(-1.0, 1.0)), x0, y0,
                  subsolver=lambda **kw: bz.PANOCplus(directions=bz.LBFGS(5, compact=True), **kw), resident=True)
    x, y = out[0], out[1]
    assert out[5] == "first_order"
    assert np.max(np.abs(x - np.clip(x, -1, 1))) <= 1e-6
    r = d["q"] * x - d["b"] + y
    viol = np.where(x > 1e-9, np.abs(r + d["lam"]), np.where(x < -1e-9, np.abs(r - d["lam"]),
                    np.maximum(np.abs(r) - d["lam"], 0)))
    assert np.max(viol) <= 1e-4
    assert not np.any(x0)


@pytest.mark.parametrize("n", [1000, 30011, 400003])
@pytest.mark.parametrize("D", ["box", "vc_pairs"])
def test_fused_start_of_a_solve_is_bitwise_neutral(bz, ref, n, D):
    """`k_begin_lip` (gradient at x + Lipschitz estimate from the gradient at x + 1, one pass) against the four
    kernels it replaces: the same gamma, f(x) and first iterates, bit for bit — also with a pairwise D, where
    the partner element moves with x + 1 as well."""
    n = n - (n % 2)
    d = bz.synth.l1_quadratic(n)
    Dset = bz.PairwiseSet("vc") if D == "vc_pairs" else bz.ClosedSet(bz.IndBox(d["lo"], d["hi"]))
    dev = (bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(), Dset)
    rng = np.random.default_rng(3)
    mu = rng.uniform(0.05, 2.0, n)
    y = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    runs = []
    try:
        for flag in ("0", "1"):
            os.environ["BZ_FUSED_BEGIN"] = flag
            prob = bz.Problem(*dev, n, n, np.float64)
            prob.set_multipliers(mu, y)
            prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), x0)
            sc0 = prob.panoc_scalars()
            for _ in range(3):
                prob.panoc_step()
            runs.append((sc0, prob.panoc_scalars(), prob.panoc_vector("x"), prob.panoc_vector("z")))
            prob.close()
    finally:
        os.environ.pop("BZ_FUSED_BEGIN", None)
    a, b = runs
    for key in ("gamma", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm"):
        assert a[0][key] == b[0][key] and a[1][key] == b[1][key], key
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


@pytest.mark.parametrize("n,iters,start", [(30011, 160, "random"), (400003, 60, "random"), (200003, 470, "zero")])
def test_history_as_iterates_and_lazy_z_are_bitwise_neutral(bz, ref, n, iters, start):
    """Two storage tricks of the one-pass compact kernel change WHAT is written, never a value:
    (1) once the last five iterations were plain ones, the stored pairs are re-formed from the last six iterates
        kept in the x / res rings (s = x_d - x, y = res - res_prev: the same subtractions) and s, y are no longer
        written; the first iteration that is not plain turns the snapshots back into pairs;
    (2) z is not stored and is re-materialised on demand;
    (3) BZ_XR=2 (the default): after one more plain iteration the residuals are not read from their ring either but
        re-evaluated from the six iterates (res = x - prox(x - gamma grad L(x)): the same operations on the same
        inputs), and res is no longer written;
    (4) BZ_UNI: uniform penalties mu (and mu*y = 0, the "zero" start) are detected and passed as numbers instead of
        being streamed.
    With all off, all on, and each alone, runs that go through gamma halvings, tau backtracks, and — the
    "zero" start run to convergence — skipped pairs and re-entries must produce identical bits."""
    d, dev, orc = make_cfg2(bz, ref, n)
    rng = np.random.default_rng(9)
    mu = np.full(n, 0.1)
    y = rng.standard_normal(n) if start == "random" else np.zeros(n)
    x0 = rng.standard_normal(n) * 0.05 if start == "random" else np.zeros(n)
    runs = []
    try:
        # (BZ_GFC pins one grid for every form of the kernel: by default the iterate-history form runs on half as
        # many workgroups, a different summation tree — the last run below, compared to rounding)
        for xr, skipz, uni, gfc in (("0", "0", "0", "2"), ("1", "1", "0", "2"), ("1", "0", "0", "2"), ("0", "1", "0", "2"),
                                    ("2", "1", "0", "2"), ("2", "0", "0", "2"), ("2", "1", "1", "2"), ("2", "1", "2", "2"),
                                    ("2", "1", "2", None)):
            os.environ["BZ_XR"], os.environ["BZ_SKIPZ"], os.environ["BZ_UNI"] = xr, skipz, uni
            if gfc is None:      # the default configuration: its own grid, backtracked trials through the one-pass kernel
                os.environ.pop("BZ_GFC", None)
                os.environ.pop("BZ_TRIALFUSE", None)
            else:
                os.environ["BZ_GFC"] = gfc
                os.environ["BZ_TRIALFUSE"] = "0"      # (rejected trials finish in the generic kernels in every form)
            prob = bz.Problem(*dev, n, n, np.float64)
            prob.set_multipliers(mu, y)
            prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=float(np.finfo(float).eps)).c_opts(), x0)
            prob.profile_enable(True)
            zs = []
            for k in range(iters):
                prob.panoc_step()
                if k % 37 == 5:
                    zs.append(prob.panoc_vector("z"))            # mid-run read-outs must not disturb anything
                if k % 41 == 17:
                    zs.append(prob.panoc_vector("res"))
            st = prob.panoc_stats()
            runs.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_vector("res"), prob.panoc_scalars(), zs,
                         (st.n_backtracks, st.n_gamma_halvings, st.n_lbfgs_skips, st.n_fused_iters), prob.profile()["misc"]["launches"],
                         prob.profile()["k_fused_iterates"]["launches"]))
            prob.close()
    finally:
        os.environ.pop("BZ_XR", None)
        os.environ.pop("BZ_SKIPZ", None)
        os.environ.pop("BZ_UNI", None)
        os.environ.pop("BZ_GFC", None)
        os.environ.pop("BZ_TRIALFUSE", None)
    base = runs[0]
    dflt = runs.pop()
    if start == "zero":      # run to convergence: the default configuration ends at the same point to rounding
        assert np.max(np.abs(dflt[0] - base[0])) <= 1e-11 * max(1.0, np.max(np.abs(base[0])))
        assert abs(dflt[3]["f_x"] - base[3]["f_x"]) <= 1e-12 * abs(base[3]["f_x"])
    assert dflt[5][3] >= iters - 16
    # the 9..11-pass form (timing category k_fused_iterates) serves every iteration, the first one included (empty
    # memory): it resumes straight after a tau backtrack, runs with a partial memory, and carries the pair of a
    # gamma-halving iteration (y = res_new(gamma/2) - res_prev(gamma)) through the gamma tag of the oldest iterate
    assert all(r[7] == 0 for r in runs[:4])
    if base[5][2] == 0:
        assert all(r[7] >= iters - 2 for r in runs[4:] + [dflt])
    for r in runs[1:]:
        assert np.array_equal(r[0], base[0]) and np.array_equal(r[1], base[1]) and np.array_equal(r[2], base[2])
        assert all(np.array_equal(a, b) for a, b in zip(r[4], base[4]))
        for key in ("gamma", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_H", "FBE"):
            assert r[3][key] == base[3][key], key
        assert r[5] == base[5]
    assert base[5][3] >= iters - 12                      # the fused pass served (almost) every iteration
    if start == "zero":
        assert base[5][2] > 0                            # pairs were skipped: the snapshots had to become pairs again
        assert runs[1][6] > runs[0][6]                   # ... by k_pairs_from_snapshots (category misc)
        assert runs[4][6] > runs[0][6]                   # ... or k_pairs_from_iterates


def test_persistent_kernel_barrier_timeout_falls_back_to_the_kernel_chain(bz, ref, monkeypatch):
    """a grid barrier of the persistent two-loop kernel that cannot complete (workgroups not all
    resident) must not lose the solve.  BZ_TEST_PERSIST_TIMEOUT=1 makes the barrier miss its target; the bounded
    polls give up, the iteration is redone with the kernel chain and the persistent form stays off: the iterates
    are those of a persist=False solve bit for bit, and the statistics say what happened."""
    n = 400_003
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.cos(np.arange(n, dtype=np.float64)), np.zeros(n)
    runs = []
    for sabotage, persist in ((True, True), (False, False)):
        if sabotage:
            monkeypatch.setenv("BZ_TEST_PERSIST_TIMEOUT", "1")
        else:
            monkeypatch.delenv("BZ_TEST_PERSIST_TIMEOUT", raising=False)
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, persist=persist, directions=bz.LBFGS(5, compact=False)).c_opts(), x0)
        for _ in range(8):
            prob.panoc_step()
        runs.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars(), prob.panoc_stats(),
                     prob.profile2()))
        prob.close()
    (xa, za, sa, sta, _), (xb, zb, sb, stb, _) = runs
    assert sta.persist_fallbacks == 1 and stb.persist_fallbacks == 0
    assert sa["k"] == sb["k"] == 9 and sta.n_grad == stb.n_grad and sta.n_prox == stb.n_prox
    assert np.array_equal(xa, xb) and np.array_equal(za, zb)
    for key in ("gamma", "f_x", "g_z", "stop_norm", "FBE"):
        assert sa[key] == sb[key]


def test_moved_bytes_accounting_of_the_one_pass_kernel(bz, ref):
    """bz_profile_get2: the bytes a launch is designed to move are its streams x n x 8.  Steady state of the
    headline family with uniform penalties and y = 0: the six last iterates, q, b in, x_d out = 9 passes."""
    n = 400_003
    d, dev, orc = make_cfg2(bz, ref, n)
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(np.full(n, 0.1), np.zeros(n))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(n))
    for _ in range(30):
        prob.panoc_step()
    prob.profile_reset()
    prob.profile_enable(True)
    st0 = prob.panoc_stats()
    for _ in range(10):
        prob.panoc_step()
    st1 = prob.panoc_stats()
    p = prob.profile2()["k_fused_iterates"]
    prob.close()
    if st1.n_backtracks == st0.n_backtracks and st1.n_lbfgs_skips == st0.n_lbfgs_skips:
        assert p["launches"] == 10 and p["timed_launches"] == 10
        assert p["bytes"] == 10 * 9 * 8 * n and p["timed_bytes"] == p["bytes"]
        assert p["form"].startswith("k_fused_compact<XR=2,UNI=2,NT=") and p["form"].endswith("TRIAL=0>")
    assert 0 < p["timed_bytes"] / (p["timed_ms"] * 1e-3) / 8e12 <= 1.0


@pytest.mark.parametrize("fam", ["headline", "l1box"])
def test_gated_prelaunch_is_bitwise_neutral(bz, ref, fam, monkeypatch):
    """The next iteration's one-pass kernel launched EARLY behind the read-back and released through its gate
    (bz_panoc_steps: the library runs the loop) against plain launches: the same bits through tau backtracks, gamma
    halvings and skipped pairs — the recalled launches leave without touching anything."""
    n = 400_003
    d, dev, orc = make_cfg2(bz, ref, n, g="l1" if fam == "headline" else "l1box")
    rng = np.random.default_rng(21)
    mu, y, x0 = np.full(n, 0.1), rng.standard_normal(n), 0.05 * rng.standard_normal(n)
    runs = []
    for gate in ("0", "1"):
        monkeypatch.setenv("BZ_GATE", gate)
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=float(np.finfo(float).eps)).c_opts(), x0)
        for chunk in (1, 7, 50, 50, 33):
            prob.panoc_steps(chunk)
            zmid = prob.panoc_vector("z")                     # between the calls nothing is pending
        st = prob.panoc_stats()
        runs.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_vector("res"), zmid, prob.panoc_scalars(), st))
        prob.close()
    a, b = runs
    for u, v in zip(a[:4], b[:4]):
        assert np.array_equal(u, v)
    for key in ("k", "gamma", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_H", "FBE"):
        assert a[4][key] == b[4][key], key
    assert a[5].n_gated_launches == 0 and a[5].n_gate_aborts == 0
    assert b[5].n_gated_launches >= 100 and b[5].n_gated_launches + b[5].n_gate_aborts <= 141 - 5
    assert (a[5].n_backtracks, a[5].n_gamma_halvings, a[5].n_lbfgs_skips, a[5].n_grad) == \
        (b[5].n_backtracks, b[5].n_gamma_halvings, b[5].n_lbfgs_skips, b[5].n_grad)
    # whole solves through the library's own loop: same counts and point with and without the gate
    outs = []
    for gate in ("0", "1"):
        monkeypatch.setenv("BZ_GATE", gate)
        outs.append(bz.alps(*dev, np.zeros(n), np.zeros(n)))
    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3] and np.array_equal(outs[0][0], outs[1][0])


def test_shared_device_context_does_not_gate(bz, ref):
    """BZ_CTX_SHARED_DEVICE: the host says the GPU is not its own — the library's loop makes plain launches only"""
    n = 200_003
    d, dev, orc = make_cfg2(bz, ref, n)
    outs = []
    for shared in (False, True):
        ctx = bz.Context(device=0, shared_device=shared)
        prob = bz.Problem(*dev, n, n, np.float64, ctx)
        prob.set_multipliers(np.full(n, 0.1), np.zeros(n))
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(n))
        prob.panoc_steps(30)
        outs.append((prob.panoc_vector("x"), prob.panoc_stats()))
        prob.close()
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1].n_gated_launches >= 20 and outs[1][1].n_gated_launches == 0


@pytest.mark.parametrize("n,persist", [(300_007, True), (100_000, True), (300_007, False)],
                         ids=["persist-capable", "below-the-persistent-size", "persist-off"])
def test_gate_timeout_falls_back_to_plain_launches(bz, ref, monkeypatch, n, persist):
    """A pre-launched pass that is not released in time (a stalled host thread; another tenant on the GPU keeping its
    first workgroup from becoming resident) leaves as a whole, the host redoes the iteration with a plain launch and keeps
    the gate off for the problem: the same bits as a solve that never used the gate, one fall-back in the statistics.
    (BZ_TEST_GATE_TIMEOUT=k: the k-th release is withheld; BZ_GATE_SPIN shortens the poll bounds from ~3 s.)
    The gate does not depend on the persistent two-loop kernel being available, nor does its fall-back (the
    redo used to exist only where that kernel was — n >= BZ_PERSIST_MIN_N = 300000 and `persist` on)."""
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    runs = []
    for gate, sab in (("0", "0"), ("1", "9")):
        monkeypatch.setenv("BZ_GATE", gate)
        monkeypatch.setenv("BZ_TEST_GATE_TIMEOUT", sab)
        monkeypatch.setenv("BZ_GATE_SPIN", "20000")
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, persist=persist).c_opts(), x0)
        prob.panoc_steps(40)
        st = prob.panoc_stats()
        runs.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars(), st))
        # the gate stays off for later solves on this problem
        monkeypatch.setenv("BZ_TEST_GATE_TIMEOUT", "0")
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, persist=persist).c_opts(), x0)
        prob.panoc_steps(10)
        runs[-1] += (prob.panoc_stats(),)
        prob.close()
    a, b = runs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for key in ("k", "gamma", "f_x", "stop_norm", "FBE"):
        assert a[2][key] == b[2][key], key
    assert a[3].n_gate_fallbacks == 0 and b[3].n_gate_fallbacks == 1
    assert 1 <= b[3].n_gated_launches <= 10 and a[3].n_grad == b[3].n_grad
    assert b[4].n_gated_launches == 0


def test_gate_timeout_inside_a_whole_solve(bz, ref, monkeypatch):
    """... and the same inside the solver's own loop (bz_alps_solve -> run_to_completion): the solve ends where the ungated one
    does, with one fall-back, and nothing is left pre-launched when the call returns (the next call on the problem works)."""
    n = 60_000
    d, dev, orc = make_cfg2(bz, ref, n)
    monkeypatch.setenv("BZ_GATE_SPIN", "20000")
    outs = []
    for gate, sab in (("0", "0"), ("1", "12")):
        monkeypatch.setenv("BZ_GATE", gate)
        monkeypatch.setenv("BZ_TEST_GATE_TIMEOUT", sab)
        prob = bz.Problem(*dev, n, n, np.float64)
        outs.append(bz.alps(*dev, np.zeros(n), np.zeros(n), problem=prob))
        monkeypatch.setenv("BZ_TEST_GATE_TIMEOUT", "0")
        prob.set_multipliers(np.full(n, 0.1), np.zeros(n))
        z, st = prob.panoc_solve(bz.PANOCplus(tol=1e-6).c_opts(), np.zeros(n))
        outs[-1] += (z, st)
        prob.close()
    a, b = outs
    assert a[5] == b[5] == "first_order" and a[2] == b[2] and a[3] == b[3]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[10], b[10]) and a[11].iters == b[11].iters
    assert a[11].n_gate_fallbacks == 0 and b[11].n_gate_fallbacks == 1 and b[11].n_gated_launches == 0


@pytest.mark.timeout(900)
def test_headline_size_library_loop_equals_single_steps_bitwise(bz, ref):
    """What bench.py TIMES against what the 30-state oracle tests STEP, at the benchmark size:
    bz_panoc_steps (the library runs the loop: the next pass pre-launched behind its gate, z not stored after the first
    20 iterations and re-materialised on demand, the non-temporal NT = 1 instantiation that n = 1e7 selects) and the same
    number of bz_panoc_step calls with x and z read after each (plain launches, z forced every state): the same bits in
    x, z, res and every scalar."""
    n = 10_000_000
    d, dev, orc = make_cfg2(bz, ref, n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    mg = float(np.finfo(float).eps)
    iters = 45
    runs = []
    for mode in ("steps", "single"):
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=mg).c_opts(), x0)
        if mode == "steps":
            prob.panoc_steps(iters)
        else:
            for _ in range(iters):
                prob.panoc_step()
                prob.panoc_vector("z")
        st = prob.panoc_stats()
        form = prob.profile2()["k_fused_iterates"]["form"]
        runs.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_vector("res"), prob.panoc_scalars(), st, form))
        prob.close()
    a, b = runs
    assert "NT=1" in a[5] and "NT=1" in b[5], (a[5], b[5])
    assert a[4].n_gated_launches >= iters - 8 and b[4].n_gated_launches == 0
    for u, v in zip(a[:3], b[:3]):
        assert np.array_equal(u, v)
    for key in ("k", "gamma", "tau", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_mem", "lbfgs_H",
                "al_z", "f_z", "FBE"):
        assert a[3][key] == b[3][key], key
    assert (a[4].n_backtracks, a[4].n_gamma_halvings, a[4].n_lbfgs_skips, a[4].n_grad, a[4].n_prox) == \
        (b[4].n_backtracks, b[4].n_gamma_halvings, b[4].n_lbfgs_skips, b[4].n_grad, b[4].n_prox)
