"""cfg 4 (dense affine constraint, demo/basispursuit.jl:38-49): affine images and full-size iterate parity.

Affine images: with c(x) = A x - b, D = ZeroSet / FreeSet and f = Zero / DiagQuadratic both c(.)
and grad L(.) are affine maps, so their values at the trial point x + d follow from the stored images of the iterates
(the linear combination that forms d) with no pass over A; a pass-over-A evaluation every `affine_refresh`-th iteration
bounds the rounding drift.  An iteration then reads A twice (the gradient at z) instead of four times.  Accepted only
because the iterates stay inside the oracle's own rounding envelope, fp64 and fp32, over 30 states — checked here."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import make_cfg4, rel, run_traces

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

_FULL = {}


def full_size_cfg4(bz, ref):
    """the 2 GiB matrix is generated once per session (splitmix64 in numpy: ~1 min)"""
    if "d" not in _FULL:
        _FULL["d"] = make_cfg4(bz, ref, 8192, 65536, np.float32, density=0.01)
    return _FULL["d"]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(64, 512), (257, 1028)])
@pytest.mark.parametrize("refresh", [8, 16, 32, 0])
def test_dense_iterates_follow_oracle_with_and_without_affine_images(bz, ref, shape, dtype, refresh):
    ny, n = shape
    d, dev, orc = make_cfg4(bz, ref, ny, n, dtype, density=0.05)
    mu, y = np.full(ny, 0.1, dtype), (0.1 * np.random.default_rng(2).standard_normal(ny)).astype(dtype)
    x0 = np.zeros(n, dtype)
    eps = float(np.finfo(dtype).eps)
    prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, x0, 30, minimum_gamma=eps, dtype=dtype, ny=ny,
                                affine_refresh=refresh)
    stats = prob.panoc_stats()
    base = 1e-9 if dtype == np.float64 else 5e-5
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert abs(g_d - g_r) <= (1e-12 if dtype == np.float64 else 1e-5) * g_r, k
        assert ex <= max(base, 100 * sens) and ez <= max(base, 100 * sens), (k, ex, ez, sens)
    if refresh:
        # 29 iterations, a pass-over-A evaluation every refresh-th (and wherever a step-size test failed on images)
        assert 29 - 29 // refresh - 6 <= stats.n_affine_images <= 29
    else:
        assert stats.n_affine_images == 0
    prob.close()


@pytest.mark.parametrize("blend", ["1", "0"])
def test_backtracked_trial_points_on_images_follow_the_oracle(bz, ref, blend, monkeypatch):
    """A tau-backtracked trial point is an affine combination of x + d and the state's z, so its images under c and grad L
    are that combination of images already held: the first backtrack of an iteration costs no pass over A (r03;
    `BZ_AFFINE_BLEND=0`: evaluated as before).  60 states of a solve that backtracks six times, against the oracle, inside
    the oracle's own rounding envelope — and the passes over A that the images save."""
    ny, n = 64, 512
    monkeypatch.setenv("BZ_AFFINE_BLEND", blend)
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float64, density=0.05)
    mu, y = np.full(ny, 0.1), 0.1 * np.random.default_rng(2).standard_normal(ny)
    prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, np.zeros(n), 60, minimum_gamma=2.3e-16, ny=ny, affine_refresh=16)
    stats = prob.panoc_stats()
    assert stats.n_backtracks >= 4
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert abs(g_d - g_r) <= 1e-12 * g_r, k
        assert ex <= max(1e-9, 100 * sens) and ez <= max(1e-9, 100 * sens), (k, ex, ez, sens)
    _BLEND_GRADS[blend] = (stats.n_grad, stats.n_dense_onepass, stats.n_backtracks)
    if len(_BLEND_GRADS) == 2:
        a, b = _BLEND_GRADS["1"], _BLEND_GRADS["0"]
        assert a[2] == b[2] and a[0] == b[0]              # the same trajectory of decisions ...
        assert a[1] <= b[1] - 2, (a, b)                   # ... with fewer passes over A (not one per backtrack: the second backtrack
                                                          # of an iteration, a trial whose step-size test fails and refresh iterations
                                                          # are evaluated)
    prob.close()


_BLEND_GRADS = {}


@pytest.mark.parametrize("fused", ["1", "0"])
def test_affine_images_halve_the_passes_over_A(bz, ref, fused, monkeypatch):
    """traffic accounting (bytes the launches are designed to move, in passes over A): an AL gradient is ONE pass with the
    one-pass kernel and two as k_gemv_n + k_gemv_t; an iteration evaluates two gradients, one of them (at the trial point)
    replaced by images between refreshes"""
    ny, n = 512, 4096
    monkeypatch.setenv("BZ_DENSE_FUSED", fused)
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float32, density=0.05)
    mu, y, x0 = np.full(ny, 0.1, np.float32), np.zeros(ny, np.float32), np.zeros(n, np.float32)
    per_grad = 1 if fused == "1" else 2
    counts = {}
    for refresh in (0, 8):
        prob = bz.Problem(*dev, n, ny, np.float32)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=1e-7, affine_refresh=refresh).c_opts(), x0)
        prob.profile_reset()
        prob.profile_enable(True)
        prob.panoc_steps(16)
        p = prob.profile2()
        st = prob.panoc_stats()
        passes = (p["gemv"]["bytes"] + p["k_gemv_t_mfma"]["bytes"]) / (4.0 * ny * n)
        counts[refresh] = (passes, st.n_backtracks, st.n_gamma_halvings, st.n_dense_onepass, p["gemv"]["form"])
        prob.close()
    assert counts[0][0] >= 2 * per_grad * 16
    # two refreshes in 16 iterations; a backtrack or a halving costs up to four more gradients (a failing step-size test on
    # images is re-run on evaluations, then the trial is repeated)
    assert counts[8][0] <= per_grad * (16 + 2 + 4 * (counts[8][1] + counts[8][2]) + 3) * 1.01
    assert (counts[0][3] > 0) == (fused == "1") and counts[0][4].startswith("k_dense_fused" if fused == "1" else "k_gemv_n")


@pytest.mark.parametrize("f", ["zero", "diag"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(512, 4096), (260, 1028)])
def test_short_kernels_as_two_launches_are_bitwise_neutral(bz, ref, shape, dtype, f, monkeypatch):
    """cfg 4's eleven short element-wise kernels either side of the pass over A run as TWO launches (k_dense_head: x_d, its
    images under grad L and c, L(x_d), the forward-backward step; k_dense_tail: the fold of the row-group partials, the pair
    with its products, the pair's images): the kernels' own bodies on the kernels' own block -> chunk maps, so every vector
    and every scalar of every iteration keeps its bits — across refreshes, backtracks and step-size halvings — and the
    iteration is five launches (head, pass over A, tail, read-back; the hand-over of the pair) instead of twelve."""
    ny, n = shape
    d, dev, orc = make_cfg4(bz, ref, ny, n, dtype, density=0.05)
    if f == "diag":
        rng = np.random.default_rng(9)
        dev = (bz.DiagQuadratic(rng.uniform(0.5, 2.0, n).astype(dtype), rng.standard_normal(n).astype(dtype)),) + tuple(dev[1:])
    mu, y = np.full(ny, 0.1, dtype), (0.1 * np.random.default_rng(2).standard_normal(ny)).astype(dtype)
    x0 = np.zeros(n, dtype)
    runs = {}
    for small in ("1", "0"):
        monkeypatch.setenv("BZ_DENSESMALL", small)
        prob = bz.Problem(*dev, n, ny, dtype)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=float(np.finfo(dtype).eps), affine_refresh=8).c_opts(), x0)
        trace = []
        for k in range(40):
            if k == 20:
                prob.profile_reset(); prob.profile_enable(True)
            prob.panoc_step()
            if k % 4 == 3 or k < 4:
                trace.append((prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_vector("res"), prob.panoc_scalars()))
        p = prob.profile2()
        st = prob.panoc_stats()
        runs[small] = (trace, sum(v["launches"] for v in p.values()), st.n_affine_images, st.n_backtracks, st.n_gamma_halvings,
                       st.n_dense_onepass)
        prob.close()
    a, b = runs["1"], runs["0"]
    assert a[2:] == b[2:] and a[2] >= 25, (a[2:], b[2:])
    for (xa, za, ra, sa), (xb, zb, rb, sb) in zip(a[0], b[0]):
        assert np.array_equal(xa, xb) and np.array_equal(za, zb) and np.array_equal(ra, rb)
        for key in ("k", "gamma", "tau", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_mem", "lbfgs_H", "FBE"):
            assert sa[key] == sb[key] or (sa[key] != sa[key] and sb[key] != sb[key]), key
    # 20 profiled iterations: a plain one is 5 launches against 12 (refresh iterations and rejected trials keep their chains)
    assert a[1] <= b[1] - 6 * 12, (a[1], b[1])


def test_affine_images_with_diag_quadratic_f_and_free_set(bz, ref):
    """the other members of the affine class: f = DiagQuadratic (its gradient is affine too), D = FreeSet"""
    ny, n = 48, 300
    rng = np.random.default_rng(4)
    A = rng.standard_normal((ny, n)) / np.sqrt(ny)
    b = rng.standard_normal(ny)
    q, bb = rng.uniform(0.5, 2.0, n), rng.standard_normal(n)
    for Dd, Dr in ((bz.ZeroSet(), ref.ZeroSet()), (bz.FreeSet(), ref.FreeSet())):
        dev = (bz.DiagQuadratic(q, bb), bz.NormL1(0.2), bz.DenseAffine(A, b), Dd)
        orc = (ref.DiagQuadratic(q, bb), ref.NormL1(0.2), ref.DenseAffine(A, b), Dr)
        mu, y = np.full(ny, 0.3), rng.standard_normal(ny)
        prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, np.zeros(n), 30, minimum_gamma=2.3e-16, ny=ny)
        for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
            assert ex <= max(1e-9, 100 * sens) and ez <= max(1e-9, 100 * sens), (k, ex, ez, sens)
        assert prob.panoc_stats().n_affine_images >= 20
        prob.close()
        a = bz.alps(*dev, np.zeros(n), np.zeros(ny))
        o = ref.alps(*orc, np.zeros(n), np.zeros(ny))
        assert a[5] == o[5] == "first_order" and a[2] == o[2]
        assert np.max(np.abs(a[0] - o[0])) <= 1e-6


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("refresh,states,onepass", [(16, 22, True), (8, 8, True), (0, 8, True), (16, 8, False)],
                         ids=["refresh16-22states", "refresh8", "no-images", "two-kernel-form"])
def test_dense_full_size_iterates_fp32(bz, ref, refresh, states, onepass, monkeypatch):
    """BASELINE config 4 at full size (A 8192 x 65536 fp32 = 2 GiB) against the numpy oracle in fp32, the MFMA
    transposed product in the loop: at the SHIPPED default affine_refresh = 16 — what `bench.py --workload cfg4` times —
    for 22 states, so that a pass-over-A refresh of the images is crossed (iteration 16) and five more iterations run on
    the refreshed images (demo/basispursuit.jl:38-49,62-66); with refresh 8 and with every
    gradient evaluated by passes over A for 8 states.

    The tolerance is an envelope around the oracle's own rounding sensitivity (its twin with long-double reductions),
    so the sensitivity itself is bounded here: an envelope that scales with the oracle's noise must not be able to
    swallow a drift of the images."""
    ny, n = 8192, 65536
    d, dev, orc = full_size_cfg4(bz, ref)
    mu, y, x0 = np.full(ny, 0.1, np.float32), np.zeros(ny, np.float32), np.zeros(n, np.float32)
    # (the library default is the one-pass kernel: every row of A read once per gradient; BZ_DENSE_FUSED=0 keeps k_gemv_n and
    # the MFMA transposed product, which stay the form of the row-sharded and dense-f paths)
    monkeypatch.setenv("BZ_DENSE_FUSED", "1" if onepass else "0")
    prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, x0, states, minimum_gamma=float(np.finfo(np.float32).eps),
                                dtype=np.float32, ny=ny, affine_refresh=refresh)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"cfg4_full_rows_refresh{refresh}{'' if onepass else '_two_kernel'}.log"), "w") as fh:
        for r in rows:
            fh.write("k=%d ex=%.3e ez=%.3e gamma_dev=%.9g gamma_ref=%.9g stop_dev=%.4e stop_ref=%.4e fused=%d sens=%.3e\n" % tuple(r))
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        # measured (r03, gpurun_out/cfg4_full_rows_refresh*.log): the oracle's own sensitivity stays below 1e-6 over the 22
        # states and the device within 1.5e-6 of it — one fp32 product over n = 65536 terms rounds to ~1e-5; no envelope
        # that scales with the oracle's noise is needed, so none is used
        assert sens <= 1e-5, (k, sens)
        assert abs(g_d - g_r) <= 1e-5 * g_r, (k, g_d, g_r)
        assert ex <= 1e-5 and ez <= 1e-5, (k, ex, ez, sens)
    stats = prob.panoc_stats()
    assert stats.n_dense_fallbacks == 0
    assert (stats.n_dense_onepass >= states - 1) if onepass else (stats.n_dense_onepass == 0)      # (every gradient at z: one pass over A)
    if refresh:
        # every iteration but the refreshes (and wherever a step-size test failed on images) ran on images
        its = states - 1
        assert its - its // refresh - 4 <= stats.n_affine_images <= its
    else:
        assert stats.n_affine_images == 0
    prob.profile_reset()
    prob.profile_enable(True)
    prob.panoc_steps(4)
    p = prob.profile2()
    if onepass:
        assert p["gemv"]["launches"] >= 4 and p["gemv"]["form"].startswith("k_dense_fused<KP=4") and p["k_gemv_t_mfma"]["launches"] == 0
        assert 8.0e12 >= p["gemv"]["timed_bytes"] / (p["gemv"]["timed_ms"] * 1e-3) >= 3.0e12
    else:
        assert p["k_gemv_t_mfma"]["launches"] >= 4 and p["k_gemv_t_mfma"]["form"] == "k_gemv_t_mfma"
        assert 8.0e12 >= p["k_gemv_t_mfma"]["timed_bytes"] / (p["k_gemv_t_mfma"]["timed_ms"] * 1e-3) >= 3.0e12
    prob.close()


# ---- the dense constraint in ONE pass over A (k_dense_fused; demo/basispursuit.jl:38-49) --------------------
ONEPASS_SHAPES = [(5, 12), (64, 512), (257, 1028), (1030, 4100), (300, 20000), (33, 3000), (2051, 16384)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", ONEPASS_SHAPES, ids=[f"{a}x{b}" for a, b in ONEPASS_SHAPES])
@pytest.mark.parametrize("D", ["zero", "free", "box"])
def test_dense_one_pass_gradient_matches_oracle_and_two_kernel_form(bz, ref, shape, dtype, D, monkeypatch):
    """gradient!(dlx, al, x) (auglagfun.jl:73-86) with c(x) = A x - b: the one-pass kernel (every row of A read once; row
    groups shared by G workgroups that exchange partial products) against the numpy oracle and against the two-kernel form
    (k_gemv_n + k_gemv_t), over ragged shapes: rows not a multiple of the 4-row tile, a last column slice with a single
    pack, one slice and many, more row groups than rows."""
    ny, n = shape
    rng = np.random.default_rng(ny * 7 + n)
    A = (rng.standard_normal((ny, n)) / np.sqrt(ny)).astype(dtype)
    b = rng.standard_normal(ny).astype(dtype)
    x = (rng.standard_normal(n) * (rng.random(n) < 0.2)).astype(dtype)
    mu = (10.0 ** rng.uniform(-2, 0, ny)).astype(dtype)
    y = rng.standard_normal(ny).astype(dtype)
    mk = {"zero": lambda m: m.ZeroSet(), "free": lambda m: m.FreeSet(),
          "box": lambda m: m.ClosedSet(m.IndBox(dtype(-0.3), dtype(0.4)) if m is ref else m.IndBox(-0.3, 0.4))}[D]
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("BZ_DENSE_FUSED", fused)
        prob = bz.Problem(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A, b), mk(bz), n, ny, dtype)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=float(np.finfo(dtype).eps)).c_opts(), x)
        onepass = prob.panoc_stats().n_dense_onepass
        assert (onepass >= 2) == (fused == "1"), onepass
        outs[fused] = prob.eval_al_gradient(x)
        prob.close()
    al = ref.AugLagFun(ref.Zero(), ref.DenseAffine(A, b), mk(ref), mu.copy(), y.copy(), x)
    g_ref = np.empty(n, dtype)
    L_ref = al.gradient(g_ref, x)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    scale = max(1.0, float(np.max(np.abs(g_ref))))
    for fused in ("1", "0"):
        g, vals = outs[fused]
        assert np.max(np.abs(g - g_ref)) <= tol * scale, (fused, np.max(np.abs(g - g_ref)))
        assert abs(vals[0] - L_ref) <= tol * max(1.0, abs(L_ref)), (fused, vals[0], L_ref)
    assert np.max(np.abs(outs["1"][0] - outs["0"][0])) <= tol * scale


def test_dense_one_pass_is_deterministic(bz, ref):
    """the partial products meet in slice order and the row groups fold in group order whatever the arrival order: two
    evaluations give the same bits"""
    ny, n = 513, 8192
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float32, density=0.05)
    mu, y = np.full(ny, 0.1, np.float32), (0.1 * np.random.default_rng(2).standard_normal(ny)).astype(np.float32)
    x = np.random.default_rng(3).standard_normal(n).astype(np.float32)
    prob = bz.Problem(*dev, n, ny, np.float32)
    prob.set_multipliers(mu, y)
    g0, v0 = prob.eval_al_gradient(x)
    for _ in range(5):
        g1, v1 = prob.eval_al_gradient(x)
        assert np.array_equal(g0, g1) and v0 == v1
    prob.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("where", ["in-the-loop", "at-the-start"])
def test_dense_one_pass_timeout_falls_back_to_two_kernels(bz, ref, monkeypatch, where):
    """The G workgroups of a row group wait for each other's partial products: with CUs held by another tenant a group can be
    partly resident.  Every poll is bounded; a group that gives up reports it, the host redoes the iteration (or the start of
    the solve) with the two-kernel form and stays there.  (BZ_TEST_DENSE_TIMEOUT=k: in the k-th launch slice 0 posts under
    tags nobody waits for; BZ_DENSE_SPIN shortens the poll bound.)"""
    ny, n = 600, 8192          # G = 4 slices
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float32, density=0.05)
    mu, y, x0 = np.full(ny, 0.1, np.float32), np.zeros(ny, np.float32), np.zeros(n, np.float32)
    monkeypatch.setenv("BZ_DENSE_SPIN", "20000")
    monkeypatch.setenv("BZ_TEST_DENSE_TIMEOUT", "9" if where == "in-the-loop" else "2")
    prob = bz.Problem(*dev, n, ny, np.float32)
    prob.set_multipliers(mu, y)
    z, st = prob.panoc_solve(bz.PANOCplus(tol=1e-4, maxit=400, minimum_gamma=1e-7).c_opts(), x0)
    assert st.n_dense_fallbacks == 1 and st.status == 0
    prob.close()
    monkeypatch.setenv("BZ_TEST_DENSE_TIMEOUT", "0")
    monkeypatch.setenv("BZ_DENSE_FUSED", "0")
    prob = bz.Problem(*dev, n, ny, np.float32)
    prob.set_multipliers(mu, y)
    z2, st2 = prob.panoc_solve(bz.PANOCplus(tol=1e-4, maxit=400, minimum_gamma=1e-7).c_opts(), x0)
    prob.close()
    assert st2.n_dense_fallbacks == 0 and st2.n_dense_onepass == 0
    assert abs(st.iters - st2.iters) <= max(3, 0.1 * st2.iters)
    assert np.max(np.abs(z - z2)) <= 1e-3 * max(1.0, np.max(np.abs(z2)))
