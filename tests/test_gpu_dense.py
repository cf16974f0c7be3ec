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


def test_affine_images_halve_the_passes_over_A(bz, ref):
    """traffic accounting: with images the gemv categories see 2 launches per iteration instead of 4"""
    ny, n = 512, 4096
    d, dev, orc = make_cfg4(bz, ref, ny, n, np.float32, density=0.05)
    mu, y, x0 = np.full(ny, 0.1, np.float32), np.zeros(ny, np.float32), np.zeros(n, np.float32)
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
        counts[refresh] = (p["gemv"]["launches"] + p["k_gemv_t_mfma"]["launches"], st.n_backtracks, st.n_gamma_halvings)
        prob.close()
    assert counts[0][0] >= 4 * 16
    assert counts[8][0] <= 2 * 16 + 2 * 2 + 4 * (counts[8][1] + counts[8][2]) + 2      # two refreshes in 16 iterations


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
@pytest.mark.parametrize("refresh,states", [(16, 22), (8, 8), (0, 8)])
def test_dense_full_size_iterates_fp32(bz, ref, refresh, states):
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
    prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, x0, states, minimum_gamma=float(np.finfo(np.float32).eps),
                                dtype=np.float32, ny=ny, affine_refresh=refresh)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"cfg4_full_rows_refresh{refresh}.log"), "w") as fh:
        for r in rows:
            fh.write("k=%d ex=%.3e ez=%.3e gamma_dev=%.9g gamma_ref=%.9g stop_dev=%.4e stop_ref=%.4e fused=%d sens=%.3e\n" % tuple(r))
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert sens <= 1e-4, (k, sens)
        assert abs(g_d - g_r) <= 1e-4 * g_r, (k, g_d, g_r)
        assert ex <= max(2e-4, 100 * sens) and ez <= max(2e-4, 100 * sens), (k, ex, ez, sens)
    stats = prob.panoc_stats()
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
    assert p["k_gemv_t_mfma"]["launches"] >= 4 and p["k_gemv_t_mfma"]["form"] == "k_gemv_t_mfma"
    assert 8.0e12 >= p["k_gemv_t_mfma"]["timed_bytes"] / (p["k_gemv_t_mfma"]["timed_ms"] * 1e-3) >= 3.0e12
    prob.close()
