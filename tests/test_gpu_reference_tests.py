"""The reference's OWN test problems (test/runtests.jl) run through the device path.

Each test mirrors the Julia test line by line — same data, same call, same assertions — with
`Bazinga.alps` replaced by the drop-in `bazinga_jl_amd.alps` (HIP kernels behind the C ABI).
    test/problems/test_verbose.jl        -> test_verbose_lasso
    test/problems/test_nonconvex_qp.jl   -> test_nonconvex_qp_tiny / _small
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("resident", [True, False])
def test_verbose_lasso(bz, resident, capfd):
    T = np.float64
    A = np.array([[1.0, -2.0, 3.0, -4.0, 5.0],
                  [2.0, -1.0, 0.0, -1.0, 3.0],
                  [-1.0, 0.0, 4.0, -3.0, 2.0],
                  [-1.0, -1.0, -1.0, 1.0, 3.0]], dtype=T)
    b = np.array([1.0, 2.0, 3.0, 4.0], dtype=T)
    m, n = A.shape
    lam = T(0.1) * np.max(np.abs(A.T @ b))
    assert type(lam) == T
    f = bz.LeastSquares(A, b)
    g = bz.NormL1(lam)
    c = bz.IdentityFunction()
    D = bz.FreeSet()
    x_star = np.array([-3.877278911564627e-01, 0, 0, 2.174149659863943e-02, 6.168435374149660e-01], dtype=T)
    TOL = 1e-4
    x0 = np.zeros(n, T)
    y0 = np.zeros(n, T)
    out = bz.alps(f, g, c, D, x0, y0, verbose=True, resident=resident)
    x, it, subit = out[0], out[2], out[3]
    assert x.dtype == T
    assert np.max(np.abs(x - x_star)) <= TOL
    assert it < 10
    assert subit < 50
    assert "initial inner tolerance" in capfd.readouterr().out      # the verbose branch ran


def _check_qp(bz, Q, q, low, upp, gamma, n):
    T = np.float64
    TOL = 1e-4
    c = bz.IdentityFunction()
    D = bz.ClosedSet(bz.IndBox(low, upp))
    f = bz.Quadratic(Q, q)
    for g in (bz.IndBox(low, upp), bz.IndFree()):
        for resident in (True, False):
            x0 = np.zeros(n, T)
            y0 = np.zeros(n, T)
            x0_backup = x0.copy()
            out = bz.alps(f, g, c, D, x0, y0, resident=resident)
            x = out[0]
            z = np.minimum(upp, np.maximum(low, x - gamma * (Q @ x + q)))
            assert np.max(np.abs(x - z)) / gamma <= TOL
            assert np.array_equal(x0, x0_backup)


def test_nonconvex_qp_tiny(bz):
    Q = np.diag([-0.5, 1.0])
    q = np.array([0.3, 0.5])
    Lip = np.max(np.diag(Q))
    _check_qp(bz, Q, q, -1.0, 1.0, 0.95 / Lip, 2)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
def test_nonconvex_qp_small(bz, k):
    rng = np.random.default_rng(k)       # Julia's Random.seed!(k) stream is not reproducible here
    n = 100
    A = rng.standard_normal((n, n))
    U, _ = np.linalg.qr(A)
    eigenvalues = 2.0 * rng.random(n) - 1.0
    Q = U @ np.diag(eigenvalues) @ U.T
    Q = 0.5 * (Q + Q.T)
    q = rng.standard_normal(n)
    Lip = np.max(np.abs(eigenvalues))
    _check_qp(bz, Q, q, -1.0, 1.0, 0.95 / Lip, n)


def test_dense_f_gradients_match_oracle(bz, ref):
    rng = np.random.default_rng(0)
    for (m, n) in ((4, 5), (33, 17), (64, 128)):
        A, b = rng.standard_normal((m, n)), rng.standard_normal(m)
        Q = rng.standard_normal((n, n))
        Q = 0.5 * (Q + Q.T)
        q = rng.standard_normal(n)
        x, mu, y = rng.standard_normal(n), rng.uniform(0.1, 1, n), rng.standard_normal(n)
        for fd, fr in ((bz.LeastSquares(A, b), ref.LeastSquares(A, b)), (bz.Quadratic(Q, q), ref.Quadratic(Q, q))):
            prob = bz.Problem(fd, bz.NormL1(0.3), bz.IdentityFunction(), bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n, n, np.float64)
            prob.set_multipliers(mu, y)
            g_dev, vals = prob.eval_al_gradient(x)
            al = ref.AugLagFun(fr, ref.IdentityFunction(), ref.ClosedSet(ref.IndBox(-1.0, 1.0)), mu.copy(), y.copy(), x)
            g_ref = np.empty(n)
            lx = al.gradient(g_ref, x)
            assert np.max(np.abs(g_dev - g_ref)) <= 1e-12 * max(1.0, np.max(np.abs(g_ref)))
            assert abs(vals[0] - lx) <= 1e-12 * max(1.0, abs(lx))
            assert abs(vals[1] - al.fx) <= 1e-12 * max(1.0, abs(al.fx))
            prob.close()


def test_plain_c_caller_of_the_abi(bz):
    """examples/panoc_capi.c: the boundary used from plain C — compiled here with gcc against
    include/bazinga_hip.h, linked to the shared library, run as its own process.  It must solve cfg 2 with
    the same outer / inner iteration counts as the Python harness does on the same splitmix64 data."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "panoc_capi")
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "panoc_capi"], check=True, capture_output=True)
    n = 200_000
    r = subprocess.run([exe, str(n), "50", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"alps: status (\d+)\s+outer (\d+)\s+inner (\d+)", r.stdout)
    assert m and "iterations/s" in r.stdout, r.stdout
    d = bz.synth.l1_quadratic(n)
    out = bz.alps(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                  bz.ClosedSet(bz.IndBox(-1.0, 1.0)), np.zeros(n), np.zeros(n),
                  subsolver=lambda **kw: bz.PANOCplus(directions=bz.LBFGS(5, compact=True), **kw), resident=True)
    assert int(m.group(1)) == 0 and out[5] == "first_order"
    assert int(m.group(2)) == out[2] and int(m.group(3)) == out[3]
