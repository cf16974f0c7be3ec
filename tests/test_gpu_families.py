"""The one-pass iterate-history kernel (k_fused_compact<XR = 2>) for every element-wise oracle family (VERDICT r1,
item 2): each family has its own instantiation (kinds and parameter streams fixed at compile time,
bz_families_dk*.hip).  For every family
  (a) the iterate-history form, the stored-pair form and the generic kernel chain are the SAME arithmetic: with the
      grids pinned to one summation tree the iterates and scalars agree bit for bit through gamma halvings and
      tau backtracks;
  (b) the first 30 PANOCplus states follow the oracle (two-loop form, the reference's operation order) within the
      north-star tolerance 1e-10.
Oracle kinds: src/proxoperators/{normL1Box,normL1Nonneg,zero}.jl, ProximalOperators NormL1 / IndBox,
src/projections/{zeroSet,freeSet,indicatorSet,vanishingConstraints,complementarityConstraints,orConstraints}.jl."""
import os
import zlib

import numpy as np
import pytest

from tests.test_gpu_parity import RTOL_ITER, iter_tol, run_traces

pytestmark = pytest.mark.gpu

# (f, g, D) — every g kind and every D class at least once, f = Zero, vector bounds given on one side only
FAMILIES = [
    ("diag", "l1", "box"),              # the headline family (also through its family-table instantiation: BZ_FAMRT)
    ("diag", "l1box", "box"),
    ("diag", "nonneg", "box"),
    ("diag", "indbox", "box"),
    ("diag", "indbox_vec", "free"),
    ("diag", "l1", "boxvec"),
    ("diag", "zero", "boxvec_lo"),      # obstacle-style: lower bound a vector, upper bound +inf
    ("zero", "l1", "box"),
    ("zero", "indbox", "boxvec"),
    ("diag", "l1", "zero"),
    ("diag", "l1box", "free"),
    ("diag", "zero", "vc"),
    ("diag", "l1", "cc"),
    ("diag", "nonneg", "eitheror"),
    ("diag", "l1", "xor"),
]


def make_family(bz, ref, n, fam, dtype=np.float64):
    f, g, D = fam
    d = bz.synth.l1_quadratic(n, dtype=dtype)
    rng = np.random.default_rng(zlib.crc32("-".join(fam).encode()))      # (hash() of a str is salted per process)
    scale = 0.2 if D in ("vc", "cc", "eitheror", "xor") else 1.0
    out = []
    for m in (bz, ref):
        ff = m.DiagQuadratic(d["q"], (scale * d["b"]).astype(dtype)) if f == "diag" else m.Zero()
        if g == "l1":
            gg = m.NormL1(0.8)
        elif g == "nonneg":
            gg = m.NormL1Nonneg(0.8)
        elif g == "l1box":
            gg = m.NormL1Box(0.8, u=np.where(np.arange(n) % 5 == 0, 0.0, 0.75).astype(dtype))
        elif g == "indbox":
            gg = m.IndBox(dtype(-0.5), dtype(0.5)) if m is ref else m.IndBox(-0.5, 0.5)
        elif g == "indbox_vec":
            r2 = np.random.default_rng(5)
            gg = m.IndBox((-r2.uniform(0.2, 1.0, n)).astype(dtype), r2.uniform(0.2, 1.0, n).astype(dtype))
        else:
            gg = m.Zero()
        r3 = np.random.default_rng(6)
        lo, hi = (-r3.uniform(0.1, 1.0, n)).astype(dtype), r3.uniform(0.1, 1.0, n).astype(dtype)
        if D == "box":
            DD = m.ClosedSet(m.IndBox(dtype(-1.0), dtype(1.0))) if m is ref else m.ClosedSet(m.IndBox(-1.0, 1.0))
        elif D == "boxvec":
            DD = m.ClosedSet(m.IndBox(lo, hi))
        elif D == "boxvec_lo":
            DD = m.ClosedSet(m.IndBox(lo, np.inf))
        elif D == "free":
            DD = m.FreeSet()
        elif D == "zero":
            DD = m.ZeroSet()
        else:
            DD = m.PairwiseSet(D)
        out.append((ff, gg, m.IdentityFunction(), DD))
    mu = np.full(n, 0.1, dtype)
    # (f = Zero: multipliers large enough that c(x) + mu*y leaves D — otherwise the subproblem is min g(x), solved at 0
    # in two iterations, and there is nothing to compare)
    y = ((30.0 if f == "zero" else 1.0) * rng.standard_normal(n)).astype(dtype)
    x0 = (0.3 * rng.standard_normal(n)).astype(dtype)
    return out[0], out[1], mu, y, x0


def _run(bz, dev, n, mu, y, x0, iters, env, dtype=np.float64, fuse=True, compact=None):
    keys = ("BZ_XR", "BZ_UNI", "BZ_GFC", "BZ_GRID", "BZ_TRIALFUSE", "BZ_FAMRT", "BZ_SKIPZ", "BZ_NT")
    old = {k: os.environ.get(k) for k in keys}
    try:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        prob = bz.Problem(*dev, n, n, dtype)
        prob.set_multipliers(mu, y)
        eps = float(np.finfo(dtype).eps)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=eps, fuse=fuse,
                                      directions=bz.LBFGS(5, compact=compact)).c_opts(), x0)
        prob.profile_enable(True)
        for _ in range(iters):
            prob.panoc_step()
        st = prob.panoc_stats()
        p = prob.profile2()
        out = (prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_vector("res"), prob.panoc_scalars(),
               (st.n_backtracks, st.n_gamma_halvings, st.n_lbfgs_skips, st.n_fused_iters),
               p["k_fused_iterates"]["launches"], p["k_fused_iterates"]["form"])
        prob.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


SCALARS = ("gamma", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_H", "FBE")


# The family table holds TWO instantiations per family: default-policy and non-temporal streams (NT = 1), the latter
# selected by size (n * sizeof(T) * streams > 340e6: every `bench.py --family` line at n = 1e7 times it).  BZ_NT is read
# at every bz_panoc_begin, so the tests below run each family through both ("half of the family kernel table
# is never compared with anything").
NT_FORMS = ["by-size", "nt"]


@pytest.mark.parametrize("nt", NT_FORMS)
@pytest.mark.parametrize("fam", FAMILIES, ids=["-".join(f) for f in FAMILIES])
def test_family_forms_are_bitwise_neutral(bz, ref, fam, nt):
    n = 60_010
    dev, orc, mu, y, x0 = make_family(bz, ref, n, fam)
    iters = 70
    pin = {"BZ_GFC": "2", "BZ_GRID": "512", "BZ_TRIALFUSE": "0"}      # one summation tree for every form
    if nt == "nt":
        pin["BZ_NT"] = "1"
    base = _run(bz, dev, n, mu, y, x0, iters, dict(pin, BZ_XR="0"))
    # stored-pair form, fused throughout but for the iterations with a tau backtrack or a gamma halving
    assert base[5] == 0 and base[4][3] >= iters - 12 - base[4][0] - base[4][1]
    variants = [dict(pin, BZ_XR="2", BZ_UNI="0"), dict(pin, BZ_XR="2"), dict(pin, BZ_XR="2", BZ_TRIALFUSE="1"),
                dict(pin, BZ_XR="2", BZ_SKIPZ="0")]
    if fam == ("diag", "l1", "box"):
        variants.append(dict(pin, BZ_XR="2", BZ_FAMRT="1", BZ_TRIALFUSE="1"))
    for env in variants:
        r = _run(bz, dev, n, mu, y, x0, iters, env)
        # the iterate-history form really ran (a skipped pair or a rejected trial sends a few iterations elsewhere)
        assert r[5] >= max(4, iters - 12 - 6 * base[4][2] - 2 * base[4][0] - base[4][1]), (env, r[5], base[4])
        if fam != ("diag", "l1", "box") or env.get("BZ_FAMRT"):
            assert "FAM=" in r[6], r[6]                                     # ... in its family instantiation
        assert ("NT=1" in r[6]) == (nt == "nt"), r[6]                      # ... with the streams the case asks for
        for a, b in zip(r[:3], base[:3]):
            assert np.array_equal(a, b), env
        for key in SCALARS:
            assert r[3][key] == base[3][key], (env, key)
        assert r[4][:3] == base[4][:3]
    # uniform penalties passed as numbers, y = 0 (UNI = 2) against streaming them
    y0 = np.zeros(n)
    u0 = _run(bz, dev, n, mu, y0, x0, 40, dict(pin, BZ_XR="2", BZ_UNI="0"))
    u2 = _run(bz, dev, n, mu, y0, x0, 40, dict(pin, BZ_XR="2", BZ_UNI="2"))
    for a, b in zip(u0[:3], u2[:3]):      # (f = Zero with y = 0 can start on a flat piece: gamma = alpha / 0, NaN in both)
        assert np.array_equal(a, b, equal_nan=True)
    # the generic kernel chain (fuse = False) on the same grid: same values (p, w come from another kernel's sums)
    g = _run(bz, dev, n, mu, y, x0, 25, dict(pin), fuse=False, compact=True)
    f = _run(bz, dev, n, mu, y, x0, 25, dict(pin, BZ_XR="2"))
    assert g[4][3] == 0 and f[4][3] >= 20 - f[4][0] - f[4][1]
    assert np.max(np.abs(g[0] - f[0])) <= 1e-11 * max(1.0, np.max(np.abs(f[0])))
    assert np.max(np.abs(g[1] - f[1])) <= 1e-11 * max(1.0, np.max(np.abs(f[1])))


@pytest.mark.parametrize("nt", NT_FORMS)
@pytest.mark.parametrize("fam", FAMILIES, ids=["-".join(f) for f in FAMILIES])
def test_family_iterates_match_oracle(bz, ref, fam, nt, monkeypatch):
    n = 20_010
    dev, orc, mu, y, x0 = make_family(bz, ref, n, fam)
    if nt == "nt":
        monkeypatch.setenv("BZ_NT", "1")
    prob, st, rows = run_traces(bz, ref, dev, orc, n, mu, y, x0, 30, minimum_gamma=float(np.finfo(float).eps))
    form = prob.profile2()["k_fused_iterates"]["form"]
    assert form.startswith("k_fused_compact<XR=2") and ("NT=1" in form) == (nt == "nt"), form
    for k, ex, ez, g_d, g_r, sn_d, sn_r, fused, sens in rows:
        assert abs(g_d - g_r) <= 1e-13 * g_r, f"gamma differs at k={k}"
        tol = iter_tol(sens)
        assert ex <= tol and ez <= tol, f"iterate mismatch at k={k}: {ex} {ez} (tol {tol})"
        assert abs(sn_d - sn_r) <= 1e-8 * max(1.0, sn_r)
    assert sum(r[7] for r in rows) >= 20
    # (the tolerance above widens only where the oracle's own rounding sensitivity exceeds 1e-12 — the f = Zero families
    # collapse onto x = 0 within a few iterations, where a relative error means nothing; the states before that must
    # have been compared at the north-star tolerance)
    assert sum(1 for r in rows if iter_tol(r[8]) == RTOL_ITER) >= (5 if fam[0] == "zero" else 25)
    prob.close()


@pytest.mark.parametrize("fam", [("diag", "l1box", "box"), ("diag", "l1", "xor"), ("zero", "indbox_vec", "boxvec")],
                         ids=["l1box-box", "l1-xor", "indboxvec-boxvec"])
@pytest.mark.parametrize("nt", NT_FORMS)
def test_family_kernels_float32(bz, ref, fam, nt):
    """fp32 packs hold four elements (two pairs): same neutrality, ragged last chunk included (n % 4 == 2)."""
    n = 50_002
    dev, orc, mu, y, x0 = make_family(bz, ref, n, fam, dtype=np.float32)
    pin = {"BZ_GFC": "2", "BZ_GRID": "512", "BZ_TRIALFUSE": "0"}
    if nt == "nt":
        pin["BZ_NT"] = "1"
    base = _run(bz, dev, n, mu, y, x0, 40, dict(pin, BZ_XR="0"), dtype=np.float32)
    # the compile-time UNI / TRIAL instantiation (fp32 has them since r03), the run-time one, penalties streamed
    for env in (dict(pin, BZ_XR="2"), dict(pin, BZ_XR="2", BZ_FAMRT="1"), dict(pin, BZ_XR="2", BZ_UNI="0")):
        r = _run(bz, dev, n, mu, y, x0, 40, env, dtype=np.float32)
        assert base[5] == 0 and r[5] >= 5 and "FAM=" in r[6] and ("NT=1" in r[6]) == (nt == "nt")
        assert ("UNI=-1" in r[6]) == bool(env.get("BZ_FAMRT")), r[6]
        for a, b in zip(r[:3], base[:3]):
            assert np.array_equal(a, b), env
        for key in SCALARS:
            assert r[3][key] == base[3][key], (env, key)
