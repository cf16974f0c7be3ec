"""Randomised ALS parity (the slack form through k_fused_slack) over many seeds: device als against the oracle's als
(development aid, like stress_sweep.py):  python tests/stress/stress_als.py 0 100"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bazinga_jl_amd as bz
from oracle import bazinga_ref as ref

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(2, 3000)) * 2
    q, b = rng.uniform(0.2, 5.0, n), rng.standard_normal(n) * 4
    fk = rng.choice(["diag", "zero"], p=[0.8, 0.2])
    if fk == "zero":
        # (f = Zero in the slack form: grad F(xs + 1) = grad F(xs), the Lipschitz estimate is 0 and gamma infinite — the
        # reference itself ends with :exception there; not a parity case)
        continue
    gk = rng.choice(["l1", "nonneg", "l1box", "indbox", "zero"])
    Dk = rng.choice(["box", "free", "zero"]) if fk == "diag" else "box"
    lam, u = float(rng.uniform(0.1, 3.0)), rng.uniform(0.0, 1.5, n)
    f_d, f_r = (bz.DiagQuadratic(q, b), ref.DiagQuadratic(q, b)) if fk == "diag" else (bz.Zero(), ref.Zero())
    g_d, g_r = {"l1": (bz.NormL1(lam), ref.NormL1(lam)), "nonneg": (bz.NormL1Nonneg(lam), ref.NormL1Nonneg(lam)),
                "l1box": (bz.NormL1Box(lam, u=u), ref.NormL1Box(lam, u=u)),
                "indbox": (bz.IndBox(-0.7, 0.9), ref.IndBox(-0.7, 0.9)), "zero": (bz.Zero(), ref.Zero())}[gk]
    l, h = -float(rng.uniform(0.2, 1.0)), float(rng.uniform(0.2, 1.0))
    D_d, D_r = {"box": (bz.ClosedSet(bz.IndBox(l, h)), ref.ClosedSet(ref.IndBox(l, h))),
                "free": (bz.FreeSet(), ref.FreeSet()), "zero": (bz.ZeroSet(), ref.ZeroSet())}[Dk]
    x0, y0 = rng.standard_normal(n) * 0.1, rng.standard_normal(n) * 0.1
    tag = f"seed {seed} n={n} f={fk} g={gk} D={Dk}"
    import time as _t
    _t0 = _t.time()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = ref.als(f_r, g_r, ref.IdentityFunction(), D_r, x0, y0, maxit=40)
        a = bz.als(f_d, g_d, bz.IdentityFunction(), D_d, x0, y0, maxit=40)
        print(tag, "|", a[5], a[2], a[3], "| oracle", o[5], o[2], o[3], round(_t.time() - _t0, 1), "s", flush=True)
        scale = max(1.0, float(np.max(np.abs(o[0]))))
        ok = (a[5] == o[5] and abs(a[2] - o[2]) <= 1 and abs(a[3] - o[3]) <= max(3, 0.3 * o[3])
              and np.max(np.abs(a[0] - o[0])) <= 2e-5 * scale)
        if not ok:
            bad += 1
            print("FAIL", tag, a[5], o[5], a[2], o[2], a[3], o[3], float(np.max(np.abs(a[0] - o[0]))), flush=True)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(e)[:200], flush=True)
    if seed % 20 == 0:
        print("seed", seed, "failures so far", bad, flush=True)
print("done", hi - lo, "seeds;", bad, "failures")
