"""Randomised whole-ALPS runs with the pairwise sets (VC / CC / either-or / XOR, adjacent and split layout) and vector
bounds — the families the seeded sweep of the suite does not draw — against the oracle (development aid).  The sets are
nonconvex: device and oracle may settle in different local solutions once rounding separates them, so the hard checks are
status, feasibility of the device's point and an objective no worse than the oracle's by more than 1e-3 relative; count
and point agreement are reported.   python tests/stress/stress_pairs.py 0 120"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bazinga_jl_amd as bz
from oracle import bazinga_ref as ref

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = same = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(11000 + seed)
    n = 2 * int(rng.integers(1, 2500))
    q, b = rng.uniform(0.2, 5.0, n), rng.standard_normal(n) * 3
    kind = str(rng.choice(["vc", "cc", "eitheror", "xor", "boxvec"]))
    layout = str(rng.choice(["adjacent", "split"]))
    gk = str(rng.choice(["l1", "nonneg", "zero", "indboxvec"]))
    lam = float(rng.uniform(0.1, 2.0))
    glo, ghi = -rng.uniform(0.5, 2.0, n), rng.uniform(0.5, 2.0, n)
    g_d, g_r = {"l1": (bz.NormL1(lam), ref.NormL1(lam)), "nonneg": (bz.NormL1Nonneg(lam), ref.NormL1Nonneg(lam)),
                "zero": (bz.Zero(), ref.Zero()), "indboxvec": (bz.IndBox(glo, ghi), ref.IndBox(glo, ghi))}[gk]
    if kind == "boxvec":
        dlo, dhi = -rng.uniform(0.1, 1.0, n), rng.uniform(0.1, 1.0, n)
        D_d, D_r = bz.ClosedSet(bz.IndBox(dlo, dhi)), ref.ClosedSet(ref.IndBox(dlo, dhi))
    else:
        D_d, D_r = bz.PairwiseSet(kind, layout=layout), ref.PairwiseSet(kind, layout=layout)
    x0, y0 = rng.standard_normal(n) * 0.1, rng.standard_normal(n) * 0.1
    tag = f"seed {seed} n={n} g={gk} D={kind}" + ("" if kind == "boxvec" else f"/{layout}")
    t0 = time.time()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = ref.alps(ref.DiagQuadratic(q, b), g_r, ref.IdentityFunction(), D_r, x0, y0, maxit=60)
        a = bz.alps(bz.DiagQuadratic(q, b), g_d, bz.IdentityFunction(), D_d, x0, y0, maxit=60)
        obj = lambda x: float(0.5 * np.sum(q * x * x) - np.sum(b * x) + g_r(x))
        s = np.empty(n); D_r.proj(s, a[0])
        infeas = float(np.max(np.abs(a[0] - s)))
        oa, oo = obj(a[0]), obj(o[0])
        err = float(np.max(np.abs(a[0] - o[0])))
        ok = a[5] == o[5] and (a[5] != "first_order" or (infeas <= 1e-4 and oa <= oo + 1e-3 * max(1.0, abs(oo))))
        same += int(err <= 1e-4 * max(1.0, float(np.max(np.abs(o[0])))))
        print(tag, "|", a[5], a[2], a[3], "| oracle", o[5], o[2], o[3], "| err %.1e infeas %.1e obj %.6g / %.6g" % (err, infeas, oa, oo),
              "%.1f s" % (time.time() - t0), "" if ok else " <== FAIL", flush=True)
        bad += 0 if ok else 1
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(e)[:200], flush=True)
print("done", hi - lo, "seeds;", bad, "failures;", same, "with the oracle's point")
