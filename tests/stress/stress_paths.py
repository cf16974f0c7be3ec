"""Randomised whole-ALPS parity on the two non-separable paths — the 5-point-stencil f (cfg 3 family: compact form on
the stencil kernels) and the dense affine constraint (cfg 4 family: affine images) — over many seeds, against the oracle
(development aid):  python tests/stress/stress_paths.py 0 60"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bazinga_jl_amd as bz
from oracle import bazinga_ref as ref

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
eps = float(np.finfo(float).eps)
for seed in range(lo, hi):
    rng = np.random.default_rng(9000 + seed)
    t0 = time.time()
    try:
        if seed % 2 == 0:
            nx, ny = int(rng.integers(3, 40)), 2 * int(rng.integers(2, 20))
            d = bz.synth.obstacle_grid(nx, ny, load=float(rng.choice([-1.0, 1.0, -3.0])))
            n = nx * ny
            dev = (bz.Stencil5ptQuadratic(nx, ny, d["b"]), bz.Zero(), bz.IdentityFunction(), bz.ClosedSet(bz.IndBox(d["psi"], np.inf)))
            orc = (ref.Stencil5ptQuadratic(nx, ny, d["b"]), ref.Zero(), ref.IdentityFunction(), ref.ClosedSet(ref.IndBox(d["psi"], np.inf)))
            x0, y0 = d["x0"].copy(), np.zeros(n)
            sub = lambda R: (lambda **k: R.PANOCplus(maxit=20000, minimum_gamma=eps, **k))
            tag = f"seed {seed} stencil {nx}x{ny}"
            tolx = 2e-4
        else:
            ny, n = int(rng.integers(4, 48)), int(rng.integers(50, 400))
            A = rng.standard_normal((ny, n)) / np.sqrt(ny)
            xt = np.where(rng.uniform(size=n) < 0.06, rng.choice([-1.0, 1.0], n), 0.0)
            b = A @ xt
            dev = (bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A, b), bz.ZeroSet())
            orc = (ref.Zero(), ref.NormL1(1.0), ref.DenseAffine(A, b), ref.ZeroSet())
            x0, y0 = np.zeros(n), np.zeros(ny)
            sub = lambda R: (lambda **k: R.PANOCplus(maxit=20000, minimum_gamma=eps, **k))
            tag = f"seed {seed} dense {ny}x{n}"
            tolx = 2e-3
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = ref.alps(*orc, x0, y0, subsolver=sub(ref))
        a = bz.alps(*dev, x0, y0, subsolver=sub(bz), resident=True)
        scale = max(1.0, float(np.max(np.abs(o[0]))))
        err = float(np.max(np.abs(a[0] - o[0])))
        ok = a[5] == o[5] and abs(a[2] - o[2]) <= 5 and err <= tolx * scale
        if seed % 2 == 1 and a[5] == "first_order":
            ok = ok and float(np.max(np.abs(A @ a[0] - b))) <= 1e-4
        print(tag, "|", a[5], a[2], a[3], "| oracle", o[5], o[2], o[3], "| err %.2e" % err, "%.1f s" % (time.time() - t0),
              "" if ok else " <== FAIL", flush=True)
        bad += 0 if ok else 1
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("ERROR", seed, repr(e)[:200], flush=True)
print("done", hi - lo, "seeds;", bad, "failures")
