import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bazinga_jl_amd as bz
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
d = bz.synth.l1_quadratic(n)
for rep in range(2):
    t0 = time.perf_counter()
    out = bz.alps(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                  bz.ClosedSet(bz.IndBox(d["lo"], d["hi"])), np.zeros(n), np.zeros(n), tol=1e-6)
    t1 = time.perf_counter()
    x, y, tot_it, tot_inner, elapsed, status = out[:6]
    print(f"rep {rep}: status {status} outer {tot_it} inner {tot_inner} elapsed(lib) {elapsed:.4f}s wall {t1-t0:.4f}s -> {tot_inner/elapsed:.0f} inner it/s")
