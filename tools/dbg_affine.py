"""development aid: affine images against the pass-over-A evaluation (small dense problem, whole ALPS)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bazinga_jl_amd as bz
ny, n = 20, 100
d = bz.synth.basis_pursuit(ny, n, dtype=np.float64, density=0.1)
dev = (bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(d["A"], d["b"]), bz.ZeroSet())
recs = {}
for refresh in (0, 8):
    rec = []
    def sub(**kw):
        inner = bz.PANOCplus(maxit=3000, minimum_gamma=2.3e-16, affine_refresh=refresh, **kw)
        def run(*, f, g, x0):
            sol, it = inner(f=f, g=g, x0=x0)
            st = inner.stats
            rec.append((it, st.gamma, st.n_gamma_halvings, st.n_backtracks, st.n_affine_images, st.stop_norm, float(np.max(np.abs(sol)))))
            return sol, it
        return run
    a = bz.alps(*dev, np.zeros(n), np.zeros(ny), subsolver=sub, subsolver_maxit=100000, resident=False, maxit=8)
    recs[refresh] = (rec, a)
    print("refresh", refresh, a[2:8])
    for r in rec:
        print("   ", r)
# first outer where they differ: replay that subproblem step by step
ra, rb = recs[0][0], recs[8][0]
k = next((i for i, (u, v) in enumerate(zip(ra, rb)) if u[0] != v[0] or abs(u[1] - v[1]) > 1e-12 * u[1]), None)
print("first differing outer:", k)
