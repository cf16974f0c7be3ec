import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import bazinga_jl_amd as bz
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import bazinga_ref as ref
def rel(a, b): return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))
for n in (1000, 20011, 200003):
    d = bz.synth.l1_quadratic(n)
    dev = (bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(), bz.ClosedSet(bz.IndBox(d["lo"], d["hi"])))
    orc = (ref.DiagQuadratic(d["q"], d["b"]), ref.NormL1(d["lam"]), ref.IdentityFunction(), ref.IndicatorSet(ref.IndBox(d["lo"], d["hi"])))
    rng = np.random.default_rng(5)
    mu, y, x0 = np.full(n, 0.1), rng.standard_normal(n), np.zeros(n)
    out = {}
    for compact in (False, True):
        prob = bz.Problem(*dev, n, n, np.float64)
        prob.set_multipliers(mu, y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, directions=bz.LBFGS(5, compact=compact)).c_opts(), x0)
        al2 = ref.AugLagFun(orc[0], orc[2], orc[3], mu.copy(), y.copy(), x0)
        it2 = ref.PANOCplusIteration(al2, ref.NonsmoothCostFun(orc[1]), x0)
        st2 = it2.init()
        errs = []
        for k in range(60):
            errs.append(max(rel(prob.panoc_vector("x"), st2.x), rel(prob.panoc_vector("z"), st2.z)))
            prob.panoc_step(); st2 = it2.step(st2)
        prob.close()
        print(n, "compact" if compact else "two-loop", "max err vs two-loop oracle over k<=10/20/30/60:",
              " ".join("%.2e" % max(errs[:m]) for m in (10, 20, 30, 60)), flush=True)
