"""Print the solver's scalars (full precision) after some iterations of cfg 2 — to diff two library builds bit for bit (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bazinga_jl_amd as bz
n = int(float(sys.argv[1])); iters = int(sys.argv[2])
d = bz.synth.l1_quadratic(n)
prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                  bz.ClosedSet(bz.IndBox(d["lo"], d["hi"])), n, n, np.float64)
rng = np.random.default_rng(5)
prob.set_multipliers(np.full(n, 0.1), rng.standard_normal(n) if len(sys.argv) > 3 else np.zeros(n))
prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10**9, minimum_gamma=np.finfo(float).eps).c_opts(), np.zeros(n))
for k in range(iters):
    prob.panoc_step()
    if k % 10 == 9:
        sc = prob.panoc_scalars()
        print(k, " ".join(f"{key}={float(sc[key]).hex()}" for key in ("gamma", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys", "lbfgs_H", "FBE")))
x = prob.panoc_vector("x")
print("xsum", float(np.sum(x)).hex(), float(np.sum(x * x)).hex())
