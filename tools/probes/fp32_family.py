"""fp32 family pass (l1box-box, n = 2e7), compile-time UNI instantiation against the run-time one (BZ_FAMRT=1): development aid."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bazinga_jl_amd as bz   # noqa: E402

n = 20_000_000
d = bz.synth.l1_quadratic(n)
q, b = d["q"].astype(np.float32), d["b"].astype(np.float32)
u = np.full(n, 0.8, np.float32)
for famrt in ("0", "1"):
    os.environ["BZ_FAMRT"] = famrt
    prob = bz.Problem(bz.DiagQuadratic(q, b), bz.NormL1Box(2.5, u=u), bz.IdentityFunction(),
                      bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n, n, np.float32)
    prob.set_multipliers(np.full(n, 0.1, np.float32), np.zeros(n, np.float32))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=1e-7).c_opts(), np.zeros(n, np.float32))
    prob.panoc_steps(30)
    prob.profile_reset(); prob.profile_enable(True, period=8)
    t0 = time.perf_counter()
    prob.panoc_steps(200)
    dt = time.perf_counter() - t0
    p = prob.profile2()["k_fused_iterates"]
    print(f"BZ_FAMRT={famrt}: {200 / dt:.0f} it/s; {p['form']} {1e3 * p['timed_ms'] / max(1, p['timed_launches']):.1f} us")
    prob.close()
