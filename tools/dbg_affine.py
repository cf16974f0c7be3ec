"""development aid: affine images against the pass-over-A evaluation, state by state (small dense problem)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bazinga_jl_amd as bz
ny, n = 20, 100
d = bz.synth.basis_pursuit(ny, n, dtype=np.float64, density=0.1)
dev = (bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(d["A"], d["b"]), bz.ZeroSet())
for refresh in (0, 8):
    prob = bz.Problem(*dev, n, ny, np.float64)
    prob.set_multipliers(np.full(ny, 0.1), np.zeros(ny))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=2.3e-16, affine_refresh=refresh).c_opts(), np.zeros(n))
    for k in range(40):
        sc = prob.panoc_scalars()
        print(refresh, k, "gamma %.6e f_x %.12e g_z %.6e stop %.6e mem %d tau %.3f" % (sc["gamma"], sc["f_x"], sc["g_z"], sc["stop_norm"], sc["lbfgs_mem"], sc["tau"]), flush=True)
        if not np.isfinite(sc["f_x"]):
            break
        prob.panoc_step()
    prob.close()
sub = lambda **kw: bz.PANOCplus(maxit=100000, minimum_gamma=2.3e-16, **kw)
a = bz.alps(*dev, np.zeros(n), np.zeros(ny), subsolver=sub, subsolver_maxit=100000, verbose=True)
print(a[2:8])
