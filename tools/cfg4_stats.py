"""cfg 4: where do the passes over A go? (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bazinga_jl_amd as bz
ny, n = 8192, 65536
d = bz.synth.basis_pursuit(ny, n, dtype=np.float32)
prob = bz.Problem(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(d["A"], d["b"]), bz.ZeroSet(), n, ny, np.float32)
prob.set_multipliers(np.full(ny, 0.1, np.float32), np.zeros(ny, np.float32))
for refresh in (8, 16, 32):
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 12, minimum_gamma=float(np.finfo(np.float32).eps), affine_refresh=refresh).c_opts(), np.zeros(n, np.float32))
    prob.panoc_steps(20)
    s0 = prob.panoc_stats()
    t0 = time.perf_counter()
    prob.panoc_steps(200)
    dt = time.perf_counter() - t0
    s1 = prob.panoc_stats()
    p = prob.profile2() if hasattr(prob, "profile2") else {}
    print(f"refresh {refresh}: {200 / dt:.1f} it/s; per iteration: AL gradients {(s1.n_grad - s0.n_grad) / 200:.3f}, from images "
          f"{(s1.n_affine_images - s0.n_affine_images) / 200:.3f}, backtracks {(s1.n_backtracks - s0.n_backtracks) / 200:.3f}, "
          f"gamma halvings {s1.n_gamma_halvings - s0.n_gamma_halvings}, stop norm {prob.panoc_scalars()['stop_norm']:.3e}")
prob.close()
