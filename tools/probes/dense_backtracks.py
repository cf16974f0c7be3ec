"""How many tau backtracks / gamma halvings / image verifications the dense basis-pursuit solves take (development aid)."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bazinga_jl_amd as bz   # noqa: E402

for ny, n, dtype in ((64, 512, np.float64), (257, 1028, np.float64), (512, 4096, np.float64), (512, 4096, np.float32)):
    d = bz.synth.basis_pursuit(ny, n, dtype=dtype, density=0.05)
    dev = (bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(d["A"], d["b"]), bz.ZeroSet())
    prob = bz.Problem(*dev, n, ny, dtype)
    prob.set_multipliers(np.full(ny, 0.1, dtype), (0.1 * np.random.default_rng(2).standard_normal(ny)).astype(dtype))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, minimum_gamma=float(np.finfo(dtype).eps), affine_refresh=16).c_opts(), np.zeros(n, dtype))
    seen = []
    for k in range(1, 201):
        prob.panoc_step()
        st = prob.panoc_stats()
        if k in (30, 60, 100, 150, 200):
            seen.append((k, st.n_backtracks, st.n_gamma_halvings, st.n_affine_images, prob.panoc_scalars()["stop_norm"]))
    print(ny, n, dtype.__name__, seen)
    prob.close()
