"""Does a dense panel that fits the 256 MB Infinity Cache stream faster on re-read?  (NEXT.md idea 4)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bazinga_jl_amd as bz

n = 65536
for ny in (128, 256, 512, 768, 1024, 2048, 8192):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((ny, n), dtype=np.float32)
    b = rng.standard_normal(ny, dtype=np.float32)
    prob = bz.Problem(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A, b), bz.ZeroSet(), n, ny, np.float32)
    prob.set_multipliers(np.full(ny, 0.1, np.float32), np.zeros(ny, np.float32))
    x = rng.standard_normal(n, dtype=np.float32)
    for _ in range(5):
        prob.eval_al_gradient(x)
    prob.profile_reset(); prob.profile_enable(True)
    for _ in range(20):
        prob.eval_al_gradient(x)
    p = prob.profile()
    mb = ny * n * 4 / 1e6
    out = []
    for k in ("gemv", "k_gemv_t_mfma"):
        v = p[k]
        if v["launches"]:
            us = 1e3 * v["total_ms"] / v["launches"]
            out.append(f"{k}: {us:7.1f} us = {mb / us / 1e6 * 1e6 / 1e3:6.2f} TB/s")
    print(f"ny={ny:5d}  A={mb:7.1f} MB  " + "  ".join(out), flush=True)
    prob.close()
