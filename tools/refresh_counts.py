"""whole-ALPS iteration counts of a small basis-pursuit problem against the affine refresh period (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bazinga_jl_amd as bz
for seed in (21, 22, 23):
    ny, n = 96, 640
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((ny, n)) / np.sqrt(ny)
    xt = np.where(rng.uniform(size=n) < 0.05, rng.choice([-1.0, 1.0], n), 0.0)
    b = A @ xt
    row = []
    for refresh in (0, 1, 4, 8, 16, 32, 64):
        o = bz.alps(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A, b), bz.ZeroSet(), np.zeros(n), np.zeros(ny),
                    subsolver=lambda **k: bz.PANOCplus(affine_refresh=refresh, **k), resident=True)
        row.append((refresh, o[2], o[3], o[5][:5], float(np.max(np.abs(A @ o[0] - b)))))
    print(seed, row)
