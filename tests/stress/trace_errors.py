"""Development aid: print device-vs-oracle iterate errors per PANOCplus state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bazinga_jl_amd as bz
from oracle import bazinga_ref as ref
from tests.test_gpu_parity import make_cfg3, make_cfg2, run_traces

for (nx, ny) in ((16, 32), (200, 128)):
    d, n, dev, orc = make_cfg3(bz, ref, nx, ny)
    prob, st, rows = run_traces(bz, ref, dev, orc, n, np.full(n, 0.1), np.zeros(n), d["x0"].copy(), 40, minimum_gamma=2.3e-16)
    print("stencil", nx, ny, " ".join("%d:%.1e/%.1e" % (r[0], max(r[1], r[2]), r[8]) for r in rows))
d, dev, orc = make_cfg2(bz, ref, 200003)
prob, st, rows = run_traces(bz, ref, dev, orc, 200003, np.full(200003, 0.1), np.zeros(200003), np.zeros(200003), 60)
print("cfg2", " ".join("%d:%.1e/%.1e" % (r[0], max(r[1], r[2]), r[8]) for r in rows))
