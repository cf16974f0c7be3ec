"""Where a whole bz_alps_solve of cfg 2 spends its time, by kernel category (development aid; device pointers)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bazinga_jl_amd as bz   # noqa: E402

n = 10_000_000
L = bz._lib
d = bz.synth.l1_quadratic(n)
prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(2.5), bz.IdentityFunction(), bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n, n, np.float64)
ao = L.AlpsOpts()
L.load().bz_alps_default_opts(C.byref(ao), L.BZ_F64)
po = bz.PANOCplus(tol=ao.inner_tol).c_opts()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
bufs = []
for _ in range(6):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n * 8) == 0
    hip.hipMemset(p, 0, n * 8)
    bufs.append(p)
hip.hipDeviceSynchronize()
st = L.AlpsStats()
for rep in range(3):
    if rep == 2:
        prob.profile_reset(); prob.profile_enable(True)
    t0 = time.perf_counter()
    L.check(L.load().bz_alps_solve(prob._h, C.byref(ao), C.byref(po), bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5], C.byref(st)))
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {1e3 * dt:.2f} ms, outer {st.tot_it}, inner {st.tot_inner_it}")
p = prob.profile2()
tot = 0.0
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["timed_ms"]):
    if v["launches"]:
        print(f"  {k:22s} launches {v['launches']:5d}  timed {v['timed_ms']:8.3f} ms  avg {1e3 * v['timed_ms'] / max(1, v['timed_launches']):8.1f} us  {v['form']}")
        tot += v["timed_ms"]
print(f"  kernel time {tot:.2f} ms of {1e3 * dt:.2f} ms (every launch evented: the wall time is inflated)")
prob.close()
