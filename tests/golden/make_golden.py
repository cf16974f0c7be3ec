"""Generates the committed golden fixtures under tests/golden/.

These are RESTATEMENT-GENERATED: the reference (Julia + un-vendored ProximalAlgorithms.jl) cannot
run in the build container, so the vectors come from oracle/bazinga_ref.py, which is itself pinned
on the reference's solution-level KATs (tests/test_oracle_kat.py).  Run:  python -m tests.golden.make_golden
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cfg2_problem(n):
    import bazinga_jl_amd as bz
    from oracle import bazinga_ref as R
    d = bz.synth.l1_quadratic(n)
    return d, (R.DiagQuadratic(d["q"], d["b"]), R.NormL1(d["lam"]), R.IdentityFunction(),
               R.ClosedSet(R.IndBox(d["lo"], d["hi"])))


def cfg2_trace(n, iters, compact=False):
    """First `iters` PANOCplus states on the cfg-2 problem with mu = 0.1, y = sin(i); compact: the L-BFGS
    operator in its compact representation (LBFGSCompactOperator)."""
    from oracle import bazinga_ref as R
    d, (f, g, c, D) = cfg2_problem(n)
    mu = np.full(n, 0.1)
    y = np.sin(np.arange(n, dtype=np.float64))
    x0 = np.zeros(n)
    al = R.AugLagFun(f, c, D, mu, y, x0)
    gF = R.NonsmoothCostFun(g)
    it = R.PANOCplusIteration(al, gF, x0, minimum_gamma=np.finfo(float).eps, directions=R.LBFGS(5, compact=compact))
    st = it.init()
    rows = []
    for k in range(1, iters + 1):
        rows.append({"k": k, "gamma": float(st.gamma), "tau": float(st.tau), "f_x": float(st.f_x),
                     "g_z": float(st.g_z), "stop_norm": float(it.stop_norm(st)),
                     "x": st.x.tolist(), "z": st.z.tolist()})
        if k < iters:
            st = it.step(st)
    return {"n": n, "iters": iters, "mu": 0.1, "y": "sin(i)", "compact": compact, "rows": rows}


def pairs_problem(n, kind):
    import bazinga_jl_amd as bz
    from oracle import bazinga_ref as R
    d = bz.synth.l1_quadratic(n)
    return d, (R.DiagQuadratic(d["q"], 0.2 * d["b"]), R.NormL1(0.3), R.IdentityFunction(), R.PairwiseSet(kind))


def pairs_alps(n, kind):
    """ALPS with D = the 2-element set `kind` over adjacent pairs (nonconvex)."""
    from oracle import bazinga_ref as R
    d, orc = pairs_problem(n, kind)
    out = R.alps(*orc, np.zeros(n), np.zeros(n))
    return {"n": n, "kind": kind, "x": out[0].tolist(), "y": out[1].tolist(), "tot_it": out[2],
            "tot_inner_it": out[3], "status": out[5]}


def cfg2_alps(n):
    from oracle import bazinga_ref as R
    d, orc = cfg2_problem(n)
    out = R.alps(*orc, np.zeros(n), np.zeros(n))
    return {"n": n, "x": out[0].tolist(), "y": out[1].tolist(), "tot_it": out[2], "tot_inner_it": out[3],
            "status": out[5], "norm_res_prim": float(out[7]), "mu": out[9].tolist()}


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "panoc_trace_cfg2_n64.json"), "w") as fh:
        json.dump(cfg2_trace(64, 25), fh)
    with open(os.path.join(here, "alps_cfg2_n256.json"), "w") as fh:
        json.dump(cfg2_alps(256), fh)
    with open(os.path.join(here, "panoc_trace_cfg2_n64_compact.json"), "w") as fh:
        json.dump(cfg2_trace(64, 25, compact=True), fh)
    with open(os.path.join(here, "alps_pairs_n128.json"), "w") as fh:
        json.dump({k: pairs_alps(128, k) for k in ("vc", "cc", "eitheror", "xor")}, fh)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
