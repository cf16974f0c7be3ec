"""Two ranks on ONE GPU through bz_panoc_steps (gated pre-launch + p2p exchange) against the single-rank solve
(development aid: python tools/dbg_gate_p2p.py [reps] [n_total] [chunk])."""
import os
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHUNKS = 10


def worker(rank, world, n_total, chunk, conn):
    try:
        import bazinga_jl_amd as bz
        bz.runtime_tuning()
        indep = os.environ.get("DBG_INDEP") == "1"      # two independent single-rank solves sharing the GPU
        if indep:
            ctx = bz.Context(device=0)
            conn.send(b"")
            conn.recv()
            lo, hi = 0, n_total // world
        else:
            ctx = bz.Context(device=0, rank=rank, nranks=world, comm_id=None)
            conn.send(ctx.p2p_export())
            ctx.p2p_connect(conn.recv(), [0] * world)
            lo, hi = bz.shard_bounds(n_total, rank, world)
        d = bz.synth.l1_quadratic(hi - lo, start=lo)
        nl = hi - lo
        prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                          bz.ClosedSet(bz.IndBox(-1.0, 1.0)), nl, nl, np.float64, ctx)
        prob.set_multipliers(np.full(nl, 0.1), np.zeros(nl))
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(nl))
        trace = []
        t0 = time.perf_counter()
        for _ in range(CHUNKS):
            prob.panoc_steps(chunk)
            sc = prob.panoc_scalars()
            trace.append((sc["k"], sc["gamma"], sc["stop_norm"], sc["FBE"], sc["lbfgs_mem"]))
        dt = time.perf_counter() - t0
        st = prob.panoc_stats()
        out = (rank, prob.panoc_vector("x"), trace, (st.n_gated_launches, st.n_gate_aborts, st.n_grad, st.n_backtracks,
                                                     st.n_gamma_halvings, st.n_lbfgs_skips), dt)
        prob.close()
        ctx.close()
        conn.send(("ok", out))
    except Exception as e:      # noqa: BLE001
        conn.send(("error", repr(e)))


def run(world, n_total, chunk):
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(world)]
    procs = [mpc.Process(target=worker, args=(r, world, n_total, chunk, pipes[r][1])) for r in range(world)]
    for p in procs:
        p.start()
    handles = [pipes[r][0].recv() for r in range(world)]
    for r in range(world):
        pipes[r][0].send(handles)
    res = []
    for r in range(world):
        assert pipes[r][0].poll(300), "rank did not answer"
        status, payload = pipes[r][0].recv()
        assert status == "ok", payload
        res.append(payload)
    for p in procs:
        p.join(60)
    return sorted(res, key=lambda t: t[0])


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n_total = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    import bazinga_jl_amd as bz
    d = bz.synth.l1_quadratic(n_total)
    prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                      bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n_total, n_total, np.float64)
    prob.set_multipliers(np.full(n_total, 0.1), np.zeros(n_total))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(n_total))
    ref_trace = []
    for _ in range(CHUNKS):
        prob.panoc_steps(chunk)
        sc = prob.panoc_scalars()
        ref_trace.append((sc["k"], sc["gamma"], sc["stop_norm"], sc["FBE"], sc["lbfgs_mem"]))
    x1 = prob.panoc_vector("x")
    prob.close()
    bad = 0
    for rep in range(reps):
        res = run(2, n_total, chunk)
        x = np.concatenate([r[1] for r in res])
        err = float(np.max(np.abs(x - x1)) / np.max(np.abs(x1)))
        first = None
        for i, (a, b, c) in enumerate(zip(res[0][2], res[1][2], ref_trace)):
            if a != b or abs(a[2] - c[2]) > 1e-8 * max(1.0, c[2]) or abs(a[1] - c[1]) > 1e-12 * c[1]:
                first = (i, a, b, c)
                break
        ok = err <= 1e-10 and first is None
        bad += 0 if ok else 1
        print(f"rep {rep}: {'ok ' if ok else 'BAD'} err_x {err:.2e} stats r0 {res[0][3]} r1 {res[1][3]} "
              f"time {res[0][4] * 1e3:.1f} / {res[1][4] * 1e3:.1f} ms" + ("" if first is None else f"\n   first divergence at chunk {first[0]}: r0 {first[1]}\n      r1 {first[2]}\n      single {first[3]}"),
              flush=True)
    print("bad runs:", bad, "of", reps)


if __name__ == "__main__":
    main()
