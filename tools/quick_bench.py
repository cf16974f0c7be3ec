"""Quick single-GPU timing of the cfg-2 PANOCplus inner iteration (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bazinga_jl_amd as bz

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fuse = int(sys.argv[3]) if len(sys.argv) > 3 else 1
persist = int(sys.argv[4]) if len(sys.argv) > 4 else 1
compact = int(sys.argv[5]) if len(sys.argv) > 5 else 0
d = bz.synth.l1_quadratic(n)
# QB_XCHG=p2p|rccl: a 1-rank context that still runs every scalar exchange (self-loop mailboxes / 1-rank
# communicator): the per-iteration cost of the multi-GPU code path minus the xGMI hop
ctx = None
if os.environ.get("QB_XCHG") == "p2p":
    ctx = bz.Context(device=0, rank=0, nranks=1)
    ctx.p2p_connect([ctx.p2p_export()], [0])
elif os.environ.get("QB_XCHG") == "p2p_unused":      # mailbox allocated and mapped, problem on the default context
    _c2 = bz.Context(device=0, rank=0, nranks=1)
    _c2.p2p_connect([_c2.p2p_export()], [0])
elif os.environ.get("QB_XCHG") == "rccl":
    ctx = bz.Context(device=0, rank=0, nranks=1, comm_id=bz.Context.unique_id())
prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                  bz.ClosedSet(bz.IndBox(d["lo"], d["hi"])), n, n, np.float64, ctx)
print(prob.ctx.info())
mu = np.full(n, 0.1); y = np.zeros(n)
prob.set_multipliers(mu, y)
opts = bz.PANOCplus(tol=0.0, maxit=10**9, minimum_gamma=np.finfo(float).eps, fuse=bool(fuse), persist=bool(persist),
                    directions=bz.LBFGS(5, compact=bool(compact))).c_opts()
prob.panoc_begin(opts, np.zeros(n))
for _ in range(20):
    prob.panoc_step()
t0 = time.perf_counter()
if os.environ.get("QB_BULK") == "1":
    prob.panoc_steps(steps)
else:
    for _ in range(steps):
        prob.panoc_step()
t1 = time.perf_counter()
sc = prob.panoc_scalars(); st = prob.panoc_stats()
its = steps / (t1 - t0)
print(f"n={n} fuse={fuse} persist={persist} compact={compact}: {its:.1f} it/s  ({1e6/its:.1f} us/it)  model 520n B/it -> {520*n*its/1e12:.3f} TB/s = {520*n*its/8e12:.3f} of 8 TB/s")
print("scalars", sc)
print("stats fused", st.n_fused_iters, "grad", st.n_grad, "prox", st.n_prox, "bt", st.n_backtracks, "halv", st.n_gamma_halvings, "skips", st.n_lbfgs_skips)
prob.profile_enable(True); prob.profile_reset()
for _ in range(50):
    prob.panoc_step()
p = prob.profile()
for k, v in p.items():
    if v["launches"]:
        print(f"  {k:20s} launches/it={v['launches']/50:5.1f}  avg={1e3*v['total_ms']/v['launches']:8.2f} us  per-it={1e3*v['total_ms']/50:8.1f} us")
