"""N > 1 path on CPU: world_size-2 gloo.

The multi-GPU design shards x in contiguous blocks and exchanges ONLY the
reductions' partial scalars: every rank all-gathers the per-rank partials and folds them in rank
order, so all ranks hold bit-identical scalars and take identical line-search decisions.  This
test runs exactly that protocol over gloo with the CPU oracle standing in for the per-shard
kernels, and checks (a) ranks never diverge and (b) the sharded solve equals the unsharded one.
The same `shard_bounds` plan and fold-in-rank-order rule are what libbazinga_hip's k_pack +
ncclAllGather + fold_src implement on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class GlooReducer:
    """Partial scalar per rank -> all_gather -> fold in rank order (deterministic, identical on all ranks)."""

    def __init__(self):
        self.world = dist.get_world_size()
        self.calls = 0

    def _fold(self, local, op):
        self.calls += 1
        t = torch.tensor([float(local)], dtype=torch.float64)
        out = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        dist.all_gather(out, t)
        vals = [float(o.item()) for o in out]
        acc = vals[0]
        for v in vals[1:]:
            acc = op(acc, v)
        return np.float64(acc)

    def sum(self, v):
        return self._fold(np.sum(v), lambda a, b: a + b)

    def dot(self, a, b):
        return self._fold(np.dot(a, b), lambda a, b: a + b)

    def max(self, v):
        return self._fold(np.max(v) if v.size else 0.0, max)

    def any(self, m):
        return bool(self._fold(float(np.any(m)), max))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q, compact=False):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bazinga_jl_amd as bz
    from oracle import bazinga_ref as R
    lo, hi = bz.shard_bounds(n, rank, world, align=16)
    d = bz.synth.l1_quadratic(hi - lo, start=lo)
    red = GlooReducer()
    R.set_reducer(red)
    orc = (R.DiagQuadratic(d["q"], d["b"]), R.NormL1(d["lam"]), R.IdentityFunction(),
           R.ClosedSet(R.IndBox(d["lo"], d["hi"])))
    sub = lambda **kw: R.PANOCplus(directions=R.LBFGS(5, compact=compact), **kw)
    out = R.alps(*orc, np.zeros(hi - lo), np.zeros(hi - lo), subsolver=sub)
    R.set_reducer(None)
    q.put((rank, lo, hi, out[0], out[1], out[2], out[3], out[5], float(out[7]), red.calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("compact", [False, True])
def test_sharded_equals_unsharded_world2(compact):
    """both evaluations of the L-BFGS operator: the two-loop recursion (2M + 1 sequential exchanges per iteration on
    the GPUs) and the compact representation, whose Gram products, p and w are all partial sums of the same pass —
    on the GPUs ONE pack of 32 scalars per iteration (k_fused_compact -> k_exchange_collect); here every scalar of
    that pack goes through the same gather-and-fold-in-rank-order rule."""
    import bazinga_jl_amd as bz
    from oracle import bazinga_ref as R
    n, world = 1000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, compact)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    d = bz.synth.l1_quadratic(n)
    ref = R.alps(R.DiagQuadratic(d["q"], d["b"]), R.NormL1(d["lam"]), R.IdentityFunction(),
                 R.ClosedSet(R.IndBox(d["lo"], d["hi"])), np.zeros(n), np.zeros(n),
                 subsolver=lambda **kw: R.PANOCplus(directions=R.LBFGS(5, compact=compact), **kw))
    # identical control flow on every rank and vs the unsharded run
    assert res[0][5:8] == res[1][5:8] == (ref[2], ref[3], ref[5])
    assert res[0][8] == res[1][8]                      # bit-identical primal residual on both ranks
    assert res[0][9] == res[1][9]                      # same number of collectives: no rank-local branches
    x = np.concatenate([r[3] for r in res])
    y = np.concatenate([r[4] for r in res])
    assert [(r[1], r[2]) for r in res] == [bz.shard_bounds(n, r, world, align=16) for r in range(world)]
    assert np.max(np.abs(x - ref[0])) <= 1e-10 * max(1.0, np.max(np.abs(ref[0])))
    assert np.max(np.abs(y - ref[1])) <= 1e-9 * max(1.0, np.max(np.abs(ref[1])))
