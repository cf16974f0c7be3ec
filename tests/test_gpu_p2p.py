"""Multi-rank path on ONE GPU: two processes share device 0, x is sharded between them and every global
scalar travels through the peer-to-peer mailboxes (HIP IPC + system-scope stores), including the phases
INSIDE the persistent two-loop kernel (each rank runs it on half of the CUs so both grids are
co-resident).  Exercises everything of the N > 1 design except a physical xGMI hop; the result must
match the single-rank solve of the unsharded problem."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL = 800_000
ITERS = 14


def _worker(rank, world, n_total, persist, conn, compact=False):
    try:
        os.environ["BZ_PERSIST_BLOCKS"] = "96"
        sys.path.insert(0, ROOT)
        import bazinga_jl_amd as bz
        ctx = bz.Context(device=0, rank=rank, nranks=world, comm_id=None)
        conn.send(ctx.p2p_export())
        handles = conn.recv()
        ctx.p2p_connect(handles, [0] * world)
        lo, hi = bz.shard_bounds(n_total, rank, world)
        d = bz.synth.l1_quadratic(hi - lo, start=lo)
        nl = hi - lo
        prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                          bz.ClosedSet(bz.IndBox(-1.0, 1.0)), nl, nl, np.float64, ctx)
        y = np.sin(np.arange(lo, hi, dtype=np.float64))
        prob.set_multipliers(np.full(nl, 0.1), y)
        prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9, persist=persist,
                                      directions=bz.LBFGS(5, compact=compact)).c_opts(), np.zeros(nl))
        prob.profile_enable(True)
        for _ in range(ITERS):
            prob.panoc_step()
        prof = prob.profile()
        out = (rank, lo, hi, prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars(),
               prof["k_twoloop_persist"]["launches"], prof["all_gather"]["launches"])
        prob.close()
        ctx.close()
        conn.send(("ok", out))
    except Exception as e:      # noqa: BLE001
        conn.send(("error", repr(e)))


def _run_sharded(world, persist, compact=False, n_total=N_TOTAL):
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(world)]
    procs = [mpc.Process(target=_worker, args=(r, world, n_total, persist, pipes[r][1], compact)) for r in range(world)]
    for p in procs:
        p.start()
    handles = [pipes[r][0].recv() for r in range(world)]
    for r in range(world):
        pipes[r][0].send(handles)
    res = []
    for r in range(world):
        assert pipes[r][0].poll(240), "rank did not answer"
        status, payload = pipes[r][0].recv()
        assert status == "ok", payload
        res.append(payload)
    for p in procs:
        p.join(60)
    return sorted(res)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("persist,compact", [(False, False), (True, False), (False, True)])
def test_two_ranks_on_one_gpu_match_single_rank(bz, persist, compact):
    res = _run_sharded(2, persist, compact)
    d = bz.synth.l1_quadratic(N_TOTAL)
    prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                      bz.ClosedSet(bz.IndBox(-1.0, 1.0)), N_TOTAL, N_TOTAL, np.float64)
    prob.set_multipliers(np.full(N_TOTAL, 0.1), np.sin(np.arange(N_TOTAL, dtype=np.float64)))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(N_TOTAL))
    for _ in range(ITERS):
        prob.panoc_step()
    x1, z1, s1 = prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars()
    prob.close()
    x = np.concatenate([r[3] for r in res])
    z = np.concatenate([r[4] for r in res])
    sa, sb = res[0][5], res[1][5]
    for key in ("gamma", "f_x", "g_z", "stop_norm", "last_ys", "lbfgs_H", "FBE"):
        assert sa[key] == sb[key], f"ranks disagree on {key}"          # bit-identical scalars on every rank
    assert sa["gamma"] == s1["gamma"] and sa["lbfgs_mem"] == 5.0
    assert np.max(np.abs(x - x1)) <= 1e-10 * np.max(np.abs(x1))
    assert np.max(np.abs(z - z1)) <= 1e-10 * np.max(np.abs(z1))
    assert abs(sa["stop_norm"] - s1["stop_norm"]) <= 1e-8 * max(1.0, s1["stop_norm"])
    if persist:
        assert all(r[6] >= ITERS - 2 for r in res)      # the persistent kernel really ran sharded
    elif compact:
        assert all(r[6] == 0 and ITERS <= r[7] <= 2 * ITERS + 12 for r in res)   # ~1 exchange per iteration
    else:
        assert all(r[6] == 0 and r[7] > 5 * ITERS for r in res)


def _worker_steps(rank, world, n_total, conn):
    """as _worker, through bz_panoc_steps (the library runs the loop: the path bench.py times)"""
    try:
        sys.path.insert(0, ROOT)
        import bazinga_jl_amd as bz
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
        for chunk in (3, 20, 17):
            prob.panoc_steps(chunk)
        st = prob.panoc_stats()
        out = (rank, prob.panoc_vector("x"), prob.panoc_scalars(), (st.n_gated_launches, st.n_gate_fallbacks, st.n_fused_iters))
        prob.close()
        ctx.close()
        conn.send(("ok", out))
    except Exception as e:      # noqa: BLE001
        conn.send(("error", repr(e)))


@pytest.mark.timeout(600)
def test_library_loop_two_ranks_on_one_gpu(bz):
    """bz_panoc_steps with x sharded over two ranks that SHARE the GPU: the gated pre-launch must stay off there (a
    resident launch polling at its gate holds its CUs, and two tenants doing that starve each other: before the contexts
    knew that they share a device this configuration ran into the gate's poll bounds and, worse, went on with partly
    executed passes), and the iterates are those of the single-rank solve."""
    mpc = mp.get_context("spawn")
    n_total = 1_000_000
    pipes = [mpc.Pipe() for _ in range(2)]
    procs = [mpc.Process(target=_worker_steps, args=(r, 2, n_total, pipes[r][1])) for r in range(2)]
    for p in procs:
        p.start()
    handles = [pipes[r][0].recv() for r in range(2)]
    for r in range(2):
        pipes[r][0].send(handles)
    res = []
    for r in range(2):
        assert pipes[r][0].poll(240), "rank did not answer"
        status, payload = pipes[r][0].recv()
        assert status == "ok", payload
        res.append(payload)
    for p in procs:
        p.join(60)
    res.sort(key=lambda t: t[0])
    d = bz.synth.l1_quadratic(n_total)
    prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                      bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n_total, n_total, np.float64)
    prob.set_multipliers(np.full(n_total, 0.1), np.zeros(n_total))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(n_total))
    prob.panoc_steps(40)
    x1, s1, st1 = prob.panoc_vector("x"), prob.panoc_scalars(), prob.panoc_stats()
    prob.close()
    assert st1.n_gated_launches >= 30                   # one tenant: the gate is in use
    x = np.concatenate([r[1] for r in res])
    assert np.max(np.abs(x - x1)) <= 1e-10 * np.max(np.abs(x1))
    for key in ("gamma", "f_x", "stop_norm", "FBE"):
        assert res[0][2][key] == res[1][2][key]
    assert abs(res[0][2]["stop_norm"] - s1["stop_norm"]) <= 1e-8 * max(1.0, s1["stop_norm"])
    assert all(r[3][0] == 0 and r[3][1] == 0 and r[3][2] >= 38 for r in res)


@pytest.mark.timeout(600)
def test_unequal_shards_take_the_same_two_loop_form(bz):
    """Shards that straddle the persistent kernel's size threshold (300 032 and 299 967 elements around
    BZ_PERSIST_MIN_N = 300 000): the ranks must agree on ONE form of the two-loop — the persistent kernel's
    in-kernel phase exchanges and the kernel chain's mailbox exchanges do not talk to each other — so the
    decision is exchanged at bz_panoc_begin and the chain is taken by both."""
    n_total = 599_999
    res = _run_sharded(2, True, False, n_total)
    assert [r[2] - r[1] for r in res] == [300_032, 299_967]
    assert all(r[6] == 0 for r in res), "one rank ran the persistent kernel while the other could not"
    d = bz.synth.l1_quadratic(n_total)
    prob = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                      bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n_total, n_total, np.float64)
    prob.set_multipliers(np.full(n_total, 0.1), np.sin(np.arange(n_total, dtype=np.float64)))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(n_total))
    for _ in range(ITERS):
        prob.panoc_step()
    x1 = prob.panoc_vector("x")
    prob.close()
    x = np.concatenate([r[3] for r in res])
    assert np.max(np.abs(x - x1)) <= 1e-10 * np.max(np.abs(x1))
    for key in ("gamma", "f_x", "stop_norm", "FBE"):
        assert res[0][5][key] == res[1][5][key]


@pytest.mark.gpu
def test_bench_launcher_two_ranks_share_one_gpu():
    """The driver's N > 1 launch line (torch.distributed.run, one process per rank) end to end on a one-GPU
    box: both ranks on device 0, x sharded, p2p mailboxes through HIP IPC, TCP rendezvous, max-over-ranks
    timing, one JSON line from rank 0.  RCCL refuses two ranks on one device, hence --no-rccl (the p2p
    scalars are then checked for rank agreement instead of against RCCL)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BZ_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "40",
           "--warmup", "10", "--size", "2e6", "--no-rccl"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["scalar_transport"] == "p2p" and d["config"]["n_per_gpu"] * 2 >= 2_000_000
    assert d["roofline"]["kernel"].startswith("bz::k_fused_compact") and d["cpu_baseline"] is None
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and 0.0 < d["roofline_iteration"]["frac"] <= d["roofline"]["frac"]
    assert d["solver"]["fused_iterations"] >= 38


# ---------------------------------------------------------------- row-block-sharded 5-point stencil (cfg 3)
GRID = (96, 128)         # rows x columns; rank r owns a block of rows


def _row_blocks(nrows, world):
    cuts = [round(r * nrows / world) for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _stencil_worker(rank, world, conn, mode):
    try:
        sys.path.insert(0, ROOT)
        import bazinga_jl_amd as bz
        nxg, nyg = GRID
        d = bz.synth.obstacle_grid(nxg, nyg, load=-1.0)
        r0, r1 = _row_blocks(nxg, world)[rank]
        sl = slice(r0 * nyg, r1 * nyg)
        nl = (r1 - r0) * nyg
        ctx = bz.Context(device=0, rank=rank, nranks=world, comm_id=None)
        conn.send(ctx.p2p_export())
        ctx.p2p_connect(conn.recv(), [0] * world)
        f = bz.Stencil5ptQuadratic(r1 - r0, nyg, d["b"][sl])
        D = bz.ClosedSet(bz.IndBox(d["psi"][sl], np.inf))
        prob = bz.Problem(f, bz.Zero(), bz.IdentityFunction(), D, nl, nl, np.float64, ctx)
        conn.send(prob.halo_export())
        hs = conn.recv()
        prob.halo_connect(hs[rank - 1] if rank > 0 else None, hs[rank + 1] if rank + 1 < world else None)
        if mode == "panoc":
            y = np.cos(np.arange(r0 * nyg, r1 * nyg, dtype=np.float64))
            prob.set_multipliers(np.full(nl, 0.05), y)
            prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), d["x0"][sl])
            for _ in range(ITERS):
                prob.panoc_step()
            out = (rank, prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars())
            prob.close()
        else:
            r = bz.alps(f, bz.Zero(), bz.IdentityFunction(), D, d["x0"][sl], np.zeros(nl), ctx=ctx, problem=prob)
            out = (rank, r[0], r[1], (r[2], r[3], r[5]))
        ctx.close()
        conn.send(("ok", out))
    except Exception as e:      # noqa: BLE001
        import traceback
        conn.send(("error", repr(e) + traceback.format_exc()[-1500:]))


def _run_stencil(world, mode):
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(world)]
    procs = [mpc.Process(target=_stencil_worker, args=(r, world, pipes[r][1], mode)) for r in range(world)]
    for p in procs:
        p.start()
    for _round in range(2):                      # mailbox handles, then halo handles
        hs = [pipes[r][0].recv() for r in range(world)]
        for r in range(world):
            pipes[r][0].send(hs)
    res = []
    for r in range(world):
        assert pipes[r][0].poll(240), "rank did not answer"
        status, payload = pipes[r][0].recv()
        assert status == "ok", payload
        res.append(payload)
    for p in procs:
        p.join(60)
    return sorted(res)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stencil_matches_single_rank(bz, world):
    """SURVEY §8(e)/(f-4), cfg 3 sharded: the grid cut into row blocks, boundary rows exchanged through
    IPC-mapped halo buffers before every stencil evaluation.  2 and 3 ranks (the middle rank has two
    neighbours) on one GPU against the unsharded solve: same iterates to the north-star tolerance."""
    nxg, nyg = GRID
    n = nxg * nyg
    d = bz.synth.obstacle_grid(nxg, nyg, load=-1.0)
    dev = (bz.Stencil5ptQuadratic(nxg, nyg, d["b"]), bz.Zero(), bz.IdentityFunction(),
           bz.ClosedSet(bz.IndBox(d["psi"], np.inf)))
    prob = bz.Problem(*dev, n, n, np.float64)
    prob.set_multipliers(np.full(n, 0.05), np.cos(np.arange(n, dtype=np.float64)))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), d["x0"])
    for _ in range(ITERS):
        prob.panoc_step()
    x1, z1, s1 = prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars()
    prob.close()
    res = _run_stencil(world, "panoc")
    x = np.concatenate([r[1] for r in res])
    z = np.concatenate([r[2] for r in res])
    for r in res[1:]:
        for key in ("gamma", "f_x", "g_z", "stop_norm", "FBE"):
            assert r[3][key] == res[0][3][key], f"ranks disagree on {key}"
    assert abs(res[0][3]["gamma"] - s1["gamma"]) <= 1e-13 * s1["gamma"]     # a reduction: summed per rank block
    assert np.max(np.abs(x - x1)) <= 1e-10 * np.max(np.abs(x1))
    assert np.max(np.abs(z - z1)) <= 1e-10 * np.max(np.abs(z1))


@pytest.mark.timeout(600)
def test_sharded_stencil_alps(bz):
    """Whole ALPS solve of the obstacle problem with the grid sharded over two ranks: same outer/inner
    iteration counts and the same membrane as the single-rank solve."""
    nxg, nyg = GRID
    n = nxg * nyg
    d = bz.synth.obstacle_grid(nxg, nyg, load=-1.0)
    o = bz.alps(bz.Stencil5ptQuadratic(nxg, nyg, d["b"]), bz.Zero(), bz.IdentityFunction(),
                bz.ClosedSet(bz.IndBox(d["psi"], np.inf)), d["x0"], np.zeros(n))
    res = _run_stencil(2, "alps")
    x = np.concatenate([r[1] for r in res])
    assert all(r[3][2] == o[5] == "first_order" for r in res)
    assert all(r[3][0] == o[2] and abs(r[3][1] - o[3]) <= max(3, 0.05 * o[3]) for r in res)
    # both runs stop at tol = 1e-6 on an ill-conditioned Laplacian after ~1000 inner iterations: the
    # resolution of the comparison is what the oracle itself shows between two summation roundings there
    # (2e-6 in x), not the per-iterate 1e-10 of the test above
    assert np.max(np.abs(x - o[0])) <= 2e-5 * max(1.0, np.max(np.abs(o[0])))
    assert np.all(x >= d["psi"] - 1e-5)


# ---------------------------------------------------------------- row-sharded dense constraint (cfg 4)
DENSE = (96, 640)        # ny x n
DENSE_WIDE = (64, 40000)  # n large enough for the all-gather of A'y to run on many workgroups


def _dense_problem(bz, dtype, shape=None):
    ny, n = shape or DENSE
    rng = np.random.default_rng(21)
    A = (rng.standard_normal((ny, n)) / np.sqrt(ny)).astype(dtype)
    xt = np.where(rng.uniform(size=n) < 0.05, rng.choice([-1.0, 1.0], n), 0.0).astype(dtype)
    b = (A.astype(np.float64) @ xt.astype(np.float64)).astype(dtype)
    return A, b


def _dense_worker(rank, world, conn, mode, dtype_name, shape=None):
    try:
        sys.path.insert(0, ROOT)
        import bazinga_jl_amd as bz
        dtype = np.dtype(dtype_name).type
        ny, n = shape or DENSE
        A, b = _dense_problem(bz, dtype, shape)
        r0, r1 = _row_blocks(ny, world)[rank]
        nyl = r1 - r0
        ctx = bz.Context(device=0, rank=rank, nranks=world, comm_id=None)
        conn.send(ctx.p2p_export())
        ctx.p2p_connect(conn.recv(), [0] * world)
        oracles = (bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A[r0:r1], b[r0:r1]), bz.ZeroSet())
        prob = bz.Problem(*oracles, n, nyl, dtype, ctx)
        conn.send(prob.allreduce_export())
        prob.allreduce_connect(conn.recv())
        if mode == "panoc":
            mu = np.full(nyl, 0.05, dtype)
            y = np.cos(np.arange(r0, r1)).astype(dtype)
            prob.set_multipliers(mu, y)
            prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(n, dtype))
            for _ in range(ITERS):
                prob.panoc_step()
            out = (rank, prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars())
            prob.close()
        else:
            r = bz.alps(*oracles, np.zeros(n, dtype), np.zeros(nyl, dtype), ctx=ctx, problem=prob,
                        tol=dtype(1e-6 if dtype is np.float64 else 1e-4), verbose=bool(os.environ.get("BZ_TEST_VERBOSE")))
            out = (rank, r[0], r[1], (r[2], r[3], r[5]))
        ctx.close()
        conn.send(("ok", out))
    except Exception as e:      # noqa: BLE001
        import traceback
        conn.send(("error", repr(e) + traceback.format_exc()[-1500:]))


def _run_dense(world, mode, dtype_name, shape=None):
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(world)]
    procs = [mpc.Process(target=_dense_worker, args=(r, world, pipes[r][1], mode, dtype_name, shape)) for r in range(world)]
    for p in procs:
        p.start()
    for _round in range(2):                      # mailbox handles, then all-reduce region handles
        hs = [pipes[r][0].recv() for r in range(world)]
        for r in range(world):
            pipes[r][0].send(hs)
    res = []
    for r in range(world):
        assert pipes[r][0].poll(240), "rank did not answer"
        status, payload = pipes[r][0].recv()
        assert status == "ok", payload
        res.append(payload)
    for p in procs:
        p.join(60)
    return sorted(res)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,dtype_name,shape", [(2, "float64", None), (3, "float64", None), (2, "float32", None),
                                                    (2, "float64", DENSE_WIDE)])
def test_row_sharded_dense_constraint_matches_single_rank(bz, world, dtype_name, shape):
    """SURVEY §8(e)/(f-4), cfg 4 sharded: the rows of A (and b, mu, y) cut into blocks, x replicated; A' yhat
    summed over the ranks in rank order through IPC-mapped regions; x-space scalars counted once, the
    constraint-space ones added up.  Every rank must hold the SAME x (bit for bit) and it must equal the
    single-rank iterate to the north-star tolerance (float32: a few ulps)."""
    dtype = np.dtype(dtype_name).type
    ny, n = shape or DENSE
    A, b = _dense_problem(bz, dtype, shape)
    prob = bz.Problem(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A, b), bz.ZeroSet(), n, ny, dtype)
    prob.set_multipliers(np.full(ny, 0.05, dtype), np.cos(np.arange(ny)).astype(dtype))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 9).c_opts(), np.zeros(n, dtype))
    for _ in range(ITERS):
        prob.panoc_step()
    x1, z1, s1 = prob.panoc_vector("x"), prob.panoc_vector("z"), prob.panoc_scalars()
    prob.close()
    res = _run_dense(world, "panoc", dtype_name, shape)
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])     # replicas stay identical
        for key in ("gamma", "f_x", "g_z", "stop_norm", "FBE"):
            assert r[3][key] == res[0][3][key], f"ranks disagree on {key}"
    tol = 1e-10 if dtype is np.float64 else 2e-4
    assert abs(res[0][3]["gamma"] - s1["gamma"]) <= (1e-13 if dtype is np.float64 else 1e-5) * s1["gamma"]
    assert np.max(np.abs(res[0][1] - x1)) <= tol * max(1e-30, np.max(np.abs(x1)))
    assert np.max(np.abs(res[0][2] - z1)) <= tol * max(1e-30, np.max(np.abs(z1)))


@pytest.mark.timeout(600)
def test_row_sharded_dense_constraint_alps(bz):
    """Whole ALPS solve of the basis-pursuit problem with the rows of A over two ranks."""
    ny, n = DENSE
    A, b = _dense_problem(bz, np.float64)
    o = bz.alps(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A, b), bz.ZeroSet(), np.zeros(n), np.zeros(ny))
    res = _run_dense(2, "alps", "float64")
    assert np.array_equal(res[0][1], res[1][1])
    y = np.concatenate([r[2] for r in res])
    assert all(r[3][2] == o[5] == "first_order" for r in res)
    # ~2000 inner iterations of a nonsmooth problem with tau backtracks: the two runs follow each other for the
    # first ~70 iterations (the 1e-10 test above) and then take different, equally valid paths to the same
    # solution — counts agree loosely, the solutions to the solver's tolerance.  (How loosely: the same problem solved
    # with affine_refresh = 0, 1, 4, 8, 16, 32, 64 — seven roundings of one algorithm — takes 11, 10, 14, 11, 13, 12, 9 outer
    # and 2278 ... 2729 inner iterations, tools/refresh_counts.py.)
    assert all(abs(r[3][0] - o[2]) <= 5 and abs(r[3][1] - o[3]) <= 0.25 * o[3] for r in res)
    assert np.max(np.abs(res[0][1] - o[0])) <= 5e-4 * max(1.0, np.max(np.abs(o[0])))
    assert np.array_equal(np.abs(res[0][1]) > 1e-3, np.abs(o[0]) > 1e-3)          # same support
    assert np.max(np.abs(A @ res[0][1] - b)) <= 1e-5
    assert y.shape == o[1].shape
