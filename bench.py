#!/usr/bin/env python3
"""bench.py — PANOCplus inner iterations/sec on the BASELINE headline workload.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md §8(d) cfg 2): l1-regularised diagonal
quadratic, n = 10^7, fp64, f = sum x(0.5 q x - b), g = 2.5||x||_1 (soft-threshold prox),
c = Identity, D = Box[-1,1], LBFGS(5), mu = default_penalty_parameter!(x0 = 0) = 0.1,
y = 0 (outer iteration 1), tol = 0 so the solver never stops; synthetic splitmix64 data.

A "step" is one PANOCplus inner iteration (one Base.iterate(iter, state)).  Inputs and all
solver state are resident in HBM before the timed region starts.

N > 1 (one process per GPU, launched by torch.distributed.run): the SAME n = 10^7 problem
with x sharded in contiguous blocks over the ranks ("scaling": "strong"); the only data
exchanged are the reductions' partial scalars (RCCL all-gather of <= 10 doubles per rank).

Output: ONE JSON line on rank 0 (contract fields + "roofline" + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
M_LBFGS = 5


def algorithmic_bytes_per_iter(n, w=8, m=M_LBFGS, n_al=2, n_fb=1, p_al=6):
    """B_iter = w n [(8m+1) + P_AL nAL + 4 nFB + 8]  (cfg 2: 65 passes)."""
    return w * n * ((8 * m + 1) + p_al * n_al + 4 * n_fb + 8)


def cpu_baseline(n, states):
    """The oracle's plain-C port (oracle/c/bz_oracle.c: one loop per Julia broadcast, no fusion) timed on
    this box's host cores on the same workload, bounded sample.  Two legs: single-threaded
    — what the reference's Julia broadcasts are — and the same loops split over the host threads."""
    from oracle import c_port
    import bazinga_jl_amd as bz
    d = bz.synth.l1_quadratic(n)
    mu, y, x0 = np.full(n, 0.1), np.zeros(n), np.zeros(n)
    kw = dict(lam=d["lam"], D="box", D_lo=d["lo"], D_hi=d["hi"], minimum_gamma=float(np.finfo(float).eps))
    c_port.load()
    t0 = time.perf_counter()
    c_port.panoc_run(d["q"], d["b"], mu, y, x0, states, **kw)
    dt = time.perf_counter() - t0
    out = {"value": round((states - 1) / dt, 4), "unit": "iterations/s", "cores": 1, "kind": "port",
           "sample": f"first {states - 1} PANOCplus iterations (plus the initial state) of the same n={n} "
                     f"workload, oracle/c/bz_oracle.c, single thread, {dt:.1f} s",
           "host_cpus": os.cpu_count(),
           # the real reference would be timed here if the box had Julia + ProximalAlgorithms
           "julia_on_box": __import__("shutil").which("julia") is not None}
    try:
        threads = int(os.environ.get("BZ_BENCH_CPU_THREADS", "0")) or min(16, len(os.sched_getaffinity(0)))
        os.environ["OMP_NUM_THREADS"] = str(threads)
        c_port.load(omp=True)
        st2 = 4 * states - 3
        c_port.panoc_run(d["q"], d["b"], mu, y, x0, 3, omp=True, **kw)          # thread pool + page warm-up
        t0 = time.perf_counter()
        c_port.panoc_run(d["q"], d["b"], mu, y, x0, st2, omp=True, **kw)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": round((st2 - 1) / dt2, 4), "unit": "iterations/s", "cores": threads,
                            "sample": f"first {st2 - 1} iterations, same loops under `omp parallel for`, "
                                      f"{threads} threads, {dt2:.1f} s"}
    except Exception as e:      # noqa: BLE001  (no OpenMP build on this box: the single-thread leg stands)
        out["all_cores"] = {"value": None, "note": repr(e)[:200]}
    return out


class SocketGroup:
    """Minimal process group over TCP (star through rank 0) for the launcher-side plumbing of N > 1:
    all-gather of small byte strings, barrier, max/min reductions.  torch.distributed would do, but
    importing torch makes libbazinga_hip bind to torch's bundled ROCm 7.0 HIP runtime instead of the
    system's 7.2 (measured: +20 us per iteration of launch overhead); the data path never uses this."""
    MAGIC = b"BZRV1"

    def __init__(self, rank, world, addr, port, timeout=120.0):
        import socket
        import struct
        self.rank, self.world, self._struct = rank, world, struct
        self.peers = []
        if rank == 0:
            srv = None
            for off in range(1, 40):
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port + off))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("no free rendezvous port")
            srv.listen(world)
            srv.settimeout(timeout)
            slots = [None] * world
            while sum(c is not None for c in slots[1:]) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                hello = self._recvn(c, len(self.MAGIC) + 4)
                if hello[:len(self.MAGIC)] != self.MAGIC:
                    c.close()
                    continue
                r = struct.unpack("<i", hello[len(self.MAGIC):])[0]
                c.sendall(b"BZOK")
                slots[r] = c
            self.peers = slots
            srv.close()
        else:
            deadline = time.time() + timeout
            sock = None
            while sock is None and time.time() < deadline:
                for off in range(1, 40):
                    try:
                        c = socket.create_connection((addr, port + off), timeout=2.0)
                        c.settimeout(5.0)
                        c.sendall(self.MAGIC + struct.pack("<i", rank))
                        if self._recvn(c, 4) == b"BZOK":
                            c.settimeout(timeout)
                            sock = c
                            break
                        c.close()
                    except OSError:
                        continue
                if sock is None:
                    time.sleep(0.2)
            if sock is None:
                raise RuntimeError("could not reach rank 0")
            self.sock = sock

    @staticmethod
    def _recvn(c, n):
        buf = b""
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("peer closed the rendezvous connection")
            buf += chunk
        return buf

    def _send(self, c, b):
        c.sendall(self._struct.pack("<i", len(b)) + b)

    def _recv(self, c):
        n = self._struct.unpack("<i", self._recvn(c, 4))[0]
        return self._recvn(c, n)

    def allgather(self, b: bytes):
        if self.rank == 0:
            parts = [b] + [self._recv(self.peers[r]) for r in range(1, self.world)]
            blob = b"".join(self._struct.pack("<i", len(x)) + x for x in parts)
            for r in range(1, self.world):
                self._send(self.peers[r], blob)
            return parts
        self._send(self.sock, b)
        blob = self._recv(self.sock)
        out, o = [], 0
        for _ in range(self.world):
            n = self._struct.unpack("<i", blob[o:o + 4])[0]
            out.append(blob[o + 4:o + 4 + n])
            o += 4 + n
        return out

    def barrier(self):
        self.allgather(b"")

    def reduce(self, x: float, op):
        vals = [self._struct.unpack("<d", v)[0] for v in self.allgather(self._struct.pack("<d", float(x)))]
        return op(vals)

    def close(self):
        for c in self.peers[1:] if self.rank == 0 else [self.sock]:
            try:
                c.close()
            except OSError:
                pass


class TorchGroup:
    """Same interface over torch.distributed (fallback if the TCP rendezvous cannot be set up)."""

    def __init__(self, rank, world, local_rank):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def allgather(self, b: bytes):
        out = [None] * self.world
        self.dist.all_gather_object(out, b)
        return out

    def barrier(self):
        self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, x, op):
        out = [None] * self.world
        self.dist.all_gather_object(out, float(x))
        return op(out)

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def side_workload(args):
    """cfg 3 / cfg 4 (single GPU): the other BASELINE configs, same JSON shape; not the headline."""
    import bazinga_jl_amd as bz
    eps64, eps32 = float(np.finfo(np.float64).eps), float(np.finfo(np.float32).eps)
    if args.workload == "cfg3":
        d = bz.synth.obstacle_grid(2048)
        n = ny = 2048 * 2048
        dt, w = np.float64, 8
        prob = bz.Problem(bz.Stencil5ptQuadratic(2048, 2048, d["b"]), bz.Zero(), bz.IdentityFunction(),
                          bz.ClosedSet(bz.IndBox(d["psi"], np.inf)), n, ny, dt)
        x0, mg = d["x0"], eps64
        b_iter = 63 * w * n            # 63 passes
        # the register-resident two-loop kernel is the largest single launch here: model (8m+1) - 4 passes
        # (the last axpy is k_axpy_dot's), it moves 4m
        cat_name, label = "k_twoloop_persist", "cfg3: 5-pt stencil QP on 2048^2 grid fp64, box D, g=Zero, LBFGS(5)"
        launch_bytes = lambda m: ((8 * m + 1) - 4) * w * n
        kernel = "bz::k_twoloop_persist<double>"
    else:
        ny, n = 8192, 65536
        d = bz.synth.basis_pursuit(ny, n, dtype=np.float32)
        dt, w = np.float32, 4
        prob = bz.Problem(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(d["A"], d["b"]), bz.ZeroSet(), n, ny, dt)
        x0, mg = np.zeros(n, dt), eps32
        b_iter = 4 * ny * n * w        # A read twice per AL gradient, 2 AL gradients per iteration
        cat_name, label = "gemv", "cfg4: basis pursuit, dense A 8192x65536 fp32, l1 prox, D=ZeroSet, LBFGS(5)"
        launch_bytes = lambda m: ny * n * w
        kernel = "bz::k_gemv_n<float> + bz::k_gemv_t_mfma (each streams A once)"
    prob.set_multipliers(np.full(ny, 0.1, dt), np.zeros(ny, dt))
    prob.panoc_begin(bz.PANOCplus(tol=0.0, maxit=10 ** 12, minimum_gamma=mg).c_opts(), x0)
    for _ in range(args.warmup):
        prob.panoc_step()
    prob.profile_reset()
    prob.profile_enable(True)
    prob.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prob.panoc_step()
    prob.ctx.synchronize()
    elapsed = time.perf_counter() - t0
    prob.profile_enable(False)
    prof_all = prob.profile()
    prof = prof_all[cat_name]
    if args.workload == "cfg4" and prof_all["k_gemv_t_mfma"]["launches"]:      # both GEMV kernels stream A once
        prof = {"launches": prof["launches"] + prof_all["k_gemv_t_mfma"]["launches"],
                "total_ms": prof["total_ms"] + prof_all["k_gemv_t_mfma"]["total_ms"]}
    sc = prob.panoc_scalars()
    its = args.steps / elapsed
    avg_s = (prof["total_ms"] / 1e3) / max(1, prof["launches"])
    achieved = launch_bytes(max(1, int(sc["lbfgs_mem"]))) / avg_s / 1e9
    print(json.dumps({
        "metric": "PANOC inner iterations/sec (%s)" % args.workload, "value": round(its, 3), "unit": "iterations/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64" if w == 8 else "f32",
        "data": "synthetic", "config": {"workload": label, "n": n, "ny": ny},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
                     "avg_launch_us": round(avg_s * 1e6, 3), "launches_per_iteration": prof["launches"] / args.steps},
        "roofline_iteration": {"algorithmic_bytes_per_iteration": int(b_iter), "achieved": round(b_iter * its / 1e9, 1),
                               "unit": "GB/s", "frac": round(b_iter * its / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels": {k: {"launches_per_iteration": round(v["launches"] / args.steps, 2),
                        "avg_us": round(1e3 * v["total_ms"] / v["launches"], 2)}
                    for k, v in prof_all.items() if v["launches"]},
        "solver": {"gamma": sc["gamma"], "stop_norm": sc["stop_norm"], "k": int(sc["k"]), "lbfgs_mem": int(sc["lbfgs_mem"])},
        "cpu_baseline": None}), flush=True)
    prob.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", "--size", dest="n", type=float, default=1e7,
                    help="global problem size (default: BASELINE cfg 2); spell it --size under torch.distributed.run, "
                         "whose own parser claims --n as an abbreviation")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 (default, the headline metric); cfg3 2048^2 stencil QP; cfg4 dense-A basis "
                         "pursuit fp32; cfg5 = cfg2 at n=1e8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-states", type=int, default=31)
    ap.add_argument("--no-fuse", action="store_true")
    ap.add_argument("--compact", action="store_true", help="(default) compact L-BFGS representation")
    ap.add_argument("--two-loop", action="store_true",
                    help="evaluate the L-BFGS operator by the two-loop recursion in the reference's operation order "
                         "(2M sequential reductions: 11 exchanges per iteration at N > 1)")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the two-loop and outer-iteration-3 side measurements")
    ap.add_argument("--no-p2p", action="store_true", help="N > 1: keep the RCCL all-gather for the scalar exchange")
    ap.add_argument("--no-rccl", action="store_true",
                    help="N > 1: do not create the RCCL communicator (p2p mailboxes only, checked for rank agreement "
                         "instead of against RCCL).  With BZ_BENCH_SAME_GPU=1 (all ranks on device 0) this rehearses "
                         "the N > 1 path on a one-GPU box, where RCCL refuses two ranks on one device")
    args = ap.parse_args()

    import bazinga_jl_amd as bz

    n = int(args.n)
    if args.workload == "cfg5":
        n = 100_000_000
    if args.workload in ("cfg3", "cfg4"):
        return side_workload(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed.run launch with that many ranks "
                         f"(WORLD_SIZE={world})")
    grp = None
    comm_id = None
    if world > 1:
        try:
            grp = SocketGroup(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                              int(os.environ.get("MASTER_PORT", "29500")))
        except Exception as e:      # noqa: BLE001
            print(f"[bench] TCP rendezvous failed ({e!r}); using torch.distributed", file=sys.stderr, flush=True)
            grp = TorchGroup(rank, world, local_rank)
        if not args.no_rccl:
            ids = grp.allgather(bz.Context.unique_id() if rank == 0 else b"")
            comm_id = ids[0]
    dev = local_rank if world > 1 else 0
    if os.environ.get("BZ_BENCH_SAME_GPU") == "1":
        dev = 0

    def agree(flag):
        return grp.reduce(flag, min) >= 1 if grp is not None else bool(flag)

    ctx, rccl_note = None, None
    try:
        if world == 1 or comm_id is not None:
            ctx = bz.Context(device=dev, rank=rank, nranks=world, comm_id=comm_id)
    except Exception as e:      # noqa: BLE001
        rccl_note = f"RCCL communicator failed: {e!r}"[:300]
        print(f"[bench] rank {rank}: {rccl_note}", file=sys.stderr, flush=True)
    if world > 1 and not agree(ctx is not None):
        if ctx is not None:
            ctx.close()
        ctx = None                              # no RCCL on this node: the p2p transport or nothing

    lo_i, hi_i = bz.shard_bounds(n, rank, world)
    nl = hi_i - lo_i
    d = bz.synth.l1_quadratic(nl, start=lo_i)
    # The L-BFGS operator is evaluated in its compact representation: the whole iteration is ONE pass over
    # 2M + 11 vectors and one reduction phase (one cross-GPU exchange at N > 1), against 2M sequential
    # reductions for the two-loop recursion.  Same operator, alternate rounding — its iterates stay as close
    # to the fp64 oracle's as the two-loop kernels' do (oracle: LBFGSCompactOperator; tests:
    # test_compact_lbfgs_*; tests/stress/err_compact_vs_twoloop.py) — and it is what LBFGS(M) means by default on this
    # path (lbfgs_compact = 2, "auto").  --two-loop times the reference's order.
    compact = not args.two_loop
    popts = bz.PANOCplus(tol=0.0, maxit=10 ** 12, minimum_gamma=float(np.finfo(float).eps),
                         fuse=not args.no_fuse, directions=bz.LBFGS(M_LBFGS, compact=compact)).c_opts()

    def make_problem(c, po=None, mu=None, y=None):
        p = bz.Problem(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                       bz.ClosedSet(bz.IndBox(d["lo"], d["hi"])), nl, nl, np.float64, c)
        p.set_multipliers(np.full(nl, 0.1) if mu is None else mu, np.zeros(nl) if y is None else y)
        p.panoc_begin(po or popts, np.zeros(nl))
        return p

    # N > 1: scalar exchange through peer-to-peer mailboxes (no collective call; the persistent two-loop
    # kernel runs sharded).  It is taken only if it reproduces the RCCL path's scalars on this node;
    # otherwise the RCCL all-gather path (always correct, slower) is timed.
    transport = "none" if world == 1 else ("rccl" if ctx is not None else None)
    p2p_note = None
    ctx_rccl = None
    if world > 1 and not (args.no_p2p and ctx is not None):
        ok, ctx2 = 1, None
        try:                                    # stage 1: map everybody's mailbox
            ctx2 = bz.Context(device=dev, rank=rank, nranks=world, comm_id=None)
            h = ctx2.p2p_export()
        except Exception as e:      # noqa: BLE001
            ok, p2p_note = 0, f"p2p export failed: {e!r}"[:300]
            h = b"\0" * 64
        hs = grp.allgather(h)
        dvs = [int(v) for v in grp.allgather(str(dev).encode())]
        if agree(ok):
            try:
                ctx2.p2p_connect(hs, dvs)
            except Exception as e:      # noqa: BLE001
                ok, p2p_note = 0, f"p2p connect failed: {e!r}"[:300]
        else:
            ok = 0
        if agree(ok):                           # stage 2: 8 iterations, same scalars as the RCCL path
            keys = ("gamma", "f_x", "g_z", "stop_norm", "FBE")
            sa = None
            if ctx is not None:
                pa = make_problem(ctx)
                for _ in range(8):
                    pa.panoc_step()
                sa = pa.panoc_scalars()
                pa.close()
            sb = None
            try:
                pb = make_problem(ctx2)
                for _ in range(8):
                    pb.panoc_step()
                sb = pb.panoc_scalars()
                pb.close()
                if sa is not None:
                    for key in keys:
                        if not abs(sa[key] - sb[key]) <= 1e-9 * max(1.0, abs(sa[key])):
                            ok, p2p_note = 0, f"p2p/rccl mismatch on {key}: {sb[key]} vs {sa[key]}"
                if not all(np.isfinite(sb[key]) for key in keys):
                    ok, p2p_note = 0, "p2p run produced non-finite scalars"
            except Exception as e:      # noqa: BLE001  (a peer never answered: every rank times out alike)
                ok, p2p_note = 0, f"p2p run failed: {e!r}"[:300]
            # every rank must hold the same bits (they take the line-search decisions independently)
            mine = repr([sb[key] for key in keys]).encode() if sb is not None else b"none"
            if len(set(grp.allgather(mine))) != 1:
                ok, p2p_note = 0, p2p_note or "ranks disagree on the p2p scalars"
            if agree(ok):
                if ctx is not None:
                    ctx_rccl = ctx
                ctx, transport = ctx2, "p2p"
                if sa is None:
                    p2p_note = "p2p checked for rank agreement only (no RCCL communicator: " + (rccl_note or "--no-rccl") + ")"
        if transport != "p2p" and p2p_note is None:
            p2p_note = "p2p rejected on another rank"
    if ctx is None:
        raise SystemExit(f"[bench] no working scalar transport at N={world}: {rccl_note}; {p2p_note}")
    def sync(ok=1):
        """Barrier + device drain that carries a health flag: the SAME group operation on every rank whatever
        happened locally, so one rank's failure cannot leave the others waiting in a different collective."""
        try:
            ctx.synchronize()
        except Exception:       # noqa: BLE001
            ok = 0
        if grp is not None:
            ok = 1 if grp.reduce(ok, min) >= 1 else 0
        try:
            ctx.synchronize()
        except Exception:       # noqa: BLE001
            ok = 0
        return ok

    ALG = ("k_twoloop_persist", "k_axpy_dot", "k_fused_sep", "k_dot", "k_fused_iterates")
    FUSED_FORMS = ("k_fused_sep", "k_fused_iterates")      # (category 1 also counts k_fused_compact's stored-pair forms)

    def timed_run(prob, steps, warmup, restart_opts=None):
        restart_opts = restart_opts or popts
        x0z = np.zeros(prob.n)
        x0z.fill(0.0)      # (pages touched now, not inside the timed bracket)
        """W untimed + K timed iterations on `prob`.  HIP events bound to each dispatch on the library's own
        stream (hipExtLaunchKernelGGL start/stop events): in the warm-up every kernel category is timed, to
        find the dominant kernel and fill the per-kernel table; in the timed region only every 8th launch of
        the dominant kernel carries events, so they do not perturb the pipeline."""
        ok, err, prof_warm, dom, st0 = 1, None, {}, "k_axpy_dot", None
        try:
            prob.profile_reset()
            prob.profile_enable(os.environ.get("BZ_BENCH_WARMPROF", "1") == "1")
            for i in range(warmup):
                prob.panoc_step()
                if i % 50 == 49 and i + 1 < warmup and prob.panoc_scalars()["stop_norm"] < 1e-12:
                    prob.panoc_begin(restart_opts, x0z)      # (as in the timed region, below)
            prof_warm = prob.profile()
            cands = {k: v for k, v in prof_warm.items() if k in ALG and v["launches"]}
            dom = max(cands, key=lambda k: cands[k]["total_ms"]) if cands else "k_axpy_dot"
            prob.profile_reset()
            mask = 1 << bz._lib.KERNEL_CATEGORIES.index(dom)
            if dom in FUSED_FORMS:      # both forms of the one-pass kernel: which one dominates is only known afterwards
                mask = sum(1 << bz._lib.KERNEL_CATEGORIES.index(k) for k in FUSED_FORMS)
            prob.profile_enable(mask, period=int(os.environ.get("BZ_BENCH_PERIOD", "8")))
            st0 = prob.panoc_stats()
        except Exception as e:      # noqa: BLE001
            ok, err = 0, repr(e)[:300]
        ok = sync(ok)
        t0 = time.perf_counter()
        restarts, carry = 0, [0, 0, 0]
        if ok:
            try:
                # K iterations through the solver's own loop (bz_panoc_steps), 50 per library call.  With tol = 0
                # the solve never stops by itself; once it has converged to rounding (stop norm < 1e-12: about
                # iteration 280 of this workload, so never within the default K) it is started again from x0 INSIDE
                # the timed bracket — iterations at the noise floor, where every other pair has <s,y> <= 0 and is
                # skipped, are not what a solve does.  Every rank sees the same scalars, so all restart together.
                left = steps
                while left > 0:
                    c = min(left, 50)
                    prob.panoc_steps(c)
                    left -= c
                    if left > 0 and prob.panoc_scalars()["stop_norm"] < 1e-12:
                        stq = prob.panoc_stats()      # (bz_panoc_begin zeroes the counters: carry them over)
                        carry = [carry[0] + stq.n_fused_iters, carry[1] + stq.n_grad, carry[2] + stq.n_prox]
                        prob.panoc_begin(restart_opts, x0z)
                        restarts += 1
            except Exception as e:      # noqa: BLE001
                ok, err = 0, repr(e)[:300]
        ok = sync(ok)
        elapsed = time.perf_counter() - t0
        if grp is not None:
            elapsed = grp.reduce(elapsed, max)
        if not ok:
            return {"failed": err or "another rank failed"}
        prob.profile_enable(False)
        prof = prob.profile()
        if dom in FUSED_FORMS:
            dom = max(FUSED_FORMS, key=lambda k: prof[k]["total_ms"])
        return {"elapsed": elapsed, "st0": st0, "st1": prob.panoc_stats(), "sc": prob.panoc_scalars(),
                "prof": prof, "prof_warm": prof_warm, "dom": dom, "restarts": restarts, "carry": carry}

    prob = make_problem(ctx)
    R = timed_run(prob, args.steps, args.warmup)
    if "failed" in R and transport == "p2p" and ctx_rccl is not None:
        # every rank sees the same verdict (sync carries it): fall back to the RCCL transport together
        p2p_note = f"p2p failed in the timed run ({R['failed']}); RCCL timed instead"
        print(f"[bench] rank {rank}: {p2p_note}", file=sys.stderr, flush=True)
        ctx, transport = ctx_rccl, "rccl"
        prob = make_problem(ctx)
        R = timed_run(prob, args.steps, args.warmup)
    if "failed" in R:
        raise SystemExit(f"[bench] rank {rank}: timed run failed: {R['failed']}")
    prob.close()
    elapsed, st0, st1, sc, prof_all, prof_warm, dom = (R[k] for k in ("elapsed", "st0", "st1", "sc", "prof", "prof_warm", "dom"))

    # N = 1 extras (rank 0 prints them beside the headline, they never replace it):
    #   two_loop : the same workload with the L-BFGS operator in the reference's two-loop operation order
    #   outer3   : the same workload inside outer iteration 3 (mu, y after two ALPS outer iterations: y != 0),
    #              SURVEY §8(d)
    extras = {}
    if world == 1 and not args.no_extras:
        if compact:
            popts_tl = bz.PANOCplus(tol=0.0, maxit=10 ** 12, minimum_gamma=float(np.finfo(float).eps),
                                    fuse=not args.no_fuse, directions=bz.LBFGS(M_LBFGS, compact=False)).c_opts()
            p2 = make_problem(ctx, popts_tl)
            r2 = timed_run(p2, args.steps, args.warmup, popts_tl)
            p2.close()
            extras["two_loop"] = {"value": round(args.steps / r2["elapsed"], 3), "unit": "iterations/s",
                                  "ms_per_step": round(1e3 * r2["elapsed"] / args.steps, 5),
                                  "note": "directions=LBFGS(5) evaluated by the two-loop recursion in the reference's operation "
                                          "order (persistent register-resident kernel); same iterates up to rounding"}
        out3 = bz.alps(bz.DiagQuadratic(d["q"], d["b"]), bz.NormL1(d["lam"]), bz.IdentityFunction(),
                       bz.ClosedSet(bz.IndBox(d["lo"], d["hi"])), np.zeros(nl), np.zeros(nl), maxit=2, ctx=ctx)
        p3 = make_problem(ctx, None, out3[9], out3[1])
        r3 = timed_run(p3, args.steps, args.warmup)
        p3.close()
        extras["outer3"] = {"value": round(args.steps / r3["elapsed"], 3), "unit": "iterations/s",
                            "ms_per_step": round(1e3 * r3["elapsed"] / args.steps, 5),
                            "note": "same workload with (mu, y) as left by two ALPS outer iterations (max|y| = %.3g)"
                                    % float(np.max(np.abs(out3[1])))}
    del d

    if rank == 0:
        its = args.steps / elapsed
        carry = R.get("carry", [0, 0, 0])
        n_fused = st1.n_fused_iters + carry[0] - st0.n_fused_iters
        n_al = (st1.n_grad + carry[1] - st0.n_grad) / args.steps
        n_fb = (st1.n_prox + carry[2] - st0.n_prox) / args.steps
        m = int(sc["lbfgs_mem"])
        w = 8
        # dominant kernel = largest total time in the timed region.  ALGORITHMIC bytes per launch follow
        # SURVEY.md §8(d) (compulsory passes under the reference's dataflow), over the local shard:
        #   k_twoloop_persist : the two-loop minus its last axpy = (8m+1) - 4 passes      (moves 4m)
        #   k_axpy_dot        : 3R+1W per step, the middle step 2R+1W -> (8m-5)/(2m-1) passes on average
        #   k_fused_sep       : last axpy + x_d (4) + 2 AL gradients (2*6) + FB step (4) + update/stop (8)
        #   k_fused_compact   : the WHOLE iteration is this one launch: (8m+1) + 12 + 4 + 8 = 65 passes at m = 5
        #                       (in steady state it moves m + 4 .. m + 6 = 9 .. 11: reads the m+1 last iterates —
        #                       the stored pairs and all m+1 residuals are re-formed from them in registers —
        #                       and q, b, plus mu and mu*y unless the penalties are uniform / the multipliers zero
        #                       (passed as numbers then); writes x_d.  z, res, s, y are never stored; the next
        #                       application's S'res, Y'res come out of the same pass.  While the ring of iterates
        #                       fills (first m+1 iterations, or after a gamma halving / a skipped pair) the
        #                       stored-pair form moves 2m + 10 = 20)
        alg_passes = {"k_twoloop_persist": (8 * m + 1) - 4,
                      "k_axpy_dot": (4.0 * (2 * m - 2) + 3.0) / (2 * m - 1) if m >= 1 else 0.0,
                      "k_fused_sep": 4 + 12 + 4 + 8, "k_dot": 2}
        real_name = dom
        if compact:
            alg_passes["k_dot"] = 2 * m + 1
            alg_passes["k_fused_sep"] = (8 * m + 1) + 12 + 4 + 8
            alg_passes["k_fused_iterates"] = alg_passes["k_fused_sep"]      # the same iteration, fewer bytes moved
            real_name = {"k_fused_sep": "k_fused_compact", "k_fused_iterates": "k_fused_compact",
                         "k_dot": "k_gram_dots"}.get(dom, dom)
        prof = prof_all[dom]
        launches_per_it = n_fused / max(1, args.steps) if dom != "k_axpy_dot" else 9.0
        bytes_per_launch = alg_passes[dom] * w * nl
        avg_s = (prof["total_ms"] / 1e3) / max(1, prof["launches"])
        achieved = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
        if os.path.exists(pmc) and world == 1 and n == 10_000_000:
            with open(pmc) as fh:
                pj = json.load(fh)
            traffic = pj.get("kernels", {}).get(real_name)
        b_iter = algorithmic_bytes_per_iter(n, n_al=2, n_fb=1)
        out = {
            "metric": "PANOC inner iterations/sec, n=10^7 l1-quadratic" if n == 10_000_000 else "PANOC inner iterations/sec, n=%d l1-quadratic" % n,
            "value": round(its, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload + ": l1-regularised diagonal quadratic, n=%d fp64, soft-threshold prox_g, "
                                   "c=Identity, D=Box[-1,1], LBFGS(5), mu=0.1, y=0, tol=0" % n,
                       "n": n, "n_per_gpu": nl, "lbfgs_memory": M_LBFGS,
                       "parallelism": "single GPU" if world == 1 else
                       f"x sharded over {world} GPUs, scalars exchanged by " +
                       ("peer-to-peer mailboxes over xGMI" if transport == "p2p" else "RCCL all-gather"),
                       "scalar_transport": transport, "p2p_note": p2p_note, "rccl_note": rccl_note,
                       "lbfgs_form": "compact representation: one pass and one reduction phase per iteration" if compact
                       else "two-loop recursion (persistent kernel, 2M-1 grid phases)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "bz::%s<double>" % real_name,
                         "kernel_form": ("steady state: history kept as iterates (template arguments XR=2, UNI per the penalties), "
                                         "own timing category" if dom == "k_fused_iterates" else None),
                         "launches_per_iteration": round(launches_per_it, 2),
                         "avg_launch_us": round(avg_s * 1e6, 3), "timed_launches": prof["launches"],
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "algorithmic_passes_per_launch": round(alg_passes[dom], 3),
                         # what the memory system actually sustained: PMC bytes / measured duration.  frac above
                         # can exceed 1 because the kernel moves fewer bytes than the reference's dataflow needs.
                         "traffic_rate": round(traffic / avg_s / 1e9, 1) if traffic and avg_s > 0 else None,
                         "traffic_frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_s > 0 else None},
            "kernels_warmup": {k: {"launches_per_iteration": round(v["launches"] / max(1, args.warmup), 2),
                                   "avg_us": round(1e3 * v["total_ms"] / v["launches"], 2)}
                               for k, v in prof_warm.items() if v["launches"]},
            "roofline_iteration": {"algorithmic_bytes_per_iteration": b_iter,
                                   "achieved": round(b_iter * its / 1e9, 1), "unit": "GB/s",
                                   "frac": round(b_iter * its / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "SURVEY §8(d) model, 65 passes at m=5, nAL=2, nFB=1"},
            "solver": {"fused_iterations": int(n_fused), "al_grads_per_it": n_al, "prox_per_it": n_fb,
                       "restarts_in_timed_region": int(R.get("restarts", 0)),
                       "lbfgs_mem": m, "gamma": sc["gamma"], "stop_norm": sc["stop_norm"],
                       "k": int(sc["k"])},
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample: ~10-30 s of CPU work whatever the size (the port runs ~2.5 it/s per 1e7 elements)
            states = max(4, int(round((args.cpu_states - 1) * min(1.0, 1.0e7 / n))) + 1)
            out["cpu_baseline"] = cpu_baseline(n, states)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    ctx.close()
    if ctx_rccl is not None and ctx_rccl is not ctx:
        ctx_rccl.close()
    if grp is not None:
        grp.barrier()
        grp.close()


if __name__ == "__main__":
    main()
