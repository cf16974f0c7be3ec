"""bench.py's launcher-side plumbing for N > 1 (TCP star through rank 0): all-gather, barrier, reductions.
World size 4 on localhost; ranks start in reverse order so the connect-retry path is exercised."""
import multiprocessing as mp
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    g = bench.SocketGroup(rank, world, "127.0.0.1", port, timeout=30)
    parts = g.allgather(("r%d" % rank).encode() * (rank + 1))
    g.barrier()
    mx = g.reduce(rank * 1.5, max)
    mn = g.reduce(1 if rank != 2 else 0, min)
    g.barrier()
    g.close()
    q.put((rank, parts, mx, mn))


def test_socket_group_world4():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in reversed(procs):
        p.start()
        time.sleep(0.05)
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(r[1] == res[0][1] for r in res)
    assert res[0][1] == [b"r0", b"r1r1", b"r2r2r2", b"r3r3r3r3"]
    assert all(r[2] == 4.5 and r[3] == 0 for r in res)
