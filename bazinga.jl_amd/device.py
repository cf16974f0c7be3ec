"""Context / problem handles over the C ABI, plus single-node multi-GPU setup."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .oracles import lower

_default_ctx = None


class Context:
    """One per process/GPU (bz_ctx).  nranks > 1: x is sharded over the ranks and the
    reductions' partial scalars are all-gathered with RCCL."""

    def __init__(self, device=0, rank=0, nranks=1, comm_id: bytes | None = None, runtime_tuning: bool = False,
                 shared_device: bool = False):
        lib = L.load()
        o = L.CtxOpts()
        o.device, o.rank, o.nranks = device, rank, nranks
        # shared_device: the GPU is not this process's own — no launch is made to wait, resident, at a gate
        o.flags = (L.BZ_CTX_RUNTIME_TUNING if runtime_tuning else 0) | (L.BZ_CTX_SHARED_DEVICE if shared_device else 0)
        self._id = None
        if comm_id is not None and len(comm_id) != 128:
            raise ValueError("comm_id must be the 128-byte id from unique_id() on rank 0")
        if comm_id is not None:
            self._id = C.create_string_buffer(comm_id, 128)
            o.comm_id = C.cast(self._id, C.c_void_p)
        h = C.c_void_p()
        L.check(lib.bz_ctx_create(C.byref(o), C.byref(h)))
        self._h = h
        self.device, self.rank, self.nranks = device, rank, nranks

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.load().bz_comm_unique_id(buf))
        return buf.raw

    def synchronize(self):
        """Drain the solver stream and the device (hipDeviceSynchronize)."""
        L.check(L.load().bz_ctx_synchronize(self._h))

    def comm_nranks(self) -> int:
        """Ranks the RCCL communicator of this context spans (0 without one)."""
        out = C.c_int32()
        L.check(L.load().bz_ctx_comm_nranks(self._h, C.byref(out)))
        return out.value

    def p2p_export(self) -> bytes:
        """This rank's mailbox IPC handle (64 bytes); all-gather them, then p2p_connect."""
        buf = C.create_string_buffer(64)
        L.check(L.load().bz_ctx_p2p_export(self._h, buf))
        return buf.raw

    def p2p_connect(self, handles, devices):
        """handles: list of nranks 64-byte handles in rank order; devices: HIP ordinal of every rank."""
        blob = C.create_string_buffer(b"".join(handles), 64 * self.nranks)
        dev = (C.c_int32 * self.nranks)(*devices)
        L.check(L.load().bz_ctx_p2p_connect(self._h, blob, dev))

    def info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int32(), C.c_int64()
        L.check(L.load().bz_device_info(self._h, name, C.byref(cus), C.byref(mem)))
        return {"arch": name.value.decode(), "cus": cus.value, "mem_bytes": mem.value}

    def close(self):
        if getattr(self, "_h", None):
            L.load().bz_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def runtime_tuning() -> int:
    """Opt in to the process-wide ROCm runtime settings of bz_runtime_tuning (include/bazinga_hip.h); call before
    the process's first HIP call.  Returns the mask of variables this call set."""
    return int(L.load().bz_runtime_tuning())


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(device=int(os.environ.get("BZ_DEVICE", "0")))
    return _default_ctx


def set_default_context(ctx: Context):
    global _default_ctx
    _default_ctx = ctx


def shard_bounds(n: int, rank: int, nranks: int, align: int = 256):
    """Contiguous block partition of [0, n) over nranks, boundaries aligned to `align`
    elements.  Returns (start, stop)."""
    per = -(-n // nranks)
    per = -(-per // align) * align
    start = min(n, rank * per)
    stop = min(n, start + per)
    return start, stop


class Problem:
    """bz_problem: the lowered (f, g, c, D) with its device-resident data and solver state."""

    def __init__(self, f, g, c, D, n, ny, dtype, ctx: Context | None = None, slack: bool = False):
        self.ctx = ctx or default_context()
        self.nx, self.ny, self.dtype, self.slack = int(n), int(ny), np.dtype(dtype), bool(slack)
        self.n = self.nx + self.ny if slack else self.nx       # length of the inner decision vector
        desc, keep = lower(f, g, c, D, self.nx, self.ny, self.dtype, slack)
        h = C.c_void_p()
        L.check(L.load().bz_problem_create(self.ctx._h, C.byref(desc), C.byref(h)))
        # structured oracles: the library has copied the data.  Generic oracles: the callback thunks (and the list
        # exceptions raised inside them are parked in) must live as long as the problem
        # split-layout pairwise D (demo/obstacle.jl:151-168): the device works in the interleaved order
        from .oracles import PairwiseSet, split_permutation
        self._perm = self._iperm = None
        if isinstance(D, PairwiseSet) and D.layout == "split" and desc.f_kind != L.BZ_F_CALLBACK:
            if slack or self.nx != self.ny:
                raise ValueError("split-layout pairwise sets: c = Identity, no slack form")
            self._perm = split_permutation(self.nx)
            self._iperm = np.argsort(self._perm)
        self.generic = desc.f_kind == L.BZ_F_CALLBACK
        self._keep = keep if self.generic else None
        del keep
        self._h = h

    # -- helpers
    def _call(self, rc):
        """check a library return code; an exception raised inside an oracle callback surfaces here"""
        if self._keep is not None and self._keep[-1]:
            err = self._keep[-1][0]
            del self._keep[-1][:]
            from .oracles import CallbackError
            raise CallbackError(f"oracle callback raised {type(err).__name__}: {err}") from err      # (rc is BZ_ERR_CALLBACK)
        L.check(rc)

    def _in(self, a, n):
        v = np.ascontiguousarray(a, dtype=self.dtype)
        if v.shape != (n,):
            raise ValueError(f"expected a vector of length {n}, got shape {v.shape}")
        if self._perm is not None:
            v = np.ascontiguousarray(v[self._perm])
        return v

    def _out(self, v):
        """device order -> caller's order (split-layout pairwise sets)"""
        return v if self._iperm is None else np.ascontiguousarray(v[self._iperm])

    def set_multipliers(self, mu, y):
        mu, y = self._in(mu, self.ny), self._in(y, self.ny)
        self._call(L.load().bz_problem_set_multipliers(self._h, mu.ctypes.data, y.ctypes.data))

    def panoc_solve(self, opts: L.PanocOpts, x0):
        x0 = self._in(x0, self.n)
        out = np.empty(self.n, self.dtype)
        st = L.PanocStats()
        self._call(L.load().bz_panoc_solve(self._h, C.byref(opts), x0.ctypes.data, out.ctypes.data, C.byref(st)))
        return self._out(out), st

    def panoc_begin(self, opts: L.PanocOpts, x0):
        x0 = self._in(x0, self.n)
        self._call(L.load().bz_panoc_begin(self._h, C.byref(opts), x0.ctypes.data))

    def halo_export(self) -> bytes:
        """Row-block-sharded Stencil5ptQuadratic: this rank's halo region (64-byte IPC handle)."""
        buf = C.create_string_buffer(64)
        L.check(L.load().bz_problem_halo_export(self._h, buf))
        return buf.raw

    def halo_connect(self, prev: bytes | None, nxt: bytes | None):
        """Handles of the previous / next rank's halo regions (None at the ends of the rank order)."""
        a = C.create_string_buffer(prev, 64) if prev else None
        b = C.create_string_buffer(nxt, 64) if nxt else None
        L.check(L.load().bz_problem_halo_connect(self._h, a, b))

    def allreduce_export(self) -> bytes:
        """Row-sharded DenseAffine: this rank's all-reduce region (64-byte IPC handle)."""
        buf = C.create_string_buffer(64)
        L.check(L.load().bz_problem_allreduce_export(self._h, buf))
        return buf.raw

    def allreduce_connect(self, handles):
        """handles: every rank's 64-byte handle, in rank order."""
        blob = C.create_string_buffer(b"".join(handles), 64 * len(handles))
        L.check(L.load().bz_problem_allreduce_connect(self._h, blob))

    def panoc_step(self):
        self._call(L.load().bz_panoc_step(self._h))

    def panoc_steps(self, k: int):
        """k consecutive Base.iterate(iter, state) steps in one library call."""
        self._call(L.load().bz_panoc_steps(self._h, int(k)))

    def panoc_finish(self):
        out = np.empty(self.n, self.dtype)
        st = L.PanocStats()
        self._call(L.load().bz_panoc_finish(self._h, out.ctypes.data, C.byref(st)))
        return self._out(out), st

    def panoc_stats(self):
        st = L.PanocStats()
        L.check(L.load().bz_panoc_finish(self._h, None, C.byref(st)))
        return st

    SCALAR_NAMES = ("k", "gamma", "tau", "f_x", "g_z", "dot_grad_res", "ss_res", "stop_norm", "last_ys",
                    "lbfgs_mem", "lbfgs_H", "al_z", "f_z", "n_backtracks", "fused", "FBE")

    def panoc_scalars(self):
        buf = (C.c_double * 16)()
        L.check(L.load().bz_panoc_scalars(self._h, buf))
        return dict(zip(self.SCALAR_NAMES, list(buf)))

    def panoc_vector(self, which):
        idx = {"x": 0, "z": 1, "res": 2, "grad_x": 3, "grad_z": 4}[which]
        out = np.empty(self.n, self.dtype)
        self._call(L.load().bz_panoc_vector(self._h, idx, out.ctypes.data))
        return self._out(out)

    def alps_solve(self, aopts: L.AlpsOpts, popts: L.PanocOpts, x0, y0):
        x0, y0 = self._in(x0, self.nx), self._in(y0, self.ny)
        x = np.empty(self.nx, self.dtype)
        y, s, mu = (np.empty(self.ny, self.dtype) for _ in range(3))
        st = L.AlpsStats()
        fn = L.load().bz_als_solve if self.slack else L.load().bz_alps_solve
        self._call(fn(self._h, C.byref(aopts), C.byref(popts), x0.ctypes.data, y0.ctypes.data,
                                       x.ctypes.data, y.ctypes.data, s.ctypes.data, mu.ctypes.data, C.byref(st)))
        return self._out(x), self._out(y), self._out(s), self._out(mu), st

    def eval_al_gradient(self, x):
        x = self._in(x, self.n)
        g = np.empty(self.n, self.dtype)
        vals = (C.c_double * 3)()
        self._call(L.load().bz_eval_al_gradient(self._h, x.ctypes.data, g.ctypes.data, vals))
        return self._out(g), tuple(vals)

    def eval_prox(self, x, gamma):
        x = self._in(x, self.n)
        z = np.empty(self.n, self.dtype)
        gz = C.c_double()
        self._call(L.load().bz_eval_prox(self._h, x.ctypes.data, float(gamma), z.ctypes.data, C.byref(gz)))
        return self._out(z), gz.value

    def eval_lbfgs(self, S, Y, v):
        v = self._in(v, self.n)
        m = len(S)
        Sa = np.ascontiguousarray(S, dtype=self.dtype).reshape(m, self.n) if m else None
        Ya = np.ascontiguousarray(Y, dtype=self.dtype).reshape(m, self.n) if m else None
        d = np.empty(self.n, self.dtype)
        L.check(L.load().bz_eval_lbfgs(self._h, m, Sa.ctypes.data if m else None, Ya.ctypes.data if m else None,
                                       v.ctypes.data, d.ctypes.data))
        return d

    def profile_enable(self, on=True, period=1):
        """on: True (all categories), False, or a bitmask over _lib.KERNEL_CATEGORIES;
        period k: only every k-th launch of each enabled category is timed."""
        mask = 0xFFFF if on is True else (0 if on is False else int(on) & 0xFFFF)
        L.check(L.load().bz_profile_enable(self._h, mask | (int(period) << 16)))

    def profile_reset(self):
        L.check(L.load().bz_profile_reset(self._h))

    def profile2(self):
        """Per kernel category: timed launches / ms / bytes (the launches that carried events), all launches and
        the bytes they were designed to move since profile_reset, and the template form of the last launch."""
        out = {}
        for i, name in enumerate(L.KERNEL_CATEGORIES):
            r = L.ProfileRec()
            L.check(L.load().bz_profile_get2(self._h, i, C.byref(r)))
            out[name] = {"timed_launches": r.timed_launches, "timed_ms": r.timed_ms, "timed_bytes": r.timed_bytes,
                         "launches": r.launches, "bytes": r.bytes, "form": r.form.decode()}
        return out

    def profile(self):
        out = {}
        for i, name in enumerate(L.KERNEL_CATEGORIES):
            n, ms = C.c_int64(), C.c_double()
            L.check(L.load().bz_profile_get(self._h, i, C.byref(n), C.byref(ms)))
            out[name] = {"launches": n.value, "total_ms": ms.value}
        return out

    def close(self):
        if getattr(self, "_h", None):
            L.load().bz_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
