"""Synthetic inputs for the BASELINE configs.

Language-independent uniform stream so C/HIP/Python/Julia produce identical bits:

    u_k(i) = (splitmix64(seed + k*2^60 + i) >> 11) * 2^-53      seed = 20241004

(the top 53 bits, so the conversion to double is exact in every language).
"""
from __future__ import annotations

import numpy as np

SEED = 20241004
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per uint64 input (vectorised, wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(k: int, n: int, start: int = 0, seed: int = SEED) -> np.ndarray:
    """u_k(start .. start+n-1) as float64 in [0,1)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) + (np.uint64(k) << np.uint64(60)) + np.uint64(start)
        idx = base + np.arange(n, dtype=np.uint64)
    z = splitmix64(idx)
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def l1_quadratic(n: int, start: int = 0, dtype=np.float64):
    """cfg 2 / cfg 5 data for elements [start, start+n): q_i = 0.1 + 9.9 u1, b_i = 10(2 u2 - 1).
    f(x)=sum x(0.5 q x - b), g = 2.5||x||_1, c = I, D = Box[-1,1]."""
    q = (0.1 + 9.9 * uniform(1, n, start)).astype(dtype)
    b = (10.0 * (2.0 * uniform(2, n, start) - 1.0)).astype(dtype)
    return {"q": q, "b": b, "lam": 2.5, "lo": -1.0, "hi": 1.0}


def obstacle_grid(nx: int = 2048, ny: int | None = None, dtype=np.float64, load: float = 1.0):
    """cfg 3: 5-pt Laplacian QP on an nx-by-ny grid, b = load*h^2 (SURVEY: load = +1; load = -1
    pushes the membrane onto the obstacle so the constraint is active), obstacle
    psi_ij = 0.05 - 0.5((i h - .5)^2 + (j h - .5)^2), h = 1/(nx+1), i,j = 1..n;
    D = Box[psi, +inf), x0 = max(0, psi)."""
    ny = nx if ny is None else ny
    h = 1.0 / (nx + 1)
    i = (np.arange(1, nx + 1) * h - 0.5) ** 2
    j = (np.arange(1, ny + 1) * (1.0 / (ny + 1)) - 0.5) ** 2
    psi = (0.05 - 0.5 * (i[:, None] + j[None, :])).reshape(-1).astype(dtype)
    b = np.full(nx * ny, load * h * h, dtype=dtype)
    return {"nx": nx, "ny": ny, "b": b, "psi": psi, "x0": np.maximum(0, psi).astype(dtype)}


def basis_pursuit(ny: int = 8192, n: int = 65536, dtype=np.float32, density: float = 0.01):
    """cfg 4: A_ij = (2 u3 - 1)/sqrt(ny) row-major ny-by-n, xtrue density-sparse +-1, b = A xtrue."""
    A = np.empty((ny, n), dtype=dtype)
    rows = max(1, (1 << 24) // n)
    s = 1.0 / np.sqrt(ny)
    for r0 in range(0, ny, rows):
        r1 = min(ny, r0 + rows)
        A[r0:r1] = ((2.0 * uniform(3, (r1 - r0) * n, r0 * n) - 1.0) * s).reshape(r1 - r0, n).astype(dtype)
    u = uniform(4, n)
    sgn = np.where(uniform(5, n) < 0.5, -1.0, 1.0)
    xtrue = np.where(u < density, sgn, 0.0).astype(dtype)
    b = (A.astype(np.float64) @ xtrue.astype(np.float64)).astype(dtype)
    return {"A": A, "b": b, "xtrue": xtrue}
