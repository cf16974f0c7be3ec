"""Random-shape sweep of the kernel-level parity tests (stencil / dense AL gradient, pairwise sets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bazinga_jl_amd as bz
from oracle import bazinga_ref as ref
import test_gpu_parity as T

def raw(f):
    return getattr(f, "__wrapped__", f)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
def run(name, fn, *args):
    global bad
    try:
        fn(bz, ref, *args)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("FAIL", name, args, repr(e)[:300], flush=True)

for i in range(60):
    nx = int(rng.integers(1, 300)); ny = 2 * int(rng.integers(1, 400))
    run("stencil", raw(T.test_stencil_al_gradient_bit_exact), (nx, ny))
for i in range(60):
    ny = int(rng.integers(1, 300)); n = int(rng.integers(1, 1500))
    run("dense64", raw(T.test_dense_al_gradient), (ny, n), np.float64)
    n32 = int(rng.choice([n, 64 * max(1, n // 64), 4 * max(1, n // 4)]))
    run("dense32", raw(T.test_dense_al_gradient), (ny, n32), np.float32)
for i in range(40):
    n = 2 * int(rng.integers(1, 200000))
    run("pairs", raw(T.test_pairwise_sets_al_gradient_bit_exact), n, str(rng.choice(["vc", "cc", "eitheror", "xor"])))
for i in range(40):
    n = int(rng.integers(1, 300000))
    run("algrad", raw(T.test_al_gradient_bit_exact), n, str(rng.choice(["box", "free", "zero"])))
    run("prox", raw(T.test_prox_bit_exact), n, str(rng.choice(["l1", "nonneg", "l1box", "l0box", "indbox", "zero"])))
print("done; failures:", bad)
for i in range(25):
    nx = int(rng.integers(2, 120)); ny = 2 * int(rng.integers(1, 120))
    run("stencil-panoc", raw(T.test_stencil_panoc_iterates_match_oracle), (nx, ny), 15, str(rng.choice(["default", "two-loop"])))
for i in range(12):
    n = int(rng.integers(1, 50000))
    run("panoc", raw(T.test_panoc_iterates_match_oracle), n, str(rng.choice(["box", "free"])),
        str(rng.choice(["default", "two-loop"])))
    run("compact", raw(T.test_compact_lbfgs_matches_compact_oracle), n, bool(rng.integers(0, 2)))
print("done (iterates); failures:", bad)
