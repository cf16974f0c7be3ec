"""Runs the randomised ALPS parity case of tests/test_gpu_parity.py over many more seeds than the suite does
(development aid: hunts for rare branch combinations — backtracks, gamma halvings, skipped pairs, resets)."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bazinga_jl_amd as bz
from oracle import bazinga_ref as ref
import test_gpu_parity as T

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
fn = T.test_randomised_kinds_alps_parity
fn = getattr(fn, "__wrapped__", fn)
for seed in range(lo, hi):
    for form in ("two-loop", "compact"):
        try:
            # the suite's seeds are 1000 + s for s < 12: shift far away from them
            fn(bz, ref, seed, form)
        except AssertionError as e:
            bad.append((seed, form, str(e)[:200]))
            print("FAIL", seed, form, str(e)[:200], flush=True)
        except Exception as e:      # noqa: BLE001
            bad.append((seed, form, repr(e)[:200]))
            print("ERROR", seed, form, repr(e)[:200], flush=True)
    if seed % 20 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done", hi - lo, "seeds x 2 forms;", len(bad), "failures")
