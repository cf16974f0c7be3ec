"""Host-side logic that needs no GPU: oracle lowering, argument errors, sharding plan,
synthetic-data generator, safeguards."""
import numpy as np
import pytest


def test_splitmix64_known_answers(bz):
    """splitmix64 reference outputs for seed 1234567 (Vigna's published test vector)."""
    s = bz.synth
    x = np.uint64(1234567)
    outs = []
    with np.errstate(over="ignore"):
        for _ in range(3):
            outs.append(int(s.splitmix64(np.array([x], dtype=np.uint64))[0]))
            x = x + np.uint64(0x9E3779B97F4A7C15)
    assert outs == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_synth_is_shard_consistent(bz):
    full = bz.synth.l1_quadratic(1000)
    a, b = bz.synth.l1_quadratic(600, start=0), bz.synth.l1_quadratic(400, start=600)
    assert np.array_equal(full["q"], np.concatenate([a["q"], b["q"]]))
    assert np.array_equal(full["b"], np.concatenate([a["b"], b["b"]]))
    assert full["q"].min() >= 0.1 and full["q"].max() <= 10.0 and np.abs(full["b"]).max() <= 10.0


@pytest.mark.parametrize("n,nranks", [(10 ** 7, 8), (10 ** 7, 3), (1000, 8), (5, 2), (10 ** 8, 8)])
def test_shard_bounds_partition(bz, n, nranks):
    prev = 0
    for r in range(nranks):
        lo, hi = bz.shard_bounds(n, r, nranks)
        assert lo == prev and lo <= hi <= n
        if hi < n:
            assert hi % 256 == 0           # 16-byte aligned vector accesses on every shard
        prev = hi
    assert prev == n


def test_lowering_kinds(bz):
    from bazinga_jl_amd.oracles import lower
    L = bz._lib
    n = 6
    q, b = np.arange(1.0, 7.0), np.ones(n)
    d, keep = lower(bz.DiagQuadratic(q, b), bz.NormL1(2.5), bz.IdentityFunction(),
                    bz.ClosedSet(bz.IndBox(-1.0, 1.0)), n, n, np.float64)
    assert (d.dtype, d.f_kind, d.g_kind, d.c_kind, d.D_kind) == (L.BZ_F64, L.BZ_F_DIAG_QUADRATIC, L.BZ_G_NORM_L1,
                                                                 L.BZ_C_IDENTITY, L.BZ_D_BOX)
    assert d.g_lambda == 2.5 and d.D_lo == -1.0 and d.D_hi == 1.0 and d.n == n and d.ny == n
    d, _ = lower(bz.Zero(), bz.IndFree(), bz.IdentityFunction(), bz.ZeroSet(), n, n, np.float32)
    assert (d.dtype, d.f_kind, d.g_kind, d.D_kind) == (L.BZ_F32, L.BZ_F_ZERO, L.BZ_G_ZERO, L.BZ_D_ZERO)
    lo = -np.ones(n)
    d, keep = lower(bz.Zero(), bz.IndBox(lo, 2.0), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float64)
    assert d.g_kind == L.BZ_G_IND_BOX and d.g_lo_vec and not d.g_hi_vec and d.g_hi == 2.0


def test_lowering_of_the_next_row_kinds(bz):
    """pairwise sets, L0 / Lp oracles, stencil and dense operands, the slack flag."""
    from bazinga_jl_amd.oracles import lower
    L = bz._lib
    n = 8
    for kind, code in (("vc", L.BZ_D_VC_PAIRS), ("cc", L.BZ_D_CC_PAIRS), ("eitheror", L.BZ_D_EITHEROR_PAIRS),
                       ("xor", L.BZ_D_XOR_PAIRS)):
        d, _ = lower(bz.Zero(), bz.Zero(), bz.IdentityFunction(), bz.PairwiseSet(kind), n, n, np.float64)
        assert d.D_kind == code
    assert bz.VanishingConstraintPairs().kind == "vc" and bz.XorPairs().kind == "xor"
    with pytest.raises(ValueError):
        bz.PairwiseSet("nand")
    with pytest.raises(ValueError):
        lower(bz.Zero(), bz.Zero(), bz.IdentityFunction(), bz.PairwiseSet("cc"), 7, 7, np.float64)     # odd ny
    u = np.linspace(0.0, 1.0, n)
    d, _ = lower(bz.Zero(), bz.NormL0Box(0.3, u=u), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float64)
    assert d.g_kind == L.BZ_G_NORM_L0_BOX and d.g_lambda == 0.3 and d.g_u
    d, _ = lower(bz.Zero(), bz.NormLpPowerBox(0.5, 0.8, u=u), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float64)
    assert d.g_kind == L.BZ_G_NORM_LP_BOX and d.g_p == 0.5 and d.g_lambda == 0.8
    d, _ = lower(bz.Stencil5ptQuadratic(2, 4, np.ones(n)), bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float64)
    assert d.f_kind == L.BZ_F_STENCIL5 and (d.f_grid_nx, d.f_grid_ny) == (2, 4)
    A = np.ones((3, n), np.float32)
    d, _ = lower(bz.Zero(), bz.NormL1(1.0), bz.DenseAffine(A, np.zeros(3, np.float32)), bz.ZeroSet(), n, 3, np.float32)
    assert d.c_kind == L.BZ_C_DENSE_AFFINE and d.ny == 3 and d.c_A and d.c_b
    d, _ = lower(bz.Zero(), bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float64, slack=True)
    assert d.slack == 1


def test_unsupported_oracles_raise(bz):
    from bazinga_jl_amd.oracles import lower

    class Rosenbrock:     # arbitrary closures cannot run on the device
        pass
    n = 4
    with pytest.raises(bz.UnsupportedOracle):
        lower(Rosenbrock(), bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float64)
    with pytest.raises(bz.UnsupportedOracle):
        lower(bz.Zero(), Rosenbrock(), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float64)
    with pytest.raises(bz.UnsupportedOracle):
        lower(bz.Zero(), bz.Zero(), Rosenbrock(), bz.FreeSet(), n, n, np.float64)
    with pytest.raises(bz.UnsupportedOracle):
        lower(bz.Zero(), bz.Zero(), bz.IdentityFunction(), Rosenbrock(), n, n, np.float64)
    with pytest.raises(bz.UnsupportedOracle):
        lower(bz.Zero(), bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), n, n, np.float16)
    with pytest.raises(ValueError):
        lower(bz.Zero(), bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), n, n + 1, np.float64)


def test_constructor_errors_mirror_reference(bz):
    with pytest.raises(ValueError):         # normL1Nonneg.jl:15-16
        bz.NormL1Nonneg(-1.0)
    with pytest.raises(ValueError):         # normL1Box.jl:18-23
        bz.NormL1Box(1.0, u=np.array([1.0, -1.0]))
    with pytest.raises(ValueError):
        bz.NormL1Box(-1.0, u=np.array([1.0]))
    mu, y, x = np.array([1.0, 0.0]), np.zeros(2), np.zeros(2)
    with pytest.raises(ValueError, match="must be positive"):     # auglagfun.jl:33-34
        bz.AugLagFun(bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), mu, y, x)
    al = bz.AugLagFun(bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), np.ones(2), y, x)
    with pytest.raises(ValueError, match="must be positive"):     # auglagfun.jl:92-93
        bz.AugLagUpdate(al, mu, y)


def test_safeguards_match_oracle(bz, ref):
    rng = np.random.default_rng(0)
    for dt in (np.float64, np.float32):
        cx, s = rng.standard_normal(50).astype(dt) * 30, rng.standard_normal(50).astype(dt)
        for objx in (-3.0, 0.5, 1e5):
            a, b = np.empty(50, dt), np.empty(50, dt)
            bz.default_penalty_parameter(a, cx, s, objx)
            ref.default_penalty_parameter(b, cx, s, objx)
            assert np.array_equal(a, b) and a.min() >= 1e-8 and a.max() <= 1e8
    y1 = np.array([1e30, -1e30, 3.0])
    y2 = y1.copy()
    bz.default_dual_safeguard(y1)
    ref.default_dual_safeguard(y2)
    assert np.array_equal(y1, y2) and np.array_equal(y1, [1e20, -1e20, 3.0])


def test_auglag_update_scalars(bz, ref):
    rng = np.random.default_rng(1)
    n = 33
    mu, y, x = rng.uniform(0.1, 1, n), rng.standard_normal(n), np.zeros(n)
    a = bz.AugLagFun(bz.Zero(), bz.IdentityFunction(), bz.FreeSet(), mu.copy(), y.copy(), x)
    b = ref.AugLagFun(ref.Zero(), ref.IdentityFunction(), ref.FreeSet(), mu.copy(), y.copy(), x)
    assert np.array_equal(a.muy, b.muy) and a.musqy == b.musqy
    mu2, y2 = mu * 0.5, y + 1
    bz.AugLagUpdate(a, mu2, y2)
    ref.AugLagUpdate(b, mu2, y2)
    assert np.array_equal(a.muy, b.muy) and a.musqy == b.musqy


def test_markstein_division_by_an_invariant_divisor_is_correctly_rounded(tmp_path):
    """`div_u` in bz_kernels.h replaces a / b (b fixed over a launch, rb = RN(1/b)) by q = a*rb followed by two
    steps q <- q + RN(a - b q)*rb with the remainder exact in an fma.  The same sequence on the CPU (libm fma)
    against the compiler's division: 4e7 random quotients over 16 orders of magnitude of b, plus divisors with
    extreme significands (all ones, one above a power of two), must agree bit for bit."""
    import subprocess
    src = tmp_path / "divu.c"
    src.write_text(r'''
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t nxt(void) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double u01(void) { return (double)(nxt() >> 11) * 0x1.0p-53; }
static double div_u(double a, double b, double rb) {
    double q = a * rb;
    double e = fma(-b, q, a);
    q = fma(e, rb, q);
    e = fma(-b, q, a);
    return fma(e, rb, q);
}
int main(void) {
    long bad = 0, n = 0;
    double special[8];
    uint64_t bits;
    bits = 0x3FFFFFFFFFFFFFFFull; memcpy(&special[0], &bits, 8);   /* significand all ones */
    bits = 0x3FF0000000000001ull; memcpy(&special[1], &bits, 8);   /* one ulp above 1 */
    bits = 0x3FB999999999999Aull; memcpy(&special[2], &bits, 8);   /* 0.1 */
    special[3] = 3.0; special[4] = 0.7; special[5] = 1e-8; special[6] = 1e8; special[7] = 1.0 / 3.0;
    for (int k = 0; k < 4008; ++k) {
        double b = k < 8 ? special[k] : pow(10.0, -8.0 + 16.0 * u01()) * (1.0 + u01());
        double rb = 1.0 / b;
        for (int i = 0; i < 10000; ++i) {
            double a = (2.0 * u01() - 1.0) * pow(10.0, -12.0 + 16.0 * u01());
            if (i == 0) a = 0.0;
            double q1 = a / b, q2 = div_u(a, b, rb);
            if (memcmp(&q1, &q2, 8) != 0) ++bad;
            ++n;
        }
    }
    printf("%ld %ld\n", n, bad);
    return 0;
}
''')
    exe = tmp_path / "divu"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == 40_080_000 and int(out[1]) == 0


def test_profile_forms_are_reproducible_from_kernel_names():
    """bench.py attaches PMC traffic to a roofline object by the template FORM the library reports; the summariser derives
    that form from rocprof's kernel names.  Every committed entry must be the image of its own kernel name, the forms of
    one workload must be distinct per instantiation, and the entries must belong to one build of the kernel sources."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import summarize_profiles as sp
    d = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    assert len(d["entries"]) > 30
    seen = {}
    for e in d["entries"]:
        assert sp.form_of(e["kernel"]) == e["form"], e
        key = (e["workload"], e["n"], e["form"])
        assert key not in seen or seen[key] == e["kernel"], f"two instantiations share the form {key}"
        seen[key] = e["kernel"]
    assert len({e["lib_sources_sha"] for e in d["entries"]}) == 1
    # the instantiations the solver names explicitly
    assert sp.form_of("bz::k_stencil_update_c<double, 5, true, true>") == "k_stencil_update_c<FULL=1,NT=1>"
    assert sp.form_of("bz::k_compact_xd<float, 5, true, false>") == "k_compact_xd<FULL=1,NT=0>"
    assert sp.form_of("bz::k_stencil_fb<double, true>") == "k_stencil_fb<NT=1>"
    assert sp.form_of("bz::k_fused_compact<double, 5, true, true, true, 2, 2, 0, 39>") == "k_fused_compact<XR=2,UNI=2,NT=1,TRIAL=0,FAM=39>"
    assert sp.form_of("bz::k_fused_compact<double, 5, true, true, true, 2, 2, 0, 35>") == "k_fused_compact<XR=2,UNI=2,NT=1,TRIAL=0>"


def test_dense_ring_kernel_has_no_spill_code_and_exact_waits():
    """k_dense_fused loads its register ring with inline asm (the compiler's own wait bookkeeping drained the prefetch): a
    spilled ring register would be stored before its data has arrived.  tools/check_dense_ring.py compiles every
    instantiation for gfx950 (no GPU needed) and checks for spill code and for the ring waits' counts."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dense_ring.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" 0 findings") == 5, r.stdout
