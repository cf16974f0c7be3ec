// bz_kernels.h — hand-written gfx950 kernels for the PANOCplus hot path.
//
// Every kernel here is a streaming, HBM-bound vector pass (arithmetic intensity
// 0.1-0.2 flop/B), so the design rules are the ones for bandwidth, not MFMA:
//   * 16 B per lane per access (double2 / float4), unit stride, 256-thread blocks,
//     a grid of <= 2048 blocks that grid-strides the vector;
//   * reductions: per-thread fp64 accumulators -> wave64 shuffle tree -> LDS across
//     the 4 waves -> one partial per block.  The NEXT kernel (or the collect
//     kernel) folds the <= 2048 partials in a fixed order, so every scalar is
//     bit-reproducible and no atomics / fences / grid barriers are needed;
//   * coefficients of the L-BFGS two-loop (alpha_i, beta_i) never visit the host:
//     each axpy+dot kernel derives its coefficient from the previous kernel's
//     partials and leaves alpha_i in a small device bank.
//
// Per-element arithmetic mirrors the CPU oracle operation for operation (build with
// -ffp-contract=off), so element-wise outputs are bit-identical to oracle/.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/bazinga_hip.h"

namespace bz {

constexpr int BLOCK   = 256;          // 4 waves of 64
constexpr int WAVES   = BLOCK / 64;
constexpr int PSTRIDE = 2048;         // max grid = partials per reduction slot

// where a consumer finds the pieces of a global scalar: `count` doubles `stride`
// apart (block partials on one GPU, per-rank packs after an all-gather).
struct ScalarSrc {
    const double* p;
    int count;
    int stride;
};

template <class T> struct PackN;
template <> struct PackN<double> { static constexpr int N = 2; };
template <> struct PackN<float>  { static constexpr int N = 4; };

template <class T> struct alignas(16) Pack {
    T v[PackN<T>::N];
};

// problem data every element-wise kernel may need (device pointers)
template <class T> struct ElemParams {
    int f_kind, g_kind, D_kind;
    int uni;               // 0: mu and mu*y are streamed; 1: every mu[i] is mu_uniform: a number, not a stream; 2: and mu*y = 0
                           // (k_muy's probe at every AugLagUpdate!; same operands, same operations, same bits)
    const T* q;
    const T* b;
    const T* mu;
    const T* muy;
    T g_lambda;
    T g_p;                 // NormLpPower*: exponent p in (0,1); g_lambda holds alpha
    const T* g_u;
    T g_lo, g_hi;
    const T* g_lo_vec;
    const T* g_hi_vec;
    T D_lo, D_hi;
    const T* D_lo_vec;
    const T* D_hi_vec;
    T mu_uniform;          // the value of every mu[i] when the penalties are uniform (k_uniform_probe), else unused
};

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // lane 0 holds the sum
}
__device__ __forceinline__ double nanmax(double a, double b) {
    return (a > b || a != a) ? a : b;   // NaN-propagating max
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_down(v, o, 64));
    return v;
}

// fold a ScalarSrc; every thread of every block gets the same bits.
// is_max: fold with max instead of sum.
__device__ __forceinline__ double fold_src(ScalarSrc s, bool is_max, double* sh /*WAVES*/) {
    double v = 0.0;
    if (is_max) {
        for (int i = threadIdx.x; i < s.count; i += BLOCK) v = nanmax(v, s.p[(size_t)i * s.stride]);
        v = wave_max(v);
    } else {
        for (int i = threadIdx.x; i < s.count; i += BLOCK) v += s.p[(size_t)i * s.stride];
        v = wave_sum(v);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = is_max ? nanmax(nanmax(sh[0], sh[1]), nanmax(sh[2], sh[3]))
                      : ((sh[0] + sh[1]) + (sh[2] + sh[3]));
    __syncthreads();
    return t;
}

// the same fold done by ONE wave, bit for bit: lane l plays threads l, l+64, l+128, l+192 of fold_src
// (unit stride).  Lets a block fold several scalars at once, one per wave.  Every lane gets the result.
__device__ __forceinline__ double fold_wave(const double* p, int count, bool is_max) {
    const int l = threadIdx.x & 63;
    // all loads first (one memory latency instead of count/256 of them); missing entries read as +0.0,
    // which is neutral for both folds (the accumulators start at +0.0 and never become -0.0)
    double v[PSTRIDE / BLOCK][WAVES];
#pragma unroll
    for (int k = 0; k < PSTRIDE / BLOCK; ++k)
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const int i = k * BLOCK + w * 64 + l;
            v[k][w] = i < count ? p[i] : 0.0;
        }
    double a[WAVES] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < PSTRIDE / BLOCK; ++k)
#pragma unroll
        for (int w = 0; w < WAVES; ++w) a[w] = is_max ? nanmax(a[w], v[k][w]) : a[w] + v[k][w];
    // the four wave reductions as one transposed butterfly (9 shuffles instead of 24): lanes split the four
    // values at the stages that pair l with l ^ 32 and l ^ 16, each then runs the rest of its value's tree —
    // the pairs (l, l ^ 32), (l, l ^ 16), ..., (l, l ^ 1) of wave_sum / wave_max, so the same bits
    auto comb = [&](double x, double y) { return is_max ? nanmax(x, y) : x + y; };
    const bool h5 = l & 32, h4 = l & 16;
    const double k0 = h5 ? a[2] : a[0], k1 = h5 ? a[3] : a[1];
    const double s0 = h5 ? a[0] : a[2], s1 = h5 ? a[1] : a[3];
    const double b0 = comb(k0, __shfl_xor(s0, 32, 64)), b1 = comb(k1, __shfl_xor(s1, 32, 64));
    double c = comb(h4 ? b1 : b0, __shfl_xor(h4 ? b0 : b1, 16, 64));      // lanes (h5, h4) hold a[2 h5 + h4]
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) c = comb(c, __shfl_xor(c, o, 64));
    const double t01 = comb(c, __shfl_xor(c, 16, 64));                     // a0 + a1 | a2 + a3
    return comb(t01, __shfl_xor(t01, 32, 64));                             // (a0 + a1) + (a2 + a3), in every lane
}

// block-reduce K accumulators and write one partial per slot.
// maxmask bit k set -> slot k is a max.
// (vb: the block's number within its section of the launch, bz_for_chunks_v; default: the launch is one section)
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&acc)[K], unsigned maxmask,
                                                   double* parts, int first_slot, int vb = -1) {
    __shared__ double sh[WAVES][K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double v = ((maxmask >> k) & 1u) ? wave_max(acc[k]) : wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        int k = threadIdx.x;
        double t = ((maxmask >> k) & 1u)
                       ? nanmax(nanmax(sh[0][k], sh[1][k]), nanmax(sh[2][k], sh[3][k]))
                       : ((sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]));
        parts[(size_t)(first_slot + k) * PSTRIDE + (vb >= 0 ? vb : (int)blockIdx.x)] = t;
    }
}

// The same for exactly 32 accumulators, as a transposed butterfly: at the stage that pairs lane l with l ^ 32 each
// lane keeps half of its slots and hands the other half to its partner, and so on down to one slot per lane
// pair — 16 + 8 + 4 + 2 + 1 + 1 = 32 shuffles per lane instead of 32 x 6.  Every slot still goes through the
// tree (l, l ^ 32), (l, l ^ 16), ..., (l, l ^ 1) that wave_sum / wave_max build with __shfl_down, and + and
// nanmax are commutative in value, so the bits are those of block_reduce_store<32>.  (Measured on
// k_fused_compact: the 32-scalar epilogue took 6.5-7 us of a 24 us pass at n = 1.25e6.)
__device__ __forceinline__ double bf_combine(double a, double b, unsigned maxmask, int slot) {
    const double s = a + b, m = nanmax(a, b);
    return ((maxmask >> slot) & 1u) ? m : s;
}
__device__ __forceinline__ void block_reduce_store32(double (&acc)[32], unsigned maxmask, double* parts,
                                                     int first_slot) {
    __shared__ double sh[WAVES][32];
    const int lane = threadIdx.x & 63;
    int base = 0;                       // first slot this lane still holds
    double a16[16], a8[8], a4[4], a2[2], a1;
    {
        const bool hi = lane & 32;
        base += hi ? 16 : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double keep = hi ? acc[j + 16] : acc[j], send = hi ? acc[j] : acc[j + 16];
            a16[j] = bf_combine(keep, __shfl_xor(send, 32, 64), maxmask, base + j);
        }
    }
    {
        const bool hi = lane & 16;
        base += hi ? 8 : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double keep = hi ? a16[j + 8] : a16[j], send = hi ? a16[j] : a16[j + 8];
            a8[j] = bf_combine(keep, __shfl_xor(send, 16, 64), maxmask, base + j);
        }
    }
    {
        const bool hi = lane & 8;
        base += hi ? 4 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double keep = hi ? a8[j + 4] : a8[j], send = hi ? a8[j] : a8[j + 4];
            a4[j] = bf_combine(keep, __shfl_xor(send, 8, 64), maxmask, base + j);
        }
    }
    {
        const bool hi = lane & 4;
        base += hi ? 2 : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double keep = hi ? a4[j + 2] : a4[j], send = hi ? a4[j] : a4[j + 2];
            a2[j] = bf_combine(keep, __shfl_xor(send, 4, 64), maxmask, base + j);
        }
    }
    {
        const bool hi = lane & 2;
        base += hi ? 1 : 0;
        const double keep = hi ? a2[1] : a2[0], send = hi ? a2[0] : a2[1];
        a1 = bf_combine(keep, __shfl_xor(send, 2, 64), maxmask, base);
    }
    a1 = bf_combine(a1, __shfl_xor(a1, 1, 64), maxmask, base);
    if ((lane & 1) == 0) sh[threadIdx.x >> 6][base] = a1;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        const double t = ((maxmask >> k) & 1u)
                             ? nanmax(nanmax(sh[0][k], sh[1][k]), nanmax(sh[2][k], sh[3][k]))
                             : ((sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]));
        parts[(size_t)(first_slot + k) * PSTRIDE + blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// 16-byte loads / stores with a scalar tail
// ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ Pack<T> ld(const T* __restrict__ p, int64_t i0, int cnt) {
    constexpr int N = PackN<T>::N;
    Pack<T> r;
    if (cnt == N) {
        r = *reinterpret_cast<const Pack<T>*>(p + i0);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) r.v[e] = (e < cnt) ? p[i0 + e] : T(1);
    }
    return r;
}
template <class T>
__device__ __forceinline__ void st(T* __restrict__ p, int64_t i0, int cnt, const Pack<T>& r) {
    constexpr int N = PackN<T>::N;
    if (cnt == N) {
        *reinterpret_cast<Pack<T>*>(p + i0) = r;
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e)
            if (e < cnt) p[i0 + e] = r.v[e];
    }
}
// non-temporal forms (streamed once per launch: do not displace what the caches hold)
template <class T> struct PackVec;
template <> struct PackVec<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct PackVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <class T, bool NT>
__device__ __forceinline__ Pack<T> ldp(const T* __restrict__ p, int64_t i0, int cnt) {
    if constexpr (!NT) return ld(p, i0, cnt);
    constexpr int N = PackN<T>::N;
    Pack<T> r;
    if (cnt == N) {
        typename PackVec<T>::type v = __builtin_nontemporal_load(reinterpret_cast<const typename PackVec<T>::type*>(p + i0));
#pragma unroll
        for (int e = 0; e < N; ++e) r.v[e] = v[e];
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) r.v[e] = (e < cnt) ? p[i0 + e] : T(1);
    }
    return r;
}
template <class T, bool NT>
__device__ __forceinline__ void stp(T* __restrict__ p, int64_t i0, int cnt, const Pack<T>& r) {
    if constexpr (!NT) { st(p, i0, cnt, r); return; }
    constexpr int N = PackN<T>::N;
    if (cnt == N) {
        typename PackVec<T>::type v;
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = r.v[e];
        __builtin_nontemporal_store(v, reinterpret_cast<typename PackVec<T>::type*>(p + i0));
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e)
            if (e < cnt) p[i0 + e] = r.v[e];
    }
}
// full packs addressed as (uniform base pointer, 32-bit byte offset): the offset is ONE vector register
// shared by every stream of a kernel, the bases stay in scalar registers (saddr addressing)
template <class T, bool NT>
__device__ __forceinline__ Pack<T> ldo(const T* __restrict__ base, unsigned byte_off) {
    using V = typename PackVec<T>::type;
    const V* p = reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off);
    V v;
    if constexpr (NT) v = __builtin_nontemporal_load(p); else v = *p;
    Pack<T> r;
#pragma unroll
    for (int e = 0; e < PackN<T>::N; ++e) r.v[e] = v[e];
    return r;
}
template <class T, bool NT>
__device__ __forceinline__ void sto(T* __restrict__ base, unsigned byte_off, const Pack<T>& r) {
    using V = typename PackVec<T>::type;
    V v;
#pragma unroll
    for (int e = 0; e < PackN<T>::N; ++e) v[e] = r.v[e];
    V* p = reinterpret_cast<V*>(reinterpret_cast<char*>(base) + byte_off);
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <class T> __device__ __forceinline__ Pack<T> splat(T s) {
    Pack<T> r;
#pragma unroll
    for (int e = 0; e < PackN<T>::N; ++e) r.v[e] = s;
    return r;
}

// canonical element->thread map shared by every kernel, so that a quantity summed
// by two different kernels is summed in the same order (fused == unfused, bitwise):
// chunk c (PackN elements) belongs to thread c mod (grid*BLOCK), chunks in increasing order.
// The body is instantiated twice: for full chunks `cnt` is the compile-time PackN (no tail branches,
// straight-line loads), and once for the ragged last chunk of the vector, which is also the last chunk
// of the thread that owns it.
template <class T, class F>
__device__ __forceinline__ void bz_for_chunks(int64_t n, F&& f) {
    constexpr int N = PackN<T>::N;
    const int64_t nfull = n / N;
    const int64_t st = (int64_t)gridDim.x * BLOCK;
    int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (; c < nfull; c += st) f(c * N, std::integral_constant<int, N>{});
    if (c == nfull && nfull * N < n) f(c * N, (int)(n - nfull * N));
}
// the same map for a SECTION of a launch that plays a grid of `vg` blocks of its own, this block being number `vb` of it
// (k_dense_head / k_dense_tail: several short element-wise kernels as sections of one launch, each with the chunk map —
// and therefore the partial sums — it has as a launch of its own)
template <class T, class F>
__device__ __forceinline__ void bz_for_chunks_v(int64_t n, int vb, int vg, F&& f) {
    constexpr int N = PackN<T>::N;
    const int64_t nfull = n / N;
    const int64_t st = (int64_t)vg * BLOCK;
    int64_t c = (int64_t)vb * BLOCK + threadIdx.x;
    for (; c < nfull; c += st) f(c * N, std::integral_constant<int, N>{});
    if (c == nfull && nfull * N < n) f(c * N, (int)(n - nfull * N));
}

// ---------------------------------------------------------------------------
// element-wise oracle arithmetic (mirrors oracle/bazinga_ref.py and oracle/c)
// ---------------------------------------------------------------------------
// max(0, x) / min(0, x) as Julia's: NaN propagates, and the zero returned for x = -0.0 / +0.0 is the literal's
template <class T> __device__ __forceinline__ T max0(T x) { return (x > T(0) || x != x) ? x : T(0); }
template <class T> __device__ __forceinline__ T min0(T x) { return (x < T(0) || x != x) ? x : T(0); }

// The pairwise sets: this element's component of the projection of the pair (x1, x2); `pos` says which
// component this element is (0: x1 = t, x2 = tp ; 1: x1 = tp, x2 = t).
template <class T> __device__ __forceinline__ T proj_pair(int kind, T t, T tp, int pos) {
    // Each kind as ONE keep-or-zero decision per component, built from ordered compares of the pair (the reference's
    // branch ladders, flattened: every path returns either the component itself or the literal zero, so the ladder is a
    // predicate).  A NaN component fails every ordered compare exactly as it falls through the reference's ifs.
    const T x1 = pos ? tp : t, x2 = pos ? t : tp;
    bool keep;
    if (kind == BZ_D_VC_PAIRS) {              // vanishingConstraints.jl:27-46  (x1 >= 0, x1*x2 >= 0)
        // x1 <= 0 -> (0, x2) ; x2 >= 0 -> (x1, x2) ; x1 + x2 > 0 -> (x1, 0) ; else (the tie x1 + x2 = 0 too) (0, x2)
        const bool a = x1 <= T(0), b = x2 >= T(0), c = x1 + x2 > T(0);
        keep = pos ? (a || b || !c) : (!a && (b || c));
    } else if (kind == BZ_D_CC_PAIRS) {       // complementarityConstraints.jl:8-20
        // both > 0: the larger one stays (x1 on a tie) ; otherwise max(0, .) of each: a component stays iff it is not
        // <= 0 (NaN stays) and the other one does not beat it
        keep = pos ? (!(x2 <= T(0)) && !(x1 >= x2)) : (!(x1 <= T(0)) && !(x2 > x1));
    } else if (kind == BZ_D_EITHEROR_PAIRS) { // orConstraints.jl:7-17
        // both < 0: the larger one is zeroed (x2 on a tie)
        const bool both = x1 < T(0) && x2 < T(0), g = x1 > x2;
        keep = pos ? !(both && !g) : !(both && g);
    } else {                                  // XOR, orConstraints.jl:24-36
        // x1 x2 > 0 (same sign, product not underflown): x1 > x2 -> (max(0,x1), min(0,x2)) else (min(0,x1), max(0,x2)):
        // of two positives the smaller is zeroed, of two negatives the larger (x1 on a tie of positives, x2 of negatives)
        const bool same = x1 * x2 > T(0), g = x1 > x2;
        keep = pos ? !(same && (g == (x2 > T(0)))) : !(same && (g != (x1 > T(0))));
    }
    return keep ? t : T(0);
}

// tp / pos: the pair partner's argument and this element's position, used by the pairwise kinds only
template <class T> __device__ __forceinline__ T proj_D(int kind, T t, T lo, T hi, T tp = T(0), int pos = 0) {
    // src/projections/{zeroSet,freeSet,indicatorSet}.jl
    if (kind == BZ_D_ZERO) return T(0);
    if (kind == BZ_D_FREE) return t;
    if (kind >= BZ_D_VC_PAIRS) return proj_pair(kind, t, tp, pos);
    return t < lo ? lo : (t > hi ? hi : t);   // IndBox prox: if x<lb lb elseif x>ub ub else x
}

// min(max(v, lo), hi) on the min/max units (2 instructions; the compare-and-select form is 6 in fp64).  For
// lo <= hi it is the value `v < lo ? lo : (v > hi ? hi : v)` for every non-NaN v (up to the sign of a zero
// result when a bound is itself a zero); a NaN v comes out as `lo`, so this form is only used where the
// result is immediately subtracted from v (v - clamp(v) is NaN either way).
__device__ __forceinline__ double clamp_mm(double v, double lo, double hi) { return __builtin_fmin(__builtin_fmax(v, lo), hi); }
__device__ __forceinline__ float clamp_mm(float v, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(v, lo), hi); }

// a / b for a divisor b that is uniform over the launch, given rb = RN(1/b) (one true division per thread):
// two Markstein refinement steps, q <- q + RN(a - b q) rb with the remainder exact in an fma.  After the
// first step q is within one ulp of a/b; with rb the correctly rounded reciprocal the second step then
// returns the correctly rounded quotient (Markstein 1990, Thm. 8.5 in Muller et al.'s Handbook) — the bits of
// the hardware division sequence (v_div_scale/rcp/5 fma/v_div_fmas/v_div_fixup) in half the instructions.
// Holds for finite a and quotients in the normal range, which is where the solver works (an infinite a gives
// NaN instead of inf: either way the iterate is lost).  The forms with and without it are compared bit for
// bit in test_history_as_iterates_and_lazy_z_are_bitwise_neutral.
__device__ __forceinline__ double div_u(double a, double b, double rb) {
    double q = a * rb;
    double e = __builtin_fma(-b, q, a);
    q = __builtin_fma(e, rb, q);
    e = __builtin_fma(-b, q, a);
    return __builtin_fma(e, rb, q);
}
__device__ __forceinline__ float div_u(float a, float b, float) { return a / b; }

template <class T> struct ALOut {
    T grad, fterm, pterm;
};

// one element of gradient!(dlx, al, x)  (src/utilities/auglagfun.jl:73-86) with
// c = Identity and an element-wise f.
template <class T>
__device__ __forceinline__ ALOut<T> al_elem(int f_kind, int D_kind, T x, T q, T b, T mu, T muy,
                                            T lo, T hi, T tp = T(0), int pos = 0, bool udiv = false, T rmu = T(0)) {
    ALOut<T> o;
    T cx = x;                       // eval!(cx, c, x)
    T t = cx + muy;                 // yupd = cx + mu*y
    T s = (D_kind == BZ_D_BOX) ? clamp_mm(t, lo, hi) : proj_D(D_kind, t, lo, hi, tp, pos);
    t = t - s;                      // yupd -= s
    o.pterm = udiv ? div_u(t * t, mu, rmu) : (t * t) / mu;         // (yupd^2)/mu, summed then halved
    T yupd = udiv ? div_u(t, mu, rmu) : t / mu;                    // yupd /= mu
    T dfx;
    if (f_kind == BZ_F_DIAG_QUADRATIC) {
        T qx = q * x;
        dfx = qx - b;
        o.fterm = x * (T(0.5) * qx - b);
    } else {
        dfx = T(0);
        o.fterm = T(0);
    }
    o.grad = dfx + yupd;            // dlx = dfx + jtv, jtv = yupd
    return o;
}

// prox!(z, g, y, gamma) element; gl = gamma*lambda.  Returns z, adds to gsum.
__device__ __forceinline__ double sqrt_rn(double v) { return __dsqrt_rn(v); }
__device__ __forceinline__ float sqrt_rn(float v) { return __fsqrt_rn(v); }

// scalar Newton solve of  min_z  alpha z^p + 0.5 (z - x)^2  over z >= 0 [and z <= u]
// (src/proxoperators/normLpNonneg.jl:44-84, normLpBox.jl:47-97); alpha = a*gamma.  zp returns z^p.
template <class T>
__device__ __noinline__ T lp_prox(T x, T p, T alpha, T u, bool box, T& zp) {
    zp = T(0);
    if (x <= T(0) || (box && u == T(0))) return T(0);
    const T ap = alpha * p;
    const T zbar = pow(T(1) / (ap * (T(1) - p)), T(1) / (p - T(2)));
    const T psi = zbar + ap * pow(zbar, p - T(1));
    if (psi >= x) return T(0);
    T z = zbar + (box ? T(0.1) : T(1));          // perturbation to the right
    for (int iter = 0; iter < 1000; ++iter) {
        const T dphi = z - x + ap * pow(z, p - T(1));
        if ((dphi < T(0) ? -dphi : dphi) <= T(1e-12)) break;
        const T ddphi = T(1) + ap * (p - T(1)) * pow(z, p - T(2));
        z -= dphi / ddphi;
    }
    const T phi0 = T(0.5) * (x * x);
    const T dz = z - x;
    const T pz = pow(z, p);
    if (phi0 <= T(0.5) * (dz * dz) + alpha * pz) return T(0);
    if (box && z > u) {
        const T du = u - x;
        const T pu = pow(u, p);
        if (T(0.5) * (du * du) + alpha * pu < phi0) { zp = pu; return u; }
        return T(0);
    }
    zp = pz;
    return z;
}

template <class T, bool LP = false>
__device__ __forceinline__ T prox_elem(int g_kind, T y, T gl, T u, T lo, T hi, T& gterm, T gp = T(0)) {
    T z;
    if (LP) {                       // the Newton/pow kinds live in their own kernel instantiation
        if (g_kind == BZ_G_NORM_LP_NONNEG) return lp_prox(y, gp, gl, T(0), false, gterm);
        if (g_kind == BZ_G_NORM_LP_BOX) return lp_prox(y, gp, gl, u, true, gterm);
    }
    switch (g_kind) {
    case BZ_G_NORM_L0_BOX: {        // normL0Box.jl:33-58 ; gterm counts the nonzeros
        z = T(0); gterm = T(0);
        if (u != T(0) && y > sqrt_rn(gl)) {
            if (y > u) {
                const T d = u - y;
                if (y * y > gl + d * d) { z = y; gterm = T(1); }
            } else {
                z = y; gterm = T(1);
            }
        }
        break;
    }
    case BZ_G_NORM_L1: {            // ProximalOperators.NormL1
        z = y - clamp_mm(y, -gl, gl);       // = y + (y <= -gl ? gl : (y >= gl ? -gl : -y)), bit for bit (gl > 0)
        gterm = z > T(0) ? z : -z;
        break;
    }
    case BZ_G_NORM_L1_NONNEG: {     // normL1Nonneg.jl:29-42
        if (y >= gl) { z = y - gl; gterm = z; } else { z = T(0); gterm = T(0); }
        break;
    }
    case BZ_G_NORM_L1_BOX: {        // normL1Box.jl:30-39  max(0, min(x-gl, u))
        T a = y - gl;
        a = a < u ? a : u;          // min(x-gl, u)
        z = a > T(0) ? a : T(0);    // max(0, .)
        gterm = z;
        break;
    }
    case BZ_G_IND_BOX: {
        z = y < lo ? lo : (y > hi ? hi : y);
        gterm = T(0);
        break;
    }
    default: z = y; gterm = T(0);
    }
    return z;
}

template <class T> struct ElemLoads {
    Pack<T> q, b, mu, muy, dlo, dhi, gu, glo, ghi;
};

template <class T, bool NT = false>
__device__ __forceinline__ void load_params(const ElemParams<T>& P, int64_t i0, int cnt,
                                            ElemLoads<T>& L, bool need_f, bool need_al,
                                            bool need_g) {
    L.q = splat(T(0)); L.b = splat(T(0));
    if (need_f && P.f_kind == BZ_F_DIAG_QUADRATIC) { L.q = ldp<T, NT>(P.q, i0, cnt); L.b = ldp<T, NT>(P.b, i0, cnt); }
    if (need_al) {
        L.mu = P.uni >= 1 ? splat(P.mu_uniform) : ldp<T, NT>(P.mu, i0, cnt);
        L.muy = P.uni >= 2 ? splat(T(0)) : ldp<T, NT>(P.muy, i0, cnt);
        L.dlo = P.D_lo_vec ? ldp<T, NT>(P.D_lo_vec, i0, cnt) : splat(P.D_lo);
        L.dhi = P.D_hi_vec ? ldp<T, NT>(P.D_hi_vec, i0, cnt) : splat(P.D_hi);
    }
    if (need_g) {
        L.gu = (P.g_kind == BZ_G_NORM_L1_BOX || P.g_kind == BZ_G_NORM_L0_BOX || P.g_kind == BZ_G_NORM_LP_BOX)
                   ? ldp<T, NT>(P.g_u, i0, cnt) : splat(T(0));
        L.glo = P.g_lo_vec ? ldp<T, NT>(P.g_lo_vec, i0, cnt) : splat(P.g_lo);
        L.ghi = P.g_hi_vec ? ldp<T, NT>(P.g_hi_vec, i0, cnt) : splat(P.g_hi);
    }
}

// ---------------------------------------------------------------------------
// K4: L-BFGS two-loop building block.
//   o   = sgn*in (+ coef*v) ; if apply_H: o *= H ; if xadd: o = xadd + o
//   out = o ; partial of sum(w .* o) -> parts[slot_out]
// coef comes from the previous kernel's partials:
//   mode 0 (loop 1): alpha = fold(src)/ys ; alphas[j] = alpha ; coef = -alpha
//   mode 1 (loop 2): beta  = fold(src)/ys ; coef = alphas[j] - beta
//   mode 2: no axpy term
// ---------------------------------------------------------------------------
template <class T> struct TailArgs {
    const T* in;
    const T* v;
    T sgn;
    int mode;
    int j;
    int apply_H;
    ScalarSrc src;
    T ys;
    T H;
    double* alphas;
};

template <class T>
__device__ __forceinline__ T tail_coef(const TailArgs<T>& a, double* sh) {
    if (a.mode == 2) return T(0);
    double tot = fold_src(a.src, false, sh);
    if (a.mode == 0) {
        T al = T(tot) / a.ys;
        if (blockIdx.x == 0 && threadIdx.x == 0) a.alphas[a.j] = (double)al;
        return -al;
    }
    T beta = T(tot) / a.ys;
    return T(a.alphas[a.j]) - beta;
}

template <class T>
__device__ __forceinline__ T tail_elem(const TailArgs<T>& a, T coef, T in, T v) {
    T o = a.sgn * in;
    if (a.mode != 2) {
        T t = coef * v;
        o = o + t;
    }
    if (a.apply_H) o = a.H * o;
    return o;
}

template <class T>
__global__ void __launch_bounds__(BLOCK)
k_axpy_dot(TailArgs<T> a, const T* w, const T* xadd, T* out /* may alias a.in */, int64_t n,
           double* __restrict__ parts, int slot_out) {
    __shared__ double sh[WAVES];
    const T coef = tail_coef(a, sh);
    double acc[1] = {0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pin = ld(a.in, i0, cnt);
        Pack<T> pv = (a.mode != 2) ? ld(a.v, i0, cnt) : splat(T(0));
        Pack<T> pw = w ? ld(w, i0, cnt) : splat(T(0));
        Pack<T> px = xadd ? ld(xadd, i0, cnt) : splat(T(0));
        Pack<T> po;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T o = tail_elem(a, coef, pin.v[e], pv.v[e]);
            if (xadd) o = px.v[e] + o;
            po.v[e] = o;
            if (w && e < cnt) acc[0] += (double)(pw.v[e] * o);
        }
        st(out, i0, cnt, po);
    });
    if (w) block_reduce_store<1>(acc, 0u, parts, slot_out);
}

// plain dot:  sum a .* (sgn*b)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_dot(const T* __restrict__ a, const T* __restrict__ b, T sgn, int64_t n,
      double* __restrict__ parts, int slot_out) {
    double acc[1] = {0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pa = ld(a, i0, cnt), pb = ld(b, i0, cnt);
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e)
            if (e < cnt) acc[0] += (double)(pa.v[e] * (sgn * pb.v[e]));
    });
    block_reduce_store<1>(acc, 0u, parts, slot_out);
}

// ---------------------------------------------------------------------------
// K1: AL gradient, element-wise kinds (c = Identity; f = Zero | DiagQuadratic)
//   slots: +0 sum f terms, +1 sum t^2/mu
// ---------------------------------------------------------------------------
// fext (dense f evaluated by the GEMV kernels, ProximalOperators LeastSquares / Quadratic):
//   0: f element-wise as given by P.f_kind
//   1: dfx = ext[i], f value comes from another kernel -> only the penalty slot (+1) is written
//   2: ext = Q x ; dfx = ext + P.b (b holds q) ; fterm = x (0.5 ext + q)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_algrad_elem(const T* __restrict__ x, ElemParams<T> P, T* __restrict__ grad, int64_t n,
              double* __restrict__ parts, int slot0, int fext, const T* __restrict__ ext) {
    double acc[2] = {0.0, 0.0};
    const int fk = fext ? BZ_F_ZERO : P.f_kind;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, fext == 0, true, false);
        Pack<T> px = ld(x, i0, cnt), pg;
        Pack<T> pe = fext ? ld(ext, i0, cnt) : splat(T(0));
        Pack<T> pq = (fext == 2) ? ld(P.b, i0, cnt) : splat(T(0));
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            ALOut<T> o = al_elem(fk, P.D_kind, px.v[e], L.q.v[e], L.b.v[e], L.mu.v[e],
                                 L.muy.v[e], L.dlo.v[e], L.dhi.v[e], px.v[e ^ 1] + L.muy.v[e ^ 1], e & 1);
            if (fext == 1) {
                o.grad = pe.v[e] + o.grad;                   // dfx + yupd  (al_elem returned 0 + yupd)
            } else if (fext == 2) {
                const T dfx = pe.v[e] + pq.v[e];
                o.fterm = px.v[e] * (T(0.5) * pe.v[e] + pq.v[e]);
                o.grad = dfx + o.grad;
            }
            pg.v[e] = o.grad;
            if (e < cnt) { acc[0] += (double)o.fterm; acc[1] += (double)o.pterm; }
        }
        if (grad) st(grad, i0, cnt, pg);
    });
    if (fext == 1) {
        double a1[1] = {acc[1]};
        block_reduce_store<1>(a1, 0u, parts, slot0 + 1);
    } else {
        block_reduce_store<2>(acc, 0u, parts, slot0);
    }
}

// Start of a solve, c = Identity with an element-wise f: gradient!(dlx, al, x) AND the local Lipschitz estimate
// lower_bound_smoothness_constant(f, I, x, grad) = ||grad(x + 1) - grad(x)|| / ||(x + 1) - x|| in one pass — the
// gradient at x + 1 never leaves the registers.  Same values, same sums as k_algrad_elem, k_add_scalar,
// k_algrad_elem, k_diff_ss2 run one after the other (18 passes -> 6).
//   slots: slot0 + 0 sum f terms, + 1 sum t^2/mu ; slot_aux + 0 sum (g(x+1) - g(x))^2, + 1 sum ((x+1) - x)^2
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_begin_lip(const T* __restrict__ x, ElemParams<T> P, T* __restrict__ grad, int64_t n,
            double* __restrict__ parts, int slot0, int slot_aux) {
    double acc[2] = {0.0, 0.0}, aux[2] = {0.0, 0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, true, true, false);
        Pack<T> px = ld(x, i0, cnt), pg;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            ALOut<T> o = al_elem(P.f_kind, P.D_kind, px.v[e], L.q.v[e], L.b.v[e], L.mu.v[e],
                                 L.muy.v[e], L.dlo.v[e], L.dhi.v[e], px.v[e ^ 1] + L.muy.v[e ^ 1], e & 1);
            const T xp = px.v[e] + T(1), xpp = px.v[e ^ 1] + T(1);
            ALOut<T> o2 = al_elem(P.f_kind, P.D_kind, xp, L.q.v[e], L.b.v[e], L.mu.v[e],
                                  L.muy.v[e], L.dlo.v[e], L.dhi.v[e], xpp + L.muy.v[e ^ 1], e & 1);
            pg.v[e] = o.grad;
            if (e < cnt) {
                acc[0] += (double)o.fterm; acc[1] += (double)o.pterm;
                T u = o2.grad - o.grad;
                T v = xp - px.v[e];
                aux[0] += (double)(u * u);
                aux[1] += (double)(v * v);
            }
        }
        st(grad, i0, cnt, pg);
    });
    block_reduce_store<2>(acc, 0u, parts, slot0);
    __syncthreads();
    block_reduce_store<2>(aux, 0u, parts, slot_aux);
}

// Start of a solve, same family: the forward-backward step at x, gradient!(., al, z) and the stop norm in one
// pass — k_fbstep, k_algrad_elem at z and k_update (with no previous state) run one after the other, z and its
// gradient never leaving the registers in between (18 passes -> 8).  Repeated once per gamma halving.
//   slots slot0 + 0..7: sum g terms, <grad L(x), res>, ||res||^2, sum f terms at z, sum t^2/mu at z, 0, 0,
//                       max |res/gamma - grad L(x) + grad L(z)|      (the layout SL_GSUM .. SL_STOP)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_begin_fb(const T* __restrict__ x, const T* __restrict__ gx, T gamma, ElemParams<T> P, T* __restrict__ z,
           T* __restrict__ res, int64_t n, double* __restrict__ parts, int slot0) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const T gl = gamma * P.g_lambda;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, true, true, true);
        Pack<T> px = ld(x, i0, cnt), pg = ld(gx, i0, cnt), pz, pr;
        T gterm[PackN<T>::N];
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T t = gamma * pg.v[e];
            T y = px.v[e] - t;
            T zz = prox_elem<T, false>(P.g_kind, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm[e], P.g_p);
            pz.v[e] = zz; pr.v[e] = px.v[e] - zz;
        }
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            ALOut<T> o = al_elem(P.f_kind, P.D_kind, pz.v[e], L.q.v[e], L.b.v[e], L.mu.v[e],
                                 L.muy.v[e], L.dlo.v[e], L.dhi.v[e], pz.v[e ^ 1] + L.muy.v[e ^ 1], e & 1);
            const T r = pr.v[e];
            T w = r / gamma;
            w = w - pg.v[e];
            w = w + o.grad;
            if (e < cnt) {
                acc[0] += (double)gterm[e];
                acc[1] += (double)(pg.v[e] * r);
                acc[2] += (double)(r * r);
                acc[3] += (double)o.fterm;
                acc[4] += (double)o.pterm;
                acc[7] = nanmax(acc[7], (double)(w < T(0) ? -w : w));
            }
        }
        st(z, i0, cnt, pz);
        st(res, i0, cnt, pr);
    });
    block_reduce_store<8>(acc, 1u << 7, parts, slot0);
}

// z (and res) of a state from its x alone, same family: z = prox_{gamma g}(x - gamma grad L(x)), res = x - z in one pass —
// k_algrad_elem + k_fbstep without the gradient's round trip through memory (10 passes -> 6..7), for the
// re-materialisation of a z the one-pass kernel did not store
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_zres_elem(const T* __restrict__ x, ElemParams<T> P, T gamma, T* __restrict__ z, T* __restrict__ res, int64_t n) {
    const T gl = gamma * P.g_lambda;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, true, true, true);
        Pack<T> px = ld(x, i0, cnt), pz, pr;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            ALOut<T> o = al_elem(P.f_kind, P.D_kind, px.v[e], L.q.v[e], L.b.v[e], L.mu.v[e],
                                 L.muy.v[e], L.dlo.v[e], L.dhi.v[e], px.v[e ^ 1] + L.muy.v[e ^ 1], e & 1);
            T t = gamma * o.grad;
            T y = px.v[e] - t;
            T gterm;
            T zz = prox_elem<T, false>(P.g_kind, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm, P.g_p);
            pz.v[e] = zz; pr.v[e] = px.v[e] - zz;
        }
        st(z, i0, cnt, pz);
        if (res) st(res, i0, cnt, pr);
    });
}

// ---------------------------------------------------------------------------
// K10 + K1: AL gradient with the 5-point-stencil quadratic f (cfg 3), c = Identity.
//   f(x) = 0.5 x'A_h x - b'x on an nx-by-ny grid (row-major, index = i*ny + j),
//   A_h = (4,-1,-1,-1,-1), homogeneous Dirichlet halo.  Per element (same order as the oracle):
//     Ax = ((((4 x_c - x_w) - x_e) - x_n) - x_s) ; dfx = Ax - b ; fterm = x_c (0.5 Ax - b)
//   Each lane owns one 16-B pack of a row (ny % pack == 0, so packs never straddle rows);
//   north/south packs are the same columns of rows i-1/i+1, west/east are one scalar each.
//   The three reads of a row (as centre, north, south) hit L2/Infinity Cache: HBM sees x once.
//   slots: +0 sum f terms, +1 sum t^2/mu.   f_only: skip the AL terms (alps.jl:39, f(x) alone).
// ---------------------------------------------------------------------------
// one pack of the stencil AL gradient; accF/accP accumulate the f and penalty terms of the valid lanes
// Row-block sharding (x sharded over the GPUs by grid rows): the row above the first local row and the row
// below the last one are the neighbour ranks' (k_halo_exchange); a null halo is the Dirichlet-0 boundary.
template <class T> struct StencilHalo {
    const T* north;      // ny values: the last row of the previous rank, or null
    const T* south;      // ny values: the first row of the next rank, or null
};
//   NTP: the parameter vectors (b, mu, mu*y, the bounds) with non-temporal loads — at grid sizes beyond the Infinity Cache
//   they are streamed once per pass and would only evict what the passes of one iteration hand to each other
//   (x_d, grad L(x_d), z, res)
template <class T, bool NTP = false>
__device__ __forceinline__ Pack<T> stencil_al_pack(const T* __restrict__ x, const ElemParams<T>& P, int64_t nx,
                                                   int64_t ny, int f_only, int64_t i0, int cnt,
                                                   const Pack<T>& xc, double& accF, double& accP,
                                                   StencilHalo<T> halo = StencilHalo<T>{nullptr, nullptr}) {
    constexpr int N = PackN<T>::N;
    const int64_t i = i0 / ny, j0 = i0 - i * ny;
    Pack<T> xn = (i > 0) ? ld(x, i0 - ny, cnt) : (halo.north ? ld(halo.north, j0, cnt) : splat(T(0)));
    Pack<T> xs = (i + 1 < nx) ? ld(x, i0 + ny, cnt) : (halo.south ? ld(halo.south, j0, cnt) : splat(T(0)));
    const T west = (j0 > 0) ? x[i0 - 1] : T(0);
    const T east = (j0 + N < ny) ? x[i0 + N] : T(0);
    Pack<T> pb = ldp<T, NTP>(P.b, i0, cnt);
    ElemLoads<T> L;
    if (!f_only) load_params<T, NTP>(P, i0, cnt, L, false, true, false);
    Pack<T> pg;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const T c = xc.v[e];
        const T w = (e == 0) ? west : xc.v[e > 0 ? e - 1 : 0];
        const T ee = (e == N - 1) ? east : xc.v[e < N - 1 ? e + 1 : N - 1];
        T Ax = T(4) * c;
        Ax = Ax - w;
        Ax = Ax - ee;
        Ax = Ax - xn.v[e];
        Ax = Ax - xs.v[e];
        const T dfx = Ax - pb.v[e];
        const T fterm = c * (T(0.5) * Ax - pb.v[e]);
        T g = dfx, pterm = T(0);
        if (!f_only) {
            T t = c + L.muy.v[e];
            T sv = proj_D(P.D_kind, t, L.dlo.v[e], L.dhi.v[e]);
            t = t - sv;
            pterm = (t * t) / L.mu.v[e];
            T yupd = t / L.mu.v[e];
            g = dfx + yupd;
        }
        pg.v[e] = g;
        if (e < cnt) { accF += (double)fterm; accP += (double)pterm; }
    }
    return pg;
}

template <class T>
__global__ void __launch_bounds__(BLOCK)
k_algrad_stencil(const T* __restrict__ x, ElemParams<T> P, int64_t nx, int64_t ny, int f_only,
                 T* __restrict__ grad, int64_t n, double* __restrict__ parts, int slot0, StencilHalo<T> halo) {
    double acc[2] = {0.0, 0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> xc = ld(x, i0, cnt);
        Pack<T> pg = stencil_al_pack(x, P, nx, ny, f_only, i0, cnt, xc, acc[0], acc[1], halo);
        if (grad) st(grad, i0, cnt, pg);
    });
    block_reduce_store<2>(acc, 0u, parts, slot0);
}

// cfg-3 fast path, first half: gradient!(.., al, x_d) and the forward-backward step at x_d in one pass
// (z_i depends only on x_d,i and gradL(x_d)_i).  Same arithmetic and summation order as
// k_algrad_stencil followed by k_fbstep.  slots: slot_f +0 f terms, +1 t^2/mu ; slot_g +0 g terms,
// +1 <g,res>, +2 ||res||^2
template <class T, bool NTP = false>
__global__ void __launch_bounds__(BLOCK)
k_stencil_fb(const T* __restrict__ x, ElemParams<T> P, int64_t nx, int64_t ny, T gamma,
             T* __restrict__ grad, T* __restrict__ z, T* __restrict__ res, int64_t n,
             double* __restrict__ parts, int slot_f, int slot_g, StencilHalo<T> halo) {
    double accF[2] = {0.0, 0.0}, accG[3] = {0.0, 0.0, 0.0};
    const T gl = gamma * P.g_lambda;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> xc = ld(x, i0, cnt);
        Pack<T> pg = stencil_al_pack<T, NTP>(x, P, nx, ny, 0, i0, cnt, xc, accF[0], accF[1], halo);
        ElemLoads<T> L;
        load_params<T, NTP>(P, i0, cnt, L, false, false, true);
        Pack<T> pz, pr;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T t = gamma * pg.v[e];
            T y = xc.v[e] - t;
            T gterm;
            T zz = prox_elem(P.g_kind, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm);
            T r = xc.v[e] - zz;
            pz.v[e] = zz; pr.v[e] = r;
            if (e < cnt) {
                accG[0] += (double)gterm;
                accG[1] += (double)(pg.v[e] * r);
                accG[2] += (double)(r * r);
            }
        }
        if (grad) st(grad, i0, cnt, pg);      // (null: the pass that follows re-forms it, k_stencil_update_c<REGX>)
        st(z, i0, cnt, pz);
        st(res, i0, cnt, pr);
    });
    block_reduce_store<2>(accF, 0u, parts, slot_f);
    block_reduce_store<3>(accG, 0u, parts, slot_g);
}

// cfg-3 fast path, second half: gradient!(.., al, z) with the L-BFGS pair and the stopping norm in the
// same pass (gradL(z) never goes to memory unless gz_out is given).  Same arithmetic and summation
// order as k_algrad_stencil followed by k_update.  slots: slot_f +0,+1 ; slot_u +0 <s,y>, +1 <y,y>, +2 max
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_stencil_update(const T* __restrict__ zp, ElemParams<T> P, int64_t nx, int64_t ny,
                 const T* __restrict__ x, const T* __restrict__ x_prev, const T* __restrict__ res,
                 const T* __restrict__ res_prev, const T* __restrict__ gx, T gamma,
                 T* __restrict__ s_new, T* __restrict__ y_new, T* __restrict__ gz_out, int64_t n,
                 double* __restrict__ parts, int slot_f, int slot_u, StencilHalo<T> halo) {
    double accF[2] = {0.0, 0.0}, accU[3] = {0.0, 0.0, 0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> zc = ld(zp, i0, cnt);
        Pack<T> pgz = stencil_al_pack(zp, P, nx, ny, 0, i0, cnt, zc, accF[0], accF[1], halo);
        Pack<T> px = ld(x, i0, cnt), pxp = ld(x_prev, i0, cnt), pr = ld(res, i0, cnt), prp = ld(res_prev, i0, cnt);
        Pack<T> pgx = ld(gx, i0, cnt), ps, py;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T sv = px.v[e] - pxp.v[e];
            T yv = pr.v[e] - prp.v[e];
            ps.v[e] = sv; py.v[e] = yv;
            T w = pr.v[e] / gamma;
            w = w - pgx.v[e];
            w = w + pgz.v[e];
            if (e < cnt) {
                accU[0] += (double)(sv * yv);
                accU[1] += (double)(yv * yv);
                accU[2] = nanmax(accU[2], (double)(w < T(0) ? -w : w));
            }
        }
        st(s_new, i0, cnt, ps);
        st(y_new, i0, cnt, py);
        if (gz_out) st(gz_out, i0, cnt, pgz);
    });
    block_reduce_store<2>(accF, 0u, parts, slot_f);
    __syncthreads();
    block_reduce_store<3>(accU, 4u, parts, slot_u);
}

// ---------------------------------------------------------------------------
// K9: dense constraint map c(x) = A x - b (demo/basispursuit.jl:38-49), A[ny][n] row-major.
// Both products are bound by the bytes of A (2 flop per 4 B); x / v stay in L2.
// ---------------------------------------------------------------------------
// cx = A p - b : one block per row (grid-strided), lanes along the contiguous columns.
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_gemv_n(const T* __restrict__ A, const T* __restrict__ p, const T* __restrict__ b,
         T* __restrict__ cx, int64_t ny, int64_t n) {
    constexpr int N = PackN<T>::N;
    __shared__ double sh[WAVES];
    const bool aligned = (n % N) == 0;      // rows start 16-B aligned only then
    const int64_t npk = aligned ? n / N : 0;
    for (int64_t r = blockIdx.x; r < ny; r += gridDim.x) {
        const T* row = A + r * n;
        T a0 = T(0), a1 = T(0);
        int64_t c = threadIdx.x;
        if (!aligned)
            for (int64_t j = threadIdx.x; j < n; j += BLOCK) a0 += row[j] * p[j];
        for (; c + BLOCK < npk; c += 2 * BLOCK) {
            // (the matrix is streamed once per product: non-temporal, so that it does not evict p and the partials)
            Pack<T> ra = ldp<T, true>(row, c * N, N);
            Pack<T> rb = ldp<T, true>(row, (c + BLOCK) * N, N);
            Pack<T> pa = *reinterpret_cast<const Pack<T>*>(p + c * N);
            Pack<T> pb = *reinterpret_cast<const Pack<T>*>(p + (c + BLOCK) * N);
#pragma unroll
            for (int e = 0; e < N; ++e) { a0 += ra.v[e] * pa.v[e]; a1 += rb.v[e] * pb.v[e]; }
        }
        for (; c < npk; c += BLOCK) {
            Pack<T> ra = ldp<T, true>(row, c * N, N);
            Pack<T> pa = *reinterpret_cast<const Pack<T>*>(p + c * N);
#pragma unroll
            for (int e = 0; e < N; ++e) a0 += ra.v[e] * pa.v[e];
        }
        double v = wave_sum((double)a0 + (double)a1);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            T dot = (T)((sh[0] + sh[1]) + (sh[2] + sh[3]));
            cx[r] = b ? dot - b[r] : dot;
        }
        __syncthreads();
    }
}

// the ny-vector part of gradient!(dlx, al, x) (auglagfun.jl:74-78):
//   t = cx + muy ; s = proj_D(t) ; t -= s ; slot +0 sum t^2/mu ; yupd = t/mu
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_yupd(const T* __restrict__ cx, ElemParams<T> P, T* __restrict__ yupd, int64_t ny,
       double* __restrict__ parts, int slot0) {
    double acc[1] = {0.0};
    bz_for_chunks<T>(ny, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, false, true, false);
        Pack<T> pc = ld(cx, i0, cnt), py;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T t = pc.v[e] + L.muy.v[e];
            T sv = proj_D(P.D_kind, t, L.dlo.v[e], L.dhi.v[e]);
            t = t - sv;
            T pterm = (t * t) / L.mu.v[e];
            py.v[e] = t / L.mu.v[e];
            if (e < cnt) acc[0] += (double)pterm;
        }
        st(yupd, i0, cnt, py);
    });
    block_reduce_store<1>(acc, 0u, parts, slot0);
}

// jtv partials: part[rc][j] = sum_{i in row chunk rc} A[i][j] v[i].  blockIdx.x = column block
// (BLOCK packs), blockIdx.y = row chunk.  Each wave-load is 1 KiB contiguous of one row.
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_gemv_t(const T* __restrict__ A, const T* __restrict__ v, T* __restrict__ part, int64_t ny,
         int64_t n, int rows_per_chunk, int64_t pstride) {
    constexpr int N = PackN<T>::N;
    const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;     // pack index along the row
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < ny) ? r0 + rows_per_chunk : ny;
    if ((n % N) != 0) {                     // rows not 16-B aligned: one column per thread
        if (c < n) {
            T a = T(0);
            for (int64_t r = r0; r < r1; ++r) a += A[r * n + c] * v[r];
            part[(int64_t)blockIdx.y * pstride + c] = a;
        }
        return;
    }
    if (c * N >= n) return;
    Pack<T> a0 = splat(T(0)), a1 = splat(T(0));
    const T* col = A + c * N;
    int64_t r = r0;
    for (; r + 1 < r1; r += 2) {
        Pack<T> ra = ldp<T, true>(col, r * n, N);
        Pack<T> rb = ldp<T, true>(col, (r + 1) * n, N);
        const T va = v[r], vb = v[r + 1];
#pragma unroll
        for (int e = 0; e < N; ++e) { a0.v[e] += ra.v[e] * va; a1.v[e] += rb.v[e] * vb; }
    }
    if (r < r1) {
        Pack<T> ra = *reinterpret_cast<const Pack<T>*>(col + r * n);
        const T va = v[r];
#pragma unroll
        for (int e = 0; e < N; ++e) a0.v[e] += ra.v[e] * va;
    }
    Pack<T> o;
#pragma unroll
    for (int e = 0; e < N; ++e) o.v[e] = a0.v[e] + a1.v[e];
    *reinterpret_cast<Pack<T>*>(part + (int64_t)blockIdx.y * pstride + c * N) = o;
}

// jtv partials on the matrix cores (fp32): v_mfma_f32_16x16x4_f32 with the matrix tile as the B
// operand — lane l feeds B[k = l>>4][col = l&15] = A[i + (l>>4)][j0 + 4(l&15) + q], one 16-B load
// per lane covering q = 0..3, so a wave-load is 4 rows x 256 B contiguous — and v as the A operand
// in row 0 only (A[m = l&15][k = l>>4] = v[i + k] for m == 0, else 0).  Row 0 of D (lanes 0..15,
// register 0) then accumulates sum_i v[i] A[i][j] as an exact k-ordered fmaf chain.  15/16 of the
// MFMA work is padding: a GEMV has no reuse, the kernel stays bound by the bytes of A (the f32 MFMA
// ingests 32 B/clk/CU = 19.7 TB/s chip-wide, 3x the HBM rate).  Four accumulators (one per q) cover
// the 40-cycle dependent-accumulator latency.  Same partial layout as k_gemv_t.
typedef float bz_f32x4 __attribute__((ext_vector_type(4)));

static __global__ void __launch_bounds__(BLOCK)
k_gemv_t_mfma(const float* __restrict__ A, const float* __restrict__ v, float* __restrict__ part,
              int64_t ny, int64_t n, int rows_per_chunk, int64_t pstride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = ((int64_t)blockIdx.x * WAVES + wave) * 64;      // 64 columns per wave
    if (j0 >= n) return;                                              // wave-uniform
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < ny) ? r0 + rows_per_chunk : ny;
    const int k = lane >> 4, c = lane & 15;
    const float* col = A + j0 + 4 * c;
    bz_f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    int64_t i = r0;
    for (; i + 8 <= r1; i += 8) {                                     // two 4-row steps in flight
        const Pack<float> ma = ldp<float, true>(col, (i + k) * n, 4);      // (non-temporal: the matrix is streamed once)
        const Pack<float> mb = ldp<float, true>(col, (i + 4 + k) * n, 4);
        const float va = (c == 0) ? v[i + k] : 0.0f;
        const float vb = (c == 0) ? v[i + 4 + k] : 0.0f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[3], acc3, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb, mb.v[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb, mb.v[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb, mb.v[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb, mb.v[3], acc3, 0, 0, 0);
    }
    for (; i < r1; i += 4) {                                          // ragged tail: rows beyond r1 feed zeros
        const bool ok = (i + k) < r1;
        Pack<float> ma = splat(0.0f);
        if (ok) ma = ldp<float, true>(col, (i + k) * n, 4);
        const float va = (ok && c == 0) ? v[i + k] : 0.0f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(va, ma.v[3], acc3, 0, 0, 0);
    }
    if (lane < 16) {                                                  // D row 0 = lanes 0..15, register 0
        Pack<float> o;
        o.v[0] = acc0[0]; o.v[1] = acc1[0]; o.v[2] = acc2[0]; o.v[3] = acc3[0];
        *reinterpret_cast<Pack<float>*>(part + (int64_t)blockIdx.y * pstride + j0 + 4 * lane) = o;
    }
}

// dlx = dfx + jtv with jtv = sum over row chunks (fixed order) ; f element-wise (Zero|DiagQuadratic)
//   slot +0: sum f terms
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_gemv_t_finish(const T* __restrict__ part, int nchunks, int64_t pstride, const T* __restrict__ x,
                ElemParams<T> P, T* __restrict__ grad, int64_t n, double* __restrict__ parts, int slot0) {
    double acc[1] = {0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        // (the partials in their fixed order — chunk order, or rank order when they are the ranks' — with eight loads in
        // flight at a time: 34 dependent load-add steps made this short kernel 14.5 us long)
        Pack<T> j = ld(part, i0, cnt);
        int k = 1;
        for (; k + 7 < nchunks; k += 8) {
            Pack<T> q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = ld(part + (int64_t)(k + u) * pstride, i0, cnt);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < PackN<T>::N; ++e) j.v[e] = j.v[e] + q[u].v[e];
        }
        for (; k < nchunks; ++k) {
            Pack<T> q = ld(part + (int64_t)k * pstride, i0, cnt);
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) j.v[e] = j.v[e] + q.v[e];
        }
        Pack<T> px = ld(x, i0, cnt), pq = splat(T(0)), pb = splat(T(0)), pg;
        if (P.f_kind == BZ_F_DIAG_QUADRATIC) { pq = ld(P.q, i0, cnt); pb = ld(P.b, i0, cnt); }
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T dfx = T(0), fterm = T(0);
            if (P.f_kind == BZ_F_DIAG_QUADRATIC) {
                T qx = pq.v[e] * px.v[e];
                dfx = qx - pb.v[e];
                fterm = px.v[e] * (T(0.5) * qx - pb.v[e]);
            }
            pg.v[e] = dfx + j.v[e];
            if (e < cnt) acc[0] += (double)fterm;
        }
        if (grad) st(grad, i0, cnt, pg);
    });
    block_reduce_store<1>(acc, 0u, parts, slot0);
}

// ---------------------------------------------------------------------------
// K4, persistent form: the whole L-BFGS two-loop in ONE launch with d held in registers.
//
// One 512-thread block per CU (8 waves, 256 VGPRs each: the CU's whole register file); thread t
// of block b owns packs c(k) = (k*NB + b)*512 + t, k < KR, i.e. up to 40 double2 = 160 VGPRs of
// d.  The 2m-1 sequentially dependent dot products become grid-wide phases separated by a
// counter barrier (agent-scope release -> atomic add -> relaxed poll -> agent-scope acquire;
// partials travel as sc1 stores/loads, every spin is bounded).  Per phase only the two history
// vectors stream from HBM (d neither re-read nor re-written):
//   traffic: res + s_1 + 2(m-1) + 1 + 2(m-1) + d_out = 4m passes   vs  8m-3 for the kernel chain.
// The arithmetic per element and the coefficient formulas are those of k_dot / k_axpy_dot.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Peer-to-peer scalar exchange (one node, x sharded over the GPUs): system-scope 8-byte stores into
// every rank's mailbox (fine-grained device memory mapped through HIP IPC), data first, then a
// release fence, then a sequence-number flag; readers poll their OWN mailbox.  Double-buffered by the
// parity of the sequence number: a rank can run at most one exchange ahead of the slowest rank,
// because completing exchange q+1 needs everybody's q+1 contribution.
// ---------------------------------------------------------------------------
struct P2PWords {            // device view of bz::P2PMailbox (bz_solver.h); same layout
    double pval[2][8];
    unsigned long long pflag[2][8];
    unsigned long long xll[2][8][32][2];   // pack exchanges: {low half | tag << 32}, {high half | tag << 32}
};
__device__ __forceinline__ void sys_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr unsigned XSPIN_LIMIT = 20000000u;

// fold this rank's block partials of slots [first, first+cnt), exchange the pack with all ranks and
// leave every rank's pack in recv[r*cnt + i] (the layout ScalarSrc{recv, nranks, cnt} expects)
constexpr int XBLOCK = 1024;              // 16 waves: up to 16 scalars folded at once
struct SlotCounts {                       // valid block partials per slot, 16 bits each (<= PSTRIDE = 2048)
    unsigned long long w[8];
    __host__ __device__ void set(int i, int c) { w[i >> 2] |= (unsigned long long)(c & 0xFFFF) << ((i & 3) * 16); }
    __device__ __forceinline__ int get(int i) const {
        unsigned long long v = w[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) if ((i >> 2) == k) v = w[k];      // compile-time indices: stays in SGPRs
        return (int)((v >> ((i & 3) * 16)) & 0xFFFFull);
    }
};
struct XchgArgs {
    const double* parts;
    SlotCounts counts;
    int first, cnt;
    unsigned maxmask;
    int rank, nranks;
    unsigned long long seq;
    double* recv;
    P2PWords* mbox_local;
    P2PWords* mbox_peer[8];
    int* timeout;
    unsigned keepmask;       // bit i clear: slot first+i holds a quantity every rank computed in full (x is
                             // replicated, e.g. with a row-sharded dense c): only rank 0's copy is counted
};
// The pack travels in "LL" form: every 8-byte word carries half a value and a 32-bit tag of the
// exchange's sequence number, so a word is valid exactly when its tag matches — no separate flag, no
// fence between data and flag, one store latency + one load latency per exchange.  8-byte stores are
// single-copy atomic, and each half is checked against its own tag, so a torn pair cannot be taken.
__device__ __host__ __forceinline__ unsigned ll_tag(unsigned long long seq) { return (unsigned)(seq & 0x7FFFFFFFull) + 1u; }
// the same tagged form towards the host's pinned mailbox: two 8-byte words per scalar, no fence; the host
// takes a scalar when both of its words carry the tag of the read-back it is waiting for
__device__ __forceinline__ void host_post(double* out, int i, double v, unsigned long long seq) {
    const unsigned long long tag = (unsigned long long)ll_tag(seq) << 32;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    unsigned long long* dst = (unsigned long long*)out + 2 * i;
    sys_store(dst + 0, tag | (bits & 0xFFFFFFFFull));
    sys_store(dst + 1, tag | (bits >> 32));
}
static __global__ void __launch_bounds__(64) k_exchange(XchgArgs a) {      // one wave per slot
    const int tid = threadIdx.x, i = blockIdx.x;
    double t = fold_wave(a.parts + (size_t)(a.first + i) * PSTRIDE, a.counts.get(i), (a.maxmask >> i) & 1u);
    if (a.rank != 0 && !((a.keepmask >> i) & 1u)) t = 0.0;
    const int par = (int)(a.seq & 1ull);
    const unsigned long long tag = (unsigned long long)ll_tag(a.seq) << 32;
    if (tid < a.nranks) {
        const int r = tid;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(t);
        unsigned long long* dst = a.mbox_peer[r]->xll[par][a.rank][i];
        sys_store(dst + 0, tag | (bits & 0xFFFFFFFFull));
        sys_store(dst + 1, tag | (bits >> 32));
        const unsigned long long* src = a.mbox_local->xll[par][r][i];
        unsigned long long w0, w1;
        unsigned spins = 0;
        for (;;) {
            w0 = sys_load(src + 0);
            w1 = sys_load(src + 1);
            if ((w0 >> 32 << 32) == tag && (w1 >> 32 << 32) == tag) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > XSPIN_LIMIT) { *a.timeout = 2; break; }
        }
        a.recv[r * a.cnt + i] = __longlong_as_double((long long)((w0 & 0xFFFFFFFFull) | (w1 << 32)));
    }
}

// exchange + fold over the ranks (rank order: identical bits everywhere) + read-back in ONE launch
struct XCollectArgs {
    XchgArgs x;
    double* host_out;
    unsigned long long ticket;
};
// One workgroup PER SCALAR (as k_collect): fold the slot, post it to every rank's mailbox, wait for every
// rank's copy, fold over the ranks, hand the total to the host.  The 32 scalars of an exchange travel
// independently — every tagged word validates itself — so nothing serialises on one workgroup (a single
// 1024-thread block doing all of it measured 9.6 us and delayed the host's view of the result by ~10 us more).
// (scalar i of the pack, by the 64 lanes of one wave; without mailboxes — a single rank with no p2p context — the
// local fold goes straight to the host)
__device__ __forceinline__ void exchange_collect_one(const XCollectArgs& b, const int i, const int tid) {
    const XchgArgs& a = b.x;
    const bool ismax = (a.maxmask >> i) & 1u;
    double t = fold_wave(a.parts + (size_t)(a.first + i) * PSTRIDE, a.counts.get(i), ismax);
    if (!a.mbox_local) {
        if (tid == 0) host_post(b.host_out, i, t, b.ticket);
        return;
    }
    if (a.rank != 0 && !((a.keepmask >> i) & 1u)) t = 0.0;
    const int par = (int)(a.seq & 1ull);
    const unsigned long long tag = (unsigned long long)ll_tag(a.seq) << 32;
    double v = 0.0;
    if (tid < a.nranks) {
        const int r = tid;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(t);
        unsigned long long* dst = a.mbox_peer[r]->xll[par][a.rank][i];
        sys_store(dst + 0, tag | (bits & 0xFFFFFFFFull));
        sys_store(dst + 1, tag | (bits >> 32));
        const unsigned long long* src = a.mbox_local->xll[par][r][i];
        unsigned long long w0, w1;
        unsigned spins = 0;
        for (;;) {
            w0 = sys_load(src + 0);
            w1 = sys_load(src + 1);
            if ((w0 >> 32 << 32) == tag && (w1 >> 32 << 32) == tag) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > XSPIN_LIMIT) { *a.timeout = 2; break; }
        }
        v = __longlong_as_double((long long)((w0 & 0xFFFFFFFFull) | (w1 << 32)));
        a.recv[r * a.cnt + i] = v;
    }
    // fold over the ranks in rank order (lane r holds rank r's copy): every lane computes the same chain
    double g = 0.0;
    for (int r = 0; r < a.nranks; ++r) {
        const double vr = __shfl(v, r, 64);
        g = ismax ? nanmax(g, vr) : g + vr;
    }
    if (tid == 0) host_post(b.host_out, i, g, b.ticket);
}
static __global__ void __launch_bounds__(64) k_exchange_collect(XCollectArgs b) {      // one WAVE per scalar
    exchange_collect_one(b, (int)blockIdx.x, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------
// K9, ONE pass over A: gradient!(dlx, al, x) with the dense constraint c(x) = A x - b
// (src/utilities/auglagfun.jl:73-86 with demo/basispursuit.jl:38-49: eval!(cx, c, x) = A x - b ;
// jtprod!(jtv, c, x, v) = A'v — SURVEY 7 H6).  As two kernels (k_gemv_n, k_gemv_t) the matrix streams from HBM
// twice per evaluation; here every row is read once: yhat_i = ((a_i.x - b_i + mu_i y_i) - s_i) / mu_i needs the whole
// row a_i before it can multiply that row again, so the row must stay on chip in between — 256 KB at n = 65536 fp32,
// half a CU's register file.  A row GROUP is therefore shared by G workgroups (one per CU) that each own a column
// slice of 512 * KP packs; the slice of x and the slice of the accumulator A'yhat stay in registers for the whole launch,
// and so does a RING of four tiles of FT = 2 rows x the slice (KP packs per row and lane).
//
// Per tile the G workgroups exchange their FT partial products through a small device-memory mailbox in "LL" form (half
// a value and a 32-bit sequence tag per 8-byte word: a word is valid exactly when its tag matches, no flag, no fence),
// fold them in slice order — the same bits in every workgroup of the group — and go on to acc += yhat_i a_i.  The
// exchange is PIPELINED one tile deep: tile p's partials are posted in step p and consumed in step p + 1, when every
// partner has long posted them, so a workgroup never waits for the slowest of its group inside a step (with the
// consumption in the same step the pass took 443 us against 374 us with the exchange compiled out: the workgroups of a
// group ran in lockstep and every step cost the slowest one's load time plus the mailbox round trip).  Ring slot
// (p + 3) mod 4 is refilled as soon as tile p - 1 has been accumulated: three tiles are in flight behind the one being
// multiplied.  Four mailbox slots per group: a workgroup is never more than a step ahead of the slowest of its group
// (step p + 1 needs everybody's partials of tile p).
//
// What keeps the loads flowing (each of these cost 10-25 % of the pass when it was missing):
//  * the polls are SCALAR loads (s_load ... glc: the scalar cache bypassed, served by L2, counted on lgkmcnt).  A wave's
//    vector-memory operations return in order, so a vector poll issued behind a prefetch does not come back before that
//    whole tile has landed (tools/probes/spoll.hip: a scalar poll sees another workgroup's agent-scope store, same XCD
//    or not, ~0.55 us per hop); a slice that stays unseen for long is also tried with vector loads, which are coherent
//    whatever the placement of the group;
//  * the rows' b, mu, mu*y and bounds come from LDS, filled once (read per tile they would queue behind the prefetch);
//  * every tile load is unconditional — a row beyond the group's last one re-reads that last row and is never used, a
//    lane beyond the last column reads column 0 against x = 0: with a branch around a load the compiler cannot count
//    the loads in flight and waits for ALL of them (vmcnt(0)) before the first product;
//  * one workgroup per CU: two (half the ring each) or narrower slices (16 or 32 partners) were 20-60 % slower.
// With blockIdx round-robin over the XCDs the G workgroups of a group are placed on ONE XCD.
// Output: c(x) (by slice 0), the partials of A'yhat per row group in k_gemv_t's layout part[group][col] — folded in
// group order by k_gemv_t_finish as before — and the penalty partial sum(t^2 / mu) per group.
// Every poll is bounded: a group whose workgroups are not all resident (another tenant holds CUs) reports a timeout
// to the host instead of hanging; the host then goes back to the two-kernel form for good.
// Both products are VALU fma: the matrix cores would need the yhat-weighted rows as a 16-row tile to fill their
// accumulators (k_gemv_t_mfma feeds them 4 rows and wastes 15/16 of the result tile) — 16x the accumulator registers
// this kernel spends on the tiles in flight; at 2 flop per 4 bytes the pass is HBM-bound either way.
// ---------------------------------------------------------------------------
constexpr int FBLOCK = 512;              // 8 waves, one workgroup per CU
constexpr int FWAVES = FBLOCK / 64;
constexpr int FT = 2;                    // rows per tile
constexpr int FNB = 4;                   // tiles in the register ring
constexpr int FG_MAX = 16;               // column slices (workgroups) per row group: at most two per polling wave
constexpr int FMS = 4;                   // mailbox slots per group (ring over the tile sequence number)
constexpr int FRC = 1024;                // rows per group, at most (their parameters sit in LDS)
constexpr unsigned FSPIN_LIMIT = 8000000u;
static_assert(FG_MAX <= 2 * FWAVES && FT == 2, "wave w polls slices w and w + 8; a slice's tagged pairs are one 32-byte scalar load");
template <class T> struct DenseFusedArgs {
    const T* A;
    const T* x;
    const T* b;
    T* cx;                               // c(x) = A x - b
    T* cx2;                              // ... a second copy (the affine images keep c at the current points), or null
    T* part;                             // A'yhat partials: part[group * pstride + col]
    int64_t pstride;
    int64_t ny, n;
    int G, ngroups;
    int64_t rows_per_group;              // a multiple of FT
    unsigned long long seq0;             // sequence number of this launch's first tile exchange
    unsigned long long* mail;            // [ngroups][FMS][FG_MAX][FT][2] tagged words
    int* timeout;
    unsigned spin;
    double* parts;
    int slot_pen;
    int sabotage;                        // (test) 1: slice 0 posts under a wrong tag: its partners' polls must give up, not hang
};
__device__ __forceinline__ void dev_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long dev_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
typedef unsigned bz_u8v __attribute__((ext_vector_type(8)));

// The ring's loads are inline asm and so are the waits on them.  Left to the compiler, the wait in front of a tile's products
// came out as vmcnt(0) — the three prefetched tiles drained at every turn of the ring: its bookkeeping merges the paths of the
// poll loops and of the loop's back edge pessimistically.  An asm load is invisible to that bookkeeping; the wait statement
// takes the tile's registers as in/out operands, so no product can be scheduled above it, and between a load and its wait
// nothing reads those registers (checked in the generated code: tools/check_dense_ring.py).
template <class T, int KP, int BI>
__device__ __forceinline__ void dense_load_tile(typename PackVec<T>::type (&ring)[FNB][FT][KP], const unsigned (&off)[KP], const T* A,
                                                const int64_t n, const int64_t r0, const int64_t r1, const int q) {
    // tile q -> ring slot BI; a row beyond the group's last one re-reads that last row (never used)
#pragma unroll
    for (int r = 0; r < FT; ++r) {
        const int64_t rr = r0 + (int64_t)q * FT + r;
        const int64_t row = rr < r1 ? rr : r1 - 1;
        const T* rp = A + row * n;
#pragma unroll
        for (int k = 0; k < KP; ++k)                                     // (streamed once: non-temporal)
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(ring[BI][r][k]) : "v"(off[k]), "s"(rp) : "memory");
    }
}
// wait until at most two tiles' worth of this wave's vector-memory operations are outstanding: the two younger tiles stay in
// flight, the tile in slot BI has landed (operations return in order; the two mailbox stores a step of wave 0 adds are older
// than the tile behind them, so the same count holds there)
template <class T, int KP, int BI>
__device__ __forceinline__ void dense_wait_tile(typename PackVec<T>::type (&ring)[FNB][FT][KP]) {
    constexpr int CNT = 2 * FT * KP;
    static_assert(FT == 2 && CNT < 64, "");
    if constexpr (KP == 4)
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ring[BI][0][0]), "+v"(ring[BI][0][1]), "+v"(ring[BI][0][2]), "+v"(ring[BI][0][3]),
                     "+v"(ring[BI][1][0]), "+v"(ring[BI][1][1]), "+v"(ring[BI][1][2]), "+v"(ring[BI][1][3]) : "n"(CNT) : "memory");
    else if constexpr (KP == 2)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ring[BI][0][0]), "+v"(ring[BI][0][1]), "+v"(ring[BI][1][0]), "+v"(ring[BI][1][1])
                     : "n"(CNT) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ring[BI][0][0]), "+v"(ring[BI][1][0]) : "n"(CNT) : "memory");
}

// (no minimum-occupancy launch bound for the narrow forms: a register cap makes the compiler SPILL ring registers right behind
// their asm loads — it believes them defined — and a spilled register is stored before its data has arrived;
// tools/check_dense_ring.py checks every instantiation for spill code)
template <class T, int KP>
__global__ void __launch_bounds__(FBLOCK)
k_dense_fused(DenseFusedArgs<T> a, ElemParams<T> P) {
    constexpr int N = PackN<T>::N;
    __shared__ double sh_w[2][FWAVES][FT];      // (by step parity: a wave may write the next step's before a late one has read this step's)
    __shared__ double sh_p[2][FG_MAX][FT];
    __shared__ int sh_dead;
    __shared__ T sh_rp[5][FRC];               // the group's rows: b, mu, mu*y, lower and upper bound
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // workgroup -> (row group, column slice): the slices of one group on one XCD
    int lin = (int)blockIdx.x;
    const int nblk = (int)gridDim.x;
    if ((nblk & 7) == 0) lin = ((int)blockIdx.x & 7) * (nblk >> 3) + ((int)blockIdx.x >> 3);
    const int group = lin / a.G, c = lin - group * a.G;
    const int64_t npk = a.n / N;
    unsigned off[KP];                    // byte offset of this lane's k-th pack inside a row (one 32-bit register per pack, rows
    bool okk[KP];                        // addressed as uniform base + offset)
    Pack<T> xs[KP], acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int64_t j = ((int64_t)c * KP + k) * FBLOCK + t;
        okk[k] = j < npk;
        off[k] = okk[k] ? (unsigned)(j * 16) : 0u;      // (a lane beyond the last column reads column 0 and multiplies it by x = 0)
        xs[k] = okk[k] ? ldo<T, false>(a.x, off[k]) : splat(T(0));
        acc[k] = splat(T(0));
    }
    if (t == 0) sh_dead = 0;
    const int64_t r0 = (int64_t)group * a.rows_per_group;
    const int64_t r1 = (r0 + a.rows_per_group < a.ny) ? r0 + a.rows_per_group : a.ny;
    const int nph = r0 < r1 ? (int)((r1 - r0 + FT - 1) / FT) : 0;
    double pen = 0.0;
    unsigned long long* const mbox = a.mail + (size_t)group * (FMS * FG_MAX * FT * 2);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    using V = typename PackVec<T>::type;
    V ring[FNB][FT][KP];
    auto load_tile = [&](auto BI, const int q) { dense_load_tile<T, KP, decltype(BI)::value>(ring, off, a.A, a.n, r0, r1, q); };
    // one slice's FT tagged pairs of tile sequence `seq` -> sh_p[par][sl][.]
    // (early: the first poll was issued at the top of the step and `v` is its destination: only its wait is left — the L2
    // round trip of a poll, ~0.5 us, runs behind the wait for the tile and its products instead of in front of the barrier)
    auto poll_slice = [&](const int par, const int sl, const unsigned long long seq, bz_u8v& v, const bool early) {
        const unsigned tag32 = ll_tag(seq);
        const unsigned long long tag = (unsigned long long)tag32 << 32;
        const unsigned long long* src = mbox + ((size_t)(seq % FMS) * FG_MAX + sl) * (FT * 2);
        unsigned spins = 0;
        bool first = early;
        for (;;) {
            if (first) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
            else asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(src) : "memory");
            first = false;
            bool ok = true;
#pragma unroll
            for (int r = 0; r < FT; ++r) ok = ok && v[4 * r + 1] == tag32 && v[4 * r + 3] == tag32;
            if (ok) {
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < FT; ++r)
                        sh_p[par][sl][r] = __longlong_as_double((long long)((unsigned long long)v[4 * r] | ((unsigned long long)v[4 * r + 2] << 32)));
                }
                return;
            }
            ++spins;
            if ((spins & 4095u) == 0u) {                              // (rare: vector loads, coherent wherever the slices run)
                unsigned long long w0 = 0, w1 = 0;
                if (lane < FT) { w0 = dev_load(src + 2 * lane); w1 = dev_load(src + 2 * lane + 1); }
                const bool mine = lane >= FT || ((w0 >> 32 << 32) == tag && (w1 >> 32 << 32) == tag);
                if (__all(mine)) {
                    if (lane < FT) sh_p[par][sl][lane] = __longlong_as_double((long long)((w0 & 0xFFFFFFFFull) | (w1 << 32)));
                    return;
                }
            }
            if (spins > a.spin) {
                if (lane == 0) { sh_dead = 1; *a.timeout = 8; }
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // Step p: the partial products of tile p are formed and posted; tile p - 1 — whose partials everybody posted a step ago —
    // gets its yhat and is accumulated, and its ring slot takes tile p + 3.  BI = p mod 4.  Every step runs the same
    // instruction stream with the same loads (steps beyond the last tile work on re-reads of the last row and skip the
    // accumulation): the number of loads in flight at every wait is then known to the compiler, which otherwise waits for
    // ALL of them — the three prefetched tiles included — at the first use.
    auto step = [&](auto BI, const int p) -> bool {
        constexpr int bi = decltype(BI)::value, bprev = (bi + FNB - 1) % FNB;
        const int par = p & 1;
        // the polls of tile p - 1's partials go out first: posted a step ago, they are there, and the round trip to L2 runs
        // behind the wait for tile p and its products (nothing between here and the poll's wait touches pv0 / pv1, and every
        // path from here reaches that wait — no wave ends with a scalar load in flight: tools/check_dense_ring.py).
        // Tried beyond this: consuming the partials two steps late — with two tiles in flight instead of three 389 us, with
        // a fifth ring slot (253 VGPRs) 353 us, the same as this form; and issuing the polls a whole step ahead, right
        // behind the barrier: slower (374 us).
        bz_u8v pv0, pv1;
        const bool early = p > 0 && a.sabotage != 2;
        const unsigned long long seqp = a.seq0 + (unsigned long long)(p - 1);
        if (early) {
            const unsigned long long* src = mbox + ((size_t)(seqp % FMS) * FG_MAX + wave_u) * (FT * 2);
            if (wave_u < a.G) asm volatile("s_load_dwordx8 %0, %1, 0x0 glc" : "=&s"(pv0) : "s"(src) : "memory");
            if (wave_u + FWAVES < a.G) asm volatile("s_load_dwordx8 %0, %1, 0x0 glc" : "=&s"(pv1) : "s"(src + (size_t)FWAVES * (FT * 2)) : "memory");
        }
        dense_wait_tile<T, KP, bi>(ring);
#pragma unroll
        for (int r = 0; r < FT; ++r) {
            T s0 = T(0), s1 = T(0);
#pragma unroll
            for (int k = 0; k < KP; ++k)
#pragma unroll
                for (int e = 0; e < N; e += 2) {
                    s0 = fma_t(ring[bi][r][k][e], xs[k].v[e], s0);
                    s1 = fma_t(ring[bi][r][k][e + 1], xs[k].v[e + 1], s1);
                }
            const double v = wave_sum((double)s0 + (double)s1);
            if (lane == 0) sh_w[par][wave][r] = v;
        }
        if (p > 0) {
            if (a.sabotage == 2) {                                    // (timing experiment: no exchange, wrong values)
                if (t < FT) for (int g = 0; g < a.G; ++g) sh_p[par][g][t] = 1.0;
            } else {
                if (wave_u < a.G) poll_slice(par, wave_u, seqp, pv0, early);
                if (wave_u + FWAVES < a.G) poll_slice(par, wave_u + FWAVES, seqp, pv1, early);
            }
        }
        __syncthreads();
        if (sh_dead) return false;
        if (t < FT) {
            const unsigned long long seq = a.seq0 + (unsigned long long)p;
            const unsigned long long tag = (unsigned long long)ll_tag(seq) << 32;
            const double s = ((sh_w[par][0][t] + sh_w[par][1][t]) + (sh_w[par][2][t] + sh_w[par][3][t])) +
                             ((sh_w[par][4][t] + sh_w[par][5][t]) + (sh_w[par][6][t] + sh_w[par][7][t]));
            const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
            unsigned long long* dst = mbox + (((size_t)(seq % FMS) * FG_MAX + c) * FT + t) * 2;
            const unsigned long long ptag = (a.sabotage == 1 && c == 0) ? tag ^ (1ull << 62) : tag;
            dev_store(dst + 0, ptag | (bits & 0xFFFFFFFFull));
            dev_store(dst + 1, ptag | (bits >> 32));
        }
        const int64_t prow0 = r0 + (int64_t)(p - 1) * FT;            // first row of tile p - 1
#pragma unroll
        for (int r = 0; r < FT; ++r) {
            const int64_t row = prow0 + r;
            if (row < r0 || row >= r1) break;                        // (uniform over the workgroup)
            const int rc = (int)(row - r0);
            double dot = 0.0;
            for (int g = 0; g < a.G; ++g) dot += sh_p[par][g][r];     // slice order: the same bits in every slice's workgroup
            // eval!(cx, c, x) ; yupd = cx + muy ; proj!(s, D, yupd) ; yupd -= s ; sum(yupd^2 / mu) ; yupd /= mu   (auglagfun.jl:74-79)
            const T dt = (T)dot;
            const T cxv = a.b ? dt - sh_rp[0][rc] : dt;
            const T mu = sh_rp[1][rc], muy = sh_rp[2][rc], lo = sh_rp[3][rc], hi = sh_rp[4][rc];
            T tt = cxv + muy;
            const T sv = proj_D(P.D_kind, tt, lo, hi);
            tt = tt - sv;
            const T pterm = (tt * tt) / mu;
            const T yh = tt / mu;
            if (c == 0 && t == 0) {
                a.cx[row] = cxv;
                if (a.cx2) a.cx2[row] = cxv;
                pen += (double)pterm;
            }
#pragma unroll
            for (int k = 0; k < KP; ++k)
#pragma unroll
                for (int e = 0; e < N; ++e) acc[k].v[e] = fma_t(yh, ring[bprev][r][k][e], acc[k].v[e]);
        }
        load_tile(std::integral_constant<int, bprev>{}, p + FNB - 1);
        return true;
    };

    if (nph > 0) {
        // the group's b, mu, mu*y and bounds into LDS, once (read per tile as vector loads they would queue behind the
        // prefetched tiles — a wave's loads return in order)
        for (int i = t; i < (int)(r1 - r0); i += FBLOCK) {
            const int64_t row = r0 + i;
            sh_rp[0][i] = a.b ? a.b[row] : T(0);
            sh_rp[1][i] = P.uni >= 1 ? P.mu_uniform : P.mu[row];
            sh_rp[2][i] = P.uni >= 2 ? T(0) : P.muy[row];
            sh_rp[3][i] = P.D_lo_vec ? P.D_lo_vec[row] : P.D_lo;
            sh_rp[4][i] = P.D_hi_vec ? P.D_hi_vec[row] : P.D_hi;
        }
        // (x is needed in registers BEFORE the ring starts: a wait the compiler places at its first use inside the loop would be
        // a vmcnt(0) there — it does not see the ring's loads — and drain the ring at every turn)
#pragma unroll
        for (int k = 0; k < KP; ++k)
#pragma unroll
            for (int e = 0; e < N; ++e) asm volatile("" : "+v"(xs[k].v[e]));
        load_tile(std::integral_constant<int, 0>{}, 0);
        load_tile(std::integral_constant<int, 1>{}, 1);
        load_tile(std::integral_constant<int, 2>{}, 2);
        const int nsteps = (nph + 1 + FNB - 1) / FNB * FNB;          // steps 0 .. nph, padded to whole turns of the ring
        for (int p = 0; p < nsteps; p += FNB) {                      // (a poll that gave up ends the workgroup: no path skips a step)
            if (!step(std::integral_constant<int, 0>{}, p)) return;
            if (!step(std::integral_constant<int, 1>{}, p + 1)) return;
            if (!step(std::integral_constant<int, 2>{}, p + 2)) return;
            if (!step(std::integral_constant<int, 3>{}, p + 3)) return;
        }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k)
        if (okk[k]) sto<T, false>(a.part + (int64_t)group * a.pstride, off[k], acc[k]);
    if (c == 0 && t == 0) a.parts[(size_t)a.slot_pen * PSTRIDE + group] = pen;
}

// Halo exchange of the row-block-sharded stencil: this rank's first grid row goes to the previous rank's
// "south" halo and its last row to the next rank's "north" halo, straight into the neighbours' HBM through
// the IPC mapping (plain 16-byte stores), then a system-scope release and a sequence-number flag; then wait
// for the two rows this rank is owed.  Double-buffered by the parity of the sequence number: a neighbour
// can run at most one exchange ahead, because finishing an exchange needs this rank's flag of the same
// number.  The halo region is fine-grained memory, so the consumer kernel's plain loads see the rows.
template <class T> struct HaloArgs {
    const T* first_row;          // ny values
    const T* last_row;
    T* prev_south;               // previous rank's south-halo row of this parity (null: no previous rank)
    T* next_north;               // next rank's north-halo row of this parity (null: no next rank)
    unsigned long long* prev_flag;   // the flag words this rank sets at its neighbours
    unsigned long long* next_flag;
    const unsigned long long* my_north_flag;   // the flag words its neighbours set here
    const unsigned long long* my_south_flag;
    unsigned long long seq;
    int64_t ny;
    int* timeout;
};
template <class T>
__global__ void __launch_bounds__(XBLOCK) k_halo_exchange(HaloArgs<T> a) {
    constexpr int N = PackN<T>::N;
    const int64_t packs = a.ny / N;
    for (int64_t c = threadIdx.x; c < packs; c += XBLOCK) {
        if (a.prev_south) st(a.prev_south, c * N, N, ld(a.first_row, c * N, N));
        if (a.next_north) st(a.next_north, c * N, N, ld(a.last_row, c * N, N));
    }
    __threadfence_system();          // every storing lane: rows before flags
    __syncthreads();
    if (threadIdx.x == 0 && a.prev_south) sys_store(a.prev_flag, a.seq);
    if (threadIdx.x == 1 && a.next_north) sys_store(a.next_flag, a.seq);
    if (threadIdx.x < 2) {
        const unsigned long long* f = threadIdx.x == 0 ? a.my_north_flag : a.my_south_flag;
        const bool owed = threadIdx.x == 0 ? a.prev_south != nullptr : a.next_north != nullptr;
        unsigned spins = 0;
        while (owed && sys_load(f) != a.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > XSPIN_LIMIT) { *a.timeout = 4; break; }
        }
    }
    __threadfence_system();
}

// Row-sharded dense constraint (x replicated, the rows of A and the ny-vectors sharded): this rank's partial
// of A' yhat (n values) goes into every rank's region, slot [parity][rank], through the IPC mapping; the caller
// then sums the nranks partials in rank order (k_gemv_t_finish with the region as its chunk array), so all
// ranks hold the same bits.  Same protocol as k_halo_exchange (plain stores, release, flag per source rank).
template <class T> struct VecXchgArgs {
    const T* local;                  // npad values
    T* peer_slot[8];                 // every rank's region, this rank's slot of this parity
    unsigned long long* peer_flag[8];
    const unsigned long long* my_flags;      // [nranks] of this parity
    unsigned long long seq;
    int64_t npad;
    int nranks;
    int* timeout;
    unsigned long long* done;        // monotonic arrival counter of this kernel's workgroups
    unsigned long long target;       // its value once every workgroup of THIS launch has arrived
};
// Many workgroups copy (2 MB per exchange at cfg-4 sizes is too much for one); each drains its stores with a
// system-scope release and arrives at a counter; the workgroup that arrives last raises the flags at the
// peers and waits for the flags this rank is owed, so the kernel ends exactly when the exchange is complete.
template <class T>
__global__ void __launch_bounds__(BLOCK) k_vec_allgather(VecXchgArgs<T> a) {
    constexpr int N = PackN<T>::N;
    __shared__ int sh_last;
    const int64_t packs = a.npad / N;
    for (int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x; c < packs; c += (int64_t)gridDim.x * BLOCK) {
        const Pack<T> v = ld(a.local, c * N, N);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < a.nranks) st(a.peer_slot[r], c * N, N, v);
    }
    __threadfence_system();          // every storing lane: its rows before its arrival
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long old = __hip_atomic_fetch_add(a.done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = (old + 1ull == a.target) ? 1 : 0;
    }
    __syncthreads();
    if (!sh_last) return;
    __threadfence_system();
    if (threadIdx.x < a.nranks) {
        sys_store(a.peer_flag[threadIdx.x], a.seq);
        unsigned spins = 0;
        while (sys_load(a.my_flags + threadIdx.x) != a.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > XSPIN_LIMIT) { *a.timeout = 5; break; }
        }
    }
    __threadfence_system();
}

// RCCL transport: fold this rank's block partials of slots [first, first+cnt) into the send buffer (one
// wave per slot)
static __global__ void __launch_bounds__(64)
k_pack(const double* parts, SlotCounts counts, int first, int cnt, unsigned maxmask, double* send, int rank,
       unsigned keepmask) {
    const int i = blockIdx.x;
    double t = fold_wave(parts + (size_t)(first + i) * PSTRIDE, counts.get(i), (maxmask >> i) & 1u);
    if (rank != 0 && !((keepmask >> i) & 1u)) t = 0.0;
    if (threadIdx.x == 0) send[first + i] = t;
}

constexpr int PBLOCK = 512;
constexpr int PWAVES = PBLOCK / 64;
constexpr int PMAXMEM = 16;
constexpr unsigned PSPIN_LIMIT = 400000u;      // ~1.5 us per poll: gives up after about half a second

template <class T> struct PersistArgs {
    const T* res;
    const T* S[PMAXMEM];     // newest first
    const T* Y[PMAXMEM];
    T ys[PMAXMEM];
    T H;
    int m;
    int nb;                  // blocks in the grid (all resident: one per CU)
    T* d_out;
    int64_t n;
    double* parts;
    double* alphas;
    unsigned long long* counter;
    unsigned long long base; // counter value when this launch starts
    int* timeout;            // host-visible flag, set if a spin gives up
    unsigned long long* abort_flag;   // device word raised with it: the other workgroups (and the later phases) stop polling
    int slot_loop1, slot_loop2;
    // multi-GPU (x sharded): phase totals are exchanged through the peers' mailboxes
    int nranks, rank;
    unsigned long long pseq; // sequence number of this launch's first phase
    P2PWords* mbox_local;
    P2PWords* mbox_peer[8];
    double* gtot;            // [2] global total of the current phase, published by block 0
    unsigned long long* gflag;
    double* final_tot;       // global <y_0, d> for the kernel that fuses the last axpy
};

__device__ __forceinline__ double block_sum512(double v, double* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = ((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]));
    __syncthreads();
    return t;
}

// grid-wide phase boundary, split in two so that the loads of the next phase can be issued between
// "arrive" and "wait": the memory system stays busy while the barrier completes.
// Every byte handed between workgroups is an 8-byte agent-scope atomic on BOTH sides (sc1 store /
// sc1 load: written through to memory, never served from a non-coherent L1/L2 copy), the storing
// lane drains its store (s_waitcnt vmcnt(0)) before it signals, and exactly one lane per workgroup
// signals: the hand-off form measured valid on gfx950 without release/acquire fences
// (MI355X_MICROARCH.md, "Valid forms" table row 1), which saves ~1.7 us + ~1.7 us per barrier.
// The arrival counter is sharded 8 ways (blockIdx % 8 = the XCD under round-robin dispatch; a
// speed assumption only) so at most nb/8 adds contend per word; 8 lanes poll the 8 shards.
//   arrive: publish this block's partial, drain, one atomic add on this block's shard
//   wait  : bounded poll until the shards sum to `target`, then every block folds the nb partials
//           in the same order -> the same bits everywhere
constexpr int PSHARDS = 8;
constexpr int PSHARD_STRIDE = 16;      // unsigned long long words: 128 B between shards

__device__ __forceinline__ void grid_arrive(double acc, double* row, unsigned long long* counter, double* sh) {
    const double part = block_sum512(acc, sh);
    if (threadIdx.x == 0) {
        __hip_atomic_store(row + blockIdx.x, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter + (blockIdx.x % PSHARDS) * PSHARD_STRIDE, 1ull, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ double grid_wait_fold(double* row, int nb, unsigned long long* counter,
                                                 unsigned long long target, int* timeout, double* sh,
                                                 unsigned long long* abort_flag = nullptr) {
    if (threadIdx.x < 64) {            // wave 0 polls: lanes 0..7 read one shard each
        unsigned spins = 0;
        for (;;) {
            unsigned long long c = 0;
            if (threadIdx.x < PSHARDS)
                c = __hip_atomic_load(counter + threadIdx.x * PSHARD_STRIDE, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
            // sum of the 8 shards (lanes >= 8 contribute 0), broadcast from lane 0
            unsigned lo = (unsigned)c, hi = (unsigned)(c >> 32);
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {
                unsigned long long t = ((unsigned long long)__shfl_down(hi, o, 64) << 32) | __shfl_down(lo, o, 64);
                c += t;
                lo = (unsigned)c; hi = (unsigned)(c >> 32);
            }
            const unsigned long long tot = ((unsigned long long)__shfl(hi, 0, 64) << 32) | __shfl(lo, 0, 64);
            if (tot >= target) break;
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            // a grid whose workgroups are not all resident never completes this barrier: the first poller to give up
            // tells the host and raises a device flag that ends everybody else's polling (this phase and the later ones)
            if (abort_flag && (spins & 255u) == 255u &&
                __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
            if (spins > PSPIN_LIMIT) {
                if (threadIdx.x == 0) {
                    *timeout = 1;
                    if (abort_flag) __hip_atomic_store(abort_flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: keep the loads below the poll
    }
    __syncthreads();
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += PBLOCK)
        v += __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return block_sum512(v, sh);
}

// Multi-GPU form of the phase boundary.  Block 0 waits for the local arrivals, folds the local
// partials, exchanges the rank totals through the mailboxes (rank-ordered sum -> identical bits on all
// ranks) and publishes the global total locally; the other blocks only wait for that publication.
template <class A>
__device__ __forceinline__ double grid_wait_fold_multi(const A& a, double* row, unsigned long long target,
                                                       unsigned long long q, double* sh) {
    const int par = (int)(q & 1ull);
    __shared__ double gsh;
    if (blockIdx.x == 0) {
        const double local = grid_wait_fold(row, a.nb, a.counter, target, a.timeout, sh);
        const int tid = threadIdx.x;
        if (tid < a.nranks) sys_store(&a.mbox_peer[tid]->pval[par][a.rank], local);
        __threadfence_system();
        __syncthreads();
        if (tid < a.nranks) sys_store(&a.mbox_peer[tid]->pflag[par][a.rank], q);
        if (tid < a.nranks) {
            unsigned spins = 0;
            while (sys_load(&a.mbox_local->pflag[par][tid]) < q) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > XSPIN_LIMIT) { *a.timeout = 3; break; }
            }
        }
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            double g = 0.0;
            for (int r = 0; r < a.nranks; ++r) g += sys_load(&a.mbox_local->pval[par][r]);
            gsh = g;
            __hip_atomic_store(a.gtot + par, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(a.gflag, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        return gsh;
    }
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(a.gflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < q) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > PSPIN_LIMIT) { *a.timeout = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        gsh = __hip_atomic_load(a.gtot + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const double g = gsh;
    __syncthreads();
    return g;
}

// Stream KR rounds of one or two vectors past the register-resident d with an explicit two-deep
// software pipeline: the loads of the next group of G rounds are issued before the current group
// is consumed, and sched_barriers keep the compiler from sinking them back to their uses (with d
// holding 160 of the 256 VGPRs its scheduler otherwise keeps only 2 loads in flight).  The first
// group is loaded by persist_prefetch, which the kernel calls BEFORE waiting on the phase barrier.
template <class T, int KR, int G, bool TWO>
__device__ __forceinline__ void persist_prefetch(const T* __restrict__ p0, const T* __restrict__ p1,
                                                 int64_t first, int stride_e, Pack<T> (&pv)[G],
                                                 Pack<T> (&pw)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g < KR) {
            const int64_t i0 = first + (int64_t)g * stride_e;
            pv[g] = *reinterpret_cast<const Pack<T>*>(p0 + i0);
            if (TWO) pw[g] = *reinterpret_cast<const Pack<T>*>(p1 + i0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// KR need not be a multiple of G: the last group is simply shorter (all bounds are compile-time).
template <class T, int KR, int G, bool TWO, class F>
__device__ __forceinline__ void persist_stream(const T* __restrict__ p0, const T* __restrict__ p1,
                                               int64_t first, int stride_e, Pack<T> (&pv)[G],
                                               Pack<T> (&pw)[G], F&& f) {
    Pack<T> qv[G], qw[G];
#pragma unroll
    for (int kb = 0; kb < KR; kb += G) {
        const bool even = ((kb / G) & 1) == 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (kb + G + g < KR) {
                const int64_t i0 = first + (int64_t)(kb + G + g) * stride_e;
                if (even) {
                    qv[g] = *reinterpret_cast<const Pack<T>*>(p0 + i0);
                    if (TWO) qw[g] = *reinterpret_cast<const Pack<T>*>(p1 + i0);
                } else {
                    pv[g] = *reinterpret_cast<const Pack<T>*>(p0 + i0);
                    if (TWO) pw[g] = *reinterpret_cast<const Pack<T>*>(p1 + i0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (kb + g < KR) {
                if (even) f(kb + g, pv[g], TWO ? pw[g] : pv[g]);
                else f(kb + g, qv[g], TWO ? qw[g] : qv[g]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class T, int KR>
__global__ void __launch_bounds__(PBLOCK, 2)
k_twoloop_persist(PersistArgs<T> a) {
    constexpr int N = PackN<T>::N;
    __shared__ double sh[PWAVES];
    __shared__ T alpha_sh[PMAXMEM];
    __shared__ int stride_sh;
    const int m = a.m, nb = a.nb;
    if (threadIdx.x == 0) stride_sh = nb * PBLOCK * N;
    __syncthreads();
    Pack<T> d[KR];
    unsigned long long target = a.base;
    unsigned long long q = a.pseq;          // multi-GPU: sequence number of the coming phase
    auto phase_total = [&](double* row) -> double {
        if (a.nranks <= 1) return grid_wait_fold(row, nb, a.counter, target, a.timeout, sh, a.abort_flag);
        const double g = grid_wait_fold_multi(a, row, target, q, sh);
        ++q;
        return g;
    };
    const int64_t first = ((int64_t)blockIdx.x * PBLOCK + threadIdx.x) * N;

    // Per phase the element stride between a thread's packs is re-read from LDS through a volatile
    // pointer: it keeps the KR pack offsets from being hoisted out of the phase loops (they would
    // occupy 2 registers per pack for the whole kernel, next to the 4 per pack that d needs).
    // Every vector this kernel touches is allocated zero-padded to KR*nb*512 packs (host side), so all
    // rounds of all threads are in-bounds and the padding contributes exact zeros: no masks, no
    // branches.  The element stride between a thread's packs is re-read per phase through a volatile
    // LDS pointer so the KR pack offsets are not hoisted out of the phase loops (2 VGPRs each).
#define BZ_P_STRIDE (*(volatile int*)&stride_sh)
    // pipeline depth: 4 rounds per group; 2 above 40 packs so that d may take up to 192 of the 256 VGPRs
    // (48 packs: 12.58 M doubles per GPU, one BASELINE config-5 shard)
    constexpr int G = KR > 40 ? 2 : 4;
    Pack<T> pv[G], pw[G];      // first group of the coming phase, loaded across the phase barrier

    // phase 0: d = -res ; <s_0, d>
    double acc = 0.0;
    persist_prefetch<T, KR, G, true>(a.res, a.S[0], first, BZ_P_STRIDE, pv, pw);
    persist_stream<T, KR, G, true>(a.res, a.S[0], first, BZ_P_STRIDE, pv, pw,
        [&](int k, const Pack<T>& r, const Pack<T>& s) {
#pragma unroll
            for (int e = 0; e < N; ++e) {
                T o = T(-1) * r.v[e];
                d[k].v[e] = o;
                acc += (double)(s.v[e] * o);
            }
        });
    // loop 1: d -= alpha_j y_j ; <s_{j+1}, d>          (j = 0 .. m-2)
    for (int j = 0; j + 1 < m; ++j) {
        double* row = a.parts + (size_t)(a.slot_loop1 + j) * PSTRIDE;
        target += nb;
        grid_arrive(acc, row, a.counter, sh);
        persist_prefetch<T, KR, G, true>(a.Y[j], a.S[j + 1], first, BZ_P_STRIDE, pv, pw);
        const double tot = phase_total(row);
        const T al = T(tot) / a.ys[j];
        if (threadIdx.x == 0) alpha_sh[j] = al;
        const T coef = -al;
        acc = 0.0;
        persist_stream<T, KR, G, true>(a.Y[j], a.S[j + 1], first, BZ_P_STRIDE, pv, pw,
            [&](int k, const Pack<T>& v, const Pack<T>& w) {
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    T t = coef * v.v[e];
                    T o = d[k].v[e] + t;
                    d[k].v[e] = o;
                    acc += (double)(w.v[e] * o);
                }
            });
    }
    // middle: d = H (d - alpha_{m-1} y_{m-1}) ; <y_{m-1}, d>
    {
        const int j = m - 1;
        double* row = a.parts + (size_t)(a.slot_loop1 + j) * PSTRIDE;
        target += nb;
        grid_arrive(acc, row, a.counter, sh);
        persist_prefetch<T, KR, G, false>(a.Y[j], a.Y[j], first, BZ_P_STRIDE, pv, pw);
        const double tot = phase_total(row);
        const T al = T(tot) / a.ys[j];
        if (threadIdx.x == 0) alpha_sh[j] = al;
        const T coef = -al;
        const T Hs = a.H;
        acc = 0.0;
        persist_stream<T, KR, G, false>(a.Y[j], a.Y[j], first, BZ_P_STRIDE, pv, pw,
            [&](int k, const Pack<T>& v, const Pack<T>&) {
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    T t = coef * v.v[e];
                    T o = d[k].v[e] + t;
                    o = Hs * o;
                    d[k].v[e] = o;
                    acc += (double)(v.v[e] * o);
                }
            });
    }
    __syncthreads();    // alpha_sh complete
    // loop 2: d += (alpha_j - beta_j) s_j ; <y_{j-1}, d>   (j = m-1 .. 1)
    for (int j = m - 1; j >= 1; --j) {
        double* row = a.parts + (size_t)(a.slot_loop2 + j) * PSTRIDE;
        target += nb;
        grid_arrive(acc, row, a.counter, sh);
        persist_prefetch<T, KR, G, true>(a.S[j], a.Y[j - 1], first, BZ_P_STRIDE, pv, pw);
        const double tot = phase_total(row);
        const T beta = T(tot) / a.ys[j];
        const T coef = alpha_sh[j] - beta;
        acc = 0.0;
        persist_stream<T, KR, G, true>(a.S[j], a.Y[j - 1], first, BZ_P_STRIDE, pv, pw,
            [&](int k, const Pack<T>& v, const Pack<T>& w) {
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    T t = coef * v.v[e];
                    T o = d[k].v[e] + t;
                    d[k].v[e] = o;
                    acc += (double)(w.v[e] * o);
                }
            });
    }
    // last partial (<y_0, d>) and d go to memory: the next kernel fuses the final axpy.  On several GPUs
    // one more phase turns the partials into the global total (the consumer cannot fold across ranks).
    if (a.nranks > 1) {
        double* row = a.parts + (size_t)(a.slot_loop2 + 0) * PSTRIDE;
        target += nb;
        grid_arrive(acc, row, a.counter, sh);
        const double tot = phase_total(row);
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.final_tot = tot;
        if (blockIdx.x == 0 && threadIdx.x < m) a.alphas[threadIdx.x] = (double)alpha_sh[threadIdx.x];
    } else {
        const double part = block_sum512(acc, sh);
        if (threadIdx.x == 0) a.parts[(size_t)(a.slot_loop2 + 0) * PSTRIDE + blockIdx.x] = part;
        if (blockIdx.x == 0 && threadIdx.x < m) a.alphas[threadIdx.x] = (double)alpha_sh[threadIdx.x];
    }
    {
        const int stride_e = BZ_P_STRIDE;
#pragma unroll
        for (int k = 0; k < KR; ++k)
            *reinterpret_cast<Pack<T>*>(a.d_out + first + (int64_t)k * stride_e) = d[k];
    }
#undef BZ_P_STRIDE
}

// ---------------------------------------------------------------------------
// K3: forward-backward step  y = x - gamma*g ; z = prox(y) ; res = x - z
//   slots: +0 sum g terms (multiply by lambda on the host), +1 <g,res>, +2 ||res||^2
//   g == nullptr: pure prox of x (used for prox_{eps g}(x0), alps.jl:38)
// ---------------------------------------------------------------------------
template <class T, bool LP = false>
__global__ void __launch_bounds__(BLOCK)
k_fbstep(const T* __restrict__ x, const T* __restrict__ g, T gamma, ElemParams<T> P,
         T* __restrict__ z, T* __restrict__ res, int64_t n, double* __restrict__ parts,
         int slot0) {
    double acc[3] = {0.0, 0.0, 0.0};
    const T gl = gamma * P.g_lambda;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, false, false, true);
        Pack<T> px = ld(x, i0, cnt);
        Pack<T> pg = g ? ld(g, i0, cnt) : splat(T(0));
        Pack<T> pz, pr;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T y = px.v[e];
            if (g) { T t = gamma * pg.v[e]; y = px.v[e] - t; }
            T gterm;
            T zz = prox_elem<T, LP>(P.g_kind, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm, P.g_p);
            T r = px.v[e] - zz;
            pz.v[e] = zz; pr.v[e] = r;
            if (e < cnt) {
                acc[0] += (double)gterm;
                acc[1] += (double)(pg.v[e] * r);
                acc[2] += (double)(r * r);
            }
        }
        st(z, i0, cnt, pz);
        if (res) st(res, i0, cnt, pr);
    });
    block_reduce_store<3>(acc, 0u, parts, slot0);
}

// ---------------------------------------------------------------------------
// ALS (slack-variable form, src/algorithms/als.jl, src/utilities/auglagfunslack.jl): the inner
// solver works on xs = [x; s] of length nx + ny.  c = Identity (ny == nx), element-wise f.
// ---------------------------------------------------------------------------
// gradient!(dFxs, F::AugLagFunSlack, xs)  (auglagfunslack.jl:78-97)
//   w = (cx + muy) - s ; Fxs = 0.5 sum w^2/mu + fx - musqy ; yupd = y + (cx - s)/mu
//   dFxs = [dfx + yupd ; -yupd]          slots: +0 sum f terms, +1 sum w^2/mu
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_algrad_slack_elem(const T* __restrict__ xs, ElemParams<T> P, const T* __restrict__ yv,
                    T* __restrict__ grad, int64_t nx, double* __restrict__ parts, int slot0) {
    double acc[2] = {0.0, 0.0};
    bz_for_chunks<T>(nx, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> px = ld(xs, i0, cnt), ps = ld(xs + nx, i0, cnt);
        Pack<T> pq = splat(T(0)), pb = splat(T(0));
        if (P.f_kind == BZ_F_DIAG_QUADRATIC) { pq = ld(P.q, i0, cnt); pb = ld(P.b, i0, cnt); }
        Pack<T> pmu = ld(P.mu, i0, cnt), pmuy = ld(P.muy, i0, cnt), py = ld(yv, i0, cnt);
        Pack<T> gx, gs;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            const T x = px.v[e], sv = ps.v[e];
            T dfx = T(0), fterm = T(0);
            if (P.f_kind == BZ_F_DIAG_QUADRATIC) {
                T qx = pq.v[e] * x;
                dfx = qx - pb.v[e];
                fterm = x * (T(0.5) * qx - pb.v[e]);
            }
            const T cx = x;
            T w = cx + pmuy.v[e];
            w = w - sv;
            const T pterm = (w * w) / pmu.v[e];
            const T r = cx - sv;
            const T yupd = py.v[e] + r / pmu.v[e];
            gx.v[e] = dfx + yupd;
            gs.v[e] = -yupd;
            if (e < cnt) { acc[0] += (double)fterm; acc[1] += (double)pterm; }
        }
        if (grad) { st(grad, i0, cnt, gx); st(grad + nx, i0, cnt, gs); }
    });
    block_reduce_store<2>(acc, 0u, parts, slot0);
}

// prox!(z, G::NonsmoothCostFunSlack, xs, gamma)  (auglagfunslack.jl:136-154) after the forward step:
//   z = [prox_g(x - gamma g_x) ; proj_D(s - gamma g_s)] ; res = xs - z
//   slots: +0 sum g terms (x part), +1 <g, res>, +2 ||res||^2 (both parts)
template <class T, bool LP = false>
__global__ void __launch_bounds__(BLOCK)
k_fbstep_slack(const T* __restrict__ xs, const T* __restrict__ g, T gamma, ElemParams<T> P,
               T* __restrict__ z, T* __restrict__ res, int64_t nx, double* __restrict__ parts,
               int slot0) {
    double acc[3] = {0.0, 0.0, 0.0};
    const T gl = gamma * P.g_lambda;
    bz_for_chunks<T>(nx, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, false, false, true);
        Pack<T> dlo = P.D_lo_vec ? ld(P.D_lo_vec, i0, cnt) : splat(P.D_lo);
        Pack<T> dhi = P.D_hi_vec ? ld(P.D_hi_vec, i0, cnt) : splat(P.D_hi);
        Pack<T> px = ld(xs, i0, cnt), ps = ld(xs + nx, i0, cnt);
        Pack<T> pgx = g ? ld(g, i0, cnt) : splat(T(0)), pgs = g ? ld(g + nx, i0, cnt) : splat(T(0));
        Pack<T> zx, zs, rx, rs;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T yx = px.v[e], ys = ps.v[e];
            if (g) { T t = gamma * pgx.v[e]; yx = px.v[e] - t; T u = gamma * pgs.v[e]; ys = ps.v[e] - u; }
            T gterm;
            const T a = prox_elem<T, LP>(P.g_kind, yx, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm, P.g_p);
            const T b = proj_D(P.D_kind, ys, dlo.v[e], dhi.v[e]);
            const T r1 = px.v[e] - a, r2 = ps.v[e] - b;
            zx.v[e] = a; zs.v[e] = b; rx.v[e] = r1; rs.v[e] = r2;
            if (e < cnt) {
                acc[0] += (double)gterm;
                acc[1] += (double)(pgx.v[e] * r1);
                acc[1] += (double)(pgs.v[e] * r2);
                acc[2] += (double)(r1 * r1);
                acc[2] += (double)(r2 * r2);
            }
        }
        st(z, i0, cnt, zx); st(z + nx, i0, cnt, zs);
        if (res) { st(res, i0, cnt, rx); st(res + nx, i0, cnt, rs); }
    });
    block_reduce_store<3>(acc, 0u, parts, slot0);
}

// ALS dual update (als.jl:82-87), c = Identity: y += (cx - s)/mu ; slot +0 max |cx - s|
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_dual_update_slack(const T* __restrict__ xs, const T* __restrict__ mu, T* __restrict__ y,
                    int64_t nx, double* __restrict__ parts, int slot0) {
    double acc[1] = {0.0};
    bz_for_chunks<T>(nx, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> px = ld(xs, i0, cnt), ps = ld(xs + nx, i0, cnt), pm = ld(mu, i0, cnt), py = ld((const T*)y, i0, cnt);
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            const T r = px.v[e] - ps.v[e];
            py.v[e] = py.v[e] + r / pm.v[e];
            if (e < cnt) acc[0] = nanmax(acc[0], (double)(r < T(0) ? -r : r));
        }
        st(y, i0, cnt, py);
    });
    block_reduce_store<1>(acc, 1u, parts, slot0);
}

// ---------------------------------------------------------------------------
// K5+K7: L-BFGS pair + stopping norm
//   s = x - x_prev ; y = res - res_prev ; slots: +0 <s,y>, +1 <y,y>,
//   +2 max |res/gamma - gx + gz|
// ---------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_update(const T* __restrict__ x, const T* __restrict__ x_prev, const T* __restrict__ res,
         const T* __restrict__ res_prev, const T* __restrict__ gx, const T* __restrict__ gz,
         T gamma, T* __restrict__ s_new, T* __restrict__ y_new, int64_t n,
         double* __restrict__ parts, int slot0) {
    double acc[3] = {0.0, 0.0, 0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> px = ld(x, i0, cnt), pr = ld(res, i0, cnt);
        Pack<T> pxp = x_prev ? ld(x_prev, i0, cnt) : px;
        Pack<T> prp = res_prev ? ld(res_prev, i0, cnt) : pr;
        Pack<T> pgx = ld(gx, i0, cnt), pgz = ld(gz, i0, cnt);
        Pack<T> ps, py;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T s = px.v[e] - pxp.v[e];
            T y = pr.v[e] - prp.v[e];
            ps.v[e] = s; py.v[e] = y;
            T w = pr.v[e] / gamma;
            w = w - pgx.v[e];
            w = w + pgz.v[e];
            if (e < cnt) {
                acc[0] += (double)(s * y);
                acc[1] += (double)(y * y);
                acc[2] = nanmax(acc[2], (double)(w < T(0) ? -w : w));
            }
        }
        if (s_new) { st(s_new, i0, cnt, ps); st(y_new, i0, cnt, py); }
    });
    block_reduce_store<3>(acc, 4u, parts, slot0);
}

// acc + a*b for the inner products that only the compact form's own kernels produce (Gram products, p, w) and
// for the linear combination d: one fused multiply-add in fp64 (one rounding instead of two, half the
// instructions); fp32 keeps the rounded product, as the oracle's fp32 arithmetic has it
__device__ __forceinline__ double mul_acc(double a, double b, double acc) { return __builtin_fma(a, b, acc); }
__device__ __forceinline__ double mul_acc(float a, float b, double acc) { return acc + (double)(a * b); }
__device__ __forceinline__ double mul_add(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float mul_add(float a, float b, float c) { float t = a * b; return c + t; }

// ---------------------------------------------------------------------------
// The separable fast path: everything between the last two-loop reduction and the
// end of the iteration in ONE pass (legal because f', c = I, proj_D and prox_g are all
// element-wise, so x_d -> gradL(x_d) -> z -> gradL(z) -> (s, y) never leave registers).
//   reads : d, S_newest, x, res_prev, q, b, mu, mu*y          (8 vectors)
//   writes: x_d, z, res, s_new, y_new [, gradL(x_d), gradL(z)] (5 vectors)
//   slots : +0 f(x_d) +1 pen(x_d) +2 gsum +3 <g,res> +4 ||res||^2
//           +5 f(z) +6 pen(z) +7 <s,y> +8 <y,y> +9 max stop
//           +10 <s_new, -res>, +11 <y_new, -res> with res the new residual: p and w of the compact form's next
//           application when this pass ran with an empty memory (k_gram_dots's arithmetic)
// ---------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_fused_sep(TailArgs<T> a, const T* __restrict__ x, const T* __restrict__ res_prev,
            ElemParams<T> P, T gamma, T* __restrict__ x_d, T* __restrict__ z,
            T* __restrict__ res, T* __restrict__ s_new, T* __restrict__ y_new,
            T* __restrict__ gx_out, T* __restrict__ gz_out, int64_t n,
            double* __restrict__ parts, int slot0) {
    __shared__ double sh[WAVES];
    const T coef = tail_coef(a, sh);
    const T gl = gamma * P.g_lambda;
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, true, true, true);
        Pack<T> pin = ld(a.in, i0, cnt);
        Pack<T> pv = (a.mode != 2) ? ld(a.v, i0, cnt) : splat(T(0));
        Pack<T> px = ld(x, i0, cnt), prp = ld(res_prev, i0, cnt);
        Pack<T> pxd, pz, pr, ps, py, pg1, pg2, pgt;
        T f1[PackN<T>::N], p1[PackN<T>::N];
        // three sweeps over the pack (trial point; gradient + FB step; gradient at z): the pairwise D kinds read the
        // pair partner's value at each of the three points
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T d = tail_elem(a, coef, pin.v[e], pv.v[e]);
            pxd.v[e] = px.v[e] + d;
        }
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            const T xd = pxd.v[e];
            ALOut<T> o1 = al_elem(P.f_kind, P.D_kind, xd, L.q.v[e], L.b.v[e], L.mu.v[e],
                                  L.muy.v[e], L.dlo.v[e], L.dhi.v[e], pxd.v[e ^ 1] + L.muy.v[e ^ 1], e & 1);
            T t = gamma * o1.grad;
            T y = xd - t;
            T zz = prox_elem(P.g_kind, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], pgt.v[e]);
            pz.v[e] = zz; pr.v[e] = xd - zz;
            pg1.v[e] = o1.grad; f1[e] = o1.fterm; p1[e] = o1.pterm;
        }
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            const T xd = pxd.v[e], zz = pz.v[e], r = pr.v[e];
            const T gterm = pgt.v[e];
            ALOut<T> o1;
            o1.grad = pg1.v[e]; o1.fterm = f1[e]; o1.pterm = p1[e];
            ALOut<T> o2 = al_elem(P.f_kind, P.D_kind, zz, L.q.v[e], L.b.v[e], L.mu.v[e],
                                  L.muy.v[e], L.dlo.v[e], L.dhi.v[e], pz.v[e ^ 1] + L.muy.v[e ^ 1], e & 1);
            T s = xd - px.v[e];
            T yy = r - prp.v[e];
            T w = r / gamma;
            w = w - o1.grad;
            w = w + o2.grad;
            ps.v[e] = s; py.v[e] = yy;
            pg2.v[e] = o2.grad;
            if (e < cnt) {
                acc[0] += (double)o1.fterm;
                acc[1] += (double)o1.pterm;
                acc[2] += (double)gterm;
                acc[3] += (double)(o1.grad * r);
                acc[4] += (double)(r * r);
                acc[5] += (double)o2.fterm;
                acc[6] += (double)o2.pterm;
                acc[7] += (double)(s * yy);
                acc[8] += (double)(yy * yy);
                acc[9] = nanmax(acc[9], (double)(w < T(0) ? -w : w));
                const T nr = T(-1) * r;
                acc[10] = mul_acc(s, nr, acc[10]);
                acc[11] = mul_acc(yy, nr, acc[11]);
            }
        }
        st(x_d, i0, cnt, pxd);
        st(z, i0, cnt, pz);
        st(res, i0, cnt, pr);
        st(s_new, i0, cnt, ps);
        st(y_new, i0, cnt, py);
        if (gx_out) st(gx_out, i0, cnt, pg1);
        if (gz_out) st(gz_out, i0, cnt, pg2);
    });
    block_reduce_store<12>(acc, 1u << 9, parts, slot0);
}

// ---------------------------------------------------------------------------
// Compact (Byrd-Nocedal-Schnabel) form of the L-BFGS operator: all 2m inner products with v = -res are
// independent, so d = H(-res) needs ONE reduction phase instead of 2m sequential ones:
//     p = S'v, w = Y'v ; u1 = M1 p - H0 M2' w ; u2 = -M2 p ; d = H0 v + S u1 + H0 Y u2
// with M1 = R^-T (D + H0 Y'Y) R^-1, M2 = R^-1 (m x m, maintained by the host from the Gram products the
// kernels return).  Same operator as the two-loop recursion (oracle: LBFGSCompactOperator); used when x is
// sharded over several GPUs, where each reduction phase is a cross-GPU exchange.  MM = compile-time
// capacity (pairs are ordered oldest -> newest).
// ---------------------------------------------------------------------------
template <class T, int MM> struct CompactVecs {
    const T* S[MM];
    const T* Y[MM];
    int m;
};
// Gated pre-launch of the one-pass kernel (iterate-history form).  What the next iteration's launch needs — which ring
// slots hold the iterates, gamma, the grid — is known while the current pass still runs, except the coefficients u1,
// u2h, H0 (and whether z is stored), which the host computes from the current pass's 32 scalars.  So the next launch is
// made EARLY, right behind the read-back kernel, with a gate: workgroup 0's first lane polls a record in pinned host
// memory; when the host has the scalars it writes the coefficients and the launch's sequence number there (or the
// number with the ABORT bit if the iteration did not end the plain way: the kernel then leaves without touching
// anything).  Workgroup 0 copies the record to device memory and raises a device flag the other workgroups poll.
// That trades the launch call + dispatch (~9 us) between two passes for one PCIe poll (~1.5 us).  Every poll is bounded.
struct GateRec {
    // HOST record (pinned, written by the host, polled over PCIe by workgroup 0's first wave): the 13 values travel in
    // "LL" form like the scalar mailboxes — every 8-byte word carries half a value and the 32-bit tag of the launch it
    // is meant for, so a word is valid exactly when its tag matches and ONE round of 26 parallel loads fetches
    // everything (a sequence number followed by the payload would be two PCIe round trips, ~1.5 us each).
    //   w[2i], w[2i+1]: value i = u1[0..4], u2h[0..4], H0, z (address bits) ; w[31]: tag | 1 = leave without running
    unsigned long long w[32];
    // DEVICE copy (written by workgroup 0, read by everybody): seq = gate_seq (go) or gate_seq | GATE_ABORT
    unsigned long long seq;
    double val[13];
};
constexpr unsigned long long GATE_ABORT = 1ull << 63;
// Workgroup 0 gives up on the host after ~3 s (a host thread that was merely descheduled comes back long before; the
// launch then leaves as a whole — ABORT for everybody — and the host reports it).  The other workgroups never take a
// decision of their own: they wait for workgroup 0's verdict, and their (much longer) bound is only there so that no wave
// can spin forever; reaching it is an error the host reports, never a silently skipped part of the pass.
constexpr unsigned GATE_SPIN_HOST = 2500000u, GATE_SPIN_DEV = 60000000u;

template <int MM> struct CompactCoef {
    // coefficients of this application, computed by the host from p = S'v, w = Y'v (v = -res) and the Gram
    // matrices:  u1 = M1 p - H0 M2' w ;  u2h = H0 * (-(M2 p))     (entries beyond m are zero)
    double u1[MM];
    double u2h[MM];
    double H0;
    // iterate-history form: the gamma the OLDEST stored iterate's residual was formed with.  It differs from the
    // current one when the oldest pair is that of an iteration that halved gamma (y = res_new(gamma/2) -
    // res_prev(gamma), as upstream has it); a halving resets the memory, so no younger iterate can differ.
    double gam0;
    int uni_rt, trial_rt;      // the kernel's UNI / TRIAL when those template arguments are -1 (family instantiations)
    // gated pre-launch (gate_seq != 0): u1, u2h, H0 above are not used, they arrive through the gate
    unsigned long long gate_seq;
    const GateRec* gate_host;  // pinned host memory (device address)
    GateRec* gate_dev;
    int* gate_timeout;
    int gate_other_stream;     // launched on another stream than the pass before it (may be resident while that one runs)
    unsigned gate_spin_host, gate_spin_dev;      // poll bounds of workgroup 0 (host record) and of the others (device flag)
    int gate_late;             // the pipelined form goes to its gate AFTER issuing the loads of its first packs
};

// K1: p_i = <s_i, -res>, w_i = <y_i, -res>   slots: slot0 + i (p), slot0 + MM + i (w)
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_gram_dots(CompactVecs<T, MM> V, const T* __restrict__ res, int64_t n, double* __restrict__ parts,
            int slot0) {
    double acc[2 * MM];
#pragma unroll
    for (int k = 0; k < 2 * MM; ++k) acc[k] = 0.0;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pr = ld(res, i0, cnt);
#pragma unroll
        for (int i = 0; i < MM; ++i) {
            if (i < V.m) {
                Pack<T> ps = ld(V.S[i], i0, cnt), py = ld(V.Y[i], i0, cnt);
#pragma unroll
                for (int e = 0; e < PackN<T>::N; ++e)
                    if (e < cnt) {
                        const T v = T(-1) * pr.v[e];
                        acc[i] = mul_acc(ps.v[e], v, acc[i]);
                        acc[MM + i] = mul_acc(py.v[e], v, acc[MM + i]);
                    }
            }
        }
    });
    block_reduce_store<2 * MM>(acc, 0u, parts, slot0);
}

template <class T, int MM>
__device__ __forceinline__ void compact_coefs(const CompactCoef<MM>& C, T (&u1)[MM], T (&u2h)[MM]) {
#pragma unroll
    for (int i = 0; i < MM; ++i) { u1[i] = (T)C.u1[i]; u2h[i] = (T)C.u2h[i]; }
}

// d for one pack:  d = H0 v + sum u1_i s_i + sum (H0 u2_i) y_i ,  v = -res
template <class T, int MM>
__device__ __forceinline__ void compact_d(int m, T H0, const T (&u1)[MM],
                                          const T (&u2h)[MM], const Pack<T>& pres, const Pack<T> (&ps)[MM],
                                          const Pack<T> (&py)[MM], Pack<T>& d) {
#pragma unroll
    for (int e = 0; e < PackN<T>::N; ++e) {
        T a = H0 * (T(-1) * pres.v[e]);
#pragma unroll
        for (int i = 0; i < MM; ++i)
            if (i < m) a = mul_add(u1[i], ps[i].v[e], a);
#pragma unroll
        for (int i = 0; i < MM; ++i)
            if (i < m) a = mul_add(u2h[i], py[i].v[e], a);
        d.v[e] = a;
    }
}

// x_d = x + H(-res) in compact form (generic path: stencil / dense / fuse = 0)
//   FULL: the memory holds MM pairs (compile-time trip counts: the 2 MM + 2 loads issue back to back) ; NT: the history
//   streams bypass the caches (they are read once per pass and exceed the Infinity Cache at the sizes that matter)
template <class T, int MM, bool FULL = false, bool NT = false>
__global__ void __launch_bounds__(BLOCK)
k_compact_xd(CompactVecs<T, MM> V, CompactCoef<MM> C, const T* __restrict__ res,
             const T* __restrict__ x, T* __restrict__ x_d, int64_t n) {
    T u1[MM], u2h[MM];
    compact_coefs<T, MM>(C, u1, u2h);
    const T H0 = (T)C.H0;
    const int m = FULL ? MM : V.m;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pres = ld(res, i0, cnt), px = ld(x, i0, cnt), ps[MM], py[MM], d, o;
#pragma unroll
        for (int i = 0; i < MM; ++i)
            if (i < m) { ps[i] = ldp<T, NT>(V.S[i], i0, cnt); py[i] = ldp<T, NT>(V.Y[i], i0, cnt); }
        compact_d<T, MM>(m, H0, u1, u2h, pres, ps, py, d);
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) o.v[e] = px.v[e] + d.v[e];
        st(x_d, i0, cnt, o);
    });
}

// Gram products of a (new) pair with the stored ones: <s_i, y_new> -> slot0 + i, <y_i, y_new> -> slot0 + MM + i
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_gram_pair(CompactVecs<T, MM> V, const T* __restrict__ y_new, int64_t n, double* __restrict__ parts,
            int slot0) {
    double acc[2 * MM];
#pragma unroll
    for (int k = 0; k < 2 * MM; ++k) acc[k] = 0.0;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pn = ld(y_new, i0, cnt);
#pragma unroll
        for (int i = 0; i < MM; ++i) {
            if (i < V.m) {
                Pack<T> ps = ld(V.S[i], i0, cnt), py = ld(V.Y[i], i0, cnt);
#pragma unroll
                for (int e = 0; e < PackN<T>::N; ++e)
                    if (e < cnt) {
                        acc[i] = mul_acc(ps.v[e], pn.v[e], acc[i]);
                        acc[MM + i] = mul_acc(py.v[e], pn.v[e], acc[MM + i]);
                    }
            }
        }
    });
    block_reduce_store<2 * MM>(acc, 0u, parts, slot0);
}

// Affine images.  When c is affine (c(x) = A x - b), D is a subspace (ZeroSet / FreeSet: yupd = (c(x) + mu*y [- 0])/mu
// is affine in x) and f is at most quadratic with a diagonal Hessian, BOTH c(x) and grad L(x) are affine maps of x.
// The trial point x_d = x + d is an affine combination of points whose images are known — d = H0 (z - x) + sum u1_i s_i
// + sum u2h_i y_i with s_i, y_i differences of earlier iterates / residuals — so its images are the same combination of
// the stored images: no pass over A.  (demo/basispursuit.jl:38-49)
//   out = base + H0 (zimg - base) + sum u1_i S[i] + sum u2h_i Y[i]       (compact_d's order of operations)
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_affine_image(CompactVecs<T, MM> V, CompactCoef<MM> C, const T* __restrict__ base, const T* __restrict__ zimg,
               T* __restrict__ out, int64_t len) {
    T u1[MM], u2h[MM];
    compact_coefs<T, MM>(C, u1, u2h);
    const T H0 = (T)C.H0;
    bz_for_chunks<T>(len, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        Pack<T> pb = ld(base, i0, cnt), pz = ld(zimg, i0, cnt), ps[MM], py[MM], o;
#pragma unroll
        for (int i = 0; i < MM; ++i)
            if (i < V.m) { ps[i] = ld(V.S[i], i0, cnt); py[i] = ld(V.Y[i], i0, cnt); }
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T a = H0 * (pz.v[e] - pb.v[e]);
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < V.m) a = mul_add(u1[i], ps[i].v[e], a);
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < V.m) a = mul_add(u2h[i], py[i].v[e], a);
            o.v[e] = pb.v[e] + a;
        }
        st(out, i0, cnt, o);
    });
}

// images of the candidate pair: s_img = xnew - xold ; y_img = (xnew - znew) - (xold - zold)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_image_pair(const T* __restrict__ xnew, const T* __restrict__ xold, const T* __restrict__ znew,
             const T* __restrict__ zold, T* __restrict__ s_img, T* __restrict__ y_img, int64_t len) {
    bz_for_chunks<T>(len, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        Pack<T> a = ld(xnew, i0, cnt), b = ld(xold, i0, cnt), c = ld(znew, i0, cnt), d = ld(zold, i0, cnt), s, y;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            s.v[e] = a.v[e] - b.v[e];
            const T rn = a.v[e] - c.v[e], ro = b.v[e] - d.v[e];
            y.v[e] = rn - ro;
        }
        st(s_img, i0, cnt, s);
        st(y_img, i0, cnt, y);
    });
}

// ---------------------------------------------------------------------------
// cfg 4 with affine images (dense c, one pass over A per iteration): the short element-wise kernels either side of
// k_dense_fused as TWO launches.  Each of them is 3-10 us of which most is launch latency, and the ones in front of the
// pass over A are issued right after the host's read-back, when the queue is empty: eleven launches cost the iteration
// ~100 us next to the 360 us pass.  The sections below are those kernels' bodies, operation for operation, on the block ->
// chunk map each has as a launch of its own (bz_for_chunks_v), with the values that one kernel wrote and the next one read
// at the same index kept in registers: same bits in every vector and every partial sum (test_gpu_dense.py).
//   k_dense_head:  blocks [0, gn): k_compact_xd -> k_affine_image (grad L) -> k_fvalue_elem -> k_fbstep   (x-space)
//                  blocks [gn, gn + gy): k_affine_image (c) -> k_yupd                                       (constraint space)
//   k_dense_tail:  blocks [0, gn): k_gemv_t_finish -> k_update_c -> k_image_pair (grad L)
//                  blocks [gn, gn + gy): k_image_pair (c)
// ---------------------------------------------------------------------------
template <class T, int MM> struct DenseHeadArgs {
    CompactVecs<T, MM> V, VG, VA;        // the L-BFGS pairs ; their images under grad L ; under c
    CompactCoef<MM> C;
    const T *res, *x;                    // k_compact_xd
    T* x_d;
    const T *gbase, *gzimg;              // grad L(x), grad L(z): the image of x_d under grad L goes to gout
    T* gout;
    const T *cbase, *czimg;              // c(x), c(z) -> cout ; yupd of it -> yupd
    T *cout, *yupd;
    T *z, *res_new;                      // k_fbstep at (x_d, gout)
    T gamma;
    int64_t n, ny;
    double* parts;
    int slot_f, slot_pen, slot_fb, gn, gy;
};
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_dense_head(DenseHeadArgs<T, MM> A, ElemParams<T> P) {
    T u1[MM], u2h[MM];
    compact_coefs<T, MM>(A.C, u1, u2h);
    const T H0 = (T)A.C.H0;
    if ((int)blockIdx.x < A.gn) {
        const int vb = blockIdx.x;
        const T gl = A.gamma * P.g_lambda;
        double accf[1] = {0.0}, acc[3] = {0.0, 0.0, 0.0};
        bz_for_chunks_v<T>(A.n, vb, A.gn, [&](const int64_t i0, const auto cnt_) {
            const int cnt = cnt_;
            // x_d = x + H(-res)   (k_compact_xd)
            Pack<T> pres = ld(A.res, i0, cnt), px = ld(A.x, i0, cnt), ps[MM], py[MM], d, xd;
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < A.V.m) { ps[i] = ld(A.V.S[i], i0, cnt); py[i] = ld(A.V.Y[i], i0, cnt); }
            compact_d<T, MM>(A.V.m, H0, u1, u2h, pres, ps, py, d);
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) xd.v[e] = px.v[e] + d.v[e];
            st(A.x_d, i0, cnt, xd);
            // grad L(x_d) as the image of x_d   (k_affine_image)
            Pack<T> pb = ld(A.gbase, i0, cnt), pz = ld(A.gzimg, i0, cnt), g;
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < A.VG.m) { ps[i] = ld(A.VG.S[i], i0, cnt); py[i] = ld(A.VG.Y[i], i0, cnt); }
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                T a = H0 * (pz.v[e] - pb.v[e]);
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < A.VG.m) a = mul_add(u1[i], ps[i].v[e], a);
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < A.VG.m) a = mul_add(u2h[i], py[i].v[e], a);
                g.v[e] = pb.v[e] + a;
            }
            st(A.gout, i0, cnt, g);
            // f(x_d)   (k_fvalue_elem, element-wise f)
            if (P.f_kind == BZ_F_DIAG_QUADRATIC) {
                Pack<T> q = ld(P.q, i0, cnt), b = ld(P.b, i0, cnt);
#pragma unroll
                for (int e = 0; e < PackN<T>::N; ++e)
                    if (e < cnt) {
                        T qx = q.v[e] * xd.v[e];
                        accf[0] += (double)(xd.v[e] * (T(0.5) * qx - b.v[e]));
                    }
            }
            // z = prox_{gamma g}(x_d - gamma grad L(x_d)), res = x_d - z   (k_fbstep)
            ElemLoads<T> L;
            load_params(P, i0, cnt, L, false, false, true);
            Pack<T> zp, rp;
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                T t = A.gamma * g.v[e];
                T y = xd.v[e] - t;
                T gterm;
                T zz = prox_elem<T, false>(P.g_kind, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm, P.g_p);
                T r = xd.v[e] - zz;
                zp.v[e] = zz; rp.v[e] = r;
                if (e < cnt) {
                    acc[0] += (double)gterm;
                    acc[1] += (double)(g.v[e] * r);
                    acc[2] += (double)(r * r);
                }
            }
            st(A.z, i0, cnt, zp);
            st(A.res_new, i0, cnt, rp);
        });
        block_reduce_store<1>(accf, 0u, A.parts, A.slot_f, vb);
        block_reduce_store<3>(acc, 0u, A.parts, A.slot_fb, vb);
    } else {
        const int vb = (int)blockIdx.x - A.gn;
        double acc[1] = {0.0};
        bz_for_chunks_v<T>(A.ny, vb, A.gy, [&](const int64_t i0, const auto cnt_) {
            const int cnt = cnt_;
            // c(x_d) as the image of x_d   (k_affine_image)
            Pack<T> pb = ld(A.cbase, i0, cnt), pz = ld(A.czimg, i0, cnt), ps[MM], py[MM], c, yu;
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < A.VA.m) { ps[i] = ld(A.VA.S[i], i0, cnt); py[i] = ld(A.VA.Y[i], i0, cnt); }
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                T a = H0 * (pz.v[e] - pb.v[e]);
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < A.VA.m) a = mul_add(u1[i], ps[i].v[e], a);
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < A.VA.m) a = mul_add(u2h[i], py[i].v[e], a);
                c.v[e] = pb.v[e] + a;
            }
            st(A.cout, i0, cnt, c);
            // the penalty term and yupd at it   (k_yupd)
            ElemLoads<T> L;
            load_params(P, i0, cnt, L, false, true, false);
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                T t = c.v[e] + L.muy.v[e];
                T sv = proj_D(P.D_kind, t, L.dlo.v[e], L.dhi.v[e]);
                t = t - sv;
                T pterm = (t * t) / L.mu.v[e];
                yu.v[e] = t / L.mu.v[e];
                if (e < cnt) acc[0] += (double)pterm;
            }
            st(A.yupd, i0, cnt, yu);
        });
        block_reduce_store<1>(acc, 0u, A.parts, A.slot_pen, vb);
    }
}

template <class T, int MM> struct DenseTailArgs {
    CompactVecs<T, MM> V;                // the stored pairs (k_update_c's Gram products)
    const T* part;                       // k_gemv_t_finish: row-group partials of A'yhat, nchunks of pstride
    int nchunks;
    int64_t pstride;
    const T *z;                          // the point of this gradient (f terms)
    T* gz;                               // grad L(z) out
    const T *x, *x_prev, *res, *res_prev, *gx;      // k_update_c
    T gamma;
    T *s_new, *y_new;
    const T *gx_prev, *gz_prev;          // k_image_pair (grad L): s_img = gx - gx_prev ; y_img = (gx - gz) - (gx_prev - gz_prev)
    T *gs_img, *gy_img;
    const T *cx, *cx_prev, *cz, *cz_prev;           // k_image_pair (c)
    T *cs_img, *cy_img;
    int64_t n, ny;
    double* parts;
    int slot_fz, slot_upd, gn, gy;
};
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_dense_tail(DenseTailArgs<T, MM> A, ElemParams<T> P) {
    if ((int)blockIdx.x < A.gn) {
        const int vb = blockIdx.x;
        constexpr int NS = 5 + 4 * MM + 2;
        double accf[1] = {0.0}, acc[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0.0;
        const int m = A.V.m;
        bz_for_chunks_v<T>(A.n, vb, A.gn, [&](const int64_t i0, const auto cnt_) {
            const int cnt = cnt_;
            // grad L(z) = df(z) + the row-group partials in their fixed order   (k_gemv_t_finish)
            Pack<T> j = ld(A.part, i0, cnt);
            int k = 1;
            for (; k + 7 < A.nchunks; k += 8) {
                Pack<T> q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = ld(A.part + (int64_t)(k + u) * A.pstride, i0, cnt);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < PackN<T>::N; ++e) j.v[e] = j.v[e] + q[u].v[e];
            }
            for (; k < A.nchunks; ++k) {
                Pack<T> q = ld(A.part + (int64_t)k * A.pstride, i0, cnt);
#pragma unroll
                for (int e = 0; e < PackN<T>::N; ++e) j.v[e] = j.v[e] + q.v[e];
            }
            Pack<T> pzz = ld(A.z, i0, cnt), pq = splat(T(0)), pbb = splat(T(0)), pgz;
            if (P.f_kind == BZ_F_DIAG_QUADRATIC) { pq = ld(P.q, i0, cnt); pbb = ld(P.b, i0, cnt); }
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                T dfx = T(0), fterm = T(0);
                if (P.f_kind == BZ_F_DIAG_QUADRATIC) {
                    T qx = pq.v[e] * pzz.v[e];
                    dfx = qx - pbb.v[e];
                    fterm = pzz.v[e] * (T(0.5) * qx - pbb.v[e]);
                }
                pgz.v[e] = dfx + j.v[e];
                if (e < cnt) accf[0] += (double)fterm;
            }
            st(A.gz, i0, cnt, pgz);
            // the pair, the stop norm and the compact form's products   (k_update_c)
            Pack<T> px = ld(A.x, i0, cnt), pxp = ld(A.x_prev, i0, cnt), pr = ld(A.res, i0, cnt), prp = ld(A.res_prev, i0, cnt);
            Pack<T> pgx = ld(A.gx, i0, cnt), ps, py, hs[MM], hy[MM];
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < m) { hs[i] = ld(A.V.S[i], i0, cnt); hy[i] = ld(A.V.Y[i], i0, cnt); }
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                T sv = px.v[e] - pxp.v[e];
                T yv = pr.v[e] - prp.v[e];
                ps.v[e] = sv; py.v[e] = yv;
                T w = pr.v[e] / A.gamma;
                w = w - pgx.v[e];
                w = w + pgz.v[e];
                if (e < cnt) {
                    acc[0] += (double)(sv * yv);
                    acc[1] += (double)(yv * yv);
                    acc[2] = nanmax(acc[2], (double)(w < T(0) ? -w : w));
                    const T nr = T(-1) * pr.v[e];
#pragma unroll
                    for (int i = 0; i < MM; ++i)
                        if (i < m) {
                            acc[3 + i] = mul_acc(hs[i].v[e], yv, acc[3 + i]);
                            acc[3 + MM + i] = mul_acc(hy[i].v[e], yv, acc[3 + MM + i]);
                            acc[3 + 2 * MM + i] = mul_acc(hs[i].v[e], nr, acc[3 + 2 * MM + i]);
                            acc[3 + 3 * MM + i] = mul_acc(hy[i].v[e], nr, acc[3 + 3 * MM + i]);
                        }
                    acc[3 + 4 * MM] = mul_acc(sv, nr, acc[3 + 4 * MM]);
                    acc[3 + 4 * MM + 1] = mul_acc(yv, nr, acc[3 + 4 * MM + 1]);
                }
            }
            st(A.s_new, i0, cnt, ps);
            st(A.y_new, i0, cnt, py);
            // the images of the candidate pair under grad L   (k_image_pair)
            Pack<T> b = ld(A.gx_prev, i0, cnt), dd = ld(A.gz_prev, i0, cnt), si, yi;
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                si.v[e] = pgx.v[e] - b.v[e];
                const T rn = pgx.v[e] - pgz.v[e], ro = b.v[e] - dd.v[e];
                yi.v[e] = rn - ro;
            }
            st(A.gs_img, i0, cnt, si);
            st(A.gy_img, i0, cnt, yi);
        });
        block_reduce_store<1>(accf, 0u, A.parts, A.slot_fz, vb);
        block_reduce_store<NS>(acc, 4u, A.parts, A.slot_upd, vb);
    } else {
        const int vb = (int)blockIdx.x - A.gn;
        // the images of the candidate pair under c   (k_image_pair)
        bz_for_chunks_v<T>(A.ny, vb, A.gy, [&](const int64_t i0, const auto cnt_) {
            const int cnt = cnt_;
            Pack<T> a = ld(A.cx, i0, cnt), b = ld(A.cx_prev, i0, cnt), c = ld(A.cz, i0, cnt), d = ld(A.cz_prev, i0, cnt), s, y;
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                s.v[e] = a.v[e] - b.v[e];
                const T rn = a.v[e] - c.v[e], ro = b.v[e] - d.v[e];
                y.v[e] = rn - ro;
            }
            st(A.cs_img, i0, cnt, s);
            st(A.cy_img, i0, cnt, y);
        });
    }
}

// history as iterates -> history as pairs: S[i] = XH[i+1] - XH[i], Y[i] = RH[i+1] - RH[i] for the MM stored pairs
// (run when an iteration leaves the plain path and the classic kernels need the difference vectors)
template <class T, int MM> struct SnapVecs {
    const T* XH[MM + 1];
    const T* RH[MM + 1];
    T* S[MM];
    T* Y[MM];
    double gam[MM + 1];      // k_pairs_from_iterates: gamma of each iterate's residual (the last one: the current gamma)
};
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK) k_pairs_from_snapshots(SnapVecs<T, MM> V, int64_t n) {
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        Pack<T> xh[MM + 1], rh[MM + 1];
#pragma unroll
        for (int i = 0; i <= MM; ++i) { xh[i] = ld(V.XH[i], i0, cnt); rh[i] = ld(V.RH[i], i0, cnt); }
#pragma unroll
        for (int i = 0; i < MM; ++i) {
            Pack<T> s, y;
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) { s.v[e] = xh[i + 1].v[e] - xh[i].v[e]; y.v[e] = rh[i + 1].v[e] - rh[i].v[e]; }
            st(V.S[i], i0, cnt, s);
            st(V.Y[i], i0, cnt, y);
        }
    });
}

// fixed-point residual of one element at a stored iterate:  z = prox_{gamma g}(x - gamma grad L(x)), res = x - z —
// operation for operation what the fused passes (and k_algrad_elem + k_fbstep) do at a trial point, so
// re-evaluating it at an iterate the rings still hold gives back the bits of the residual computed then
// (xpart: the pair partner's iterate value, read by the pairwise D kinds only)
template <class T>
__device__ __forceinline__ T resid_elem(int fk, int dk, int gk, T xv, const ElemLoads<T>& L, int e, T gamma, T gl,
                                        T& zz, bool udiv = false, T rmu = T(0), T xpart = T(0)) {
    ALOut<T> o = al_elem(fk, dk, xv, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], L.dlo.v[e], L.dhi.v[e],
                         xpart + L.muy.v[e ^ 1], e & 1, udiv, rmu);
    T t = gamma * o.grad;
    T y = xv - t;
    T gterm;
    zz = prox_elem(gk, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm);
    return xv - zz;
}

// history as iterates, residuals not stored -> history as pairs (any element-wise family, run-time kinds: this pass
// is rare): S[i] = XH[i+1] - XH[i], Y[i] = r(XH[i+1]) - r(XH[i]) with r re-evaluated, plus the residual and z of the
// newest iterate — everything the classic kernels need when an iteration leaves the plain path
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_pairs_from_iterates(SnapVecs<T, MM> V, int m, ElemParams<T> P, T* __restrict__ res_cur,
                      T* __restrict__ z_cur, int64_t n) {
    // m <= MM stored pairs: XH[0..m] are the iterates (XH[m] the newest, repeated in the entries beyond it)
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, true, true, true);
        Pack<T> xh[MM + 1], rh[MM + 1], pz;
#pragma unroll
        for (int i = 0; i <= MM; ++i) {
            xh[i] = ld(V.XH[i], i0, cnt);
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) {
                T zz;
                const T gi = (T)V.gam[i];
                rh[i].v[e] = resid_elem<T>(P.f_kind, P.D_kind, P.g_kind, xh[i].v[e], L, e, gi, gi * P.g_lambda, zz, false,
                                           T(0), xh[i].v[e ^ 1]);
                if (i == MM) pz.v[e] = zz;
            }
        }
#pragma unroll
        for (int i = 0; i < MM; ++i) {
            Pack<T> sp, yp;
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) { sp.v[e] = xh[i + 1].v[e] - xh[i].v[e]; yp.v[e] = rh[i + 1].v[e] - rh[i].v[e]; }
            if (i < m) {
                st(V.S[i], i0, cnt, sp);
                st(V.Y[i], i0, cnt, yp);
            }
        }
        st(res_cur, i0, cnt, rh[MM]);
        st(z_cur, i0, cnt, pz);
    });
}

// k_update with the compact form's products in the same pass (generic kernel chain: dense c, generic oracles, rejected
// trials): besides <s,y>, <y,y> and the stop norm, the Gram products of the candidate pair with the stored pairs and
// the products of all pairs with the new residual — the slot layout of k_fused_compact from slot0 + 7 on — so the chain
// needs ONE read-back per trial instead of three (k_update, k_gram_pair, k_gram_dots).  Same arithmetic as those
// kernels (mul_acc), same summation tree on the same grid.
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_update_c(CompactVecs<T, MM> V, const T* __restrict__ x, const T* __restrict__ x_prev, const T* __restrict__ res,
           const T* __restrict__ res_prev, const T* __restrict__ gx, const T* __restrict__ gz, T gamma,
           T* __restrict__ s_new, T* __restrict__ y_new, int64_t n, double* __restrict__ parts, int slot0) {
    constexpr int NS = 5 + 4 * MM + 2;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    const int m = V.m;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        Pack<T> px = ld(x, i0, cnt), pxp = ld(x_prev, i0, cnt), pr = ld(res, i0, cnt), prp = ld(res_prev, i0, cnt);
        Pack<T> pgx = ld(gx, i0, cnt), pgz = ld(gz, i0, cnt), ps, py, hs[MM], hy[MM];
#pragma unroll
        for (int i = 0; i < MM; ++i)
            if (i < m) { hs[i] = ld(V.S[i], i0, cnt); hy[i] = ld(V.Y[i], i0, cnt); }
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T sv = px.v[e] - pxp.v[e];
            T yv = pr.v[e] - prp.v[e];
            ps.v[e] = sv; py.v[e] = yv;
            T w = pr.v[e] / gamma;
            w = w - pgx.v[e];
            w = w + pgz.v[e];
            if (e < cnt) {
                acc[0] += (double)(sv * yv);
                acc[1] += (double)(yv * yv);
                acc[2] = nanmax(acc[2], (double)(w < T(0) ? -w : w));
                const T nr = T(-1) * pr.v[e];
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < m) {
                        acc[3 + i] = mul_acc(hs[i].v[e], yv, acc[3 + i]);
                        acc[3 + MM + i] = mul_acc(hy[i].v[e], yv, acc[3 + MM + i]);
                        acc[3 + 2 * MM + i] = mul_acc(hs[i].v[e], nr, acc[3 + 2 * MM + i]);
                        acc[3 + 3 * MM + i] = mul_acc(hy[i].v[e], nr, acc[3 + 3 * MM + i]);
                    }
                acc[3 + 4 * MM] = mul_acc(sv, nr, acc[3 + 4 * MM]);
                acc[3 + 4 * MM + 1] = mul_acc(yv, nr, acc[3 + 4 * MM + 1]);
            }
        }
        st(s_new, i0, cnt, ps);
        st(y_new, i0, cnt, py);
    });
    block_reduce_store<NS>(acc, 4u, parts, slot0 + 7);
}

// cfg-3 fast path with the compact L-BFGS representation: k_stencil_update plus, in the same pass, everything the
// next application of the operator needs — the Gram products of the new pair with the stored pairs and the products
// of all pairs with the new residual (k_fused_compact's slot layout, so the host reads one block of scalars):
//   slot0 + 5,6: f terms, t^2/mu at z ; + 7 <s,y>, + 8 <y,y>, + 9 max stop
//   slot0 + 10 + i: <s_i, y_new> ; + 10 + MM + i: <y_i, y_new> ; + 10 + 2MM + i: <s_i, -res> ; + 10 + 3MM + i: <y_i, -res> ;
//   then <s_new, -res>, <y_new, -res>
// (slots slot0 + 0..4 are k_stencil_fb's).  With it the iteration has ONE reduction phase and no persistent
// two-loop kernel with its 2m-1 grid barriers: x_d (k_compact_xd), k_stencil_fb, this.
// REGX (r03): grad L(x_d) and res are not read but re-formed here — res = x_d - z from the two packs this pass loads anyway, and
// grad L(x_d) by the stencil on x_d (its north / south / west / east re-reads are cache hits: the pass streams x_d already) —
// the operations of k_stencil_fb on the same operands, so the same bits, for two read streams less here and one write stream
// less there (k_stencil_fb with grad = null): 39 -> 36 passes over n per iteration.  halo_x: x_d's halo rows (sharded grid).
// (measured on cfg 3: REGX = 2, both re-formed: 126 us against 117 us for this pass — the second stencil costs more than the two
// streams it saves ; REGX = 1, res only — a subtraction of two packs the pass holds anyway: the default)
template <class T, int MM, bool FULL = false, bool NT = false, int REGX = 0>
__global__ void __launch_bounds__(BLOCK)
k_stencil_update_c(CompactVecs<T, MM> V, const T* __restrict__ zp, ElemParams<T> P, int64_t nx, int64_t ny,
                   const T* __restrict__ x, const T* __restrict__ x_prev, const T* __restrict__ res,
                   const T* __restrict__ res_prev, const T* __restrict__ gx, T gamma,
                   T* __restrict__ s_new, T* __restrict__ y_new, int64_t n,
                   double* __restrict__ parts, int slot0, StencilHalo<T> halo, StencilHalo<T> halo_x = StencilHalo<T>()) {
    constexpr int NS = 5 + 4 * MM + 2;
    double accF[2] = {0.0, 0.0}, acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    const int m = FULL ? MM : V.m;
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        Pack<T> zc = ld(zp, i0, cnt);
        Pack<T> pgz = stencil_al_pack<T, NT>(zp, P, nx, ny, 0, i0, cnt, zc, accF[0], accF[1], halo);
        Pack<T> px = ld(x, i0, cnt), pxp = ldp<T, NT>(x_prev, i0, cnt), pr, prp = ldp<T, NT>(res_prev, i0, cnt);
        Pack<T> pgx, ps, py, hs[MM], hy[MM];
        if constexpr (REGX >= 1) {
#pragma unroll
            for (int e = 0; e < PackN<T>::N; ++e) pr.v[e] = px.v[e] - zc.v[e];
        } else {
            pr = ld(res, i0, cnt);
        }
        if constexpr (REGX >= 2) {
            double dump0 = 0.0, dump1 = 0.0;      // (the value terms at x_d were summed by k_stencil_fb)
            pgx = stencil_al_pack<T, NT>(x, P, nx, ny, 0, i0, cnt, px, dump0, dump1, halo_x);
        } else {
            pgx = ld(gx, i0, cnt);
        }
#pragma unroll
        for (int i = 0; i < MM; ++i)
            if (i < m) { hs[i] = ldp<T, NT>(V.S[i], i0, cnt); hy[i] = ldp<T, NT>(V.Y[i], i0, cnt); }
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T sv = px.v[e] - pxp.v[e];
            T yv = pr.v[e] - prp.v[e];
            ps.v[e] = sv; py.v[e] = yv;
            T w = pr.v[e] / gamma;
            w = w - pgx.v[e];
            w = w + pgz.v[e];
            if (e < cnt) {
                acc[0] += (double)(sv * yv);
                acc[1] += (double)(yv * yv);
                acc[2] = nanmax(acc[2], (double)(w < T(0) ? -w : w));
                const T nr = T(-1) * pr.v[e];
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < m) {
                        acc[3 + i] = mul_acc(hs[i].v[e], yv, acc[3 + i]);
                        acc[3 + MM + i] = mul_acc(hy[i].v[e], yv, acc[3 + MM + i]);
                        acc[3 + 2 * MM + i] = mul_acc(hs[i].v[e], nr, acc[3 + 2 * MM + i]);
                        acc[3 + 3 * MM + i] = mul_acc(hy[i].v[e], nr, acc[3 + 3 * MM + i]);
                    }
                acc[3 + 4 * MM] = mul_acc(sv, nr, acc[3 + 4 * MM]);
                acc[3 + 4 * MM + 1] = mul_acc(yv, nr, acc[3 + 4 * MM + 1]);
            }
        }
        st(s_new, i0, cnt, ps);
        st(y_new, i0, cnt, py);
    });
    block_reduce_store<2>(accF, 0u, parts, slot0 + 5);
    __syncthreads();
    block_reduce_store<NS>(acc, 4u, parts, slot0 + 7);
}

// The separable fast path with the compact direction: ONE pass computes d from (res, S, Y), then x_d, both AL
// gradients, the FB step, the new pair, its Gram products with the stored pairs and the stop norm.
//   reads : res, S[m], Y[m], x, q, b, mu, mu*y   writes: x_d, z, res, s_new, y_new
//   slots : slot0 + 0..9 as k_fused_sep ; + 10 + i: <s_i, y_new> ; + 10 + MM + i: <y_i, y_new> ;
//           + 10 + 2MM + i: <s_i, -res> ; + 10 + 3MM + i: <y_i, -res> ; then <s_new, -res>, <y_new, -res>
//           with res the NEW residual: the p and w of the next application, whichever pairs it keeps —
//           so the whole iteration is this one pass (S and Y are in registers here anyway)
//   XR = 1: V.S / V.Y are the last MM iterates / residuals before (x, res_prev) and the pairs are re-formed
//           in registers (s_new, y_new not written)
//   XR = 2: as 1, and the residuals are not read either but re-evaluated from the iterates (resid_elem):
//           reads the MM+1 iterates, q, b, mu, mu*y ; writes x_d only (res too if `res` is not null)
//   UNI = 1: every mu[i] is the same number (P.mu_uniform; alps.jl:42 from a start with c(x0) in D, and
//            alps.jl:97 scales all of them alike) -> not streamed ; UNI = 2: and mu*y = 0 (first subproblem from
//            y0 = 0) -> not streamed either.  Same operands, same operations (the + 0 stays), so the same bits.
//   TRIAL (XR = 2 only): the trial point is GIVEN in x_d (a tau-blend of the rejected x + d and z: read, not
//            written) instead of formed as x + d: a backtracked trial is then this one pass too — both
//            gradients, FB step, pair, its Gram products, p, w, stop norm — with nothing materialised.
// Oracle family of the iterate-history form (XR = 2), fixed at compile time: the kinds of f, g and D and which of
// their parameters are vectors (= streams the pass must prefetch).  Every element-wise family the solver lowers
// has its own instantiation of the one-pass kernel (bz_families_*.hip); the headline family (cfg 2 / cfg 5)
// additionally fixes UNI and TRIAL at compile time.
constexpr int FAM_F_ZERO = 0, FAM_F_DIAG = 1;
constexpr int FAM_G_ZERO = 0, FAM_G_L1 = 1, FAM_G_L1NONNEG = 2, FAM_G_L1BOX = 3, FAM_G_INDBOX = 4, FAM_G_INDBOX_VEC = 5;
constexpr int FAM_G_COUNT = 6;
constexpr int FAM_D_ZERO = 0, FAM_D_FREE = 1, FAM_D_BOX = 2, FAM_D_BOX_VEC = 3;
constexpr int FAM_D_VC = 4, FAM_D_CC = 5, FAM_D_EITHEROR = 6, FAM_D_XOR = 7;      // BZ_D_VC_PAIRS + (k - 4): one kind per
constexpr int FAM_D_COUNT = 8;      // instantiation (all four in one kernel is 13 000 instructions: past the instruction cache)
constexpr int fam_code(int fk, int gk, int dk) { return fk | (gk << 1) | (dk << 4); }
constexpr int fam_fk(int fam) { return fam & 1; }
constexpr int fam_gk(int fam) { return (fam >> 1) & 7; }
constexpr int fam_dk(int fam) { return fam >> 4; }
constexpr int FAM_HEADLINE = fam_code(FAM_F_DIAG, FAM_G_L1, FAM_D_BOX);

//   UNI / TRIAL = -1: taken at run time from C.uni_rt / C.trial_rt (the family instantiations: a wave-uniform branch
//   around two loads and one store costs nothing next to a third of the instantiations)
template <class T, int MM, bool NT, bool SPEC, bool OFF32 = false, int XR = 0, int UNI = 0, int TRIAL = 0,
          int FAM = FAM_HEADLINE>
__global__ void __launch_bounds__(BLOCK)
k_fused_compact(CompactVecs<T, MM> V, CompactCoef<MM> C, const T* __restrict__ x,
                const T* __restrict__ res_prev, ElemParams<T> P, T gamma, T* __restrict__ x_d,
                T* __restrict__ z, T* __restrict__ res, T* __restrict__ s_new, T* __restrict__ y_new,
                int64_t n, double* __restrict__ parts, int slot0) {
    // SPEC (stored-pair forms): the headline family (cfg 2 / cfg 5) with everything uniform known at compile time —
    // f = DiagQuadratic, g = NormL1, D = Box with scalar bounds, full memory: no kind switches, no optional
    // streams, far fewer live scalar registers (the generic body spills ~300 SGPRs to VGPR lanes).
    // XR = 2: the kinds come from FAM.
    constexpr bool FAMILY = XR == 2;
    constexpr int GKC = fam_gk(FAM), DKC = fam_dk(FAM);
    const int fk = FAMILY ? fam_fk(FAM) : (SPEC ? (int)BZ_F_DIAG_QUADRATIC : P.f_kind);
    const int gk = FAMILY ? (GKC >= FAM_G_INDBOX ? (int)BZ_G_IND_BOX : GKC) : (SPEC ? (int)BZ_G_NORM_L1 : P.g_kind);
    const int dk = FAMILY ? (DKC >= FAM_D_VC ? (int)BZ_D_VC_PAIRS + (DKC - FAM_D_VC) : (DKC >= FAM_D_BOX ? (int)BZ_D_BOX : DKC))
                          : (SPEC ? (int)BZ_D_BOX : P.D_kind);
    const int uni = UNI >= 0 ? UNI : C.uni_rt;
    const bool trial = TRIAL >= 0 ? (TRIAL != 0) : (C.trial_rt != 0);
    const int m = SPEC ? MM : V.m;
    T u1[MM], u2h[MM];
    compact_coefs<T, MM>(C, u1, u2h);
    T H0 = (T)C.H0;
    T gl = gamma * P.g_lambda;
    __shared__ unsigned long long gate_sh;
    // the gate of a pre-launched pass (see GateRec); returns false when the host recalled the launch
    auto gate_wait = [&]() -> bool {
        if constexpr (XR == 2) {
        if (C.gate_seq != 0ull) {
            static_assert(MM <= 5, "GateRec holds five coefficients of each kind");
            const unsigned long long tag = (unsigned long long)ll_tag(C.gate_seq) << 32;
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                unsigned long long sq = 0ull;
                unsigned spins = 0;
                if (blockIdx.x == 0) {
                    unsigned long long word = 0ull;
                    for (;;) {
                        if (lane < 32) word = sys_load(&C.gate_host->w[lane]);
                        const bool mine = lane < 32 && (word >> 32 << 32) == tag;
                        const unsigned long long ok = __ballot(mine);
                        if (ok & (1ull << 31)) { sq = C.gate_seq | GATE_ABORT; break; }
                        if ((ok & 0x3FFFFFFull) == 0x3FFFFFFull) { sq = C.gate_seq; break; }
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > C.gate_spin_host) { sq = C.gate_seq | GATE_ABORT; if (lane == 0) *C.gate_timeout = 6; break; }
                    }
                    // even lanes 0, 2, .., 24 assemble value lane/2 from their word and the next lane's
                    const unsigned lo = (unsigned)word, hi = (unsigned)__shfl((unsigned)word, lane + 1, 64);
                    if (!(sq & GATE_ABORT) && lane < 26 && (lane & 1) == 0) {
                        const unsigned long long bits = (unsigned long long)lo | ((unsigned long long)hi << 32);
                        __hip_atomic_store(&C.gate_dev->val[lane >> 1], __longlong_as_double((long long)bits), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&C.gate_dev->seq, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (lane == 0) {
                    for (;;) {
                        sq = __hip_atomic_load(&C.gate_dev->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((sq & ~GATE_ABORT) == C.gate_seq) break;
                        // (other stream: ~0.7 us between polls — 255 pollers must not load the fabric while a pass streams)
                        if (C.gate_other_stream) __builtin_amdgcn_s_sleep(24); else __builtin_amdgcn_s_sleep(2);
                        if (++spins > C.gate_spin_dev) { sq = C.gate_seq | GATE_ABORT; *C.gate_timeout = 7; break; }
                    }
                }
                if (lane == 0) gate_sh = sq;
            }
            __syncthreads();
            if (gate_sh & GATE_ABORT) return false;
            // this kernel may have been resident while the previous pass (another stream) was still writing what it
            // is about to read: every wave takes an agent-scope acquire before its first load
            if (C.gate_other_stream) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
            for (int i = 0; i < MM; ++i) {
                u1[i] = (T)__hip_atomic_load(&C.gate_dev->val[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u2h[i] = (T)__hip_atomic_load(&C.gate_dev->val[5 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            H0 = (T)__hip_atomic_load(&C.gate_dev->val[10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z = (T*)(uintptr_t)__double_as_longlong(__hip_atomic_load(&C.gate_dev->val[11], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        }
        return true;
    };
    // the pipelined form issues the loads of its first packs BEFORE it goes to the gate (they do not depend on the
    // coefficients): the gate wait hides their latency.  Not when the launch may be resident while the previous pass
    // still writes (other stream): its first load must come after the acquire.
    constexpr bool PIPE_ = SPEC && OFF32 && XR == 2;
    const bool gate_late = PIPE_ && !C.gate_other_stream && C.gate_late;
    if (!gate_late) { if (!gate_wait()) return; }
    if (SPEC && !OFF32) {      // keep the per-application coefficients in vector registers: scalar ones are the scarce kind here
#pragma unroll
        for (int i = 0; i < MM; ++i) { asm volatile("" : "+v"(u1[i])); asm volatile("" : "+v"(u2h[i])); }
        asm volatile("" : "+v"(H0));
        asm volatile("" : "+v"(gl));
        asm volatile("" : "+v"(gamma));
    }
    // fp64, iterate-history form: the 10 divisions by mu per element (and the one by gamma) go through div_u,
    // on one reciprocal per launch (uniform penalties) or per element
    constexpr bool UDIV = XR == 2 && sizeof(T) == 8;
    T rmu_u = T(0), rgam = T(0);
    if constexpr (UDIV) { rgam = T(1) / gamma; if (uni >= 1) rmu_u = T(1) / P.mu_uniform; }
    const T gam0 = (T)C.gam0, gl0 = gam0 * P.g_lambda;      // XR = 2: gamma (and gamma*lambda) of the oldest iterate's residual
    constexpr int NS = 10 + 4 * MM + 2;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    // XR = 2 reads few enough streams (MM+1 iterates + the family's parameter vectors) to keep the NEXT packs' loads in
    // flight while this pack's ~500-800 instructions run: a software pipeline in registers, two packs deep for the
    // families with up to 11 streams, one pack deep beyond (the register file holds three stages of 11 packs)
    constexpr bool PIPE = SPEC && OFF32 && XR == 2;
    constexpr int NSTREAMS = MM + 1 + (fam_fk(FAM) ? 2 : 0) + (UNI >= 2 ? 0 : (UNI == 1 ? 1 : 2)) + (TRIAL == 0 ? 0 : 1) +
                             (GKC == FAM_G_L1BOX ? 1 : 0) + (GKC == FAM_G_INDBOX_VEC ? 2 : 0) + (DKC == FAM_D_BOX_VEC ? 2 : 0);
    constexpr int DEPTH = NSTREAMS <= 11 ? 2 : 1;
    struct Stage { Pack<T> q, b, mu, muy, px, ps[MM], xt, gu, glo, ghi, dlo, dhi; };
    auto load_stage = [&](Stage& S, unsigned bo) {
        asm volatile("" : "+v"(bo));
        if (fk == BZ_F_DIAG_QUADRATIC) { S.q = ldo<T, NT>(P.q, bo); S.b = ldo<T, NT>(P.b, bo); }
        if (uni < 1) S.mu = ldo<T, NT>(P.mu, bo);
        if (uni < 2) S.muy = ldo<T, NT>(P.muy, bo);
        S.px = ldo<T, NT>(x, bo);
        if (trial) S.xt = ldo<T, NT>((const T*)x_d, bo);
#pragma unroll
        for (int i = 0; i < MM; ++i) S.ps[i] = ldo<T, NT>(V.S[i], bo);
        if constexpr (GKC == FAM_G_L1BOX) S.gu = ldo<T, NT>(P.g_u, bo);
        if constexpr (GKC == FAM_G_INDBOX_VEC) {
            S.glo = P.g_lo_vec ? ldo<T, NT>(P.g_lo_vec, bo) : splat(P.g_lo);
            S.ghi = P.g_hi_vec ? ldo<T, NT>(P.g_hi_vec, bo) : splat(P.g_hi);
        }
        if constexpr (DKC == FAM_D_BOX_VEC) {
            S.dlo = P.D_lo_vec ? ldo<T, NT>(P.D_lo_vec, bo) : splat(P.D_lo);
            S.dhi = P.D_hi_vec ? ldo<T, NT>(P.D_hi_vec, bo) : splat(P.D_hi);
        }
    };
    auto body = [&](const int64_t i0, const auto cnt_, const auto staged_, const Stage& SG) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        // full chunks of a vector shorter than 4 GiB: every stream is (scalar base, one shared 32-bit offset)
        constexpr bool O32 = SPEC && OFF32 && !std::is_integral<std::remove_cv_t<decltype(cnt_)>>::value;
        constexpr bool STAGED = std::remove_cv_t<decltype(staged_)>::value;
        constexpr int N = PackN<T>::N;
        unsigned bo = (unsigned)(i0 * (int64_t)sizeof(T));
        if constexpr (O32) asm volatile("" : "+v"(bo));      // opaque: no per-stream 64-bit pointer induction variables
        ElemLoads<T> L;
        Pack<T> px, prp, ps[MM], py[MM], d, rmu, pxt;
        if (trial) {
            if constexpr (STAGED) pxt = SG.xt; else pxt = ldp<T, NT>((const T*)x_d, i0, cnt);
        }
        if constexpr (STAGED) {
            if (fk == BZ_F_DIAG_QUADRATIC) { L.q = SG.q; L.b = SG.b; } else { L.q = splat(T(0)); L.b = splat(T(0)); }
            if (uni >= 1) L.mu = splat(P.mu_uniform); else L.mu = SG.mu;
            if (uni >= 2) L.muy = splat(T(0)); else L.muy = SG.muy;
            if constexpr (DKC == FAM_D_BOX_VEC) { L.dlo = SG.dlo; L.dhi = SG.dhi; } else { L.dlo = splat(P.D_lo); L.dhi = splat(P.D_hi); }
            if constexpr (GKC == FAM_G_L1BOX) L.gu = SG.gu; else L.gu = splat(T(0));
            if constexpr (GKC == FAM_G_INDBOX_VEC) { L.glo = SG.glo; L.ghi = SG.ghi; } else { L.glo = splat(P.g_lo); L.ghi = splat(P.g_hi); }
            px = SG.px;
#pragma unroll
            for (int i = 0; i < MM; ++i) ps[i] = SG.ps[i];
        } else if constexpr (O32 && !FAMILY) {
            L.q = ldo<T, NT>(P.q, bo); L.b = ldo<T, NT>(P.b, bo);
            if (P.uni >= 1) L.mu = splat(P.mu_uniform); else L.mu = ldo<T, NT>(P.mu, bo);
            if (P.uni >= 2) L.muy = splat(T(0)); else L.muy = ldo<T, NT>(P.muy, bo);
            L.dlo = splat(P.D_lo); L.dhi = splat(P.D_hi);
            L.gu = splat(T(0)); L.glo = splat(T(0)); L.ghi = splat(T(0));
            px = ldo<T, NT>(x, bo);
            prp = ldo<T, NT>(res_prev, bo);
#pragma unroll
            for (int i = 0; i < MM; ++i) {
                ps[i] = ldo<T, NT>(V.S[i], bo);
                py[i] = ldo<T, NT>(V.Y[i], bo);
            }
        } else {
            if (SPEC && !FAMILY) {
                L.q = ldp<T, NT>(P.q, i0, cnt); L.b = ldp<T, NT>(P.b, i0, cnt);
                if (P.uni >= 1) L.mu = splat(P.mu_uniform); else L.mu = ldp<T, NT>(P.mu, i0, cnt);
                if (P.uni >= 2) L.muy = splat(T(0)); else L.muy = ldp<T, NT>(P.muy, i0, cnt);
                L.dlo = splat(P.D_lo); L.dhi = splat(P.D_hi);
                L.gu = splat(T(0)); L.glo = splat(T(0)); L.ghi = splat(T(0));
            } else {
                // run-time kinds (generic stored-pair body; the ragged last chunk of a family kernel, whose kinds
                // are those of P by construction)
                load_params<T, NT>(P, i0, cnt, L, true, true, true);
                if (FAMILY && uni >= 1) L.mu = splat(P.mu_uniform);
                if (FAMILY && uni >= 2) L.muy = splat(T(0));
            }
            px = ldp<T, NT>(x, i0, cnt);
            if constexpr (XR != 2) prp = ldp<T, NT>(res_prev, i0, cnt);
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < m) {
                    ps[i] = ldp<T, NT>(V.S[i], i0, cnt);
                    if constexpr (XR != 2) py[i] = ldp<T, NT>(V.Y[i], i0, cnt);
                }
        }
        if constexpr (XR == 2) {
            // ... and the residuals re-evaluated at those iterates instead of read: the same operations on the
            // same inputs as when they were first computed (gamma, mu, mu*y have not changed since: any
            // iteration that changes them leaves this mode), so again the same bits
            Pack<T> rr[MM + 1];
#pragma unroll
            for (int e = 0; e < N; ++e) {
                if constexpr (UDIV) rmu.v[e] = (uni >= 1) ? rmu_u : T(1) / L.mu.v[e];
#pragma unroll
                for (int i = 0; i <= MM; ++i) {
                    T zz;
                    const T gi = (i == 0) ? gam0 : gamma;
                    const T gli = (i == 0) ? gl0 : gl;
                    rr[i].v[e] = resid_elem<T>(fk, dk, gk, (i < MM) ? ps[i].v[e] : px.v[e], L, e, gi, gli, zz, UDIV,
                                               rmu.v[e], (i < MM) ? ps[i].v[e ^ 1] : px.v[e ^ 1]);
                }
            }
            prp = rr[MM];
#pragma unroll
            for (int i = 0; i < MM; ++i)
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const T nx = (i + 1 < MM) ? ps[i + 1].v[e] : px.v[e];
                    ps[i].v[e] = nx - ps[i].v[e];
                    py[i].v[e] = rr[i + 1].v[e] - rr[i].v[e];
                }
        } else if constexpr (XR == 1) {
            // history kept as ITERATES: V.S[i], V.Y[i] are the snapshots x_{k-MM+i}, res_{k-MM+i} (x and res_prev the
            // newest), and the pairs are their successive differences — the very subtractions that produced the
            // stored s and y (s = x_d - x, y = res - res_prev), so the same bits, for two write streams less
#pragma unroll
            for (int i = 0; i < MM; ++i)
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const T nx = (i + 1 < MM) ? ps[i + 1].v[e] : px.v[e];
                    const T nr = (i + 1 < MM) ? py[i + 1].v[e] : prp.v[e];
                    ps[i].v[e] = nx - ps[i].v[e];
                    py[i].v[e] = nr - py[i].v[e];
                }
        }
        if (!trial) compact_d<T, MM>(m, H0, u1, u2h, prp, ps, py, d);
        // the trial point, then gradient + forward-backward step, then the gradient at z: three sweeps over the
        // pack's elements, because the pairwise D kinds read the pair partner's value at each of the three points
        Pack<T> pxd, pz, pr, pss, pyy, pg1, pgt;
#pragma unroll
        for (int e = 0; e < N; ++e) pxd.v[e] = trial ? pxt.v[e] : px.v[e] + d.v[e];
        T f1[N], p1[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const T xd = pxd.v[e];
            ALOut<T> o1 = al_elem(fk, dk, xd, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], L.dlo.v[e], L.dhi.v[e],
                                  pxd.v[e ^ 1] + L.muy.v[e ^ 1], e & 1, UDIV, rmu.v[e]);
            T t = gamma * o1.grad;
            T y = xd - t;
            T zz = prox_elem(gk, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], pgt.v[e]);
            pz.v[e] = zz; pr.v[e] = xd - zz;
            pg1.v[e] = o1.grad; f1[e] = o1.fterm; p1[e] = o1.pterm;
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const T xd = pxd.v[e], zz = pz.v[e], r = pr.v[e];
            ALOut<T> o2 = al_elem(fk, dk, zz, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], L.dlo.v[e], L.dhi.v[e],
                                  pz.v[e ^ 1] + L.muy.v[e ^ 1], e & 1, UDIV, rmu.v[e]);
            T sv = xd - px.v[e];
            T yy = r - prp.v[e];
            T w = UDIV ? div_u(r, gamma, rgam) : r / gamma;
            w = w - pg1.v[e];
            w = w + o2.grad;
            pss.v[e] = sv; pyy.v[e] = yy;
            if (e < cnt) {
                acc[0] += (double)f1[e];
                acc[1] += (double)p1[e];
                acc[2] += (double)pgt.v[e];
                acc[3] += (double)(pg1.v[e] * r);
                acc[4] += (double)(r * r);
                acc[5] += (double)o2.fterm;
                acc[6] += (double)o2.pterm;
                acc[7] += (double)(sv * yy);
                acc[8] += (double)(yy * yy);
                acc[9] = nanmax(acc[9], (double)(w < T(0) ? -w : w));
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < m) {
                        acc[10 + i] = mul_acc(ps[i].v[e], yy, acc[10 + i]);
                        acc[10 + MM + i] = mul_acc(py[i].v[e], yy, acc[10 + MM + i]);
                    }
                const T nr = T(-1) * r;
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < m) {
                        acc[10 + 2 * MM + i] = mul_acc(ps[i].v[e], nr, acc[10 + 2 * MM + i]);
                        acc[10 + 3 * MM + i] = mul_acc(py[i].v[e], nr, acc[10 + 3 * MM + i]);
                    }
                acc[10 + 4 * MM] = mul_acc(sv, nr, acc[10 + 4 * MM]);
                acc[10 + 4 * MM + 1] = mul_acc(yy, nr, acc[10 + 4 * MM + 1]);
            }
        }
        if constexpr (O32) {
            if (!trial) sto<T, NT>(x_d, bo, pxd);
            if (z) sto<T, NT>(z, bo, pz);
            if (XR != 2 || res) sto<T, NT>(res, bo, pr);
            if constexpr (XR == 0) { sto<T, NT>(s_new, bo, pss); sto<T, NT>(y_new, bo, pyy); }
        } else {
            if (!trial) stp<T, NT>(x_d, i0, cnt, pxd);
            if (z) stp<T, NT>(z, i0, cnt, pz);
            if (XR != 2 || res) stp<T, NT>(res, i0, cnt, pr);
            if constexpr (XR == 0) {
                stp<T, NT>(s_new, i0, cnt, pss);
                stp<T, NT>(y_new, i0, cnt, pyy);
            }
        }
    };
    if constexpr (PIPE) {
        // the chunk -> thread map of bz_for_chunks, with the loads of the next chunks issued before chunk c is
        // consumed (past the end a thread re-requests the last full chunk: no branch, nothing out of bounds)
        constexpr int N = PackN<T>::N;
        const int64_t nfull = n / N;
        const int64_t stride = (int64_t)gridDim.x * BLOCK;
        int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        auto clampc = [&](int64_t k) { return k < nfull ? k : (nfull - 1); };
        auto fetch = [&](Stage& S, int64_t k) { load_stage(S, (unsigned)(clampc(k) * N * (int64_t)sizeof(T))); };
        auto use = [&](const Stage& S, int64_t k) { body(k * N, std::integral_constant<int, N>{}, std::true_type{}, S); };
        Stage sa, sb;
        if constexpr (DEPTH == 2) {
            // two packs ahead: with one, a wave has 8..10 KB in flight and the pass is bound by latency x concurrency
            // (three stages used in rotation, the loop unrolled by three: no register copies between iterations)
            Stage sc;
            if (c < nfull) { fetch(sa, c); fetch(sb, c + stride); }
            if (gate_late) { if (!gate_wait()) return; }
            for (;;) {
                if (c >= nfull) break;
                fetch(sc, c + 2 * stride); use(sa, c); c += stride;
                if (c >= nfull) break;
                fetch(sa, c + 2 * stride); use(sb, c); c += stride;
                if (c >= nfull) break;
                fetch(sb, c + 2 * stride); use(sc, c); c += stride;
            }
        } else {
            if (c < nfull) fetch(sa, c);
            if (gate_late) { if (!gate_wait()) return; }
            for (;;) {
                if (c >= nfull) break;
                fetch(sb, c + stride); use(sa, c); c += stride;
                if (c >= nfull) break;
                fetch(sa, c + stride); use(sb, c); c += stride;
            }
        }
        if (c == nfull && nfull * N < n) body(c * N, (int)(n - nfull * N), std::false_type{}, sa);
    } else {
        Stage none;
        bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) { body(i0, cnt_, std::false_type{}, none); });
    }
    if constexpr (NS == 32) block_reduce_store32(acc, 1u << 9, parts, slot0);
    else block_reduce_store<NS>(acc, 1u << 9, parts, slot0);
}

// ---------------------------------------------------------------------------
// The slack (ALS) form of the fused iteration, compact L-BFGS representation with stored pairs: the whole iteration of
// PANOCplus on xs = [x; s] (src/algorithms/als.jl:68-72, src/utilities/auglagfunslack.jl:59-154) in ONE pass.  With
// c = Identity and an element-wise f the lifted problem couples x_i with s_i only, so the chain
//   d -> xs_d = xs + d -> grad F(xs_d) -> [prox_g; proj_D] -> grad F(z) -> pair -> stop norm
// stays in registers per index i, as it does per element in k_fused_compact: reads res, S[m], Y[m], xs (both halves),
// q, b, mu, mu*y, y; writes xs_d, z, res, s_new, y_new — (2m + 7) passes over the lifted vector + 5 over n instead
// of the 4m + 24 of the kernel chain (k_compact_xd, k_algrad_slack_elem x 2, k_fbstep_slack, k_update_c).
// Element arithmetic: that of k_algrad_slack_elem / k_fbstep_slack / k_update_c, operation for operation; the reductions
// run over the index i with both halves' contributions, another summation tree than the chain's (which runs over the
// lifted vector), so the scalars agree with it to rounding.  Slots: those of k_fused_compact.
// ---------------------------------------------------------------------------
template <class T> struct SlackOut { T gx, gs, fterm, pterm; };
// (udiv: mu is uniform over the launch and rmu = RN(1 / mu): the quotients through two Markstein steps, div_u — the bits of
// the hardware division in half the instructions)
template <class T>
__device__ __forceinline__ SlackOut<T> slack_elem(int f_kind, T x, T sv, T q, T b, T mu, T muy, T y, bool udiv = false, T rmu = T(0)) {
    SlackOut<T> o;
    T dfx = T(0);
    o.fterm = T(0);
    if (f_kind == BZ_F_DIAG_QUADRATIC) {
        T qx = q * x;
        dfx = qx - b;
        o.fterm = x * (T(0.5) * qx - b);
    }
    const T cx = x;
    T w = cx + muy;
    w = w - sv;
    o.pterm = udiv ? div_u(w * w, mu, rmu) : (w * w) / mu;
    const T r = cx - sv;
    const T yupd = y + (udiv ? div_u(r, mu, rmu) : r / mu);
    o.gx = dfx + yupd;
    o.gs = -yupd;
    return o;
}
template <class T, int MM, bool NT>
__global__ void __launch_bounds__(BLOCK)
k_fused_slack(CompactVecs<T, MM> V, CompactCoef<MM> C, const T* __restrict__ xs, const T* __restrict__ res_prev,
              ElemParams<T> P, const T* __restrict__ yv, T gamma, T* __restrict__ xs_d, T* __restrict__ z,
              T* __restrict__ res, T* __restrict__ s_new, T* __restrict__ y_new, int64_t nx,
              double* __restrict__ parts, int slot0) {
    constexpr int N = PackN<T>::N;
    const int m = V.m;
    T u1[MM], u2h[MM];
    compact_coefs<T, MM>(C, u1, u2h);
    const T H0 = (T)C.H0;
    const T gl = gamma * P.g_lambda;
    constexpr int NS = 10 + 4 * MM + 2;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    bz_for_chunks<T>(nx, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        ElemLoads<T> L;
        load_params<T, NT>(P, i0, cnt, L, true, true, true);
        Pack<T> dlo = P.D_lo_vec ? ldp<T, NT>(P.D_lo_vec, i0, cnt) : splat(P.D_lo);
        Pack<T> dhi = P.D_hi_vec ? ldp<T, NT>(P.D_hi_vec, i0, cnt) : splat(P.D_hi);
        Pack<T> pyv = P.uni >= 2 ? splat(T(0)) : ldp<T, NT>(yv, i0, cnt);
        // both halves of every lifted vector: h = 0 the x part, h = 1 the s part
        Pack<T> px[2], prp[2], ps[2][MM], py[2][MM], d[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t o = i0 + (h ? nx : 0);
            px[h] = ldp<T, NT>(xs, o, cnt);
            prp[h] = ldp<T, NT>(res_prev, o, cnt);
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < m) { ps[h][i] = ldp<T, NT>(V.S[i], o, cnt); py[h][i] = ldp<T, NT>(V.Y[i], o, cnt); }
            compact_d<T, MM>(m, H0, u1, u2h, prp[h], ps[h], py[h], d[h]);
        }
        Pack<T> pxd[2], pz[2], pr[2], pss[2], pyy[2];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const T xd = px[0].v[e] + d[0].v[e], sd = px[1].v[e] + d[1].v[e];
            pxd[0].v[e] = xd; pxd[1].v[e] = sd;
            // grad F(xs_d)   (k_algrad_slack_elem)
            const SlackOut<T> o1 = slack_elem(P.f_kind, xd, sd, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], pyv.v[e]);
            // forward-backward step   (k_fbstep_slack)
            T t = gamma * o1.gx;
            const T yx = xd - t;
            T u = gamma * o1.gs;
            const T ys = sd - u;
            T gterm;
            const T a = prox_elem(P.g_kind, yx, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm);
            const T b = proj_D(P.D_kind, ys, dlo.v[e], dhi.v[e]);
            const T r1 = xd - a, r2 = sd - b;
            pz[0].v[e] = a; pz[1].v[e] = b; pr[0].v[e] = r1; pr[1].v[e] = r2;
            // grad F(z)
            const SlackOut<T> o2 = slack_elem(P.f_kind, a, b, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], pyv.v[e]);
            // the pair and the stopping norm   (k_update_c)
            const T sv0 = xd - px[0].v[e], sv1 = sd - px[1].v[e];
            const T yy0 = r1 - prp[0].v[e], yy1 = r2 - prp[1].v[e];
            pss[0].v[e] = sv0; pss[1].v[e] = sv1; pyy[0].v[e] = yy0; pyy[1].v[e] = yy1;
            T w0 = r1 / gamma;
            w0 = w0 - o1.gx;
            w0 = w0 + o2.gx;
            T w1 = r2 / gamma;
            w1 = w1 - o1.gs;
            w1 = w1 + o2.gs;
            if (e < cnt) {
                acc[0] += (double)o1.fterm;
                acc[1] += (double)o1.pterm;
                acc[2] += (double)gterm;
                acc[3] += (double)(o1.gx * r1);
                acc[3] += (double)(o1.gs * r2);
                acc[4] += (double)(r1 * r1);
                acc[4] += (double)(r2 * r2);
                acc[5] += (double)o2.fterm;
                acc[6] += (double)o2.pterm;
                acc[7] += (double)(sv0 * yy0);
                acc[7] += (double)(sv1 * yy1);
                acc[8] += (double)(yy0 * yy0);
                acc[8] += (double)(yy1 * yy1);
                acc[9] = nanmax(acc[9], (double)(w0 < T(0) ? -w0 : w0));
                acc[9] = nanmax(acc[9], (double)(w1 < T(0) ? -w1 : w1));
                const T nr0 = T(-1) * r1, nr1 = T(-1) * r2;
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < m) {
                        acc[10 + i] = mul_acc(ps[0][i].v[e], yy0, acc[10 + i]);
                        acc[10 + i] = mul_acc(ps[1][i].v[e], yy1, acc[10 + i]);
                        acc[10 + MM + i] = mul_acc(py[0][i].v[e], yy0, acc[10 + MM + i]);
                        acc[10 + MM + i] = mul_acc(py[1][i].v[e], yy1, acc[10 + MM + i]);
                        acc[10 + 2 * MM + i] = mul_acc(ps[0][i].v[e], nr0, acc[10 + 2 * MM + i]);
                        acc[10 + 2 * MM + i] = mul_acc(ps[1][i].v[e], nr1, acc[10 + 2 * MM + i]);
                        acc[10 + 3 * MM + i] = mul_acc(py[0][i].v[e], nr0, acc[10 + 3 * MM + i]);
                        acc[10 + 3 * MM + i] = mul_acc(py[1][i].v[e], nr1, acc[10 + 3 * MM + i]);
                    }
                acc[10 + 4 * MM] = mul_acc(sv0, nr0, acc[10 + 4 * MM]);
                acc[10 + 4 * MM] = mul_acc(sv1, nr1, acc[10 + 4 * MM]);
                acc[10 + 4 * MM + 1] = mul_acc(yy0, nr0, acc[10 + 4 * MM + 1]);
                acc[10 + 4 * MM + 1] = mul_acc(yy1, nr1, acc[10 + 4 * MM + 1]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t o = i0 + (h ? nx : 0);
            stp<T, NT>(xs_d, o, cnt, pxd[h]);
            if (z) stp<T, NT>(z, o, cnt, pz[h]);
            stp<T, NT>(res, o, cnt, pr[h]);
            stp<T, NT>(s_new, o, cnt, pss[h]);
            stp<T, NT>(y_new, o, cnt, pyy[h]);
        }
    });
    if constexpr (NS == 32) block_reduce_store32(acc, 1u << 9, parts, slot0);
    else block_reduce_store<NS>(acc, 1u << 9, parts, slot0);
}

// ---------------------------------------------------------------------------
// The slack form with the history as ITERATES (r03; the headline kernel's XR = 2 idea on xs = [x; s]): the stored pairs are
// differences of consecutive iterates and of their fixed-point residuals, and the residual of an iterate is a function of
// that iterate alone — res = xs - [prox_{gamma g}; proj_D](xs - gamma grad F(xs)) with gamma, mu, mu*y, y fixed along
// the run (src/utilities/auglagfunslack.jl:78-97,136-154).  So the pass reads the m + 1 last iterates (both halves), re-
// evaluates their residuals in registers — the operations of k_fused_slack on the same operands, so the same bits — forms
// the pairs by the very subtractions that produced the stored ones, and from there on IS k_fused_slack; it writes xs_d
// only (z on request): 2 (m + 1) + 2 passes over n plus the parameter vectors instead of 2 (2m + 7): 1.5 GB instead of
// 2.96 GB per iteration at n = 1e7, m = 5.  XH[0..m]: the iterates, oldest first, XH[m] the current one; gam0: the step
// size the OLDEST iterate's residual was formed with (CompactCoef::gam0; every younger one: gamma).
// ---------------------------------------------------------------------------
template <class T, int MM> struct SlackIterates {
    const T* XH[MM + 1];
    int m;
};
template <class T>
__device__ __forceinline__ void slack_resid(const ElemParams<T>& P, T x, T sv, const ElemLoads<T>& L, int e, T dlo, T dhi, T yv,
                                            T gam, T& rx, T& rs, T& zx, T& zs, bool udiv = false, T rmu = T(0)) {
    const SlackOut<T> o = slack_elem(P.f_kind, x, sv, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], yv, udiv, rmu);
    T t = gam * o.gx;
    const T yx = x - t;
    T u = gam * o.gs;
    const T ys = sv - u;
    T gterm;
    zx = prox_elem(P.g_kind, yx, gam * P.g_lambda, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm);
    zs = proj_D(P.D_kind, ys, dlo, dhi);
    rx = x - zx;
    rs = sv - zs;
}
// FULL: the memory holds MM pairs (the steady state): compile-time trip counts, so the 2 (MM + 1) iterate loads issue back
// to back with no branch between them (with a branch around a load the compiler waits for every load at the first use)
// UNI >= 0 (with FULL): the fast instantiations — f = DiagQuadratic, no vector-valued parameters of g or D, the penalties
// streamed (0), mu a number (1), mu and mu*y = y = 0 numbers (2): every load of the pass is then unconditional too.
// UNI = -1: run-time everything.
// KIND = 1 (fast only): g = NormL1 and D = Box as compile-time facts too (the ALS form of cfg 2: with run-time kinds the
// seven evaluations of prox_g / proj_D per index are ladders of wave-uniform branches, a third of the pass's instructions).
// DEPTH: packs of loads kept in flight ahead of the one being consumed (0: none, the loads of a pack issue at its start).
template <class T, int MM, bool NT, bool FULL = false, int UNI = -1, int KIND = 0, int DEPTH = 0>
__global__ void __launch_bounds__(BLOCK)
k_fused_slack_xr(SlackIterates<T, MM> V, CompactCoef<MM> C, ElemParams<T> P, const T* __restrict__ yv, T gamma,
                 T* __restrict__ xs_d, T* __restrict__ z, int64_t nx, double* __restrict__ parts, int slot0) {
    constexpr int N = PackN<T>::N;
    static_assert(UNI < 0 || FULL, "the fast instantiations are for a full memory");
    const int m = FULL ? MM : V.m;
    static_assert((KIND == 0 && DEPTH == 0) || UNI >= 0, "compile-time kinds and the pipeline belong to the fast instantiations");
    if constexpr (UNI >= 0) { P.f_kind = BZ_F_DIAG_QUADRATIC; P.uni = UNI; }
    if constexpr (KIND == 1) { P.g_kind = BZ_G_NORM_L1; P.D_kind = BZ_D_BOX; }
    T u1[MM], u2h[MM];
    compact_coefs<T, MM>(C, u1, u2h);
    const T H0 = (T)C.H0;
    const T gl = gamma * P.g_lambda;
    const T gam0 = (T)C.gam0;
    // uniform penalties (k_muy's probe): mu is a number, not a stream, and its quotients go through div_u; zero multipliers
    // as well: mu*y and y are the number 0 (same operands, same operations, same bits as the streamed forms)
    const bool udiv = P.uni >= 1 && std::is_same<T, double>::value;
    const T rmu = udiv ? T(1) / P.mu_uniform : T(0);
    constexpr int NS = 10 + 4 * MM + 2;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    // The fast instantiations run a software pipeline in registers, two packs ahead (the headline kernel's, k_fused_compact):
    // the 14-17 loads of the thread's NEXT pack — every stream as scalar base + ONE shared 32-bit offset — are issued before
    // this pack's ~1500 instructions run, so a wave always has a full pack of loads in flight.  Needs nx * sizeof(T) < 4 GiB.
    constexpr bool PIPE = DEPTH > 0;
    struct Stage { Pack<T> q, b, mu, muy, yv, xh[2][MM + 1]; };
    auto load_stage = [&](Stage& S, unsigned bo) {
        asm volatile("" : "+v"(bo));
        S.q = ldo<T, NT>(P.q, bo); S.b = ldo<T, NT>(P.b, bo);
        if constexpr (UNI < 1) S.mu = ldo<T, NT>(P.mu, bo);
        if constexpr (UNI < 2) { S.muy = ldo<T, NT>(P.muy, bo); S.yv = ldo<T, NT>(yv, bo); }
#pragma unroll
        for (int i = 0; i <= MM; ++i) {
            S.xh[0][i] = ldo<T, NT>(V.XH[i], bo);
            S.xh[1][i] = ldo<T, NT>(V.XH[i] + nx, bo);
        }
    };
    auto body = [&](const int64_t i0, const auto cnt_, const auto staged_, const Stage& SG) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        constexpr bool STAGED = std::remove_cv_t<decltype(staged_)>::value;
        ElemLoads<T> L;
        Pack<T> dlo, dhi, pyv;
        Pack<T> xh[2][MM + 1], rh[2][MM + 1];
        if constexpr (STAGED) {
            L.q = SG.q; L.b = SG.b;
            L.mu = splat(P.mu_uniform); L.muy = splat(T(0)); pyv = splat(T(0));
            if constexpr (UNI < 1) L.mu = SG.mu;
            if constexpr (UNI < 2) { L.muy = SG.muy; pyv = SG.yv; }
            L.dlo = splat(P.D_lo); L.dhi = splat(P.D_hi); L.gu = splat(T(0)); L.glo = splat(P.g_lo); L.ghi = splat(P.g_hi);
            dlo = L.dlo; dhi = L.dhi;
#pragma unroll
            for (int i = 0; i <= MM; ++i) { xh[0][i] = SG.xh[0][i]; xh[1][i] = SG.xh[1][i]; }
        } else {
        if constexpr (UNI >= 0) {
            L.q = ldp<T, NT>(P.q, i0, cnt); L.b = ldp<T, NT>(P.b, i0, cnt);
            L.mu = splat(P.mu_uniform); L.muy = splat(T(0)); pyv = splat(T(0));
            if constexpr (UNI < 1) L.mu = ldp<T, NT>(P.mu, i0, cnt);
            if constexpr (UNI < 2) { L.muy = ldp<T, NT>(P.muy, i0, cnt); pyv = ldp<T, NT>(yv, i0, cnt); }
            L.dlo = splat(P.D_lo); L.dhi = splat(P.D_hi); L.gu = splat(T(0)); L.glo = splat(P.g_lo); L.ghi = splat(P.g_hi);
            dlo = L.dlo; dhi = L.dhi;
        } else {
            load_params<T, NT>(P, i0, cnt, L, true, true, true);
            dlo = P.D_lo_vec ? ldp<T, NT>(P.D_lo_vec, i0, cnt) : splat(P.D_lo);
            dhi = P.D_hi_vec ? ldp<T, NT>(P.D_hi_vec, i0, cnt) : splat(P.D_hi);
            pyv = P.uni >= 2 ? splat(T(0)) : ldp<T, NT>(yv, i0, cnt);
        }
        // the iterates, both halves (h = 0 the x part, h = 1 the s part), and their residuals
#pragma unroll
        for (int i = 0; i <= MM; ++i)
            if (i <= m) {
                xh[0][i] = ldp<T, NT>(V.XH[i], i0, cnt);
                xh[1][i] = ldp<T, NT>(V.XH[i], i0 + nx, cnt);
            }
        }
#pragma unroll
        for (int i = 0; i <= MM; ++i)
            if (i <= m) {
                const T gi = (i == 0 && m > 0) ? gam0 : gamma;
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    T zx, zs;
                    slack_resid(P, xh[0][i].v[e], xh[1][i].v[e], L, e, dlo.v[e], dhi.v[e], pyv.v[e], gi, rh[0][i].v[e], rh[1][i].v[e], zx, zs, udiv, rmu);
                }
            }
        Pack<T> px[2], prp[2], ps[2][MM], py[2][MM], d[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < m) {
#pragma unroll
                    for (int e = 0; e < N; ++e) {
                        ps[h][i].v[e] = xh[h][i + 1].v[e] - xh[h][i].v[e];
                        py[h][i].v[e] = rh[h][i + 1].v[e] - rh[h][i].v[e];
                    }
                }
            // (the current iterate and its residual: entry m)
            px[h] = xh[h][0]; prp[h] = rh[h][0];
#pragma unroll
            for (int i = 1; i <= MM; ++i)
                if (i == m) { px[h] = xh[h][i]; prp[h] = rh[h][i]; }
            compact_d<T, MM>(m, H0, u1, u2h, prp[h], ps[h], py[h], d[h]);
        }
        Pack<T> pxd[2], pz[2];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const T xd = px[0].v[e] + d[0].v[e], sd = px[1].v[e] + d[1].v[e];
            pxd[0].v[e] = xd; pxd[1].v[e] = sd;
            // grad F(xs_d)   (k_algrad_slack_elem)
            const SlackOut<T> o1 = slack_elem(P.f_kind, xd, sd, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], pyv.v[e], udiv, rmu);
            // forward-backward step   (k_fbstep_slack)
            T t = gamma * o1.gx;
            const T yx = xd - t;
            T u = gamma * o1.gs;
            const T ys = sd - u;
            T gterm;
            const T a = prox_elem(P.g_kind, yx, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm);
            const T b = proj_D(P.D_kind, ys, dlo.v[e], dhi.v[e]);
            const T r1 = xd - a, r2 = sd - b;
            pz[0].v[e] = a; pz[1].v[e] = b;
            // grad F(z)
            const SlackOut<T> o2 = slack_elem(P.f_kind, a, b, L.q.v[e], L.b.v[e], L.mu.v[e], L.muy.v[e], pyv.v[e], udiv, rmu);
            // the pair and the stopping norm   (k_update_c)
            const T sv0 = xd - px[0].v[e], sv1 = sd - px[1].v[e];
            const T yy0 = r1 - prp[0].v[e], yy1 = r2 - prp[1].v[e];
            T w0 = r1 / gamma;
            w0 = w0 - o1.gx;
            w0 = w0 + o2.gx;
            T w1 = r2 / gamma;
            w1 = w1 - o1.gs;
            w1 = w1 + o2.gs;
            if (e < cnt) {
                acc[0] += (double)o1.fterm;
                acc[1] += (double)o1.pterm;
                acc[2] += (double)gterm;
                acc[3] += (double)(o1.gx * r1);
                acc[3] += (double)(o1.gs * r2);
                acc[4] += (double)(r1 * r1);
                acc[4] += (double)(r2 * r2);
                acc[5] += (double)o2.fterm;
                acc[6] += (double)o2.pterm;
                acc[7] += (double)(sv0 * yy0);
                acc[7] += (double)(sv1 * yy1);
                acc[8] += (double)(yy0 * yy0);
                acc[8] += (double)(yy1 * yy1);
                acc[9] = nanmax(acc[9], (double)(w0 < T(0) ? -w0 : w0));
                acc[9] = nanmax(acc[9], (double)(w1 < T(0) ? -w1 : w1));
                const T nr0 = T(-1) * r1, nr1 = T(-1) * r2;
#pragma unroll
                for (int i = 0; i < MM; ++i)
                    if (i < m) {
                        acc[10 + i] = mul_acc(ps[0][i].v[e], yy0, acc[10 + i]);
                        acc[10 + i] = mul_acc(ps[1][i].v[e], yy1, acc[10 + i]);
                        acc[10 + MM + i] = mul_acc(py[0][i].v[e], yy0, acc[10 + MM + i]);
                        acc[10 + MM + i] = mul_acc(py[1][i].v[e], yy1, acc[10 + MM + i]);
                        acc[10 + 2 * MM + i] = mul_acc(ps[0][i].v[e], nr0, acc[10 + 2 * MM + i]);
                        acc[10 + 2 * MM + i] = mul_acc(ps[1][i].v[e], nr1, acc[10 + 2 * MM + i]);
                        acc[10 + 3 * MM + i] = mul_acc(py[0][i].v[e], nr0, acc[10 + 3 * MM + i]);
                        acc[10 + 3 * MM + i] = mul_acc(py[1][i].v[e], nr1, acc[10 + 3 * MM + i]);
                    }
                acc[10 + 4 * MM] = mul_acc(sv0, nr0, acc[10 + 4 * MM]);
                acc[10 + 4 * MM] = mul_acc(sv1, nr1, acc[10 + 4 * MM]);
                acc[10 + 4 * MM + 1] = mul_acc(yy0, nr0, acc[10 + 4 * MM + 1]);
                acc[10 + 4 * MM + 1] = mul_acc(yy1, nr1, acc[10 + 4 * MM + 1]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (STAGED) {
                unsigned bo = (unsigned)(i0 * (int64_t)sizeof(T));
                asm volatile("" : "+v"(bo));
                sto<T, NT>(h ? xs_d + nx : xs_d, bo, pxd[h]);
                if (z) sto<T, NT>(h ? z + nx : z, bo, pz[h]);
            } else {
                const int64_t o = i0 + (h ? nx : 0);
                stp<T, NT>(xs_d, o, cnt, pxd[h]);
                if (z) stp<T, NT>(z, o, cnt, pz[h]);
            }
        }
    };
    if constexpr (PIPE) {
        // the chunk -> thread map of bz_for_chunks (so the sums are those of every other form on this grid, bit for bit); past
        // the end a thread re-requests the last full chunk: no branch around a load, nothing out of bounds
        const int64_t nfull = nx / N;
        const int64_t stride = (int64_t)gridDim.x * BLOCK;
        int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        auto fetch = [&](Stage& S, int64_t k) { load_stage(S, (unsigned)((k < nfull ? k : nfull - 1) * N * (int64_t)sizeof(T))); };
        auto use = [&](const Stage& S, int64_t k) { body(k * N, std::integral_constant<int, N>{}, std::true_type{}, S); };
        Stage sa, sb;
        if constexpr (DEPTH == 2) {
            Stage sc;
            if (c < nfull) { fetch(sa, c); fetch(sb, c + stride); }
            for (;;) {      // three stages in rotation, the loop unrolled by three: no register copies between iterations
                if (c >= nfull) break;
                fetch(sc, c + 2 * stride); use(sa, c); c += stride;
                if (c >= nfull) break;
                fetch(sa, c + 2 * stride); use(sb, c); c += stride;
                if (c >= nfull) break;
                fetch(sb, c + 2 * stride); use(sc, c); c += stride;
            }
        } else {
            if (c < nfull) fetch(sa, c);
            for (;;) {
                if (c >= nfull) break;
                fetch(sb, c + stride); use(sa, c); c += stride;
                if (c >= nfull) break;
                fetch(sa, c + stride); use(sb, c); c += stride;
            }
        }
        if (c == nfull && nfull * N < nx) body(c * N, (int)(nx - nfull * N), std::false_type{}, sa);
    } else {
        Stage none;
        bz_for_chunks<T>(nx, [&](const int64_t i0, const auto cnt_) { body(i0, cnt_, std::false_type{}, none); });
    }
    if constexpr (NS == 32) block_reduce_store32(acc, 1u << 9, parts, slot0);
    else block_reduce_store<NS>(acc, 1u << 9, parts, slot0);
}

// ... and back to stored pairs when an iteration leaves the plain path (a rejected trial point, a skipped pair, a gamma
// halving): S[i] = XH[i+1] - XH[i], Y[i] = r(XH[i+1]) - r(XH[i]) with r re-evaluated, plus the residual and z of the
// current iterate (the slack twin of k_pairs_from_iterates; gam[i]: the step size of each iterate's residual)
template <class T, int MM>
__global__ void __launch_bounds__(BLOCK)
k_pairs_from_iterates_slack(SnapVecs<T, MM> V, int m, ElemParams<T> P, const T* __restrict__ yv, T* __restrict__ res_cur,
                            T* __restrict__ z_cur, int64_t nx) {
    constexpr int N = PackN<T>::N;
    bz_for_chunks<T>(nx, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, true, true, true);
        Pack<T> dlo = P.D_lo_vec ? ld(P.D_lo_vec, i0, cnt) : splat(P.D_lo);
        Pack<T> dhi = P.D_hi_vec ? ld(P.D_hi_vec, i0, cnt) : splat(P.D_hi);
        Pack<T> pyv = P.uni >= 2 ? splat(T(0)) : ld(yv, i0, cnt);
        Pack<T> xh[2][MM + 1], rh[2][MM + 1], pz[2];
#pragma unroll
        for (int i = 0; i <= MM; ++i) {
            xh[0][i] = ld(V.XH[i], i0, cnt);
            xh[1][i] = ld(V.XH[i], i0 + nx, cnt);
            const T gi = (T)V.gam[i];
#pragma unroll
            for (int e = 0; e < N; ++e) {
                T zx, zs;
                slack_resid(P, xh[0][i].v[e], xh[1][i].v[e], L, e, dlo.v[e], dhi.v[e], pyv.v[e], gi, rh[0][i].v[e], rh[1][i].v[e], zx, zs);
                if (i == MM) { pz[0].v[e] = zx; pz[1].v[e] = zs; }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t o = i0 + (h ? nx : 0);
#pragma unroll
            for (int i = 0; i < MM; ++i) {
                Pack<T> sp, yp;
#pragma unroll
                for (int e = 0; e < N; ++e) { sp.v[e] = xh[h][i + 1].v[e] - xh[h][i].v[e]; yp.v[e] = rh[h][i + 1].v[e] - rh[h][i].v[e]; }
                if (i < m) {
                    st(V.S[i], o, cnt, sp);
                    st(V.Y[i], o, cnt, yp);
                }
            }
            st(res_cur, o, cnt, rh[h][MM]);
            st(z_cur, o, cnt, pz[h]);
        }
    });
}

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
// K6: x = tau*x_d + (1-tau)*z_curr
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_blend(const T* __restrict__ x_d, const T* __restrict__ z_curr, T tau, T omt,
        T* __restrict__ x, int64_t n) {
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> a = ld(x_d, i0, cnt), b = ld(z_curr, i0, cnt), o;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T p = tau * a.v[e];
            T q = omt * b.v[e];
            o.v[e] = p + q;
        }
        st(x, i0, cnt, o);
    });
}

// out = in + c
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_add_scalar(const T* __restrict__ in, T c, T* __restrict__ out, int64_t n) {
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> a = ld(in, i0, cnt), o;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) o.v[e] = a.v[e] + c;
        st(out, i0, cnt, o);
    });
}

// slots: +0 sum (a-b)^2, +1 sum (c-d)^2     (lower_bound_smoothness_constant)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_diff_ss2(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
           const T* __restrict__ d, int64_t n, double* __restrict__ parts, int slot0) {
    double acc[2] = {0.0, 0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pa = ld(a, i0, cnt), pb = ld(b, i0, cnt), pc = ld(c, i0, cnt), pd = ld(d, i0, cnt);
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e)
            if (e < cnt) {
                T u = pa.v[e] - pb.v[e];
                T v = pc.v[e] - pd.v[e];
                acc[0] += (double)(u * u);
                acc[1] += (double)(v * v);
            }
    });
    block_reduce_store<2>(acc, 0u, parts, slot0);
}

// ---------------------------------------------------------------------------
// K8: outer-loop element-wise work (alps.jl:62,72-84,97 ; auglagfun.jl:95-98 ;
// safeguards.jl:2-18), kept on the device so ny-vectors never cross PCIe
// ---------------------------------------------------------------------------
// AugLagUpdate!: muy = mu.*y ; slots +0 sum muy*y, +1 max(mu<=0)
//   do_clamp: first y = clamp(y, -1e20, 1e20), stored back (default_dual_safeguard!, alps.jl:62 — k_clamp_scale's
//             arithmetic) ; slot_probe >= 0: also k_uniform_probe's three maxima (mu and mu*y are in registers here)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_muy(const T* __restrict__ mu, T* __restrict__ y, T* __restrict__ muy, int64_t n,
      double* __restrict__ parts, int slot0, int do_clamp, int slot_probe) {
    double acc[2] = {0.0, 0.0}, pr[3] = {0.0, 0.0, 0.0};
    const T m0 = mu[0];
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pm = ld(mu, i0, cnt), py = ld((const T*)y, i0, cnt), o;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            if (do_clamp) {
                double w = (double)py.v[e];
                w = w < 1e20 ? w : 1e20;      // min(y, 1e20)
                w = w > -1e20 ? w : -1e20;    // max(-1e20, .)
                py.v[e] = (T)w;
            }
            T m = pm.v[e] * py.v[e];
            o.v[e] = m;
            if (e < cnt) {
                acc[0] += (double)(m * py.v[e]);
                acc[1] = nanmax(acc[1], (pm.v[e] <= T(0)) ? 1.0 : 0.0);
                pr[0] = nanmax(pr[0], (double)pm.v[e]);
                pr[1] = nanmax(pr[1], pm.v[e] == m0 ? 0.0 : 1.0);
                pr[2] = nanmax(pr[2], (double)(m < T(0) ? -m : m));
            }
        }
        if (do_clamp) st(y, i0, cnt, py);
        st(muy, i0, cnt, o);
    });
    block_reduce_store<2>(acc, 2u, parts, slot0);
    if (slot_probe >= 0) block_reduce_store<3>(pr, 7u, parts, slot_probe);
}

// are the penalties uniform, are the scaled multipliers zero?  slots (all max, all >= 0 as the folds assume):
//   +0 max mu, +1 max (mu[i] != mu[0]), +2 max |mu*y|
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_uniform_probe(const T* __restrict__ mu, const T* __restrict__ muy, int64_t n, double* __restrict__ parts,
                int slot0) {
    double acc[3] = {0.0, 0.0, 0.0};
    const T m0 = mu[0];
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;
        Pack<T> pm = ld(mu, i0, cnt), py = ld(muy, i0, cnt);
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e)
            if (e < cnt) {
                acc[0] = nanmax(acc[0], (double)pm.v[e]);
                acc[1] = nanmax(acc[1], pm.v[e] == m0 ? 0.0 : 1.0);
                acc[2] = nanmax(acc[2], (double)(py.v[e] < T(0) ? -py.v[e] : py.v[e]));
            }
    });
    block_reduce_store<3>(acc, 7u, parts, slot0);
}

// dual update with c = Identity (alps.jl:72-84):
//   cx = x ; y = cx + muy ; s = proj_D(y) ; y -= s ; y /= mu ; slot +0 max |cx - s|
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_dual_update(const T* __restrict__ cx, ElemParams<T> P, T* __restrict__ y, T* __restrict__ s,
              int64_t n, double* __restrict__ parts, int slot0) {
    double acc[1] = {0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        ElemLoads<T> L;
        load_params(P, i0, cnt, L, false, true, false);
        Pack<T> pc = ld(cx, i0, cnt), py, ps;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T t = pc.v[e] + L.muy.v[e];
            T sv = proj_D(P.D_kind, t, L.dlo.v[e], L.dhi.v[e], pc.v[e ^ 1] + L.muy.v[e ^ 1], e & 1);
            t = t - sv;
            t = t / L.mu.v[e];
            py.v[e] = t; ps.v[e] = sv;
            T r = pc.v[e] - sv;
            if (e < cnt) acc[0] = nanmax(acc[0], (double)(r < T(0) ? -r : r));
        }
        st(y, i0, cnt, py);
        st(s, i0, cnt, ps);
    });
    block_reduce_store<1>(acc, 1u, parts, slot0);
}

// s = proj_D(cx) ; mu = clamp(0.1*max(1, 0.5 (cx-s)^2)/max(1,objx), 1e-8, 1e8)
// (alps.jl:41-42, safeguards.jl:13-18; Float64 literals, stored back into T)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_penalty_init(const T* __restrict__ cx, ElemParams<T> P, double denom, T* __restrict__ s,
               T* __restrict__ mu, int64_t n) {
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> pc = ld(cx, i0, cnt), ps, pm;
        Pack<T> dlo = P.D_lo_vec ? ld(P.D_lo_vec, i0, cnt) : splat(P.D_lo);
        Pack<T> dhi = P.D_hi_vec ? ld(P.D_hi_vec, i0, cnt) : splat(P.D_hi);
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T sv = proj_D(P.D_kind, pc.v[e], dlo.v[e], dhi.v[e], pc.v[e ^ 1], e & 1);
            T d = pc.v[e] - sv;
            double d2 = (double)(d * d);
            double h = 0.5 * d2;
            T m = (T)((h > 1.0 ? h : 1.0) / denom);
            m = (T)((double)m * 0.1);
            double mm = (double)m;
            mm = mm < 1e8 ? mm : 1e8;
            mm = mm > 1e-8 ? mm : 1e-8;
            ps.v[e] = sv; pm.v[e] = (T)mm;
        }
        st(s, i0, cnt, ps);
        st(mu, i0, cnt, pm);
    });
}

// v = clamp(v, lo, hi)  (default_dual_safeguard!) ; v *= c
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_clamp_scale(T* __restrict__ v, double lo, double hi, T scale, int do_clamp, int64_t n) {
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> a = ld((const T*)v, i0, cnt), o;
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            T t = a.v[e];
            if (do_clamp) {
                double w = (double)t;
                w = w < hi ? w : hi;      // min(y, 1e20)
                w = w > lo ? w : lo;      // max(-1e20, .)
                t = (T)w;
            } else {
                t = t * scale;
            }
            o.v[e] = t;
        }
        st(v, i0, cnt, o);
    });
}

// slot +0: max |v|   (verbose display: ||res||_inf)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_absmax(const T* __restrict__ v, int64_t n, double* __restrict__ parts, int slot0) {
    double acc[1] = {0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> a = ld(v, i0, cnt);
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e)
            if (e < cnt) acc[0] = nanmax(acc[0], (double)(a.v[e] < T(0) ? -a.v[e] : a.v[e]));
    });
    block_reduce_store<1>(acc, 1u, parts, slot0);
}

// f(x) alone for element-wise f (alps.jl:39): slot +0 sum f terms
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_fvalue_elem(const T* __restrict__ x, ElemParams<T> P, int64_t n, double* __restrict__ parts,
              int slot0, const T* __restrict__ ext) {
    double acc[1] = {0.0};
    bz_for_chunks<T>(n, [&](const int64_t i0, const auto cnt_) {
        const int cnt = cnt_;      // compile-time PackN in the main loop, run-time only for the ragged last chunk
        Pack<T> px = ld(x, i0, cnt);
        Pack<T> q = splat(T(0)), b = splat(T(0));
        if (P.f_kind == BZ_F_DIAG_QUADRATIC) { q = ld(P.q, i0, cnt); b = ld(P.b, i0, cnt); }
        if (ext) { q = ld(ext, i0, cnt); b = ld(P.b, i0, cnt); }      // dense Quadratic: q := Q x, b := q
#pragma unroll
        for (int e = 0; e < PackN<T>::N; ++e) {
            if (e < cnt && ext) {
                acc[0] += (double)(px.v[e] * (T(0.5) * q.v[e] + b.v[e]));
            } else if (e < cnt && P.f_kind == BZ_F_DIAG_QUADRATIC) {
                T qx = q.v[e] * px.v[e];
                acc[0] += (double)(px.v[e] * (T(0.5) * qx - b.v[e]));
            }
        }
    });
    block_reduce_store<1>(acc, 0u, parts, slot0);
}

// ---------------------------------------------------------------------------
// Broyden() directions (ProximalAlgorithms; the modified Broyden update of the PANOC papers, Themelis & Patrinos,
// "SuperMann", IEEE TAC 2019, §VI-A): a dense n-by-n operator H, row-major; the tiny-n demos only
// ---------------------------------------------------------------------------
// H = I
template <class T>
__global__ void __launch_bounds__(BLOCK) k_set_identity(T* __restrict__ H, int64_t n) {
    const int64_t tot = n * n;
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < tot; k += (int64_t)gridDim.x * BLOCK)
        H[k] = (k / n == k % n) ? T(1) : T(0);
}
// H += ((s - Hy) * inv_denom) (x) sH        (rank-one update, element (i, j): u_i * sH_j)
template <class T>
__global__ void __launch_bounds__(BLOCK)
k_rank1_update(T* __restrict__ H, const T* __restrict__ s, const T* __restrict__ Hy, const T* __restrict__ sH,
               T inv_denom, int64_t n) {
    const int64_t tot = n * n;
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < tot; k += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = k / n, j = k - i * n;
        const T u = (s[i] - Hy[i]) * inv_denom;
        H[k] = H[k] + u * sH[j];
    }
}

// ---------------------------------------------------------------------------
// scalar plumbing
// ---------------------------------------------------------------------------
// one host-given number as a reduction slot with a single partial (so that it can travel through an exchange)
static __global__ void __launch_bounds__(64) k_fill_slot(double* parts, int slot, double v) {
    if (threadIdx.x == 0) parts[(size_t)slot * PSTRIDE] = v;
}
constexpr int MAX_COLLECT = 40;
struct CollectArgs {
    ScalarSrc src[MAX_COLLECT];
    unsigned maxmask;
    int n;
    unsigned long long ticket;      // sequence number of this read-back
};
// one block per source: fold it and post it (tagged words, host_post) to the host-mapped mailbox `out`; the
// host spins on the tags instead of paying a blocking stream synchronisation.  (One 1024-thread block folding
// all 32 sources, 16 at a time, measured slower: 8.6 us against 5.5 us for 32 small blocks.)
__global__ void __launch_bounds__(BLOCK) k_collect(CollectArgs a, double* out);

}  // namespace bz
