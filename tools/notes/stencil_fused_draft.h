// DRAFT, NOT COMPILED INTO THE LIBRARY (round 3 notes; NEXT.md "cfg 3: both stencil passes in one").
// This version is CORRECT — integrated behind BZ_STENCIL_FUSED it passed the stencil parity tests (oracle iterates, 1e-10 against
// the two-pass form over ragged tilings) — but SLOWER than the two passes it replaces: 204 us against 38 + 116 us at 2048^2.
// Why: every load sits behind a validity test (grid edge, ragged tile), so the compiler waits for all loads in flight at each
// use — about three serialised memory latencies per tile row, 18 row steps per tile, two rounds of tiles.  A second version with a
// compile-time INTERIOR path (no tests, unconditional loads) let the compiler's straight-line code for the 10 + 8 unrolled rows
// run to 256 VGPRs + 256 AGPRs + 740 spills.  What the next attempt needs: the row loops NOT unrolled (grad L(x_d) and res of the
// tile recomputed or parked in LDS instead of 2 x SF_TR register packs), uniform-base + 32-bit-offset addressing, one row's loads
// issued as a batch (inline asm as in k_dense_fused if the compiler will not), interior tiles only on the fast path.
// ---------------------------------------------------------------------------
// cfg 3: BOTH stencil passes of an iteration in one (r03).  k_stencil_fb writes grad L(x_d), z and res, and
// k_stencil_update_c reads them back because grad L(z) needs z's four neighbours: 30 streams for the two passes.  Here a
// workgroup owns a TILE of SF_TR rows x 256 packs: it forms z on the tile plus a one-cell halo (the halo rows by the same
// lanes — two more turns of the row loop —, the halo columns by the tile's first and last lane, one scalar each) and keeps it
// in LDS; grad L(x_d) and res of the tile's SF_TR packs per lane stay in registers across the one barrier; grad L(z) then
// takes its neighbours from LDS.  HBM sees x_d once (plus the halo rows, out of L2 mostly), b and the bounds twice (the second
// time out of L2), x_prev, res_prev, the history; res, z, s_new, y_new go out: ~19 streams instead of 30.
// Same element arithmetic as stencil_al_pack / k_stencil_fb / k_stencil_update_c, operation for operation; the reductions run
// over tiles instead of the canonical chunk order, so the scalars agree with the two-pass form to rounding, not bit for bit.
// Dirichlet-0 outside the grid; single rank (a row-sharded grid keeps the two passes with their halo exchange).
// ---------------------------------------------------------------------------
constexpr int SF_TR = 16;
template <class T> constexpr int sf_row_stride() { return BLOCK * PackN<T>::N + 2; }            // elements of T per LDS row
template <class T> constexpr size_t sf_lds_bytes() { return (size_t)(SF_TR + 2) * sf_row_stride<T>() * sizeof(T); }

// z = prox_{gamma g}(x - gamma grad L(x)) at ONE grid point (r, c): the tile's halo columns
template <class T>
__device__ __forceinline__ T stencil_z_point(const T* __restrict__ x, const ElemParams<T>& P, int64_t nx, int64_t ny, int64_t r,
                                             int64_t c, T gamma, T gl) {
    const int64_t i = r * ny + c;
    const T xc = x[i];
    const T w = c > 0 ? x[i - 1] : T(0), ee = c + 1 < ny ? x[i + 1] : T(0);
    const T xn = r > 0 ? x[i - ny] : T(0), xs = r + 1 < nx ? x[i + ny] : T(0);
    T Ax = T(4) * xc;
    Ax = Ax - w;
    Ax = Ax - ee;
    Ax = Ax - xn;
    Ax = Ax - xs;
    const T dfx = Ax - P.b[i];
    const T mu = P.uni >= 1 ? P.mu_uniform : P.mu[i];
    const T muy = P.uni >= 2 ? T(0) : P.muy[i];
    const T lo = P.D_lo_vec ? P.D_lo_vec[i] : P.D_lo, hi = P.D_hi_vec ? P.D_hi_vec[i] : P.D_hi;
    T t = xc + muy;
    const T sv = proj_D(P.D_kind, t, lo, hi);
    t = t - sv;
    const T yupd = t / mu;
    const T g = dfx + yupd;
    const T tt = gamma * g;
    const T y = xc - tt;
    const bool hu = P.g_u && (P.g_kind == BZ_G_NORM_L1_BOX || P.g_kind == BZ_G_NORM_L0_BOX || P.g_kind == BZ_G_NORM_LP_BOX);
    T gterm;
    return prox_elem(P.g_kind, y, gl, hu ? P.g_u[i] : T(0), P.g_lo_vec ? P.g_lo_vec[i] : P.g_lo, P.g_hi_vec ? P.g_hi_vec[i] : P.g_hi, gterm);
}

template <class T, int MM, bool FULL = false, bool NT = false>
__global__ void __launch_bounds__(BLOCK)
k_stencil_fused(CompactVecs<T, MM> V, const T* __restrict__ xd, ElemParams<T> P, int64_t nx, int64_t ny,
                const T* __restrict__ x_prev, const T* __restrict__ res_prev, T gamma, T* __restrict__ res_out,
                T* __restrict__ z_out, T* __restrict__ s_new, T* __restrict__ y_new, int tiles_c,
                double* __restrict__ parts, int slot_f, int slot_g, int slot0) {
    constexpr int N = PackN<T>::N;
    constexpr int RS = sf_row_stride<T>();
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    T* const zt = reinterpret_cast<T*>(sf_smem);                       // zt[rr * RS + 1 + column in the tile], rr = 0 .. SF_TR + 1
    const int t = threadIdx.x;
    const int tc = (int)blockIdx.x % tiles_c, tr = (int)blockIdx.x / tiles_c;
    const int64_t R0 = (int64_t)tr * SF_TR;
    const int64_t npk_row = ny / N;
    const int64_t pk = (int64_t)tc * BLOCK + t;                        // this lane's pack within a row
    const bool lane_ok = pk < npk_row;
    const int64_t c0 = pk * N;                                         // its first column
    const int64_t tile_c0 = (int64_t)tc * BLOCK * N;                   // the tile's first column
    const int64_t tile_c1 = (tile_c0 + (int64_t)BLOCK * N < ny) ? tile_c0 + (int64_t)BLOCK * N : ny;      // one past its last
    const T gl = gamma * P.g_lambda;
    const int m = FULL ? MM : V.m;
    double accF[2] = {0.0, 0.0}, accG[3] = {0.0, 0.0, 0.0}, accF2[2] = {0.0, 0.0};
    constexpr int NS = 5 + 4 * MM + 2;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    Pack<T> gxr[SF_TR], rsr[SF_TR];                                    // grad L(x_d) and res of this lane's packs: stay in registers

    // ---- stage 1: z on the tile + halo -> LDS ; grad L(x_d), res of the tile's own rows -> registers (and res, z -> HBM)
    Pack<T> xm = splat(T(0)), x0 = splat(T(0)), xp = splat(T(0));      // x_d at rows r - 1, r, r + 1 of this lane's pack
    {
        const int64_t ra = R0 - 2, rb = R0 - 1;
        if (lane_ok && ra >= 0) xm = ld(xd, ra * ny + c0, N);
        if (lane_ok && rb >= 0 && rb < nx) x0 = ld(xd, rb * ny + c0, N);
    }
#pragma unroll
    for (int rr = 0; rr < SF_TR + 2; ++rr) {      // (unrolled: gxr / rsr are indexed at compile time)
        const int64_t r = R0 - 1 + rr;
        xp = (lane_ok && r + 1 >= 0 && r + 1 < nx) ? ld(xd, (r + 1) * ny + c0, N) : splat(T(0));
        Pack<T> pz = splat(T(0));
        const bool row_ok = r >= 0 && r < nx;
        if (row_ok && lane_ok) {
            const int64_t i0 = r * ny + c0;
            const T west = c0 > 0 ? xd[i0 - 1] : T(0);
            const T east = c0 + N < ny ? xd[i0 + N] : T(0);
            Pack<T> pb = ldp<T, NT>(P.b, i0, N);
            ElemLoads<T> L;
            load_params<T, NT>(P, i0, N, L, false, true, true);
            const bool own = rr >= 1 && rr <= SF_TR;
            Pack<T> pg, pr;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const T c = x0.v[e];
                const T w = (e == 0) ? west : x0.v[e > 0 ? e - 1 : 0];
                const T ee = (e == N - 1) ? east : x0.v[e < N - 1 ? e + 1 : N - 1];
                T Ax = T(4) * c;
                Ax = Ax - w;
                Ax = Ax - ee;
                Ax = Ax - xm.v[e];
                Ax = Ax - xp.v[e];
                const T dfx = Ax - pb.v[e];
                const T fterm = c * (T(0.5) * Ax - pb.v[e]);
                T tt = c + L.muy.v[e];
                const T sv = proj_D(P.D_kind, tt, L.dlo.v[e], L.dhi.v[e]);
                tt = tt - sv;
                const T pterm = (tt * tt) / L.mu.v[e];
                const T yupd = tt / L.mu.v[e];
                const T g = dfx + yupd;
                pg.v[e] = g;
                T u = gamma * g;
                const T y = c - u;
                T gterm;
                const T zz = prox_elem(P.g_kind, y, gl, L.gu.v[e], L.glo.v[e], L.ghi.v[e], gterm);
                const T rv = c - zz;
                pz.v[e] = zz; pr.v[e] = rv;
                if (own) {
                    accF[0] += (double)fterm; accF[1] += (double)pterm;
                    accG[0] += (double)gterm;
                    accG[1] += (double)(g * rv);
                    accG[2] += (double)(rv * rv);
                }
            }
            if (own) {
#pragma unroll
                for (int q = 0; q < SF_TR; ++q)
                    if (q == rr - 1) { gxr[q] = pg; rsr[q] = pr; }
                stp<T, false>(res_out, i0, N, pr);
                if (z_out) stp<T, false>(z_out, i0, N, pz);
            }
        }
        // this lane's z pack (zero outside the grid) and, from the tile's first / last lane, the halo columns
        T* zrow = zt + (size_t)rr * RS;
#pragma unroll
        for (int e = 0; e < N; ++e) zrow[1 + t * N + e] = pz.v[e];
        if (t == 0) zrow[0] = (row_ok && tile_c0 > 0) ? stencil_z_point(xd, P, nx, ny, r, tile_c0 - 1, gamma, gl) : T(0);
        if (t == BLOCK - 1) zrow[1 + BLOCK * N] = (row_ok && tile_c1 < ny && tile_c1 == tile_c0 + (int64_t)BLOCK * N)
                                                      ? stencil_z_point(xd, P, nx, ny, r, tile_c1, gamma, gl) : T(0);
        xm = x0; x0 = xp;
    }
    __syncthreads();

    // ---- stage 2: grad L(z) from LDS, the pair, the stop norm, the compact form's products
    if (lane_ok) {
#pragma unroll
        for (int q = 0; q < SF_TR; ++q) {
            const int64_t r = R0 + q;
            if (r >= nx) break;
            const int64_t i0 = r * ny + c0;
            const T* zc = zt + (size_t)(q + 1) * RS + 1 + t * N;
            const T* zn = zc - RS;
            const T* zs = zc + RS;
            Pack<T> pb = ldp<T, NT>(P.b, i0, N);
            ElemLoads<T> L;
            load_params<T, NT>(P, i0, N, L, false, true, false);
            Pack<T> px = ld(xd, i0, N), pxp = ldp<T, NT>(x_prev, i0, N), prp = ldp<T, NT>(res_prev, i0, N), hs[MM], hy[MM];
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < m) { hs[i] = ldp<T, NT>(V.S[i], i0, N); hy[i] = ldp<T, NT>(V.Y[i], i0, N); }
            Pack<T> ps, py;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const T c = zc[e];
                T Ax = T(4) * c;
                Ax = Ax - zc[e - 1];
                Ax = Ax - zc[e + 1];
                Ax = Ax - zn[e];
                Ax = Ax - zs[e];
                const T dfx = Ax - pb.v[e];
                const T fterm = c * (T(0.5) * Ax - pb.v[e]);
                T tt = c + L.muy.v[e];
                const T svp = proj_D(P.D_kind, tt, L.dlo.v[e], L.dhi.v[e]);
                tt = tt - svp;
                const T pterm = (tt * tt) / L.mu.v[e];
                const T yupd = tt / L.mu.v[e];
                const T gz = dfx + yupd;
                accF2[0] += (double)fterm; accF2[1] += (double)pterm;
                const T rv = rsr[q].v[e];
                const T sv = px.v[e] - pxp.v[e];
                const T yv = rv - prp.v[e];
                ps.v[e] = sv; py.v[e] = yv;
                T w = rv / gamma;
                w = w - gxr[q].v[e];
                w = w + gz;
                acc[0] += (double)(sv * yv);
                acc[1] += (double)(yv * yv);
                acc[2] = nanmax(acc[2], (double)(w < T(0) ? -w : w));
                const T nr = T(-1) * rv;
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
            stp<T, false>(s_new, i0, N, ps);
            stp<T, false>(y_new, i0, N, py);
        }
    }
    block_reduce_store<2>(accF, 0u, parts, slot_f);
    block_reduce_store<3>(accG, 0u, parts, slot_g);
    __syncthreads();
    block_reduce_store<2>(accF2, 0u, parts, slot0 + 5);
    __syncthreads();
    block_reduce_store<NS>(acc, 4u, parts, slot0 + 7);
}

