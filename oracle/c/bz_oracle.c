/* bz_oracle.c — plain-C CPU restatement of the PANOCplus inner solve on the
 * separable BASELINE configs (cfg 2 / cfg 5: DiagQuadratic f, Identity c,
 * Box/Free/Zero D, L1-family g).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/bazinga_ref.py for the parity statement:
 * solution-level KATs pinned, iterate level "parity unpinned").  Used (a) to
 * cross-check the numpy restatement at sizes numpy is slow at and (b) as
 * bench.py's cpu_baseline, kind "port": single-threaded and REFERENCE-SHAPED,
 * i.e. one loop per Julia broadcast statement, no fusion across statements,
 * the same temporaries the Julia code allocates:
 *   gradient!(dlx, al::AugLagFun, x)    src/utilities/auglagfun.jl:73-86
 *   prox!(z, g::NonsmoothCostFun, ...)  src/utilities/nonsmoothcostfun.jl:17-22
 *   IdentityFunction eval!/jtprod!      test/definitions/identityFunction.jl:3-13
 *   proj! Zero/Free/IndicatorSet        src/projections/*.jl
 *   PANOCplus / LBFGS / f_model         ProximalAlgorithms.jl (external; restated,
 *                                       call site src/algorithms/alps.jl:64-66)
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef double real;

/* cpu_baseline "all cores" build (oracle/Makefile: libbz_oracle_omp.so, -fopenmp): the SAME
 * reference-shaped loops, each one split over the host threads; reductions become per-thread partial
 * sums (a different rounding — this build is only ever timed, never used as a checker). */
#ifdef _OPENMP
#define PFOR _Pragma("omp parallel for schedule(static)")
#define PRED_ACC _Pragma("omp parallel for schedule(static) reduction(+:acc)")
#else
#define PFOR
#define PRED_ACC
#endif

enum { D_ZERO = 0, D_FREE = 1, D_BOX = 2 };
enum { G_ZERO = 0, G_L1 = 1, G_L1_NONNEG = 2, G_L1_BOX = 3, G_IND_BOX = 4 };

typedef struct {
    int64_t n;
    int D_kind, g_kind;
    const real *q, *b;          /* f(x) = sum x(0.5 q x - b) */
    real lam;                   /* g */
    const real* g_u;
    real g_lo, g_hi;
    real D_lo, D_hi;
    /* AugLagFun state (auglagfun.jl:11-26) */
    real *mu, *y, *muy, *cx, *s, *yupd, *dfx, *jtv, *tmp;
    real musqy, fx;
    int64_t n_grad;
} AL;

/* ---- Julia Base reductions, restated: pairwise sum with 1024-element leaves */
static real pairwise_sum(const real* v, int64_t n) {
#ifdef _OPENMP
    {
        real s = 0;
        _Pragma("omp parallel for schedule(static) reduction(+:s)")
        for (int64_t i = 0; i < n; ++i) s += v[i];
        return s;
    }
#endif
    if (n <= 1024) {
        real s = 0;
        for (int64_t i = 0; i < n; ++i) s += v[i];
        return s;
    }
    int64_t h = n / 2;
    return pairwise_sum(v, h) + pairwise_sum(v + h, n - h);
}
static real dot(const real* a, const real* b, int64_t n) {   /* BLAS ddot shape: 1 pass */
    real s = 0;
#ifdef _OPENMP
    _Pragma("omp parallel for schedule(static) reduction(+:s)")
#endif
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
static real norm2(const real* a, int64_t n) { return sqrt(dot(a, a, n)); }

/* ---- oracles ------------------------------------------------------------- */
static void proj_D(const AL* al, real* s, const real* v) {
    const int64_t n = al->n;
    if (al->D_kind == D_ZERO) { PFOR for (int64_t i = 0; i < n; ++i) s[i] = 0; }
    else if (al->D_kind == D_FREE) { PFOR for (int64_t i = 0; i < n; ++i) s[i] = v[i]; }
    else { PFOR for (int64_t i = 0; i < n; ++i) s[i] = v[i] < al->D_lo ? al->D_lo : (v[i] > al->D_hi ? al->D_hi : v[i]); }
}
static real f_gradient(const AL* al, real* dfx, const real* x) {
    const int64_t n = al->n;
    real fx = 0;
    if (!al->q) { PFOR for (int64_t i = 0; i < n; ++i) dfx[i] = 0; return 0; }
    real* t = al->tmp;
    PFOR for (int64_t i = 0; i < n; ++i) { real qx = al->q[i] * x[i]; dfx[i] = qx - al->b[i]; t[i] = x[i] * (0.5 * qx - al->b[i]); }
    fx = pairwise_sum(t, n);
    return fx;
}
static real g_prox(const AL* al, real* z, const real* x, real gamma) {
    const int64_t n = al->n;
    const real gl = gamma * al->lam;
    real acc = 0;
    switch (al->g_kind) {
    case G_L1:
        PRED_ACC for (int64_t i = 0; i < n; ++i) {
            z[i] = x[i] + (x[i] <= -gl ? gl : (x[i] >= gl ? -gl : -x[i]));
            acc += z[i] > 0 ? z[i] : -z[i];
        }
        return al->lam * acc;
    case G_L1_NONNEG:
        PRED_ACC for (int64_t i = 0; i < n; ++i) { if (x[i] >= gl) { z[i] = x[i] - gl; acc += z[i]; } else z[i] = 0; }
        return al->lam * acc;
    case G_L1_BOX:
        PRED_ACC for (int64_t i = 0; i < n; ++i) { real a = x[i] - gl; a = a < al->g_u[i] ? a : al->g_u[i]; z[i] = a > 0 ? a : 0; acc += z[i]; }
        return al->lam * acc;
    case G_IND_BOX:
        PFOR for (int64_t i = 0; i < n; ++i) z[i] = x[i] < al->g_lo ? al->g_lo : (x[i] > al->g_hi ? al->g_hi : x[i]);
        return 0;
    default:
        PFOR for (int64_t i = 0; i < n; ++i) z[i] = x[i];
        return 0;
    }
}

/* gradient!(dlx, al, x): one loop per broadcast statement (auglagfun.jl:73-86) */
static real al_gradient(AL* al, real* dlx, const real* x) {
    const int64_t n = al->n;
    al->n_grad++;
    PFOR for (int64_t i = 0; i < n; ++i) al->cx[i] = x[i];                       /* eval!(cx, c, x)          */
    PFOR for (int64_t i = 0; i < n; ++i) al->yupd[i] = al->cx[i] + al->muy[i];   /* yupd .= cx .+ muy        */
    proj_D(al, al->s, al->yupd);                                            /* proj!(s, D, yupd)        */
    PFOR for (int64_t i = 0; i < n; ++i) al->yupd[i] -= al->s[i];                /* yupd .-= s               */
    PFOR for (int64_t i = 0; i < n; ++i) al->tmp[i] = (al->yupd[i] * al->yupd[i]) / al->mu[i];  /* temp  */
    real lx = 0.5 * pairwise_sum(al->tmp, n);                               /* 0.5*sum(...)             */
    PFOR for (int64_t i = 0; i < n; ++i) al->yupd[i] /= al->mu[i];               /* yupd ./= mu              */
    al->fx = f_gradient(al, al->dfx, x);                                    /* gradient!(dfx, f, x)     */
    lx += al->fx;
    lx -= al->musqy;
    PFOR for (int64_t i = 0; i < n; ++i) al->jtv[i] = al->yupd[i];               /* jtprod!(jtv, c, x, yupd) */
    PFOR for (int64_t i = 0; i < n; ++i) dlx[i] = al->dfx[i] + al->jtv[i];       /* dlx .= dfx .+ jtv        */
    return lx;
}

/* ---- L-BFGS operator (ProximalAlgorithms LBFGSOperator) -------------------- */
typedef struct {
    int M, currmem, curridx;   /* curridx 1-based, 0 = empty */
    real **s_M, **y_M, *ys_M, *alphas, H;
} LBFGS;

static void lbfgs_update(LBFGS* L, const real* s, const real* y, int64_t n, real* ys_out) {
    real ys = dot(s, y, n);
    *ys_out = ys;
    if (ys > 0) {
        L->curridx += 1; if (L->curridx > L->M) L->curridx = 1;
        L->currmem += 1; if (L->currmem > L->M) L->currmem = L->M;
        L->ys_M[L->curridx - 1] = ys;
        memcpy(L->s_M[L->curridx - 1], s, n * sizeof(real));
        memcpy(L->y_M[L->curridx - 1], y, n * sizeof(real));
        real yty = dot(y, y, n);
        L->H = ys / yty;
    }
}
static void lbfgs_mul(LBFGS* L, real* d, const real* v, int64_t n) {
    memcpy(d, v, n * sizeof(real));
    int idx = L->curridx;
    for (int i = 0; i < L->currmem; ++i) {
        real a = dot(L->s_M[idx - 1], d, n) / L->ys_M[idx - 1];
        L->alphas[idx - 1] = a;
        const real* y = L->y_M[idx - 1];
        PFOR for (int64_t k = 0; k < n; ++k) d[k] -= a * y[k];
        idx -= 1; if (idx == 0) idx = L->M;
    }
    PFOR for (int64_t k = 0; k < n; ++k) d[k] = L->H * d[k];
    for (int i = 0; i < L->currmem; ++i) {
        idx += 1; if (idx > L->M) idx = 1;
        real beta = dot(L->y_M[idx - 1], d, n) / L->ys_M[idx - 1];
        real c = L->alphas[idx - 1] - beta;
        const real* s = L->s_M[idx - 1];
        PFOR for (int64_t k = 0; k < n; ++k) d[k] += c * s[k];
    }
}

static real f_model(real f_x, const real* grad, const real* res, real Lc, int64_t n) {
    real nr = norm2(res, n);
    return f_x - dot(grad, res, n) + (Lc / 2) * (nr * nr);
}

/* ---- public entry ----------------------------------------------------------
 * Runs `iters` states (initial state counts as 1) of PANOCplus with tol = 0 on the
 * AL subproblem defined by (q,b | lam.. | D | mu,y).  Outputs x, z of the last state,
 * stats[0..7] = {gamma, tau, f_x, g_z, stop_norm, n_grad, n_backtracks, n_halvings}.
 * trace (optional, iters*4 doubles): gamma, f_x, g_z, stop_norm per state.       */
int bzo_panoc_run(int64_t n, const real* q, const real* b, int g_kind, real lam, const real* g_u,
                  real g_lo, real g_hi, int D_kind, real D_lo, real D_hi, const real* mu_in,
                  const real* y_in, const real* x0, int64_t iters, int M, real minimum_gamma,
                  real* x_out, real* z_out, double* stats, double* trace) {
    const real alpha = 0.95, beta = 0.5, eps = DBL_EPSILON;
    const int max_bt = 20;
    AL al; memset(&al, 0, sizeof(al));
    al.n = n; al.q = q; al.b = b; al.g_kind = g_kind; al.lam = lam; al.g_u = g_u; al.g_lo = g_lo; al.g_hi = g_hi;
    al.D_kind = D_kind; al.D_lo = D_lo; al.D_hi = D_hi;
    const size_t nb = (size_t)n * sizeof(real);
    real** bufs[] = {&al.mu, &al.y, &al.muy, &al.cx, &al.s, &al.yupd, &al.dfx, &al.jtv, &al.tmp};
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); ++i) { *bufs[i] = (real*)malloc(nb); if (!*bufs[i]) return -1; }
    /* AugLagUpdate! (auglagfun.jl:91-101) */
    for (int64_t i = 0; i < n; ++i) { if (mu_in[i] <= 0) return -6; }
    memcpy(al.mu, mu_in, nb); memcpy(al.y, y_in, nb);
    PFOR for (int64_t i = 0; i < n; ++i) al.muy[i] = al.mu[i] * al.y[i];
    PFOR for (int64_t i = 0; i < n; ++i) al.tmp[i] = al.muy[i] * al.y[i];
    al.musqy = 0.5 * pairwise_sum(al.tmp, n);

    real *x, *gx, *yv, *z, *res, *x_prev, *res_prev, *d, *x_d, *gxd, *z_curr, *gz, *t1, *t2;
    real** sb[] = {&x, &gx, &yv, &z, &res, &x_prev, &res_prev, &d, &x_d, &gxd, &z_curr, &gz, &t1, &t2};
    for (size_t i = 0; i < sizeof(sb) / sizeof(sb[0]); ++i) { *sb[i] = (real*)malloc(nb); if (!*sb[i]) return -1; }
    LBFGS L; L.M = M; L.currmem = 0; L.curridx = 0; L.H = 1;
    L.s_M = (real**)malloc(M * sizeof(real*)); L.y_M = (real**)malloc(M * sizeof(real*));
    L.ys_M = (real*)calloc(M, sizeof(real)); L.alphas = (real*)calloc(M, sizeof(real));
    for (int i = 0; i < M; ++i) { L.s_M[i] = (real*)calloc(n, sizeof(real)); L.y_M[i] = (real*)calloc(n, sizeof(real)); }

    /* Base.iterate(iter) */
    memcpy(x, x0, nb);
    real f_x = al_gradient(&al, gx, x);
    PFOR for (int64_t i = 0; i < n; ++i) t1[i] = x[i] + 1;                 /* xeps = x .+ 1 */
    al_gradient(&al, t2, t1);
    PFOR for (int64_t i = 0; i < n; ++i) t2[i] = t2[i] - gx[i];
    real nrm_g = norm2(t2, n);
    PFOR for (int64_t i = 0; i < n; ++i) t2[i] = t1[i] - x[i];
    real gamma = alpha / (nrm_g / norm2(t2, n));
    PFOR for (int64_t i = 0; i < n; ++i) yv[i] = x[i] - gamma * gx[i];
    real g_z = g_prox(&al, z, yv, gamma);
    PFOR for (int64_t i = 0; i < n; ++i) res[i] = x[i] - z[i];
    int64_t n_bt = 0, n_halv = 0;
    real tau = 0;
    real f_z;
    {   /* backtrack_stepsize! */
        real f_z_upp = f_model(f_x, gx, res, alpha / gamma, n);
        f_z = al_gradient(&al, gz, z);
        real tol = 10 * eps * (1 + fabs(f_z));
        while (f_z > f_z_upp + tol && gamma >= minimum_gamma) {
            gamma /= 2; n_halv++;
            PFOR for (int64_t i = 0; i < n; ++i) yv[i] = x[i] - gamma * gx[i];
            g_z = g_prox(&al, z, yv, gamma);
            PFOR for (int64_t i = 0; i < n; ++i) res[i] = x[i] - z[i];
            f_z_upp = f_model(f_x, gx, res, alpha / gamma, n);
            f_z = al_gradient(&al, gz, z);
            tol = 10 * eps * (1 + fabs(f_z));
        }
    }
    real stop = 0;
    for (int64_t k = 1;; ++k) {
        /* default_stopping_criterion: norm(res/gamma - gx + gz, Inf)  (allocating broadcast) */
        PFOR for (int64_t i = 0; i < n; ++i) t1[i] = res[i] / gamma - gx[i] + gz[i];
        stop = 0;
#ifdef _OPENMP
        _Pragma("omp parallel for schedule(static) reduction(max:stop)")
#endif
        for (int64_t i = 0; i < n; ++i) { real a = fabs(t1[i]); if (a > stop || a != a) stop = a; }
        if (trace) { trace[4 * (k - 1) + 0] = gamma; trace[4 * (k - 1) + 1] = f_x; trace[4 * (k - 1) + 2] = g_z; trace[4 * (k - 1) + 3] = stop; }
        if (k >= iters) break;
        /* Base.iterate(iter, state) */
        memcpy(x_prev, x, nb); memcpy(res_prev, res, nb);
        real FBE_x = f_model(f_x, gx, res, alpha / gamma, n) + g_z;
        PFOR for (int64_t i = 0; i < n; ++i) t1[i] = -res[i];              /* -state.res (allocates) */
        lbfgs_mul(&L, d, t1, n);
        tau = 1;
        memcpy(t2, d, nb);                                            /* mul!(Ad, I, d)         */
        PFOR for (int64_t i = 0; i < n; ++i) x_d[i] = x[i] + d[i];
        real f_xd = al_gradient(&al, gxd, x_d);
        memcpy(x, x_d, nb); memcpy(gx, gxd, nb); f_x = f_xd;
        memcpy(z_curr, z, nb);
        real sigma = beta * (0.5 / gamma) * (1 - alpha);
        real tol = 10 * eps * (1 + fabs(FBE_x));
        real nr = norm2(res, n);
        real threshold = FBE_x - sigma * (nr * nr) + tol;
        for (int kk = 1; kk <= max_bt; ++kk) {
            PFOR for (int64_t i = 0; i < n; ++i) yv[i] = x[i] - gamma * gx[i];
            g_z = g_prox(&al, z, yv, gamma);
            PFOR for (int64_t i = 0; i < n; ++i) res[i] = x[i] - z[i];
            real f_z_upp = f_model(f_x, gx, res, alpha / gamma, n);
            f_z = al_gradient(&al, gz, z);
            tol = 10 * eps * (1 + fabs(f_z));
            if (f_z > f_z_upp + tol && gamma >= minimum_gamma) {
                gamma *= 0.5; n_halv++; sigma *= 2;
                L.currmem = 0; L.curridx = 0; L.H = 1;
                continue;
            }
            real FBE_new = f_z_upp + g_z;
            if (FBE_new <= threshold || kk >= max_bt) break;
            tau = (kk >= max_bt - 1) ? 0 : tau / 2; n_bt++;
            PFOR for (int64_t i = 0; i < n; ++i) x[i] = tau * x_d[i] + (1 - tau) * z_curr[i];
            f_x = al_gradient(&al, gx, x);
        }
        PFOR for (int64_t i = 0; i < n; ++i) t1[i] = x[i] - x_prev[i];
        PFOR for (int64_t i = 0; i < n; ++i) t2[i] = res[i] - res_prev[i];
        real ys;
        lbfgs_update(&L, t1, t2, n, &ys);
    }
    memcpy(x_out, x, nb); memcpy(z_out, z, nb);
    stats[0] = gamma; stats[1] = tau; stats[2] = f_x; stats[3] = g_z; stats[4] = stop;
    stats[5] = (double)al.n_grad; stats[6] = (double)n_bt; stats[7] = (double)n_halv;
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); ++i) free(*bufs[i]);
    for (size_t i = 0; i < sizeof(sb) / sizeof(sb[0]); ++i) free(*sb[i]);
    for (int i = 0; i < M; ++i) { free(L.s_M[i]); free(L.y_M[i]); }
    free(L.s_M); free(L.y_M); free(L.ys_M); free(L.alphas);
    return 0;
}
