/* panoc_capi.c — the drop-in boundary used from plain C (no Python, no Julia): builds BASELINE config 2
 * (l1-regularised diagonal quadratic, soft-threshold prox, c = Identity, D = Box[-1,1]) with the same
 * splitmix64 data bench.py uses, runs ALPS through bz_alps_solve and times PANOCplus iterations through
 * bz_panoc_begin / bz_panoc_steps.  SURVEY §8(b): "who calls it: Julia shim via ccall; Python ctypes harness;
 * C++ bench binary".
 *
 *   cc -O2 -Iinclude examples/panoc_capi.c -Lbazinga.jl_amd -lbazinga_hip -Wl,-rpath,$PWD/bazinga.jl_amd -lm -o panoc_capi
 *   ./panoc_capi [n] [steps] [lbfgs_compact: 0 two-loop | 1 compact | 2 auto (default)]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bazinga_hip.h"

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* u_k(i) = splitmix64(seed + k*2^60 + i) * 2^-64, seed = 20241004    */
static double uniform(int k, uint64_t i) {
    const uint64_t seed = 20241004ull;
    return (double)(splitmix64(seed + ((uint64_t)k << 60) + i) >> 11) * (1.0 / 9007199254740992.0);
}
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != BZ_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bz_last_error());   \
            return 1;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : 1000000;
    const int64_t steps = argc > 2 ? atoll(argv[2]) : 200;
    const int compact = argc > 3 ? atoi(argv[3]) : 2;      /* bz_panoc_opts.lbfgs_compact: 0 two-loop, 1 compact, 2 auto */
    double *q = malloc(n * sizeof(double)), *b = malloc(n * sizeof(double)), *mu = malloc(n * sizeof(double));
    double *y = calloc(n, sizeof(double)), *x0 = calloc(n, sizeof(double)), *x = malloc(n * sizeof(double));
    double *s = malloc(n * sizeof(double)), *yo = malloc(n * sizeof(double)), *muo = malloc(n * sizeof(double));
    if (!q || !b || !mu || !y || !x0 || !x || !s || !yo || !muo) return 2;
    for (int64_t i = 0; i < n; ++i) {
        q[i] = 0.1 + 9.9 * uniform(1, (uint64_t)i);
        b[i] = 10.0 * (2.0 * uniform(2, (uint64_t)i) - 1.0);
        mu[i] = 0.1;
    }
    bz_ctx_opts co;
    memset(&co, 0, sizeof(co));
    co.nranks = 1;
    bz_ctx* ctx = NULL;
    CHECK(bz_ctx_create(&co, &ctx));
    char name[256];
    int32_t cus = 0;
    int64_t mem = 0;
    CHECK(bz_device_info(ctx, name, &cus, &mem));
    printf("%s on %s (%d CUs, %.0f GB)\n", bz_version(), name, cus, (double)mem / 1e9);

    bz_problem_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = BZ_F64; d.f_kind = BZ_F_DIAG_QUADRATIC; d.g_kind = BZ_G_NORM_L1; d.c_kind = BZ_C_IDENTITY;
    d.D_kind = BZ_D_BOX; d.n = n; d.ny = n; d.f_q = q; d.f_b = b; d.g_lambda = 2.5; d.D_lo = -1.0; d.D_hi = 1.0;
    bz_problem* p = NULL;
    CHECK(bz_problem_create(ctx, &d, &p));

    /* the whole outer loop on the device (alps.jl:7-117) */
    bz_alps_opts ao;
    bz_panoc_opts po;
    bz_alps_default_opts(&ao, BZ_F64);
    bz_panoc_default_opts(&po);
    po.lbfgs_compact = compact;
    bz_alps_stats as;
    double t0 = now_s();
    CHECK(bz_alps_solve(p, &ao, &po, x0, y, x, yo, s, muo, &as));
    double t1 = now_s();
    double viol = 0.0;
    for (int64_t i = 0; i < n; ++i) viol = fmax(viol, fabs(x[i]) - 1.0);
    printf("alps: status %d  outer %lld  inner %lld  %.3f s  max box violation %.2e\n", as.status,
           (long long)as.tot_it, (long long)as.tot_inner_it, t1 - t0, viol);

    /* PANOCplus iterations per second on the first AL subproblem (mu = 0.1, y = 0, tol = 0) */
    CHECK(bz_problem_set_multipliers(p, mu, y));
    po.tol = 0.0; po.maxit = 1000000000000ll; po.minimum_gamma = 2.220446049250313e-16;
    CHECK(bz_panoc_begin(p, &po, x0));
    CHECK(bz_panoc_steps(p, 20));
    CHECK(bz_ctx_synchronize(ctx));
    t0 = now_s();
    CHECK(bz_panoc_steps(p, steps));
    CHECK(bz_ctx_synchronize(ctx));
    t1 = now_s();
    double sc[16];
    CHECK(bz_panoc_scalars(p, sc));
    printf("panoc: n=%lld %s L-BFGS  %.1f iterations/s  (%.1f us/iteration)  k=%.0f gamma=%.6g stop_norm=%.3e\n",
           (long long)n, compact == 0 ? "two-loop" : compact == 1 ? "compact" : "default (auto)", (double)steps / (t1 - t0), 1e6 * (t1 - t0) / (double)steps,
           sc[0], sc[1], sc[7]);
    bz_problem_destroy(p);
    bz_ctx_destroy(ctx);
    free(q); free(b); free(mu); free(y); free(x0); free(x); free(s); free(yo); free(muo);
    return as.status == 0 && viol <= 1e-5 ? 0 : 3;
}
