/* bazinga_hip.h — C ABI of libbazinga_hip.so
 *
 * MI355X (gfx950) implementation of the PANOCplus inner proximal-gradient solve
 * that Bazinga.alps runs on every augmented-Lagrangian subproblem.
 *
 * The reference (aldma/Bazinga.jl) has no FFI: its seam is the Julia-level
 * `subsolver` keyword of `alps` (src/algorithms/alps.jl:24,64-66) plus the
 * duck-typed f/g/c/D oracle protocol (README.md:17-20, src/Bazinga.jl:11-16).
 * Each entry point below cites the reference interface it replaces; the Julia
 * `ccall` stubs a maintainer would add are in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; all vectors are dense, contiguous, of the
 *    problem's dtype (double for BZ_F64, float for BZ_F32);
 *  - input/output pointers may be host OR device pointers (copies use
 *    hipMemcpyDefault); the library copies inputs into its own HBM buffers and
 *    never writes through an input pointer (x0/y0 are never mutated — pinned by
 *    test/problems/test_nonconvex_qp.jl:36);
 *  - every function returns BZ_OK (0) or a negative error code;
 *    bz_last_error() gives the message of the calling thread's last failure;
 *  - handles are not thread-safe; distinct handles are independent;
 *  - all calls block until their results are on the host.
 */
#ifndef BAZINGA_HIP_H
#define BAZINGA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------ */
#define BZ_OK                 0
#define BZ_ERR_ARG           -1   /* bad argument / size / kind combination           */
#define BZ_ERR_HIP           -2   /* HIP runtime failure (no device, OOM, launch)     */
#define BZ_ERR_UNSUPPORTED   -3   /* oracle kind not lowered to the device            */
#define BZ_ERR_STATE         -4   /* call order (e.g. step before begin)              */
#define BZ_ERR_COMM          -5   /* RCCL failure                                     */
#define BZ_ERR_MU            -6   /* "parameters `mu` must be positive"
                                     (src/utilities/auglagfun.jl:33-34,92-93)        */
#define BZ_ERR_CALLBACK      -7   /* a generic-oracle callback failed (bz_callback_abort) */

/* ---- numeric type ----------------------------------------------------- */
#define BZ_F64 0
#define BZ_F32 1

/* ---- oracle kinds (the structured types the Julia shim pattern-matches) */
/* f: smooth cost.  gradient!(dfx,f,x)->fx, f(x)  (src/Bazinga.jl:16)             */
#define BZ_F_ZERO            0   /* src/proxoperators/zero.jl:13-20                   */
#define BZ_F_DIAG_QUADRATIC  1   /* sum x_i(0.5 q_i x_i - b_i): diagonal case of
                                    ProximalOperators.Quadratic (test_nonconvex_qp.jl:14) */
#define BZ_F_STENCIL5        2   /* 0.5 x'A_h x - b'x, A_h = 5-pt Laplacian, Dirichlet */
#define BZ_F_LEAST_SQUARES   3   /* 0.5||A x - b||^2, ProximalOperators.LeastSquares
                                    (test/problems/test_verbose.jl:22)                 */
#define BZ_F_QUADRATIC       4   /* 0.5 x'Qx + q'x, dense symmetric Q, ProximalOperators.
                                    Quadratic (test/problems/test_nonconvex_qp.jl:14)  */
/* g: proximable cost.  prox!(z,g,x,gamma)->g(z)                                     */
#define BZ_G_ZERO            0   /* zero.jl:22-25, ProximalOperators.Zero / IndFree   */
#define BZ_G_NORM_L1         1   /* ProximalOperators.NormL1(lambda) (test_verbose.jl:23) */
#define BZ_G_NORM_L1_NONNEG  2   /* src/proxoperators/normL1Nonneg.jl:29-42           */
#define BZ_G_NORM_L1_BOX     3   /* src/proxoperators/normL1Box.jl:30-39              */
#define BZ_G_IND_BOX         4   /* ProximalOperators.IndBox (test_nonconvex_qp.jl:15)*/
#define BZ_G_NORM_L0_BOX     5   /* src/proxoperators/normL0Box.jl:33-58              */
#define BZ_G_NORM_LP_NONNEG  6   /* alpha*sum x^p, x >= 0: normLpNonneg.jl:14-90      */
#define BZ_G_NORM_LP_BOX     7   /* alpha*sum x^p, 0 <= x <= u: normLpBox.jl:11-97    */
/* c: constraint map.  eval!(cx,c,x), jtprod!(jtv,c,x,v)                             */
#define BZ_C_IDENTITY        0   /* test/definitions/identityFunction.jl:3-13         */
#define BZ_C_DENSE_AFFINE    1   /* A x - b, demo/basispursuit.jl:38-49               */
/* D: closed set.  proj!(s,D,v)                                                      */
#define BZ_D_ZERO            0   /* src/projections/zeroSet.jl:17-20                  */
#define BZ_D_FREE            1   /* src/projections/freeSet.jl:17-20                  */
#define BZ_D_BOX             2   /* ClosedSet(IndBox(lo,hi)), indicatorSet.jl:8-11    */
/* pairwise sets over ADJACENT pairs (cx[2j], cx[2j+1]), ny even — the layout of demo/mpvca.jl:105-106,147-148
 * and demo/eitheror.jl:79-88,123-130; c = Identity, no slack.  The 2-element projections they apply:      */
#define BZ_D_VC_PAIRS        3   /* project_onto_VC_set!       src/projections/vanishingConstraints.jl:27-46   */
#define BZ_D_CC_PAIRS        4   /* project_onto_CC_set!       src/projections/complementarityConstraints.jl:8-20 */
#define BZ_D_EITHEROR_PAIRS  5   /* project_onto_EITHEROR_set! src/projections/orConstraints.jl:7-17           */
#define BZ_D_XOR_PAIRS       6   /* project_onto_XOR_set!      src/projections/orConstraints.jl:24-36          */

/* Generic (user-defined) oracles — whatever the structured kinds above do not cover, e.g. the closures of
 * demo/rosenbrock.jl:39-80 (BASELINE config 1): the four oracles are HOST CALLBACKS with the reference's own
 * protocol (README.md:17-20; the generic fallbacks the package itself defines are src/Bazinga.jl:49-84).  The host
 * evaluates f/grad f, prox_g, c, J'v and proj_D and assembles the AL gradient exactly as auglagfun.jl:73-86 does;
 * the L-BFGS two-loop, the line-search vector work and every reduction stay on the device.  All four kinds must be
 * the CALLBACK kind together (a host language wraps its structured types in callbacks when it mixes them);
 * single rank, no slack form.  Every vector a callback sees is a host array of the problem's dtype.        */
#define BZ_F_CALLBACK        5
#define BZ_G_CALLBACK        8
#define BZ_C_CALLBACK        2
#define BZ_D_CALLBACK        7
/* gradient!(dfx, f, x) -> f(x)                       src/Bazinga.jl:16, demo/rosenbrock.jl:45-50          */
typedef double (*bz_f_gradient_fn)(void* user, const void* x, void* dfx, int64_t n);
/* prox!(z, g, x, gamma) -> g(z)                      src/utilities/nonsmoothcostfun.jl:17-22              */
typedef double (*bz_g_prox_fn)(void* user, const void* x, double gamma, void* z, int64_t n);
/* eval!(cx, c, x)                                    demo/rosenbrock.jl:67-70                             */
typedef void (*bz_c_eval_fn)(void* user, const void* x, void* cx, int64_t n, int64_t ny);
/* jtprod!(jtv, c, x, v)                              demo/rosenbrock.jl:71-74                             */
typedef void (*bz_c_jtprod_fn)(void* user, const void* x, const void* v, void* jtv, int64_t n, int64_t ny);
/* proj!(s, D, v)                                     demo/rosenbrock.jl:77-80                             */
typedef void (*bz_D_proj_fn)(void* user, const void* v, void* s, int64_t ny);

/* Error channel of the callbacks (r03).  A callback cannot unwind through the library's C frames: one that failed
 * (an exception in the host language — the reference's oracles `error(...)` freely, e.g. src/utilities/auglagfun.jl:33)
 * calls bz_callback_abort() before it returns whatever it has; the library call in progress on this thread then ends
 * with BZ_ERR_CALLBACK right after that callback, without evaluating anything on the buffers it left behind.  The flag
 * is per thread and is cleared by every library call that takes a bz_problem.                               */
void bz_callback_abort(void);

typedef struct bz_ctx     bz_ctx;
typedef struct bz_problem bz_problem;

/* ---- context: one per process/GPU ------------------------------------- */
typedef struct {
    int32_t device;       /* HIP device ordinal                                      */
    int32_t rank;         /* this rank, 0..nranks-1                                  */
    int32_t nranks;       /* 1 = single GPU; >1 = x sharded, scalars all-gathered    */
    int32_t flags;        /* BZ_CTX_* bits, 0 = none                                 */
    const void* comm_id;  /* 128-byte id from bz_comm_unique_id on rank 0: RCCL communicator
                             (nranks==1 + id: 1-rank communicator, for testing; NULL with
                             nranks>1: p2p mailboxes must be connected instead)        */
} bz_ctx_opts;

/* bz_ctx_opts.flags */
#define BZ_CTX_RUNTIME_TUNING 1   /* apply bz_runtime_tuning() before this context's first HIP call         */
#define BZ_CTX_SHARED_DEVICE  2   /* the GPU is not this process's own (other processes, other HIP users in this one): no
                                     launch is ever made to wait, resident, at a gate — a polling launch holds its CUs, and
                                     tenants that do that starve each other ("Who decides at the gate").  Costs the
                                     ~6 us per iteration the gated pre-launch saves.  Implied between ranks that
                                     bz_ctx_p2p_connect finds on one device.                                            */

/* ROCm runtime settings for a launch-latency-bound host loop (one short kernel chain per PANOC iteration with
 * the host in the loop): HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) and HSA_ENABLE_INTERRUPT=0
 * (polled completion signals); worth ~1.5 us per iteration.  They are PROCESS-WIDE environment settings read
 * when the HIP runtime initialises, so the library never applies them on its own: the host opts in, before the
 * process's first HIP call, either here or with BZ_CTX_RUNTIME_TUNING on its first bz_ctx_create.  Variables
 * the process has already set are left alone.  Returns a bit mask: bit 0 HIP_FORCE_DEV_KERNARG was set by
 * this call, bit 1 HSA_ENABLE_INTERRUPT was.                                                              */
int  bz_runtime_tuning(void);

int  bz_comm_unique_id(void* id128);                 /* fills 128 bytes (RCCL id)     */
int  bz_ctx_create(const bz_ctx_opts* opts, bz_ctx** out);
void bz_ctx_destroy(bz_ctx* ctx);
int  bz_ctx_synchronize(bz_ctx* ctx);                /* drains the solver stream and the device */
/* number of ranks the context's RCCL communicator spans (ncclCommCount), 0 without a communicator      */
int  bz_ctx_comm_nranks(bz_ctx* ctx, int32_t* out);
/* Peer-to-peer scalar mailboxes (single node, optional; replaces the RCCL all-gather and lets the
 * persistent two-loop kernel run sharded): every rank exports the 64-byte HIP IPC handle of its
 * mailbox, the launcher all-gathers the handles and the device ordinals, every rank connects.      */
int  bz_ctx_p2p_export(bz_ctx* ctx, void* handle64);
int  bz_ctx_p2p_connect(bz_ctx* ctx, const void* handles /* nranks*64 bytes */, const int32_t* devices);
const char* bz_last_error(void);
const char* bz_version(void);
/* fills name (<=255 chars), compute-unit count and total device memory in bytes */
int  bz_device_info(bz_ctx* ctx, char* name256, int32_t* cus, int64_t* mem_bytes);

/* ---- problem descriptor: the f/g/c/D oracles of Bazinga.alps ---------- */
/* On nranks>1 every rank passes its contiguous shard (n, ny = local sizes).
 * Scalars (lambda, lo, hi) are used when the matching *_vec pointer is NULL.        */
typedef struct {
    int32_t dtype;                 /* BZ_F64 | BZ_F32                                 */
    int32_t f_kind, g_kind, c_kind, D_kind;
    int32_t slack;                 /* 0: ALPS subproblem on x (auglagfun.jl); 1: ALS subproblem on
                                      xs = [x; s] of length n + ny (auglagfunslack.jl)      */
    int64_t n;                     /* length of x (local shard)                       */
    int64_t ny;                    /* length of y / c(x) (local shard)                */
    /* f */
    const void* f_q;               /* DIAG_QUADRATIC: q[n]                            */
    const void* f_b;               /* DIAG_QUADRATIC / STENCIL5: b[n]                 */
    int64_t     f_grid_nx, f_grid_ny; /* STENCIL5: grid rows, cols (row-major, n=nx*ny) */
    const void* f_A;               /* LEAST_SQUARES: A[f_rows][n]; QUADRATIC: Q[n][n]; row-major;
                                      f_b then holds b[f_rows] resp. q[n]                */
    int64_t     f_rows;
    /* g */
    double      g_lambda;          /* NORM_L1*, NORM_L0_BOX: lambda >= 0; NORM_LP_*: alpha */
    double      g_p;               /* NORM_LP_*: exponent p in (0,1)                  */
    const void* g_u;               /* NORM_L1_BOX / NORM_L0_BOX / NORM_LP_BOX: u[n] >= 0 */
    double      g_lo, g_hi;        /* IND_BOX scalar bounds                           */
    const void* g_lo_vec;          /* IND_BOX vector bounds (or NULL)                 */
    const void* g_hi_vec;
    /* c */
    const void* c_A;               /* DENSE_AFFINE: A[ny][n] row-major                */
    const void* c_b;               /* DENSE_AFFINE: b[ny]                             */
    /* D */
    double      D_lo, D_hi;        /* BOX scalar bounds (+-inf allowed)               */
    const void* D_lo_vec;          /* BOX vector bounds (or NULL)                     */
    const void* D_hi_vec;
    /* generic oracles (all four kinds BZ_*_CALLBACK) */
    void*            cb_user;      /* handed back to every callback                   */
    bz_f_gradient_fn cb_f_gradient;
    bz_g_prox_fn     cb_g_prox;
    bz_c_eval_fn     cb_c_eval;
    bz_c_jtprod_fn   cb_c_jtprod;
    bz_D_proj_fn     cb_D_proj;
} bz_problem_desc;

int  bz_problem_create(bz_ctx* ctx, const bz_problem_desc* desc, bz_problem** out);
void bz_problem_destroy(bz_problem* p);

/* BZ_F_STENCIL5 with x sharded over the ranks of a node (cfg 3): the grid is cut into
 * row blocks in rank order — desc.f_grid_nx is THIS rank's number of rows, desc.n = f_grid_nx * f_grid_ny —
 * and before every stencil evaluation the boundary rows are exchanged with the two neighbour ranks
 * straight through IPC-mapped device buffers (no collective).  After bz_problem_create on a context whose
 * mailboxes are connected (bz_ctx_p2p_connect): export this problem's halo region, hand the 64-byte
 * handles to the neighbours (any host channel), connect.  prev64 / next64 are ignored on the first / last
 * rank (pass NULL there).                                                                              */
int bz_problem_halo_export(bz_problem* p, void* handle64);
int bz_problem_halo_connect(bz_problem* p, const void* prev64, const void* next64);

/* BZ_C_DENSE_AFFINE with the ROWS of A sharded over the ranks of a node (cfg 4): desc.ny,
 * c_A, c_b, mu, y are this rank's rows; x (desc.n) is replicated on every rank, which does all n-vector
 * work in full; the one n-vector that couples the ranks, A' yhat, is summed in rank order through
 * IPC-mapped regions.  After bz_problem_create (mailboxes connected): export, all-gather the 64-byte
 * handles (nranks * 64 bytes, rank order), connect.                                                   */
int bz_problem_allreduce_export(bz_problem* p, void* handle64);
int bz_problem_allreduce_connect(bz_problem* p, const void* handles);

/* ---- inner solver: replaces  ProximalAlgorithms.PANOCplus(...)(f=alFun,g=gFun,x0=x)
 *      at src/algorithms/alps.jl:64-66 ------------------------------------------- */
typedef struct {
    double  tol;             /* stop when ||res/gamma - gradL(x) + gradL(z)||_inf <= tol */
    int64_t maxit;           /* IterativeAlgorithm maxit (default 1000)               */
    int32_t freq;            /* display frequency when verbose (default 10)           */
    int32_t verbose;
    double  minimum_gamma;   /* default 1e-7                                          */
    double  alpha;           /* default 0.95                                          */
    double  beta;            /* default 0.5                                           */
    int32_t max_backtracks;  /* default 20                                            */
    int32_t lbfgs_memory;    /* directions = LBFGS(M), default 5; 0 = NoAcceleration() */
    int32_t fuse;            /* 1: use the single-pass fused kernel when the problem
                                is separable (elementwise f, c = Identity); 0: always
                                the generic kernel chain.  Results are bit-identical. */
    int32_t persist;         /* 1: run the L-BFGS two-loop as ONE persistent launch with d
                                register-resident when the vector fits (<= 48 packs/thread)
                                and is long enough; 0: one kernel per two-loop step.
                                Same arithmetic, different (fixed) summation tree.       */
    int32_t lbfgs_compact;   /* how the L-BFGS operator (ProximalAlgorithms.LBFGS) is evaluated:
                                0: two-loop recursion, the reference's operation order (2M sequentially
                                   dependent reductions per application);
                                1: compact (Byrd-Nocedal-Schnabel) representation (M <= 5): no sequential
                                   reductions, so for c = Identity with an element-wise f the whole
                                   iteration is ONE streaming pass over M+4..2M+10 vectors and one reduction
                                   phase (one cross-GPU exchange) instead of 2M+1.  The same operator, an
                                   alternate rounding: its iterates track the fp64 oracle as closely as the
                                   two-loop kernels' do;
                                2 (default): 1 where that one-pass kernel applies, on the stencil and affine-image
                                   paths, and — one rank — wherever the two-loop would otherwise run as a chain of 2M
                                   kernels (a vector beyond the persistent launch's register capacity, or too short
                                   for its grid barriers); 0 (the persistent two-loop launch) elsewhere.          */
    int32_t affine_refresh;  /* affine images (c = DenseAffine, D = ZeroSet / FreeSet, f = Zero / DiagQuadratic:
                                c(.) and grad L(.) are affine maps).  k >= 1 (default 16): c(x + d) and grad L(x + d)
                                are formed from the stored images of the iterates — the linear combination that
                                forms d, applied to their images: no pass over A — and evaluated with the two passes
                                over A every k-th iteration, which bounds the rounding drift (k image steps add about 12 k eps of
                                relative error: at 16 below what one fp32 product over n = 65536 terms rounds to); with it an iteration
                                reads A twice instead of four times.  1: every evaluation passes over A (no images
                                used); 0: no image bookkeeping at all (the reference's dataflow).  Needs the compact
                                L-BFGS form (lbfgs_compact != 0, lbfgs_memory <= 5).                          */
    int32_t directions;      /* the `directions` keyword of PANOCplus (demo/rosenbrock.jl:96-103):
                                BZ_DIR_LBFGS (default): LBFGS(lbfgs_memory), lbfgs_memory = 0: NoAcceleration();
                                BZ_DIR_ANDERSON: AndersonAcceleration(lbfgs_memory), 1 <= lbfgs_memory <= 5;
                                BZ_DIR_BROYDEN:  Broyden() — a dense n-by-n operator, n <= 4096                 */
    int32_t reserved;
    double  broyden_theta_bar; /* Broyden(theta_bar = 0.2)                                                    */
    /* the step-size keywords of PANOCplus (upstream `Lf = nothing`, `gamma = Lf === nothing ? nothing :
     * alpha / Lf`, `adaptive = gamma === nothing`); what a warm-started outer loop passes down (bz_alps_opts.warm_start) */
    double  gamma;           /* 0 (default) = nothing: gamma = alpha / lower_bound_smoothness_constant(f, I, x, grad) ;
                                > 0: the initial step size, no Lipschitz estimate (one AL gradient less per solve)    */
    double  Lf;              /* 0 (default) = nothing; > 0 and gamma == 0: gamma = alpha / Lf                         */
    int32_t adaptive;        /* -1 (default): true exactly when neither gamma nor Lf is given; 1: backtrack_stepsize!
                                (gamma halvings) at the start and inside every line-search trial; 0: gamma is kept   */
    int32_t reserved2;
} bz_panoc_opts;
#define BZ_DIR_LBFGS    0
#define BZ_DIR_ANDERSON 1
#define BZ_DIR_BROYDEN  2

void bz_panoc_default_opts(bz_panoc_opts* o);

typedef struct {
    int64_t iters;            /* k returned by the solver (initial state counts as 1) */
    double  f_z;              /* f at the last point the AL functor was evaluated
                                 -> alFun.fx  (alps.jl:68)                            */
    double  g_z;              /* g at the returned z -> gFun.gz (alps.jl:68)          */
    double  al_z;             /* augmented Lagrangian value at that point             */
    double  gamma;
    double  tau;
    double  stop_norm;
    int64_t n_grad;           /* AL-gradient evaluations                              */
    int64_t n_prox;           /* forward-backward (prox) steps                        */
    int64_t n_backtracks;     /* tau halvings                                         */
    int64_t n_gamma_halvings;
    int64_t n_fused_iters;    /* iterations served by the fused fast path             */
    int64_t n_lbfgs_skips;    /* updates rejected because <s,y> <= 0                  */
    double  elapsed_s;
    int32_t status;           /* 0 converged, 1 maxit, 2 NaN encountered              */
    int32_t persist_fallbacks;/* times the persistent two-loop kernel's grid barrier timed out (its workgroups
                                 were not co-resident) and the solve went on with the kernel chain (0 or 1)   */
    int64_t n_affine_images;  /* AL gradients formed from stored images instead of two passes over A          */
    int64_t n_gated_launches; /* iterations whose one-pass kernel was launched EARLY, behind the previous read-back,
                                 and released through its gate (bz_panoc_steps / bz_panoc_solve / bz_alps_solve:
                                 wherever the library itself runs the loop); ~6 us per iteration; BZ_GATE=0 turns
                                 it off                                                                           */
    int64_t n_gate_aborts;    /* ... and early launches recalled because the iteration did not end the plain way   */
    int64_t n_gate_fallbacks; /* times an early launch gave up at its gate — the host did not release it within ~3 s, or its
                                 first workgroup was not resident in time because another tenant holds the GPU's CUs —
                                 and the iteration was redone without the gate, which then stays off for this problem
                                 (0 or 1; one rank only: with several ranks it is BZ_ERR_COMM)                        */
    int64_t n_dense_onepass;  /* (r03) AL gradients with c = DenseAffine evaluated in ONE pass over A (k_dense_fused: c(x), yhat and
                                 A'yhat from one read of every row, demo/basispursuit.jl:38-49) instead of two products         */
    int64_t n_dense_fallbacks;/* times that kernel's row groups timed out waiting for each other (workgroups not all resident:
                                 the GPU has another tenant) and the solve went on with the two-kernel form (0 or 1)           */
} bz_panoc_stats;

/* Multipliers/penalties of the current subproblem:  AugLagUpdate!(alFun, mu, y)
 * (src/utilities/auglagfun.jl:91-101).  Returns BZ_ERR_MU if any mu <= 0.           */
int bz_problem_set_multipliers(bz_problem* p, const void* mu, const void* y);

/* One-shot subsolve.  x_out receives state.z.                                      */
int bz_panoc_solve(bz_problem* p, const bz_panoc_opts* o, const void* x0,
                   void* x_out, bz_panoc_stats* stats);

/* Step-wise form of the same solve (used by parity tests and bench.py):
 *   begin  = Base.iterate(iter)           (initial state, k = 1)
 *   step   = Base.iterate(iter, state)    (k += 1)
 *   finish = copy out state.z + stats                                               */
int bz_panoc_begin(bz_problem* p, const bz_panoc_opts* o, const void* x0);
int bz_panoc_step(bz_problem* p);
/* k consecutive steps in one call (the same as calling bz_panoc_step k times: the stopping
 * criterion is the caller's to test, through bz_panoc_scalars)                          */
int bz_panoc_steps(bz_problem* p, int64_t k);
int bz_panoc_finish(bz_problem* p, void* x_out, bz_panoc_stats* stats);
/* scalars of the current state: out[0..15] =
 *  {k, gamma, tau, f_x(AL value at x), g_z, <gradL(x),res>, ||res||^2, stop_norm,
 *   last <s,y>, lbfgs currmem, lbfgs H, AL value at z, f(z), n_backtracks(last it),
 *   fused(last it), FBE(x)}                                                          */
int bz_panoc_scalars(bz_problem* p, double* out16);
/* copies a state vector to the host: which = 0:x 1:z 2:res 3:gradL(x) 4:gradL(z)
 * (3 and 4 are recomputed on demand when the fused path did not materialise them)   */
int bz_panoc_vector(bz_problem* p, int32_t which, void* out);

/* ---- outer loop with device-resident vectors: replaces Bazinga.alps
 *      (src/algorithms/alps.jl:7-117) for lowered oracle kinds ---------------------- */
typedef struct {
    double  tol_prim, tol_dual, inner_tol;
    int64_t maxit;
    double  theta_penalty, kappa_penalty, kappa_tol;
    int64_t subsolver_maxit;   /* only the threshold of alps.jl:70                    */
    int32_t verbose;
    int32_t warm_start;        /* SURVEY 8(f-1), an OPT-IN deviation from alps.jl:64 (0, the default, is the reference):
                                  bit 0: from the second subproblem on the subsolver starts at the step size gamma the
                                  previous subproblem ended with (`subsolver(tol, verbose; gamma = gamma_prev, adaptive =
                                  true)`) instead of estimating a Lipschitz constant again: one AL gradient (two passes
                                  over a dense A) and the first halvings less per subproblem.  gamma never grows inside
                                  PANOCplus, so after the penalties shrink it is at most too small, never unsafe.      */
} bz_alps_opts;

void bz_alps_default_opts(bz_alps_opts* o, int32_t dtype);

typedef struct {
    int64_t tot_it, tot_inner_it;
    double  elapsed_s;
    int32_t status;            /* 0 :first_order 1 :max_iter 2 :exception 3 :unknown  */
    int32_t reserved;
    double  inner_tol, norm_res_prim, objective;
} bz_alps_stats;

/* x[n], y[ny], s[ny], mu[ny] are outputs (alps.jl:115); x0[n], y0[ny] inputs.       */
int bz_alps_solve(bz_problem* p, const bz_alps_opts* ao, const bz_panoc_opts* po,
                  const void* x0, const void* y0,
                  void* x, void* y, void* s, void* mu, bz_alps_stats* stats);

/* Bazinga.als (src/algorithms/als.jl:7-120), the slack-variable sibling: same arguments and outputs;
 * the problem must have been created with desc.slack = 1.                                            */
int bz_als_solve(bz_problem* p, const bz_alps_opts* ao, const bz_panoc_opts* po,
                 const void* x0, const void* y0,
                 void* x, void* y, void* s, void* mu, bz_alps_stats* stats);

/* ---- single oracle evaluations on the device (kernel-level parity tests) ---- */
/* gradient!(dlx, al, x) -> lx   (auglagfun.jl:73-86).  vals = {lx, fx, 0.5*sum t^2/mu} */
int bz_eval_al_gradient(bz_problem* p, const void* x, void* dlx, double* vals3);
/* prox!(z, gFun, x, gamma) -> g(z)   (nonsmoothcostfun.jl:17-22)                    */
int bz_eval_prox(bz_problem* p, const void* x, double gamma, void* z, double* gz);
/* d = H v with the L-BFGS two-loop over m stored pairs (newest last);
 * S, Y are m*n host arrays, ys[m] = <s_i,y_i>, H0 scalar.                           */
int bz_eval_lbfgs(bz_problem* p, int32_t m, const void* S, const void* Y,
                  const void* v, void* d);

/* ---- in-library kernel timing (HIP events on the solver's stream) ----------- */
/* category: 0 k_axpy_dot (two-loop step), 1 k_fused_sep (fused separable iteration),
 *           2 AL gradient, 3 forward-backward step, 4 L-BFGS update/stop norm,
 *           5 scalar collect, 6 pack + all-gather, 7 misc, 8 k_dot (first two-loop dot),
 *           9 dense GEMV kernels (vector ALU), 10 k_twoloop_persist, 11 k_gemv_t_mfma,
 *           12 k_fused_compact in its steady-state form (history kept as iterates: reads the M+1 last
 *              iterates and the problem data, writes x_d; the other forms of that kernel count under 1),
 *           13 k_stencil_fb, 14 k_stencil_update (the two passes of a stencil-f iteration), 15 formation of x_d on
 *              the unfused paths (k_compact_xd / the two-loop's last axpy).
 * mask: bits 0..15: bit c enables timing of category c (0 = off); bits 16..31: sampling period k
 *       (0/1 = every launch, k = every k-th launch of each enabled category).          */
#define BZ_NUM_KERNEL_CATEGORIES 16
int bz_profile_enable(bz_problem* p, int32_t mask);
int bz_profile_get(bz_problem* p, int32_t category, int64_t* launches, double* total_ms);
int bz_profile_reset(bz_problem* p);
/* The same with the bytes each launch is designed to move (its read + write streams x their length x
 * sizeof(T): what the implemented dataflow must move, not the reference's): over the launches that carried
 * timing events (timed_*: a sustained rate is timed_bytes / timed_ms) and over EVERY launch of the category
 * since bz_profile_reset (launches, bytes: moved bytes per iteration).  form: the template form of the
 * category's last launch (e.g. "k_fused_compact<XR=2,UNI=2,NT=1>"), so that a hardware-counter profile can
 * be matched to the instantiation that actually ran.                                                      */
typedef struct {
    int64_t timed_launches;
    double  timed_ms;
    double  timed_bytes;
    int64_t launches;
    double  bytes;
    char    form[96];
} bz_profile_rec;
int bz_profile_get2(bz_problem* p, int32_t category, bz_profile_rec* out);

#ifdef __cplusplus
}
#endif
#endif /* BAZINGA_HIP_H */
