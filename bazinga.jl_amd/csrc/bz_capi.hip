// bz_capi.hip — the extern "C" surface declared in include/bazinga_hip.h.
#include <cstdlib>
#include <cstring>
#include <string>

#include "bz_solver.h"

struct bz_ctx {
    bz::Ctx c;
};
struct bz_problem {
    bz_ctx* ctx;
    bz::SolverBase* s;
    int dtype;
    int device;       // kept here so destroy never dereferences a context that was freed first
};

namespace {
thread_local std::string g_err;

// ROCm runtime settings for a launch-latency-bound host loop (one short kernel chain per PANOC iteration,
// the host in the loop between them):
//   HIP_FORCE_DEV_KERNARG=1  kernel arguments live in device memory: the command processor does not fetch
//                            them over PCIe at every launch
//   HSA_ENABLE_INTERRUPT=0   completion signals are polled instead of interrupt-driven
// measured together, same box, alternating runs at n = 1.25e6: 53.7 -> 52.0 us per iteration on average and a
// narrower spread (52.2-56.3 -> 51.3-52.4); box-to-box differences are larger than that.  They change the
// behaviour of every HIP user in the process, so they are applied only when the host asks (bz_runtime_tuning /
// BZ_CTX_RUNTIME_TUNING), never from a library constructor.
int apply_runtime_tuning() {
    int mask = 0;
    if (!std::getenv("HIP_FORCE_DEV_KERNARG")) { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); mask |= 1; }
    if (!std::getenv("HSA_ENABLE_INTERRUPT")) { setenv("HSA_ENABLE_INTERRUPT", "0", 0); mask |= 2; }
    return mask;
}

template <class F> int guard(F&& f) {
    try {
        f();
        return BZ_OK;
    } catch (const bz::Error& e) {
        g_err = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_err = e.what();
        return BZ_ERR_ARG;
    } catch (...) {      // nothing may cross the extern "C" boundary
        g_err = "unknown C++ exception";
        return BZ_ERR_HIP;
    }
}
// HIP's current device is per host thread: every entry point that takes a handle selects the handle's device
// first (a caller on another thread — a migrated Julia task, a Python worker — would otherwise allocate and
// launch on device 0)
void on_device(int device) {
    bz::callback_abort_flag() = false;      // (a stale request from outside any library call)
    BZ_HIP(hipSetDevice(device));
}
void need(const void* p, const char* what) {
    if (!p) throw bz::Error(BZ_ERR_ARG, std::string("null argument: ") + what);
}
}  // namespace

extern "C" {

const char* bz_last_error(void) { return g_err.c_str(); }
const char* bz_version(void) { return "bazinga-hip 0.2 (gfx950)"; }
int bz_runtime_tuning(void) { return apply_runtime_tuning(); }
void bz_callback_abort(void) { bz::callback_abort_flag() = true; }

int bz_comm_unique_id(void* id128) {
    return guard([&] {
        need(id128, "id128");
        static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
        ncclUniqueId id;
        BZ_NCCL(ncclGetUniqueId(&id));
        std::memcpy(id128, &id, sizeof(id));
    });
}

int bz_ctx_create(const bz_ctx_opts* o, bz_ctx** out) {
    return guard([&] {
        need(o, "opts"); need(out, "out");
        if (o->nranks < 1 || o->rank < 0 || o->rank >= o->nranks)
            throw bz::Error(BZ_ERR_ARG, "rank/nranks out of range");
        if (o->flags & ~(BZ_CTX_RUNTIME_TUNING | BZ_CTX_SHARED_DEVICE)) throw bz::Error(BZ_ERR_ARG, "unknown bz_ctx_opts.flags bit");
        if (o->flags & BZ_CTX_RUNTIME_TUNING) (void)apply_runtime_tuning();      // before the first HIP call below
        int ndev = 0;
        BZ_HIP(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw bz::Error(BZ_ERR_HIP, "no HIP device visible");
        if (o->device < 0 || o->device >= ndev) throw bz::Error(BZ_ERR_ARG, "device ordinal out of range");
        auto* c = new bz_ctx();
        try {
            c->c.device = o->device; c->c.rank = o->rank; c->c.nranks = o->nranks;
            c->c.shared_device = (o->flags & BZ_CTX_SHARED_DEVICE) != 0;
            BZ_HIP(hipSetDevice(o->device));
            BZ_HIP(hipStreamCreateWithFlags(&c->c.stream, hipStreamNonBlocking));
            // nranks == 1 with a comm_id builds a 1-rank communicator: exercises the all-gather
            // plumbing on a single GPU (tests)
            // nranks > 1 without a comm_id: the caller will connect the p2p mailboxes instead of RCCL
            if (o->comm_id) {
                ncclUniqueId id;
                std::memcpy(&id, o->comm_id, sizeof(id));
                BZ_NCCL(ncclCommInitRank(&c->c.comm, o->nranks, id, o->rank));
            }
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
    });
}

void bz_ctx_destroy(bz_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->c.device);
    delete ctx;
}

int bz_ctx_p2p_export(bz_ctx* ctx, void* handle64) {
    return guard([&] { need(ctx, "ctx"); need(handle64, "handle64"); on_device(ctx->c.device); bz::p2p_export(&ctx->c, handle64); });
}
int bz_ctx_p2p_connect(bz_ctx* ctx, const void* handles, const int32_t* devices) {
    return guard([&] { need(ctx, "ctx"); need(handles, "handles"); on_device(ctx->c.device); bz::p2p_connect(&ctx->c, handles, devices); });
}

int bz_ctx_synchronize(bz_ctx* ctx) {
    return guard([&] {
        need(ctx, "ctx");
        BZ_HIP(hipSetDevice(ctx->c.device));
        BZ_HIP(hipStreamSynchronize(ctx->c.stream));
        BZ_HIP(hipDeviceSynchronize());
    });
}

int bz_ctx_comm_nranks(bz_ctx* ctx, int32_t* out) {
    return guard([&] {
        need(ctx, "ctx"); need(out, "out");
        int cnt = 0;
        if (ctx->c.comm) BZ_NCCL(ncclCommCount(ctx->c.comm, &cnt));
        *out = cnt;
    });
}

int bz_device_info(bz_ctx* ctx, char* name256, int32_t* cus, int64_t* mem_bytes) {
    return guard([&] {
        need(ctx, "ctx");
        hipDeviceProp_t p;
        BZ_HIP(hipGetDeviceProperties(&p, ctx->c.device));
        if (name256) { std::strncpy(name256, p.gcnArchName, 255); name256[255] = 0; }
        if (cus) *cus = p.multiProcessorCount;
        if (mem_bytes) *mem_bytes = (int64_t)p.totalGlobalMem;
    });
}

int bz_problem_create(bz_ctx* ctx, const bz_problem_desc* d, bz_problem** out) {
    return guard([&] {
        need(ctx, "ctx"); need(d, "desc"); need(out, "out");
        BZ_HIP(hipSetDevice(ctx->c.device));
        auto* p = new bz_problem{ctx, nullptr, d->dtype, ctx->c.device};
        try {
            p->s = bz::make_solver(&ctx->c, *d);
        } catch (...) {
            delete p;
            throw;
        }
        *out = p;
    });
}

void bz_problem_destroy(bz_problem* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    delete p->s;
    delete p;
}

void bz_panoc_default_opts(bz_panoc_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->tol = 1e-8; o->maxit = 1000; o->freq = 10; o->verbose = 0;
    o->minimum_gamma = 1e-7; o->alpha = 0.95; o->beta = 0.5;
    o->max_backtracks = 20; o->lbfgs_memory = 5; o->fuse = 1; o->persist = 1; o->lbfgs_compact = 2;
    o->affine_refresh = 16; o->directions = BZ_DIR_LBFGS; o->broyden_theta_bar = 0.2;
    o->gamma = 0.0; o->Lf = 0.0; o->adaptive = -1; o->reserved2 = 0;
}

void bz_alps_default_opts(bz_alps_opts* o, int32_t dtype) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    const double tol = dtype == BZ_F32 ? (double)1e-6f : 1e-6;   // eltype(x0)(1e-6), alps.jl:14
    o->tol_prim = tol; o->tol_dual = tol; o->inner_tol = std::cbrt(tol);
    o->maxit = 100; o->theta_penalty = 0.8; o->kappa_penalty = 0.5; o->kappa_tol = 0.1;
    o->subsolver_maxit = 1000000000LL; o->verbose = 0;
}

int bz_problem_set_multipliers(bz_problem* p, const void* mu, const void* y) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(mu, "mu"); need(y, "y"); p->s->set_multipliers(mu, y); });
}

int bz_panoc_solve(bz_problem* p, const bz_panoc_opts* o, const void* x0, void* x_out,
                   bz_panoc_stats* st) {
    return guard([&] {
        need(p, "problem"); on_device(p->device); need(o, "opts"); need(x0, "x0"); need(x_out, "x_out");
        p->s->solve(*o, x0, x_out, st);
    });
}

int bz_panoc_begin(bz_problem* p, const bz_panoc_opts* o, const void* x0) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(o, "opts"); need(x0, "x0"); p->s->begin(*o, x0); });
}
int bz_panoc_step(bz_problem* p) {
    return guard([&] { need(p, "problem"); on_device(p->device); p->s->step(); });
}
int bz_panoc_steps(bz_problem* p, int64_t k) {
    return guard([&] {
        need(p, "problem"); on_device(p->device);
        if (k < 0) throw bz::Error(BZ_ERR_ARG, "k must be nonnegative");
        p->s->steps(k);
    });
}
int bz_panoc_finish(bz_problem* p, void* x_out, bz_panoc_stats* st) {
    return guard([&] { need(p, "problem"); on_device(p->device); p->s->finish(x_out, st); });
}
int bz_panoc_scalars(bz_problem* p, double* out16) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(out16, "out16"); p->s->scalars(out16); });
}
int bz_panoc_vector(bz_problem* p, int32_t which, void* out) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(out, "out"); p->s->vector(which, out); });
}

int bz_alps_solve(bz_problem* p, const bz_alps_opts* ao, const bz_panoc_opts* po, const void* x0,
                  const void* y0, void* x, void* y, void* s, void* mu, bz_alps_stats* st) {
    return guard([&] {
        need(p, "problem"); on_device(p->device); need(ao, "alps opts"); need(po, "panoc opts");
        need(x0, "x0"); need(y0, "y0"); need(x, "x"); need(y, "y"); need(s, "s"); need(mu, "mu");
        p->s->alps(*ao, *po, x0, y0, x, y, s, mu, st);
    });
}

int bz_als_solve(bz_problem* p, const bz_alps_opts* ao, const bz_panoc_opts* po, const void* x0,
                 const void* y0, void* x, void* y, void* s, void* mu, bz_alps_stats* st) {
    return guard([&] {
        need(p, "problem"); on_device(p->device); need(ao, "alps opts"); need(po, "panoc opts");
        need(x0, "x0"); need(y0, "y0"); need(x, "x"); need(y, "y"); need(s, "s"); need(mu, "mu");
        p->s->als(*ao, *po, x0, y0, x, y, s, mu, st);
    });
}

int bz_eval_al_gradient(bz_problem* p, const void* x, void* dlx, double* vals3) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(x, "x"); need(vals3, "vals3"); p->s->eval_al_gradient(x, dlx, vals3); });
}
int bz_eval_prox(bz_problem* p, const void* x, double gamma, void* z, double* gz) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(x, "x"); need(z, "z"); need(gz, "gz"); p->s->eval_prox(x, gamma, z, gz); });
}
int bz_eval_lbfgs(bz_problem* p, int32_t m, const void* S, const void* Y, const void* v, void* d) {
    return guard([&] {
        need(p, "problem"); on_device(p->device); need(v, "v"); need(d, "d");
        if (m > 0) { need(S, "S"); need(Y, "Y"); }
        p->s->eval_lbfgs(m, S, Y, v, d);
    });
}

int bz_problem_halo_export(bz_problem* p, void* handle64) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(handle64, "handle64"); p->s->halo_export(handle64); });
}
int bz_problem_halo_connect(bz_problem* p, const void* prev64, const void* next64) {
    return guard([&] { need(p, "problem"); on_device(p->device); p->s->halo_connect(prev64, next64); });
}

int bz_problem_allreduce_export(bz_problem* p, void* handle64) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(handle64, "handle64"); p->s->allreduce_export(handle64); });
}
int bz_problem_allreduce_connect(bz_problem* p, const void* handles) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(handles, "handles"); p->s->allreduce_connect(handles); });
}

int bz_profile_enable(bz_problem* p, int32_t on) {
    return guard([&] { need(p, "problem"); on_device(p->device); p->s->profile_enable((unsigned)on); });
}
int bz_profile_get(bz_problem* p, int32_t category, int64_t* launches, double* total_ms) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(launches, "launches"); need(total_ms, "total_ms"); p->s->profile_get(category, launches, total_ms); });
}
int bz_profile_get2(bz_problem* p, int32_t category, bz_profile_rec* out) {
    return guard([&] { need(p, "problem"); on_device(p->device); need(out, "out"); p->s->profile_get2(category, out); });
}
int bz_profile_reset(bz_problem* p) {
    return guard([&] { need(p, "problem"); on_device(p->device); p->s->profile_reset(); });
}

}  // extern "C"
