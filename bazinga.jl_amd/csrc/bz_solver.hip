// bz_solver.hip — device-resident PANOCplus / ALPS driver (host side) + kernel launches.
//
// Restates, for the lowered oracle kinds, what the reference runs at
// src/algorithms/alps.jl:64-66: ProximalAlgorithms.PANOCplus(...)(f=alFun, g=gFun, x0=x)
// with alFun = AugLagFun (src/utilities/auglagfun.jl) and gFun = NonsmoothCostFun
// (src/utilities/nonsmoothcostfun.jl).  The scalar control flow below is the same
// as oracle/bazinga_ref.py (PANOCplusIteration.init/step), which documents the
// provenance of the restatement.
#include "bz_solver.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <type_traits>

namespace bz {

Ctx::~Ctx() {
    for (int r = 0; r < P2P_MAXRANKS; ++r)
        if (mbox_opened[r] && mbox_peer[r]) (void)hipIpcCloseMemHandle(mbox_peer[r]);
    if (mbox_local) (void)hipFree(mbox_local);
    if (comm) (void)ncclCommDestroy(comm);
    if (stream) (void)hipStreamDestroy(stream);
}

static_assert(sizeof(P2PMailbox) == sizeof(P2PWords), "host and device mailbox layouts differ");

// allocate this rank's mailbox (fine-grained: peers' system-scope stores must be visible to a running
// kernel) and hand out its IPC handle
void p2p_export(Ctx* ctx, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle is 64 bytes");
    if (ctx->nranks > P2P_MAXRANKS) throw Error(BZ_ERR_ARG, "p2p supports at most 8 ranks");
    BZ_HIP(hipSetDevice(ctx->device));
    if (!ctx->mbox_local) {
        void* p = nullptr;
        BZ_HIP(hipExtMallocWithFlags(&p, sizeof(P2PMailbox), hipDeviceMallocFinegrained));
        BZ_HIP(hipMemset(p, 0, sizeof(P2PMailbox)));
        BZ_HIP(hipDeviceSynchronize());
        ctx->mbox_local = (P2PMailbox*)p;
    }
    hipIpcMemHandle_t h;
    BZ_HIP(hipIpcGetMemHandle(&h, ctx->mbox_local));
    std::memcpy(handle64, &h, sizeof(h));
}

// map every rank's mailbox; from here on scalar exchanges bypass RCCL
void p2p_connect(Ctx* ctx, const void* handles, const int32_t* devices) {
    if (!ctx->mbox_local) throw Error(BZ_ERR_STATE, "bz_ctx_p2p_export must be called first");
    BZ_HIP(hipSetDevice(ctx->device));
    const unsigned char* hs = (const unsigned char*)handles;
    for (int r = 0; r < ctx->nranks; ++r) {
        if (r == ctx->rank) { ctx->mbox_peer[r] = ctx->mbox_local; continue; }
        if (devices && devices[r] == ctx->device) ctx->shared_device = true;
        if (devices && devices[r] != ctx->device) {
            hipError_t e = hipDeviceEnablePeerAccess(devices[r], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                throw Error(BZ_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, hs + (size_t)r * 64, 64);
        void* p = nullptr;
        BZ_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        ctx->mbox_peer[r] = (P2PMailbox*)p;
        ctx->mbox_opened[r] = true;
    }
    ctx->p2p_on = true;
}

// ---------------------------------------------------------------------------
// one block per scalar: fold source blockIdx.x and write it to (host-mapped) out
__global__ void __launch_bounds__(BLOCK) k_collect(CollectArgs a, double* out) {
    __shared__ double sh[WAVES];
    const int i = blockIdx.x;
    double t = fold_src(a.src[i], (a.maxmask >> i) & 1u, sh);
    if (threadIdx.x == 0) host_post(out, i, t, a.ticket);
}

// the same for unit-stride sources with ONE wave per scalar (fold_wave gives fold_src's bits): a 64-thread
// workgroup needs no LDS and no barrier
__global__ void __launch_bounds__(64) k_collect_w(CollectArgs a, double* out) {
    const int i = blockIdx.x;
    const double t = fold_wave(a.src[i].p, a.src[i].count, (a.maxmask >> i) & 1u);
    if (threadIdx.x == 0) host_post(out, i, t, a.ticket);
}

// the persistent two-loop kernel's grid barrier gave up (its workgroups were not all resident): single-rank
// solves catch this, switch to the kernel chain for good and redo the iteration
// a pre-launched pass gave up at its gate (the host did not release it in time, or workgroup 0 was not resident in
// time because somebody else holds the CUs): it has written nothing that the same pass, launched again, does not
// write again — single-rank solves catch this, redo the iteration without the gate and leave the gate off
struct GateTimeout : Error {
    explicit GateTimeout(int code) : Error(BZ_ERR_COMM, code == 6 ? "a pre-launched pass timed out at its gate (the host never released it)"
                                                                   : "a pre-launched pass: workgroups timed out waiting for workgroup 0 to open the gate") {}
};
struct DenseFusedTimeout : Error {
    DenseFusedTimeout() : Error(BZ_ERR_HIP, "one-pass dense kernel: a row group's workgroups timed out waiting for each other (not all resident?)") {}
};
struct PersistTimeout : Error {
    PersistTimeout() : Error(BZ_ERR_HIP, "persistent two-loop kernel: grid barrier timed out (blocks not co-resident?)") {}
};

// set by bz_callback_abort() from inside a host callback (callbacks run on the thread that made the library call)
bool& callback_abort_flag() {
    static thread_local bool flag = false;
    return flag;
}

enum Cat : int { C_TWOLOOP = 0, C_FUSED = 1, C_ALGRAD = 2, C_FB = 3, C_UPDATE = 4,
                 C_COLLECT = 5, C_GATHER = 6, C_MISC = 7, C_DOT = 8, C_GEMV = 9, C_PERSIST = 10, C_GEMV_MFMA = 11, C_FUSED_IT = 12,
                 C_STENCIL_FB = 13, C_STENCIL_UPD = 14, C_XD = 15 };

template <class T> class Solver final : public SolverBase {
   public:
    Solver(Ctx* c, const bz_problem_desc& d) : ctx(c), desc(d), n(d.n), ny(d.ny), nx(d.n), slack(d.slack != 0) {
        cur_ = ctx->stream;
        if (n <= 0 || ny < 0) throw Error(BZ_ERR_ARG, "n must be positive");
        if (d.c_kind == BZ_C_IDENTITY && ny != n)
            throw Error(BZ_ERR_ARG, "c = Identity requires ny == n");
        if (slack) {
            // ALS: the inner solver works on xs = [x; s]; from here on `n` is the length of that vector
            if (d.c_kind != BZ_C_IDENTITY || (d.f_kind != BZ_F_ZERO && d.f_kind != BZ_F_DIAG_QUADRATIC))
                throw Error(BZ_ERR_UNSUPPORTED, "slack (ALS) form: c = Identity and element-wise f only");
            if (nx % PackN<T>::N != 0)
                throw Error(BZ_ERR_ARG, "slack (ALS) form: n must be a multiple of 16 bytes");
            if (ctx->nranks > 1) throw Error(BZ_ERR_UNSUPPORTED, "slack (ALS) form is not sharded");
            n = nx + ny;
        }
        {
            // generic (user-defined) oracles: host callbacks, all four together
            const int ncb = (d.f_kind == BZ_F_CALLBACK) + (d.g_kind == BZ_G_CALLBACK) + (d.c_kind == BZ_C_CALLBACK) +
                            (d.D_kind == BZ_D_CALLBACK);
            if (ncb != 0 && ncb != 4)
                throw Error(BZ_ERR_ARG, "generic oracles: f, g, c and D must all be the CALLBACK kind together");
            generic_ = ncb == 4;
            if (generic_) {
                if (!d.cb_f_gradient || !d.cb_g_prox || !d.cb_c_eval || !d.cb_c_jtprod || !d.cb_D_proj)
                    throw Error(BZ_ERR_ARG, "generic oracles: a callback pointer is null");
                if (slack || ctx->nranks > 1)
                    throw Error(BZ_ERR_UNSUPPORTED, "generic oracles: single rank, no slack form");
                if (ny <= 0) throw Error(BZ_ERR_ARG, "generic oracles: ny must be positive");
            }
        }
        if (d.c_kind != BZ_C_IDENTITY && d.c_kind != BZ_C_DENSE_AFFINE && !generic_)
            throw Error(BZ_ERR_UNSUPPORTED, "constraint kind not lowered to the device");
        if (d.c_kind == BZ_C_DENSE_AFFINE) {
            if (ny <= 0 || !d.c_A || !d.c_b) throw Error(BZ_ERR_ARG, "DenseAffine needs A[ny][n] and b[ny]");
            if (d.f_kind == BZ_F_STENCIL5) throw Error(BZ_ERR_UNSUPPORTED, "Stencil5pt f with a dense c");
            // nranks > 1: the ROWS of A (and b, mu, y: ny = this rank's rows) are sharded, x is replicated; the
            // n-vector A' yhat is summed over the ranks through IPC-mapped regions (bz_problem_allreduce_*)
            if (ctx->nranks > 1 && (!ctx->p2p_on || slack))
                throw Error(BZ_ERR_UNSUPPORTED, "a row-sharded DenseAffine needs the p2p mailboxes and no slack");
        }
        if ((d.f_kind < BZ_F_ZERO || d.f_kind > BZ_F_QUADRATIC) && !generic_)
            throw Error(BZ_ERR_UNSUPPORTED, "smooth-cost kind not lowered to the device");
        dense_f = d.f_kind == BZ_F_LEAST_SQUARES || d.f_kind == BZ_F_QUADRATIC;
        if (dense_f) {
            if (!d.f_A || !d.f_b || d.f_rows <= 0) throw Error(BZ_ERR_ARG, "dense f needs its matrix, vector and row count");
            if (d.f_kind == BZ_F_QUADRATIC && d.f_rows != n) throw Error(BZ_ERR_ARG, "Quadratic: Q must be n-by-n");
            if (d.c_kind != BZ_C_IDENTITY) throw Error(BZ_ERR_UNSUPPORTED, "dense f with a dense c");
            if (ctx->nranks > 1) throw Error(BZ_ERR_UNSUPPORTED, "dense f is not sharded");
        }
        if (d.f_kind == BZ_F_STENCIL5) {
            if (d.f_grid_nx <= 0 || d.f_grid_ny <= 0 || d.f_grid_nx * d.f_grid_ny != n)
                throw Error(BZ_ERR_ARG, "Stencil5pt: grid nx*ny must equal n");
            if (d.f_grid_ny % PackN<T>::N != 0)
                throw Error(BZ_ERR_ARG, "Stencil5pt: grid columns must be a multiple of 16 bytes");
            // nranks > 1: the grid is sharded by row blocks in rank order (f_grid_nx = this rank's rows); the
            // halo rows travel through IPC-mapped buffers (bz_problem_halo_export / _connect)
            if (ctx->nranks > 1 && !ctx->p2p_on)
                throw Error(BZ_ERR_UNSUPPORTED, "a sharded Stencil5pt needs the p2p mailboxes (bz_ctx_p2p_connect)");
        }
        if ((d.g_kind < BZ_G_ZERO || d.g_kind > BZ_G_NORM_LP_BOX) && !generic_)
            throw Error(BZ_ERR_ARG, "unknown g kind");
        if ((d.D_kind < BZ_D_ZERO || d.D_kind > BZ_D_XOR_PAIRS) && !generic_) throw Error(BZ_ERR_ARG, "unknown D kind");
        if (d.D_kind >= BZ_D_VC_PAIRS && d.D_kind <= BZ_D_XOR_PAIRS) {
            // adjacent pairs live inside one 16-byte pack: only the element-wise kernels (c = Identity) see them
            if (d.c_kind != BZ_C_IDENTITY || slack || d.f_kind == BZ_F_STENCIL5)
                throw Error(BZ_ERR_UNSUPPORTED, "pairwise D sets need c = Identity, an element-wise or dense f and no slack");
            if (ny % 2 != 0) throw Error(BZ_ERR_ARG, "pairwise D sets need an even number of constraints");
        }
        if ((d.g_kind == BZ_G_NORM_L1 || d.g_kind == BZ_G_NORM_L1_NONNEG ||
             d.g_kind == BZ_G_NORM_L1_BOX || d.g_kind == BZ_G_NORM_L0_BOX) && d.g_lambda < 0)
            throw Error(BZ_ERR_ARG, "parameter lambda must be nonnegative");
        BZ_HIP(hipSetDevice(ctx->device));
        const int64_t nchunks = (n + PackN<T>::N - 1) / PackN<T>::N;
        {
            hipDeviceProp_t prop;
            BZ_HIP(hipGetDeviceProperties(&prop, ctx->device));
            num_cus = prop.multiProcessorCount;
            pblocks = num_cus;
            // BZ_PERSIST_BLOCKS: run the persistent kernel on fewer CUs (two ranks sharing one GPU in tests)
            if (const char* e = getenv("BZ_PERSIST_BLOCKS")) pblocks = std::max(1, std::min(num_cus, atoi(e)));
            // persistent two-loop: one 512-thread block per CU, KR register packs per thread; its vectors
            // are zero-padded to KR*num_cus*512 packs so that every round is in-bounds (no masks)
            const int64_t kneed = (nchunks + (int64_t)pblocks * PBLOCK - 1) / ((int64_t)pblocks * PBLOCK);
            persist_kr = (kneed <= 48 && pblocks > 0 && pblocks <= PSTRIDE) ? persist_round_kr((int)kneed) : 0;
            // the hand-rolled grid barrier needs every workgroup resident at once: ask the runtime how many 512-thread
            // workgroups of this instantiation fit a CU (a non-resident grid would spin until the bounded polls give
            // up; that case — another process or stream holding CUs — is caught at run time, see step())
            if (persist_kr && persist_occupancy(persist_kr) * num_cus < pblocks) persist_kr = 0;
            vcap = n;
            if (persist_kr) vcap = std::max<int64_t>(n, (int64_t)persist_kr * pblocks * PBLOCK * PackN<T>::N);
        }
        int g = (int)std::min<int64_t>(PSTRIDE, std::max<int64_t>(1, (nchunks + BLOCK - 1) / BLOCK));
        if (const char* e = getenv("BZ_GRID")) g = std::max(1, std::min(PSTRIDE, atoi(e)));
        grid = g;
        const int64_t nychunks = (ny + PackN<T>::N - 1) / PackN<T>::N;
        grid_y = (int)std::min<int64_t>(grid, std::max<int64_t>(1, (nychunks + BLOCK - 1) / BLOCK));
        if (d.c_kind == BZ_C_IDENTITY) grid_y = grid;
        if (slack) {
            const int64_t c2 = (nx / PackN<T>::N + BLOCK - 1) / BLOCK;
            grid_y = (int)std::min<int64_t>(PSTRIDE, std::max<int64_t>(1, c2));
        }
        npad = ((n + PackN<T>::N - 1) / PackN<T>::N) * PackN<T>::N;
        if (dense_f) {
            frows = d.f_rows;
            FA_.alloc((size_t)frows * n);
            BZ_HIP(hipMemcpyAsync(FA_.p, d.f_A, (size_t)frows * n * sizeof(T), hipMemcpyDefault, ctx->stream));
            BZ_HIP(hipStreamSynchronize(ctx->stream));
            upload(fb_, d.f_b, d.f_kind == BZ_F_LEAST_SQUARES ? frows : n);
            FR_.alloc(std::max<int64_t>(frows, n));
            fscale = d.f_kind == BZ_F_LEAST_SQUARES ? T(0.5) : T(1);
            if (d.f_kind == BZ_F_LEAST_SQUARES) {
                DFX_.alloc(npad);
                plan_chunks(frows, f_rows_per_chunk, f_nrowchunks);
                GT_.alloc((size_t)f_nrowchunks * npad);
            }
        }
        if (d.c_kind == BZ_C_DENSE_AFFINE) {
            A_.alloc((size_t)ny * n);
            BZ_HIP(hipMemcpyAsync(A_.p, d.c_A, (size_t)ny * n * sizeof(T), hipMemcpyDefault, ctx->stream));
            BZ_HIP(hipStreamSynchronize(ctx->stream));
            upload(cb_, d.c_b, ny);
            CX_.alloc(ny); YU_.alloc(ny);
            plan_chunks(ny, rows_per_chunk, nrowchunks);
            x_replicated = ctx->nranks > 1;
            dense_fused_plan();
            GT_.alloc((size_t)std::max(nrowchunks, df_groups_) * npad);
            if (x_replicated) JL_.alloc(npad);
            affine_ok_ = !x_replicated && !slack && (d.D_kind == BZ_D_ZERO || d.D_kind == BZ_D_FREE) &&
                         (d.f_kind == BZ_F_ZERO || d.f_kind == BZ_F_DIAG_QUADRATIC);
            if (affine_ok_) {
                CXS_.alloc(ny); CZS_.alloc(ny); CXD_.alloc(ny); CZN_.alloc(ny);
            }
        }

        std::memset(&P, 0, sizeof(P));
        P.f_kind = d.f_kind; P.g_kind = d.g_kind; P.D_kind = d.D_kind;
        if (d.f_kind == BZ_F_DIAG_QUADRATIC) {
            if (!d.f_q || !d.f_b) throw Error(BZ_ERR_ARG, "DiagQuadratic needs q and b");
            upload(q_, d.f_q, nx); upload(b_, d.f_b, nx);
            P.q = q_.p; P.b = b_.p;
        }
        if (dense_f) P.b = fb_.p;                          // Quadratic: q, read by the element-wise kernels
        if (d.f_kind == BZ_F_STENCIL5) {
            if (!d.f_b) throw Error(BZ_ERR_ARG, "Stencil5pt needs b");
            upload(b_, d.f_b, n);
            P.b = b_.p;
        }
        P.g_lambda = (T)d.g_lambda;
        P.g_p = (T)d.g_p;
        lp_g = d.g_kind == BZ_G_NORM_LP_NONNEG || d.g_kind == BZ_G_NORM_LP_BOX;
        if (lp_g) {
            if (!(d.g_p > 0)) throw Error(BZ_ERR_ARG, "p must be positive");
            if (!(d.g_p < 1)) throw Error(BZ_ERR_ARG, "p must be smaller than one");
            if (d.g_lambda < 0) throw Error(BZ_ERR_ARG, "alpha must be nonnegative");
        }
        if (d.g_kind == BZ_G_NORM_L1_BOX || d.g_kind == BZ_G_NORM_L0_BOX || d.g_kind == BZ_G_NORM_LP_BOX) {
            if (!d.g_u) throw Error(BZ_ERR_ARG, "NormL1Box / NormL0Box / NormLpPowerBox need u");
            upload(gu_, d.g_u, nx); P.g_u = gu_.p;
        }
        P.g_lo = (T)d.g_lo; P.g_hi = (T)d.g_hi;
        if (d.g_kind == BZ_G_IND_BOX) {
            if (d.g_lo_vec) { upload(glo_, d.g_lo_vec, nx); P.g_lo_vec = glo_.p; }
            if (d.g_hi_vec) { upload(ghi_, d.g_hi_vec, nx); P.g_hi_vec = ghi_.p; }
        }
        P.D_lo = (T)d.D_lo; P.D_hi = (T)d.D_hi;
        if (d.D_kind == BZ_D_BOX) {
            if (d.D_lo_vec) { upload(dlo_, d.D_lo_vec, ny); P.D_lo_vec = dlo_.p; }
            if (d.D_hi_vec) { upload(dhi_, d.D_hi_vec, ny); P.D_hi_vec = dhi_.p; }
        }
        mu_.alloc(ny); muy_.alloc(ny); ymul_.alloc(ny); sproj_.alloc(ny);
        if (generic_) {
            const size_t nn = (size_t)std::max<int64_t>(n, ny);
            for (auto* v : {&hx_, &hg_, &hy_, &hz_, &hres_, &hdfx_, &hjtv_}) v->assign(nn, T(0));
            for (auto* v : {&hcx_, &ht_, &hs_, &hmu_, &hmuy_, &hyv_}) v->assign((size_t)ny, T(0));
        }
        P.mu = mu_.p; P.muy = muy_.p;
        for (auto& b : X_) b.alloc(vcap);
        for (auto& b : RES_) b.alloc(vcap);
        for (auto& b : Z_) b.alloc(vcap);
        GX_.alloc(vcap); GZ_.alloc(vcap); D_.alloc(vcap); TMP_.alloc(vcap);
        if (affine_ok_) { GXN_.alloc(vcap); GZN_.alloc(vcap); }
        parts_.alloc((size_t)SL_COUNT * PSTRIDE);
        BZ_HIP(hipMemsetAsync(parts_.p, 0, (size_t)SL_COUNT * PSTRIDE * sizeof(double), ctx->stream));
        alphas_.alloc(MAX_MEM + 1);
        send_.alloc(SL_COUNT);
        recv_.alloc((size_t)SL_COUNT * std::max(1, ctx->nranks));
        BZ_HIP(hipHostMalloc((void**)&host_out_, sizeof(double) * 2 * MAX_COLLECT, hipHostMallocMapped));
        std::memset(host_out_, 0, sizeof(double) * 2 * MAX_COLLECT);
        BZ_HIP(hipHostGetDevicePointer((void**)&host_out_dev_, host_out_, 0));
        BZ_HIP(hipHostMalloc((void**)&ptimeout_, sizeof(int), hipHostMallocMapped));
        *ptimeout_ = 0;
        BZ_HIP(hipHostGetDevicePointer((void**)&ptimeout_dev_, ptimeout_, 0));
        pcounter_.alloc(PSHARDS * PSHARD_STRIDE);      // zero-filled by alloc
        pgflag_.alloc(2); pglobal_.alloc(4);
        for (int s = 0; s < SL_COUNT; ++s) { grp_first[s] = s; grp_cnt[s] = 1; slot_n[s] = grid; }
        slot_n[SL_OUTER] = slot_n[SL_OUTER + 1] = grid_y;
        BZ_HIP(hipStreamSynchronize(ctx->stream));
    }

    ~Solver() override {
        gate_abort();
        if (gate_stream_) { (void)hipStreamSynchronize(gate_stream_); (void)hipStreamDestroy(gate_stream_); }
        (void)hipStreamSynchronize(ctx->stream);
        if (gate_host_) (void)hipHostFree(gate_host_);
        for (auto& r : prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto& e : ev_pool) (void)hipEventDestroy(e);
        for (int r = 0; r < P2P_MAXRANKS; ++r)
            if (ar_peer_[r] && ar_peer_[r] != ar_local_) (void)hipIpcCloseMemHandle(ar_peer_[r]);
        if (ar_local_) (void)hipFree(ar_local_);
        if (halo_prev_) (void)hipIpcCloseMemHandle(halo_prev_);
        if (halo_next_) (void)hipIpcCloseMemHandle(halo_next_);
        if (halo_local_) (void)hipFree(halo_local_);
        if (host_out_) (void)hipHostFree(host_out_);
        if (ptimeout_) (void)hipHostFree(ptimeout_);
    }

    // ------------------------------------------------------------------ API
    void set_multipliers(const void* mu, const void* y) override {
        copy_in(mu_.p, mu, ny);
        copy_in(ymul_.p, y, ny);
        aug_lag_update();
    }

    void begin(const bz_panoc_opts& o, const void* x0_host) override {
        copy_in(X_[0].p, x0_host, n);
        begin_dev(o, X_[0].p);
    }

    void solve(const bz_panoc_opts& o, const void* x0, void* x_out, bz_panoc_stats* st) override {
        begin(o, x0);
        run_to_completion();
        finish(x_out, st);
    }

    bool should_stop() const override {
        return k_ >= opt.maxit || (double)stop_norm_ <= opt.tol;
    }

    void finish(void* x_out, bz_panoc_stats* st) override {
        require_active();
        if (x_out) { ensure_z(false); copy_out(x_out, Z_[zc].p, n); }
        if (st) fill_stats(st);
    }

    void scalars(double* o) override {
        require_active();
        o[0] = (double)k_; o[1] = (double)gamma; o[2] = (double)tau; o[3] = (double)f_x;
        o[4] = (double)g_z; o[5] = (double)dot_gr; o[6] = (double)ss_res; o[7] = stop_norm_;
        o[8] = (double)last_ys; o[9] = (double)order.size(); o[10] = (double)H;
        o[11] = (double)f_z_al; o[12] = (double)fraw_last; o[13] = (double)last_nbt;
        o[14] = last_fused ? 1.0 : 0.0; o[15] = (double)fbe_last;
    }

    void vector(int which, void* out) override {
        require_active();
        switch (which) {
        case 0: copy_out(out, X_[xc].p, n); break;
        case 1: ensure_z(false); copy_out(out, Z_[zc].p, n); break;
        case 2: ensure_z(); copy_out(out, RES_[rc].p, n); break;
        case 3:
            if (!gx_valid) { algrad(X_[xc].p, GX_.p, SL_AUX); gx_valid = true; }
            copy_out(out, GX_.p, n); break;
        case 4:
            ensure_z();
            if (!gz_valid) { algrad(Z_[zc].p, GZ_.p, SL_AUX); gz_valid = true; }
            copy_out(out, GZ_.p, n); break;
        default: throw Error(BZ_ERR_ARG, "unknown vector id");
        }
    }

    void eval_al_gradient(const void* x, void* dlx, double* vals3) override {
        copy_in(TMP_.p, x, n);
        algrad(TMP_.p, D_.p, SL_AUX);           // (exchanges its two slots itself)
        auto v = collect({SL_AUX, SL_AUX + 1}, 0u);
        T half_pen = T(0.5) * T(v[1]);
        vals3[0] = (double)al_value(v[0], v[1]);
        vals3[1] = (double)f_value(v[0]);
        vals3[2] = (double)half_pen;
        if (dlx) copy_out(dlx, D_.p, n);
    }

    void eval_prox(const void* x, double gam, void* z, double* gz) override {
        copy_in(TMP_.p, x, n);
        fbstep(TMP_.p, nullptr, (T)gam, D_.p, nullptr, SL_GSUM);
        gather(SL_GSUM, 3, 0u);
        auto v = collect({SL_GSUM}, 0u);
        *gz = (double)g_value(v[0]);
        copy_out(z, D_.p, n);
    }

    void eval_lbfgs(int m, const void* S, const void* Y, const void* v, void* d) override {
        if (m < 0 || m > MAX_MEM) throw Error(BZ_ERR_ARG, "bad pair count");
        active = false;
        M = std::max(1, m);
        alloc_history();
        lbfgs_reset_all();
        rc = 0; xc = 0;
        const T* Sh = (const T*)S; const T* Yh = (const T*)Y;
        for (int i = 0; i < m; ++i) {   // oldest first, as update! would have seen them
            const int slot = spare;
            copy_in(S_[slot].p, Sh + (size_t)i * n, n);
            copy_in(Y_[slot].p, Yh + (size_t)i * n, n);
            mv(2); launch(C_MISC, k_dot<T>, grid, (const T*)S_[slot].p, (const T*)Y_[slot].p, T(1), n, parts_.p, (int)SL_YS);
            mv(1); launch(C_MISC, k_dot<T>, grid, (const T*)Y_[slot].p, (const T*)Y_[slot].p, T(1), n, parts_.p, (int)SL_YTY);
            gather(SL_YS, 2, 0u);
            auto r = collect({SL_YS, SL_YTY}, 0u);
            lbfgs_insert((T)r[0], (T)r[1]);
        }
        // d = H * v  == two-loop applied to -(-v)
        std::vector<T> neg(n);
        const T* vh = (const T*)v;
        for (int64_t i = 0; i < n; ++i) neg[i] = -vh[i];
        copy_in(RES_[rc].p, neg.data(), n);
        TailArgs<T> t = two_loop();
        mv(t.mode != 2 ? 3 : 2);
        launch(C_TWOLOOP, k_axpy_dot<T>, grid, t, (const T*)nullptr, (const T*)nullptr, D_.p, n,
               parts_.p, 0);
        copy_out(d, D_.p, n);
    }

    void profile_enable(unsigned mask) override {
        prof_mask = mask & 0xFFFFu;
        prof_period = std::max(1u, mask >> 16);          // upper 16 bits: time every k-th launch only
        for (auto& c : prof_count) c = 0;
    }
    void profile_reset() override {
        drain_prof();
        for (int c = 0; c < BZ_NUM_KERNEL_CATEGORIES; ++c) {
            prof_ms[c] = 0; prof_n[c] = 0; bytes_all_[c] = 0; bytes_timed_[c] = 0; launches_all_[c] = 0;
        }
    }
    void profile_get2(int cat, bz_profile_rec* r) override {
        if (cat < 0 || cat >= BZ_NUM_KERNEL_CATEGORIES) throw Error(BZ_ERR_ARG, "bad category");
        drain_prof();
        std::memset(r, 0, sizeof(*r));
        r->timed_launches = prof_n[cat]; r->timed_ms = prof_ms[cat]; r->timed_bytes = bytes_timed_[cat];
        r->launches = launches_all_[cat]; r->bytes = bytes_all_[cat];
        std::strncpy(r->form, form_[cat].c_str(), sizeof(r->form) - 1);
    }
    void profile_get(int cat, int64_t* launches, double* ms) override {
        if (cat < 0 || cat >= BZ_NUM_KERNEL_CATEGORIES) throw Error(BZ_ERR_ARG, "bad category");
        drain_prof();
        *launches = prof_n[cat];
        *ms = prof_ms[cat];
    }

    // ------------------------------------------------------- alps (alps.jl:7-117)
    void alps(const bz_alps_opts& ao, const bz_panoc_opts& po, const void* x0, const void* y0,
              void* xo, void* yo, void* so, void* muo, bz_alps_stats* st) override {
        if (slack) throw Error(BZ_ERR_STATE, "bz_alps_solve on a slack (ALS) problem: use bz_als_solve");
        if (ao.warm_start & ~1) throw Error(BZ_ERR_ARG, "bz_alps_opts.warm_start: unknown bit (bit 0: the step size)");
        auto t0 = std::chrono::steady_clock::now();
        const T epsT = std::numeric_limits<T>::epsilon();
        T* x = X_[0].p;
        copy_in(TMP_.p, x0, n);
        // prox!(x, gFun, x0, eps(T))                                   alps.jl:38
        fbstep(TMP_.p, nullptr, epsT, x, nullptr, SL_GSUM);
        gather(SL_GSUM, 3, 0u);
        // objx = f(x) + gFun.gz                                        alps.jl:39
        fvalue(x, SL_AUX);
        auto v0 = collect({SL_GSUM, SL_AUX}, 0u);
        T gz0 = g_value(v0[0]);
        T objx = f_value(v0[1]) + gz0;
        // eval!(cx,c,x); proj!(s,D,cx); default_penalty_parameter!     alps.jl:40-42
        const double denom = std::max(1.0, (double)objx);
        const bool dense_c = desc.c_kind == BZ_C_DENSE_AFFINE;
        if (dense_c) eval_c(x);
        if (generic_) {
            // eval!(cx, c, x) ; proj!(s, D, cx) ; default_penalty_parameter!   (alps.jl:40-42, safeguards.jl:13-18:
            // Float64 literals, stored back into T — the arithmetic of k_penalty_init)
            copy_out(hx_.data(), x, n);
            cb_c_eval(hx_.data(), hcx_.data(), n, ny);
            cb_D_proj(hcx_.data(), hs_.data(), ny);
            for (int64_t i = 0; i < ny; ++i) {
                const T dd = hcx_[i] - hs_[i];
                const double h = 0.5 * (double)(dd * dd);
                T mm = (T)((h > 1.0 ? h : 1.0) / denom);
                mm = (T)((double)mm * 0.1);
                double w = (double)mm;
                w = w < 1e8 ? w : 1e8;
                w = w > 1e-8 ? w : 1e-8;
                hmu_[i] = (T)w;
            }
            copy_in(sproj_.p, hs_.data(), ny);
            copy_in(mu_.p, hmu_.data(), ny);
        } else {
        mv(3 + (P.D_lo_vec ? 1 : 0) + (P.D_hi_vec ? 1 : 0), ny);
        launch(C_MISC, k_penalty_init<T>, grid_y, dense_c ? (const T*)CX_.p : (const T*)x /* cx = x */, P, denom,
               sproj_.p, mu_.p, ny);
        }
        copy_in(ymul_.p, y0, ny);                                    // y .= y0
        double norm_res_prim = 0, norm_res_prim_old = 0;
        bool have_old = false, have_res = false;
        int64_t tot_it = 0, tot_inner = 0;
        double inner_tol = ao.inner_tol;
        bool solved = false, tired = tot_it >= ao.maxit, broken = std::isnan((double)objx);
        if (ao.verbose) {
            std::printf("[ Info: initial inner tolerance %g\n", inner_tol);
        }
        bool can_stop = solved || tired || broken;
        bz_panoc_opts po2 = po;
        while (!can_stop) {
            ++tot_it;
            po2.tol = inner_tol;                                     // alps.jl:64
            po2.verbose = ao.verbose;
            // opt-in (bz_alps_opts.warm_start bit 0): subsolver(tol, verbose; gamma = gamma_prev, adaptive = true)
            if ((ao.warm_start & 1) && tot_it > 1 && (double)gamma > 0.0) { po2.gamma = (double)gamma; po2.adaptive = 1; }
            // dual_safeguard(y, cx)  alps.jl:62  +  AugLagUpdate!  alps.jl:65, one pass
            aug_lag_update(true);
            begin_dev(po2, x);                                       // alps.jl:66
            run_to_completion();
            const int64_t sub_it = k_;
            ensure_z(false);
            x = Z_[zc].p;                                            // x .= sub_sol
            objx = fraw_last + g_z;                                  // alps.jl:68
            tot_inner += sub_it;
            const bool sub_solved = sub_it < ao.subsolver_maxit;     // alps.jl:70
            // dual update + primal residual                          alps.jl:72-84
            if (dense_c) eval_c(x);                                  // eval!(cx, c, x)  alps.jl:72
            if (generic_) {
                copy_out(hx_.data(), x, n);
                cb_c_eval(hx_.data(), hcx_.data(), n, ny);          // eval!(cx, c, x)      alps.jl:72
                for (int64_t i = 0; i < ny; ++i) hyv_[i] = hcx_[i] + hmuy_[i];         // y .= cx .+ muy       :74
                cb_D_proj(hyv_.data(), hs_.data(), ny);             // proj!(s, D, y)       :75
                double nrm = 0.0;
                for (int64_t i = 0; i < ny; ++i) {
                    T t = hyv_[i] - hs_[i];                                            // y .-= s              :80
                    hyv_[i] = t / hmu_[i];                                             // y ./= mu             :81
                    const T r = hcx_[i] - hs_[i];
                    const double ar = (double)(r < T(0) ? -r : r);
                    if (ar > nrm || ar != ar) nrm = ar;                                // norm(cx - s, Inf)    :84
                }
                copy_in(ymul_.p, hyv_.data(), ny);
                copy_in(sproj_.p, hs_.data(), ny);
                fill_slot(SL_OUTER, nrm);
            } else {
            mv(3 + pstreams(false, true, false), ny);
            launch(C_MISC, k_dual_update<T>, grid_y, dense_c ? (const T*)CX_.p : (const T*)x, P, ymul_.p, sproj_.p,
                   ny, parts_.p, (int)SL_OUTER);
            }
            gather(SL_OUTER, 1, 1u, 1u);
            auto r = collect({SL_OUTER}, 1u);
            norm_res_prim_old = norm_res_prim; have_old = have_res;
            norm_res_prim = r[0]; have_res = true;
            solved = (inner_tol <= ao.tol_dual && sub_solved) && (norm_res_prim <= ao.tol_prim);
            tired = tot_it >= ao.maxit;
            broken = std::isnan((double)objx);
            can_stop = solved || tired || broken;
            if (!can_stop) {
                if (have_old &&
                    norm_res_prim > std::max(ao.theta_penalty * norm_res_prim_old, ao.tol_prim)) {
                    mv(2, ny); launch(C_MISC, k_clamp_scale<T>, grid_y, mu_.p, 0.0, 0.0, (T)ao.kappa_penalty, 0, ny);
                }
                inner_tol = std::max(ao.kappa_tol * inner_tol, ao.tol_dual);
            }
            // next subproblem starts from x (kept in the z buffer): copy to a state buffer
            // (the z buffer BECOMES the first state buffer: the next bz_panoc_begin recomputes z anyway)
            if (!can_stop) {
                std::swap(X_[0].p, Z_[zc].p);
                std::swap(X_[0].n, Z_[zc].n);
                x = X_[0].p;
            }
        }
        copy_out(xo, x, n);
        copy_out(yo, ymul_.p, ny);
        copy_out(so, sproj_.p, ny);
        copy_out(muo, mu_.p, ny);
        if (st) {
            st->tot_it = tot_it; st->tot_inner_it = tot_inner;
            st->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            st->status = solved ? 0 : (tired ? 1 : (broken ? 2 : 3));
            st->inner_tol = inner_tol; st->norm_res_prim = norm_res_prim;
            st->objective = (double)objx;
        }
    }

    // ------------------------------------------------------- als (als.jl:7-120)
    void als(const bz_alps_opts& ao, const bz_panoc_opts& po, const void* x0, const void* y0,
             void* xo, void* yo, void* so, void* muo, bz_alps_stats* st) override {
        if (!slack) throw Error(BZ_ERR_STATE, "bz_als_solve needs a problem created with desc.slack = 1");
        if (ao.warm_start & ~1) throw Error(BZ_ERR_ARG, "bz_alps_opts.warm_start: unknown bit (bit 0: the step size)");
        auto t0 = std::chrono::steady_clock::now();
        const T epsT = std::numeric_limits<T>::epsilon();
        T* xs = X_[0].p;                                   // [x; s]
        copy_in(TMP_.p, x0, nx);
        // prox!(x, gFun, x0, eps(T)) ; objx = f(x) + gFun.gz              als.jl:41-42
        mv(2 + pstreams(false, false, true), nx);
        if (lp_g) launch(C_FB, k_fbstep<T, true>, grid_y, (const T*)TMP_.p, (const T*)nullptr, epsT, P, xs, (T*)nullptr, nx, parts_.p, (int)SL_GSUM);
        else launch(C_FB, k_fbstep<T, false>, grid_y, (const T*)TMP_.p, (const T*)nullptr, epsT, P, xs, (T*)nullptr, nx, parts_.p, (int)SL_GSUM);
        for (int k = 0; k < 3; ++k) slot_n[SL_GSUM + k] = grid_y;
        fvalue(xs, SL_AUX);
        auto v0 = collect({SL_GSUM, SL_AUX}, 0u);
        T objx = f_value(v0[1]) + g_value(v0[0]);
        // eval!(cx,c,x); proj!(s,D,cx); default_penalty_parameter!          als.jl:43-45   (s lands in xs[nx:])
        mv(3 + (P.D_lo_vec ? 1 : 0) + (P.D_hi_vec ? 1 : 0), ny);
        launch(C_MISC, k_penalty_init<T>, grid_y, (const T*)xs, P, std::max(1.0, (double)objx), xs + nx, mu_.p, ny);
        copy_in(ymul_.p, y0, ny);
        double norm_res_prim = 0, norm_res_prim_old = 0;
        bool have_old = false, have_res = false;
        int64_t tot_it = 0, tot_inner = 0;
        double inner_tol = ao.inner_tol;
        bool solved = false, tired = tot_it >= ao.maxit, broken = std::isnan((double)objx);
        if (ao.verbose) std::printf("[ Info: initial inner tolerance %g\n", inner_tol);
        bool can_stop = solved || tired || broken;
        bz_panoc_opts po2 = po;
        while (!can_stop) {
            ++tot_it;
            po2.tol = inner_tol; po2.verbose = ao.verbose;
            if ((ao.warm_start & 1) && tot_it > 1 && (double)gamma > 0.0) { po2.gamma = (double)gamma; po2.adaptive = 1; }
            aug_lag_update(true);                                    // dual_safeguard + AugLagUpdate!(fSlack, mu, y)
            begin_dev(po2, xs);                                      // sub_solver(f=fSlack, g=gSlack, x0=xSlack)
            run_to_completion();
            const int64_t sub_it = k_;
            ensure_z(false);                                         // (the one-pass kernel keeps z in registers until it is asked for)
            xs = Z_[zc].p;                                           // xSlack .= sub_sol
            objx = fraw_last + g_z;                                  // f(x) + gSlack.gz       als.jl:79
            tot_inner += sub_it;
            const bool sub_solved = sub_it < ao.subsolver_maxit;
            // y += (cx - s)/mu ; ||cx - s||_inf                      als.jl:82-87
            mv(5, nx);
            launch(C_MISC, k_dual_update_slack<T>, grid_y, (const T*)xs, (const T*)mu_.p, ymul_.p, nx, parts_.p,
                   (int)SL_OUTER);
            slot_n[SL_OUTER] = grid_y;
            gather(SL_OUTER, 1, 1u);
            auto r = collect({SL_OUTER}, 1u);
            norm_res_prim_old = norm_res_prim; have_old = have_res;
            norm_res_prim = r[0]; have_res = true;
            solved = (inner_tol <= ao.tol_dual && sub_solved) && (norm_res_prim <= ao.tol_prim);
            tired = tot_it >= ao.maxit;
            broken = std::isnan((double)objx);
            can_stop = solved || tired || broken;
            if (!can_stop) {
                if (have_old && norm_res_prim > std::max(ao.theta_penalty * norm_res_prim_old, ao.tol_prim)) {
                    mv(2, ny); launch(C_MISC, k_clamp_scale<T>, grid_y, mu_.p, 0.0, 0.0, (T)ao.kappa_penalty, 0, ny);
                }
                inner_tol = std::max(ao.kappa_tol * inner_tol, ao.tol_dual);
                BZ_HIP(hipMemcpyAsync(X_[0].p, xs, n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
                xs = X_[0].p;
            }
        }
        copy_out(xo, xs, nx);
        copy_out(so, xs + nx, ny);
        copy_out(yo, ymul_.p, ny);
        copy_out(muo, mu_.p, ny);
        if (st) {
            st->tot_it = tot_it; st->tot_inner_it = tot_inner;
            st->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            st->status = solved ? 0 : (tired ? 1 : (broken ? 2 : 3));
            st->inner_tol = inner_tol; st->norm_res_prim = norm_res_prim;
            st->objective = (double)objx;
        }
    }

   private:
    // ------------------------------------------------------------ plumbing
    Ctx* ctx;
    bz_problem_desc desc;
    int64_t n, ny;
    int64_t nx;                              // length of x (== n unless slack: then n = nx + ny)
    bool slack;
    int grid = 1, grid_y = 1;
    ElemParams<T> P;
    DBuf<T> q_, b_, gu_, glo_, ghi_, dlo_, dhi_, mu_, muy_, ymul_, sproj_;
    // x and res live in rings long enough to keep the last CM+1 iterates alive (history as iterates, see
    // xr_run_): CM+1 snapshots + the slot being written (+ one more x slot for the tau blend)
    static constexpr int NXR = CM + 3, NRR = CM + 2;
    DBuf<T> X_[NXR], RES_[NRR], Z_[2], GX_, GZ_, D_, TMP_;
    DBuf<T> A_, cb_, CX_, YU_, GT_;          // DenseAffine c: A[ny][n], b, c(x), yupd, A'v row-chunk partials
    int rows_per_chunk = 1, nrowchunks = 1;
    DBuf<T> FA_, fb_, FR_, DFX_;             // dense f: matrix, vector, residual / Qx, gradient of f
    bool dense_f = false, lp_g = false;
    // generic oracles (host callbacks): host mirrors of the vectors the callbacks read and write
    bool generic_ = false;
    std::vector<T> hx_, hg_, hy_, hz_, hres_, hdfx_, hjtv_, hcx_, ht_, hs_, hmu_, hmuy_, hyv_;
    // the five host callbacks; a callback that failed says so through bz_callback_abort() (it cannot unwind through the
    // C frames): the library call in progress then ends with BZ_ERR_CALLBACK as soon as the callback has returned,
    // instead of iterating on whatever the failed callback left in its output buffers
    static void cb_check() {
        if (callback_abort_flag()) { callback_abort_flag() = false; throw Error(BZ_ERR_CALLBACK, "an oracle callback failed (bz_callback_abort)"); }
    }
    void cb_c_eval(const T* x, T* cx, int64_t n_, int64_t ny_) { desc.cb_c_eval(desc.cb_user, x, cx, n_, ny_); cb_check(); }
    void cb_D_proj(const T* v, T* s, int64_t ny_) { desc.cb_D_proj(desc.cb_user, v, s, ny_); cb_check(); }
    void cb_c_jtprod(const T* x, const T* v, T* jtv, int64_t n_, int64_t ny_) { desc.cb_c_jtprod(desc.cb_user, x, v, jtv, n_, ny_); cb_check(); }
    double cb_f_gradient(const T* x, T* dfx, int64_t n_) { const double v = desc.cb_f_gradient(desc.cb_user, x, dfx, n_); cb_check(); return v; }
    double cb_g_prox(const T* x, double gam, T* z, int64_t n_) { const double v = desc.cb_g_prox(desc.cb_user, x, gam, z, n_); cb_check(); return v; }
    void fill_slot(int slot, double v) {
        launch_b(C_MISC, k_fill_slot, 1, 64, parts_.p, slot, v);
        slot_n[slot] = 1;
    }
    // gradient!(dlx, al, x) with the oracles evaluated on the host, statement by statement as
    // src/utilities/auglagfun.jl:73-86 (the value-only form :58-69 is the same minus dfx and jtv)
    void algrad_generic(const T* x, T* grad, int slot0) {
        copy_out(hx_.data(), x, n);
        cb_c_eval(hx_.data(), hcx_.data(), n, ny);                  // eval!(cx, c, x)          :74
        for (int64_t i = 0; i < ny; ++i) ht_[i] = hcx_[i] + hmuy_[i];                  // yupd .= cx .+ muy        :75
        cb_D_proj(ht_.data(), hs_.data(), ny);                      // proj!(s, D, yupd)        :76
        double pen = 0.0;
        for (int64_t i = 0; i < ny; ++i) {
            T t = ht_[i] - hs_[i];                                                     // yupd .-= s               :77
            pen += (double)((t * t) / hmu_[i]);                                        // sum(yupd.^2 ./ mu)       :78
            ht_[i] = t / hmu_[i];                                                      // yupd ./= mu              :79
        }
        const double fx = cb_f_gradient(hx_.data(), hdfx_.data(), n);   // fx = gradient!(dfx, f, x)  :80
        if (grad) {
            cb_c_jtprod(hx_.data(), ht_.data(), hjtv_.data(), n, ny);   // jtprod!(jtv, c, x, yupd)   :83
            for (int64_t j = 0; j < n; ++j) hg_[j] = hdfx_[j] + hjtv_[j];              // dlx .= dfx .+ jtv        :84
            copy_in(grad, hg_.data(), n);
        }
        fill_slot(slot0, fx);
        fill_slot(slot0 + 1, pen);
    }
    // y = x - gamma g ; z = prox!(., g, y, gamma) ; res = x - z   (nonsmoothcostfun.jl:17-22 and the caller's
    // forward step), with the sums k_fbstep returns
    void fbstep_generic(const T* x, const T* g, T gam, T* z, T* res, int slot0) {
        copy_out(hx_.data(), x, n);
        if (g) copy_out(hg_.data(), g, n);
        for (int64_t j = 0; j < n; ++j) {
            T yv = hx_[j];
            if (g) { T t = gam * hg_[j]; yv = hx_[j] - t; }
            hy_[j] = yv;
        }
        const double gz = cb_g_prox(hy_.data(), (double)gam, hz_.data(), n);
        double dot = 0.0, ss = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            const T r = hx_[j] - hz_[j];
            hres_[j] = r;
            if (g) dot += (double)(hg_[j] * r);
            ss += (double)(r * r);
        }
        copy_in(z, hz_.data(), n);
        if (res) copy_in(res, hres_.data(), n);
        fill_slot(slot0, gz);
        fill_slot(slot0 + 1, dot);
        fill_slot(slot0 + 2, ss);
    }
    int64_t frows = 0, npad = 0;
    int f_rows_per_chunk = 1, f_nrowchunks = 1;
    T fscale = T(1);                         // f(x) = fscale * (sum of the f partials)
    // persistent two-loop
    DBuf<unsigned long long> pcounter_, pgflag_;
    DBuf<double> pglobal_;                   // [0..1] phase totals (double-buffered), [2] final <y_0,d>
    int pblocks = 0;                         // blocks of the persistent grid (= CUs unless overridden)
    unsigned long long pbase = 0;
    int* ptimeout_ = nullptr;                // host-mapped
    int* ptimeout_dev_ = nullptr;
    int num_cus = 0, persist_kr = 0;
    int64_t vcap = 0;                        // allocated elements per n-vector (>= n, zero-padded)
    bool persist_ok = false;
    bool stencil_fast_ = false;              // Stencil5pt f with the two fused stencil passes (see step())
    // Affine images (dense affine c, D = ZeroSet / FreeSet, f = Zero / DiagQuadratic: c(.) and grad L(.) are affine maps):
    // the images of the trial point x + d are formed from the stored images of the iterates — the same linear
    // combination that forms d — instead of two passes over A; a pass-over-A evaluation every `aff_refresh_`
    // iterations stops rounding drift.  State: c and grad L at the current x and z (CXS_, GX_, CZS_, GZ_), their
    // candidates (CXD_, CZN_, GXN_, GZN_) and the images of every stored pair (AS_, AY_: ny-vectors; GS_, GY_: n).
    bool affine_ok_ = false, aff_track_ = false;
    int aff_refresh_ = 8, aff_count_ = 0;
    int64_t n_affine_ = 0, n_affine_verify_ = 0;
    T* cx_keep_ = nullptr;                   // algrad (dense c): also leave c(point) here
    DBuf<T> CXS_, CZS_, CXD_, CZN_, GXN_, GZN_;
    std::vector<DBuf<T>> AS_, AY_, GS_, GY_;
    CompactVecs<T, CM> image_vecs(bool ny_space) const {      // logical (oldest first) view, as compact_vecs()
        CompactVecs<T, CM> V;
        std::memset(&V, 0, sizeof(V));
        V.m = (int)order.size();
        for (int i = 0; i < V.m; ++i) {
            const int s = order[V.m - 1 - i];
            V.S[i] = ny_space ? AS_[s].p : GS_[s].p;
            V.Y[i] = ny_space ? AY_[s].p : GY_[s].p;
        }
        return V;
    }
    bool persist_broken_ = false;            // a grid barrier timed out once on this problem: the kernel chain from then on
    bool persist_sabotage_ = false;          // BZ_TEST_PERSIST_TIMEOUT=1: make the barrier miss its target (tests the fallback)
    std::vector<DBuf<T>> S_, Y_;
    DBuf<double> parts_, alphas_, send_, recv_;
    double* host_out_ = nullptr;             // pinned mailbox: {value, ticket} per collected scalar
    double* host_out_dev_ = nullptr;
    unsigned long long collect_seq = 0;
    int grp_first[SL_COUNT], grp_cnt[SL_COUNT];
    int slot_n[SL_COUNT];                    // number of valid block partials per slot

    // solver state (host scalars)
    bz_panoc_opts opt{};
    bool active = false, fused_ok = false, gx_valid = false, gz_valid = false;
    // The one-pass compact kernel computes z in registers and does not store it: nothing in a plain iteration
    // reads it back (the next iterate is x_d, the stopping test uses grad L(z) formed in the same pass), and
    // the store is the dearest of the kernel's streams (-8 % of its time).  Who does need it — a tau backtrack
    // (z_curr), the caller asking for the solution — gets it re-materialised bit for bit from x and gamma.
    bool z_valid = true;
    // History as iterates: once the last CM iterations were plain ones that each inserted their pair, the CM
    // stored pairs are the successive differences of the last CM+1 iterates, which the long x / res rings
    // still hold — the fused pass then reads those snapshots instead of S and Y (same number of streams,
    // same bits: s = x_d - x, y = res - res_prev are re-formed by the very subtraction that made them) and
    // stops WRITING s and y (two of its dearest streams).  The first iteration that is not a plain one
    // turns the snapshots back into pairs (k_pairs_from_snapshots) and the classic kernels take over.
    // One step further (headline family, after CM+1 such iterations): the residual of an iterate is a function
    // of that iterate alone (res = x - prox(x - gamma grad L(x)), with gamma, mu, mu*y fixed along the run), so
    // the fused pass re-evaluates the CM+1 residuals from the CM+1 iterates instead of reading them, and
    // stops writing res as well: reads CM+1 iterates + q, b, mu, mu*y, writes x_d.  Same operations on the
    // same inputs as when each residual was first computed -> the same bits.
    int xr_run_ = 0;             // consecutive plain, pair-inserting iterations so far
    int xr_env_ = 2, skipz_env_ = 1;     // BZ_XR / BZ_SKIPZ, read at every bz_panoc_begin (tests toggle them)
    int gfc_env_ = 0, trialfuse_env_ = 1, fused_begin_env_ = 1;      // BZ_GFC / BZ_TRIALFUSE / BZ_FUSED_BEGIN, likewise
    int slackfast_env_ = 1;      // BZ_SLACKFAST=0: the slack iterate-history pass always in its run-time-kinds instantiation
    int suc_grid_env_ = 1;       // BZ_SUC_GRID=k: k_stencil_update_c on k workgroups per CU (default 1; 0: the problem's grid)
    int stencil_regx_env_ = 1;   // BZ_STENCIL_REGX=0|1|2: cfg 3's second pass reads res and grad L(x_d) / re-forms res (default) / re-forms both (slower)
    int affblend_env_ = 1;       // BZ_AFFINE_BLEND=0: a tau-backtracked point of cfg 4 is always evaluated with a pass over A (no images)
    int densesmall_env_ = 1;     // BZ_DENSESMALL=0: cfg 4's short kernels either side of the pass over A as launches of their own (k_dense_head / k_dense_tail off)
    int slackkind_env_ = 1;      // BZ_SLACKKIND=0: its fast instantiations with run-time kinds of g and D
    int slackdepth_env_ = 1;     // BZ_SLACKDEPTH=0: ... without the one-pack-ahead register pipeline (232 against 227 us per pass)
    int nt_env_ = -1;            // BZ_NT: -1 (default) non-temporal streams by working-set size, 0 / 1 forced
    int famrt_env_ = 0;          // BZ_FAMRT=1: the headline family through its family-table instantiation (run-time UNI / TRIAL)
    bool sy_stale_ = false;      // S_/Y_ do not hold the stored pairs (they live in the rings)
    bool rh_stale_ = false;      // ... and the residual ring was not written either during this run
    double gring_[NXR] = {0};    // the gamma the residual of each iterate in the ring was (or would be) formed with
    bool res_valid = true;       // RES_[rc] holds the residual of the current state
    void materialize_pairs() {
        if (!sy_stale_) return;
        SnapVecs<T, CM> V;
        std::memset(&V, 0, sizeof(V));
        const int m = (int)order.size();      // (< CM only in the re-evaluating form, soon after a memory reset)
        for (int i = 0; i <= CM; ++i) {
            const int back = std::max(0, m - i);
            V.XH[i] = X_[(xc - back + NXR) % NXR].p;
            V.RH[i] = RES_[(rc - back + NRR) % NRR].p;
            V.gam[i] = back ? gring_[(xc - back + NXR) % NXR] : (double)gamma;
        }
        for (int i = 0; i < m; ++i) { V.S[i] = S_[order[m - 1 - i]].p; V.Y[i] = Y_[order[m - 1 - i]].p; }
        if (rh_stale_ && slack) {
            // (the lifted vector: both halves of every iterate in, both halves of the pairs, the residual and z out)
            mv(2 * ((m + 1) + 2 * m + 2) + pstreams(true, true, true) + (P.uni >= 2 ? 0 : 1), nx);
            launch(C_MISC, k_pairs_from_iterates_slack<T, CM>, grid_y, V, m, P, (const T*)ymul_.p, RES_[rc].p, Z_[zc].p, nx);
            res_valid = true; z_valid = true;
        } else if (rh_stale_) {
            // (must run before gamma changes: the residuals are re-evaluated with the gamma of this run)
            mv((m + 1) + pstreams(true, true, true) + 2 * m + 2);
            launch(C_MISC, k_pairs_from_iterates<T, CM>, grid, V, m, P, RES_[rc].p, Z_[zc].p, n);
            res_valid = true; z_valid = true;
        } else {
            if (m != CM) throw Error(BZ_ERR_STATE, "history as snapshots with a partial memory");
            mv(2 * (CM + 1) + 2 * CM);
            launch(C_MISC, k_pairs_from_snapshots<T, CM>, grid, V, n);
        }
        sy_stale_ = false;
    }
    int xc = 0, rc = 0, zc = 0;
    T alpha = T(0.95), beta = T(0.5), min_gamma = T(1e-7), musqy = T(0);
    T gamma_given_ = T(0);       // bz_panoc_opts.gamma / alpha / Lf (0: estimate a Lipschitz constant)
    bool adaptive_ = true;       // upstream's `adaptive`: gamma halvings at the start and inside the line search
    T gamma = T(0), tau = T(0), f_x = T(0), g_z = T(0), dot_gr = T(0), ss_res = T(0);
    T f_z_al = T(0), fraw_last = T(0), last_ys = T(0), fbe_last = T(0);
    double stop_norm_ = 0;
    int64_t k_ = 0, n_grad = 0, n_prox = 0, n_bt = 0, n_halv = 0, n_fused = 0, n_skips = 0;
    int last_nbt = 0;
    bool last_fused = false;
    std::chrono::steady_clock::time_point t_begin;
    // L-BFGS ring: M+1 physical slots, `order` newest first, `spare` receives the candidate pair
    int M = 5;
    std::deque<int> order;
    std::vector<int> freeslots;
    int spare = 0;
    T ys_[MAX_MEM + 1];
    T H = T(1);
    // `directions` other than L-BFGS (bz_panoc_opts.directions)
    int dir_kind_ = BZ_DIR_LBFGS;
    T broyden_theta_bar_ = T(0.2);
    DBuf<T> HB_, BHy_, BsH_;                 // Broyden: the dense operator (n x n, row-major), H y, H's
    int b_rpc_ = 1, b_nch_ = 1;
    // Broyden: D_ = H res (the caller forms x_d = x - D_ through the returned tail)
    TailArgs<T> broyden_dir() {
        gemv_rows(HB_.p, n, RES_[rc].p, (const T*)nullptr, D_.p);
        TailArgs<T> t;
        std::memset(&t, 0, sizeof(t));
        t.alphas = alphas_.p;
        t.in = D_.p; t.v = nullptr; t.sgn = T(-1); t.mode = 2; t.apply_H = 0; t.H = T(1);
        t.src = ScalarSrc{parts_.p, 0, 1}; t.ys = T(1);
        return t;
    }
    void broyden_reset() {
        launch(C_MISC, k_set_identity<T>, (int)std::min<int64_t>(PSTRIDE, (n * n + BLOCK - 1) / BLOCK), HB_.p, n);
    }
    // update!(H, s, y) — the pair sits in S_[spare], Y_[spare]:
    //   Hy = H y ; sH = s'H ; delta = <Hy, s> / <s, s> ; theta = 1 if |delta| >= theta_bar else
    //   (1 - sgn(delta) theta_bar) / (1 - delta), sgn(0) = 1 ; H += (s - Hy) / <s, (1/theta - 1) s + Hy> * sH
    void broyden_update() {
        const T* sv = S_[spare].p;
        const T* yv = Y_[spare].p;
        gemv_rows(HB_.p, n, yv, (const T*)nullptr, BHy_.p);
        gemv_cols(HB_.p, n, sv, b_rpc_, b_nch_);
        {
            ElemParams<T> Pz = P;
            Pz.f_kind = BZ_F_ZERO;
            launch(C_MISC, k_gemv_t_finish<T>, grid, (const T*)GT_.p, b_nch_, npad, sv, Pz, BsH_.p, n, parts_.p, (int)SL_SCRATCH);
        }
        launch(C_MISC, k_dot<T>, grid, (const T*)BHy_.p, sv, T(1), n, parts_.p, (int)SL_AUX);
        launch(C_MISC, k_dot<T>, grid, sv, sv, T(1), n, parts_.p, (int)SL_AUX + 1);
        slot_n[SL_AUX] = slot_n[SL_AUX + 1] = grid;
        gather(SL_AUX, 2, 0u);
        auto v = collect({SL_AUX, SL_AUX + 1}, 0u);
        const T hys = T(v[0]), ss = T(v[1]);
        if (!(ss > T(0))) return;
        const T delta = hys / ss;
        T theta = T(1);
        if (std::abs(delta) < broyden_theta_bar_) {
            const T sg = delta >= T(0) ? T(1) : T(-1);
            theta = (T(1) - sg * broyden_theta_bar_) / (T(1) - delta);
        }
        const T denom = (T(1) / theta - T(1)) * ss + hys;
        if (denom == T(0) || denom != denom) return;
        launch(C_MISC, k_rank1_update<T>, (int)std::min<int64_t>(PSTRIDE, (n * n + BLOCK - 1) / BLOCK), HB_.p, sv,
               (const T*)BHy_.p, (const T*)BsH_.p, T(1) / denom, n);
    }
    // Anderson: a = (Y'Y)^-1 Y'v by elimination with complete pivoting on the Gram matrix (pivots below 1e-14 of the
    // largest are treated as a rank deficiency: their coefficient is zero)
    void anderson_coefficients(int m, const double* w, double* a) const {
        double A[CM * CM], b[CM];
        int perm[CM];
        for (int i = 0; i < m; ++i) { b[i] = w[i]; perm[i] = i; for (int j = 0; j < m; ++j) A[i * CM + j] = Gyy[i * CM + j]; }
        double amax = 0.0;
        for (int i = 0; i < m; ++i) amax = std::max(amax, std::abs(A[i * CM + i]));
        int rank = 0;
        for (int k = 0; k < m; ++k) {
            int pi = k, pj = k;
            double best = 0.0;
            for (int i = k; i < m; ++i)
                for (int j = k; j < m; ++j)
                    if (std::abs(A[i * CM + j]) > best) { best = std::abs(A[i * CM + j]); pi = i; pj = j; }
            if (!(best > 1e-14 * amax)) break;
            if (pi != k) { for (int j = 0; j < m; ++j) std::swap(A[k * CM + j], A[pi * CM + j]); std::swap(b[k], b[pi]); }
            if (pj != k) { for (int i = 0; i < m; ++i) std::swap(A[i * CM + k], A[i * CM + pj]); std::swap(perm[k], perm[pj]); }
            for (int i = k + 1; i < m; ++i) {
                const double f = A[i * CM + k] / A[k * CM + k];
                for (int j = k; j < m; ++j) A[i * CM + j] -= f * A[k * CM + j];
                b[i] -= f * b[k];
            }
            rank = k + 1;
        }
        double z[CM] = {0};
        for (int k = rank - 1; k >= 0; --k) {
            double acc = b[k];
            for (int j = k + 1; j < rank; ++j) acc -= A[k * CM + j] * z[j];
            z[k] = acc / A[k * CM + k];
        }
        for (int i = 0; i < m; ++i) a[i] = 0.0;
        for (int k = 0; k < rank; ++k) a[perm[k]] = z[k];
    }
    // compact form: Gram products of the stored pairs in logical order (oldest first), CM x CM
    bool compact_ok = false;
    int gm = 0;
    double Gsy[CM * CM], Gyy[CM * CM];
    // p = S'(-res), w = Y'(-res) at the current state (logical order, oldest first), when the accepted
    // trial delivered them (k_fused_compact): the next application then needs no reduction pass at all
    double p_new_ = 0.0, w_new_ = 0.0;      // <s_new, -res>, <y_new, -res> of the candidate pair
    bool pw_valid = false;
    double hp_[CM] = {0}, hw_[CM] = {0};

    // ---- gated pre-launch of the next iteration's one-pass kernel (see GateRec in bz_kernels.h) ----
    struct GatePlan {                        // everything the launch needs except the coefficients and the z store
        const T* S[CM];
        const T* x;
        T* xd;
        double gam0, gamma;
        int uni, gfc, fam, m_now;
        bool nt, table;
        bool operator==(const GatePlan& o) const {
            for (int i = 0; i < CM; ++i) if (S[i] != o.S[i]) return false;
            return x == o.x && xd == o.xd && gam0 == o.gam0 && gamma == o.gamma && uni == o.uni && gfc == o.gfc &&
                   fam == o.fam && m_now == o.m_now && nt == o.nt && table == o.table;
        }
    };
    // The early launch goes on the OTHER of two streams, so that it is dispatched (and, registers permitting, resident)
    // while the current pass still runs instead of queueing behind the read-back kernel; when it is released the
    // solver's launches move over to that stream (everything on the old one has completed by then: the host has the
    // read-back's scalars in hand).  Between library calls the solver is always back on the context's stream.
    hipStream_t cur_ = nullptr, gate_stream_ = nullptr, gate_on_ = nullptr;
    GateRec* gate_host_ = nullptr;           // pinned host memory
    GateRec* gate_host_dev_ = nullptr;       // ... its device address
    DBuf<GateRec> gate_dev_;
    unsigned long long gate_seq_ = 0;
    bool gate_pending_ = false, more_coming_ = false;
    int gate_env_ = 1;
    bool gate_broken_ = false;
    int gate_sabotage_ = 0;                  // (test) the n-th release is withheld: the launch must time out at its gate
    int64_t n_gate_fallbacks_ = 0;
    GatePlan gate_plan_{};
    double gate_bytes_ = 0.0;
    int64_t n_gated_ = 0, n_gate_aborts_ = 0;
    void gate_alloc() {
        if (gate_host_) return;
        BZ_HIP(hipHostMalloc((void**)&gate_host_, sizeof(GateRec), hipHostMallocMapped));
        std::memset(gate_host_, 0, sizeof(GateRec));
        BZ_HIP(hipHostGetDevicePointer((void**)&gate_host_dev_, gate_host_, 0));
        gate_dev_.alloc(1);
        BZ_HIP(hipStreamCreateWithFlags(&gate_stream_, hipStreamNonBlocking));
    }
    // back on the context's stream, nothing outstanding on the other one (end of every library-run loop)
    void gate_quiesce() {
        gate_abort();
        if (cur_ != ctx->stream) { BZ_HIP(hipStreamSynchronize(cur_)); cur_ = ctx->stream; }
    }
    // the plan of the iterate-history launch at ring position xc_ with m_now stored pairs (false: that form does not apply)
    bool gate_make_plan(int xc_, int m_now, const double* gring, int xr_run, GatePlan& pl) const {
        const int fam = fused_family();
        static const int spec_env = std::getenv("BZ_SPEC") ? std::atoi(std::getenv("BZ_SPEC")) : 1;
        static const int off32_env = std::getenv("BZ_OFF32") ? std::atoi(std::getenv("BZ_OFF32")) : 1;
        const int nt_env = nt_env_;
        const bool small = off32_env && (double)vcap * sizeof(T) < 4.0e9;
        if (!(xr_env_ >= 2 && small && fam >= 0 && xr_run >= m_now)) return false;
        for (int i = 1; i < m_now; ++i)
            if (gring[(xc_ - m_now + i + NXR) % NXR] != (double)gamma) return false;
        std::memset(&pl, 0, sizeof(pl));
        for (int i = 0; i < CM; ++i) {
            const int slot = (xc_ - std::max(0, m_now - i) + NXR) % NXR;
            pl.S[i] = X_[slot].p;
            if (i == 0) pl.gam0 = gring[slot];
        }
        pl.x = X_[xc_].p; pl.xd = X_[(xc_ + 1) % NXR].p;
        pl.gamma = (double)gamma; pl.uni = uni_; pl.fam = fam; pl.m_now = m_now;
        pl.gfc = gfc_env_ > 0 ? std::min(grid, gfc_env_ * std::max(1, num_cus)) : std::min(grid, std::max(1, num_cus));
        const int streams = (m_now + 1) + pstreams(true, true, true) + 1;      // (uniform penalties / zero multipliers are not streams)
        pl.nt = nt_env >= 0 ? nt_env != 0 : (double)n * sizeof(T) * streams > 340e6;
        pl.table = !(spec_env && fam == FAM_HEADLINE) || famrt_env_;
        return true;
    }
    void gate_launch(const GatePlan& pl, CompactCoef<CM> C2, T* zarg, bool trial_unused = false) {
        CompactVecs<T, CM> XV;
        XV.m = CM;
        for (int i = 0; i < CM; ++i) { XV.S[i] = pl.S[i]; XV.Y[i] = nullptr; }
        C2.gam0 = pl.gam0;
        const T gam = (T)pl.gamma;
        if (pl.table) {
            C2.uni_rt = pl.uni; C2.trial_rt = 0;
            // the plain pass with uniform penalties has compile-time instantiations (fp64); run-time UNI / TRIAL otherwise
            static const int famct_env = std::getenv("BZ_FAMCT") ? std::atoi(std::getenv("BZ_FAMCT")) : 1;
            FusedFn<T> fn = (famct_env && !famrt_env_) ? family_kernel<T>(pl.fam, pl.nt, pl.uni) : nullptr;
            const bool ct = fn != nullptr;
            if (!fn) fn = family_kernel<T>(pl.fam, pl.nt, -1);
            if (!fn) throw Error(BZ_ERR_STATE, "no one-pass kernel instantiation for this oracle family");
            form_[C_FUSED_IT] = std::string("k_fused_compact<XR=2,UNI=") + (ct ? std::to_string(pl.uni) : std::string("-1")) + ",NT=" +
                                std::to_string(pl.nt ? 1 : 0) + ",TRIAL=" + (ct ? "0" : "-1") + ",FAM=" + std::to_string(pl.fam) + ">";
            launch(C_FUSED_IT, fn, pl.gfc, XV, C2, pl.x, (const T*)nullptr, P, gam, pl.xd, zarg, (T*)nullptr, (T*)nullptr,
                   (T*)nullptr, n, parts_.p, (int)SL_TRIAL);
        } else {
            form_[C_FUSED_IT] = std::string("k_fused_compact<XR=2,UNI=") + char('0' + pl.uni) + (pl.nt ? ",NT=1" : ",NT=0") + ",TRIAL=0>";
#define BZ_LAUNCH_G(NT_, UNI_)                                                                                    \
    launch(C_FUSED_IT, k_fused_compact<T, CM, NT_, true, true, 2, UNI_>, pl.gfc, XV, C2, pl.x, (const T*)nullptr, P, gam, \
           pl.xd, zarg, (T*)nullptr, (T*)nullptr, (T*)nullptr, n, parts_.p, (int)SL_TRIAL)
            if (pl.nt) { if (pl.uni == 2) BZ_LAUNCH_G(true, 2); else if (pl.uni == 1) BZ_LAUNCH_G(true, 1); else BZ_LAUNCH_G(true, 0); }
            else { if (pl.uni == 2) BZ_LAUNCH_G(false, 2); else if (pl.uni == 1) BZ_LAUNCH_G(false, 1); else BZ_LAUNCH_G(false, 0); }
#undef BZ_LAUNCH_G
        }
    }
    // launch the NEXT iteration's pass now, gated on the host record
    void gate_prelaunch(const GatePlan& pl) {
        gate_alloc();
        CompactCoef<CM> C2;
        std::memset(&C2, 0, sizeof(C2));
        static const int glate_env = std::getenv("BZ_GATELATE") ? std::atoi(std::getenv("BZ_GATELATE")) : 1;
        C2.gate_late = glate_env;
        // (BZ_GATE_SPIN: the poll bounds, for the test of the fall-back)
        static const unsigned spin_env = std::getenv("BZ_GATE_SPIN") ? (unsigned)std::atoll(std::getenv("BZ_GATE_SPIN")) : 0u;
        C2.gate_spin_host = spin_env ? spin_env : GATE_SPIN_HOST;
        C2.gate_spin_dev = spin_env ? 8u * spin_env : GATE_SPIN_DEV;
        C2.gate_seq = ++gate_seq_; C2.gate_host = gate_host_dev_; C2.gate_dev = gate_dev_.p; C2.gate_timeout = ptimeout_dev_;
        const int streams = (pl.m_now + 1) + pstreams(true, true, true) + 1;
        mv(streams);
        gate_bytes_ = pending_bytes_;
        hipStream_t here = cur_;
        gate_on_ = gate_env_ != 2 ? cur_ : ((cur_ == ctx->stream) ? gate_stream_ : ctx->stream);
        C2.gate_other_stream = gate_on_ != cur_ ? 1 : 0;
        cur_ = gate_on_;
        try {
            gate_launch(pl, C2, (T*)nullptr);
        } catch (...) {
            cur_ = here;
            throw;
        }
        cur_ = here;
        gate_plan_ = pl; gate_pending_ = true;
    }
    void gate_release(const CompactCoef<CM>& C, T* zstore) {
        if (gate_sabotage_ > 0 && --gate_sabotage_ == 0) {      // (test) forget this release: the launch times out at its gate
            cur_ = gate_on_; gate_pending_ = false; ++n_gated_;
            return;
        }
        // 12 values (u1, u2h, H0, the z address) as tagged half-words: any order, each word validates itself
        double vals[13] = {0};
        for (int i = 0; i < CM; ++i) { vals[i] = C.u1[i]; vals[5 + i] = C.u2h[i]; }
        vals[10] = C.H0;
        const unsigned long long zbits = (unsigned long long)(uintptr_t)zstore;
        std::memcpy(&vals[11], &zbits, sizeof(double));
        const unsigned long long tag = (unsigned long long)ll_tag(gate_seq_) << 32;
        volatile unsigned long long* w = gate_host_->w;
        for (int i = 0; i < 13; ++i) {
            unsigned long long bits;
            std::memcpy(&bits, &vals[i], sizeof(bits));
            w[2 * i] = tag | (bits & 0xFFFFFFFFull);
            w[2 * i + 1] = tag | (bits >> 32);
        }
        std::atomic_thread_fence(std::memory_order_release);
        cur_ = gate_on_;                         // the rest of this iteration queues behind the released pass
        gate_pending_ = false; ++n_gated_;
        if (zstore) bytes_all_[C_FUSED_IT] += (double)n * sizeof(T);
    }
    void gate_abort() {
        if (!gate_pending_) return;
        *(volatile unsigned long long*)&gate_host_->w[31] = ((unsigned long long)ll_tag(gate_seq_) << 32) | 1ull;
        std::atomic_thread_fence(std::memory_order_release);
        gate_pending_ = false; ++n_gate_aborts_;
        bytes_all_[C_FUSED_IT] -= gate_bytes_; launches_all_[C_FUSED_IT] -= 1;      // (it left without moving anything)
    }
    bool prof_would_pick(int cat) const {
        return ((prof_mask >> cat) & 1u) && (prof_count[cat] % prof_period) == 0;
    }

    // profiling
    struct ProfRec { int cat; hipEvent_t a, b; double bytes; };
    unsigned prof_mask = 0, prof_period = 1;
    // Bytes every launch is DESIGNED to move (its read + write streams x their length x sizeof(T)), noted by mv()
    // right before the launch: counted for every launch of a category (bytes_all_) and for the launches that
    // carry timing events (bytes_timed_), so that a sustained rate is moved bytes / measured time of the SAME
    // launches whatever mix of kernel forms ran.  form_[cat]: template form of the category's last launch.
    double pending_bytes_ = 0.0;
    const char* pending_form_ = nullptr;
    void nm(const char* kernel_name) { pending_form_ = kernel_name; }
    double bytes_all_[BZ_NUM_KERNEL_CATEGORIES] = {0}, bytes_timed_[BZ_NUM_KERNEL_CATEGORIES] = {0};
    int64_t launches_all_[BZ_NUM_KERNEL_CATEGORIES] = {0};
    std::string form_[BZ_NUM_KERNEL_CATEGORIES];
    void mv(double passes, int64_t len = -1) { pending_bytes_ += passes * (double)(len < 0 ? n : len) * sizeof(T); }
    // parameter vectors an element-wise kernel streams (load_params)
    int pstreams(bool need_f, bool need_al, bool need_g) const {
        int k = 0;
        if (need_f && P.f_kind == BZ_F_DIAG_QUADRATIC) k += 2;
        if (need_al) k += 2 - (P.uni >= 1 ? 1 : 0) - (P.uni >= 2 ? 1 : 0) + (P.D_lo_vec ? 1 : 0) + (P.D_hi_vec ? 1 : 0);
        if (need_g) k += (P.g_u && (P.g_kind == BZ_G_NORM_L1_BOX || P.g_kind == BZ_G_NORM_L0_BOX || P.g_kind == BZ_G_NORM_LP_BOX) ? 1 : 0) +
                         (P.g_lo_vec ? 1 : 0) + (P.g_hi_vec ? 1 : 0);
        return k;
    }
    void account(int cat, ProfRec* r) {
        bytes_all_[cat] += pending_bytes_; launches_all_[cat] += 1;
        if (r) r->bytes = pending_bytes_;
        pending_bytes_ = 0.0;
        if (pending_form_) { form_[cat] = pending_form_; pending_form_ = nullptr; }
    }
    unsigned prof_count[BZ_NUM_KERNEL_CATEGORIES] = {0};
    bool prof_pick(int cat) {
        if (!((prof_mask >> cat) & 1u)) return false;
        return (prof_count[cat]++ % prof_period) == 0;
    }
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[BZ_NUM_KERNEL_CATEGORIES] = {0};
    int64_t prof_n[BZ_NUM_KERNEL_CATEGORIES] = {0};

    hipEvent_t get_event() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e; BZ_HIP(hipEventCreate(&e)); return e;
    }
    void drain_prof() {
        if (prof_recs.empty()) return;
        BZ_HIP(hipStreamSynchronize(ctx->stream));
        if (gate_stream_) BZ_HIP(hipStreamSynchronize(gate_stream_));
        for (auto& r : prof_recs) {
            float ms = 0; BZ_HIP(hipEventElapsedTime(&ms, r.a, r.b));
            prof_ms[r.cat] += ms; prof_n[r.cat] += 1; bytes_timed_[r.cat] += r.bytes;
            ev_pool.push_back(r.a); ev_pool.push_back(r.b);
        }
        prof_recs.clear();
    }

    template <class K, class... A> void launch(int cat, K kernel, int g, A... args) {
        launch_b(cat, kernel, g, BLOCK, args...);
    }
    template <class K, class... A> void launch_b(int cat, K kernel, int g, int block, A... args) {
        ProfRec r{cat, nullptr, nullptr, 0.0};
        const bool prof_on = prof_pick(cat);
        account(cat, prof_on ? &r : nullptr);
        if (prof_on) {
            // start/stop events bound to the dispatch itself: kernel time without the launch gap
            r.a = get_event(); r.b = get_event();
            hipExtLaunchKernelGGL(kernel, dim3(g), dim3(block), 0, cur_, r.a, r.b, 0, args...);
            prof_recs.push_back(r);
            if (prof_recs.size() > 8192) drain_prof();
        } else {
            hipLaunchKernelGGL(kernel, dim3(g), dim3(block), 0, cur_, args...);
        }
        BZ_HIP(hipGetLastError());
    }

    template <class K, class... A> void launch2d(int cat, K kernel, int gx, int gy, A... args) {
        ProfRec r{cat, nullptr, nullptr, 0.0};
        const bool prof_on = prof_pick(cat);
        account(cat, prof_on ? &r : nullptr);
        if (prof_on) {
            r.a = get_event(); r.b = get_event();
            hipExtLaunchKernelGGL(kernel, dim3(gx, gy), dim3(BLOCK), 0, cur_, r.a, r.b, 0, args...);
            prof_recs.push_back(r);
        } else {
            hipLaunchKernelGGL(kernel, dim3(gx, gy), dim3(BLOCK), 0, cur_, args...);
        }
        BZ_HIP(hipGetLastError());
    }

    void upload(DBuf<T>& dst, const void* src, int64_t cnt) {
        dst.alloc(cnt);
        copy_in(dst.p, src, cnt);
    }
    void copy_in(T* dst, const void* src, int64_t cnt) {
        if (!src) throw Error(BZ_ERR_ARG, "null input pointer");
        BZ_HIP(hipMemcpyAsync(dst, src, cnt * sizeof(T), hipMemcpyDefault, ctx->stream));
        BZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    void copy_out(void* dst, const T* src, int64_t cnt) {
        if (!dst) throw Error(BZ_ERR_ARG, "null output pointer");
        BZ_HIP(hipMemcpyAsync(dst, src, cnt * sizeof(T), hipMemcpyDefault, ctx->stream));
        BZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    void require_active() const {
        if (!active) throw Error(BZ_ERR_STATE, "no solve in progress (call bz_panoc_begin first)");
    }

    // ---- row-block-sharded stencil: halo rows through IPC-mapped fine-grained buffers -------------------
    // region layout: rows[parity][side][ny] of T (side 0 = north halo, written by the previous rank; side 1 =
    // south halo, written by the next rank), then flags[parity][side] (64-byte aligned)
    size_t halo_rows_bytes() const { return ((size_t)4 * desc.f_grid_ny * sizeof(T) + 63) / 64 * 64; }
    size_t halo_bytes() const { return halo_rows_bytes() + 64; }
    static T* halo_row(void* base, int par, int side, int64_t ny) { return (T*)base + (size_t)(par * 2 + side) * ny; }
    unsigned long long* halo_flag(void* base, int par, int side) const {
        return (unsigned long long*)((char*)base + halo_rows_bytes()) + (par * 2 + side);
    }
    bool sharded_stencil() const { return desc.f_kind == BZ_F_STENCIL5 && ctx->nranks > 1; }
   public:
    void halo_export(void* handle64) override {
        if (!sharded_stencil()) throw Error(BZ_ERR_STATE, "halo buffers exist only for a sharded Stencil5pt problem");
        BZ_HIP(hipSetDevice(ctx->device));
        if (!halo_local_) {
            BZ_HIP(hipExtMallocWithFlags(&halo_local_, halo_bytes(), hipDeviceMallocFinegrained));
            BZ_HIP(hipMemset(halo_local_, 0, halo_bytes()));
            BZ_HIP(hipDeviceSynchronize());
        }
        hipIpcMemHandle_t h;
        BZ_HIP(hipIpcGetMemHandle(&h, halo_local_));
        std::memcpy(handle64, &h, sizeof(h));
    }
    // handles of the previous and the next rank's halo regions (ignored at the two ends of the rank order)
    void halo_connect(const void* prev64, const void* next64) override {
        if (!halo_local_) throw Error(BZ_ERR_STATE, "bz_problem_halo_export must be called first");
        BZ_HIP(hipSetDevice(ctx->device));
        auto open = [&](const void* h64, void** out) {
            if (!h64) throw Error(BZ_ERR_ARG, "missing neighbour halo handle");
            hipIpcMemHandle_t h;
            std::memcpy(&h, h64, sizeof(h));
            BZ_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
        };
        if (ctx->rank > 0 && !halo_prev_) open(prev64, &halo_prev_);
        if (ctx->rank + 1 < ctx->nranks && !halo_next_) open(next64, &halo_next_);
        halo_connected_ = true;
    }
   private:
    // exchange the boundary rows of v with the neighbours; the halo the stencil kernel then reads
    StencilHalo<T> halo_exchange(const T* v) {
        if (!sharded_stencil()) return StencilHalo<T>{nullptr, nullptr};
        if (!halo_connected_) throw Error(BZ_ERR_STATE, "sharded Stencil5pt: bz_problem_halo_connect has not been called");
        const int64_t gny = desc.f_grid_ny, rows = desc.f_grid_nx;
        const unsigned long long seq = ++hseq_;
        const int par = (int)(seq & 1ull);
        HaloArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        a.first_row = v; a.last_row = v + (size_t)(rows - 1) * gny;
        a.seq = seq; a.ny = gny; a.timeout = ptimeout_dev_;
        if (halo_prev_) { a.prev_south = halo_row(halo_prev_, par, 1, gny); a.prev_flag = halo_flag(halo_prev_, par, 1); }
        if (halo_next_) { a.next_north = halo_row(halo_next_, par, 0, gny); a.next_flag = halo_flag(halo_next_, par, 0); }
        a.my_north_flag = halo_flag(halo_local_, par, 0);
        a.my_south_flag = halo_flag(halo_local_, par, 1);
        launch_b(C_GATHER, k_halo_exchange<T>, 1, XBLOCK, a);
        return StencilHalo<T>{halo_prev_ ? halo_row(halo_local_, par, 0, gny) : nullptr,
                              halo_next_ ? halo_row(halo_local_, par, 1, gny) : nullptr};
    }
    // ---- row-sharded dense constraint: x replicated, A' yhat summed over the ranks --------------------
    // region layout: slots[parity][rank][npad] of T, then flags[parity][rank]
    // ---- dense constraint in ONE pass over A (k_dense_fused) ------------------------------------------
    bool dense_fused_broken_ = false;
    int64_t n_dense_fallbacks_ = 0, n_dense_onepass_ = 0;
    int df_kp_ = 0, df_G_ = 0, df_groups_ = 0;
    int64_t df_rpg_ = 0;
    unsigned long long df_seq_ = 1;
    DBuf<unsigned long long> df_mail_;
    // the plan for an ny x n matrix on this device, or df_kp_ = 0 when the kernel does not apply
    template <int KP_> static int dense_occupancy() {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_dense_fused<T, KP_>, FBLOCK, 0) != hipSuccess) {
            (void)hipGetLastError();
            occ = 0;
        }
        return occ;
    }
    void dense_fused_plan() {
        df_kp_ = 0;
        df_env_ = std::getenv("BZ_DENSE_FUSED") ? std::atoi(std::getenv("BZ_DENSE_FUSED")) : 1;
        df_sabotage_ = std::getenv("BZ_TEST_DENSE_TIMEOUT") ? std::atoi(std::getenv("BZ_TEST_DENSE_TIMEOUT")) : 0;
        df_spin_ = std::getenv("BZ_DENSE_SPIN") ? (unsigned)std::atoll(std::getenv("BZ_DENSE_SPIN")) : 0u;
        constexpr int N = PackN<T>::N;
        if (desc.c_kind != BZ_C_DENSE_AFFINE || x_replicated || slack || (n % N) != 0) return;
        if (desc.D_kind >= BZ_D_VC_PAIRS) return;
        const int64_t npk = n / N;
        // KP packs per row and lane: kp_full fills a CU's registers with ONE workgroup's ring of tiles — the fewest slices per
        // row group, so the fewest partners per exchange (narrower slices: 16 or 32 partners, and several workgroups per CU,
        // measured 20-60 % slower on cfg 4); the narrower forms serve matrices with few columns
        const int kp_full = std::is_same<T, float>::value ? 4 : 2;
        int kp = kp_full;
        if (const char* e = std::getenv("BZ_DENSE_KP")) kp = std::max(1, std::min(kp_full, std::atoi(e)));
        while (kp > 1 && kp != 2 && kp != 4) --kp;
        while (kp > 1 && (int64_t)FBLOCK * (kp / 2) >= npk) kp /= 2;       // (a matrix narrower than one slice: more row groups instead)
        while (kp < kp_full && (npk + (int64_t)FBLOCK * kp - 1) / ((int64_t)FBLOCK * kp) > FG_MAX) kp *= 2;
        const int64_t G = (npk + (int64_t)FBLOCK * kp - 1) / ((int64_t)FBLOCK * kp);
        // co-resident workgroups: what the runtime says fits a CU, at most what the launch bounds ask for (the groups' workgroups
        // wait for each other: a grid beyond the resident set would only ever time out)
        int occ = 0;
        if (kp == 1) occ = dense_occupancy<1>();
        else if (kp == 2) occ = dense_occupancy<2>();
        else if constexpr (std::is_same<T, float>::value) occ = dense_occupancy<4>();
        const int64_t slots = (int64_t)num_cus * std::min(occ, kp_full / kp);
        if (G > FG_MAX || G > slots) return;
        int64_t groups = std::max<int64_t>(1, std::min<int64_t>(slots / G, (ny + FT - 1) / FT));
        int64_t rpg = (ny + groups - 1) / groups;
        rpg = (rpg + FT - 1) / FT * FT;
        groups = (ny + rpg - 1) / rpg;
        if (rpg > FRC) return;                                          // (a group's row parameters must fit the kernel's LDS cache)
        df_kp_ = kp; df_G_ = (int)G; df_groups_ = (int)groups; df_rpg_ = rpg;
        df_mail_.alloc((size_t)groups * FMS * FG_MAX * FT * 2);
    }
    bool dense_fused_on() const { return df_env_ && df_kp_ > 0 && !dense_fused_broken_; }
    int df_env_ = 1, df_sabotage_ = 0;       // BZ_DENSE_FUSED / BZ_TEST_DENSE_TIMEOUT, read when the problem is created
    unsigned df_spin_ = 0;                   // BZ_DENSE_SPIN
    // gradient!'s dense part in one pass over A: c(point) -> CX_ (and cx_keep_), the row-group partials of A'yhat -> GT_,
    // the penalty partials -> slot_pen
    void dense_fused_launch(const T* x, int slot_pen) {
        DenseFusedArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        a.A = A_.p; a.x = x; a.b = cb_.p; a.cx = CX_.p; a.cx2 = cx_keep_; a.part = GT_.p; a.pstride = npad;
        a.ny = ny; a.n = n; a.G = df_G_; a.ngroups = df_groups_; a.rows_per_group = df_rpg_;
        a.seq0 = df_seq_; a.mail = df_mail_.p; a.timeout = ptimeout_dev_;
        a.spin = df_spin_ ? df_spin_ : FSPIN_LIMIT;
        a.parts = parts_.p; a.slot_pen = slot_pen;
        df_seq_ += (unsigned long long)(df_rpg_ / FT) + 2ull * FNB;      // (every step posts, the padding steps of the last ring turn too)
        // (test) slice 0 of every group posts under tags nobody waits for: the polls give up
        if (df_sabotage_ > 0 && ++dense_sabotage_count_ == df_sabotage_) a.sabotage = 1;
        if (df_sabotage_ == -2) a.sabotage = 2;      // (timing experiment: no exchange)
        // the matrix once; x, b and the penalty vectors over the rows; c(x) out; the row-group partials
        mv((double)ny, n); mv(1, n); mv(2 + pstreams(false, true, false) + (cx_keep_ ? 1 : 0), ny); mv(df_groups_, npad);
        const int g = df_groups_ * df_G_;
        if (df_kp_ == 1) { nm("k_dense_fused<KP=1>"); launch_b(C_GEMV, k_dense_fused<T, 1>, g, FBLOCK, a, P); }
        else if (df_kp_ == 2) { nm("k_dense_fused<KP=2>"); launch_b(C_GEMV, k_dense_fused<T, 2>, g, FBLOCK, a, P); }
        else if constexpr (std::is_same<T, float>::value) { nm("k_dense_fused<KP=4>"); launch_b(C_GEMV, k_dense_fused<T, 4>, g, FBLOCK, a, P); }
        else throw Error(BZ_ERR_STATE, "k_dense_fused: no instantiation");
        slot_n[slot_pen] = df_groups_;
        ++n_dense_onepass_;
    }
    int dense_sabotage_count_ = 0;
    bool x_replicated = false;
    DBuf<T> JL_;                     // this rank's partial of A' yhat
    void* ar_local_ = nullptr;
    void* ar_peer_[P2P_MAXRANKS] = {};
    bool ar_connected_ = false;
    DBuf<unsigned long long> ar_done_;
    unsigned long long ar_done_base_ = 0;
    unsigned long long arseq_ = 0;
    size_t ar_slots_bytes() const { return ((size_t)2 * ctx->nranks * npad * sizeof(T) + 63) / 64 * 64; }
    size_t ar_bytes() const { return ar_slots_bytes() + 2 * P2P_MAXRANKS * sizeof(unsigned long long); }
    T* ar_slot(void* base, int par, int r) const { return (T*)base + (size_t)(par * ctx->nranks + r) * npad; }
    unsigned long long* ar_flag(void* base, int par, int r) const {
        return (unsigned long long*)((char*)base + ar_slots_bytes()) + (par * P2P_MAXRANKS + r);
    }
   public:
    void allreduce_export(void* handle64) override {
        if (!x_replicated) throw Error(BZ_ERR_STATE, "all-reduce regions exist only for a row-sharded DenseAffine problem");
        BZ_HIP(hipSetDevice(ctx->device));
        if (!ar_local_) {
            BZ_HIP(hipExtMallocWithFlags(&ar_local_, ar_bytes(), hipDeviceMallocFinegrained));
            BZ_HIP(hipMemset(ar_local_, 0, ar_bytes()));
            BZ_HIP(hipDeviceSynchronize());
        }
        hipIpcMemHandle_t h;
        BZ_HIP(hipIpcGetMemHandle(&h, ar_local_));
        std::memcpy(handle64, &h, sizeof(h));
    }
    void allreduce_connect(const void* handles) override {
        if (!ar_local_) throw Error(BZ_ERR_STATE, "bz_problem_allreduce_export must be called first");
        if (!handles) throw Error(BZ_ERR_ARG, "null handles");
        BZ_HIP(hipSetDevice(ctx->device));
        for (int r = 0; r < ctx->nranks; ++r) {
            if (r == ctx->rank) { ar_peer_[r] = ar_local_; continue; }
            if (ar_peer_[r]) continue;
            hipIpcMemHandle_t h;
            std::memcpy(&h, (const char*)handles + (size_t)r * 64, 64);
            BZ_HIP(hipIpcOpenMemHandle(&ar_peer_[r], h, hipIpcMemLazyEnablePeerAccess));
        }
        ar_connected_ = true;
    }
   private:
    // sum JL_ over the ranks: returns the chunk array (nranks chunks of npad) k_gemv_t_finish then folds
    const T* allreduce_partials() {
        if (!ar_connected_) throw Error(BZ_ERR_STATE, "row-sharded DenseAffine: bz_problem_allreduce_connect has not been called");
        const unsigned long long seq = ++arseq_;
        const int par = (int)(seq & 1ull);
        VecXchgArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        a.local = JL_.p; a.seq = seq; a.npad = npad; a.nranks = ctx->nranks; a.timeout = ptimeout_dev_;
        for (int r = 0; r < ctx->nranks; ++r) {
            a.peer_slot[r] = ar_slot(ar_peer_[r], par, ctx->rank);
            a.peer_flag[r] = ar_flag(ar_peer_[r], par, ctx->rank);
        }
        a.my_flags = ar_flag(ar_local_, par, 0);
        const int64_t packs = npad / PackN<T>::N;
        const int g = (int)std::max<int64_t>(1, std::min<int64_t>(64, (packs + BLOCK - 1) / BLOCK));
        if (!ar_done_.p) ar_done_.alloc(16);
        ar_done_base_ += (unsigned long long)g;
        a.done = ar_done_.p; a.target = ar_done_base_;
        launch_b(C_GATHER, k_vec_allgather<T>, g, BLOCK, a);
        return ar_slot(ar_local_, par, 0);
    }

    void* halo_local_ = nullptr;
    void* halo_prev_ = nullptr;
    void* halo_next_ = nullptr;
    bool halo_connected_ = false;
    unsigned long long hseq_ = 0;

    // multi-GPU: fold this rank's block partials of slots [first, first+cnt) and all-gather
    // z of the CURRENT state into Z_[zc] if the last fused pass skipped its store:
    //   z = prox_{gamma g}(x - gamma grad L(x))   — the arithmetic of k_algrad_elem + k_fbstep, which the fused
    //   passes reproduce bit for bit (test_fused_equals_generic_bitwise)
    // (likewise the residual res = x - z into RES_[rc] when the pass did not store it)
    // need_res = false: the caller reads z only (the solution of a subproblem, alps.jl:67): a z that the last pass stored is
    // enough, whatever the state of res (the iterate-history passes never write it)
    void ensure_z(bool need_res = true) {
        if (z_valid && (res_valid || !need_res)) return;
        const int fb_env = fused_begin_env_;
        if (fb_env && desc.c_kind == BZ_C_IDENTITY && !slack && !dense_f && !lp_g &&
            (desc.f_kind == BZ_F_ZERO || desc.f_kind == BZ_F_DIAG_QUADRATIC)) {
            mv(1 + pstreams(true, true, true) + 1 + (res_valid ? 0 : 1));
            launch(C_FB, k_zres_elem<T>, grid, (const T*)X_[xc].p, P, gamma, Z_[zc].p,
                   res_valid ? (T*)nullptr : RES_[rc].p, n);      // the two kernels below in one pass
        } else {
            algrad(X_[xc].p, D_.p, SL_AUX);                  // scratch gradient: GX_/GZ_ keep their meaning
            fbstep(X_[xc].p, D_.p, gamma, Z_[zc].p, res_valid ? (T*)nullptr : RES_[rc].p, SL_ZS);
        }
        z_valid = true; res_valid = true;
    }
    // ymask (row-sharded dense c only, where x-space quantities are computed in full by every rank): the
    // slots that are sums over THIS rank's constraint rows and must be added up; the others count once
    void gather(int first, int cnt, unsigned maxmask, unsigned ymask = 0u) {
        if (!ctx->multi()) return;
        const unsigned keepmask = x_replicated ? ymask : ~0u;
        if (ctx->p2p_on) {
            if (cnt > P2P_PACK) throw Error(BZ_ERR_ARG, "pack too large for the p2p mailbox");
            XchgArgs a;
            std::memset(&a, 0, sizeof(a));
            a.parts = parts_.p; a.first = first; a.cnt = cnt; a.maxmask = maxmask;
            for (int i = 0; i < cnt; ++i) a.counts.set(i, slot_n[first + i]);
            a.rank = ctx->rank; a.nranks = ctx->nranks; a.seq = ++ctx->xseq;
            a.recv = recv_.p + (size_t)first * ctx->nranks;
            a.mbox_local = (P2PWords*)ctx->mbox_local;
            for (int r = 0; r < ctx->nranks; ++r) a.mbox_peer[r] = (P2PWords*)ctx->mbox_peer[r];
            a.timeout = ptimeout_dev_; a.keepmask = keepmask;
            launch_b(C_GATHER, k_exchange, cnt, 64, a);
            for (int s = first; s < first + cnt; ++s) { grp_first[s] = first; grp_cnt[s] = cnt; }
            return;
        }
        if (cnt > 32) throw Error(BZ_ERR_ARG, "pack too large");
        SlotCounts counts;
        std::memset(&counts, 0, sizeof(counts));
        for (int i = 0; i < cnt; ++i) counts.set(i, slot_n[first + i]);
        launch_b(C_GATHER, k_pack, cnt, 64, (const double*)parts_.p, counts, first, cnt, maxmask, send_.p, ctx->rank,
                 keepmask);
        BZ_NCCL(ncclAllGather(send_.p + first, recv_.p + (size_t)first * ctx->nranks, cnt, ncclDouble,
                              ctx->comm, cur_));
        for (int s = first; s < first + cnt; ++s) { grp_first[s] = first; grp_cnt[s] = cnt; }
    }
    // the arguments of the fold + (p2p) exchange + read-back of slots [first, first + cnt): with mailboxes the exchange over
    // the ranks, without (one rank, no p2p context) the local fold alone
    XCollectArgs make_xcollect(int first, int cnt, unsigned maxmask) {
        if (cnt > P2P_PACK) throw Error(BZ_ERR_ARG, "pack too large for the p2p mailbox");
        XCollectArgs b;
        std::memset(&b, 0, sizeof(b));
        XchgArgs& a = b.x;
        a.parts = parts_.p; a.first = first; a.cnt = cnt; a.maxmask = maxmask;
        for (int i = 0; i < cnt; ++i) a.counts.set(i, slot_n[first + i]);
        a.rank = 0; a.nranks = 1;
        if (ctx->p2p_on) {
            a.rank = ctx->rank; a.nranks = ctx->nranks; a.seq = ++ctx->xseq;
            a.recv = recv_.p + (size_t)first * ctx->nranks;
            a.mbox_local = (P2PWords*)ctx->mbox_local;
            for (int r = 0; r < ctx->nranks; ++r) a.mbox_peer[r] = (P2PWords*)ctx->mbox_peer[r];
            for (int s = first; s < first + cnt; ++s) { grp_first[s] = first; grp_cnt[s] = cnt; }
        }
        a.timeout = ptimeout_dev_; a.keepmask = ~0u;
        b.host_out = host_out_dev_;
        b.ticket = ++collect_seq;
        return b;
    }
    // p2p transport: k_exchange and the read-back in one launch; returns the ticket wait_host must see
    unsigned long long exchange_collect(int first, int cnt, unsigned maxmask) {
        XCollectArgs b = make_xcollect(first, cnt, maxmask);
        launch_b(C_GATHER, k_exchange_collect, cnt, 64, b);
        return b.ticket;
    }
    ScalarSrc src(int slot) const {
        if (!ctx->multi()) return ScalarSrc{parts_.p + (size_t)slot * PSTRIDE, slot_n[slot], 1};
        const int f = grp_first[slot], c = grp_cnt[slot];
        return ScalarSrc{recv_.p + (size_t)f * ctx->nranks + (slot - f), ctx->nranks, c};
    }
    // fold the listed slots (bit i of maxmask: i-th listed slot is a max) and read them back
    std::vector<double> collect_range(int first, int cnt, unsigned maxmask) {
        CollectArgs a;
        a.n = cnt; a.maxmask = maxmask;
        for (int i = 0; i < cnt; ++i) a.src[i] = src(first + i);
        return collect_run(a);
    }
    std::vector<double> collect(std::initializer_list<int> slots, unsigned maxmask) {
        CollectArgs a;
        a.n = 0; a.maxmask = maxmask;
        for (int s : slots) {
            a.src[a.n] = src(s);
            ++a.n;
        }
        return collect_run(a);
    }
    // launch the read-back of slots [first, first + cnt) without waiting; wait_host(cnt, ticket) later
    unsigned long long collect_launch_range(int first, int cnt, unsigned maxmask) {
        CollectArgs a;
        a.n = cnt; a.maxmask = maxmask;
        for (int i = 0; i < cnt; ++i) a.src[i] = src(first + i);
        a.ticket = ++collect_seq;
        static const int wave_env = std::getenv("BZ_COLLECT_WAVE") ? std::atoi(std::getenv("BZ_COLLECT_WAVE")) : 1;
        bool unit = wave_env != 0;
        for (int i = 0; i < a.n; ++i) unit = unit && a.src[i].stride == 1;
        if (unit) launch_b(C_COLLECT, k_collect_w, a.n, 64, a, host_out_dev_);
        else launch(C_COLLECT, k_collect, a.n, a, host_out_dev_);
        return a.ticket;
    }
    std::vector<double> collect_run(CollectArgs& a) {
        a.ticket = ++collect_seq;
        static const int wave_env = std::getenv("BZ_COLLECT_WAVE") ? std::atoi(std::getenv("BZ_COLLECT_WAVE")) : 1;
        bool unit = wave_env != 0;
        for (int i = 0; i < a.n; ++i) unit = unit && a.src[i].stride == 1;
        if (unit) launch_b(C_COLLECT, k_collect_w, a.n, 64, a, host_out_dev_);
        else launch(C_COLLECT, k_collect, a.n, a, host_out_dev_);
        return wait_host(a.n, a.ticket);
    }
    // read n scalars from the pinned mailbox once both tagged words of each carry this read-back's tag
    std::vector<double> wait_host(int cnt, unsigned long long ticket) {
        struct { int n; } a{cnt};
        const unsigned long long tag = (unsigned long long)ll_tag(ticket) << 32;
        const unsigned long long himask = 0xFFFFFFFF00000000ull;
        // spin on the tickets in pinned host memory (a few microseconds after the kernel's stores land);
        // bounded: on a fault or a hang fall through to the blocking synchronisation, which reports it
        {
            volatile unsigned long long* ho = (volatile unsigned long long*)host_out_;
            const auto t_start = std::chrono::steady_clock::now();
            bool done = false;
            for (unsigned spin = 0; !done; ++spin) {
                done = true;
                for (int i = 0; i < a.n; ++i)
                    if ((ho[2 * i] & himask) != tag || (ho[2 * i + 1] & himask) != tag) { done = false; break; }
                if (!done && (spin & 0x3FFu) == 0x3FFu) {
                    if (!gate_pending_ && hipStreamQuery(cur_) != hipErrorNotReady) break;     // finished or failed
                    if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(30)) break;
                }
            }
            bool complete = done;
            if (!done) {
                gate_abort();
                BZ_HIP(hipStreamSynchronize(cur_));
                std::atomic_thread_fence(std::memory_order_acquire);
                // everything queued has run: the scalars are there now, or they will never be (a device poll that gave
                // up is reported below; anything else must not be read as numbers)
                complete = true;
                for (int i = 0; i < a.n; ++i)
                    if ((ho[2 * i] & himask) != tag || (ho[2 * i + 1] & himask) != tag) { complete = false; break; }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (!complete && !*ptimeout_)
                throw Error(BZ_ERR_HIP, "a read-back did not arrive although the stream is idle (no kernel posted these scalars)");
        }
        if (*ptimeout_) {
            const int code = *ptimeout_;
            *ptimeout_ = 0;
            if (code == 1 && ctx->nranks == 1) throw PersistTimeout();
            if (code == 6 || code == 7) throw GateTimeout(code);
            if (code == 8) {      // (whoever can redo its work does; every later evaluation takes the two-kernel form)
                dense_fused_broken_ = true; ++n_dense_fallbacks_;
                throw DenseFusedTimeout();
            }
            throw Error(code == 1 ? BZ_ERR_HIP : BZ_ERR_COMM,
                        code == 1 ? "persistent two-loop kernel: grid barrier timed out (blocks not co-resident?)"
                        : code == 2 ? "p2p scalar exchange timed out waiting for a peer rank"
                        : code == 4 ? "stencil halo exchange timed out waiting for a neighbour rank"
                        : code == 5 ? "dense-constraint all-reduce timed out waiting for a peer rank"
                        : code == 6 ? "a pre-launched pass timed out at its gate (the host never released it)"
                        : code == 7 ? "a pre-launched pass: workgroups timed out waiting for workgroup 0 to open the gate"
                                    : "persistent two-loop kernel: p2p phase exchange timed out waiting for a peer rank");
        }
        std::vector<double> out(a.n);
        const unsigned long long* hw = (const unsigned long long*)host_out_;
        for (int i = 0; i < a.n; ++i) {
            const unsigned long long bits = (hw[2 * i] & 0xFFFFFFFFull) | (hw[2 * i + 1] << 32);
            std::memcpy(&out[i], &bits, sizeof(double));
        }
        return out;
    }

    // forward-backward step kernel; the Newton/pow prox kinds use their own instantiation so the
    // common kinds keep their register budget
    void fbstep(const T* x, const T* g, T gam, T* z, T* res, int slot0) {
        if (generic_) { fbstep_generic(x, g, gam, z, res, slot0); return; }
        if (slack) {
            for (int k = 0; k < 3; ++k) slot_n[slot0 + k] = grid_y;
            mv(2 * (1 + (g ? 1 : 0) + 1 + (res ? 1 : 0)) + pstreams(false, false, true) + (P.D_lo_vec ? 1 : 0) + (P.D_hi_vec ? 1 : 0), nx);
            if (lp_g) launch(C_FB, k_fbstep_slack<T, true>, grid_y, x, g, gam, P, z, res, nx, parts_.p, slot0);
            else launch(C_FB, k_fbstep_slack<T, false>, grid_y, x, g, gam, P, z, res, nx, parts_.p, slot0);
            return;
        }
        for (int k = 0; k < 3; ++k) slot_n[slot0 + k] = grid;
        mv(1 + (g ? 1 : 0) + 1 + (res ? 1 : 0) + pstreams(false, false, true));
        if (lp_g) launch(C_FB, k_fbstep<T, true>, grid, x, g, gam, P, z, res, n, parts_.p, slot0);
        else launch(C_FB, k_fbstep<T, false>, grid, x, g, gam, P, z, res, n, parts_.p, slot0);
    }
    T al_value(double fsum, double pensum) const {   // auglagfun.jl:78,81-82
        T lx = T(0.5) * T(pensum);
        lx += f_value(fsum);
        lx -= musqy;
        return lx;
    }
    T f_value(double fsum) const { return fscale == T(1) ? T(fsum) : fscale * T(fsum); }
    T g_value(double gsum) const {
        switch (desc.g_kind) {
        case BZ_G_NORM_L1: case BZ_G_NORM_L1_NONNEG: case BZ_G_NORM_L1_BOX: case BZ_G_NORM_L0_BOX:
        case BZ_G_NORM_LP_NONNEG: case BZ_G_NORM_LP_BOX:
            return P.g_lambda * T(gsum);
        case BZ_G_CALLBACK: return T(gsum);      // the callback returned g(z) itself
        default: return T(0);
        }
    }

    // gradient!(dlx, al, x) on the device; partials -> slot0 (f terms), slot0+1 (t^2/mu)
    // row chunks of the transposed product: enough blocks to fill the chip, fixed summation order
    void plan_chunks(int64_t rows, int& rpc, int& nch) const {
        const int64_t colblocks = std::max<int64_t>(1, (n / PackN<T>::N + BLOCK - 1) / BLOCK);
        const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(rows, (2048 + colblocks - 1) / colblocks));
        rpc = (int)((rows + chunks - 1) / chunks);
        nch = (int)((rows + rpc - 1) / rpc);
    }
    // out = M p - b (b may be null): M[rows][n] row-major
    void gemv_rows(const T* M, int64_t rows, const T* p, const T* b, T* out) {
        mv((double)rows, n); mv(b ? 2 : 1, rows); mv(1, n);      // the matrix once, x, b and the result
        nm("k_gemv_n");
        launch(C_GEMV, k_gemv_n<T>, (int)std::min<int64_t>(rows, 65535), M, p, b, out, rows, n);
    }
    // GT_ partials of M' v over row chunks.  fp32 with n % 64 == 0 runs on the matrix cores
    // (v_mfma_f32_16x16x4_f32), everything else on the vector ALUs; both are bound by the bytes of M.
    void gemv_cols(const T* M, int64_t rows, const T* v, int rpc, int nch) {
        if constexpr (std::is_same<T, float>::value) {
            if (n % 64 == 0 && !getenv("BZ_GEMV_VALU")) {
                mv((double)rows, n); mv(1, rows); mv(nch, npad);      // the matrix once, v, the row-chunk partials
                nm("k_gemv_t_mfma");
                launch2d(C_GEMV_MFMA, k_gemv_t_mfma, (int)((n / 64 + WAVES - 1) / WAVES), nch, (const float*)M,
                         (const float*)v, (float*)GT_.p, rows, n, rpc, npad);
                return;
            }
        }
        const bool aligned = (n % PackN<T>::N) == 0;
        const int colblocks = (int)(((aligned ? n / PackN<T>::N : n) + BLOCK - 1) / BLOCK);
        mv((double)rows, n); mv(1, rows); mv(nch, npad); nm("k_gemv_t");
        launch2d(C_GEMV, k_gemv_t<T>, colblocks, nch, M, v, GT_.p, rows, n, rpc, npad);
    }
    // eval!(cx, c, x) for the dense constraint
    void eval_c(const T* x) { gemv_rows(A_.p, ny, x, cb_.p, CX_.p); }
    // dense f: leaves what k_algrad_elem / k_fvalue_elem need in FR_ / DFX_ and the f partials in slot0
    //   LeastSquares: r = A x - b ; slot0 <- <r,r> ; DFX = A' r        (ProximalOperators: 0.5||Ax-b||^2)
    //   Quadratic:    FR = Q x (value and gradient finished element-wise)
    void dense_f_eval(const T* x, int slot0, bool need_grad) {
        if (desc.f_kind == BZ_F_LEAST_SQUARES) {
            gemv_rows(FA_.p, frows, x, fb_.p, FR_.p);
            const int gm = (int)std::min<int64_t>(grid, std::max<int64_t>(1, (frows / PackN<T>::N + BLOCK) / BLOCK));
            mv(1, frows);
            launch(C_MISC, k_dot<T>, gm, (const T*)FR_.p, (const T*)FR_.p, T(1), frows, parts_.p, slot0);
            slot_n[slot0] = gm;
            if (need_grad) {
                gemv_cols(FA_.p, frows, FR_.p, f_rows_per_chunk, f_nrowchunks);
                ElemParams<T> Pz = P;
                Pz.f_kind = BZ_F_ZERO;
                mv(f_nrowchunks + 2);
                launch(C_MISC, k_gemv_t_finish<T>, grid, (const T*)GT_.p, f_nrowchunks, npad, x, Pz, DFX_.p, n,
                       parts_.p, (int)SL_SCRATCH);
            }
        } else {
            gemv_rows(FA_.p, n, x, (const T*)nullptr, FR_.p);
        }
    }
    void algrad(const T* x, T* grad, int slot0) {
        if (generic_) { algrad_generic(x, grad, slot0); return; }
        if (desc.c_kind == BZ_C_DENSE_AFFINE && dense_fused_on()) {
            // one pass over A: c(x), yhat and the row-group partials of A'yhat (k_dense_fused), then the fold + f terms
            slot_n[slot0] = grid;
            dense_fused_launch(x, slot0 + 1);
            mv(df_groups_ + 2 + pstreams(true, false, false));
            launch(C_MISC, k_gemv_t_finish<T>, grid, (const T*)GT_.p, df_groups_, npad, x, P, grad, n, parts_.p, slot0);
            gather(slot0, 2, 0u, 2u);
            return;
        }
        if (desc.c_kind == BZ_C_DENSE_AFFINE) {
            eval_c(x);                                                        // cx = A x - b
            if (cx_keep_) BZ_HIP(hipMemcpyAsync(cx_keep_, CX_.p, ny * sizeof(T), hipMemcpyDeviceToDevice, cur_));
            mv(2 + pstreams(false, true, false), ny);
            launch(C_MISC, k_yupd<T>, grid_y, (const T*)CX_.p, P, YU_.p, ny, parts_.p, slot0 + 1);
            slot_n[slot0] = grid; slot_n[slot0 + 1] = grid_y;
            gemv_cols(A_.p, ny, YU_.p, rows_per_chunk, nrowchunks);           // jtv = A' yupd (row-chunk partials)
            if (x_replicated) {
                // this rank's rows only: fold the chunks, sum over the ranks (rank order), then finish
                ElemParams<T> Pz = P;
                Pz.f_kind = BZ_F_ZERO;
                mv(nrowchunks + 2);
                launch(C_MISC, k_gemv_t_finish<T>, grid, (const T*)GT_.p, nrowchunks, npad, x, Pz, JL_.p, n, parts_.p,
                       (int)SL_SCRATCH);
                const T* chunks = allreduce_partials();
                mv(ctx->nranks + 2 + pstreams(true, false, false));
                launch(C_MISC, k_gemv_t_finish<T>, grid, chunks, ctx->nranks, npad, x, P, grad, n, parts_.p, slot0);
            } else {
                mv(nrowchunks + 2 + pstreams(true, false, false));
                launch(C_MISC, k_gemv_t_finish<T>, grid, (const T*)GT_.p, nrowchunks, npad, x, P, grad, n, parts_.p, slot0);
            }
            gather(slot0, 2, 0u, 2u);         // slot0: f terms (x-space) ; slot0 + 1: penalty terms (this rank's rows)
            return;
        }
        slot_n[slot0] = slot_n[slot0 + 1] = grid;
        if (slack) {
            slot_n[slot0] = slot_n[slot0 + 1] = grid_y;
            mv(2 + pstreams(true, false, false) + 3 + 2, nx);
            launch(C_ALGRAD, k_algrad_slack_elem<T>, grid_y, x, P, (const T*)ymul_.p, grad, nx, parts_.p, slot0);
            gather(slot0, 2, 0u);
            return;
        }
        if (dense_f) {
            dense_f_eval(x, slot0, true);
            if (desc.f_kind == BZ_F_LEAST_SQUARES)
                { mv(3 + pstreams(false, true, false)); launch(C_ALGRAD, k_algrad_elem<T>, grid, x, P, grad, n, parts_.p, slot0, 1, (const T*)DFX_.p); }
            else
                { mv(4 + pstreams(false, true, false)); launch(C_ALGRAD, k_algrad_elem<T>, grid, x, P, grad, n, parts_.p, slot0, 2, (const T*)FR_.p); }
        } else if (desc.f_kind == BZ_F_STENCIL5) {
            mv(3 + pstreams(false, true, false));      // x (its north / south / west / east re-reads are cache hits), b, grad
            launch(C_ALGRAD, k_algrad_stencil<T>, grid, x, P, (int64_t)desc.f_grid_nx, (int64_t)desc.f_grid_ny, 0,
                   grad, n, parts_.p, slot0, halo_exchange(x));
        } else {
            mv(2 + pstreams(true, true, false));
            launch(C_ALGRAD, k_algrad_elem<T>, grid, x, P, grad, n, parts_.p, slot0, 0, (const T*)nullptr);
        }
        gather(slot0, 2, 0u);
    }
    // f(x) alone (alps.jl:39): partial sums -> slot0
    void fvalue(const T* x, int slot0) {
        if (generic_) {      // f(x) through the gradient callback (the reference's generic f(x) needs no more)
            copy_out(hx_.data(), x, n);
            fill_slot(slot0, cb_f_gradient(hx_.data(), hdfx_.data(), n));
            return;
        }
        slot_n[slot0] = grid;
        if (slack) {      // f on the x part only
            slot_n[slot0] = grid_y;
            mv(1 + pstreams(true, false, false), nx);
            launch(C_MISC, k_fvalue_elem<T>, grid_y, x, P, nx, parts_.p, slot0, (const T*)nullptr);
            gather(slot0, 1, 0u);
            return;
        }
        if (dense_f) {
            dense_f_eval(x, slot0, false);
            if (desc.f_kind == BZ_F_QUADRATIC)
                { mv(3); launch(C_MISC, k_fvalue_elem<T>, grid, x, P, n, parts_.p, slot0, (const T*)FR_.p); }
        } else if (desc.f_kind == BZ_F_STENCIL5) {
            mv(2);
            launch(C_MISC, k_algrad_stencil<T>, grid, x, P, (int64_t)desc.f_grid_nx, (int64_t)desc.f_grid_ny, 1,
                   (T*)nullptr, n, parts_.p, slot0, halo_exchange(x));
        } else {
            mv(1 + pstreams(true, false, false));
            launch(C_MISC, k_fvalue_elem<T>, grid, x, P, n, parts_.p, slot0, (const T*)nullptr);
        }
        gather(slot0, 1, 0u);
    }

    // AugLagUpdate!(al, mu, y)  (auglagfun.jl:91-101) on the device copies mu_, ymul_ ; safeguard: the dual
    // safeguard of alps.jl:62 applied to y first, in the same pass
    void aug_lag_update(bool safeguard = false) {
        // Uniform penalties / zero multipliers (this rank's part of them): the one-pass kernel then takes mu as a
        // number and does not stream mu (nor mu*y).  alps.jl:42 gives every constraint the same mu when c(x0) is
        // in D, alps.jl:97 scales them alike, and y0 = 0 holds through the first subproblem — the longest one.
        uni_ = 0;
        const int uni_env = std::getenv("BZ_UNI") ? std::atoi(std::getenv("BZ_UNI")) : 2;      // (tests toggle it)
        const bool probe = uni_env && (fused_family() >= 0 || (slack && !lp_g) ||
                                       (desc.f_kind == BZ_F_STENCIL5 && desc.c_kind == BZ_C_IDENTITY && !slack && !lp_g));
        for (int k = 0; k < 3; ++k) slot_n[SL_GP + k] = grid_y;
        slot_n[SL_OUTER] = slot_n[SL_OUTER + 1] = grid_y;
        mv(3 + (safeguard ? 1 : 0), ny);
        launch(C_MISC, k_muy<T>, grid_y, (const T*)mu_.p, ymul_.p, muy_.p, ny, parts_.p, (int)SL_OUTER,
               safeguard ? 1 : 0, probe ? (int)SL_GP : -1);
        gather(SL_OUTER, 2, 2u, 3u);
        std::vector<double> v, u;
        if (probe && !ctx->multi()) {
            // one read-back for both groups; the probe's slots are this rank's own (no exchange: every form of
            // the kernel gives the same bits)
            auto a5 = collect({SL_OUTER, SL_OUTER + 1, SL_GP, SL_GP + 1, SL_GP + 2}, 2u | (7u << 2));
            v = {a5[0], a5[1]}; u = {a5[2], a5[3], a5[4]};
        } else {
            v = collect({SL_OUTER, SL_OUTER + 1}, 2u);
            if (probe) {
                CollectArgs a;
                a.n = 3; a.maxmask = 7u;
                for (int k = 0; k < 3; ++k) a.src[k] = ScalarSrc{parts_.p + (size_t)(SL_GP + k) * PSTRIDE, grid_y, 1};
                u = collect_run(a);
            }
        }
        if (v[1] > 0.0) throw Error(BZ_ERR_MU, "parameters `mu` must be positive");
        musqy = T(0.5) * T(v[0]);
        if (generic_) { copy_out(hmu_.data(), mu_.p, ny); copy_out(hmuy_.data(), muy_.p, ny); }
        if (probe && u[1] == 0.0 && u[0] > 0.0) {
            uni_ = (u[2] == 0.0 && uni_env >= 2) ? 2 : 1;
            P.mu_uniform = (T)u[0];
        }
        P.uni = uni_;      // every kernel that takes the penalties through load_params takes them as numbers then
    }
    // the oracle family of the iterate-history one-pass kernel (fam_code, bz_kernels.h), or -1: c = Identity, an
    // element-wise f, any element-wise g but the Newton / L0 kinds, any D
    int fused_family() const {
        if (desc.c_kind != BZ_C_IDENTITY || slack || ny != n || lp_g) return -1;
        int fk, gk, dk;
        switch (desc.f_kind) {
        case BZ_F_ZERO: fk = FAM_F_ZERO; break;
        case BZ_F_DIAG_QUADRATIC: fk = FAM_F_DIAG; break;
        default: return -1;
        }
        switch (desc.g_kind) {
        case BZ_G_ZERO: gk = FAM_G_ZERO; break;
        case BZ_G_NORM_L1: gk = FAM_G_L1; break;
        case BZ_G_NORM_L1_NONNEG: gk = FAM_G_L1NONNEG; break;
        case BZ_G_NORM_L1_BOX: gk = FAM_G_L1BOX; break;
        case BZ_G_IND_BOX: gk = (P.g_lo_vec || P.g_hi_vec) ? FAM_G_INDBOX_VEC : FAM_G_INDBOX; break;
        default: return -1;
        }
        switch (desc.D_kind) {
        case BZ_D_ZERO: dk = FAM_D_ZERO; break;
        case BZ_D_FREE: dk = FAM_D_FREE; break;
        case BZ_D_BOX: dk = (P.D_lo_vec || P.D_hi_vec) ? FAM_D_BOX_VEC : FAM_D_BOX; break;
        case BZ_D_VC_PAIRS: dk = FAM_D_VC; break;
        case BZ_D_CC_PAIRS: dk = FAM_D_CC; break;
        case BZ_D_EITHEROR_PAIRS: dk = FAM_D_EITHEROR; break;
        case BZ_D_XOR_PAIRS: dk = FAM_D_XOR; break;
        default: return -1;
        }
        return fam_code(fk, gk, dk);
    }
    int uni_ = 0;

    // --------------------------------------------------------------- L-BFGS
    void alloc_history() {
        if ((int)S_.size() == M + 1) return;
        S_ = std::vector<DBuf<T>>(M + 1);
        Y_ = std::vector<DBuf<T>>(M + 1);
        for (int i = 0; i <= M; ++i) { S_[i].alloc(vcap); Y_[i].alloc(vcap); }
        if (affine_ok_) {
            AS_ = std::vector<DBuf<T>>(M + 1); AY_ = std::vector<DBuf<T>>(M + 1);
            GS_ = std::vector<DBuf<T>>(M + 1); GY_ = std::vector<DBuf<T>>(M + 1);
            for (int i = 0; i <= M; ++i) { AS_[i].alloc(ny); AY_[i].alloc(ny); GS_[i].alloc(vcap); GY_[i].alloc(vcap); }
        }
    }
    void lbfgs_reset_all() {
        gm = 0; pw_valid = false;
        order.clear(); freeslots.clear();
        spare = 0;
        for (int i = M; i >= 1; --i) freeslots.push_back(i);
        H = T(1);
    }
    void lbfgs_reset() {                 // reset!(H): currmem = curridx = 0, H = 1
        if (dir_kind_ == BZ_DIR_BROYDEN) broyden_reset();
        gm = 0; pw_valid = false;
        for (int s : order) freeslots.push_back(s);
        order.clear();
        H = T(1);
    }
    // Gram products of the stored pairs after inserting a pair whose products with them are sy[i], yy[i]
    // (i = logical index, oldest first): drop the oldest when the ring is full, append row/column
    void gram_insert(double* sy, double* yy, double ys, double yty) {
        int m = gm;
        if (m == M) {                    // the oldest pair is overwritten
            for (int i = 1; i < m; ++i)
                for (int j = 1; j < m; ++j) { Gsy[(i - 1) * CM + (j - 1)] = Gsy[i * CM + j]; Gyy[(i - 1) * CM + (j - 1)] = Gyy[i * CM + j]; }
            for (int i = 1; i < m; ++i) { sy[i - 1] = sy[i]; yy[i - 1] = yy[i]; hp_[i - 1] = hp_[i]; hw_[i - 1] = hw_[i]; }
            --m;
        }
        hp_[m] = p_new_; hw_[m] = w_new_;
        for (int i = 0; i < m; ++i) {
            Gsy[i * CM + m] = sy[i]; Gsy[m * CM + i] = 0.0;
            Gyy[i * CM + m] = yy[i]; Gyy[m * CM + i] = yy[i];
        }
        Gsy[m * CM + m] = ys; Gyy[m * CM + m] = yty;
        gm = m + 1;
    }
    // M1 = R^-T (D + H0 Y'Y) R^-1 and M2 = R^-1 (same loops as LBFGSCompactOperator.coefficient_matrices)
    void compact_matrices(double H0, double* M1, double* M2) const {
        const int m = gm;
        double Ri[CM * CM] = {0}, B[CM * CM] = {0}, T1[CM * CM] = {0};
        for (int j = 0; j < m; ++j) {
            Ri[j * CM + j] = 1.0 / Gsy[j * CM + j];
            for (int i = j - 1; i >= 0; --i) {
                double acc = 0.0;
                for (int k = i + 1; k <= j; ++k) acc += Gsy[i * CM + k] * Ri[k * CM + j];
                Ri[i * CM + j] = -acc / Gsy[i * CM + i];
            }
        }
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) B[i * CM + j] = H0 * Gyy[i * CM + j] + (i == j ? Gsy[i * CM + i] : 0.0);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                double acc = 0.0;
                for (int k = 0; k <= j; ++k) acc += B[i * CM + k] * Ri[k * CM + j];
                T1[i * CM + j] = acc;
            }
        for (int i = 0; i < CM * CM; ++i) { M1[i] = 0.0; M2[i] = Ri[i]; }
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                double acc = 0.0;
                for (int k = 0; k <= i; ++k) acc += Ri[k * CM + i] * T1[k * CM + j];
                M1[i * CM + j] = acc;
            }
    }
    CompactVecs<T, CM> compact_vecs() const {      // logical (oldest first) view of the physical ring
        CompactVecs<T, CM> V;
        std::memset(&V, 0, sizeof(V));
        V.m = (int)order.size();
        for (int i = 0; i < V.m; ++i) { V.S[i] = S_[order[V.m - 1 - i]].p; V.Y[i] = Y_[order[V.m - 1 - i]].p; }
        return V;
    }
    // coefficient block for the kernels that apply the operator.  p = S'(-res), w = Y'(-res) normally came
    // back with the previous iteration's scalars; otherwise (the accepted trial was not the fused one)
    // they take their own pass and read-back
    CompactCoef<CM> compact_prepare(const CompactVecs<T, CM>& V) {
        const int m = V.m;
        if (m == 0) {
            for (int i = 0; i < CM; ++i) { hp_[i] = 0.0; hw_[i] = 0.0; }
            pw_valid = true;
        }
        if (!pw_valid) {
            for (int k = 0; k < 2 * CM; ++k) slot_n[SL_GP + k] = grid;
            mv(2 * m + 1);
            launch(C_DOT, k_gram_dots<T, CM>, grid, V, (const T*)RES_[rc].p, n, parts_.p, (int)SL_GP);
            gather(SL_GP, 2 * CM, 0u);
            auto pv = collect({SL_GP + 0, SL_GP + 1, SL_GP + 2, SL_GP + 3, SL_GP + 4, SL_GP + 5, SL_GP + 6,
                               SL_GP + 7, SL_GP + 8, SL_GP + 9}, 0u);
            for (int i = 0; i < CM; ++i) { hp_[i] = i < m ? pv[i] : 0.0; hw_[i] = i < m ? pv[CM + i] : 0.0; }
            pw_valid = true;
        }
        CompactCoef<CM> C;
        std::memset(&C, 0, sizeof(C));
        C.H0 = (double)H;
        if (dir_kind_ == BZ_DIR_ANDERSON) {
            // d = v + (S - Y) a , a = (Y'Y)^-1 Y'v : the compact kernels' linear combination with u1 = a, H0 u2 = -a
            double a[CM] = {0};
            anderson_coefficients(m, hw_, a);
            C.H0 = 1.0;
            for (int i = 0; i < CM; ++i) { C.u1[i] = i < m ? a[i] : 0.0; C.u2h[i] = i < m ? -a[i] : 0.0; }
            return C;
        }
        double M1[CM * CM], M2[CM * CM];
        compact_matrices(C.H0, M1, M2);
        // same loops as LBFGSCompactOperator.__call__ (rows/columns beyond m are zero)
        for (int i = 0; i < CM; ++i) {
            double a = 0.0, b = 0.0, c = 0.0;
            for (int j = 0; j < CM; ++j) a += M1[i * CM + j] * (j < m ? hp_[j] : 0.0);
            for (int j = 0; j < CM; ++j) b += M2[j * CM + i] * (j < m ? hw_[j] : 0.0);
            for (int j = 0; j < CM; ++j) c += M2[i * CM + j] * (j < m ? hp_[j] : 0.0);
            C.u1[i] = i < m ? a - C.H0 * b : 0.0;
            C.u2h[i] = i < m ? C.H0 * (-c) : 0.0;
        }
        return C;
    }

    void lbfgs_insert(T ys, T yty, const double* sy = nullptr, const double* yy = nullptr) {     // update!(H, s, y) when <s,y> > 0
        if (M == 0) return;              // NoAcceleration: nothing is stored, H stays 1
        if (compact_ok) {
            double z[CM] = {0};
            double a[CM], b[CM];
            for (int i = 0; i < CM; ++i) { a[i] = sy ? sy[i] : z[i]; b[i] = yy ? yy[i] : z[i]; }
            gram_insert(a, b, (double)ys, (double)yty);
        }
        order.push_front(spare);
        ys_[spare] = ys;
        if ((int)order.size() > M) { spare = order.back(); order.pop_back(); }
        else { spare = freeslots.back(); freeslots.pop_back(); }
        H = dir_kind_ == BZ_DIR_ANDERSON ? T(1) : ys / yty;
    }

    // d = H(-res) up to the last axpy, which the caller fuses with what follows
    // same result through the persistent kernel (one launch, d register-resident)
    TailArgs<T> two_loop_persist() {
        TailArgs<T> t;
        std::memset(&t, 0, sizeof(t));
        t.alphas = alphas_.p;
        const int m = (int)order.size();
        PersistArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        a.res = RES_[rc].p;
        for (int j = 0; j < m; ++j) { a.S[j] = S_[order[j]].p; a.Y[j] = Y_[order[j]].p; a.ys[j] = ys_[order[j]]; }
        a.H = H; a.m = m; a.d_out = D_.p; a.n = n; a.parts = parts_.p; a.alphas = alphas_.p;
        a.counter = pcounter_.p; a.base = pbase + (persist_sabotage_ ? 1ull : 0ull); a.timeout = ptimeout_dev_;
        a.abort_flag = pgflag_.p + 1;
        a.slot_loop1 = SL_LOOP1; a.slot_loop2 = SL_LOOP2;
        a.nb = persist_blocks();
        const bool multi = ctx->nranks > 1;
        const int nphases = 2 * m - 1 + (multi ? 1 : 0);
        a.nranks = ctx->nranks; a.rank = ctx->rank; a.pseq = ctx->pseq + 1;
        if (multi) {
            a.mbox_local = (P2PWords*)ctx->mbox_local;
            for (int r = 0; r < ctx->nranks; ++r) a.mbox_peer[r] = (P2PWords*)ctx->mbox_peer[r];
            a.gtot = pglobal_.p; a.gflag = pgflag_.p; a.final_tot = pglobal_.p + 2;
            ctx->pseq += nphases;
        }
        pbase += (unsigned long long)nphases * a.nb;
        mv(4 * m, vcap);
        form_[C_PERSIST] = "k_twoloop_persist<KR=" + std::to_string(persist_kr) + ">";
        switch (persist_kr) {
#define BZ_KR_CASE(K) case K: launch_persist(k_twoloop_persist<T, K>, a); break;
        BZ_KR_CASE(1) BZ_KR_CASE(2) BZ_KR_CASE(3) BZ_KR_CASE(4) BZ_KR_CASE(5) BZ_KR_CASE(6) BZ_KR_CASE(7) BZ_KR_CASE(8)
        BZ_KR_CASE(10) BZ_KR_CASE(12) BZ_KR_CASE(14) BZ_KR_CASE(16) BZ_KR_CASE(20) BZ_KR_CASE(24) BZ_KR_CASE(28)
        BZ_KR_CASE(32) BZ_KR_CASE(36) BZ_KR_CASE(40) BZ_KR_CASE(44) BZ_KR_CASE(48)
#undef BZ_KR_CASE
        default: throw Error(BZ_ERR_STATE, "persistent two-loop: no instantiation for this size");
        }
        slot_n[SL_LOOP2 + 0] = a.nb;
        t.in = D_.p; t.sgn = T(1); t.v = S_[order[0]].p; t.mode = 1; t.j = 0; t.apply_H = 0; t.H = T(1);
        t.src = multi ? ScalarSrc{pglobal_.p + 2, 1, 1} : src(SL_LOOP2 + 0);
        t.ys = ys_[order[0]];
        return t;
    }
    int persist_blocks() const { return pblocks; }
    static int persist_occupancy(int kr) {
        int nb = 0;
        hipError_t e = hipErrorInvalidValue;
        switch (kr) {
#define BZ_KR_CASE(K) case K: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_twoloop_persist<T, K>, PBLOCK, 0); break;
        BZ_KR_CASE(1) BZ_KR_CASE(2) BZ_KR_CASE(3) BZ_KR_CASE(4) BZ_KR_CASE(5) BZ_KR_CASE(6) BZ_KR_CASE(7) BZ_KR_CASE(8)
        BZ_KR_CASE(10) BZ_KR_CASE(12) BZ_KR_CASE(14) BZ_KR_CASE(16) BZ_KR_CASE(20) BZ_KR_CASE(24) BZ_KR_CASE(28)
        BZ_KR_CASE(32) BZ_KR_CASE(36) BZ_KR_CASE(40) BZ_KR_CASE(44) BZ_KR_CASE(48)
#undef BZ_KR_CASE
        default: break;
        }
        if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
        return nb;
    }
    // register packs per thread: the smallest instantiated count >= the need (rounds past the need
    // stream zero padding, so the steps are finer where the relative waste would be larger)
    static int persist_round_kr(int kneed) {
        static const int ks[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 36, 40, 44, 48};
        for (int k : ks) if (k >= kneed) return k;
        return 0;
    }
    template <class K> void launch_persist(K kernel, const PersistArgs<T>& a) {
        ProfRec r{C_PERSIST, nullptr, nullptr, 0.0};
        const bool prof_on = prof_pick(C_PERSIST);
        account(C_PERSIST, prof_on ? &r : nullptr);
        if (prof_on) {
            r.a = get_event(); r.b = get_event();
            hipExtLaunchKernelGGL(kernel, dim3(a.nb), dim3(PBLOCK), 0, cur_, r.a, r.b, 0, a);
            prof_recs.push_back(r);
        } else {
            hipLaunchKernelGGL(kernel, dim3(a.nb), dim3(PBLOCK), 0, cur_, a);
        }
        BZ_HIP(hipGetLastError());
    }

    TailArgs<T> two_loop() {
        TailArgs<T> t;
        std::memset(&t, 0, sizeof(t));
        t.alphas = alphas_.p;
        const int m = (int)order.size();
        slot_n[SL_LOOP2 + 0] = grid;
        const T* res = RES_[rc].p;
        if (m == 0) {
            t.in = res; t.v = nullptr; t.sgn = T(-1); t.mode = 2; t.apply_H = 1; t.H = H;
            t.src = ScalarSrc{parts_.p, 0, 1}; t.ys = T(1);
            return t;
        }
        mv(2);
        launch(C_DOT, k_dot<T>, grid, (const T*)S_[order[0]].p, res, T(-1), n, parts_.p, SL_LOOP1 + 0);
        gather(SL_LOOP1 + 0, 1, 0u);
        for (int j = 0; j + 1 < m; ++j) {        // loop 1: d -= alpha_j y_j ; <s_{j+1}, d>
            TailArgs<T> a = t;
            a.in = (j == 0) ? res : (const T*)D_.p; a.sgn = (j == 0) ? T(-1) : T(1);
            a.v = Y_[order[j]].p; a.mode = 0; a.j = j; a.apply_H = 0; a.H = T(1);
            a.src = src(SL_LOOP1 + j); a.ys = ys_[order[j]];
            mv(4); nm("k_axpy_dot");
            launch(C_TWOLOOP, k_axpy_dot<T>, grid, a, (const T*)S_[order[j + 1]].p, (const T*)nullptr,
                   D_.p, n, parts_.p, SL_LOOP1 + j + 1);
            gather(SL_LOOP1 + j + 1, 1, 0u);
        }
        {                                          // d = H (d - alpha_{m-1} y_{m-1}) ; <y_{m-1}, d>
            TailArgs<T> a = t;
            const int j = m - 1;
            a.in = (m == 1) ? res : (const T*)D_.p; a.sgn = (m == 1) ? T(-1) : T(1);
            a.v = Y_[order[j]].p; a.mode = 0; a.j = j; a.apply_H = 1; a.H = H;
            a.src = src(SL_LOOP1 + j); a.ys = ys_[order[j]];
            mv(3);
            launch(C_TWOLOOP, k_axpy_dot<T>, grid, a, (const T*)Y_[order[j]].p, (const T*)nullptr, D_.p,
                   n, parts_.p, SL_LOOP2 + j);
            gather(SL_LOOP2 + j, 1, 0u);
        }
        for (int j = m - 1; j >= 1; --j) {         // loop 2: d += (alpha_j - beta_j) s_j ; <y_{j-1}, d>
            TailArgs<T> a = t;
            a.in = D_.p; a.sgn = T(1); a.v = S_[order[j]].p; a.mode = 1; a.j = j; a.apply_H = 0;
            a.H = T(1); a.src = src(SL_LOOP2 + j); a.ys = ys_[order[j]];
            mv(4);
            launch(C_TWOLOOP, k_axpy_dot<T>, grid, a, (const T*)Y_[order[j - 1]].p, (const T*)nullptr,
                   D_.p, n, parts_.p, SL_LOOP2 + j - 1);
            gather(SL_LOOP2 + j - 1, 1, 0u);
        }
        t.in = D_.p; t.sgn = T(1); t.v = S_[order[0]].p; t.mode = 1; t.j = 0; t.apply_H = 0; t.H = T(1);
        t.src = src(SL_LOOP2 + 0); t.ys = ys_[order[0]];
        return t;
    }

    // ------------------------------------------------ Base.iterate(iter)  (k = 1)
    void begin_dev(const bz_panoc_opts& o, const T* x0_dev) {
        try {
            begin_impl(o, x0_dev);
        } catch (const DenseFusedTimeout&) {
            // (the start of a solve writes nothing it does not write again: once more, in the two-kernel form)
            if (ctx->nranks > 1) throw;
            BZ_HIP(hipStreamSynchronize(cur_));
            *ptimeout_ = 0;
            std::fprintf(stderr, "Warning: the one-pass dense kernel timed out (is the GPU shared?); using the two-kernel form\n");
            begin_impl(o, X_[0].p);
        }
    }
    void begin_impl(const bz_panoc_opts& o, const T* x0_dev) {
        opt = o;
        if (o.lbfgs_memory < 0 || o.lbfgs_memory > MAX_MEM)
            throw Error(BZ_ERR_ARG, "lbfgs_memory must be in 0..16 (0 = NoAcceleration)");
        if (o.max_backtracks < 1) throw Error(BZ_ERR_ARG, "max_backtracks must be >= 1");
        M = o.lbfgs_memory;
        if (o.directions != BZ_DIR_LBFGS && o.directions != BZ_DIR_ANDERSON && o.directions != BZ_DIR_BROYDEN)
            throw Error(BZ_ERR_ARG, "unknown `directions`");
        dir_kind_ = o.directions;
        if (dir_kind_ == BZ_DIR_ANDERSON && (M < 1 || M > CM))
            throw Error(BZ_ERR_ARG, "AndersonAcceleration(n): 1 <= n <= 5");
        if (dir_kind_ == BZ_DIR_BROYDEN) {
            if (n > 4096) throw Error(BZ_ERR_UNSUPPORTED, "Broyden() keeps a dense n-by-n operator: n <= 4096");
            if (ctx->nranks > 1) throw Error(BZ_ERR_UNSUPPORTED, "Broyden() is not sharded");
            M = 0;                                   // no pair history: the operator is the matrix
            broyden_theta_bar_ = (T)o.broyden_theta_bar;
            if (!HB_.p) {
                HB_.alloc((size_t)n * n); BHy_.alloc(npad); BsH_.alloc(npad);
                plan_chunks(n, b_rpc_, b_nch_);
                if (GT_.n < (size_t)b_nch_ * npad) GT_.alloc((size_t)b_nch_ * npad);
            }
        }
        alloc_history();
        lbfgs_reset_all();
        if (dir_kind_ == BZ_DIR_BROYDEN) broyden_reset();
        if (o.lbfgs_compact < 0 || o.lbfgs_compact > 2) throw Error(BZ_ERR_ARG, "lbfgs_compact must be 0, 1 or 2 (auto)");
        if (o.lbfgs_compact == 1 && M > CM) throw Error(BZ_ERR_ARG, "lbfgs_compact supports lbfgs_memory <= 5");
        alpha = (T)o.alpha; beta = (T)o.beta; min_gamma = (T)o.minimum_gamma;
        // Lf = nothing, gamma = Lf === nothing ? nothing : alpha / Lf, adaptive = gamma === nothing   (upstream's keywords)
        if (o.gamma < 0.0 || o.Lf < 0.0 || o.gamma != o.gamma || o.Lf != o.Lf) throw Error(BZ_ERR_ARG, "gamma and Lf must be >= 0 (0 = nothing)");
        if (o.adaptive < -1 || o.adaptive > 1) throw Error(BZ_ERR_ARG, "adaptive must be -1 (default), 0 or 1");
        gamma_given_ = o.gamma > 0.0 ? (T)o.gamma : (o.Lf > 0.0 ? alpha / (T)o.Lf : T(0));
        // (upstream tests `iter.gamma === nothing || iter.adaptive == true` at both halving sites: without a given step
        // size the estimate is always backtracked, whatever `adaptive` says)
        adaptive_ = !(gamma_given_ > T(0)) || o.adaptive == 1;
        // (the slack form of ALS too: x_i couples with s_i only — k_fused_slack; no pairwise D there, it needs the partner)
        fused_ok = o.fuse && desc.c_kind == BZ_C_IDENTITY && !lp_g && (!slack || desc.D_kind < BZ_D_VC_PAIRS) &&
                   (desc.f_kind == BZ_F_ZERO || desc.f_kind == BZ_F_DIAG_QUADRATIC);
        // auto: the compact representation where it makes the whole iteration one pass (the fused separable
        // path, memory within its capacity), the two-loop recursion everywhere else
        // (... and the stencil path: x_d, k_stencil_fb, k_stencil_update_c with ONE reduction phase per iteration
        // instead of the persistent two-loop kernel's 2m - 1 grid barriers, or 2m + 1 exchanges when sharded)
        stencil_fast_ = desc.f_kind == BZ_F_STENCIL5 && o.fuse && !lp_g && !slack;
        if (o.affine_refresh < 0) throw Error(BZ_ERR_ARG, "affine_refresh must be >= 0");
        aff_refresh_ = o.affine_refresh;
        static const int aff_env = std::getenv("BZ_AFFINE") ? std::atoi(std::getenv("BZ_AFFINE")) : -1;
        if (aff_env >= 0) aff_refresh_ = aff_env;
        aff_track_ = affine_ok_ && aff_refresh_ > 0 && o.lbfgs_compact != 0 && M >= 1 && M <= CM && dir_kind_ == BZ_DIR_LBFGS;
        aff_count_ = 0; n_affine_ = 0; n_gated_ = 0; n_gate_aborts_ = 0; n_dense_onepass_ = 0;
        compact_ok = M >= 1 && (o.lbfgs_compact == 1 || dir_kind_ == BZ_DIR_ANDERSON ||
                                (o.lbfgs_compact == 2 && (fused_ok || stencil_fast_ || aff_track_) && M <= CM));
        {
            // persistent two-loop: d must fit the register files (<= 40 packs per thread, one 512-thread
            // block per CU) and the vector must be long enough for 2m-1 grid barriers to beat 2m launches
            int64_t min_n = 300000;      // below this 2m short launches beat 2m-1 grid barriers (~5 us each)
            if (const char* e = getenv("BZ_PERSIST_MIN_N")) min_n = atoll(e);
            // with several ranks the phases need the p2p mailboxes (RCCL cannot be called from a kernel)
            persist_ok = o.persist && (!ctx->multi() || ctx->p2p_on) && persist_kr > 0 && n >= min_n && !x_replicated &&
                         !persist_broken_;
            if (ctx->nranks > 1 && !ctx->multi())
                throw Error(BZ_ERR_STATE, "nranks > 1 needs an RCCL communicator or connected p2p mailboxes");
            persist_sabotage_ = std::getenv("BZ_TEST_PERSIST_TIMEOUT") && std::atoi(std::getenv("BZ_TEST_PERSIST_TIMEOUT")) != 0;
            if (ctx->nranks > 1 && o.persist && M >= 1) {
                // the shards may straddle a threshold (length, register budget): the phases of the persistent kernel
                // and the exchanges of the kernel chain do not talk to each other, so all ranks must take the same
                // form — the persistent one only if every rank can
                launch_b(C_MISC, k_fill_slot, 1, 64, parts_.p, (int)SL_AUX, persist_ok ? 0.0 : 1.0);
                slot_n[SL_AUX] = 1;
                gather(SL_AUX, 1, 1u);
                if (collect({SL_AUX}, 1u)[0] > 0.0) persist_ok = false;
            }
            // auto (lbfgs_compact = 2), one rank: wherever the two-loop would run as a CHAIN of 2m kernels (a vector beyond the
            // persistent kernel's register capacity — e.g. the lifted vector [x; s] of ALS at n = 1e7 — or too short for its
            // grid barriers) the compact form does the same work in two launches and 4m + 11 passes instead of 8m + 1
            // (ALS at n = 1e7: 741 against 519 it/s).  Several ranks keep the rule above: they must agree on one form.
            if (!compact_ok && o.lbfgs_compact == 2 && M >= 1 && M <= CM && dir_kind_ == BZ_DIR_LBFGS && !ctx->multi() && !persist_ok &&
                !generic_)
                compact_ok = true;
        }
        t_begin = std::chrono::steady_clock::now();
        k_ = 1; n_grad = n_prox = n_bt = n_halv = n_fused = n_skips = 0;
        last_nbt = 0; last_fused = false; tau = T(0); last_ys = T(0); fbe_last = T(0);
        xc = 0; rc = 0; zc = 0; z_valid = true; xr_run_ = 0; sy_stale_ = false; rh_stale_ = false; res_valid = true;
        xr_env_ = std::getenv("BZ_XR") ? std::atoi(std::getenv("BZ_XR")) : 2;
        gfc_env_ = std::getenv("BZ_GFC") ? std::atoi(std::getenv("BZ_GFC")) : 0;
        trialfuse_env_ = std::getenv("BZ_TRIALFUSE") ? std::atoi(std::getenv("BZ_TRIALFUSE")) : 1;
        fused_begin_env_ = std::getenv("BZ_FUSED_BEGIN") ? std::atoi(std::getenv("BZ_FUSED_BEGIN")) : 1;
        skipz_env_ = std::getenv("BZ_SKIPZ") ? std::atoi(std::getenv("BZ_SKIPZ")) : 1;
        famrt_env_ = std::getenv("BZ_FAMRT") ? std::atoi(std::getenv("BZ_FAMRT")) : 0;
        nt_env_ = std::getenv("BZ_NT") ? std::atoi(std::getenv("BZ_NT")) : -1;
        slackfast_env_ = std::getenv("BZ_SLACKFAST") ? std::atoi(std::getenv("BZ_SLACKFAST")) : 1;
        slackkind_env_ = std::getenv("BZ_SLACKKIND") ? std::atoi(std::getenv("BZ_SLACKKIND")) : 1;
        densesmall_env_ = std::getenv("BZ_DENSESMALL") ? std::atoi(std::getenv("BZ_DENSESMALL")) : 1;
        affblend_env_ = std::getenv("BZ_AFFINE_BLEND") ? std::atoi(std::getenv("BZ_AFFINE_BLEND")) : 1;
        stencil_regx_env_ = std::getenv("BZ_STENCIL_REGX") ? std::atoi(std::getenv("BZ_STENCIL_REGX")) : 1;
        suc_grid_env_ = std::getenv("BZ_SUC_GRID") ? std::atoi(std::getenv("BZ_SUC_GRID")) : 1;
        slackdepth_env_ = std::getenv("BZ_SLACKDEPTH") ? std::atoi(std::getenv("BZ_SLACKDEPTH")) : 1;
        // BZ_GATE: 0 off; 1 (default) the early launch queues behind the read-back on the solver's own stream; 2 on the other
        // stream (resident while the previous pass runs: measured slower, kept for the record)
        // Several ranks: off unless asked for (BZ_GATE=1).  A launch that misses its gate cannot be redone there (the peers
        // have consumed this rank's scalars: BZ_ERR_COMM), and gated launches on distinct devices have never run on
        // hardware — bench.py asks for them after checking, on the node it runs on, that they reproduce the plain launches.
        gate_env_ = std::getenv("BZ_GATE") ? std::atoi(std::getenv("BZ_GATE")) : (ctx->nranks > 1 ? 0 : 1);
        // a resident launch polling at its gate holds its CUs: with another tenant on the GPU (a rank of this very job in
        // a one-GPU rehearsal, or whoever made an earlier launch of this problem miss its gate) the two starve each other
        if (ctx->shared_device || gate_broken_) gate_env_ = 0;
        gate_sabotage_ = std::getenv("BZ_TEST_GATE_TIMEOUT") ? std::atoi(std::getenv("BZ_TEST_GATE_TIMEOUT")) : 0;
        if (gate_env_ && fused_ok) gate_alloc();      // (pinned record, device copy, second stream: not inside an iteration)
        gate_quiesce();
        if (x0_dev != X_[0].p)
            BZ_HIP(hipMemcpyAsync(X_[0].p, x0_dev, n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
        const T eps = std::numeric_limits<T>::epsilon();
        T* x = X_[xc].p;
        // grad_f_x, f_x = gradient(f, x)
        const int lip_env = fused_begin_env_;
        if (lip_env && desc.c_kind == BZ_C_IDENTITY && !slack && !dense_f &&
            (desc.f_kind == BZ_F_ZERO || desc.f_kind == BZ_F_DIAG_QUADRATIC)) {
            // gradient at x and the Lipschitz estimate in one pass (k_begin_lip)
            slot_n[SL_FXD] = slot_n[SL_FXD + 1] = grid;
            mv(2 + pstreams(true, true, false));
            launch(C_ALGRAD, k_begin_lip<T>, grid, (const T*)x, P, GX_.p, n, parts_.p, (int)SL_FXD, (int)SL_AUX);
            gather(SL_FXD, 2, 0u);
            n_grad += 2; gx_valid = true;
        } else {
            if (aff_track_) cx_keep_ = CXS_.p;
            algrad(x, GX_.p, SL_FXD); ++n_grad; gx_valid = true;
            cx_keep_ = nullptr;
            if (!(gamma_given_ > T(0))) {
                // gamma = alpha / lower_bound_smoothness_constant(f, I, x, grad_f_x)
                mv(2);
                launch(C_MISC, k_add_scalar<T>, grid, (const T*)x, T(1), TMP_.p, n);
                algrad(TMP_.p, GZ_.p, SL_FZ); ++n_grad;
                mv(4);
                launch(C_MISC, k_diff_ss2<T>, grid, (const T*)GZ_.p, (const T*)GX_.p, (const T*)TMP_.p, (const T*)x, n,
                       parts_.p, (int)SL_AUX);
            }
        }
        if (gamma_given_ > T(0)) {
            // gamma given (or Lf): no Lipschitz estimate; the one-pass start above computed it for free and it is ignored
            auto v = collect({SL_FXD, SL_PXD}, 0u);
            f_x = al_value(v[0], v[1]);
            gamma = gamma_given_;
        } else {
            slot_n[SL_AUX] = slot_n[SL_AUX + 1] = grid;
            gather(SL_AUX, 2, 0u);
            auto v = collect({SL_FXD, SL_PXD, SL_AUX, SL_AUX + 1}, 0u);
            f_x = al_value(v[0], v[1]);
            const T Lest = std::sqrt(T(v[2])) / std::sqrt(T(v[3]));
            gamma = alpha / Lest;
        }
        // y = x - gamma grad ; z, g_z = prox(g, y, gamma) ; res = x - z ; backtrack_stepsize!
        T f_z = T(0);
        const bool fused_fb = lip_env && desc.c_kind == BZ_C_IDENTITY && !slack && !dense_f && !lp_g &&
                              (desc.f_kind == BZ_F_ZERO || desc.f_kind == BZ_F_DIAG_QUADRATIC);
        double stop0 = 0.0;
        for (;;) {
            std::vector<double> v;
            if (fused_fb) {
                // FB step, gradient at z and stop norm in one pass (k_begin_fb)
                for (int k = 0; k < 8; ++k) slot_n[SL_GSUM + k] = grid;
                mv(4 + pstreams(true, true, true));
                launch(C_FB, k_begin_fb<T>, grid, (const T*)x, (const T*)GX_.p, gamma, P, Z_[zc].p, RES_[rc].p, n,
                       parts_.p, (int)SL_GSUM);
                gather(SL_GSUM, 8, 1u << 7);
                ++n_prox; ++n_grad; gz_valid = false;
                v = collect({SL_GSUM, SL_DOT, SL_SS, SL_FZ, SL_PZ, SL_STOP}, 1u << 5);
                stop0 = v[5];
            } else {
                fbstep(x, GX_.p, gamma, Z_[zc].p, RES_[rc].p, SL_GSUM);
                gather(SL_GSUM, 3, 0u);
                ++n_prox;
                if (aff_track_) cx_keep_ = CZS_.p;
                algrad(Z_[zc].p, GZ_.p, SL_FZ); ++n_grad; gz_valid = true;
                cx_keep_ = nullptr;
                v = collect({SL_GSUM, SL_DOT, SL_SS, SL_FZ, SL_PZ}, 0u);
            }
            g_z = g_value(v[0]); dot_gr = T(v[1]); ss_res = T(v[2]);
            f_z = al_value(v[3], v[4]); fraw_last = f_value(v[3]); f_z_al = f_z;
            const T nr = std::sqrt(ss_res);
            const T f_z_upp = f_x - dot_gr + ((alpha / gamma) / T(2)) * (nr * nr);
            const T tol = T(10) * eps * (T(1) + std::abs(f_z));
            // (upstream: `if (iter.gamma === nothing || iter.adaptive == true)` backtrack_stepsize!)
            // (an infinite gamma — a zero Lipschitz estimate: grad F(x + 1) = grad F(x), e.g. f = Zero in the slack form — cannot be
            // halved: the reference's loop compares NaNs there and leaves; here f(z) may come out +inf, so say it explicitly)
            if (adaptive_ && std::isfinite((double)gamma) && f_z > f_z_upp + tol && gamma >= min_gamma) {
                gamma = gamma / T(2); ++n_halv;
                continue;
            }
            break;
        }
        if (gamma < min_gamma)
            std::fprintf(stderr, "Warning: stepsize `gamma` became too small (%g)\n", (double)gamma);
        if (fused_fb) {
            stop_norm_ = stop0;
        } else {
            for (int k = 0; k < 3; ++k) slot_n[SL_YS + k] = grid;
            mv(4);
            launch(C_UPDATE, k_update<T>, grid, (const T*)x, (const T*)nullptr, (const T*)RES_[rc].p,
                   (const T*)nullptr, (const T*)GX_.p, (const T*)GZ_.p, gamma, (T*)nullptr, (T*)nullptr, n,
                   parts_.p, (int)SL_YS);
            gather(SL_YS, 3, 4u);
            auto v = collect({SL_STOP}, 1u);
            stop_norm_ = v[0];
        }
        gring_[xc] = (double)gamma;
        active = true;
    }

    void run_to_completion() {
        LoopGuard guard{this};
        for (;;) {
            const bool stop = should_stop();
            if (stop) gate_abort();              // (a pass pre-launched for an iteration that will not happen)
            if (opt.verbose && (stop || (opt.freq > 0 && k_ % opt.freq == 0))) display();
            if (stop) break;
            more_coming_ = k_ + 1 < opt.maxit;   // the solver's own loop: another iteration follows unless this one stops it
            step();
        }
    }

    void display() {
        ensure_z();
        mv(1);
        slot_n[SL_AUX] = grid;
        launch(C_MISC, k_absmax<T>, grid, (const T*)RES_[rc].p, n, parts_.p, (int)SL_AUX);
        gather(SL_AUX, 1, 1u);
        auto v = collect({SL_AUX}, 1u);
        std::printf("%5lld | %.3e | %.3e | %.3e\n", (long long)k_, (double)gamma, v[0] / (double)gamma,
                    (double)tau);
    }

    // ---------------------------------------- Base.iterate(iter, state)  (k += 1)
   public:
    // (leaving a library-run loop, normally or by an exception: no pass stays pre-launched at its gate, the flag that
    // allows pre-launching is down, the solver is back on the context's stream)
    struct LoopGuard {
        Solver* s;
        ~LoopGuard() {
            s->more_coming_ = false;
            try { s->gate_quiesce(); } catch (...) {}
        }
    };
    void steps(int64_t k) override {
        LoopGuard guard{this};
        for (int64_t i = 0; i < k; ++i) {
            more_coming_ = i + 1 < k;            // (the caller asked for all k: the next pass may be launched early)
            step();
        }
    }
    void step() override {
        require_active();
        // Two things can make an iteration fail without having committed anything of the state, and both are reported
        // through the next read-back: a grid barrier of the persistent two-loop kernel that cannot complete (its
        // workgroups are not all resident: another stream or process holds CUs), and a pass pre-launched behind its gate
        // that gave up there (the host was descheduled for seconds, or the GPU has another tenant).  The iteration is
        // simply redone — with the kernel chain, without the gate — and the form that failed stays off for this
        // problem.  Whatever else goes wrong leaves with no launch waiting at a gate.
        const int64_t sv[7] = {k_, n_grad, n_prox, n_bt, n_halv, n_fused, n_skips};
        const unsigned long long sv_pseq = ctx->pseq;
        auto restore = [&]() {
            k_ = sv[0]; n_grad = sv[1]; n_prox = sv[2]; n_bt = sv[3]; n_halv = sv[4]; n_fused = sv[5]; n_skips = sv[6];
            gx_valid = false; gz_valid = false;
        };
        try {
            try {
                step_impl();
            } catch (const GateTimeout&) {
                // (several ranks: the others have taken this rank's stale scalars for good ones — not recoverable here)
                if (ctx->nranks > 1) throw;
                gate_abort();              // (the pass launched early for the iteration after this one)
                BZ_HIP(hipStreamSynchronize(cur_));
                cur_ = ctx->stream;
                *ptimeout_ = 0;
                restore();
                gate_broken_ = true; gate_env_ = 0; ++n_gate_fallbacks_;
                std::fprintf(stderr, "Warning: a pre-launched pass timed out at its gate (is the GPU shared?); gated pre-launch is off for this problem\n");
                step_impl();
            } catch (const DenseFusedTimeout&) {
                // the one-pass dense kernel's workgroups wait for each other (k_dense_fused): with CUs held by somebody else a
                // row group can be partly resident — the two-kernel form from now on, and the iteration again
                if (ctx->nranks > 1) throw;
                gate_abort();
                BZ_HIP(hipStreamSynchronize(cur_));
                cur_ = ctx->stream;
                *ptimeout_ = 0;
                restore();
                aff_count_ = aff_refresh_;      // (the images of this iteration's points were being formed: evaluate afresh)
                std::fprintf(stderr, "Warning: the one-pass dense kernel timed out (is the GPU shared?); using the two-kernel form\n");
                step_impl();
            } catch (const PersistTimeout&) {
                if (!persist_ok) throw;
                BZ_HIP(hipStreamSynchronize(ctx->stream));
                *ptimeout_ = 0;
                BZ_HIP(hipMemsetAsync(pcounter_.p, 0, PSHARDS * PSHARD_STRIDE * sizeof(unsigned long long), ctx->stream));
                BZ_HIP(hipMemsetAsync(pgflag_.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
                BZ_HIP(hipStreamSynchronize(ctx->stream));
                pbase = 0; ctx->pseq = sv_pseq;
                restore();
                persist_ok = false; persist_broken_ = true; ++n_persist_fallbacks_;
                std::fprintf(stderr, "Warning: persistent two-loop kernel timed out at its grid barrier; using the kernel chain\n");
                step_impl();
            }
        } catch (...) {
            more_coming_ = false;
            try { gate_quiesce(); } catch (...) {}
            throw;
        }
        if (!more_coming_) gate_quiesce();
    }
    int64_t n_persist_fallbacks_ = 0;
   private:
    void step_impl() {
        ++k_;
        const T eps = std::numeric_limits<T>::epsilon();
        const int max_bt = opt.max_backtracks;
        // FBE at the current state
        const T nr0 = std::sqrt(ss_res);
        const T FBE_x = (f_x - dot_gr + ((alpha / gamma) / T(2)) * (nr0 * nr0)) + g_z;
        fbe_last = FBE_x;
        // direction d = H(-res): all but the last axpy
        // (headline family: the one-pass kernel also serves an EMPTY memory — d = H0 (-res), all coefficients zero —
        // so the first iteration of a solve is a 3..5-stream pass too instead of k_fused_sep's 12)
        // (the stencil fast path also serves a row-sharded grid: the halo rows of x_d and of z travel before the two
        // passes, and with the compact form the iteration has ONE scalar exchange — the 32 slots of k_stencil_update_c)
        const bool stencil_fast_now = stencil_fast_ && (!ctx->multi() || (ctx->p2p_on && compact_ok));
        const bool use_compact = compact_ok && (!order.empty() || (fused_ok && fused_family() >= 0) ||
                                                stencil_fast_now || aff_track_);
        const bool use_persist = persist_ok && !order.empty() && !use_compact;
        CompactVecs<T, CM> CV;
        CompactCoef<CM> CC;
        TailArgs<T> tail;
        if (use_compact) { CV = compact_vecs(); CC = compact_prepare(CV); std::memset(&tail, 0, sizeof(tail)); }
        else if (dir_kind_ == BZ_DIR_BROYDEN) { if (!res_valid) ensure_z(); tail = broyden_dir(); }
        else tail = use_persist ? two_loop_persist() : two_loop();
        double gsy[CM] = {0}, gyy[CM] = {0};
        double tp[CM] = {0}, tw[CM] = {0}, tpn = 0.0, twn = 0.0;      // next p, w as measured by the fused trial
        constexpr int NFC = 10 + 4 * CM + 2;                          // slots of k_fused_compact
        static_assert(SL_TRIAL + NFC <= SL_AUX, "k_fused_compact's slots overlap the next group");
        const int m_at_trial = (int)order.size();
        bool tail_used = false, z_skipped = false, res_skipped = false;
        unsigned long long tail_ticket = 0;
        bool gram_from_trial = false;
        tau = T(1);
        const int xp = xc, xd = (xc + 1) % NXR, xb = (xc + 2) % NXR;
        const int rp = rc, rn = (rc + 1) % NRR, zp = zc, zn = 1 - zc;
        int xcur = xd;
        bool have_trial = false, fused_this = false, reset_this = false, sep_trial = false;
        bool img_trial = false;      // grad L (and f) at the trial point x + d are affine images, not evaluations
        // a backtracked trial point can go through the one-pass kernel too ("trial given" variant) when this
        // iteration's first trial did: what that launch used is kept here
        bool trial_ok = false, trial_nt = false;
        bool head_on = false, head_fb = false;      // cfg 4: k_dense_head serves this iteration ; ... and has made the FB step of its first trial
        bool state_imgs = false, halved_here = false;      // cfg 4: the state's images are valid ; gamma was halved inside this step
        int trial_uni = 0, trial_gfc = 0, trial_fam = -1;
        bool trial_table = false;
        CompactVecs<T, CM> trial_XV;
        CompactCoef<CM> trial_CC;
        double sep_p = 0.0, sep_w = 0.0;      // <s_new, -res>, <y_new, -res> as measured by a k_fused_sep trial
        if (fused_ok && use_compact) {
            // 176 VGPRs -> two 256-thread blocks per CU: one resident round of blocks (each block pays the
            // coefficient prologue and a 20-slot reduction epilogue once)
            const int gfc_env = gfc_env_;
            int gfc = std::min(grid, (gfc_env > 0 ? gfc_env : 2) * std::max(1, num_cus));
            // non-temporal loads/stores once the working set (2M + 11 vectors) no longer fits the 256 MB Infinity
            // Cache.  Measured fused-pass times, default policy vs non-temporal: n = 1.25e6 (210 MB) 39.0 / 44.5 us,
            // 1.8e6 (302 MB) 51.0 / 60.4, 2.5e6 (420 MB) 90.1 / 81.5, 5e6 (840 MB) 171 / 159, 1e7 322 / 314.
            const int nt_env = nt_env_;      // (BZ_NT, read at every bz_panoc_begin: the tests run both instantiations)
            // headline family with everything uniform fixed at compile time (see the kernel)
            static const int spec_env = std::getenv("BZ_SPEC") ? std::atoi(std::getenv("BZ_SPEC")) : 1;
            const int fam = fused_family();
            const bool headline = spec_env && fam == FAM_HEADLINE;
            const bool spec = headline && CV.m == CM;
#define BZ_LAUNCH_FC(NT_, SPEC_)                                                                                  \
    launch(C_FUSED, k_fused_compact<T, CM, NT_, SPEC_>, gfc, CV, CC, (const T*)X_[xp].p, (const T*)RES_[rp].p, P, \
           gamma, X_[xd].p, zstore, RES_[rn].p, S_[spare].p, Y_[spare].p, n, parts_.p, (int)SL_TRIAL)
            // z is the solution the caller reads when the solve stops: once the stop norm is within a factor 10 of
            // the tolerance, store it (one more write stream for the last iteration or two) rather than
            // re-materialise it afterwards with two generic kernels (the same bits either way)
            // ... and during the first 20 iterations of a solve: ALPS subproblems are often that short (13 of them
            // with 180 inner iterations in all on cfg 2), and a stored z costs a tenth of re-materialising one
            // (tol = 0: the caller has said the solve never stops by itself — bench.py's timed region, the step-wise parity
            // tests — so no early stop is being prepared for; whoever asks for z gets it re-materialised, same bits)
            const bool near_stop = (double)stop_norm_ <= 10.0 * opt.tol || (k_ <= 20 && opt.tol > 0.0);
            T* const zstore = (skipz_env_ && !near_stop) ? (T*)nullptr : Z_[zn].p;
            z_skipped = zstore == nullptr;
            static const int off32_env = std::getenv("BZ_OFF32") ? std::atoi(std::getenv("BZ_OFF32")) : 1;
            const bool small = off32_env && (double)vcap * sizeof(T) < 4.0e9;
            const bool off32 = small && spec;
            // 0: stored pairs; 1: pairs re-formed from the iterate / residual rings (full memory only); 2: residuals
            // re-evaluated too — possible as soon as every stored pair is a difference of ring neighbours, also
            // with a partial memory (the absent pairs are x - x = 0 with zero coefficients)
            const int m_now = (int)order.size();
            int xr = 0;
            // (the slack form of ALS has its own iterate-history kernel, k_fused_slack_xr: BZ_XR >= 2, any element-wise kinds)
            if (xr_env_ && small && (fam >= 0 || slack) && xr_run_ >= m_now) {
                if (xr_env_ >= 2) xr = 2;
                else if (headline && m_now == CM && !rh_stale_) xr = 1;
                // (only the oldest stored iterate may carry another gamma — see CompactCoef::gam0)
                for (int i = 1; i < m_now; ++i)
                    if (gring_[(xc - m_now + i + NXR) % NXR] != (double)gamma) xr = 0;
                if (xr == 1 && gring_[(xc - m_now + NXR) % NXR] != (double)gamma) xr = 0;
            }
            if (sy_stale_ && !xr) materialize_pairs();
            if (xr != 2 && !res_valid) ensure_z();
            const int uni = xr == 2 ? uni_ : 0;
            // the iterate-history form keeps two packs of loads in flight per wave and runs best with ONE wave per
            // SIMD (n = 1e7: 134 vs 140 us; 1.25e6: 28.5 vs 30.2 us): the wave has the vector ALU to itself and
            // the 32-scalar epilogue runs half as often.  (A different grid is a different summation tree: the
            // forms then agree to rounding, not bit for bit — BZ_GFC pins one grid for all of them.)
            // (the slack form's fast instantiations: a full memory, f = DiagQuadratic, no vector-valued parameters of g or D —
            // pipelined like the headline kernel, 256 VGPRs + spill AGPRs: one workgroup per CU there too)
            const bool slack_fast = slack && xr == 2 && slackfast_env_ && m_now == CM && desc.f_kind == BZ_F_DIAG_QUADRATIC &&
                                    pstreams(false, true, true) == 2 - std::min(2, (int)P.uni) &&
                                    !(P.g_u && (P.g_kind == BZ_G_NORM_L1_BOX || P.g_kind == BZ_G_NORM_L0_BOX));
            // ... with the kinds fixed too (g = NormL1, D = Box: the ALS form of cfg 2), one pack of loads ahead
            const bool slack_hk = slack_fast && slackkind_env_ && P.g_kind == BZ_G_NORM_L1 && P.D_kind == BZ_D_BOX;
            if (xr == 2 && gfc_env <= 0 && (!slack || (slack_hk && slackdepth_env_ > 0))) gfc = std::min(grid, std::max(1, num_cus));
            for (int k = 0; k < NFC; ++k) slot_n[SL_TRIAL + k] = gfc;
            // (the vectors this pass touches: history + x_d + z + the parameter vectors (+ res, s, y))
            const int xr2_streams = (m_now + 1) + pstreams(true, true, true) + 1;      // (pstreams leaves out what travels as numbers)
            const int nvec = (xr == 2 ? xr2_streams : 2 * CM + 5 + pstreams(true, true, true)) + (zstore ? 1 : 0);
            const bool nt = nt_env >= 0 ? nt_env != 0 : (double)n * sizeof(T) * nvec > 340e6;
            if (gate_pending_ && xr != 2) gate_abort();
            if (xr == 2 && slack) {
                // the m + 1 last iterates of the lifted vector (both halves), the parameter vectors, y ; xs_d (z) out
                SlackIterates<T, CM> SV;
                std::memset(&SV, 0, sizeof(SV));
                SV.m = m_now;
                for (int i = 0; i <= m_now; ++i) SV.XH[i] = X_[(xc - (m_now - i) + NXR) % NXR].p;
                CC.gam0 = gring_[(xc - m_now + NXR) % NXR];
                const int slack_streams = 2 * (m_now + 1) + pstreams(true, true, true) + (P.uni >= 2 ? 0 : 1) + 2;      // (pstreams counts D's vector bounds)
                const bool snt = nt_env_ >= 0 ? nt_env_ != 0 : (double)nx * sizeof(T) * (slack_streams + (zstore ? 2 : 0)) > 340e6;
                mv(slack_streams + (zstore ? 2 : 0), nx);
                form_[C_FUSED_IT] = std::string("k_fused_slack_xr") + (snt ? "<NT=1>" : "<NT=0>");
#define BZ_LAUNCH_SXR(NT_, FULL_, UNI_, KIND_, DEPTH_)                                                             \
    launch(C_FUSED_IT, k_fused_slack_xr<T, CM, NT_, FULL_, UNI_, KIND_, DEPTH_>, gfc, SV, CC, P, (const T*)ymul_.p, gamma, X_[xd].p, zstore, nx, \
           parts_.p, (int)SL_TRIAL)
#define BZ_LAUNCH_SXR_U(NT_, KIND_, DEPTH_)                                                                        \
    do { if (P.uni >= 2) BZ_LAUNCH_SXR(NT_, true, 2, KIND_, DEPTH_); else if (P.uni == 1) BZ_LAUNCH_SXR(NT_, true, 1, KIND_, DEPTH_); \
         else BZ_LAUNCH_SXR(NT_, true, 0, KIND_, DEPTH_); } while (0)
                // (the fast instantiations: a full memory, f = DiagQuadratic, no vector-valued parameters of g or D)
                const bool fast = slack_fast;
                if (fast) {
                    const int sdepth = slackdepth_env_;
                    const bool hk = slack_hk;
                    form_[C_FUSED_IT] += hk ? "(fast,l1-box)" : "(fast)";
                    if (hk && sdepth >= 1) { if (snt) BZ_LAUNCH_SXR_U(true, 1, 1); else BZ_LAUNCH_SXR_U(false, 1, 1); }
                    else if (hk) { if (snt) BZ_LAUNCH_SXR_U(true, 1, 0); else BZ_LAUNCH_SXR_U(false, 1, 0); }
                    else { if (snt) BZ_LAUNCH_SXR_U(true, 0, 0); else BZ_LAUNCH_SXR_U(false, 0, 0); }
                } else if (m_now == CM) { if (snt) BZ_LAUNCH_SXR(true, true, -1, 0, 0); else BZ_LAUNCH_SXR(false, true, -1, 0, 0); }
                else { if (snt) BZ_LAUNCH_SXR(true, false, -1, 0, 0); else BZ_LAUNCH_SXR(false, false, -1, 0, 0); }
#undef BZ_LAUNCH_SXR_U
#undef BZ_LAUNCH_SXR
                sy_stale_ = true; rh_stale_ = true; res_skipped = true;
                trial_ok = false;      // (a tau-backtracked point finishes in the generic chain, after the pairs are re-materialised)
            } else if (xr == 2) {
                CompactVecs<T, CM> XV;
                XV.m = CM;
                for (int i = 0; i < CM; ++i) {
                    const int slot = (xc - std::max(0, m_now - i) + NXR) % NXR;      // (beyond m: x itself)
                    XV.S[i] = X_[slot].p;
                    XV.Y[i] = nullptr;
                    if (i == 0) CC.gam0 = gring_[slot];
                }
                const bool table = !headline || famrt_env_;
                GatePlan cur;
                std::memset(&cur, 0, sizeof(cur));
                for (int i = 0; i < CM; ++i) cur.S[i] = XV.S[i];
                cur.x = X_[xp].p; cur.xd = X_[xd].p; cur.gam0 = CC.gam0; cur.gamma = (double)gamma; cur.uni = uni; cur.gfc = gfc;
                cur.fam = fam; cur.m_now = m_now; cur.nt = nt; cur.table = table;
                if (gate_pending_ && cur == gate_plan_) {
                    // this very launch was made early, behind the previous iteration's read-back: hand it its coefficients
                    gate_release(CC, zstore);
                } else {
                    gate_abort();
                    // streams: the m_now + 1 distinct iterates (x among them), the family's parameter vectors (mu / mu*y
                    // unless passed as numbers); x_d (z)
                    mv(xr2_streams + (zstore ? 1 : 0));
                    CompactCoef<CM> C2 = CC;
                    C2.gate_seq = 0ull;
                    gate_launch(cur, C2, zstore);
                }
                sy_stale_ = true; rh_stale_ = true; res_skipped = true;
                const int tf_now = trialfuse_env_;
                trial_ok = tf_now != 0; trial_nt = nt; trial_uni = uni; trial_gfc = gfc; trial_XV = XV; trial_CC = CC;
                trial_table = table; trial_fam = fam;
            } else if (xr) {
                CompactVecs<T, CM> XV;
                XV.m = CM;
                for (int i = 0; i < CM; ++i) {
                    XV.S[i] = X_[(xc - CM + i + NXR) % NXR].p;
                    XV.Y[i] = RES_[(rc - CM + i + NRR) % NRR].p;
                }
                mv(2 * (CM + 1) + pstreams(true, true, true) + 2 + (zstore ? 1 : 0));
                form_[C_FUSED] = std::string("k_fused_compact<XR=1") + (nt ? ",NT=1>" : ",NT=0>");
                if (nt)
                    launch(C_FUSED, k_fused_compact<T, CM, true, true, true, 1>, gfc, XV, CC, (const T*)X_[xp].p,
                           (const T*)RES_[rp].p, P, gamma, X_[xd].p, zstore, RES_[rn].p, (T*)nullptr, (T*)nullptr, n, parts_.p,
                           (int)SL_TRIAL);
                else
                    launch(C_FUSED, k_fused_compact<T, CM, false, true, true, 1>, gfc, XV, CC, (const T*)X_[xp].p,
                           (const T*)RES_[rp].p, P, gamma, X_[xd].p, zstore, RES_[rn].p, (T*)nullptr, (T*)nullptr, n, parts_.p,
                           (int)SL_TRIAL);
                sy_stale_ = true;
            } else
#define BZ_LAUNCH_FC3(NT_)                                                                                        \
    launch(C_FUSED, k_fused_compact<T, CM, NT_, true, true>, gfc, CV, CC, (const T*)X_[xp].p, (const T*)RES_[rp].p, P, \
           gamma, X_[xd].p, zstore, RES_[rn].p, S_[spare].p, Y_[spare].p, n, parts_.p, (int)SL_TRIAL)
            if (slack) {
                // the lifted vector [x; s]: res, S[m], Y[m], xs ; xs_d, res, s_new, y_new (z) — both halves — and over n the
                // parameter vectors and the multipliers y
                mv(2 * (2 + 2 * CV.m + 4 + (zstore ? 1 : 0)) + pstreams(true, true, true) + (P.uni >= 2 ? 0 : 1), nx);
                form_[C_FUSED] = std::string("k_fused_slack") + (nt ? "<NT=1>" : "<NT=0>");
                if (nt)
                    launch(C_FUSED, k_fused_slack<T, CM, true>, gfc, CV, CC, (const T*)X_[xp].p, (const T*)RES_[rp].p, P,
                           (const T*)ymul_.p, gamma, X_[xd].p, zstore, RES_[rn].p, S_[spare].p, Y_[spare].p, nx, parts_.p,
                           (int)SL_TRIAL);
                else
                    launch(C_FUSED, k_fused_slack<T, CM, false>, gfc, CV, CC, (const T*)X_[xp].p, (const T*)RES_[rp].p, P,
                           (const T*)ymul_.p, gamma, X_[xd].p, zstore, RES_[rn].p, S_[spare].p, Y_[spare].p, nx, parts_.p,
                           (int)SL_TRIAL);
            } else {
                // stored pairs: res, S[m], Y[m], x + the parameter vectors ; x_d, res, s_new, y_new (z)
                mv(2 + 2 * CV.m + pstreams(true, true, true) + 4 + (zstore ? 1 : 0));
                form_[C_FUSED] = std::string("k_fused_compact<XR=0") + (spec ? ",SPEC=1" : ",SPEC=0") + (nt ? ",NT=1>" : ",NT=0>");
                if (off32 && nt) BZ_LAUNCH_FC3(true);
                else if (off32) BZ_LAUNCH_FC3(false);
                else if (nt && spec) BZ_LAUNCH_FC(true, true);
                else if (nt) BZ_LAUNCH_FC(true, false);
                else if (spec) BZ_LAUNCH_FC(false, true);
                else BZ_LAUNCH_FC(false, false);
            }
#undef BZ_LAUNCH_FC3
#undef BZ_LAUNCH_FC
            // the NEXT iteration's pass, assuming this one ends the plain way (trial accepted, pair inserted, same gamma):
            // ring one step on, one more pair
            GatePlan nxt;
            bool have_plan = false;
            if (xr == 2 && gate_env_ && more_coming_ && !opt.verbose && !prof_would_pick(C_FUSED_IT)) {
                double gr[NXR];
                for (int i = 0; i < NXR; ++i) gr[i] = gring_[i];
                gr[xd] = (double)gamma;
                have_plan = gate_make_plan(xd, std::min(m_now + 1, M), gr, xr_run_ + 1, nxt);
            }
            if (ctx->p2p_on) {
                // exchange + fold over the ranks + read-back in one launch (no k_collect)
                tail_ticket = exchange_collect(SL_TRIAL, NFC, 1u << 9);
                tail_used = true;
            } else {
                gather(SL_TRIAL, NFC, 1u << 9);
                if (xr == 2 && gate_env_) {      // the read-back kernel now, so that the next pass can queue right behind it
                    tail_ticket = collect_launch_range(SL_TRIAL, NFC, 1u << 9);
                    tail_used = true;
                }
            }
            if (have_plan) gate_prelaunch(nxt);
            have_trial = true; fused_this = true; gx_valid = false; gz_valid = false; gram_from_trial = true;
            n_grad += 2; n_prox += 1;
        } else if (fused_ok && !slack) {
            if (!res_valid) ensure_z();
            for (int k = 0; k < 12; ++k) slot_n[SL_TRIAL + k] = grid;
            mv((tail.mode != 2 ? 2 : 1) + 2 + pstreams(true, true, true) + 5);
            form_[C_FUSED] = "k_fused_sep";
            launch(C_FUSED, k_fused_sep<T>, grid, tail, (const T*)X_[xp].p, (const T*)RES_[rp].p, P, gamma,
                   X_[xd].p, Z_[zn].p, RES_[rn].p, S_[spare].p, Y_[spare].p, (T*)nullptr, (T*)nullptr, n,
                   parts_.p, (int)SL_TRIAL);
            gather(SL_TRIAL, 12, 1u << 9);
            sep_trial = true;
            have_trial = true; fused_this = true; gx_valid = false; gz_valid = false;
            n_grad += 2; n_prox += 1;
        } else {
            if (!res_valid) ensure_z();
            // cfg 4 with images: x_d, its images under grad L and c, L(x_d) and the forward-backward step in ONE launch
            // (k_dense_head; single rank, element-wise f, the common prox kinds)
            head_on = densesmall_env_ && use_compact && aff_track_ && !stencil_fast_now && !generic_ && !ctx->multi() && !lp_g &&
                      !dense_f && gx_valid && gz_valid && aff_count_ + 1 < aff_refresh_ &&
                      (desc.f_kind == BZ_F_ZERO || desc.f_kind == BZ_F_DIAG_QUADRATIC);
            // x_d = x + d ; gradient at x_d ; state.x = x_d
            if (head_on) {
            } else if (use_compact) {
                mv(2 * CV.m + 3);
                // (full memory + a history beyond the Infinity Cache: compile-time trip counts, non-temporal history loads)
                static const int xdnt_env = std::getenv("BZ_XDNT") ? std::atoi(std::getenv("BZ_XDNT")) : 1;
                const bool hist_nt = xdnt_env && (double)n * sizeof(T) * (2 * CV.m + 3) > 340e6;
                // (the template form in the name: a hardware-counter profile is matched to the instantiation that ran)
                nm(CV.m == CM ? (hist_nt ? "k_compact_xd<FULL=1,NT=1>" : "k_compact_xd<FULL=1,NT=0>") : "k_compact_xd<FULL=0,NT=0>");
                if (CV.m == CM && hist_nt)
                    launch(C_XD, k_compact_xd<T, CM, true, true>, grid, CV, CC, (const T*)RES_[rp].p, (const T*)X_[xp].p, X_[xd].p, n);
                else if (CV.m == CM)
                    launch(C_XD, k_compact_xd<T, CM, true, false>, grid, CV, CC, (const T*)RES_[rp].p, (const T*)X_[xp].p, X_[xd].p, n);
                else
                launch(C_XD, k_compact_xd<T, CM>, grid, CV, CC, (const T*)RES_[rp].p, (const T*)X_[xp].p,
                       X_[xd].p, n);
            } else {
                mv((tail.mode != 2 ? 2 : 1) + 2); nm("k_axpy_dot(x_d)");
                launch(C_XD, k_axpy_dot<T>, grid, tail, (const T*)nullptr, (const T*)X_[xp].p, X_[xd].p, n,
                       parts_.p, 0);
            }
            if (stencil_fast_now) {
                // stencil fast path: {AL gradient at x_d + FB step} and {AL gradient at z + pair + stop norm}
                // as two passes; same partial sums as the four generic kernels of the first trial
                for (int sidx = SL_FXD; sidx <= SL_STOP; ++sidx) slot_n[sidx] = grid;
                // (uniform penalties / zero multipliers travel as numbers, P.uni: the two stencil passes stream mu and mu*y
                // otherwise — 4 of the iteration's 43 passes)
                // (r03: with the compact form the second pass re-forms grad L(x_d) and res from x_d and z — k_stencil_update_c<REGX> —
                // so this pass does not write the gradient and that one reads neither: 39 -> 36 passes over n per iteration)
                const int regx = use_compact ? std::max(0, std::min(2, stencil_regx_env_)) : 0;
                mv(2 + pstreams(false, true, true) + (regx >= 2 ? 2 : 3));        // x_d, b + parameters ; (grad,) z, res
                const StencilHalo<T> halo_x = halo_exchange(X_[xd].p);
                static const int fbnt_env = std::getenv("BZ_XDNT") ? std::atoi(std::getenv("BZ_XDNT")) : 1;
                nm(fbnt_env && (double)n * sizeof(T) * 12 > 340e6 ? "k_stencil_fb<NT=1>" : "k_stencil_fb<NT=0>");
                if (fbnt_env && (double)n * sizeof(T) * 12 > 340e6)
                    launch(C_STENCIL_FB, k_stencil_fb<T, true>, grid, (const T*)X_[xd].p, P, (int64_t)desc.f_grid_nx,
                           (int64_t)desc.f_grid_ny, gamma, regx >= 2 ? (T*)nullptr : GX_.p, Z_[zn].p, RES_[rn].p, n, parts_.p, (int)SL_FXD,
                           (int)SL_GSUM, halo_x);
                else
                launch(C_STENCIL_FB, k_stencil_fb<T>, grid, (const T*)X_[xd].p, P, (int64_t)desc.f_grid_nx,
                       (int64_t)desc.f_grid_ny, gamma, regx >= 2 ? (T*)nullptr : GX_.p, Z_[zn].p, RES_[rn].p, n, parts_.p, (int)SL_FXD,
                       (int)SL_GSUM, halo_x);
                const StencilHalo<T> halo_z = halo_exchange(Z_[zn].p);
                if (use_compact) {
                    // ... with the Gram products of the new pair and the next application's p, w in the same pass
                    // (r03: this pass — 19 streams, 27 accumulators — runs best with ONE workgroup per CU, one resident round and a
                    // 27-slot epilogue per CU: 102 us against 108 with two or four and 114 on the problem's grid of 2048, at 2048^2 ;
                    // the other two passes want the largest grid.  BZ_SUC_GRID=k: k per CU, 0: `grid`.)
                    const int g_upd = suc_grid_env_ > 0 ? std::min(grid, suc_grid_env_ * std::max(1, num_cus)) : grid;
                    for (int sidx = 0; sidx < NFC; ++sidx) slot_n[SL_TRIAL + sidx] = sidx < 5 ? grid : g_upd;      // (slots 0..4: k_stencil_fb's)
                    mv(2 + pstreams(false, true, false) + (5 - regx) + 2 + 2 * CV.m);
#define BZ_LAUNCH_SUC_R(FULL_, NT_, REGX_)                                                                        \
    launch(C_STENCIL_UPD, k_stencil_update_c<T, CM, FULL_, NT_, REGX_>, g_upd, CV, (const T*)Z_[zn].p, P, (int64_t)desc.f_grid_nx, \
           (int64_t)desc.f_grid_ny, (const T*)X_[xd].p, (const T*)X_[xp].p, (const T*)RES_[rn].p, (const T*)RES_[rp].p, \
           (const T*)GX_.p, gamma, S_[spare].p, Y_[spare].p, n, parts_.p, (int)SL_TRIAL, halo_z, halo_x)
#define BZ_LAUNCH_SUC(FULL_, NT_)                                                                                 \
    do { if (regx >= 2) BZ_LAUNCH_SUC_R(FULL_, NT_, 2); else if (regx == 1) BZ_LAUNCH_SUC_R(FULL_, NT_, 1);      \
         else BZ_LAUNCH_SUC_R(FULL_, NT_, 0); } while (0)
                    {
                        static const int xdnt_env = std::getenv("BZ_XDNT") ? std::atoi(std::getenv("BZ_XDNT")) : 1;
                        const bool hist_nt = xdnt_env && (double)n * sizeof(T) * (2 * CV.m + 12) > 340e6;
                        if (regx >= 2) nm(CV.m == CM ? (hist_nt ? "k_stencil_update_c<FULL=1,NT=1,REGX=2>" : "k_stencil_update_c<FULL=1,NT=0,REGX=2>")
                                                     : "k_stencil_update_c<FULL=0,NT=0,REGX=2>");
                        else if (regx == 1) nm(CV.m == CM ? (hist_nt ? "k_stencil_update_c<FULL=1,NT=1,REGX=1>" : "k_stencil_update_c<FULL=1,NT=0,REGX=1>")
                                                          : "k_stencil_update_c<FULL=0,NT=0,REGX=1>");
                        else nm(CV.m == CM ? (hist_nt ? "k_stencil_update_c<FULL=1,NT=1>" : "k_stencil_update_c<FULL=1,NT=0>")
                                           : "k_stencil_update_c<FULL=0,NT=0>");
                        if (CV.m == CM && hist_nt) BZ_LAUNCH_SUC(true, true);
                        else if (CV.m == CM) BZ_LAUNCH_SUC(true, false);
                        else BZ_LAUNCH_SUC(false, false);
                    }
#undef BZ_LAUNCH_SUC
#undef BZ_LAUNCH_SUC_R
                    if (ctx->p2p_on) {
                        // exchange + fold over the ranks + read-back of all 32 slots in one launch
                        tail_ticket = exchange_collect(SL_TRIAL, NFC, 1u << 9);
                        tail_used = true;
                    }
                    gram_from_trial = true;
                } else {
                mv(2 + pstreams(false, true, false) + 5 + 2);   // z, b + parameters, x_d, x, res, res_prev, grad ; s, y
                nm("k_stencil_update");
                launch(C_STENCIL_UPD, k_stencil_update<T>, grid, (const T*)Z_[zn].p, P, (int64_t)desc.f_grid_nx,
                       (int64_t)desc.f_grid_ny, (const T*)X_[xd].p, (const T*)X_[xp].p, (const T*)RES_[rn].p,
                       (const T*)RES_[rp].p, (const T*)GX_.p, gamma, S_[spare].p, Y_[spare].p, (T*)nullptr, n, parts_.p,
                       (int)SL_FZ, (int)SL_YS, halo_z);
                }
                have_trial = true; gx_valid = regx < 2; gz_valid = false;
                n_grad += 2; n_prox += 1;
            } else if (aff_track_) {
                // gradient (and c) at x_d into the candidate buffers, then trade: GX_ = grad L(x_d), GXN_ = grad L(x_prev)
                state_imgs = gx_valid && gz_valid;      // (the images of this state's x and z are what GX_, GZ_, CXS_, CZS_ hold)
                if (use_compact && gx_valid && gz_valid && aff_count_ + 1 < aff_refresh_) {
                    ++aff_count_; ++n_affine_; img_trial = true;
                    CompactVecs<T, CM> VA = image_vecs(true), VG = image_vecs(false);
                    if (head_on) {
                        DenseHeadArgs<T, CM> a;
                        std::memset(&a, 0, sizeof(a));
                        a.V = CV; a.VG = VG; a.VA = VA; a.C = CC;
                        a.res = RES_[rp].p; a.x = X_[xp].p; a.x_d = X_[xd].p;
                        a.gbase = GX_.p; a.gzimg = GZ_.p; a.gout = GXN_.p;
                        a.cbase = CXS_.p; a.czimg = CZS_.p; a.cout = CXD_.p; a.yupd = YU_.p;
                        a.z = Z_[zn].p; a.res_new = RES_[rn].p; a.gamma = gamma;
                        a.n = n; a.ny = ny; a.parts = parts_.p;
                        a.slot_f = SL_FXD; a.slot_pen = SL_PXD; a.slot_fb = SL_GSUM; a.gn = grid; a.gy = grid_y;
                        slot_n[SL_FXD] = grid; slot_n[SL_PXD] = grid_y;
                        for (int kk = 0; kk < 3; ++kk) slot_n[SL_GSUM + kk] = grid;
                        // (what the six kernels move: k_compact_xd, the two images, f, yupd, the FB step)
                        mv(2 * CV.m + 3); mv(2 * VA.m + 3, ny); mv(2 * VG.m + 3); mv(1 + pstreams(true, false, false));
                        mv(2 + pstreams(false, true, false), ny); mv(4 + pstreams(false, false, true));
                        nm("k_dense_head");
                        launch(C_MISC, k_dense_head<T, CM>, grid + grid_y, a, P);
                        head_fb = true;
                    } else {
                    mv(2 * VA.m + 3, ny); nm("k_affine_image");
                    launch(C_MISC, k_affine_image<T, CM>, grid_y, VA, CC, (const T*)CXS_.p, (const T*)CZS_.p, CXD_.p, ny);
                    mv(2 * VG.m + 3);
                    launch(C_MISC, k_affine_image<T, CM>, grid, VG, CC, (const T*)GX_.p, (const T*)GZ_.p, GXN_.p, n);
                    // the value L(x_d): f element-wise, the penalty from the image of c
                    slot_n[SL_FXD] = grid; slot_n[SL_PXD] = grid_y;
                    mv(1 + pstreams(true, false, false));
                    launch(C_MISC, k_fvalue_elem<T>, grid, (const T*)X_[xd].p, P, n, parts_.p, (int)SL_FXD, (const T*)nullptr);
                    mv(2 + pstreams(false, true, false), ny);
                    launch(C_MISC, k_yupd<T>, grid_y, (const T*)CXD_.p, P, YU_.p, ny, parts_.p, (int)SL_PXD);
                    }
                } else {
                    aff_count_ = 0;
                    cx_keep_ = CXD_.p;
                    algrad(X_[xd].p, GXN_.p, SL_FXD);
                    cx_keep_ = nullptr;
                }
                std::swap(GX_.p, GXN_.p); std::swap(GX_.n, GXN_.n);
                ++n_grad; gx_valid = true;
            } else {
                algrad(X_[xd].p, GX_.p, SL_FXD); ++n_grad; gx_valid = true;
            }
        }
        T sigma = beta * (T(0.5) / gamma) * (T(1) - alpha);
        const T tol0 = T(10) * eps * (T(1) + std::abs(FBE_x));
        const T threshold = FBE_x - sigma * (nr0 * nr0) + tol0;
        std::vector<double> v;
        int nbt = 0;
        bool gen_gram = false;      // the generic trial just launched carried the compact form's products (k_update_c)
        int m_gram = m_at_trial;    // ... measured against a memory of this many pairs
        for (int k = 1; k <= max_bt; ++k) {
            if (!have_trial) {
                if (!gx_valid) { algrad(X_[xcur].p, GX_.p, SL_FXD); gx_valid = true; }
                if (head_fb) head_fb = false;      // (k_dense_head made this step already)
                else fbstep(X_[xcur].p, GX_.p, gamma, Z_[zn].p, RES_[rn].p, SL_GSUM);
                gather(SL_GSUM, 3, 0u);
                ++n_prox;
                T* const gz_dst = aff_track_ ? GZN_.p : GZ_.p;      // (affine images: grad L(z_prev) is still needed)
                if (aff_track_) cx_keep_ = CZN_.p;
                // cfg 4: the fold of the row-group partials, the pair with its products and the pair's images in ONE launch
                // behind the pass over A (k_dense_tail)
                const bool tail_on = densesmall_env_ && aff_track_ && compact_ok && !generic_ && !ctx->multi() &&
                                     desc.c_kind == BZ_C_DENSE_AFFINE && dense_fused_on();
                if (tail_on) {
                    slot_n[SL_FZ] = grid;
                    dense_fused_launch(Z_[zn].p, SL_FZ + 1);
                    ++n_grad; gz_valid = true;
                    cx_keep_ = nullptr;
                    const CompactVecs<T, CM> VG = compact_vecs();
                    DenseTailArgs<T, CM> a;
                    std::memset(&a, 0, sizeof(a));
                    a.V = VG; a.part = GT_.p; a.nchunks = df_groups_; a.pstride = npad;
                    a.z = Z_[zn].p; a.gz = gz_dst;
                    a.x = X_[xcur].p; a.x_prev = X_[xp].p; a.res = RES_[rn].p; a.res_prev = RES_[rp].p; a.gx = GX_.p;
                    a.gamma = gamma; a.s_new = S_[spare].p; a.y_new = Y_[spare].p;
                    a.gx_prev = GXN_.p; a.gz_prev = GZ_.p; a.gs_img = GS_[spare].p; a.gy_img = GY_[spare].p;
                    a.cx = CXD_.p; a.cx_prev = CXS_.p; a.cz = CZN_.p; a.cz_prev = CZS_.p;
                    a.cs_img = AS_[spare].p; a.cy_img = AY_[spare].p;
                    a.n = n; a.ny = ny; a.parts = parts_.p; a.slot_fz = SL_FZ; a.slot_upd = SL_YS; a.gn = grid; a.gy = grid_y;
                    for (int kk = 0; kk < 3 + 4 * CM + 2; ++kk) slot_n[SL_YS + kk] = grid;
                    // (what the four kernels move: the fold + f terms, k_update_c, the two image pairs)
                    mv(df_groups_ + 2 + pstreams(true, false, false)); mv(8 + 2 * VG.m); mv(6, ny); mv(6);
                    nm("k_dense_tail");
                    launch(C_UPDATE, k_dense_tail<T, CM>, grid + grid_y, a, P);
                    gather(SL_FZ, 2, 0u, 2u);
                    gather(SL_YS, 3 + 4 * CM + 2, 4u);
                    gen_gram = true; m_gram = VG.m; tail_used = false;
                } else {
                algrad(Z_[zn].p, gz_dst, SL_FZ); ++n_grad; gz_valid = true;
                cx_keep_ = nullptr;
                if (compact_ok) {
                    // the pair, the stop norm AND the compact form's products (Gram products of the candidate pair, the
                    // next application's p, w) in one pass and one read-back — against the memory as it is NOW
                    const CompactVecs<T, CM> VG = compact_vecs();
                    for (int kk = 0; kk < 3 + 4 * CM + 2; ++kk) slot_n[SL_YS + kk] = grid;
                    mv(8 + 2 * VG.m); nm("k_update_c");
                    launch(C_UPDATE, k_update_c<T, CM>, grid, VG, (const T*)X_[xcur].p, (const T*)X_[xp].p,
                           (const T*)RES_[rn].p, (const T*)RES_[rp].p, (const T*)GX_.p, (const T*)gz_dst, gamma,
                           S_[spare].p, Y_[spare].p, n, parts_.p, (int)SL_TRIAL);
                    gather(SL_YS, 3 + 4 * CM + 2, 4u);
                    gen_gram = true; m_gram = VG.m; tail_used = false;
                } else {
                for (int kk = 0; kk < 3; ++kk) slot_n[SL_YS + kk] = grid;
                mv(8);
                launch(C_UPDATE, k_update<T>, grid, (const T*)X_[xcur].p, (const T*)X_[xp].p,
                       (const T*)RES_[rn].p, (const T*)RES_[rp].p, (const T*)GX_.p, (const T*)gz_dst, gamma,
                       S_[spare].p, Y_[spare].p, n, parts_.p, (int)SL_YS);
                gather(SL_YS, 3, 4u);
                }
                if (aff_track_) {
                    // images of the candidate pair (s = x - x_prev, y = res - res_prev) under c and grad L
                    mv(6, ny);
                    launch(C_MISC, k_image_pair<T>, grid_y, (const T*)CXD_.p, (const T*)CXS_.p, (const T*)CZN_.p,
                           (const T*)CZS_.p, AS_[spare].p, AY_[spare].p, ny);
                    mv(6);
                    launch(C_MISC, k_image_pair<T>, grid, (const T*)GX_.p, (const T*)GXN_.p, (const T*)GZN_.p,
                           (const T*)GZ_.p, GS_[spare].p, GY_[spare].p, n);
                }
                }
            }
            if ((have_trial && gram_from_trial) || gen_gram) {
                v = (have_trial && tail_used) ? wait_host(NFC, tail_ticket) : collect_range(SL_TRIAL, NFC, 1u << 9);
                gram_from_trial = true; gen_gram = false;
                for (int i = 0; i < CM; ++i) {
                    gsy[i] = v[10 + i]; gyy[i] = v[10 + CM + i];
                    tp[i] = v[10 + 2 * CM + i]; tw[i] = v[10 + 3 * CM + i];
                }
                tpn = v[10 + 4 * CM]; twn = v[10 + 4 * CM + 1];
            } else if (have_trial && sep_trial) {
                static_assert(SL_TRIAL == SL_FXD && SL_STOP == SL_TRIAL + 9 && SL_GU == SL_TRIAL + 10, "k_fused_sep's slots");
                v = collect_range(SL_TRIAL, 12, 1u << 9);
                sep_p = v[10]; sep_w = v[11];
                gram_from_trial = false;
            } else {
                v = collect({SL_FXD, SL_PXD, SL_GSUM, SL_DOT, SL_SS, SL_FZ, SL_PZ, SL_YS, SL_YTY, SL_STOP},
                            1u << 9);
                gram_from_trial = false;
            }
            have_trial = false;
            f_x = al_value(v[0], v[1]);
            g_z = g_value(v[2]); dot_gr = T(v[3]); ss_res = T(v[4]);
            const T f_z = al_value(v[5], v[6]);
            fraw_last = f_value(v[5]); f_z_al = f_z;
            const T nr = std::sqrt(ss_res);
            const T f_z_upp = f_x - dot_gr + ((alpha / gamma) / T(2)) * (nr * nr);
            const T tol = T(10) * eps * (T(1) + std::abs(f_z));
            const bool halve = adaptive_ && std::isfinite((double)gamma) && f_z > f_z_upp + tol && gamma >= min_gamma;
            if (halve && img_trial) {
                // The step-size test compares f(z) with a model built on f(x) and grad L(x) to within 10 eps: an image
                // (a linear combination, not an evaluation) is not consistent with f(z) to that level near convergence,
                // and a value a few ulps low would halve gamma again and again at the same point.  A FAILING test is
                // therefore never trusted on images: evaluate f and grad L at this x with the two passes over A and
                // run the trial again (this does not consume one of the max_backtracks trials).
                img_trial = false; aff_count_ = 0; ++n_affine_verify_;
                cx_keep_ = CXD_.p;
                algrad(X_[xcur].p, GX_.p, SL_FXD); ++n_grad; gx_valid = true;
                cx_keep_ = nullptr;
                --k;
                continue;
            }
            const T FBE_new = f_z_upp + g_z;
            // not a plain iteration: z of the state this step started from may be needed (z_curr below), and
            // it must be formed with the gamma of that state
            // ... and the classic kernels that finish this iteration need the stored pairs (and the residual of
            // that state) as vectors
            // a pass pre-launched for the next iteration assumed this trial is accepted and its pair inserted
            if (gate_pending_ && !(k == 1 && !halve && (FBE_new <= threshold || k >= max_bt) && T(v[7]) > T(0))) gate_abort();
            if (halve) trial_ok = false;
            if (sy_stale_ && !trial_ok && (halve || !(FBE_new <= threshold || k >= max_bt))) materialize_pairs();
            if ((!z_valid || !res_valid) && (halve || !(FBE_new <= threshold || k >= max_bt))) ensure_z();
            if (halve) {
                halved_here = true;
                gamma = gamma * T(0.5); ++n_halv;
                if (gamma < min_gamma)
                    std::fprintf(stderr, "Warning: stepsize `gamma` became too small (%g)\n", (double)gamma);
                sigma = sigma * T(2);   // (as upstream: sigma is updated, the threshold is kept)
                lbfgs_reset();
                fused_this = false; reset_this = true;
                continue;
            }
            if (FBE_new <= threshold || k >= max_bt) break;
            tau = (k >= max_bt - 1) ? T(0) : tau / T(2);
            ++nbt; ++n_bt;
            mv(3);
            launch(C_MISC, k_blend<T>, grid, (const T*)X_[xd].p, (const T*)Z_[zp].p, tau, T(1) - tau,
                   X_[xb].p, n);
            xcur = xb;
            fused_this = false;
            if (trial_ok) {
                // the blended point through the one-pass kernel: given in X_[xb], evaluated against the same ring
                // of iterates; z and res of the new state are stored (Z_[zn], RES_[rn])
                for (int kk = 0; kk < NFC; ++kk) slot_n[SL_TRIAL + kk] = trial_gfc;
#define BZ_LAUNCH_FCT(NT_, UNI_)                                                                                  \
    launch(C_FUSED_IT, k_fused_compact<T, CM, NT_, true, true, 2, UNI_, 1>, trial_gfc, trial_XV, trial_CC,     \
           (const T*)X_[xp].p, (const T*)nullptr, P, gamma, X_[xb].p, Z_[zn].p, RES_[rn].p, (T*)nullptr,         \
           (T*)nullptr, n, parts_.p, (int)SL_TRIAL)
                // the iterates, the parameter vectors (mu, mu*y unless numbers), the trial point ; z, res
                mv((m_at_trial + 1) + pstreams(true, true, true) + 1 + 2);
                if (trial_table) {
                    trial_CC.uni_rt = trial_uni; trial_CC.trial_rt = 1;
                    FusedFn<T> fn = family_kernel<T>(trial_fam, trial_nt);
                    form_[C_FUSED_IT] = "k_fused_compact<XR=2,UNI=-1,NT=" + std::to_string(trial_nt ? 1 : 0) + ",TRIAL=-1,FAM=" + std::to_string(trial_fam) + ">";
                    launch(C_FUSED_IT, fn, trial_gfc, trial_XV, trial_CC, (const T*)X_[xp].p, (const T*)nullptr, P, gamma,
                           X_[xb].p, Z_[zn].p, RES_[rn].p, (T*)nullptr, (T*)nullptr, n, parts_.p, (int)SL_TRIAL);
                } else {
                form_[C_FUSED_IT] = std::string("k_fused_compact<XR=2,UNI=") + char('0' + trial_uni) + (trial_nt ? ",NT=1" : ",NT=0") + ",TRIAL=1>";
                if (trial_nt) { if (trial_uni == 2) BZ_LAUNCH_FCT(true, 2); else if (trial_uni == 1) BZ_LAUNCH_FCT(true, 1); else BZ_LAUNCH_FCT(true, 0); }
                else { if (trial_uni == 2) BZ_LAUNCH_FCT(false, 2); else if (trial_uni == 1) BZ_LAUNCH_FCT(false, 1); else BZ_LAUNCH_FCT(false, 0); }
                }
#undef BZ_LAUNCH_FCT
                if (ctx->p2p_on) {
                    tail_ticket = exchange_collect(SL_TRIAL, NFC, 1u << 9);
                    tail_used = true;
                } else {
                    gather(SL_TRIAL, NFC, 1u << 9);
                    tail_used = false;
                }
                have_trial = true; gram_from_trial = true; gx_valid = false; gz_valid = false;
                n_grad += 2; n_prox += 1;
            } else if (affblend_env_ && aff_track_ && state_imgs && gx_valid && nbt == 1 && !halved_here && compact_ok && !generic_ &&
                       !ctx->multi() && aff_count_ + 1 < aff_refresh_) {
                // cfg 4, first tau backtrack of an iteration: the blended point is an affine combination of x_d and the state's z,
                // whose images under c and grad L are at hand — its images are the same combination (k_blend's operations), no
                // pass over A.  (As for x_d: a failing step-size test on images is re-run on evaluations, img_trial.)  The rejected
                // trial's z images (CZN_, GZN_) are dead: they take the results and trade places.
                ++aff_count_; img_trial = true;      // (n_affine_images counts iterations whose trial point x + d went on images)
                mv(3, ny);
                launch(C_MISC, k_blend<T>, grid_y, (const T*)CXD_.p, (const T*)CZS_.p, tau, T(1) - tau, CZN_.p, ny);
                std::swap(CXD_.p, CZN_.p); std::swap(CXD_.n, CZN_.n);
                mv(3);
                launch(C_MISC, k_blend<T>, grid, (const T*)GX_.p, (const T*)GZ_.p, tau, T(1) - tau, GZN_.p, n);
                std::swap(GX_.p, GZN_.p); std::swap(GX_.n, GZN_.n);
                slot_n[SL_FXD] = grid; slot_n[SL_PXD] = grid_y;
                mv(1 + pstreams(true, false, false));
                launch(C_MISC, k_fvalue_elem<T>, grid, (const T*)X_[xb].p, P, n, parts_.p, (int)SL_FXD, (const T*)nullptr);
                mv(2 + pstreams(false, true, false), ny);
                launch(C_MISC, k_yupd<T>, grid_y, (const T*)CXD_.p, P, YU_.p, ny, parts_.p, (int)SL_PXD);
                ++n_grad; gx_valid = true;
            } else {
                if (aff_track_) { cx_keep_ = CXD_.p; aff_count_ = 0; img_trial = false; }
                algrad(X_[xb].p, GX_.p, SL_FXD); ++n_grad; gx_valid = true;
                cx_keep_ = nullptr;
            }
        }
        if (aff_track_) {
            // the accepted state's images become the current ones
            std::swap(GZ_.p, GZN_.p); std::swap(GZ_.n, GZN_.n);
            std::swap(CZS_.p, CZN_.p); std::swap(CZS_.n, CZN_.n);
            std::swap(CXS_.p, CXD_.p); std::swap(CXS_.n, CXD_.n);
        }
        // update!(H, x - x_prev, res - res_prev): the pair sits in the spare slot
        const T ys = T(v[7]), yty = T(v[8]);
        last_ys = ys;
        // p, w for the next application: valid iff the accepted point is the one the fused trial measured
        // and the memory was not reset meanwhile (gram_insert shifts them along with the Gram matrices)
        pw_valid = compact_ok && gram_from_trial && (int)order.size() == m_gram;
        if (pw_valid) {
            for (int i = 0; i < CM; ++i) { hp_[i] = i < m_gram ? tp[i] : 0.0; hw_[i] = i < m_gram ? tw[i] : 0.0; }
            p_new_ = tpn; w_new_ = twn;
        } else if (compact_ok && sep_trial && fused_this && m_at_trial == 0 && order.empty()) {
            // first iteration of a solve (empty memory): the k_fused_sep pass measured the new pair's p and w
            pw_valid = true;
            for (int i = 0; i < CM; ++i) { hp_[i] = 0.0; hw_[i] = 0.0; }
            p_new_ = sep_p; w_new_ = sep_w;
        }
        if (dir_kind_ == BZ_DIR_BROYDEN) {
            broyden_update();                    // (no curvature test: every pair updates the operator)
        } else if (ys > T(0) || dir_kind_ == BZ_DIR_ANDERSON) {
            if (compact_ok && !gram_from_trial && !order.empty()) {
                // the accepted pair is not the one the fused trial measured: its Gram products with the
                // stored pairs come from their own pass
                for (int k = 0; k < 2 * CM; ++k) slot_n[SL_GU + k] = grid;
                mv(2 * (int)order.size() + 1);
                launch(C_DOT, k_gram_pair<T, CM>, grid, compact_vecs(), (const T*)Y_[spare].p, n, parts_.p,
                       (int)SL_GU);
                gather(SL_GU, 2 * CM, 0u);
                auto gv = collect({SL_GU + 0, SL_GU + 1, SL_GU + 2, SL_GU + 3, SL_GU + 4, SL_GU + 5, SL_GU + 6,
                                   SL_GU + 7, SL_GU + 8, SL_GU + 9}, 0u);
                for (int i = 0; i < CM; ++i) { gsy[i] = gv[i]; gyy[i] = gv[CM + i]; }
            }
            lbfgs_insert(ys, yty, gsy, gyy);
        } else {
            ++n_skips;
            materialize_pairs();         // (history as iterates: the window stops being contiguous here)
        }
        // A tau-backtracked point sits in the blend buffer: trade the two buffers so that the accepted iterate is
        // the next one of the ring whatever produced it — the stored pairs stay the successive differences of
        // the ring's last iterates (and residuals), and the run below goes on through backtracks
        if (xcur == xb && fused_ok && compact_ok) {
            std::swap(X_[xd].p, X_[xb].p);
            std::swap(X_[xd].n, X_[xb].n);
            xcur = xd;
        }
        // history as iterates is possible after CM iterations in a row that each inserted their pair, with no
        // change of gamma (which resets the memory) in between
        // (xr_run_: how many of the newest stored pairs are differences of ring neighbours.  The pair of an
        // iteration that halved gamma is one too — y = res_new(gamma/2) - res_prev(gamma), as upstream has it —
        // because every iterate in the ring remembers the gamma of its residual, gring_)
        xr_run_ = (fused_ok && compact_ok && (ys > T(0) || dir_kind_ == BZ_DIR_ANDERSON) && xcur == xd) ? (reset_this ? 1 : xr_run_ + 1) : 0;
        gring_[xcur] = (double)gamma;
        if (xr_run_ == 0) rh_stale_ = false;       // (whatever broke the run has materialised the pairs above)
        stop_norm_ = v[9];
        xc = xcur; rc = rn; zc = zn;
        z_valid = !(z_skipped && fused_this);      // the generic trial writes z; an accepted fused one may not have
        res_valid = !(res_skipped && fused_this);  // ... nor res
        last_nbt = nbt; last_fused = fused_this;
        if (fused_this) ++n_fused;
    }

   private:
    void fill_stats(bz_panoc_stats* st) {
        std::memset(st, 0, sizeof(*st));
        st->iters = k_; st->f_z = (double)fraw_last; st->g_z = (double)g_z; st->al_z = (double)f_z_al;
        st->gamma = (double)gamma; st->tau = (double)tau; st->stop_norm = stop_norm_;
        st->n_grad = n_grad; st->n_prox = n_prox; st->n_backtracks = n_bt; st->n_gamma_halvings = n_halv;
        st->n_fused_iters = n_fused; st->n_lbfgs_skips = n_skips;
        st->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        st->status = std::isnan((double)f_x) ? 2 : ((double)stop_norm_ <= opt.tol ? 0 : 1);
        st->persist_fallbacks = (int32_t)n_persist_fallbacks_;
        st->n_affine_images = n_affine_;
        st->n_gated_launches = n_gated_;
        st->n_gate_aborts = n_gate_aborts_;
        st->n_gate_fallbacks = n_gate_fallbacks_;
        st->n_dense_onepass = n_dense_onepass_;
        st->n_dense_fallbacks = n_dense_fallbacks_;
    }
};

// one-pass kernel of an oracle family: the instantiations live in bz_families_dk*.hip (one file per D class, so that
// they compile in parallel)
template <class T> FusedFn<T> family_kernel(int fam, bool nt, int uni) {
    switch (fam_dk(fam)) {
    case FAM_D_ZERO: return family_kernel_dk<T, FAM_D_ZERO>(fam, nt, uni);
    case FAM_D_FREE: return family_kernel_dk<T, FAM_D_FREE>(fam, nt, uni);
    case FAM_D_BOX: return family_kernel_dk<T, FAM_D_BOX>(fam, nt, uni);
    case FAM_D_BOX_VEC: return family_kernel_dk<T, FAM_D_BOX_VEC>(fam, nt, uni);
    case FAM_D_VC: return family_kernel_dk<T, FAM_D_VC>(fam, nt, uni);
    case FAM_D_CC: return family_kernel_dk<T, FAM_D_CC>(fam, nt, uni);
    case FAM_D_EITHEROR: return family_kernel_dk<T, FAM_D_EITHEROR>(fam, nt, uni);
    case FAM_D_XOR: return family_kernel_dk<T, FAM_D_XOR>(fam, nt, uni);
    default: return nullptr;
    }
}

SolverBase* make_solver(Ctx* ctx, const bz_problem_desc& d) {
    if (d.dtype == BZ_F64) return new Solver<double>(ctx, d);
    if (d.dtype == BZ_F32) return new Solver<float>(ctx, d);
    throw Error(BZ_ERR_ARG, "dtype must be BZ_F64 or BZ_F32");
}

}  // namespace bz
