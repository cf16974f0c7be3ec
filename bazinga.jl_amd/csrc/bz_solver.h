// bz_solver.h — host-side driver of the device-resident PANOCplus solve.
//
// The host keeps only scalars (gamma, tau, FBE, L-BFGS ring bookkeeping) and takes
// the line-search / step-size decisions exactly as the reference's solver does
// (ProximalAlgorithms.PANOCplus as called from src/algorithms/alps.jl:64-66); all
// n-vectors live in HBM for the whole solve and only scalars cross PCIe: one
// pinned-memory read-back per trial point.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "bz_kernels.h"

namespace bz {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define BZ_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            throw ::bz::Error(BZ_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define BZ_NCCL(expr)                                                                        \
    do {                                                                                     \
        ncclResult_t _r = (expr);                                                            \
        if (_r != ncclSuccess)                                                               \
            throw ::bz::Error(BZ_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
    } while (0)

// Peer-to-peer scalar mailbox (single node): every rank owns a small fine-grained device buffer that all
// ranks map through HIP IPC; scalars are exchanged by system-scope stores straight into the peers'
// mailboxes over xGMI, tagged with a sequence number — no collective library call, no host involvement.
constexpr int P2P_MAXRANKS = 8;
constexpr int P2P_PACK = 32;               // doubles per pack exchange
struct P2PMailbox {                        // layout of one rank's mailbox (all words written by peers)
    double pval[2][P2P_MAXRANKS];                          // persistent-kernel phase totals
    unsigned long long pflag[2][P2P_MAXRANKS];
    unsigned long long xll[2][P2P_MAXRANKS][P2P_PACK][2];  // pack exchanges (k_exchange), tagged half-words
};
struct Ctx {
    int device = 0, rank = 0, nranks = 1;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    // p2p
    bool p2p_on = false;
    bool shared_device = false;            // another rank of this job runs on the same GPU (one-GPU rehearsals): no resident, polling launches
    P2PMailbox* mbox_local = nullptr;
    P2PMailbox* mbox_peer[P2P_MAXRANKS] = {};
    bool mbox_opened[P2P_MAXRANKS] = {};
    unsigned long long pseq = 0, xseq = 0;  // sequence numbers, identical on all ranks by construction
    bool multi() const { return comm != nullptr || p2p_on; }
    ~Ctx();
};

template <class T> struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    void alloc(size_t count) {      // zero-filled, with >= 8 elements of slack behind the payload
        release();
        n = count;
        if (count) {
            BZ_HIP(hipMalloc((void**)&p, (count + 8) * sizeof(T)));
            // hipMemset on device memory is asynchronous to the host and runs on the NULL stream, which
            // the solver's non-blocking stream does not wait for: drain it before anyone uses the buffer
            BZ_HIP(hipMemsetAsync(p, 0, (count + 8) * sizeof(T), nullptr));
            BZ_HIP(hipStreamSynchronize(nullptr));
        }
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    ~DBuf() { release(); }
};

// reduction slots (each slot = PSTRIDE block partials)
enum Slot : int {
    SL_LOOP1 = 0,      // +j, j < M : <s_j, d>
    SL_LOOP2 = 16,     // +j, j < M : <y_j, d>
    SL_TRIAL = 32,     // 10 slots of the fused kernel / trial sequence
    SL_FXD = 32, SL_PXD = 33, SL_GSUM = 34, SL_DOT = 35, SL_SS = 36,
    SL_FZ = 37, SL_PZ = 38, SL_YS = 39, SL_YTY = 40, SL_STOP = 41,
    SL_GU = 42,        // compact L-BFGS: Gram products of the new pair, 2*CM slots right behind the trial slots
    SL_GN = 52,        // compact L-BFGS: next p, w (2*CM + 2 slots) right behind them: one exchange for all 32
    SL_AUX = 64,       // 2 slots: Lipschitz estimate / misc
    SL_OUTER = 66,     // 2 slots: outer loop
    SL_SCRATCH = 68,   // sink for partials nobody reads
    SL_GP = 69,        // compact L-BFGS: p = S'v, w = Y'v from their own pass, 2*CM slots
    SL_ZS = 79,        // 3 slots: sink of the forward-backward step that re-materialises z
    SL_COUNT = 82
};
constexpr int MAX_MEM = 16;
constexpr int CM = 5;            // capacity of the compact L-BFGS form (pairs)

// the one-pass iteration kernel of an oracle family (k_fused_compact<T, CM, NT, true, true, 2, -1, -1, FAM>)
template <class T>
using FusedFn = void (*)(CompactVecs<T, CM>, CompactCoef<CM>, const T*, const T*, ElemParams<T>, T, T*, T*, T*, T*, T*,
                         int64_t, double*, int);
template <class T, int DK> FusedFn<T> family_kernel_dk(int fam, bool nt, int uni);      // bz_families_dk<DK>.hip
template <class T> FusedFn<T> family_kernel(int fam, bool nt, int uni = -1);

struct SolverBase {
    virtual ~SolverBase() = default;
    virtual void set_multipliers(const void* mu, const void* y) = 0;
    virtual void begin(const bz_panoc_opts& o, const void* x0_host) = 0;
    virtual void step() = 0;
    virtual void steps(int64_t k) = 0;
    virtual bool should_stop() const = 0;
    virtual void finish(void* x_out, bz_panoc_stats* st) = 0;
    virtual void scalars(double* out16) = 0;
    virtual void vector(int which, void* out) = 0;
    virtual void solve(const bz_panoc_opts& o, const void* x0, void* x_out, bz_panoc_stats* st) = 0;
    virtual void alps(const bz_alps_opts& ao, const bz_panoc_opts& po, const void* x0,
                      const void* y0, void* x, void* y, void* s, void* mu, bz_alps_stats* st) = 0;
    virtual void als(const bz_alps_opts& ao, const bz_panoc_opts& po, const void* x0,
                     const void* y0, void* x, void* y, void* s, void* mu, bz_alps_stats* st) = 0;
    virtual void eval_al_gradient(const void* x, void* dlx, double* vals3) = 0;
    virtual void eval_prox(const void* x, double gamma, void* z, double* gz) = 0;
    virtual void eval_lbfgs(int m, const void* S, const void* Y, const void* v, void* d) = 0;
    virtual void halo_export(void* handle64) = 0;
    virtual void halo_connect(const void* prev64, const void* next64) = 0;
    virtual void allreduce_export(void* handle64) = 0;
    virtual void allreduce_connect(const void* handles) = 0;
    virtual void profile_enable(unsigned mask) = 0;
    virtual void profile_get(int cat, int64_t* launches, double* ms) = 0;
    virtual void profile_reset() = 0;
    virtual void profile_get2(int cat, bz_profile_rec* out) = 0;
};

bool& callback_abort_flag();      // thread-local: bz_callback_abort() was called (generic oracles)
void p2p_export(Ctx* ctx, void* handle64);
void p2p_connect(Ctx* ctx, const void* handles, const int32_t* devices);

SolverBase* make_solver(Ctx* ctx, const bz_problem_desc& d);

}  // namespace bz
