// Launch call -> first instruction of the kernel, as seen from the host through a pinned flag, for different
// kernel-argument sizes and grids.  (Development probe: what the ~9 us between "scalars on the host" and "next
// pass running" are made of.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Small { unsigned long long* flag; unsigned long long seq; };
struct Big { unsigned long long* flag; unsigned long long seq; double pad[56]; };      // 464 bytes
template <class A> __global__ void __launch_bounds__(256) k(A a) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <class A> double run(hipStream_t s, unsigned long long* flag, int grid, double* call_us) {
    std::vector<double> t, c;
    for (int r = 1; r <= 600; ++r) {
        A a{}; a.flag = flag; a.seq = (unsigned long long)r;
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k<A>, dim3(grid), dim3(256), 0, s, a);
        auto t1 = std::chrono::steady_clock::now();
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned long long)r) {}
        auto t2 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(s);
        t.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
        c.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end()); std::sort(c.begin(), c.end());
    *call_us = c[c.size() / 2];
    return t[t.size() / 2];
}
struct V11 { const double* S[5]; const double* Y[5]; int m; };
struct C13 { double u1[5], u2h[5], H0, gam0; };
struct P18 { int a, b, c, d; const double *q, *bb, *mu, *muy; double gl, gp; const double* gu; double glo, ghi; const double *glv, *ghv; double dlo, dhi; const double *dlv, *dhv; double muu; };
__global__ void __launch_bounds__(256) kmany(V11 v, C13 c, const double* x, const double* rp, P18 p, double gamma, double* xd, double* z,
                                             double* res, double* sn, double* yn, long n, double* parts, int slot0,
                                             unsigned long long* flag, unsigned long long seq) {
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
    {
        unsigned long long* flag;
        CK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent));
        *flag = 0;
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        std::vector<double> t, c;
        V11 v{}; C13 cc{}; P18 pp{};
        // live device allocations as in a solver (40 vectors of 80 MB), real pointers in the top-level arguments
        std::vector<double*> bufs(40, nullptr);
        for (auto& b : bufs) CK(hipMalloc((void**)&b, 80u << 20));
        for (int pass = 0; pass < 2; ++pass) {
        t.clear(); c.clear();
        for (int r = 1; r <= 600; ++r) {
            double* A = pass ? bufs[r % 8] : nullptr;
            double* B = pass ? bufs[8 + r % 8] : nullptr;
            double* Cc = pass ? bufs[16 + r % 8] : nullptr;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(kmany, dim3(256), dim3(256), 0, s, v, cc, (const double*)A, (const double*)B, pp, 0.5,
                               Cc, pass ? bufs[24] : nullptr, pass ? bufs[25] : nullptr, pass ? bufs[26] : nullptr,
                               pass ? bufs[27] : nullptr, 10L, pass ? bufs[28] : nullptr, 0, flag, (unsigned long long)(r + 1000 * pass));
            auto t1 = std::chrono::steady_clock::now();
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned long long)(r + 1000 * pass)) {}
            auto t2 = std::chrono::steady_clock::now();
            (void)hipStreamSynchronize(s);
            t.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
            c.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(t.begin(), t.end()); std::sort(c.begin(), c.end());
        std::printf("16 separate arguments (the one-pass kernel's signature), %s: launch call %.2f us, call start -> flag on host %.2f us\n",
                    pass ? "real device pointers, 40 live 80 MB allocations" : "null pointers", c[c.size() / 2], t[t.size() / 2]);
        }
        for (auto& b : bufs) (void)hipFree(b);
        *flag = 0;
    }
    unsigned long long* flag;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent));
    *flag = 0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double c;
    for (int grid : {1, 256, 512}) {
        double a = run<Small>(s, flag, grid, &c);
        std::printf("grid %4d, 16-byte kernarg : launch call %.2f us, call start -> flag on host %.2f us\n", grid, c, a);
        *flag = 0;
        double b = run<Big>(s, flag, grid, &c);
        std::printf("grid %4d, 464-byte kernarg: launch call %.2f us, call start -> flag on host %.2f us\n", grid, c, b);
        *flag = 0;
    }
    return 0;
}
