// Variants of launch_latency.hip closer to the library's situation: (a) a register-heavy kernel with LDS, (b) a
// second small kernel launched right behind the first (the read-back kernel), (c) the process also links RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) heavy(unsigned long long* flag, unsigned long long seq, const double* in, double* out, int n) {
    __shared__ double sh[4][32];
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    double a[110];
#pragma unroll
    for (int i = 0; i < 110; ++i) a[i] = in ? in[threadIdx.x + i * 256] : (double)i;
    for (int k = 0; k < n; ++k)
#pragma unroll
        for (int i = 0; i < 110; ++i) a[i] = a[i] * 1.0000001 + a[(i + 7) % 110];
    double s = 0;
#pragma unroll
    for (int i = 0; i < 110; ++i) s += a[i];
    sh[threadIdx.x >> 6][threadIdx.x & 31] = s;
    __syncthreads();
    if (out) out[blockIdx.x * 256 + threadIdx.x] = sh[0][threadIdx.x & 31];
}
__global__ void __launch_bounds__(64) small2(double* out) { if (out) out[threadIdx.x] = 1.0; }
int main() {
    int v = 0; ncclGetVersion(&v);
    unsigned long long* flag;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent));
    *flag = 0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int second = 0; second < 4; ++second) {
        const bool nosync = second >= 2;      // never synchronise the stream: only the flag tells the host
        std::vector<double> t, c, sy;
        for (int r = 1; r <= 600; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(heavy, dim3(256), dim3(256), 0, s, flag, (unsigned long long)(r + 1000 * second), (const double*)nullptr, (double*)nullptr, 0);
            auto t1 = std::chrono::steady_clock::now();
            if (second & 1) hipLaunchKernelGGL(small2, dim3(32), dim3(64), 0, s, (double*)nullptr);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned long long)(r + 1000 * second)) {}
            auto t2 = std::chrono::steady_clock::now();
            if (!nosync) {
                auto s0 = std::chrono::steady_clock::now();
                (void)hipStreamSynchronize(s);
                sy.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s0).count());
            } else {
                for (volatile int w = 0; w < 3000; ++w) {}      // let the kernels retire
            }
            t.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
            c.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(t.begin(), t.end()); std::sort(c.begin(), c.end());
        std::sort(sy.begin(), sy.end());
        std::printf("rccl %d linked, register-heavy kernel%s%s: launch call %.2f us, call start -> flag on host %.2f us, sync %.2f us\n", v,
                    (second & 1) ? " + a second launch right behind it" : "", nosync ? ", stream never synchronised" : "",
                    c[c.size() / 2], t[t.size() / 2], sy.empty() ? 0.0 : sy[sy.size() / 2]);
    }
    return 0;
}
