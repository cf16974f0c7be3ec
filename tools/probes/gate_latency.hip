// How fast can a resident kernel be released by the host?  One wave polls a flag in pinned host memory and
// acknowledges in pinned host memory; the host measures flag -> ack round trips.  (Development probe for the
// "pre-launched pass gated on a flag" idea in NEXT.md; not part of the library.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void gate(volatile unsigned long long* flag, volatile unsigned long long* ack, int rounds) {
    for (int r = 1; r <= rounds; ++r) {
        unsigned spins = 0;
        while (__hip_atomic_load((unsigned long long*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)r) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 200000000u) return;      // bounded
        }
        __hip_atomic_store((unsigned long long*)ack, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void empty() {}

int main() {
    unsigned long long *flag, *ack;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&ack, 64, hipHostMallocCoherent));
    *flag = 0; *ack = 0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int rounds = 2000;
    hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, s, flag, ack, rounds);
    std::vector<double> rt;
    for (int r = 1; r <= rounds; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(flag, (unsigned long long)r, __ATOMIC_RELEASE);
        while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) < (unsigned long long)r) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { std::printf("timeout\n"); return 2; }
        }
        rt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        for (volatile int k = 0; k < 2000; ++k) {}
    }
    CK(hipStreamSynchronize(s));
    std::sort(rt.begin(), rt.end());
    std::printf("flag->ack round trip (host write, GPU poll over PCIe, GPU write, host poll): median %.2f us, p10 %.2f, p90 %.2f\n",
                rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10]);
    // for comparison: launch + completion of an empty kernel seen through a pinned flag is not measurable this
    // way; time hipLaunchKernelGGL + hipStreamSynchronize instead
    std::vector<double> lt;
    for (int r = 0; r < 500; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, s);
        CK(hipStreamSynchronize(s));
        lt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(lt.begin(), lt.end());
    std::printf("empty kernel launch + hipStreamSynchronize: median %.2f us\n", lt[lt.size() / 2]);
    return 0;
}
