// Can the host store directly into fine-grained DEVICE memory (large BAR), and how long until a resident kernel sees it?
// (development probe for the gated pre-launch: the gate record in device memory instead of pinned host memory)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void waiter(volatile unsigned long long* flag, unsigned long long* ack_host, int rounds) {
    for (int r = 1; r <= rounds; ++r) {
        unsigned spins = 0;
        while (__hip_atomic_load((unsigned long long*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)r) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 50000000u) return;
        }
        __hip_atomic_store(ack_host, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int main() {
    void* dev = nullptr;
    CK(hipExtMallocWithFlags(&dev, 4096, hipDeviceMallocFinegrained));
    CK(hipMemset(dev, 0, 4096));
    unsigned long long* ack = nullptr; unsigned long long* ack_dev = nullptr;
    CK(hipHostMalloc((void**)&ack, 64, hipHostMallocMapped));
    *ack = 0;
    CK(hipHostGetDevicePointer((void**)&ack_dev, ack, 0));
    hipPointerAttribute_t at;
    CK(hipPointerGetAttributes(&at, dev));
    printf("finegrained device memory: hostPointer=%p devicePointer=%p type=%d\n", at.hostPointer, at.devicePointer, (int)at.type);
    CK(hipDeviceSynchronize());
    const int rounds = 2000;
    hipLaunchKernelGGL(waiter, dim3(1), dim3(64), 0, 0, (volatile unsigned long long*)dev, ack_dev, rounds);
    volatile unsigned long long* hp = (volatile unsigned long long*)dev;      // host store straight into device memory
    double tot = 0, mx = 0;
    for (int r = 1; r <= rounds; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        *hp = (unsigned long long)r;
        __sync_synchronize();
        while (*(volatile unsigned long long*)ack != (unsigned long long)r) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { printf("timeout at round %d\n", r); return 2; }
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        tot += us; mx = us > mx ? us : mx;
    }
    CK(hipDeviceSynchronize());
    printf("host store -> device poll -> host ack round trip: mean %.2f us, max %.2f us over %d rounds\n", tot / rounds, mx, rounds);
    return 0;
}
