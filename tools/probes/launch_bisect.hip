// Which of the things the library does before its steady-state loop makes a kernel launch slower?  Measures launch
// call time and call -> kernel-start-visible-on-host after each candidate action.  (Development probe.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) kflag(unsigned long long* flag, unsigned long long seq, double* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (out) out[blockIdx.x * 256 + threadIdx.x] = 1.0;
}
__global__ void kpost(double* hostmapped) { hostmapped[threadIdx.x] = 2.0; }
static unsigned long long* flag; static hipStream_t s; static unsigned long long seq = 0;
static void measure(const char* what) {
    std::vector<double> t, c;
    for (int r = 0; r < 400; ++r) {
        ++seq;
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(kflag, dim3(256), dim3(256), 0, s, flag, seq, (double*)nullptr);
        auto t1 = std::chrono::steady_clock::now();
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
        auto t2 = std::chrono::steady_clock::now();
        for (volatile int w = 0; w < 3000; ++w) {}
        t.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
        c.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end()); std::sort(c.begin(), c.end());
    std::printf("%-58s launch call %.2f us, call -> kernel start seen %.2f us\n", what, c[200], t[200]);
}
int main() {
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent));
    *flag = 0;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    measure("fresh process");
    std::vector<double*> bufs(40, nullptr);
    for (auto& b : bufs) { CK(hipMalloc((void**)&b, 80u << 20)); CK(hipMemsetAsync(b, 0, 80u << 20, nullptr)); CK(hipStreamSynchronize(nullptr)); }
    measure("after 40 x hipMalloc + hipMemsetAsync on the NULL stream");
    std::vector<double> host(10u << 20, 1.0);
    CK(hipMemcpyAsync(bufs[0], host.data(), 80u << 20, hipMemcpyDefault, s)); CK(hipStreamSynchronize(s));
    measure("after a pageable H2D hipMemcpyAsync on the stream");
    CK(hipMemcpyAsync(host.data(), bufs[0], 80u << 20, hipMemcpyDefault, s)); CK(hipStreamSynchronize(s));
    measure("after a pageable D2H hipMemcpyAsync on the stream");
    double* hm; double* hd;
    CK(hipHostMalloc((void**)&hm, 4096, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void**)&hd, hm, 0));
    hipLaunchKernelGGL(kpost, dim3(1), dim3(64), 0, s, hd); CK(hipStreamSynchronize(s));
    measure("after a kernel storing to hipHostMallocMapped memory");
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipExtLaunchKernelGGL(kflag, dim3(256), dim3(256), 0, s, a, b, 0, flag, ++seq, (double*)nullptr);
    CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    measure("after one hipExtLaunchKernelGGL with start/stop events");
    CK(hipDeviceSynchronize());
    measure("after hipDeviceSynchronize");
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    measure("after hipGetDeviceProperties");
    CK(hipMemcpyAsync(bufs[1], bufs[0], 80u << 20, hipMemcpyDeviceToDevice, s)); CK(hipStreamSynchronize(s));
    measure("after a D2D hipMemcpyAsync on the stream");
    CK(hipMemset(bufs[2], 0, 1024)); CK(hipDeviceSynchronize());
    measure("after hipMemset (NULL stream) + device sync");
    return 0;
}
