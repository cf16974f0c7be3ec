// Does a scalar load with GLC (s_load_dwordx2 ... glc: bypasses the scalar cache, served by L2) see another workgroup's
// agent-scope stores — same XCD, other XCD?  And what does a ping-pong through such words cost against vector sc1 polls?
// build: hipcc -O3 --offload-arch=gfx950 spoll.hip -o spoll ; run: ./spoll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long sload(const unsigned long long* p) {
    u2v v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return ((unsigned long long)v[1] << 32) | v[0];
}
__device__ __forceinline__ unsigned long long vload(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vstore(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// blocks a and b ping-pong `rounds` times through words w[0] (a -> b) and w[8] (b -> a); mode 0: vector polls, 1: scalar polls
__global__ void pingpong(unsigned long long* w, int a, int b, int rounds, int mode, long long* cycles, int* fail) {
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    if (threadIdx.x != 0) return;
    const bool isA = (int)blockIdx.x == a;
    unsigned long long* mine = isA ? w : w + 8;
    const unsigned long long* theirs = isA ? w + 8 : w;
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (isA) vstore(mine, (unsigned long long)r);
        unsigned spins = 0;
        for (;;) {
            const unsigned long long v = mode ? sload(theirs) : vload(theirs);
            if (v >= (unsigned long long)r) break;
            if (++spins > 2000000u) { *fail = r; return; }
        }
        if (!isA) vstore(mine, (unsigned long long)r);
    }
    if (isA) *cycles = wall_clock64() - t0;
}
int main() {
    unsigned long long* w;
    long long* cyc; int* fail;
    CK(hipMalloc(&w, 4096)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&fail, 4));
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 255}};
    const char* names[3] = {"blocks 0,8 (same XCD if round-robin)", "blocks 0,1 (neighbouring XCDs)", "blocks 0,255"};
    for (int mode = 0; mode < 2; ++mode)
        for (int p = 0; p < 3; ++p) {
            CK(hipMemset(w, 0, 4096)); CK(hipMemset(cyc, 0, 8)); CK(hipMemset(fail, 0, 4));
            const int rounds = 2000;
            hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, w, pairs[p][0], pairs[p][1], rounds, mode, cyc, fail);
            CK(hipDeviceSynchronize());
            long long c; int f;
            CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            printf("%s polls, %s: %s, round trip %.3f us\n", mode ? "scalar(glc)" : "vector(sc1)", names[p],
                   f ? "NEVER SAW the store (timeout)" : "ok", f ? 0.0 : (double)c / rounds / 100.0);      // wall_clock64: 100 MHz
        }
    return 0;
}
