// Can the 256 MB Infinity Cache carry the newest iterate(s) from one pass of the headline kernel to the next?
// (development probe)  The pass reads the last M+1 = 6 iterates of a ring + q + b and writes the next iterate.  Every
// stream has its own cache policy: the streams marked 'keep' use the default policy (allocate), the others are
// non-temporal.  If non-temporal streams do not displace what the default-policy streams left in the Infinity Cache,
// the write of pass k and the newest read(s) of pass k+1 never reach HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NR = 8;
struct Ptrs { const double2* p[NR]; double2* out; };
template <bool NT> __device__ __forceinline__ double2 ldv(const double2* p) {
    if (NT) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void stv(double2* p, double2 v) {
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}
// KEEP: the KEEP newest iterates (streams 0..KEEP-1) are read with the default policy; WKEEP: the write allocates
template <int KEEP, bool WKEEP, bool REV>
__global__ void __launch_bounds__(256) pass(Ptrs a, long npk, int flip) {
    const long stride = (long)gridDim.x * 256;
    double2 acc = {0, 0};
    long c0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (long cc = c0; cc < npk; cc += stride) {
        const long c = (REV && flip) ? npk - 1 - cc : cc;
        double2 v[NR];
#pragma unroll
        for (int s = 0; s < NR; ++s) v[s] = (s < KEEP) ? ldv<false>(a.p[s] + c) : ldv<true>(a.p[s] + c);
        double2 t = {0, 0};
#pragma unroll
        for (int s = 0; s < NR; ++s) { t.x += v[s].x; t.y += v[s].y; }
        acc.x += t.x; acc.y += t.y;
        stv<!WKEEP>(a.out + c, t);
    }
    if (acc.x == 1.2345e300) a.out[0] = acc;
}
constexpr int RING = 8;       // M + 3
template <int KEEP, bool WKEEP, bool REV> double run(double2** ring, double2* q, double2* b, long npk, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 40;
    int k = 0;
    auto launch = [&]() {
        Ptrs a;
        for (int s = 0; s < 6; ++s) a.p[s] = ring[((k - s) % RING + RING) % RING];      // newest first
        a.p[6] = q; a.p[7] = b;
        a.out = ring[(k + 1) % RING];
        hipLaunchKernelGGL((pass<KEEP, WKEEP, REV>), dim3(grid), dim3(256), 0, 0, a, npk, k & 1);
        ++k;
    };
    for (int i = 0; i < 8; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10000000L, npk = n / 2;
    double2* ring[RING]; double2 *q, *b;
    for (int s = 0; s < RING; ++s) { CK(hipMalloc((void**)&ring[s], n * 8)); CK(hipMemset(ring[s], 0, n * 8)); }
    CK(hipMalloc((void**)&q, n * 8)); CK(hipMemset(q, 0, n * 8));
    CK(hipMalloc((void**)&b, n * 8)); CK(hipMemset(b, 0, n * 8));
    CK(hipDeviceSynchronize());
    const double gb = n * 8 / 1e9;
#define RUN(KEEP, WKEEP, REV, grid) { double us = run<KEEP, WKEEP, REV>(ring, q, b, npk, grid); \
    printf("n %ld keep-reads %d keep-write %d alternate-direction %d grid %4d : %7.1f us  (9 passes: %6.0f GB/s equivalent)\n", n, KEEP, WKEEP, REV, grid, us, 9 * gb / us * 1e6); }
    for (int grid : {256, 512, 2048}) {
        RUN(0, false, false, grid) RUN(0, true, false, grid) RUN(1, true, false, grid) RUN(2, true, false, grid) RUN(3, true, false, grid)
        RUN(8, true, false, grid)
        RUN(1, true, true, grid) RUN(2, true, true, grid) RUN(1, false, true, grid)
    }
    return 0;
}
