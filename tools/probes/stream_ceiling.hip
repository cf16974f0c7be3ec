// What does HBM sustain for the headline pass's stream mix (8 reads + 1 write of 80 MB each, fp64, 16 B per lane)?
// (development probe: the ceiling `k_fused_compact<XR=2>` is measured against)
//   NT    non-temporal loads / stores
//   DEPTH packs per stream in flight per lane (register software pipeline depth)
//   grid  workgroups (256 threads) — 256 = one wave per SIMD, 512 = two, ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NS = 12;
struct Ptrs { const double2* p[NS]; double2* out[2]; };
template <bool NT> __device__ __forceinline__ double2 ldv(const double2* p) {
    if (NT) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void stv(double2* p, double2 v) {
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}
template <int NR, int NW, int DEPTH, bool NTL, bool NTS, int BLK>
__global__ void __launch_bounds__(BLK) pass(Ptrs a, long npk) {
    const long stride = (long)gridDim.x * BLK;
    double2 acc = {0, 0};
    long c = (long)blockIdx.x * BLK + threadIdx.x;
    for (; c + (DEPTH - 1) * stride < npk; c += DEPTH * stride) {
        double2 v[DEPTH][NR];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int s = 0; s < NR; ++s) v[d][s] = ldv<NTL>(a.p[s] + c + d * stride);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            double2 t = {0, 0};
#pragma unroll
            for (int s = 0; s < NR; ++s) { t.x += v[d][s].x; t.y += v[d][s].y; }
            acc.x += t.x; acc.y += t.y;
#pragma unroll
            for (int w = 0; w < NW; ++w) stv<NTS>(a.out[w] + c + d * stride, t);
        }
    }
    for (; c < npk; c += stride) {
        double2 t = {0, 0};
#pragma unroll
        for (int s = 0; s < NR; ++s) { double2 v = ldv<NTL>(a.p[s] + c); t.x += v.x; t.y += v.y; }
        acc.x += t.x; acc.y += t.y;
#pragma unroll
        for (int w = 0; w < NW; ++w) stv<NTS>(a.out[w] + c, t);
    }
    if (acc.x == 1.2345e300) a.out[0][0] = acc;
}
template <int NR, int NW, int DEPTH, bool NTL, bool NTS, int BLK> double run(Ptrs a, long npk, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pass<NR, NW, DEPTH, NTL, NTS, BLK>), dim3(grid), dim3(BLK), 0, 0, a, npk);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pass<NR, NW, DEPTH, NTL, NTS, BLK>), dim3(grid), dim3(BLK), 0, 0, a, npk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10000000L, npk = n / 2;
    Ptrs a;
    for (int s = 0; s < NS; ++s) { void* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); a.p[s] = (const double2*)p; }
    for (int w = 0; w < 2; ++w) { void* p; CK(hipMalloc(&p, n * 8)); a.out[w] = (double2*)p; }
    CK(hipDeviceSynchronize());
    const double gb = n * 8 / 1e9;
#define RUN(NR, NW, D, NTL, NTS, BLK, grid) { double us = run<NR, NW, D, NTL, NTS, BLK>(a, npk, grid); \
    printf("%2dR+%dW depth %d ntl %d nts %d blk %4d grid %5d : %7.1f us  %6.0f GB/s\n", NR, NW, D, NTL, NTS, BLK, grid, us, (NR + NW) * gb / us * 1e6); }
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        RUN(1, 1, 1, false, false, 256, grid) RUN(1, 1, 4, true, true, 256, grid)
        RUN(8, 1, 1, false, false, 256, grid) RUN(8, 1, 1, true, true, 256, grid) RUN(8, 1, 1, true, false, 256, grid)
        RUN(8, 1, 2, false, false, 256, grid) RUN(8, 1, 2, true, true, 256, grid)
        RUN(8, 1, 4, true, true, 256, grid)
        RUN(8, 0, 2, true, true, 256, grid)
    }
    for (int grid : {256, 512, 1024}) {
        RUN(8, 1, 1, true, true, 512, grid) RUN(8, 1, 2, true, true, 512, grid)
        RUN(8, 1, 1, true, true, 1024, grid) RUN(8, 1, 2, true, true, 1024, grid)
    }
    return 0;
}
