// How does the chunk -> thread map of a many-stream pass affect the sustained HBM rate?  (development probe, cfg-3 sizes)
//   map 0: canonical (chunk c belongs to thread c mod (grid*256): consecutive waves read consecutive KB of every stream)
//   map U: every wave reads U consecutive 1-KB pieces of each stream before it moves on (U KB contiguous per stream and wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NS = 22;
struct Ptrs { const double2* p[NS]; double2* out[2]; };
template <int NR, int NW, int U>
__global__ void __launch_bounds__(256) pass(Ptrs a, long npk) {
    const long stride = (long)gridDim.x * 256;
    double2 acc = {0, 0};
    if (U == 0) {
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < npk; c += stride) {
            double2 v[NR];
#pragma unroll
            for (int s = 0; s < NR; ++s) v[s] = a.p[s][c];
            double2 t = {0, 0};
#pragma unroll
            for (int s = 0; s < NR; ++s) { t.x += v[s].x; t.y += v[s].y; }
            acc.x += t.x; acc.y += t.y;
#pragma unroll
            for (int w = 0; w < NW; ++w) a.out[w][c] = t;
        }
    } else {
        const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, nwaves = stride >> 6;
        for (long base = wave * (64L * U); base < npk; base += nwaves * (64L * U)) {
#pragma unroll
            for (int u = 0; u < (U ? U : 1); ++u) {
                const long c = base + (long)u * 64 + lane;
                if (c < npk) {
                    double2 v[NR];
#pragma unroll
                    for (int s = 0; s < NR; ++s) v[s] = a.p[s][c];
                    double2 t = {0, 0};
#pragma unroll
                    for (int s = 0; s < NR; ++s) { t.x += v[s].x; t.y += v[s].y; }
                    acc.x += t.x; acc.y += t.y;
#pragma unroll
                    for (int w = 0; w < NW; ++w) a.out[w][c] = t;
                }
            }
        }
    }
    if (acc.x == 1.2345e300) a.out[0][0] = acc;
}
template <int NR, int NW, int U> double run(Ptrs a, long npk, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pass<NR, NW, U>), dim3(grid), dim3(256), 0, 0, a, npk);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pass<NR, NW, U>), dim3(grid), dim3(256), 0, 0, a, npk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}
int main() {
    const long n = 2048L * 2048, npk = n / 2;
    Ptrs a;
    for (int s = 0; s < NS; ++s) { void* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); a.p[s] = (const double2*)p; }
    for (int w = 0; w < 2; ++w) { void* p; CK(hipMalloc(&p, n * 8)); a.out[w] = (double2*)p; }
    CK(hipDeviceSynchronize());
    const double gb = n * 8 / 1e9;
    for (int grid : {2048, 1024, 512}) {
#define RUN(NR, NW, U) { double us = run<NR, NW, U>(a, npk, grid); printf("grid %4d  %2dR+%dW  map U=%d : %7.1f us  %6.0f GB/s\n", grid, NR, NW, U, us, (NR + NW) * gb / us * 1e6); }
        RUN(20, 2, 0) RUN(20, 2, 2) RUN(20, 2, 4) RUN(20, 2, 8)
        RUN(12, 1, 0) RUN(12, 1, 4)
        RUN(8, 1, 0) RUN(8, 1, 4)
#undef RUN
    }
    return 0;
}
