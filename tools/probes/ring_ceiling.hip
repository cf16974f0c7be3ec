// The headline pass's stream mix (8 reads + 1 write of 80 MB, fp64, 16 B per lane) with the loads as inline asm and explicit
// vmcnt waits — a register ring NB packs deep per stream, persistent workgroups — against tools/probes/stream_ceiling.hip,
// whose compiler-scheduled loads reach 118-121 us.  (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));
constexpr int NR = 8;
struct Ptrs { const double* p[NR]; double* out; };
template <int BI, int NB> __device__ __forceinline__ void ld_slot(v2 (&ring)[NB][NR], const Ptrs& a, unsigned off) {
#pragma unroll
    for (int s = 0; s < NR; ++s) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(ring[BI][s]) : "v"(off), "s"(a.p[s]) : "memory");
}
template <int BI, int NB, int CNT> __device__ __forceinline__ void wait_slot(v2 (&ring)[NB][NR]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ring[BI][0]), "+v"(ring[BI][1]), "+v"(ring[BI][2]), "+v"(ring[BI][3]), "+v"(ring[BI][4]),
                 "+v"(ring[BI][5]), "+v"(ring[BI][6]), "+v"(ring[BI][7]) : "n"(CNT) : "memory");
}
template <int BI, int NB> __device__ __forceinline__ void do_slot(v2 (&ring)[NB][NR], const Ptrs& a, unsigned off, unsigned off_next, v2& acc, bool store) {
    wait_slot<BI, NB, (NB - 1) * (NR + 1)>(ring);
    v2 t = {0, 0};
#pragma unroll
    for (int s = 0; s < NR; ++s) t += ring[BI][s];
    acc += t;
    if (store) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(t), "s"(a.out) : "memory");
    ld_slot<BI, NB>(ring, a, off_next);
}
// every workgroup owns a contiguous range of packs; lane t of it takes packs t, t + BLK, ...
template <int NB, int BLK> __global__ void __launch_bounds__(BLK) ring_pass(Ptrs a, long npk, int do_store) {
    const long per = (npk + gridDim.x - 1) / gridDim.x;
    const long p0 = blockIdx.x * per, p1 = p0 + per < npk ? p0 + per : npk;
    const long steps = (p1 - p0 + BLK - 1) / BLK;
    v2 ring[NB][NR];
    v2 acc = {0, 0};
    auto off_of = [&](long k) -> unsigned { long pk = p0 + k * BLK + threadIdx.x; if (pk >= p1) pk = p1 - 1; return (unsigned)(pk * 16); };
    static_assert(NB == 4, "");
    ld_slot<0, NB>(ring, a, off_of(0)); ld_slot<1, NB>(ring, a, off_of(1)); ld_slot<2, NB>(ring, a, off_of(2)); ld_slot<3, NB>(ring, a, off_of(3));
    const long nst = (steps + NB - 1) / NB * NB;
    for (long k = 0; k < nst; k += NB) {
        do_slot<0, NB>(ring, a, off_of(k), off_of(k + 4), acc, do_store && k < steps && p0 + k * BLK + threadIdx.x < p1);
        do_slot<1, NB>(ring, a, off_of(k + 1), off_of(k + 5), acc, do_store && k + 1 < steps && p0 + (k + 1) * BLK + threadIdx.x < p1);
        do_slot<2, NB>(ring, a, off_of(k + 2), off_of(k + 6), acc, do_store && k + 2 < steps && p0 + (k + 2) * BLK + threadIdx.x < p1);
        do_slot<3, NB>(ring, a, off_of(k + 3), off_of(k + 7), acc, do_store && k + 3 < steps && p0 + (k + 3) * BLK + threadIdx.x < p1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 1.2345e300) a.out[0] = acc.x;
}
template <int NB, int BLK> double run(Ptrs a, long npk, int grid, int st) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring_pass<NB, BLK>), dim3(grid), dim3(BLK), 0, 0, a, npk, st);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ring_pass<NB, BLK>), dim3(grid), dim3(BLK), 0, 0, a, npk, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10000000L, npk = n / 2;
    Ptrs a;
    for (int s = 0; s < NR; ++s) { void* p; CK(hipMalloc(&p, n * 8 + 4096)); CK(hipMemset(p, 0, n * 8)); a.p[s] = (const double*)p; }
    { void* p; CK(hipMalloc(&p, n * 8 + 4096)); a.out = (double*)p; }
    CK(hipDeviceSynchronize());
    const double gb = n * 8 / 1e9;
    for (int st : {1, 0})
        for (int grid : {256, 512, 1024}) {
            double u1 = run<4, 256>(a, npk, grid, st), u2 = run<4, 512>(a, npk, grid, st);
            printf("asm ring 8R+%dW depth 4: grid %4d blk 256: %7.1f us %6.0f GB/s | blk 512: %7.1f us %6.0f GB/s\n", st, grid, u1, (8 + st) * gb / u1 * 1e6,
                   u2, (8 + st) * gb / u2 * 1e6);
        }
    return 0;
}
