// Round 5: is an order-independent (exact, fixed-point) accumulation of the rates affordable?  Rate of no-return 64-bit INTEGER
// atomic adds against the f64 atomic add of the shipped kernel, in the sweep's access shape (every wave adds to 64 consecutive
// 8-byte slots of a pseudo-random 512-byte row of a 134 MB array), and of the pair an exact accumulator would need: two u64 adds
// per visit into two separate 134 MB arrays (high and low word; interleaving them in one array touches twice the sectors).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_u64.hip -o atomic_u64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>      // 0: f64 atomic add   1: u64 atomic add   2: two u64 adds, two arrays   3: f64 add + u64 add, two arrays   4: two f64 adds, two arrays   5: two f64 adds, interleaved in one array (a[2 id], a[2 id + 1]; a must hold 2 n)
__global__ __launch_bounds__(256) void k(double *a, double *b, unsigned nrows, int rounds, unsigned seed)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned x = seed + wave * 2654435761u;
    for (int r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const size_t id = (size_t)((x >> 8) % (nrows - 1)) * 64 + lane + ((x >> 3) & 7u);     // runs start anywhere, as in the kernel
        if (MODE == 0) atomicAdd(a + id, 1.0);
        if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long *>(a) + id, 3ULL);
        if (MODE == 2) { atomicAdd(reinterpret_cast<unsigned long long *>(a) + id, 3ULL); atomicAdd(reinterpret_cast<unsigned long long *>(b) + id, 5ULL); }
        if (MODE == 3) { atomicAdd(a + id, 1.0); atomicAdd(reinterpret_cast<unsigned long long *>(b) + id, 5ULL); }
        if (MODE == 4) { atomicAdd(a + id, 1.0); atomicAdd(b + id, 1.0); }
        if (MODE == 5) { atomicAdd(a + 2 * id, 1.0); atomicAdd(a + 2 * id + 1, 1.0); }
    }
}
template <int MODE>
double run(double *a, double *b, size_t n, int blocks, int rounds)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned nrows = (unsigned)(n / 64);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, a, b, nrows, rounds, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, a, b, nrows, rounds, 7u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return (double)blocks * 256 * rounds / (best * 1e-3);      // visits per second
}
int main()
{
    const size_t n = (size_t)256 * 256 * 256;
    double *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 8); hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 8);
    for (int blocks : {16384}) for (int rounds : {64, 512}) {
        printf("blocks %d rounds %d, visits/s:  f64 add %.3e   u64 add %.3e   two u64 adds (two arrays) %.3e   f64 + u64 %.3e   two f64 adds %.3e   two f64 adds interleaved %.3e\n", blocks, rounds,
               run<0>(a, b, n, blocks, rounds), run<1>(a, b, n, blocks, rounds), run<2>(a, b, n, blocks, rounds), run<3>(a, b, n, blocks, rounds), run<4>(a, b, n, blocks, rounds), run<5>(a, b, n, blocks, rounds));
    }
    return 0;
}
