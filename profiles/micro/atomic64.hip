// Rate of no-return global_atomic_add_f64 in the sweep's access shape: every wave adds 64 consecutive doubles
// (512 B) of a pseudo-random 512-B row of a 134 MB array (256^3 f64), N rounds per wave.  Compared with plain
// stores and f32 atomics of the same shape.   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k(T *buf, unsigned nrows, int rounds, unsigned seed)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned x = seed + wave * 2654435761u;
    for (int r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const unsigned row = (x >> 8) % nrows;
        T *p = buf + (size_t)row * 64 + lane;
        if (MODE == 0) atomicAdd(p, (T)1.0);
        else if (MODE == 1) __builtin_nontemporal_store((T)r, p);
        else *p = *p + (T)1.0;
    }
}
template <typename T, int MODE>
double run(T *buf, size_t n, int blocks, int rounds)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned nrows = (unsigned)(n / 64);
    hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, buf, nrows, rounds, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, MODE>), dim3(blocks), dim3(256), 0, 0, buf, nrows, rounds, 7u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * rounds * sizeof(T) / (ms * 1e-3) / 1e12;
}
int main()
{
    const size_t n = (size_t)256 * 256 * 256;
    double *d; hipMalloc(&d, n * 8); hipMemset(d, 0, n * 8);
    for (int blocks : {2048, 16384}) for (int rounds : {64, 512}) {
        printf("blocks %5d rounds %3d:  atomic f64 %.2f TB/s (%.2e adds/s)   atomic f32 %.2f TB/s   store f64 %.2f TB/s   load+store f64 %.2f TB/s\n",
               blocks, rounds, run<double, 0>(d, n, blocks, rounds), run<double, 0>(d, n, blocks, rounds) * 1e12 / 8,
               run<float, 0>((float *)d, 2 * n, blocks, rounds), run<double, 1>(d, n, blocks, rounds), run<double, 2>(d, n, blocks, rounds));
    }
    return 0;
}
