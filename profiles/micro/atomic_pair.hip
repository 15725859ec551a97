// Two f64 atomics per visited cell (the non-isothermal sweep: Gamma and the heating rate): into two separate arrays at the
// same index, as k_sweep_shell_fast<HEAT> does, against ONE array of interleaved pairs (the two adds of a cell land in the
// same 32-byte sector).  Access shape of the sweep: a wave covers 64 consecutive cells of a pseudo-random row of 256^3 cells.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_pair.hip
// Measured (round 3, one MI355X): one atomic per visit 1.86e11 visits/s; two arrays 9.3e10 visits/s (the same 1.86e11 adds/s);
// interleaved pairs 4.55e10 visits/s -- HALF: an atomic instruction costs by the 64-byte sectors it touches (8 for 64
// consecutive doubles, 16 at stride 2), not by the lines that end up in DRAM.  The heating rates therefore stay in their own
// array, and the non-isothermal sweep's ceiling is 9.3e10 visits/s (it runs at 7.2 - 7.7e10, DESIGN s8a).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double *a, double *b, unsigned nrows, int rounds, unsigned seed)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned x = seed + wave * 2654435761u;
    for (int r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const unsigned row = (x >> 8) % nrows;
        const size_t c = (size_t)row * 64 + lane;
        if (MODE == 0) { atomicAdd(a + c, 1.0); }
        else if (MODE == 1) { atomicAdd(a + c, 1.0); atomicAdd(b + c, 2.0); }
        else { atomicAdd(a + 2 * c, 1.0); atomicAdd(a + 2 * c + 1, 2.0); }
    }
}
template <int MODE> double run(double *a, double *b, size_t n, int blocks, int rounds)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned nrows = (unsigned)(n / 64);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, a, b, nrows, rounds, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, a, b, nrows, rounds, 7u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * rounds / (ms * 1e-3);      // visits per second
}
int main()
{
    const size_t n = (size_t)256 * 256 * 256;
    double *a, *b; hipMalloc(&a, 2 * n * 8); hipMalloc(&b, n * 8); hipMemset(a, 0, 2 * n * 8); hipMemset(b, 0, n * 8);
    for (int blocks : {2048, 16384}) for (int rounds : {64, 512})
        printf("blocks %5d rounds %3d:  one atomic %.3e visits/s   two arrays %.3e visits/s   interleaved pairs %.3e visits/s\n", blocks, rounds,
               run<0>(a, b, n, blocks, rounds), run<1>(a, b, n, blocks, rounds), run<2>(a, b, n, blocks, rounds));
    return 0;
}
