// Round 5: does any allocation kind let the f64 rate atomics execute in an XCD's L2 instead of at the memory side?  The sweep's access
// shape (every wave adds to 64 consecutive 8-byte slots of a pseudo-random row), a 134 MB array (beyond the L2s) and a 2 MB
// window (L2-resident), allocated with hipMalloc / hipExtMallocWithFlags(uncached | fine-grained) / hipMallocManaged.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_alloc.hip -o atomic_alloc
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double *a, unsigned nrows, int rounds, unsigned seed)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned x = seed + wave * 2654435761u;
    for (int r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const size_t id = (size_t)((x >> 8) % (nrows - 1)) * 64 + lane + ((x >> 3) & 7u);
        atomicAdd(a + id, 1.0);
    }
}
double rate(double *a, size_t n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 16384, rounds = 256;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a, (unsigned)(n / 64), 8, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a, (unsigned)(n / 64), rounds, 7u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return (double)blocks * 256 * rounds / (best * 1e-3);
}
int main()
{
    const size_t big = (size_t)256 * 256 * 256, small = (size_t)256 * 1024;      // 134 MB, 2 MB
    const char *names[4] = {"hipMalloc", "hipExtMallocWithFlags(uncached)", "hipExtMallocWithFlags(fine-grained)", "hipMallocManaged"};
    for (int kind = 0; kind < 4; ++kind) {
        double *a = nullptr;
        hipError_t e = hipSuccess;
        if (kind == 0) e = hipMalloc(&a, big * 8);
        if (kind == 1) e = hipExtMallocWithFlags((void **)&a, big * 8, hipDeviceMallocUncached);
        if (kind == 2) e = hipExtMallocWithFlags((void **)&a, big * 8, hipDeviceMallocFinegrained);
        if (kind == 3) e = hipMallocManaged(&a, big * 8);
        if (e != hipSuccess || !a) { printf("%-40s allocation failed (%s)\n", names[kind], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipMemset(a, 0, big * 8); hipDeviceSynchronize();
        printf("%-40s f64 adds/s: 134 MB array %.3e   2 MB window %.3e\n", names[kind], rate(a, big), rate(a, small));
        hipFree(a);
    }
    return 0;
}
