// Where do f64 atomic adds execute?  Agent scope (atomicAdd) goes to the memory side on MI355X (1.5 TB/s of added
// bytes, atomic64.hip).  This probe gives every XCD its own replica of the array (chosen by HW_REG_XCC_ID) and adds
// with WORKGROUP / WAVEFRONT scope (no sc bits): if those run in the XCD's L2 they should approach the plain
// load+store rate -- and must still lose no add, since all adders of a replica share that L2.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 7u; }
template <int SCOPE, bool REPL>
__global__ __launch_bounds__(256) void k(double *buf, size_t nper, unsigned nrows, int rounds, unsigned seed, unsigned *xcc_seen)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1u << xcc);
    double *base = REPL ? buf + (size_t)xcc * nper : buf;
    unsigned x = seed + wave * 2654435761u;
    for (int r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const unsigned row = (x >> 8) % nrows;
        double *p = base + (size_t)row * 64 + lane;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}
template <int SCOPE, bool REPL>
void run(double *d, size_t nper, int blocks, int rounds, const char *name)
{
    unsigned *seen; hipMalloc(&seen, 4); hipMemset(seen, 0, 4);
    hipMemset(d, 0, nper * 8 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned nrows = (unsigned)(nper / 64);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<SCOPE, REPL>), dim3(blocks), dim3(256), 0, 0, d, nper, nrows, rounds, 7u, seen);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // every add is +1.0: the grand total over all replicas must equal the number of adds exactly
    std::vector<double> h(nper * 8);
    hipMemcpy(h.data(), d, nper * 8 * 8, hipMemcpyDeviceToHost);
    double tot = 0; for (double v : h) tot += v;
    unsigned s; hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost);
    const double adds = (double)blocks * 256 * rounds;
    printf("%-34s %.2f TB/s  (%.2e adds/s)  lost adds: %.0f of %.0f   xcc mask %02x\n", name, adds * 8 / (ms * 1e-3) / 1e12,
           adds / (ms * 1e-3), adds - tot, adds, s);
    hipFree(seen);
}
int main()
{
    const size_t nper = (size_t)256 * 256 * 256;          // 134 MB per replica
    double *d; hipMalloc(&d, nper * 8 * 8);
    for (int rounds : {64, 256}) {
        printf("rounds %d, 16384 blocks\n", rounds);
        run<0, false>(d, nper, 16384, rounds, "agent scope, one array");
        run<0, true>(d, nper, 16384, rounds, "agent scope, per-XCD replicas");
        run<1, true>(d, nper, 16384, rounds, "workgroup scope, per-XCD replicas");
        run<2, true>(d, nper, 16384, rounds, "wavefront scope, per-XCD replicas");
        run<1, false>(d, nper, 16384, rounds, "workgroup scope, ONE array (unsafe)");
    }
    return 0;
}
