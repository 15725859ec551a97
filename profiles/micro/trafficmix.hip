// What the MEMORY SYSTEM alone allows for the per-shell sweep's traffic mix, with no arithmetic: every "visit" of the
// shipped kernel moves  (A) one n_HI double from a 134 MB grid (a wave reads 64 consecutive cells of a pseudo-random
// row),  (B) the shell plane of the previous shell: two rows of 8-byte values per three cells, streamed once
// (non-temporal),  (C) one 8-byte plane store (non-temporal),  (D) one f64 atomic add into a second 134 MB grid at the
// n_HI position.  This kernel issues exactly those accesses for 3 "cells" per thread like k_sweep_shell_fast and
// nothing else; each stream can be switched off.  Result: visits per second of the mix = the no-compute ceiling of
// the DESIGN (not of the chip: a design that keeps the planes on chip would drop B and C).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics trafficmix.hip -o trafficmix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)

// Round 3, one box: all four streams 1.192e11 visits/s with 512-B aligned runs, 1.142e11 (-4 %) with runs that start anywhere
// (atomics alone 1.86e11 -> 1.69e11, -9 %).  The kernel's runs start anywhere: its real no-compute ceiling is ~4 % below the
// aligned figure.  Aligning them would cost 3.5 - 7 % idle lanes (rows padded to 8 cells at q = 100): not pursued.
// MASK bit 16 (round 3): the 64-cell runs of the n_HI loads and of the atomics start at a pseudo-random cell of the row instead
// of a 512-byte boundary, as in the kernel, where a run starts wherever the source's shell does (9 64-byte sectors per
// wave instruction instead of 8, 5 128-byte lines instead of 4)
template <int MASK>      // 1: n_HI loads  2: plane loads  4: plane stores  8: atomics
__global__ __launch_bounds__(256) void k_mix(const double *__restrict__ nhi, double *__restrict__ gam, const double *__restrict__ pin,
                                              double *__restrict__ pout, unsigned nrows, size_t plane_elems, unsigned seed)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned wave = (unsigned)(t >> 6), lane = threadIdx.x & 63;
    unsigned x = seed + wave * 2654435761u;
    double acc = 0.0;
    // the plane streams: thread t owns elements [3t, 3t+3) of the output and reads 4 rows x 2 columns of the input
    // (the column neighbour is the adjacent lane's element: an L1 hit, as in the kernel)
    const size_t o = (t * 3) % (plane_elems - 8);
    if (MASK & 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc += __builtin_nontemporal_load(pin + (o + r) % plane_elems);
            acc += __builtin_nontemporal_load(pin + (o + r + 1) % plane_elems);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        x = x * 1664525u + 1013904223u;
        const size_t id = (size_t)((x >> 8) % (nrows - 1)) * 64 + lane + ((MASK & 16) ? ((x >> 3) & 7u) : 0u);
        double v = 1.0;
        // MASK bit 32 (round 5): the n_HI loads come from a 2 MB window (L2-resident on every XCD) -- what an XCD-aware,
        // mesh-plane-ordered work list could at best do for that stream (the atomics execute at the memory side regardless)
        if (MASK & 1) v = (MASK & 32) ? nhi[(size_t)((x >> 8) % 4096u) * 64 + lane + ((x >> 3) & 7u)] : __builtin_nontemporal_load(nhi + id);
        if (MASK & 4) __builtin_nontemporal_store(v + acc, pout + o + c);
        if (MASK & 8) atomicAdd(gam + id, v * 1e-30);
        acc += v;
    }
    if (acc == -1.0) pout[0] = acc;
}

// The cell-major ("gather") alternative: a thread owns 3 mesh cells and loops over G sources whose face lies in the cells'
// plane: one n_HI load and ONE atomic per cell, the plane streams per (cell, source) as before.  G*3 visits per thread.
template <int G>
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ nhi, double *__restrict__ gam, const double *__restrict__ pin,
                                                 double *__restrict__ pout, unsigned nrows, size_t plane_elems, unsigned seed)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned wave = (unsigned)(t >> 6), lane = threadIdx.x & 63;
    unsigned x = seed + wave * 2654435761u;
    double acc[3] = {0.0, 0.0, 0.0};
    size_t id[3];
    double v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        x = x * 1664525u + 1013904223u;
        id[c] = (size_t)((x >> 8) % nrows) * 64 + lane;
        v[c] = __builtin_nontemporal_load(nhi + id[c]);
    }
    for (int g = 0; g < G; ++g) {
        const size_t o = (((size_t)g * gridDim.x * 256 + t) * 3) % (plane_elems - 8);     // every source's plane is its own coalesced stream
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a += __builtin_nontemporal_load(pin + (o + r) % plane_elems);
            a += __builtin_nontemporal_load(pin + (o + r + 1) % plane_elems);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { __builtin_nontemporal_store(v[c] + a, pout + o + c); acc[c] += v[c] + a; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicAdd(gam + id[c], acc[c] * 1e-30);
}

template <int G>
double run_gather(const double *nhi, double *gam, const double *pin, double *pout, unsigned nrows, size_t plane_elems, size_t nthreads)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (int)(nthreads / G / 256);
    hipLaunchKernelGGL((k_gather<G>), dim3(blocks), dim3(256), 0, 0, nhi, gam, pin, pout, nrows, plane_elems, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_gather<G>), dim3(blocks), dim3(256), 0, 0, nhi, gam, pin, pout, nrows, plane_elems, 7u + rep);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return (double)blocks * 256 * 3 * G / (best * 1e-3);
}

constexpr int kReps = 10;
template <int MASK>
double run(const double *nhi, double *gam, const double *pin, double *pout, unsigned nrows, size_t plane_elems, size_t nthreads)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (int)(nthreads / 256);
    hipLaunchKernelGGL((k_mix<MASK>), dim3(blocks), dim3(256), 0, 0, nhi, gam, pin, pout, nrows, plane_elems, 1u);
    hipLaunchKernelGGL((k_mix<MASK>), dim3(blocks), dim3(256), 0, 0, nhi, gam, pin, pout, nrows, plane_elems, 2u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < kReps; ++rep) {      // best of kReps launches of ~3 ms (round 5: three were too few -- the figure varied more than the kernel it is compared with)
        hipEventRecord(a);
        hipLaunchKernelGGL((k_mix<MASK>), dim3(blocks), dim3(256), 0, 0, nhi, gam, pin, pout, nrows, plane_elems, 7u + rep);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return (double)nthreads * 3 / (best * 1e-3);      // visits per second
}

#ifdef TRAFFICMIX_LIB
// The same measurement as a function, for bench.py (ctypes): the memory-only ceiling of the sweep's traffic mix ON THE BOX
// the bench runs on, taken in the bench process after the timed region -- so that `roofline.frac_of_memory_only_mix` does not
// divide a number of this box by a constant measured on another one (boxes differ by +-5 %).
//   mask: 31 = all four streams, runs starting anywhere (what the kernel does); 63 = the same with the n_HI loads served by the
//         L2s (what the plane-ordered mapping of round 5 approaches); 15 = 512-B aligned runs; see k_mix
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -DTRAFFICMIX_LIB -shared -fPIC trafficmix.hip -o libtrafficmix.so
extern "C" int c2r_micro_traffic_mix(int device, int mask, double *visits_per_s)
{
    if (!visits_per_s) return -1;
    CK(hipSetDevice(device));
    const size_t n = (size_t)256 * 256 * 256;
    const size_t nthreads = (size_t)1 << 27;
    const size_t plane_elems = nthreads * 3 + 64;
    double *nhi = nullptr, *gam = nullptr, *pin = nullptr, *pout = nullptr;
    CK(hipMalloc(&nhi, n * 8)); CK(hipMalloc(&gam, n * 8)); CK(hipMalloc(&pin, plane_elems * 8)); CK(hipMalloc(&pout, plane_elems * 8));
    CK(hipMemset(nhi, 0, n * 8)); CK(hipMemset(gam, 0, n * 8)); CK(hipMemset(pin, 0, plane_elems * 8)); CK(hipMemset(pout, 0, plane_elems * 8));
    const unsigned nrows = (unsigned)(n / 64);
    double v = 0.0;
    switch (mask) {
        case 31: v = run<31>(nhi, gam, pin, pout, nrows, plane_elems, nthreads); break;
        case 63: v = run<63>(nhi, gam, pin, pout, nrows, plane_elems, nthreads); break;      // ... with every n_HI load an L2 hit
        case 15: v = run<15>(nhi, gam, pin, pout, nrows, plane_elems, nthreads); break;
        case 24: v = run<24>(nhi, gam, pin, pout, nrows, plane_elems, nthreads); break;
        case 6:  v = run<6>(nhi, gam, pin, pout, nrows, plane_elems, nthreads); break;
        default: hipFree(nhi); hipFree(gam); hipFree(pin); hipFree(pout); return -2;
    }
    hipFree(nhi); hipFree(gam); hipFree(pin); hipFree(pout);
    *visits_per_s = v;
    return 0;
}
#else
int main()
{
    const size_t n = (size_t)256 * 256 * 256;                 // the 256^3 mesh
    const size_t nthreads = (size_t)1 << 27;                  // 4.0e8 visits per launch (a q = 100 shell of 1000 sources is 2.4e8)
    const size_t plane_elems = nthreads * 3 + 64;             // 3.2 GB per plane stream: far beyond L2 and MALL, like 1000 sources' planes
    double *nhi, *gam, *pin, *pout;
    CK(hipMalloc(&nhi, n * 8)); CK(hipMalloc(&gam, n * 8)); CK(hipMalloc(&pin, plane_elems * 8)); CK(hipMalloc(&pout, plane_elems * 8));
    CK(hipMemset(nhi, 0, n * 8)); CK(hipMemset(gam, 0, n * 8)); CK(hipMemset(pin, 0, plane_elems * 8)); CK(hipMemset(pout, 0, plane_elems * 8));
    const unsigned nrows = (unsigned)(n / 64);
#define RUN(M, what) printf("%-58s %.3e visits/s\n", what, run<M>(nhi, gam, pin, pout, nrows, plane_elems, nthreads))
    RUN(15, "all four streams (the shipped design's mix)");
    RUN(31, "all four streams, n_HI / atomic runs not 512-B aligned");
    RUN(63, "all four streams, unaligned, n_HI from an L2-resident 2 MB window");
    RUN(24, "atomics only, runs not 512-B aligned");
    RUN(25, "n_HI loads + atomics only, runs not 512-B aligned");
    RUN(7,  "without the Gamma atomics");
    RUN(13, "without the plane loads");
    RUN(11, "without the plane stores");
    RUN(9,  "n_HI loads + atomics only (planes on chip)");
    RUN(1,  "n_HI loads only");
    RUN(8,  "atomics only");
    RUN(6,  "plane loads + stores only");
    printf("%-58s %.3e visits/s\n", "cell-major, 2 sources per cell (1 n_HI load + 1 atomic per cell)", run_gather<2>(nhi, gam, pin, pout, nrows, plane_elems, nthreads));
    printf("%-58s %.3e visits/s\n", "cell-major, 4 sources per cell", run_gather<4>(nhi, gam, pin, pout, nrows, plane_elems, nthreads));
    printf("%-58s %.3e visits/s\n", "cell-major, 8 sources per cell", run_gather<8>(nhi, gam, pin, pout, nrows, plane_elems, nthreads));
    return 0;
}
#endif
