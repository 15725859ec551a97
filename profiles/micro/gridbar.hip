// Micro-benchmark: what a grid-wide barrier inside ONE persistent kernel costs on MI355X, against the 1.8 us per
// dependent kernel of a hipGraph replay (graphlat.hip).  Monotonic counter in device memory, agent-scope release /
// acquire fences so the planes one workgroup writes are visible to every other (the XCDs' L2s are not coherent with
// each other).  Each round every workgroup writes a value, crosses the barrier and checks the value of the workgroup
// "opposite" to it (another XCD), so the fences are proven sufficient, not just timed.  Bounded spin: no hang.
//   variants: all workgroups participate | only workgroups on XCD 0 (blockIdx % 8 == 0) participate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *count, unsigned target, int *err)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                                       // release: my stores reach memory
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000) { *err = 1; ok = false; break; }
        }
        __threadfence();                                                       // acquire
    }
    __syncthreads();
    return ok;
}

// Variant without contended atomics: every workgroup stores the round number into its OWN slot; workgroup 0 polls all
// slots with one load per thread, then publishes the round in a release word every other workgroup polls.
__device__ __forceinline__ bool grid_barrier_flags(unsigned *arrive, unsigned *go, unsigned round, int wg, int nwg, int *err)
{
    __syncthreads();
    __shared__ int sh_ok;
    if (threadIdx.x == 0) { sh_ok = 1; __threadfence(); __hip_atomic_store(&arrive[wg * 16], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (wg == 0) {
        for (int w = threadIdx.x; w < nwg; w += blockDim.x) {
            long spins = 0;
            while (__hip_atomic_load(&arrive[w * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round)
                if (++spins > 4000000) { *err = 1; sh_ok = 0; break; }
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(go, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 4000000) { *err = 1; sh_ok = 0; break; }
        }
    }
    if (threadIdx.x == 0) __threadfence();
    __syncthreads();
    return sh_ok != 0;
}

template <bool FENCE>
__global__ void __launch_bounds__(256) k_rounds_flags(unsigned *arrive, unsigned *go, double *slots, int rounds, int *err, int *bad)
{
    const int wg = blockIdx.x, nwg = gridDim.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) slots[(size_t)(r & 1) * nwg + wg] = (double)(r + 1) * (wg + 1);
        if (!grid_barrier_flags(arrive, go, (unsigned)(r + 1), wg, nwg, err)) return;
        int other = (wg + nwg / 2 + 1) % nwg;
        double got = __builtin_nontemporal_load(&slots[(size_t)(r & 1) * nwg + other]);
        if (threadIdx.x == 0 && got != (double)(r + 1) * (other + 1)) atomicAdd(bad, 1);
        acc += got;
    }
    if (acc == -1.0) slots[0] = acc;
}

template <bool ONE_XCD>
__global__ void __launch_bounds__(256) k_rounds(unsigned *count, double *slots, int rounds, int work, int *err, int *bad)
{
    int wg = blockIdx.x, nwg = gridDim.x;
    if (ONE_XCD) { if (wg % 8) return; wg /= 8; nwg = (nwg + 7) / 8; }
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        // "work": `work` dependent fmas per thread, then one plane value per workgroup
        double v = (double)(r + 1) * (wg + 1);
        for (int i = 0; i < work; ++i) v = __builtin_fma(v, 1.0, 0.0);
        if (threadIdx.x == 0) slots[(size_t)(r & 1) * nwg + wg] = v;
        if (!grid_barrier(count, (unsigned)(r + 1) * nwg, err)) return;
        int other = (wg + nwg / 2 + 1) % nwg;
        double got = __builtin_nontemporal_load(&slots[(size_t)(r & 1) * nwg + other]);
        if (threadIdx.x == 0 && got != (double)(r + 1) * (other + 1)) atomicAdd(bad, 1);
        acc += got;
    }
    if (acc == -1.0) slots[0] = acc;
}

int main()
{
    unsigned *count; double *slots; int *err, *bad;
    CK(hipMalloc(&count, 4)); CK(hipMalloc(&slots, 2 * 4096 * 8)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int rounds = 2000;
    for (int one_xcd = 0; one_xcd < 2; ++one_xcd)
        for (int nwg : {8, 32, 64, 128, 256, 512}) {
            int grid = one_xcd ? nwg * 8 : nwg;
            if (grid > 2048) continue;                         // all must be co-resident: 256 CUs x 8 workgroups of 256
            for (int work : {0, 200}) {
                double best = 1e30; int herr = 0, hbad = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(count, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(bad, 0, 4, st));
                    CK(hipStreamSynchronize(st));
                    auto t0 = std::chrono::steady_clock::now();
                    if (one_xcd) k_rounds<true><<<grid, 256, 0, st>>>(count, slots, rounds, work, err, bad);
                    else k_rounds<false><<<grid, 256, 0, st>>>(count, slots, rounds, work, err, bad);
                    CK(hipStreamSynchronize(st));
                    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
                    if (us < best) best = us;
                    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
                    if (herr) break;
                }
                printf("%s %4d workgroups, %3d fma of work: %.2f us per round%s%s\n", one_xcd ? "XCD0 only" : "all XCDs ", nwg, work, best,
                       herr ? "  SPIN TIMEOUT" : "", hbad ? "  STALE DATA SEEN" : "");
                if (herr) return 1;
            }
        }
    unsigned *arrive, *go;
    CK(hipMalloc(&arrive, 4096 * 64)); CK(hipMalloc(&go, 64));
    for (int nwg : {8, 32, 64, 128, 256, 512}) {
        double best = 1e30; int herr = 0, hbad = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(arrive, 0, 4096 * 64, st)); CK(hipMemsetAsync(go, 0, 64, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(bad, 0, 4, st));
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            k_rounds_flags<true><<<nwg, 256, 0, st>>>(arrive, go, slots, rounds, err, bad);
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
            if (us < best) best = us;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            if (herr) break;
        }
        printf("flag slots %4d workgroups: %.2f us per round%s%s\n", nwg, best, herr ? "  SPIN TIMEOUT" : "", hbad ? "  STALE DATA SEEN" : "");
        if (herr) return 1;
    }
    return 0;
}
