// Micro-benchmark: what ONE dependent step costs when the dependence is carried inside a single launch -- the workgroups
// of "shell" d wait for the workgroups of shell d-1 through a counter in device memory -- against the same chain as
// dependent kernel launches (stream order; graphlat.hip: 1.8 us per dependent node of a replayed hipGraph plus the
// kernel's own ramp and drain).  Each workgroup reads 3 x 256 "plane" values its neighbour in the previous shell wrote,
// runs `work` dependent FMAs, writes its own and signals; every value read is checked, so the visibility protocol is
// proven, not just timed.  Block index = shell * width + column: a waiting workgroup only ever waits for lower block
// indices (the dispatcher hands workgroups out in index order), and every spin is bounded.
//   FENCE variant: ordinary stores, agent-scope release fence by every wave (buffer_wbl2 sc1), barrier, atomic add;
//                  consumer: poll, acquire fence (buffer_inv sc1), barrier, ordinary loads
//   SC1 variant:   plane stores and loads carry sc1 (device scope: they bypass the XCD-local L2 contents), a wave waits
//                  vmcnt(0) for its own stores before the barrier; no cache-wide write-back or invalidate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));

template <int AUX> __device__ __forceinline__ double ld(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, AUX));
}
template <int AUX> __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, v), r, (int)off, 0, AUX);
}

// planes: [2][width][768]; counter[d] counts the finished workgroups of shell d
template <bool SC1, bool CHAINED>
__global__ void __launch_bounds__(256) k_chain(double *planes, unsigned *counter, int width, int d0, int work, int *err, int *bad,
                                               const double *__restrict__ other)
{
    const int d = CHAINED ? (int)blockIdx.x / width : d0, c = (int)blockIdx.x % width;
    const unsigned bytes = 2u * width * 768u * 8u;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(planes, 0, bytes, 0x00020000);
    // independent prologue: something to fetch that does not depend on the previous shell (n_HI in the sweep)
    double pre = other[(size_t)blockIdx.x * 256 + threadIdx.x];
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    if (CHAINED && d > 0) {
        if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_load(&counter[d - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)width) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000) { *err = 1; ok = 0; break; }
            }
            if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!ok) return;
    }
    constexpr int A = SC1 ? 16 : 0;
    const int src = (c + 1) % width;
    double v[3];
    if (d > 0) {
        const unsigned base = (((unsigned)((d - 1) & 1) * width + src) * 768u + threadIdx.x) * 8u;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = ld<A>(r, base + k * 2048u);
        bool wrong = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) wrong |= v[k] != (double)(d - 1) * 1000.0 + src + k;
        if (wrong) atomicAdd(bad, 1);
    } else { v[0] = v[1] = v[2] = 0.0; }
    double x = v[0] + v[1] + v[2] + pre;
    for (int i = 0; i < work; ++i) x = __builtin_fma(x, 1.0, 0.0);
    const double keep = (x == -1.0) ? 1.0 : 0.0;               // never: keeps the chain alive
    const unsigned obase = (((unsigned)(d & 1) * width + c) * 768u + threadIdx.x) * 8u;
#pragma unroll
    for (int k = 0; k < 3; ++k) st<A>(r, obase + k * 2048u, (double)d * 1000.0 + c + k + keep);
    if (CHAINED) {
        if (SC1) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this wave's stores are acknowledged at device scope
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&counter[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main()
{
    const int maxw = 256, maxd = 64;
    double *planes, *other; unsigned *counter; int *err, *bad;
    CK(hipMalloc(&planes, 2ull * maxw * 768 * 8)); CK(hipMalloc(&other, (size_t)maxw * maxd * 256 * 8));
    CK(hipMemset(other, 0, (size_t)maxw * maxd * 256 * 8));
    CK(hipMalloc(&counter, maxd * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
    hipStream_t st_; CK(hipStreamCreate(&st_));
    for (int depth : {5, 40})
        for (int width : {1, 6, 24, 60, 150}) {
            for (int work : {0, 300}) {
                double res[3] = {1e30, 1e30, 1e30}; int herr = 0, hbad[3] = {0, 0, 0};
                // the launch chain as a captured graph, replayed
                hipGraph_t graph; hipGraphExec_t exec;
                CK(hipStreamBeginCapture(st_, hipStreamCaptureModeGlobal));
                for (int d = 0; d < depth; ++d) k_chain<false, false><<<width, 256, 0, st_>>>(planes, counter, width, d, work, err, bad, other);
                CK(hipStreamEndCapture(st_, &graph));
                CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                CK(hipGraphLaunch(exec, st_)); CK(hipStreamSynchronize(st_));
                for (int variant = 0; variant < 3; ++variant)
                    for (int rep = 0; rep < 5; ++rep) {
                        CK(hipMemsetAsync(counter, 0, maxd * 4, st_)); CK(hipMemsetAsync(err, 0, 4, st_)); CK(hipMemsetAsync(bad, 0, 4, st_));
                        CK(hipStreamSynchronize(st_));
                        auto t0 = std::chrono::steady_clock::now();
                        if (variant == 0) k_chain<false, true><<<width * depth, 256, 0, st_>>>(planes, counter, width, 0, work, err, bad, other);
                        else if (variant == 1) k_chain<true, true><<<width * depth, 256, 0, st_>>>(planes, counter, width, 0, work, err, bad, other);
                        else CK(hipGraphLaunch(exec, st_));
                        CK(hipStreamSynchronize(st_));
                        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                        if (us < res[variant]) res[variant] = us;
                        int e, b; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
                        herr |= e; hbad[variant] |= b;
                    }
                printf("depth %2d width %3d work %3d: fence %7.2f us (%5.2f/shell)%s  sc1 %7.2f us (%5.2f/shell)%s  graph of launches %7.2f us (%5.2f/shell)%s%s\n",
                       depth, width, work, res[0], res[0] / depth, hbad[0] ? " STALE" : "", res[1], res[1] / depth, hbad[1] ? " STALE" : "",
                       res[2], res[2] / depth, hbad[2] ? " STALE" : "", herr ? "  SPIN TIMEOUT" : "");
                CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
                if (herr) return 1;
            }
        }
    return 0;
}
