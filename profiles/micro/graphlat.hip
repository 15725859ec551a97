#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(double *x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0000001 + 1e-9; }
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
int main() {
    double *d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int chain = 70, reps = 200;
    for (int blocks : {1, 64, 1024}) {
        // stream launches
        for (int i = 0; i < chain; ++i) tiny<<<blocks, 256, 0, st>>>(d, 1 << 17);
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) { for (int i = 0; i < chain; ++i) tiny<<<blocks, 256, 0, st>>>(d, 1 << 17); CK(hipStreamSynchronize(st)); }
        double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < chain; ++i) tiny<<<blocks, 256, 0, st>>>(d, 1 << 17);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("blocks %4d: chain of %d dependent kernels: stream %.1f us (%.2f/kernel), graph %.1f us (%.2f/kernel)\n", blocks, chain, us_stream, us_stream / chain, us_graph, us_graph / chain);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
