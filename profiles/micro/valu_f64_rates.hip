// Round 5: issue cost of the f64 vector instructions the sweep kernel is made of, one SIMD's view: 8 waves per SIMD (as the
// plane-ordered kernel runs), every wave a stream of N independent chains of ONE instruction kind; cycles per wave-instruction
// = elapsed shader cycles x waves-per-SIMD-normalised.  hipcc --offload-arch=gfx950 -O3 valu_f64_rates.hip -o valu_f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, double seed, int iters)
{
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + 0.001 * (threadIdx.x + i);
    const double c = 1.0000001, d = 0.9999999;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) x[i] = __builtin_fma(x[i], c, d);
            if (KIND == 1) x[i] = __builtin_amdgcn_rcp(x[i]);
            if (KIND == 2) x[i] = __builtin_amdgcn_rsq(x[i]);
            if (KIND == 3) x[i] = x[i] * c;
            if (KIND == 4) x[i] = x[i] + d;
            if (KIND == 5) x[i] = __builtin_fmax(x[i], d);
            if (KIND == 6) x[i] = __builtin_amdgcn_fract(x[i]) + c;            // fract + add
            if (KIND == 7) x[i] = __builtin_amdgcn_frexp_mant(x[i]) + c;       // frexp_mant + add
            if (KIND == 8) x[i] = (double)(int)x[i] + d;                       // cvt_i32_f64 + cvt_f64_i32 + add
            if (KIND == 9) x[i] = (double)__builtin_amdgcn_rcpf((float)x[i]);  // cvt_f32_f64 + rcp_f32 + cvt_f64_f32
            if (KIND == 10) x[i] = __builtin_sqrt(x[i]);                        // library sqrt
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 1.2345e-300) out[0] = s;
}
template <int KIND> double run(const char *name, int per_iter_extra)
{
    double *out; hipMalloc(&out, 8);
    const int iters = 4000, blocks = 256 * 8;          // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, 1.5, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, 1.5, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // per SIMD: 8 waves x iters x 8 instructions (of the kind; the "+ add" kinds carry one more each)
    const double inst_per_simd = 8.0 * iters * 8;
    const double ns_per_inst = best * 1e6 / inst_per_simd;
    printf("%-44s %7.2f ns per wave-instruction-group on a SIMD  (= %5.1f cycles at 2.4 GHz)\n", name, ns_per_inst, ns_per_inst * 2.4);
    hipFree(out);
    return ns_per_inst;
}
int main()
{
    run<0>("v_fma_f64", 0); run<3>("v_mul_f64", 0); run<4>("v_add_f64", 0); run<5>("v_max_f64", 0);
    run<1>("v_rcp_f64", 0); run<2>("v_rsq_f64", 0);
    run<6>("v_fract_f64 + v_add_f64", 0); run<7>("v_frexp_mant_f64 + v_add_f64", 0);
    run<8>("cvt f64->i32->f64 + v_add_f64", 0); run<9>("cvt f64->f32, v_rcp_f32, cvt f32->f64", 0);
    run<10>("sqrt (library, f64)", 0);
    return 0;
}
