// How accurate are v_rcp_f64 / v_rsq_f64 on gfx950?  (decides how many Newton steps an exact division needs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(unsigned long long seed, int n, double *out /* [2]: max rel err rcp, rsq (as ulp of 2^-52) */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long x = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i + 1);
    double mr = 0, ms = 0;
    for (int k = 0; k < n; ++k) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const double m = 1.0 + (double)(x >> 11) * (1.0 / 9007199254740992.0);   // [1,2)
        const double r = __builtin_amdgcn_rcp(m);
        const double e = fabs(__builtin_fma(-m, r, 1.0));      // |1 - m r|, exact to one rounding
        mr = fmax(mr, e);
        const double m4 = m * ((x & 1) ? 2.0 : 1.0);             // [1,4)
        const double y = __builtin_amdgcn_rsq(m4);
        const double e2 = fabs(__builtin_fma(-m4 * y, y, 1.0));  // |1 - m y^2| ~ 2 * rel err
        ms = fmax(ms, e2);
    }
    atomicMax((unsigned long long *)&out[0], (unsigned long long)__double_as_longlong(mr));
    atomicMax((unsigned long long *)&out[1], (unsigned long long)__double_as_longlong(ms));
}
int main()
{
    double *d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
    k<<<4096, 256>>>(12345ULL, 2000, d);
    double h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("samples %.3g\nrcp_f64: max |1 - x*rcp(x)| = %.3e = 2^%.2f\nrsq_f64: max |1 - x*rsq(x)^2| = %.3e = 2^%.2f\n", 4096.0 * 256 * 2000, h[0], log2(h[0]), h[1], log2(h[1]));
    return 0;
}
