// Device code: the two kernels of the context's life cycle (code-object load, self-test of the division helpers).
// Included by api.hip only.
#pragma once
#include "kernels_common.hpp"

namespace c2r {

// c2r_create launches this once: the first launch of any kernel of the library makes the runtime load the whole code object
// onto the device (milliseconds) -- set-up, not something the first evolve3D of a run should pay
__global__ void k_load_code_object(int *out) { if (out) *out = 1; }

// Device self-test of the division helpers against the compiler's IEEE division (c2r_selftest).
__global__ void k_selftest_div(int n, double d_uniform, double rd_uniform, unsigned long long seed,
                               unsigned int *mismatch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i + 1);
    auto next = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    // operands spread over the magnitudes the kernels meet: 1e-40 .. 1e130
    auto rnd = [&](double lo10, double hi10) {
        const double u = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
        const double m = 1.0 + (double)(next() >> 11) * (1.0 / 9007199254740992.0);
        return m * exp10(lo10 + (hi10 - lo10) * u);
    };
    const double num = rnd(-40, 130), den = rnd(-10, 80), w = rnd(-0.3, 7);
    unsigned int bad = 0;
    if (fdiv(num, den) != num / den) bad |= 1;
    if (frcp(w) != 1.0 / w) bad |= 2;
    const double nu = rnd(-30, 30);
    if (udiv(nu, d_uniform, rd_uniform, 1) != nu / d_uniform) bad |= 4;
    const double xs = 1.0 + 2.0 * (double)(next() >> 11) * (1.0 / 9007199254740992.0);     // path^2 lies in [1, 3]
    if (fsqrt(xs) != sqrt(xs)) bad |= 8;
    if (bad) atomicAdd(mismatch, 1u);
}

}  // namespace c2r
