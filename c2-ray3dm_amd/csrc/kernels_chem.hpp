// Device code, part 3 of 4: the global pass (evolve0D_global + do_chemistry + doric + thermal), the photon-statistics
// sums and the fixed-order reductions.  Included by chemistry.hip only (it defines non-template kernels).
#pragma once
#include "kernels_common.hpp"

namespace c2r {

// ---- global pass -------------------------------------------------------------------------------
struct ChemParams {
    const StepBlock *step;        // dt, brech0, acolh0, recpow, clumping, sqrtt, expt, zp, dzdt below are filled from step->chem at kernel entry
    double dt, eps, min_frac_change, min_frac_atoms, abu_c, deltht_small;
    double brech0, acolh0;        // doric.f90:73,78 evaluated on the host for the step's temperature
    double bh00, recpow;          // brech0 = clumping*bh00*recpow when clumping comes from a grid
    const float *clump;           // clumping_grid (clumping_module.F90:116) or null
    int max_iter;
    // STATS variant: the four mesh sums of photonstatistics.F90 over (xh_intermed, xh_av) as this pass leaves them
    double clumping, colh0, sqrtt, expt;
    double *stat_partial;         // [4][gridDim.x]
    // THERMAL variant (c2ray_parameters.f90:28 isothermal=.false.): temperature_grid, phiheat_grid, the cooling curve
    float *temper;                // temperature_module.F90:35: (current, average, intermed) f32 per cell
    const double *phiheat;        // evolve_data.F90:42
    const double *cool;           // cooling.f90:27 cie_cool(1:cool_points), linear
    double cool_mintemp, cool_dtemp;
    int cool_points, thermal_max_steps;
    double k_B, gamma1, minitemp, rel_denergy, rate_floor, time_tol;   // tped.f90, atomic.f90:25, c2ray_parameters.f90:108-110, thermal.f90:117,160
    double zp, dzdt;              // cosmology.F90:198-225 cosmo_cool = e*2/(1+zred)*dzdt (dzdt = 0: not cosmological)
    double temph0, albpow;        // doric.f90:73-78 at the cell's own temperature
    double tconv_rel, tconv_abs;  // evolve_point.F90:387-388
    // drivers built with -DALLFRAC (c2r_params.allfrac): the stored neutral fractions, the (:,:,:,0) halves of xh / xh_av / xh_intermed
    // (null: the shipped build, neutral = 1 - ionized)
    const double *xh0; double *xh_av0, *xh_intermed0;
};

// cooling.f90:38-59 coolin
__device__ __forceinline__ double coolin_dev(const ChemParams &c, double nucldens, double eldens, double temp0)
{
    const double tpos = (log10(temp0) - c.cool_mintemp) / c.cool_dtemp + 1.0;
    const int itpos = min(c.cool_points - 1, max(1, (int)tpos));
    const double dtpos = tpos - (double)itpos;
    const int itpos1 = min(c.cool_points, itpos + 1);
    const double c0 = c.cool[itpos - 1], c1 = c.cool[itpos1 - 1];
    return nucldens * eldens * (c0 + (c1 - c0) * dtpos);
}

// thermal.f90:22-189: explicit sub-stepping of the internal energy, each sub-step limited to rel_denergy of the
// thermal time scale.  t_final / t_average are left untouched when t_initial <= minitemp (:83).
__device__ __forceinline__ void thermal_dev(const ChemParams &c, double t_initial, double &t_final, double &t_average,
                                            double ndens_electron, double nd, double h_old1, double h_av1, double h1, double heating)
{
    const double ne_old = nd * (h_old1 + c.abu_c), ne_av = nd * (h_av1 + c.abu_c), ne_new = nd * (h1 + c.abu_c);   // tped.f90:81
    double e_int = (nd + ne_old) * c.k_B * t_initial / c.gamma1;                    // :66 temper2pressr/(gamma1)
    const double cosmo_cool_rate = e_int * 2.0 / c.zp * c.dzdt;                     // :73-76, cosmology.F90:223
    if (!(t_initial > c.minitemp)) return;
    double cumulative = 0.0, avg = 0.0, t_int = t_initial;
    int i_heating = 0;
    for (;;) {
        i_heating++;
        const double cooling = coolin_dev(c, nd, ndens_electron, t_int) + cosmo_cool_rate;        // :104
        const double rate = fmax(c.rate_floor, fabs(cooling - heating));
        const double timescale = e_int / fabs(rate);
        const double dt_thermal = c.rel_denergy * timescale;
        const double dt_ode = fmin(dt_thermal, c.dt - cumulative);                  // :127
        e_int = e_int + dt_ode * (heating - cooling);
        avg = avg + 0.5 * t_int * dt_ode;
        t_int = e_int * c.gamma1 / (c.k_B * (nd + ne_av));                          // :137 pressr2temper
        avg = avg + 0.5 * t_int * dt_ode;
        if (t_int < c.minitemp) {                                                   // :147-153
            e_int = (nd + ne_av) * c.k_B * c.minitemp;
            t_int = c.minitemp;
        }
        cumulative = cumulative + dt_ode;
        if (cumulative >= c.dt || fabs(cumulative - c.dt) < c.time_tol * c.dt) break;   // :160
        if (i_heating > c.thermal_max_steps) break;                                 // :163
    }
    t_average = c.dt > 0.0 ? avg / c.dt : t_initial;                                // :168-172
    t_final = e_int * c.gamma1 / (c.k_B * (nd + ne_new));                           // :175
}

// evolve0D_global (evolve_point.F90:305-406) + do_chemistry (:410-555) + doric (doric.f90:33-134).
// Fixed grid, grid-stride: block partial sums of xh_intermed land in sum_partial[blockIdx.x].
// STATS: also what k_photon_sums(xh_intermed, xh_av) would return after this pass -- the values are in registers here --
// accumulated in the same order over the same grid, so the sums are bit-identical to the separate kernel's and the
// 20 bytes per cell it reads are saved (evolve.F90:570 calculate_photon_statistics after every global pass).
// THERMAL: the non-isothermal do_chemistry -- doric at the cell's own (time-averaged) temperature, thermal after every
// doric call (evolve_point.F90:515-527), the temperature clause of the global convergence test (:387-388) and
// set_temperature_point (:553; f32 stores of %intermed and %average).
template <bool STATS, bool THERMAL>
__global__ __launch_bounds__(256) void k_global_pass(ChemParams c, size_t ncell, const float *__restrict__ ndens,
                                                     const double *__restrict__ xh, double *__restrict__ xh_av,
                                                     double *__restrict__ xh_intermed,
                                                     const double *__restrict__ phih, double *sum_partial,
                                                     unsigned long long *conv_flag, unsigned int *chem_fail,
                                                     const int *gate = nullptr)
{
    __shared__ double sm[4];
    if (gate && *gate != 0) return;
    {   // the step's constants (dt, doric's rate coefficients at the step's temperature, cosmo_cool's redshift): device-resident
        const C2R_AS4 ChemStep &st = ((const C2R_AS4 StepBlock *)c.step)->chem;
        c.dt = st.dt; c.brech0 = st.brech0; c.acolh0 = st.acolh0; c.recpow = st.recpow; c.clumping = st.clumping;
        c.sqrtt = st.sqrtt; c.expt = st.expt; c.zp = st.zp; c.dzdt = st.dzdt;
    }
    double lsum = 0.0;
    double st_h0 = 0.0, st_h1 = 0.0, st_tr = 0.0, st_tc = 0.0;
    unsigned int nconv = 0, nfail = 0;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < ncell; id += (size_t)gridDim.x * 256) {
        const double h_old1 = fmax(c.eps, xh[id]);
        const double xav_in = xh_av[id];
        double hav1 = fmax(c.eps, xav_in);
        const bool allfrac = c.xh0 != nullptr;                      // (block-uniform)
        const double xav0_in = allfrac ? c.xh_av0[id] : 0.0;
        // evolve_point.F90:341-346 (ALLFRAC: both stored, both floored) / :347-353
        const double h_old0 = allfrac ? fmax(c.eps, c.xh0[id]) : 1.0 - h_old1;
        double hav0 = allfrac ? fmax(c.eps, xav0_in) : 1.0 - hav1;
        const double nd = (double)ndens[id];
        const double gamma = phih[id];
        double brech0 = c.clump ? (double)c.clump[id] * c.bh00 * c.recpow : c.brech0;   // evolve_point.F90:443-445
        double acolh0 = c.acolh0;
        // get_temperature_point (temperature_module.F90:133-151); temperature_end = temperature_start (:436)
        double t_start_cur = 0.0, t_start_avg = 0.0, t_end_avg = 0.0, t_end_int = 0.0, heat = 0.0;
        if (THERMAL) {
            t_start_cur = (double)c.temper[3 * id]; t_start_avg = (double)c.temper[3 * id + 1]; t_end_int = (double)c.temper[3 * id + 2];
            t_end_avg = t_start_avg;
            heat = c.phiheat[id];                                    // evolve_point.F90:364
        }
        double h1 = h_old1, h0 = h_old0;
        int nit = 0;
        for (;;) {
            nit++;
            const double yh0_av_old = hav0;
            const double de = nd * (hav1 + c.abu_c);                 // tped.f90:81
            if (THERMAL) {                                           // doric.f90:73-78 at temperature_end%average
                const double cl = c.clump ? (double)c.clump[id] : c.clumping;
                brech0 = cl * c.bh00 * pow(t_end_avg / 1e4, c.albpow);
                acolh0 = c.colh0 * sqrt(t_end_avg) * exp(-c.temph0 / t_end_avg);
            }
            const double aih0 = gamma + de * acolh0;
            const double delth = aih0 + de * brech0;
            const double eq1 = aih0 / delth;
            const double eq0 = de * brech0 / delth;
            const double deltht = delth * c.dt;
            const double ee = exp(-deltht);
            h1 = (h_old1 - eq1) * ee + eq1;
            h0 = (h_old0 - eq0) * ee + eq0;
            if (h0 < c.eps) { h0 = c.eps; h1 = 1.0 - c.eps; }
            const double avg = deltht < c.deltht_small ? 1.0 : (1.0 - ee) / deltht;
            hav1 = eq1 + (h_old1 - eq1) * avg;
            hav0 = 1.0 - hav1;
            if (hav0 < c.eps) hav0 = c.eps;
            if (THERMAL)                                             // :518-527 (de from the new average)
                thermal_dev(c, t_start_cur, t_end_int, t_end_avg, nd * (hav1 + c.abu_c), nd, h_old1, hav1, h1, heat);
            // :531-538: the temperature clause compares temperature_end%current with its copy from the iteration
            // before; thermal never writes %current, so it is |0/T| < 1e-3: true for every finite T > 0
            if (fabs((hav0 - yh0_av_old) / hav0) < c.min_frac_change || hav0 < c.min_frac_atoms) break;
            if (nit > c.max_iter) { nfail++; break; }
        }
        const double yh0_old = allfrac ? xav0_in : 1.0 - fmax(c.eps, xav_in);   // evolve_point.F90:375 (ALLFRAC: as stored) / :378-379
        bool notconv = fabs(hav0 - yh0_old) > c.min_frac_change && fabs((hav0 - yh0_old) / hav0) > c.min_frac_change &&
                       hav0 > c.min_frac_atoms;
        double t_stat = 0.0;
        if (THERMAL) {
            const float f_int = (float)t_end_int, f_avg = (float)t_end_avg;       // set_temperature_point, :553
            c.temper[3 * id + 2] = f_int; c.temper[3 * id + 1] = f_avg;
            t_stat = (double)f_avg;                                                // :381 get_temperature_point again
            notconv = notconv || (fabs((t_start_avg - t_stat) / t_stat) > c.tconv_rel && fabs(t_start_avg - t_stat) > c.tconv_abs);
        }
        if (notconv) nconv++;
        xh_intermed[id] = h1;
        xh_av[id] = hav1;
        if (allfrac) { c.xh_intermed0[id] = h0; c.xh_av0[id] = hav0; }   // :395-398
        lsum += h1;
        if (STATS) {                                           // k_photon_sums with xl = xh_intermed, xr = xh_av, same expressions
            st_h0 += nd * (allfrac ? h0 : 1.0 - h1);
            st_h1 += nd * h1;
            const double y1 = hav1, y0 = allfrac ? hav0 : 1.0 - y1;
            const double de = nd * (y1 + c.abu_c);
            const double cl = c.clump ? (double)c.clump[id] : c.clumping;
            if (THERMAL) {                                         // photonstatistics.F90:167-177 at temperature%average
                st_tr += nd * y1 * de * cl * c.bh00 * pow(t_stat / 1e4, c.albpow);
                st_tc += nd * y0 * de * c.colh0 * sqrt(t_stat) * exp(-c.temph0 / t_stat);
            } else {
            st_tr += nd * y1 * de * cl * c.bh00 * c.recpow;
            st_tc += nd * y0 * de * c.colh0 * c.sqrtt * c.expt;
            }
        }
    }
    const double tot = block_sum_256(lsum, sm);
    if (threadIdx.x == 0) sum_partial[blockIdx.x] = tot;
    if (STATS) {
        double v[4] = {st_h0, st_h1, st_tr, st_tc};
        for (int m = 0; m < 4; ++m) {
            __syncthreads();
            const double t4 = block_sum_256(v[m], sm);
            if (threadIdx.x == 0) c.stat_partial[(size_t)m * gridDim.x + blockIdx.x] = t4;
        }
    }
    // integer counts: order-independent
    for (int off = 32; off > 0; off >>= 1) { nconv += __shfl_down(nconv, off, 64); nfail += __shfl_down(nfail, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (nconv) atomicAdd(conv_flag, (unsigned long long)nconv);
        if (nfail) atomicAdd(chem_fail, nfail);
    }
}

// photonstatistics.F90:104-217: the four mesh sums of state_before/state_after/total_rates in one
// pass (h0, h1 from xh_l; recombination and collisional-ionization sums from xh_r).
// partial is [4][gridDim.x].
__global__ __launch_bounds__(256) void k_photon_sums(size_t ncell, const float *__restrict__ ndens,
                                                     const double *__restrict__ xl, const double *__restrict__ xr,
                                                     double abu_c, double clumping, const float *__restrict__ clump,
                                                     double bh00, double recpow, double colh0, double sqrtt, double expt,
                                                     double *partial, const float *__restrict__ temper, double albpow,
                                                     double temph0, const double *__restrict__ xl0, const double *__restrict__ xr0)
{   // temper != null: non-isothermal run, the rate coefficients at every cell's temperature%average (:167)
    // xl0 / xr0 != null: -DALLFRAC drivers, the stored neutral fractions of xl / xr (photonstatistics.F90:117-118, :158-159)
    __shared__ double sm[4];
    double h0 = 0.0, h1 = 0.0, tr = 0.0, tc = 0.0;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < ncell; id += (size_t)gridDim.x * 256) {
        const double nd = (double)ndens[id];
        const double x = xl[id];
        h0 += nd * (xl0 ? xl0[id] : 1.0 - x);
        h1 += nd * x;
        const double y1 = xr[id], y0 = xr0 ? xr0[id] : 1.0 - y1;
        const double de = nd * (y1 + abu_c);
        const double cl = clump ? (double)clump[id] : clumping;
        if (temper) {
            const double t = (double)temper[3 * id + 1];
            tr += nd * y1 * de * cl * bh00 * pow(t / 1e4, albpow);
            tc += nd * y0 * de * colh0 * sqrt(t) * exp(-temph0 / t);
            continue;
        }
        tr += nd * y1 * de * cl * bh00 * recpow;              // photonstatistics.F90:166-168, left to right
        tc += nd * y0 * de * colh0 * sqrtt * expt;            // :169-172
    }
    double v[4] = {h0, h1, tr, tc};
    for (int m = 0; m < 4; ++m) {
        const double tot = block_sum_256(v[m], sm);
        if (threadIdx.x == 0) partial[(size_t)m * gridDim.x + blockIdx.x] = tot;
        __syncthreads();
    }
}

// set_final_temperature_point (temperature_module.F90:172-183): %current = %intermed on convergence (evolve.F90:220)
__global__ __launch_bounds__(256) void k_final_temperature(size_t ncell, float *temper)
{
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < ncell; id += (size_t)gridDim.x * 256)
        temper[3 * id] = temper[3 * id + 2];
}

__global__ __launch_bounds__(256) void k_sum_partial(size_t n, const double *__restrict__ a, double *partial)
{
    __shared__ double sm[4];
    double v = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v += a[i];
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// out[m] = sum of partial[m][0..n) in a fixed order, m = blockIdx.x; `out` may be mapped host memory.
__global__ __launch_bounds__(256) void k_sum_final(int n, const double *partial, double *out, const int *gate = nullptr)
{
    __shared__ double sm[4];
    if (gate && *gate != 0) return;
    partial += (size_t)blockIdx.x * n;
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// End of a global pass: the sum of xh_intermed and the two counters go to the host's pinned scalars
// (mapped pointers); the counters are left at zero for the next pass.
__global__ __launch_bounds__(256) void k_pass_final(int n, const double *partial, unsigned long long *conv,
                                                    unsigned int *chem_fail, double *host_sum,
                                                    unsigned long long *host_conv, unsigned int *host_fail,
                                                    const int *gate = nullptr, unsigned long long *dev_seq = nullptr,
                                                    unsigned long long *host_seq = nullptr)
{
    // dev_seq / host_seq (fused iteration): a counter of completed passes, stored to the host LAST -- the host polls it
    // instead of blocking in a stream synchronize (whose wake-up is a tenth of a 0.26 ms iteration)
    __shared__ double sm[4];
    if (gate && *gate != 0) return;
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) {
        *host_sum = tot; *host_conv = *conv; *host_fail = *chem_fail;
        *conv = 0ULL; *chem_fail = 0u;
        if (host_seq) {
            const unsigned long long sq = *dev_seq + 1ULL;
            *dev_seq = sq;
            __threadfence_system();
            __hip_atomic_store(host_seq, sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace c2r
