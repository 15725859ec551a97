// The global pass (evolve0D_global over the mesh: do_chemistry + doric + thermal), the photon-statistics sums and the
// fixed-order reductions, on the host side.  Kernels: kernels_chem.hpp.
#include "ctx.hpp"
#include "kernels_chem.hpp"

namespace c2r {

// the four mesh sums of photonstatistics.F90 into dst[4] (device-visible: mapped pinned memory), no host wait
int photon_sums_launch(Ctx *ctx, int which_l, int which_r, double *dst)
{
    const c2r_params &p = ctx->prm;
    // photonstatistics.F90:166-172: same rate coefficients as doric, host libm
    hipLaunchKernelGGL(k_photon_sums, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell,
                       (const float *)ctx->grid[0], (const double *)ctx->grid[which_l],
                       (const double *)ctx->grid[which_r], p.abu_c, (double)ctx->clumping, (const float *)ctx->d_clump,
                       p.bh00, pow(ctx->temper / 1e4, p.albpow), p.colh0, sqrt(ctx->temper),
                       exp(-p.temph0 / ctx->temper), ctx->d_sum_partial, ctx->thermal ? (const float *)ctx->grid[6] : nullptr,
                       p.albpow, p.temph0, ctx->allfrac ? (const double *)ctx->grid[which_l + 6] : nullptr,
                       ctx->allfrac ? (const double *)ctx->grid[which_r + 6] : nullptr);
    hipLaunchKernelGGL(k_sum_final, dim3(4), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_sum_partial, dst);
    HIP_TRY(hipGetLastError());
    return C2R_OK;
}

// the launches of a global pass, no host wait; gate: see k_transpose_xy
int global_pass_enqueue(Ctx *ctx, double dt, double *stats_dst, size_t cell_off, size_t cell_cnt, const int *gate, bool count_pass)
{
    const c2r_params &p = ctx->prm;
    ChemParams cp{};
    // (dt and the step's rate coefficients reach the kernel through the step block -- callers have run sync_step with this
    // dt --; the by-value copies below only document what the kernel overwrites at entry)
    cp.step = reinterpret_cast<const StepBlock *>(ctx->d_step);
    cp.dt = 0.0 * dt; cp.eps = p.epsilon; cp.min_frac_change = p.minimum_fractional_change;
    cp.min_frac_atoms = p.minimum_fraction_of_atoms; cp.abu_c = p.abu_c; cp.deltht_small = C2R_DELTHT_SMALL;
    cp.max_iter = p.max_chem_iter;
    // doric.f90:73,78 -- temperature is uniform (isothermal), so both rate coefficients are
    // per-call constants; evaluated with the host libm like the reference does at run time
    cp.bh00 = p.bh00; cp.clump = ctx->d_clump ? ctx->d_clump + cell_off : nullptr;
    cp.colh0 = p.colh0;
    cp.stat_partial = ctx->d_stat_partial;
    if (ctx->allfrac) {      // the (:,:,:,0) halves (evolve_point.F90:341-346, :394-399)
        cp.xh0 = (const double *)ctx->grid[7] + cell_off; cp.xh_av0 = (double *)ctx->grid[8] + cell_off; cp.xh_intermed0 = (double *)ctx->grid[9] + cell_off;
    }
    if (ctx->thermal) {
        const c2r_thermal_params &t = ctx->tprm;
        if (t.cosmological && !ctx->have_zred) FAIL(C2R_ESTATE, "non-isothermal run: c2r_set_redshift has not been called (cosmo_cool needs zred)");
        cp.temper = (float *)ctx->grid[6] + 3 * cell_off; cp.phiheat = (const double *)ctx->grid[5] + cell_off; cp.cool = ctx->d_cool;
        cp.cool_mintemp = t.cool_mintemp; cp.cool_dtemp = t.cool_dtemp; cp.cool_points = t.cool_points;
        cp.thermal_max_steps = t.thermal_max_steps;
        cp.k_B = t.k_B; cp.gamma1 = t.gamma1; cp.minitemp = t.minitemp; cp.rel_denergy = t.relative_denergy;
        cp.rate_floor = t.thermal_rate_floor; cp.time_tol = t.thermal_time_tol;
        cp.temph0 = p.temph0; cp.albpow = p.albpow;
        cp.tconv_rel = t.temp_conv_rel; cp.tconv_abs = t.temp_conv_abs;
    }
    prof_begin(ctx, ctx->ev_chem, ctx->ev_chem_used);
#define C2R_LAUNCH_GLOBAL(S, T) hipLaunchKernelGGL((k_global_pass<S, T>), dim3(kSumBlocks), dim3(256), 0, ctx->stream, cp, cell_cnt, \
                           (const float *)ctx->grid[0] + cell_off, (const double *)ctx->grid[1] + cell_off, (double *)ctx->grid[2] + cell_off, \
                           (double *)ctx->grid[3] + cell_off, (const double *)ctx->grid[4] + cell_off, ctx->d_sum_partial, ctx->d_conv, \
                           ctx->d_chemfail, gate)
    if (stats_dst) { if (ctx->thermal) C2R_LAUNCH_GLOBAL(true, true); else C2R_LAUNCH_GLOBAL(true, false); }
    else { if (ctx->thermal) C2R_LAUNCH_GLOBAL(false, true); else C2R_LAUNCH_GLOBAL(false, false); }
#undef C2R_LAUNCH_GLOBAL
    prof_end(ctx, ctx->ev_chem, ctx->ev_chem_used);
    if (stats_dst)
        hipLaunchKernelGGL(k_sum_final, dim3(4), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_stat_partial, stats_dst, gate);
    // last: its final store is the pass counter a fused iteration's host polls (count_pass)
    hipLaunchKernelGGL(k_pass_final, dim3(1), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_sum_partial, ctx->d_conv,
                       ctx->d_chemfail, &ctx->d_hsc->sum, &ctx->d_hsc->conv, &ctx->d_hsc->chemfail, gate,
                       count_pass ? ctx->d_seq : nullptr, count_pass ? &ctx->d_hsc->seq : nullptr);
    HIP_TRY(hipGetLastError());
    return C2R_OK;
}

// global_pass (evolve.F90:499-573); stats_dst (device-visible, 4 doubles, or null): the photon-statistics sums of
// (xh_intermed, xh_av) as the pass leaves them, from the same kernel
int global_pass_impl(Ctx *ctx, double dt, int64_t *conv_flag, double *sum_xh1, double *stats_dst, size_t cell_off, size_t cell_cnt)
{
    if (cell_cnt == (size_t)-1) cell_cnt = ctx->ncell;        // (a slab [cell_off, cell_off+cell_cnt): slab chemistry)
    ctx->step_dt = dt;
    int rc = sync_step(ctx);
    if (rc) return rc;
    rc = global_pass_enqueue(ctx, dt, stats_dst, cell_off, cell_cnt, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    if (conv_flag) *conv_flag = (int64_t)ctx->h_sc->conv;
    if (sum_xh1) *sum_xh1 = ctx->h_sc->sum;
    return C2R_OK;
}


int final_temperature_enqueue(Ctx *ctx)
{
    hipLaunchKernelGGL(k_final_temperature, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell, (float *)ctx->grid[6]);
    HIP_TRY(hipGetLastError());
    return C2R_OK;
}

}  // namespace c2r

using namespace c2r;

extern "C" {

int c2r_set_final_temperature(c2r_ctx *c)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (!ctx->thermal) return C2R_OK;                     // temperature_module.F90:181: nothing to do when isothermal
    HIP_TRY(hipSetDevice(ctx->prm.device));
    hipLaunchKernelGGL(k_final_temperature, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell, (float *)ctx->grid[6]);
    HIP_TRY(hipGetLastError());
    return C2R_OK;
}

// evolve0D_global(dt,pos,conv_flag) (evolve_point.F90:305-406) for ONE cell (pos 1-based) on the caller's arrays: the
// same kernel as the mesh-wide pass on a one-cell slab.  Non-isothermal contexts: phiheat_grid / temperature_grid (3 x f32 per
// cell) of the caller as well.  conv_flag is incremented when the cell has not converged.  Slow by construction.
int c2r_global_pass_cell_host(c2r_ctx *c, double dt, const int32_t pos[3], const float *ndens, const double *xh, double *xh_av,
                              double *xh_intermed, const double *phih_grid, const double *phiheat_grid, float *temperature_grid,
                              int32_t *conv_flag)
{
    if (!c || !pos || !ndens || !xh || !xh_av || !xh_intermed || !phih_grid) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    const c2r_params &p = ctx->prm;
    for (int d = 0; d < 3; ++d) if (pos[d] < 1 || pos[d] > p.mesh[d]) FAIL(C2R_EINVAL, "evolve0D_global: mesh position out of range");
    if (ctx->thermal && (!phiheat_grid || !temperature_grid)) FAIL(C2R_EINVAL, "non-isothermal run: evolve0D_global needs phiheat_grid and temperature_grid");
    const size_t idx = (size_t)(pos[0] - 1) + (size_t)p.mesh[0] * ((size_t)(pos[1] - 1) + (size_t)p.mesh[1] * (size_t)(pos[2] - 1));
    hipStream_t st = ctx->stream;
    // (-DALLFRAC drivers: xh / xh_av / xh_intermed are (mesh,0:1) -- the neutral half first, the ionized half ncell further on)
    const size_t ion = ctx->allfrac ? ctx->ncell : 0;
    HIP_TRY(hipMemcpyAsync((float *)ctx->grid[0] + idx, ndens + idx, sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync((double *)ctx->grid[1] + idx, xh + ion + idx, sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync((double *)ctx->grid[2] + idx, xh_av + ion + idx, sizeof(double), hipMemcpyHostToDevice, st));
    if (ctx->allfrac) {
        HIP_TRY(hipMemcpyAsync((double *)ctx->grid[7] + idx, xh + idx, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((double *)ctx->grid[8] + idx, xh_av + idx, sizeof(double), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync((double *)ctx->grid[4] + idx, phih_grid + idx, sizeof(double), hipMemcpyHostToDevice, st));
    if (ctx->thermal) {
        HIP_TRY(hipMemcpyAsync((double *)ctx->grid[5] + idx, phiheat_grid + idx, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((float *)ctx->grid[6] + 3 * idx, temperature_grid + 3 * idx, 3 * sizeof(float), hipMemcpyHostToDevice, st));
    }
    int64_t nonconv = 0;
    if ((rc = global_pass_impl(ctx, dt, &nonconv, nullptr, nullptr, idx, 1))) return rc;
    HIP_TRY(hipMemcpyAsync(xh_av + ion + idx, (double *)ctx->grid[2] + idx, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(xh_intermed + ion + idx, (double *)ctx->grid[3] + idx, sizeof(double), hipMemcpyDeviceToHost, st));
    if (ctx->allfrac) {
        HIP_TRY(hipMemcpyAsync(xh_av + idx, (double *)ctx->grid[8] + idx, sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(xh_intermed + idx, (double *)ctx->grid[9] + idx, sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (ctx->thermal)
        HIP_TRY(hipMemcpyAsync(temperature_grid + 3 * idx, (float *)ctx->grid[6] + 3 * idx, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (conv_flag) *conv_flag += (int32_t)nonconv;
    return C2R_OK;
}

int c2r_global_pass_host(c2r_ctx *c, double dt, const float *ndens, const double *xh, double *xh_av,
                         double *xh_intermed, const double *phih_grid, int64_t *conv_flag)
{
    if (!c || !ndens || !xh || !xh_av || !xh_intermed || !phih_grid) return C2R_EINVAL;
    int rc;
    Ctx *ctx = C(c);
    if ((rc = c2r_upload(c, 0, ndens))) return rc;
    if ((rc = copy_in(ctx, 1, xh))) return rc;                 // (-DALLFRAC drivers: both halves of the (mesh,0:1) arrays)
    if ((rc = copy_in(ctx, 2, xh_av))) return rc;
    if ((rc = c2r_upload(c, 4, phih_grid))) return rc;
    if ((rc = c2r_global_pass(c, dt, conv_flag, nullptr))) return rc;
    if ((rc = copy_out(ctx, 2, xh_av))) return rc;
    if ((rc = copy_out(ctx, 3, xh_intermed))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return C2R_OK;
}

int c2r_sum(c2r_ctx *c, int32_t which, double *sum)
{
    if (!c || !sum || which < 1 || which > 9 || which == 5 || which == 6) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (which >= 7 && !ctx->allfrac) FAIL(C2R_ESTATE, "arrays 7 - 9 (the stored neutral fractions) exist with c2r_params.allfrac only");
    HIP_TRY(hipSetDevice(ctx->prm.device));
    hipLaunchKernelGGL(k_sum_partial, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell,
                       (const double *)ctx->grid[which], ctx->d_sum_partial);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_sum_partial, &ctx->d_hsc->sum);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *sum = ctx->h_sc->sum;
    return C2R_OK;
}

int c2r_photon_sums(c2r_ctx *c, int32_t which_l, int32_t which_r, double out[4])
{
    if (!c || !out || which_l < 1 || which_l > 3 || which_r < 1 || which_r > 3) return C2R_EINVAL;     // (-DALLFRAC drivers: the arrays' stored neutral halves are used too)
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    if ((rc = photon_sums_launch(ctx, which_l, which_r, ctx->d_hsc->four))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int m = 0; m < 4; ++m) out[m] = ctx->h_sc->four[m];
    return C2R_OK;
}

int c2r_global_pass(c2r_ctx *c, double dt, int64_t *conv_flag, double *sum_xh1)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    return global_pass_impl(ctx, dt, conv_flag, sum_xh1, nullptr, 0, (size_t)-1);
}

}  // extern "C"
