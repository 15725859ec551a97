/* c2ray_hip.h -- C ABI of the MI355X-native C2-Ray evolve hot path.
 *
 * This is the drop-in boundary.  The reference (garrelt/C2-Ray3Dm) exposes the path as Fortran
 * module procedures over module-global arrays; a Fortran shim (c2-ray3dm_amd/fortran/
 * evolve_hip.F90) keeps those names and forwards to the entry points below through
 * ISO_C_BINDING.  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *  - plain C types only; all grids are column-major (Fortran order) N1 x N2 x N3, 1-based mesh
 *    positions, exactly the storage of the reference's allocatable module arrays;
 *  - "host" entry points take host pointers owned by the caller (the Fortran driver) and never
 *    retain them past the call; "dev" entry points work on device buffers that are either owned
 *    by the context or bound from outside (c2r_bind_device_buffers: e.g. torch tensors);
 *  - every function returns 0 on success, a negative C2R_E* code or a positive hipError_t on
 *    failure (the reference has no status returns: it warns and continues, evolve.F90:230,
 *    evolve_point.F90:541; non-convergence is therefore reported in c2r_report, not as an error);
 *  - one context per process and GPU; calls on one context are serialised by the caller, as
 *    evolve3D is in the reference (called from the main thread of each MPI rank).
 */
#ifndef C2RAY_HIP_H
#define C2RAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly what this header declares is exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define C2R_OK            0
#define C2R_EINVAL      (-1)   /* bad argument */
#define C2R_ESTATE      (-2)   /* call order: tables / step scalars / sources / buffers missing */
#define C2R_ENOMEM      (-3)
#define C2R_ECALLBACK   (-4)   /* the all-reduce callback or the iteration hook failed */

#define C2R_MAX_ITER_LOG 128

#define C2R_DEVICE_AUTO  (-1)
#define C2R_SWEEP_EXACT   0
#define C2R_SWEEP_FAST    1

typedef struct c2r_ctx c2r_ctx;

/* Numerical parameters of the path: compile-time `parameter`s in the reference, run-time here
 * so nothing is baked in.  c2r_default_params() fills the shipped values (include/
 * c2ray_constants.h cites each defining line). */
typedef struct c2r_params {
    int32_t mesh[3];                 /* sizes.f90:33  mesh(1:3) */
    int32_t device;                  /* HIP device ordinal, or C2R_DEVICE_AUTO: the process's local rank as the launcher
                                      * exports it (C2R_DEVICE, LOCAL_RANK, OMPI_COMM_WORLD_LOCAL_RANK, MV2_COMM_WORLD_LOCAL_RANK,
                                      * MPI_LOCALRANKID, PMI_LOCAL_RANK, SLURM_LOCALID; first one set) modulo the visible devices */
    int32_t subboxsize;              /* c2ray_parameters.f90:54 */
    int32_t max_subbox;              /* c2ray_parameters.f90:61 */
    int32_t numtau;                  /* radiation_sizes.f90:14 (tables hold numtau+1 entries) */
    int32_t max_outer_iter;          /* evolve.F90:228 (100) */
    int32_t max_chem_iter;           /* evolve_point.F90:541 (400) */
    int32_t deterministic_rates;     /* 0: Gamma accumulated with f64 atomics (reproducible to rounding);
                                      * 1: per-source Gamma grids reduced in source order: bit-reproducible, and
                                      *    the summation order of the serial reference (evolve_point.F90:283);
                                      *    costs 16 B x N^3 of scratch per source in flight */
    int32_t sweep_mode;              /* C2R_SWEEP_EXACT (0; opt-in since round 6): every f64 operation of evolve0D's geometry and of cinterp
                                      *    (column_density.f90:29-271) in the reference's order, IEEE-exact division and
                                      *    sqrt: column densities bit-identical to the Fortran.  The RATE of a cell
                                      *    (photoion_rates, radiation_photoionrates.F90:71-317) is evaluated by the routine both
                                      *    modes share (kernels_common.hpp rates_fast: table position straight out of the logarithm,
                                      *    one 2^-48 reciprocal): inside |dGamma| <= 1e-13 Gamma + 2e-14 W of the oracle, the
                                      *    bound this mode has always stated (the rounding of the table position dominates it
                                      *    in the reference's own libm too).  The photon loss through a sub-box surface is taken
                                      *    from the same routine, so the keep / retire decision of evolve_source.F90:128-131
                                      *    compares a loss good to ~1e-13 with loss_fraction x flux: the integer results (sub-box
                                      *    counts, visited cells) are tolerance-bound in BOTH modes -- equal to the reference's on
                                      *    every fixture and in every fuzz run so far, not equal by construction;
                                      * C2R_SWEEP_FAST (1; what c2r_default_params sets, what the Fortran shim and the Python host run
                                      *    unless told otherwise, and what bench.py's headline times): the interpolation and the geometry re-associated as well (factored
                                      *    weights, 2^-48 reciprocals): same integer results, column densities within 1e-11 and
                                      *    rates within |dGamma| <= 1e-12 Gamma + 2e-14 W of the oracle (W = sum_s (1+tau_in)
                                      *    photo_in / (vol_ph n_HI): the rate that passes THROUGH the cell; worst plain relative
                                      *    error measured 7.6e-8, in cells whose own rate is ~1e-7 of W): the tolerances the GPU
                                      *    tests apply live in tests/_util.py (TOL); ~1.2x the throughput.  Only this field
                                      *    selects the mode: the library reads no environment variable for it (the Fortran shim
                                      *    and the Python host have their own switches) */
    int32_t allfrac;                 /* 1: the driver was built with -DALLFRAC (ionfractions_module.F90:19-50; no shipped makefile defines it):
                                      *    xh, xh_av, xh_intermed are (mesh,0:1) and the NEUTRAL fraction is stored, not derived as
                                      *    1 - x.  evolve0D then takes n_HI from the stored neutral fraction (evolve_point.F90:131-134),
                                      *    evolve0D_global reads and writes both (:341-346, :394-399), Test 2 sums the stored one
                                      *    (evolve.F90:179-181), the photon statistics count it.  The device holds the (:,:,:,0) halves
                                      *    as arrays 7 (xh), 8 (xh_av), 9 (xh_intermed); arrays 1 - 3 stay the (:,:,:,1) halves; every
                                      *    HOST pointer named xh / xh_av / xh_intermed below is then the driver's whole (mesh,0:1)
                                      *    array (neutral half first).  0 (default): the shipped build */
    double  epsilon;                 /* c2ray_parameters.f90:31 */
    double  convergence_fraction;    /* :25 */
    double  minimum_fractional_change; /* :34 */
    double  minimum_fraction_of_atoms; /* :40 */
    double  loss_fraction;           /* :67 */
    double  max_coldensh;            /* evolve_point.F90:95 */
    double  tau_photo_limit;         /* radiation_photoionrates.F90:244 */
    double  sigma_HI;                /* cgsphotoconstants.f90:24 */
    double  minlogtau, dlogtau;      /* radiation_tables.F90:45-47 */
    double  weight_floor;            /* column_density.f90:289 */
    double  sqrt2, sqrt3, pi;        /* column_density.f90:52-53, mathconstants.f90:21 */
    double  abu_c;                   /* abundances.f90:26 */
    double  bh00, albpow, colh0, temph0; /* cgsconstants.f90:64-86 */
    double  S_star;                  /* radiation_sed_parameters: table normalisation */
    size_t  scratch_bytes;           /* cap on per-source sweep scratch (0 = 1/4 of free HBM) */
} c2r_params;

/* What evolve3D reports through its log (evolve.F90:205-210,249-251,559-566), per call. */
typedef struct c2r_report {
    int32_t niter;                   /* outer iterations executed */
    int32_t converged;               /* 1: xh updated (evolve.F90:218); 0: gave up, xh unchanged (:228) */
    int64_t conv_flag;               /* non-converged cells of the last global pass */
    int64_t conv_criterion;          /* evolve.F90:162 */
    int64_t sum_nbox_all;            /* evolve_source.F90:46 */
    int64_t visited;                 /* (cell,source) pairs executed over all iterations (this rank) */
    double  photon_loss_all;         /* evolve_data.F90 photon_loss_all(1) */
    double  seconds_sweep;           /* wall time inside pass_all_sources, all iterations (one rank: the whole iteration, see c2r_iterate) */
    double  seconds_chem;            /* wall time inside global_pass, all iterations (one rank: 0, it is not waited for separately) */
    int32_t chem_not_converged;      /* cells that hit max_chem_iter in the last global pass */
    int32_t timing_split;            /* 1: seconds_sweep and seconds_chem are separate waits (several ranks); 0: one rank -- the whole
                                      * iteration is one wait and is booked under seconds_sweep, seconds_chem is 0.  With slab
                                      * chemistry the pass's collectives are part of seconds_chem */
    int64_t it_conv_flag[C2R_MAX_ITER_LOG];        /* [k]: global pass of iteration k+1; after a restart from
                                                    * iteration r, [r-1] is the pass that follows the dump read */
    int64_t it_sum_nbox[C2R_MAX_ITER_LOG];
    double  it_rel_change_xh1[C2R_MAX_ITER_LOG];   /* Test-2 values seen before iteration k+2 */
    double  it_rel_change_xh0[C2R_MAX_ITER_LOG];
    double  it_sum_xh1[C2R_MAX_ITER_LOG];          /* sum(xh_intermed) after global pass k+1 */
    /* photon statistics of the step, as photonstatistics.F90 leaves them after evolve3D
     * (state_before :104, state_after :190, total_rates :137, total_ionizations :222,
     * report_photonstatistics :254) */
    double  h0_before, h1_before, h0_after, h1_after;
    double  totrec, totcollisions, dh0, total_ion, totalsrc, photcons;
    double  it_photcons[C2R_MAX_ITER_LOG];         /* conservation ratio logged after each global pass */
    /* where the wall time of a host-pointer call (c2r_evolve3d, _restart, _thermal) went: the copies are timed with HIP
     * events on the context's stream (they are enqueued without a host wait in between), the call as a whole with the host
     * clock.  The device-resident entries (c2r_evolve3d_dev) leave the two copy times at zero. */
    double  seconds_upload;          /* host -> device: ndens, xh (restart: xh_av, xh_intermed, phih_grid too) */
    double  seconds_download;        /* device -> host: xh, phih_grid (+ whatever else the caller passed non-null) */
    double  seconds_total;           /* the whole call */
} c2r_report;

/* Sum `count` doubles at device pointer `buf` over all ranks, in place, on `stream`
 * (replaces MPI_ALLREDUCE of phih_grid, evolve.F90:599-602).  Return 0 on success. */
typedef int (*c2r_allreduce_fn)(void *user, void *dev_buf, size_t count, void *hip_stream);

/* ---- life cycle ------------------------------------------------------------------------ */
int  c2r_default_params(c2r_params *p);
/* evolve_ini (evolve_data.F90:73): allocates the device mirrors of the hot-path arrays. */
int  c2r_create(c2r_ctx **ctx, const c2r_params *p);
void c2r_destroy(c2r_ctx *ctx);
const char *c2r_last_error(const c2r_ctx *ctx);
/* One line describing how the context runs: the device and how C2R_DEVICE_AUTO resolved it (which launcher variable),
 * the sweep mode, the rate accumulation, rank/nranks, and a WARNING when C2R_DEVICE_AUTO found no local-rank variable
 * although nranks > 1 (also printed to stderr by c2r_set_rank).  Valid until the next call on the context. */
const char *c2r_info(c2r_ctx *ctx);
/* Run on a caller-provided hipStream_t (NULL = the context's own stream). */
int  c2r_set_stream(c2r_ctx *ctx, void *hip_stream);
/* Switches of the launch schedule, for A/B measurements and tests.  The library reads NO environment variable for any of them
 * (nor for anything else but the device of C2R_DEVICE_AUTO): a host that wants one calls this.  None changes a result beyond the
 * order in which the f64 atomics of different sources land in the rate arrays; sub-box counts, visited cells and the photon loss are
 * the same bits under every setting (tests/test_gpu_chains.py, test_gpu_few_sources.py, test_gpu_xcd_order.py).  Setting one drops
 * every captured launch sequence.  C2R_EINVAL for an unknown name.
 *   name                  default   meaning
 *   graph                 1         0: no launch sequence is ever captured and replayed (hipGraph)
 *   chain_graph           1         0: chains in flight (64 - 768 sources per round) are driven launch by launch in every pass;
 *                                      1: from the second pass on a chain's launch sequence is one replayed hipGraph (sweep.hip run_chains)
 *   chain_tail            1         0: the tail of an iteration (totals, fold of the transposed rates, global pass) is never enqueued behind
 *                                      the replayed chains' device-side gate: the host waits for the chains first
 *   fused_iter            1         0: c2r_iterate always runs its three steps in turn (no whole-iteration graph for <= 32 sources)
 *   fuse_small            1         0: the first sub-boxes run shell by shell instead of in k_sweep_box_fused
 *   fold_source_cell      1         0: k_source_cells is always its own launch
 *   pair_shells           1         0: never two shells per launch (look-ahead pairs of <= 32 sources)
 *   sched_hint            1         0: the host always runs exactly one sub-box ahead of the device
 *   spin_wait, poll_wait  1         0: host waits block (hipStreamSynchronize / hipEventSynchronize) instead of polling first
 *   stream_hint           -1        non-temporal cache policy of the shell kernels: -1 by mesh size (n_HI >= 64 MB), 0 off, 1 on
 *   xcd_order             -1        plane-ordered block mapping of the far shells: -1 from xcd_min_per_plane sources per mesh plane, 0 never, 1 always
 *   xcd_min_per_plane     1.5       ... that threshold;   xcd_min_alive 0.9: ... while this share of the batch is still traced;   xcd_qmin 16: ... from this shell
 *   chains                0         n > 0: a pass runs as n chains in flight (1 - 4) whatever its source count; 0: the library's rule
 *   batch_cap             0         n > 0: at most n sources in flight per round (tests of passes cut into several rounds)
 *   sparse_exchange       1         0: c2r_allreduce_rates always reduces the whole grid;   sparse_fraction 0.5: largest packed volume (units of the mesh)
 *   exchange_overlap      0         = c2r_set_exchange_overlap;   exchange_overlap_min 64: fewest sources per rank it applies to */
int  c2r_set_option(c2r_ctx *ctx, const char *name, double value);

/* ---- inputs the driver owns ------------------------------------------------------------ */
/* stellar_photo_thick_table / _thin_table (0:NumTau,1), built once by rad_ini
 * (radiation_tables.F90:95-126). n = numtau+1. */
int  c2r_set_tables(c2r_ctx *ctx, const double *thick, const double *thin, int32_t n);
/* The second source type of photoion_rates -- builds of the driver with use_xray_SED=.true. (sed_parameters.f90:56):
 * `phi = phi + photo_lookuptable(..., NormFlux_xray(nsrc), "P", vol)` (radiation_photoionrates.F90:133-137): the same table
 * positions, its own thick / thin tables (xray_photo_thick_table / _thin_table(0:NumTau,1), radiation_tables.F90:80-81) and a
 * second normalised flux per source (NormFlux_xray: column 5 of the source list / S_star_xray, sourceprops.F90:63, :381, :631).
 * A source is traced while its STELLAR flux leaves its sub-box (evolve_source.F90:119: total_source_flux counts that alone);
 * where NormFlux_xray > 0 the X-ray rate is added cell by cell and its photons count in photo_out.
 * c2r_set_xray_tables(ctx, thick, thin, numtau+1): switch it on (NULL, NULL: off).
 * c2r_set_xray_heat_tables(ctx, heat_thick, heat_thin, numtau+1): non-isothermal contexts (c2r_set_thermal) need the type's
 *   heating tables too (xray_heat_thick/thin_table, radiation_tables.F90:84-85; heat_lookuptable "P", :165-171) before a pass.
 * c2r_set_xray_sources(ctx, normflux_xray, nsrc): after every c2r_set_sources with a new list; sources without a value: 0.
 * (The reference fills these tables from an array it never sets, radiation_tables.F90:367: they are inputs here -- e.g.
 * c2r_build_tables with C2R_SED_POWER_LAW.) */
int  c2r_set_xray_tables(c2r_ctx *ctx, const double *thick, const double *thin, int32_t n);
int  c2r_set_xray_heat_tables(c2r_ctx *ctx, const double *heat_thick, const double *heat_thin, int32_t n);
int  c2r_set_xray_sources(c2r_ctx *ctx, const double *normflux_xray, int32_t nsrc);

/* Per-time-step scalars the driver recomputes before every evolve3D call (C2Ray.F90:367-376):
 * dr(1:3), vol (grid.F90, cosmology.F90:161-193), coldensh_LLS (LLS.F90:178-182),
 * clumping (clumping_module.F90:74), temper_val (temperature_module.F90:34).  They (and dt, the redshift) reach the kernels
 * through a small device-resident block refreshed when a value changes, never through captured kernel arguments: a new time
 * step costs one small copy, not a re-capture of a replayed launch sequence. */
int  c2r_set_step(c2r_ctx *ctx, const double dr[3], double vol, double coldensh_LLS,
                  float clumping, double temper);
/* Non-default LLS treatment (c2ray_parameters.f90:80-99 type_of_LLS, evolve_point.F90:186-196):
 * 1 homogeneous column per cell (coldensh_LLS of c2r_set_step; the default), 2 position-dependent
 * column LLS_grid(i,j,k) (LLS.F90:199-212; host f32 N^3, copied), 3 hard barrier: no transfer beyond
 * R_max_LLS (proper cm, LLS.F90:191). */
int  c2r_set_lls(c2r_ctx *ctx, int32_t type_of_LLS, const float *lls_grid, double R_max_LLS);
/* Position-dependent clumping factor clumping_grid(i,j,k) (type_of_clumping 3-5; clumping_point,
 * clumping_module.F90:106-118) used by doric and the photon statistics; NULL returns to the scalar
 * clumping of c2r_set_step. */
int  c2r_set_clumping_grid(c2r_ctx *ctx, const float *clump_grid);
/* ---- non-isothermal runs (c2ray_parameters.f90:28 isothermal=.false.; the shipped value is .true.) ----------
 * Parameters of the heating/cooling path: `parameter`s of c2ray_parameters.f90:105-110, thermal.f90, tped.f90,
 * atomic.f90:23-25, cosmoparms.f90:28-41, radiation_photoionrates.F90:333, evolve_point.F90:387-388.
 * c2r_default_thermal fills the shipped values; cool_mintemp / cool_dtemp come from the cooling table file
 * (setup_cool, cooling.f90:64-87: temp(1) and temp(2)-temp(1) of tables/corocool.tab). */
typedef struct c2r_thermal_params {
    double tau_heat_limit;           /* radiation_photoionrates.F90:333 */
    double k_B, gamma1;              /* cgsconstants.f90:34, atomic.f90:25 */
    double minitemp, relative_denergy;        /* c2ray_parameters.f90:108,110 */
    double thermal_rate_floor, thermal_time_tol;   /* thermal.f90:117 (1d-50), :160 (1e-6) */
    double temp_conv_rel, temp_conv_abs;      /* evolve_point.F90:387-388 (0.1, 100 K) */
    double H0, Omega0;               /* cosmoparms.f90:30,41 (cosmo_cool, cosmology.F90:198-225) */
    double cool_mintemp, cool_dtemp; /* cooling.f90:78-79: log10 T of the first table row, row spacing */
    int32_t cool_points;             /* cooling.f90:26 temppoints */
    int32_t thermal_max_steps;       /* thermal.f90:163 */
    int32_t cosmological;            /* c2ray_parameters.f90:105 */
    int32_t reserved0;
} c2r_thermal_params;
int  c2r_default_thermal(c2r_thermal_params *t);
/* Switches the context to the non-isothermal path (t = NULL: back to isothermal): heat_thick / heat_thin =
 * stellar_heat_thick_table / _thin_table(0:NumTau,1) (radiation_tables.F90:521-543; n = numtau+1), cie_cool = the
 * cooling curve as setup_cool holds it (cooling.f90:83: 10**table, cool_points values).  From then on the sweep also
 * accumulates phi%heat into phiheat_grid (array 5; evolve_point.F90:285-286, heat_lookuptable
 * radiation_photoionrates.F90:323-417), the all-reduce covers it (evolve.F90:604-609), and the global pass runs
 * doric at every cell's own temperature followed by thermal (thermal.f90:22; evolve_point.F90:515-527) on
 * temperature_grid (array 6: temperature_module.F90:35, (current, average, intermed) f32 per cell), with the
 * temperature clause of the convergence test (:387-388) and set_final_temperature_point on convergence (evolve.F90:220).
 * With deterministic_rates the heating rates are reduced in source order like Gamma (twice the per-source scratch). */
int  c2r_set_thermal(c2r_ctx *ctx, const c2r_thermal_params *t, const double *heat_thick, const double *heat_thin,
                     int32_t n, const double *cie_cool);
/* cosmology.F90:42 zred at the time the driver calls evolve3D (redshift_evol of the middle of the step,
 * C2Ray.F90:368): cosmo_cool's redshift.  Per time step, non-isothermal runs only. */
int  c2r_set_redshift(c2r_ctx *ctx, double zred);
/* set_final_temperature_point (temperature_module.F90:172-183) for hosts that drive the loop piecewise. */
int  c2r_set_final_temperature(c2r_ctx *ctx);

/* srcpos(3,NumSrc) (1-based, may lie outside the mesh: wrapped at use) and
 * NormFlux_stellar(1:NumSrc) (sourceprops.F90:121-123,167-168).  Also allocates the sweep scratch of this rank's share (shell
 * planes, the pinned staging block: set-up, as evolve_ini's allocations are) if it is not large enough yet. */
int  c2r_set_sources(c2r_ctx *ctx, const int32_t *srcpos, const double *normflux, int32_t nsrc);
/* MPI rank / size of the static source distribution (master_slave.F90:85:
 * do ns1=1+rank,NumSrc,npr) and the collective that replaces MPI_ALLREDUCE. */
int  c2r_set_rank(c2r_ctx *ctx, int32_t rank, int32_t nranks, c2r_allreduce_fn fn, void *user);

/* Slab chemistry (an alternative to evolve.F90:548-555 + :599, where every rank all-reduces the rates and then
 * runs evolve0D_global over the WHOLE mesh): the rates are reduce-scattered by z-slabs (rank r ends up with the sum over
 * ranks of ITS slab of phih_grid / phiheat_grid), every rank runs the chemistry on its slab only, and the pass's outputs --
 * xh_av, xh_intermed, temperature_grid -- are all-gathered; the non-converged counts are summed through the all-reduce
 * callback, the sums that drive the convergence tests and the photon statistics are taken over the gathered arrays exactly
 * as the replicated pass takes them (the same decisions, bit for bit, on every rank), and when the step ends the rates are
 * gathered once so that phih_grid is complete everywhere.  Both callbacks work IN PLACE on a full-size device array, the
 * slabs given as nranks offsets and counts (c2r_slab): reduce-scatter in f64 ELEMENTS (after it only the own slab is
 * defined), all-gather in BYTES (every rank's slab is valid on entry in its own array).  Return 0 on success.
 * Worth it where the global pass is heavy (non-isothermal runs: 0.7 ms replicated at 256^3) or the ranks many; it moves
 * 3 array-slabs per iteration where the all-reduce moves 2.  NULL, NULL switches it off. */
typedef int (*c2r_reduce_scatter_fn)(void *user, void *dev_buf, const size_t *elem_offset, const size_t *elem_count,
                                     int32_t nranks, void *hip_stream);
typedef int (*c2r_allgather_fn)(void *user, void *dev_buf, const size_t *byte_offset, const size_t *byte_count,
                                int32_t nranks, void *hip_stream);
int  c2r_set_slab_chemistry(c2r_ctx *ctx, c2r_reduce_scatter_fn rs, c2r_allgather_fn ag, void *user);
/* The z-slab of `rank` among `nranks`: whole z-planes, the first (mesh(3) mod nranks) ranks one more; in cells. */
int  c2r_slab(const c2r_ctx *ctx, int32_t rank, int32_t nranks, size_t *cell_offset, size_t *cell_count);

/* Cost-balanced alternative to the static distribution (the reference's answer to
 * imbalance is its master/worker scheduler, master_slave.F90:124-330): the 0-based global indices
 * of the sources THIS rank sweeps, in sweep order.  idx=NULL returns to the static rule.
 * c2r_set_sources resets it.  c2r_last_nbox returns the sub-box count each of this rank's sources
 * ended with in the last c2r_pass_sources (its cost is the volume of that box). */
int  c2r_set_source_share(c2r_ctx *ctx, const int32_t *idx, int32_t n);
int  c2r_last_nbox(c2r_ctx *ctx, int32_t *nbox, int32_t n);
/* The same cost balancing INSIDE the library, for hosts that only call c2r_evolve3d (the Fortran shim): after
 * every pass each rank contributes the sub-box counts of the sources it swept to one small all-reduce (through the
 * callback of c2r_set_rank), and before the next pass every rank computes the same longest-processing-time
 * partition by the cell count of each source's last sub-box.  The first pass after c2r_set_sources / c2r_set_rank
 * uses the static rule.  An explicit c2r_set_source_share takes precedence while it is in force. */
int  c2r_set_balance(c2r_ctx *ctx, int32_t on);
/* The 0-based global indices of the sources this rank sweeps (swept in the last pass, once one has run):
 * *n receives their number, idx (capacity cap) the first min(*n, cap) of them. */
int  c2r_source_share(c2r_ctx *ctx, int32_t *idx, int32_t cap, int32_t *n);
/* The partition c2r_set_balance uses, as a pure host function (no context, no GPU): rank `rank`'s share of
 * nsrc sources with the given costs over nranks ranks; idx has room for nsrc entries. */
int  c2r_balanced_shares(const int64_t *cost, int32_t nsrc, int32_t nranks, int32_t rank, int32_t *idx, int32_t *n);
/* do_grid_master / do_grid_slave (master_slave.F90:124-330: what do_grid selects when npr > min_numproc_master_slave = 10) -- the
 * sources of a pass are handed out ON REQUEST instead of by a fixed rule: a rank that has swept what it had asks for more, so a
 * rank whose sources end early takes more of them.  next(user, pass, want, &first, &count) is the host's queue: it hands the
 * calling rank the next count <= want sources [first, first + count) (0-based) of pass number `pass` and count = 0 once none are
 * left (the reference's master answers a worker's request with the next source number, :170-230; an MPI host implements the queue
 * with MPI_Fetch_and_op on a counter in rank 0's window, a thread host with an atomic; one counter per parity of `pass`, the other
 * one reset when a pass starts, needs no extra synchronisation because a collective follows every pass).  chunk: how many sources
 * a rank asks for at a time (the reference: 1); a chunk is swept as one round of launches, so chunks of 64 and more keep the GPU
 * busy and smaller ones balance finer.  Unlike the reference's master, every rank sweeps.  The shape of the photon-loss sums
 * follows the whole list's size, so per-source results do not depend on who swept what; Gamma differs from the static rule's
 * by the association of the sums over ranks.  c2r_source_share / c2r_last_nbox describe what the rank swept in the last pass.
 * Excludes deterministic_rates (C2R_ESTATE), takes precedence over c2r_set_balance and c2r_set_source_share.  NULL: off. */
typedef int (*c2r_next_sources_fn)(void *user, int64_t pass, int32_t want, int32_t *first, int32_t *count);
int  c2r_set_source_queue(c2r_ctx *ctx, c2r_next_sources_fn next, void *user, int32_t chunk);
/* The device the context runs on (resolves C2R_DEVICE_AUTO). */
int  c2r_get_device(const c2r_ctx *ctx, int32_t *device);

/* ---- device buffers -------------------------------------------------------------------- */
/* Use caller-owned device arrays (N^3 each; ndens f32, the rest f64) instead of the
 * context's own.  Any pointer may be NULL to keep the context's buffer. */
int  c2r_bind_device_buffers(c2r_ctx *ctx, void *ndens, void *xh, void *xh_av,
                             void *xh_intermed, void *phih_grid);
/* which: 0 ndens, 1 xh, 2 xh_av, 3 xh_intermed, 4 phih_grid; non-isothermal runs (context-owned, C2R_ESTATE otherwise):
 * 5 phiheat_grid (f64), 6 temperature_grid (3 x f32 per cell: current, average, intermed); c2r_params.allfrac (context-owned,
 * C2R_ESTATE otherwise): 7, 8, 9 = the stored neutral fractions of xh, xh_av, xh_intermed (N^3 f64 each; 1 - 3 are then the ionized
 * halves).  These three entries move ONE N^3 array per call in every build. */
int  c2r_device_ptr(c2r_ctx *ctx, int32_t which, void **ptr);
int  c2r_upload(c2r_ctx *ctx, int32_t which, const void *host);
int  c2r_download(c2r_ctx *ctx, int32_t which, void *host);

/* ---- the path, piecewise (device-resident data) ------------------------------------------ */
/* set_rates_to_zero (evolve.F90:430-440) */
int  c2r_zero_rates(c2r_ctx *ctx);
/* pass_all_sources (evolve.F90:444-495) for this rank's sources: do_grid_static
 * (master_slave.F90:74-96) -> do_source (evolve_source.F90:58) -> evolve0D
 * (evolve_point.F90:83) -> cinterp (column_density.f90:29) + photoion_rates
 * (radiation_photoionrates.F90:71).  Reads ndens, xh_av; accumulates into phih_grid.
 * Does NOT reduce across ranks (c2r_allreduce_rates does). */
int  c2r_pass_sources(c2r_ctx *ctx, double *photon_loss, int64_t *sum_nbox, int64_t *visited);
/* The all-reduce of the rates overlapped with the sweep.  on = 1: where a pass would be followed by an all-reduce of the WHOLE
 * grid (several ranks, no slab chemistry, f64 atomics, isothermal, at least 64 sources on this rank, rates not sparse by the
 * previous pass's sub-boxes), c2r_pass_sources sweeps this rank's sources as two halves into two pairs of accumulators and
 * hands the first half's rates to the all-reduce callback -- on a second stream -- while the second half is swept; both
 * reduced halves are then added.  c2r_allreduce_rates after such a pass only refreshes the per-source sub-box list.  The
 * result is the plain path's up to the association of the sums (1e-16 relative).  Every rank must make the same choice, and
 * the callback must honour its stream argument (the RCCL binding and the torch.distributed host do).  Off by default. */
int  c2r_set_exchange_overlap(c2r_ctx *ctx, int32_t on);

/* mpi_accumulate_grid_quantities (evolve.F90:577-616) through the callback; no-op for 1 rank.
 * Sparse form (on by default): the rates of a pass are non-zero only inside the sources' final sub-boxes, which every rank
 * learns through one small all-reduce of the sub-box counts; while the boxes' volumes add up to at most half the mesh only
 * they travel (packed box after box in source order, reduced through the same callback, written back) -- the result is the
 * all-reduce's, element by element the same sum over the ranks.  A cold 256^3 x 1000 step exchanges 8 % of N^3 x 8 B per
 * iteration instead of all of it.  c2r_exchange_stats: calls so far, how many went packed, bytes of the last call / in all. */
int  c2r_allreduce_rates(c2r_ctx *ctx);
int  c2r_exchange_stats(c2r_ctx *ctx, int64_t *calls, int64_t *sparse_calls, int64_t *bytes_last, int64_t *bytes_total);
/* do_source (evolve_source.F90:58) for ONE source ns (1-based), for tests: optionally returns
 * the source's full coldensh_out grid (evolve_data.F90 coldensh_out) to a host array. */
int  c2r_do_source(c2r_ctx *ctx, int32_t ns, double *coldensh_out_host, double *photon_loss_src,
                   int32_t *nbox, int64_t *visited);
/* global_pass (evolve.F90:499-573): evolve0D_global + do_chemistry + doric over the mesh.
 * sum_xh1 (optional) receives sum(xh_intermed) after the pass. */
int  c2r_global_pass(c2r_ctx *ctx, double dt, int64_t *conv_flag, double *sum_xh1);
/* One outer iteration of evolve3D's loop on a SINGLE rank (evolve.F90:243-269): set_rates_to_zero + pass_all_sources +
 * global_pass, the results of c2r_pass_sources and c2r_global_pass together.  Same arithmetic as the three calls; where
 * the sources are few (<= 32, one batch) the whole iteration is one replayed hipGraph with one host wait -- the launches
 * after the pass are recorded behind it and gated on the device by "every source has retired" -- which is what a
 * 128^3 x 1-source iteration is made of.  C2R_ESTATE with more than one rank (a collective belongs between
 * the pass and the global pass there). */
int  c2r_iterate(c2r_ctx *ctx, double dt, double *photon_loss, int64_t *sum_nbox, int64_t *visited, int64_t *conv_flag,
                 double *sum_xh1);
/* The four mesh sums of photonstatistics.F90 in one pass over device arrays `which_l`/`which_r`
 * (1 xh, 2 xh_av, 3 xh_intermed): out = { sum n(1-x_l), sum n x_l, recombination sum, collisional
 * ionization sum } (state_before/_after :104-217, total_rates :137-185), unscaled by vol and dt. */
int  c2r_photon_sums(c2r_ctx *ctx, int32_t which_l, int32_t which_r, double out[4]);
/* sum() of one of the device arrays (evolve.F90:183) */
int  c2r_sum(c2r_ctx *ctx, int32_t which, double *sum);

/* ---- the path, piecewise, on the driver's host arrays (Fortran shim: do_source, global_pass) ----- */
/* do_source(dt,ns1,niter) (evolve_source.F90:58) with the module arrays it touches as arguments:
 * reads ndens, xh_av; ADDS the source's rates into phih_grid (evolve_point.F90:283); fills the
 * source's coldensh_out (N^3, may be NULL); returns photon_loss_src (:216) and nbox (:219).
 * ns = 1..NumSrc as in c2r_do_source. */
int  c2r_do_source_host(c2r_ctx *ctx, int32_t ns, const float *ndens, const double *xh_av,
                        double *phih_grid, double *coldensh_out, double *photon_loss_src, int32_t *nbox);
/* do_grid(dt,niter) (master_slave.F90:53-96; do_grid_static's loop `do ns1=1+rank,NumSrc,npr: call do_source`, or the
 * cost-balanced share once c2r_set_balance has learnt the costs) with the module arrays it touches as arguments: reads
 * ndens, xh_av; ADDS the rates of all of this rank's sources into phih_grid (and phiheat_grid in a non-isothermal run,
 * NULL otherwise); returns what the loop adds to photon_loss(1) (evolve_source.F90:216) and to sum_nbox (:219).  One
 * pass on the device for all of the rank's sources -- not NumSrc separate do_source calls. */
int  c2r_do_grid_host(c2r_ctx *ctx, const float *ndens, const double *xh_av, double *phih_grid, double *phiheat_grid,
                      double *photon_loss, int64_t *sum_nbox);
/* evolve0D(dt,rtpos,ns,niter) (evolve_point.F90:83-299) for ONE cell of source ns (1-based) on the caller's arrays -- the per-cell
 * call the reference's sweep routines make (evolve_source.F90:227-591): rtpos is the unwrapped mesh position, last_l / last_r
 * the current sub-box (evolve_data.F90:70-71); does nothing when coldensh_out(pos) is already set (:125); otherwise cinterp from
 * the caller's coldensh_out (column_density.f90:29-271, bit-identical), the rate, coldensh_out(pos) = ..., phih_grid(pos) += ...,
 * phiheat_grid(pos) += ... (non-isothermal; else may be null), *photon_loss_src += ... when the cell lies on the sub-box surface
 * (:288-293).  A launch and a few small copies per cell: slow by construction, for hosts that walk the cells themselves and for
 * tests; c2r_do_source / c2r_pass_sources are the product path. */
int  c2r_evolve0d_host(c2r_ctx *ctx, int32_t ns, const int32_t rtpos[3], const int32_t last_l[3], const int32_t last_r[3],
                       const float *ndens, const double *xh_av, double *coldensh_out, double *phih_grid, double *phiheat_grid,
                       double *photon_loss_src);
/* evolve0D_global(dt,pos,conv_flag) (evolve_point.F90:305-406) for ONE cell (pos 1-based): do_chemistry with the collected
 * rates and the global convergence test, the same kernel as the mesh-wide pass on a one-cell slab; *conv_flag is incremented
 * when the cell has not converged (:384-391).  phiheat_grid / temperature_grid (3 x f32 per cell): non-isothermal contexts. */
int  c2r_global_pass_cell_host(c2r_ctx *ctx, double dt, const int32_t pos[3], const float *ndens, const double *xh, double *xh_av,
                               double *xh_intermed, const double *phih_grid, const double *phiheat_grid, float *temperature_grid,
                               int32_t *conv_flag);
/* global_pass(conv_flag,dt) (evolve.F90:499): evolve0D_global over the mesh with the host arrays.  In a non-isothermal
 * context phiheat_grid and temperature_grid are used as they lie on the device (c2r_upload arrays 5 and 6 first, c2r_download
 * array 6 afterwards). */
int  c2r_global_pass_host(c2r_ctx *ctx, double dt, const float *ndens, const double *xh, double *xh_av,
                          double *xh_intermed, const double *phih_grid, int64_t *conv_flag);

/* ---- the path, whole ------------------------------------------------------------------- */
/* evolve3D(time,dt,restart=0) (evolve.F90:83-281) on the device-resident arrays. */
int  c2r_evolve3d_dev(c2r_ctx *ctx, double dt, c2r_report *rep);
/* evolve3D(time,dt,restart/=0) (evolve.F90:153-157): the caller has loaded an iteration dump
 * (start_from_dump, evolve.F90:328-426: niter, photon_loss_all, phih_grid, xh_av, xh_intermed)
 * into the device arrays; runs the global pass and continues the convergence loop from `niter`. */
int  c2r_evolve3d_restart_dev(c2r_ctx *ctx, double dt, int32_t niter, double photon_loss_all,
                              c2r_report *rep);
/* evolve3D on the driver's host arrays: uploads ndens and xh, runs the loop, downloads xh,
 * xh_av, xh_intermed, phih_grid (any output pointer may be NULL).  This is what the Fortran
 * shim calls. */
int  c2r_evolve3d(c2r_ctx *ctx, double dt, const float *ndens, double *xh, double *xh_av,
                  double *xh_intermed, double *phih_grid, c2r_report *rep);
/* evolve3D(time,dt,restart/=0) on the driver's host arrays, after the shim's start_from_dump
 * (evolve.F90:328-426) has read niter, photon_loss_all, phih_grid, xh_av, xh_intermed from the dump. */
int  c2r_evolve3d_restart(c2r_ctx *ctx, double dt, int32_t niter, double photon_loss_all,
                          const float *ndens, double *xh, double *xh_av, double *xh_intermed,
                          double *phih_grid, c2r_report *rep);
/* evolve3D of a non-isothermal run on the driver's host arrays (restart_niter < 0: restart=0; otherwise the shim's
 * start_from_dump has read niter, photon_loss_all and ALL of the arrays below from the dump, evolve.F90:364-375):
 * additionally uploads temperature_grid, downloads phiheat_grid and temperature_grid. */
int  c2r_evolve3d_thermal(c2r_ctx *ctx, double dt, int32_t restart_niter, double photon_loss_all, const float *ndens,
                          double *xh, double *xh_av, double *xh_intermed, double *phih_grid, double *phiheat_grid,
                          float *temperature_grid, c2r_report *rep);
/* Called after every outer iteration (global pass done, stream idle) at the point where the
 * reference checks the wall clock and writes an iteration dump (evolve.F90:271-275); the hook may
 * c2r_download() arrays 2,3,4 (xh_av, xh_intermed, phih_grid) and, in a non-isothermal run, 5 and 6 (phiheat_grid,
 * temperature_grid: the two extra records of such a dump, evolve.F90:314-317).  Non-zero return aborts (C2R_ECALLBACK). */
typedef int (*c2r_iteration_fn)(void *user, int32_t niter, double photon_loss_all);
int  c2r_set_iteration_hook(c2r_ctx *ctx, c2r_iteration_fn fn, void *user);

/* ---- rate tables (one-time set-up; host code, needs no GPU and no context) ------------------ */
/* SED and table parameters: compile-time `parameter`s of sed_parameters.f90, radiation_sizes.f90,
 * radiation_tables.F90:45-47 and the cgs constant modules. */
#define C2R_SED_BLACK_BODY 1      /* sed_parameters.f90:26 stellar_SED_type: black body (the shipped value) */
#define C2R_SED_POWER_LAW  2      /* ... power law in photon number, frequency**(-pl_index) (radiation_tables.F90:455-466) */
typedef struct c2r_sed_params {
    double T_eff, S_star, min_freq, max_freq;   /* black body, sed_parameters.f90 */
    double pl_index_cross_section;              /* radiation_sizes.f90:85 */
    double hplanck, k_B, two_pi_over_c_square, R_solar, pi;
    double minlogtau, maxlogtau;
    int32_t numtau;
    int32_t sed_type;                           /* C2R_SED_BLACK_BODY (0 means the same) or C2R_SED_POWER_LAW */
    double  pl_index;                           /* sed_parameters.f90:40: photon-number index of the power-law SED */
    int32_t grey, reserved1;                    /* c2ray_parameters.f90:43 grey: frequency-independent cross section
                                                 * (radiation_tables.F90:338-347) */
} c2r_sed_params;
int  c2r_default_sed(c2r_sed_params *p);
/* the shipped power-law parameters (sed_parameters.f90:38-45: index 3, 1e48 photons/s between the HI and HeII edges) */
int  c2r_default_sed_power_law(c2r_sed_params *p);
/* rad_ini (radiation_tables.F90:95-126): spectrum_parms, setup_scalingfactors,
 * romberg_initialisation, spec_diag, spec_integration for the black-body or the power-law source
 * (sed_type) with NumFreqBnd=1, frequency-dependent or grey opacity.  Fills stellar_photo_thick_table / _thin_table(0:numtau); n = numtau+1.
 * R_star (optional) returns the rescaled black-body radius the reference logs. */
int  c2r_build_tables(const c2r_sed_params *sed, double *thick, double *thin, int32_t n, double *R_star);
/* The heating tables of a non-isothermal run (fill_heating_integrands_HI + make_heat_tables_HI,
 * radiation_tables.F90:455-543): the photo integrands times hplanck*(nu - ion_freq_HI). */
int  c2r_build_heat_tables(const c2r_sed_params *sed, double ion_freq_HI, double *heat_thick, double *heat_thin, int32_t n);

/* Device self-test: the kernels' division helpers (bare Newton-Raphson core, reciprocal-multiply
 * for launch-invariant divisors) against the compiler's IEEE division on 4x2^22 pseudo-random
 * operand sets; *mismatches must come back 0. */
int  c2r_selftest(c2r_ctx *ctx, int64_t *mismatches);

/* ---- measurement -------------------------------------------------------------------------- */
/* HIP-event timing of the two hot kernels on the context's stream (bench.py's roofline leg).
 * mode 0: off; 1: an event pair around every k_sweep_shell launch (the roofline measurement; the extra
 * barrier packets cost ~3 % when launches are short, e.g. 125 sources per GPU); 2: one pair per sub-box
 * (5 launches plus the small kernels between them: cheaper, slightly pessimistic).  Any mode resets the
 * counters.  sweep_ms is the sum of the intervals, sweep_launches the k_sweep_shell launches they cover. */
int  c2r_profile(c2r_ctx *ctx, int32_t mode);
int  c2r_profile_read(c2r_ctx *ctx, double *sweep_ms, int64_t *sweep_launches,
                      double *chem_ms, int64_t *chem_launches);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* C2RAY_HIP_H */
