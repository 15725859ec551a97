// Device code of the C2-Ray evolve hot path for gfx950 (MI355X, CDNA4) -- part 1 of 4: what every translation unit of the
// library shares: the kernels' argument structs, the f64 helpers (IEEE-exact division and square root without the generic
// expansion, the table-driven log10), the rate arithmetic of one (cell, source), buffer addressing.
// (kernels_sweep.hpp: the sweep; kernels_chem.hpp: the global pass and the reductions; kernels_exchange.hpp: packing.)
//
// Design:
//  * The short-characteristics sweep of one source is causal only from one Chebyshev shell
//    (cube surface |d|_inf = q) to the next: every upstream cell that cinterp gives a non-zero
//    weight lies in shell q-1 (column_density.f90:108,173,226).  So shell q of ALL sources of a
//    batch is one launch; its 24q^2+2 cells per source are independent.
//  * A source's column densities live only in two "shell" buffers (planes of the 6 cube faces,
//    ping-pong by q parity), not in an N^3 array per source (evolve_data.F90 coldensh_out).
//  * f64 throughout, -ffp-contract=off: statement order follows the reference so that results
//    agree with the Fortran to rounding of the transcendental functions only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c2r {

#ifndef C2R_BLOCK
#define C2R_BLOCK 256
#endif
constexpr int kFusedQmaxK = 10;    // (= kFusedQmax of ctx.hpp, static_assert in sweep.hip) the fused first sub-boxes reach shell q <= this: their LDS planes are (2 q + 1)^2
constexpr int kBlock = C2R_BLOCK;   // threads per block of the sweep; a face's owned rectangle is flattened into tiles of kBlock

typedef double v2f64 __attribute__((ext_vector_type(2)));
constexpr int kLogTab = 64;              // intervals of the log10 table (log10_tab)

// The scalars of a time step live in DEVICE memory, not in the kernel arguments: the cell size and volume (cosmological
// expansion: C2Ray.F90:360-376 changes them every step), the homogeneous LLS column, what a shell derives from them, and
// the global pass's step constants (dt; doric.f90:73-78 at the step's temperature; cosmo_cool's redshift).  A replayed
// hipGraph bakes its kernel arguments in; with these behind a pointer the captured launch sequence of a small batch stays
// valid from time step to time step (no re-capture: 0.15 ms per step where an outer iteration takes 0.1 - 0.25 ms) and no
// setter can leave a stale constant in a captured node.  The host refreshes the block when a value changes (sync_step).
// Kernels read it through constant-address-space views (step_of, shell_step): scalar loads.
struct ShellStep { double d2axis[3]; double path_scale, lls_scale; };      // per shell q: (dr_d q)^2, dr[0]/q, coldensh_LLS/q (1/q with an LLS grid)
struct ChemStep { double dt, brech0, acolh0, recpow, clumping, sqrtt, expt, zp, dzdt; };
struct StepBlock {
    double dr[3], dr2[3], vol, coldensh_LLS, inv_dr0;
    int exact_udiv_dr0, n_shell;
    ChemStep chem;
    // ShellStep shell[n_shell] follows (KParams::shell_step points at it)
};

struct KParams {
    int n[3];
    int hl[3], hr[3];          // trace limits around a source: -hl..+hr (evolve_source.F90:100-102)
    const StepBlock *step;         // dr, dr2, vol, coldensh_LLS, inv_dr0 (step_of)
    const ShellStep *shell_step;   // [Qmax + 1]
    double sigma, wfloor, sqrt2, sqrt3, fourpi;
    double max_coldensh, tau_limit, minlogtau, dlogtau, numtau_d, eps;
    // correctly rounded reciprocals of launch-invariant divisors (exact division in 3 FMAs, see udiv)
    double inv_dlogtau;
    int exact_udiv;            // 0: dlogtau fails the precondition of udiv -> plain IEEE division (dr[0]: StepBlock::exact_udiv_dr0)
    int numtau;
    int R, P;                  // plane centre offset and pitch (P = 2R+1)
    size_t PP;                 // P*P
    // n_HI = max(1-max(xh_av,eps),eps)*ndens per cell (evolve_point.F90:137-146, doric.f90:153), the only
    // way the sweep uses xh_av and ndens: evaluated once per cell and pass by k_prepare_nhi instead of
    // once per (cell, source); nhi is [k][j][i] (i fastest), nhi_T [k][i][j] (j fastest) for the +-x
    // faces, whose waves run along y
    const double *nhi, *nhi_T;
    double *phih;
    double *phih_T;            // Gamma of the +-x faces, added back after the pass
    // non-default physics switches (c2ray_parameters.f90:80-99)
    int lls_type;              // 1 homogeneous, 2 per-cell grid, 3 hard barrier
    double R_max2;             // R_max_LLS^2 (type 3)
    const float *lls, *lls_T;  // LLS_grid and its (x,y)-transposed replica (type 2)
    double *gbox_h;            // ... and per-source heating rates of a non-isothermal run, same layout (null: isothermal)
    double *gbox;              // deterministic mode: [S_batch][2][ncell] per-source Gamma ([0] x-fastest cells of
                               // z/y faces, [1] y-fastest cells of x faces); null: atomics into phih/phih_T
    const double *thick, *thin;
    // non-isothermal runs (c2ray_parameters.f90:28 isothermal=.false.; HEAT kernels only, EXT & 1): heating tables
    // stellar_heat_thick/thin_table (padded like thick/thin), phiheat_grid and its transposed accumulator
    const double *hthick, *hthin;
    double *heat, *heat_T;
    double tau_heat_limit;    // radiation_photoionrates.F90:333
    const v2f64 *logtab;      // [kLogTab] {r_i, -log10 r_i} for log10_tab
    // tolerance ("fast") mode of the sweep (c2r_params.sweep_mode = 1, k_sweep_shell_fast)
    const v2f64 *odtab;       // [kLogTab] {r_i, 1 + (-log10 r_i - minlogtau)/dlogtau}: table position of tau = 1/r_i
    double od_per_e, od_per_ln; // log10(2)/dlogtau, log10(e)/dlogtau
    const int    *srcpos;      // 3 x S_batch (unwrapped, 1-based: cinterp's real(i0) needs it)
    const int    *srcw;        // 3 x S_batch wrapped to 0..N-1
    const double *normflux;    // S_batch
    // builds with use_xray_SED=.true. (sed_parameters.f90:56; XRAY kernels only, EXT & 2): the second source type of
    // photoion_rates (radiation_photoionrates.F90:133-137) -- its tables (padded like thick/thin) and NormFlux_xray per source
    const double *xthick, *xthin;
    const double *xhthick, *xhthin;   // ... and its heating tables (xray_heat_thick/thin_table; HEAT and XRAY together, EXT == 3)
    const double *normflux_x;  // S_batch
    double *planes;            // [S_batch][2][6][P][P]
};

// Cells of one cube face that this face OWNS in shell q, as a rectangle in plane coordinates
// (a,b), already clipped to the trace limits; flattened row-major into tiles of 256 threads.
struct FaceRect {
    int a_lo, wa, b_lo, wb;
    unsigned magic;            // t / wa == umulhi(t, magic) for t < wa*wb (0: wa == 1)
    int ntiles;                // tiles of k_sweep_shell (groups of kRows rows); 0: face absent from this shell
    int pp, npr;               // row groups of the rows b >= 0; row groups in all (k_sweep_shell walks wa x npr groups)
};

struct ShellArgs {
    int q;
    int has_boundary;
    int buf_prev, buf_cur;       // fast mode: which of a source's two plane sets holds the previous shell / receives this one
                                 // ((q-1)&1, q&1 while every launch is one shell; the look-ahead pairs advance two shells per set)
    int tiles_max;               // grid.x; loss_partial is [n_active][6][tiles_max]
    int boxR[3], boxL[3];        // limits of the current sub-box (last_r/last_l - srcpos)
    double alam;                 // (q-0.5)/q, column_density.f90:112 (sign cancels)
    double dp2, inv_dp2;         // q*q and its correctly rounded reciprocal
    double inv_q;                // fast mode: 1/q  (what else a shell derives from the step's scalars: KParams::shell_step[q])
    FaceRect face[6];
    const int *active;           // compacted list of local source indices
    const int *n_active;         // its length on the device: the grid may be sized by an older, larger count
    double *loss_partial;
    double *dbg_cdout;           // optional N^3 coldensh_out of the (single) source, else null
};

__device__ __forceinline__ int pmod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }

#define C2R_AS4 __attribute__((address_space(4)))
// constant-address-space views of the step block: uniform addresses, so the loads are scalar (s_load) whatever else the kernel writes
__device__ __forceinline__ const C2R_AS4 StepBlock &step_of(const KParams &p) { return *(const C2R_AS4 StepBlock *)p.step; }
__device__ __forceinline__ const C2R_AS4 ShellStep &shell_step(const KParams &p, int q) { return ((const C2R_AS4 ShellStep *)p.shell_step)[q]; }

// ---- IEEE-exact f64 division without the generic expansion ---------------------------------------
// hipcc expands a/b into div_scale x2, rcp, 4 fma, mul, fma, div_fmas, div_fixup.  The scaling and
// fix-up only matter for operands near the exponent limits; every quotient of this kernel is far
// inside the normal range, so the bare Newton-Raphson core gives the same correctly rounded
// result in 8 instructions.  c2r_selftest compares both forms bit for bit on the device.
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double fdiv(double n, double d)
{
    const double r = rcp_nr(d);
    const double q = n * r;
    const double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r, q);
}
__device__ __forceinline__ double frcp(double d)        // 1.0/d
{
    const double r = rcp_nr(d);
    const double rem = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(rem, r, r);
}
// sqrt for arguments of order one (here 1 <= x <= 3: the path-length factor).  hipcc's expansion wraps the
// same Goldschmidt/Newton core in exponent scaling and class fix-ups for tiny, huge and special inputs
// (22 instructions); the bare core (10) returns the same correctly rounded root inside the normal range.
__device__ __forceinline__ double fsqrt(double x)
{
#ifdef C2R_SQRT_GENERIC
    return sqrt(x);
#else
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
#endif
}
// n/d for a launch-invariant d with rd = RN(1/d) from the host (Markstein: q' = RN(q + r*rd) with
// r = n - q*d exact is the correctly rounded quotient unless d's significand is all ones; the
// host checks that and clears exact_udiv otherwise).
__device__ __forceinline__ double udiv(double n, double d, double rd, int exact)
{
    if (!exact) return n / d;
    const double q = n * rd;
    const double r = __builtin_fma(-q, d, n);
    return __builtin_fma(r, rd, q);
}

// log10 for the table position (radiation_photoionrates.F90:195).  The device library's log10
// spends 105 VALU instructions on double-double arithmetic to stay under 1 ulp RELATIVE error;
// the table position needs ABSOLUTE accuracy in log10(tau) (od = 1 + (lt+20)/0.012), which the
// classic argument reduction x = 2^e * m, m in [sqrt(1/2), sqrt(2)), log(m) = 2 atanh(s) with
// s = f/(2+f) and a degree-7 minimax polynomial in s^2 (the published fdlibm e_log.c / e_log10.c
// scheme and coefficients) delivers in ~32 instructions: error <= ~1 ulp of the result, the same
// class as glibc's log10 that the reference calls.  -DC2R_LOG10_OCML selects the library version.
__device__ __forceinline__ double log10_pos(double x)      // x > 0, normal
{
#ifdef C2R_LOG10_OCML
    return log10(x);
#else
    double m = __builtin_amdgcn_frexp_mant(x);              // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                      // [sqrt(1/2), sqrt(2))
    e = lo ? e - 1 : e;
    const double dk = (double)e;
    const double f = m - 1.0;
    const double s = f * rcp_nr(2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                                        3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01,
                                        1.818357216161805012e-01), 2.857142874366239149e-01),
                                        6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double lm = f - (hfsq - s * (hfsq + R));           // log(m)
    // dk*log10_2hi is exact (low 32 bits of the constant are zero)
    const double hi = __builtin_fma(lm, 4.34294481903251816668e-01, dk * 3.01029995663611771306e-01);
    return __builtin_fma(dk, 3.69423907715893078616e-13, hi);
#endif
}

// The same log10 with a 64-interval table held in LDS (one private copy per wave, filled by
// wave_log_table): x = 2^e * m, m in [0.5,1); interval i = top 6 fraction bits of m; with r_i ~ 1/c_i
// (c_i the interval centre) z = m*r_i - 1 is exact to an fma rounding and |z| <= 2^-7, so
// log(m) = -log(r_i) + log1p(z) needs a degree-7 series only: 17 VALU instructions and one 16-byte LDS
// read instead of 32 VALU -- the LDS port is otherwise idle in this kernel, the VALU port is what binds it.
// Entries: .x = r_i, .y = -log10(r_i) (host, long double).  Error <= ~1 ulp of the result like log10_pos.
__device__ __forceinline__ double log10_tab(double x, const v2f64 *__restrict__ tab)   // x > 0, normal; tab in LDS
{
    const double m = __builtin_amdgcn_frexp_mant(x);        // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    const unsigned i = ((unsigned)__double2hiint(m) >> 14) & 63u;
    const v2f64 rt = tab[i];
    const double z = __builtin_fma(m, rt.x, -1.0);
    // log1p(z) = z + z^2 * (-1/2 + z/3 - z^2/4 + z^3/5 - z^4/6 + z^5/7); next term z^8/8 < 2^-59
    double P = __builtin_fma(z, 1.0 / 7.0, -1.0 / 6.0);
    P = __builtin_fma(z, P, 0.2);
    P = __builtin_fma(z, P, -0.25);
    P = __builtin_fma(z, P, 1.0 / 3.0);
    P = __builtin_fma(z, P, -0.5);
    const double l1p = __builtin_fma(z * z, P, z);
    return __builtin_fma((double)e, 3.01029995663981198017e-01, __builtin_fma(l1p, 4.34294481903251816668e-01, rt.y));
}
// Every wave keeps its own copy of the table in LDS: filled with all 64 lanes active at the top of the
// kernel, read later by the same wave only, so no barrier is needed (same-wave LDS accesses are ordered).
__device__ __forceinline__ const v2f64 *wave_log_table(const v2f64 *__restrict__ g, v2f64 *lds /* [waves][64] */)
{
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    v2f64 *mine = lds + (tid & ~63u);
    mine[lane] = g[lane];
    return mine;
}

// radiation_photoionrates.F90:184-208  set_tau_table_positions
struct TauPos { int ip, ip1; double res; };
__device__ __forceinline__ TauPos tau_pos(double tau, const KParams &p, const v2f64 *__restrict__ ltab)
{
#ifdef C2R_LOG10_NOTAB
    const double lt = log10_pos(fmax(1.0e-20, tau));
#else
    const double lt = log10_tab(fmax(1.0e-20, tau), ltab);
#endif
    const double od = fmin(p.numtau_d, fmax(0.0, 1.0 + udiv(lt - p.minlogtau, p.dlogtau, p.inv_dlogtau, p.exact_udiv)));
    TauPos t;
    t.ip = (int)od;
    t.res = od - (double)t.ip;
    t.ip1 = min(p.numtau, t.ip + 1);
    return t;
}
// radiation_photoionrates.F90:212-228  read_table
__device__ __forceinline__ double read_table(const double *__restrict__ tab, const TauPos &t)
{
    // the device tables carry one extra element equal to the last (tab[numtau+1] = tab[numtau]), so the
    // two neighbours tab(ip), tab(ip1 = min(numtau, ip+1)) are always tab[ip], tab[ip+1]: one 16-byte load
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    const d2u v = *reinterpret_cast<const d2u *>(tab + t.ip);
    return v.x + (v.y - v.x) * t.res;
}

// radiation_photoionrates.F90:71-179, :233-317 for NumFreqBnd=1, stellar table.
// Returns photo_cell_HI (already divided by vol_ph); out = photo_out.
// EXT & 1 (HEAT): also phi%heat of heat_lookuptable (:323-417) from the same two table positions.
template <int EXT = 0>
__device__ __forceinline__ double photoion(const KParams &p, const v2f64 *__restrict__ ltab, double cd_in, double cd_out,
                                           double vol_ph, double nflux, double &p_out, double *heat = nullptr, double nflux_x = 0.0)
{
    const double tau_in = cd_in * p.sigma, tau_out = cd_out * p.sigma;
    const TauPos pin = tau_pos(tau_in, p, ltab);
    const double p_in = nflux * read_table(p.thick, pin);
    double p_cell;
    TauPos pout = pin;
    const bool thick_cell = fabs(tau_out - tau_in) > p.tau_limit;
    if (thick_cell) {
        pout = tau_pos(tau_out, p, ltab);
        p_out = nflux * read_table(p.thick, pout);
        p_cell = p_in - p_out;
    } else {
        p_cell = nflux * (tau_out - tau_in) * read_table(p.thin, pin);
        p_out = p_in - p_cell;
    }
    if (EXT & 1) {
        const double h_in = nflux * read_table(p.hthick, pin);                         // :384
        if (fabs(tau_out - tau_in) > p.tau_heat_limit) {                               // :388
            if (!thick_cell) pout = tau_pos(tau_out, p, ltab);                         // (only if tau_heat_limit < tau_photo_limit)
            *heat = fdiv(h_in - nflux * read_table(p.hthick, pout), vol_ph);
        } else {
            const double tau_cell = (cd_out - cd_in) * p.sigma;                        // :104, :146
            *heat = fdiv(nflux * tau_cell * read_table(p.hthin, pin), vol_ph);         // :396-400
        }
    }
    double rate = fdiv(p_cell, vol_ph);
    if ((EXT & 2) && nflux_x > 0.0) {             // :133-137  phi = phi + photo_lookuptable(..., NormFlux_xray(nsrc), "P", vol)
        const double x_in = nflux_x * read_table(p.xthick, pin);
        double x_cell, x_out;
        if (thick_cell) {
            x_out = nflux_x * read_table(p.xthick, pout);
            x_cell = x_in - x_out;
        } else {
            x_cell = nflux_x * (tau_out - tau_in) * read_table(p.xthin, pin);
            x_out = x_in - x_cell;
        }
        p_out = p_out + x_out;
        rate = rate + fdiv(x_cell, vol_ph);
        if (EXT & 1) {                            // :165-171  phi = phi + heat_lookuptable(..., NormFlux_xray(nsrc), "P", ...)
            const double hx_in = nflux_x * read_table(p.xhthick, pin);
            if (fabs(tau_out - tau_in) > p.tau_heat_limit) *heat = *heat + fdiv(hx_in - nflux_x * read_table(p.xhthick, pout), vol_ph);
            else *heat = *heat + fdiv(nflux_x * ((cd_out - cd_in) * p.sigma) * read_table(p.xhthin, pin), vol_ph);
        }
    }
    return rate;
}

// Deterministic block sum (fixed order): wave shuffles, then the 4 wave sums in order.
__device__ __forceinline__ double block_sum_256(double v, double *sm /* >= 4 doubles */)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) { const int nw = (blockDim.x * blockDim.y) >> 6; for (int w = 0; w < nw; ++w) r += sm[w]; }
    return r;   // valid in thread 0
}

// ---- the rate arithmetic of one (cell, source), shared by both sweep modes -------------------------------------
// radiation_photoionrates.F90:71-317 + evolve_point.F90:262 restated so that it costs ~60 instead of ~150 vector
// instructions; what it gives up against the statement-by-statement form (photoion above, kept for the source cell) is
// far inside the Gamma tolerance both modes state (tests/_util.TOL: the rounding of the table position dominates either):
//  * the table position 1+(log10 tau-minlogtau)/dlogtau (:195-199) comes straight out of the log evaluation: the per-wave
//    LDS table holds positions instead of logarithms, the two scale factors are folded into the last two FMAs;
//  * reciprocals by v_rcp_f64 + one Newton step (2^-48) instead of the correctly rounded quotient;
//  * Gamma = NormFlux (T_in - T_out) / (vol_ph n_HI) with ONE reciprocal (:262-263, evolve_point.F90:262).
__device__ __forceinline__ double rcp1(double d)        // 1/d to 2^-48
{
    const double r = __builtin_amdgcn_rcp(d);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}
// table position od = min(numtau, 1 + (log10(max(1e-20,tau)) - minlogtau)/dlogtau); tab = wave's LDS copy of p.odtab
__device__ __forceinline__ double tau_od(double tau, const KParams &p, const v2f64 *__restrict__ tab)
{
    const double x = fmax(1.0e-20, tau);
    const double m = __builtin_amdgcn_frexp_mant(x);        // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    const unsigned i = ((unsigned)__double2hiint(m) >> 14) & 63u;
    const v2f64 rt = tab[i];
    const double z = __builtin_fma(m, rt.x, -1.0);
    double P = __builtin_fma(z, 1.0 / 7.0, -1.0 / 6.0);
    P = __builtin_fma(z, P, 0.2);
    P = __builtin_fma(z, P, -0.25);
    P = __builtin_fma(z, P, 1.0 / 3.0);
    P = __builtin_fma(z, P, -0.5);
    const double l1p = __builtin_fma(z * z, P, z);
    const double od = __builtin_fma((double)e, p.od_per_e, __builtin_fma(l1p, p.od_per_ln, rt.y));
    return fmin(p.numtau_d, od);
}
// read_table at position od >= 1 (radiation_photoionrates.F90:212-228); tables padded by one element
__device__ __forceinline__ double table_at(const double *__restrict__ tab, double od)
{
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    const d2u v = *reinterpret_cast<const d2u *>(tab + (int)od);
    return __builtin_fma(v.y - v.x, __builtin_amdgcn_fract(od), v.x);
}

// photo-ionization (and heating) rate of a cell from its entry / exit columns; vol_ph = 4 pi dist2 path; volnhi = vol_ph n_HI.
// p_out: photo_out, the photons leaving the cell (NormFlux x the thick-table value at the exit column, both source types), for
// the photon loss.
template <int EXT>
__device__ __forceinline__ double rates_fast(const KParams &p, const v2f64 *__restrict__ ltab, const double *__restrict__ thick,
                                             const double cd_in, const double cd_out, const double nflux, const double nflux_x,
                                             const double volnhi, const double vol_ph, double &p_out, double &heat)
{
    const double tau_in = cd_in * p.sigma, tau_out = cd_out * p.sigma;
    const double od_in = tau_od(tau_in, p, ltab);
    const double t_in = table_at(thick, od_in);
    double dT, t_out, od_out = od_in;
    const bool thick_cell = fabs(tau_out - tau_in) > p.tau_limit;
    if (thick_cell) {
        od_out = tau_od(tau_out, p, ltab);
        t_out = table_at(thick, od_out);
        dT = t_in - t_out;
    } else {
        dT = (tau_out - tau_in) * table_at(p.thin, od_in);
        t_out = t_in - dT;
    }
    p_out = nflux * t_out;                                                 // photo_out (radiation_photoionrates.F90:292, :302)
    if (EXT & 1) {       // heat_lookuptable (radiation_photoionrates.F90:323-417) at the same table positions
        const double h_in = table_at(p.hthick, od_in);
        double dH;
        if (fabs(tau_out - tau_in) > p.tau_heat_limit) {
            if (!thick_cell) od_out = tau_od(tau_out, p, ltab);
            dH = h_in - table_at(p.hthick, od_out);
        } else dH = ((cd_out - cd_in) * p.sigma) * table_at(p.hthin, od_in);
        heat = (nflux * dH) * rcp1(vol_ph);                                // phi%heat = .../vol_ph
    }
    const double r = rcp1(volnhi);
    double gamma = (nflux * dT) * r;                                       // photo_cell_HI / (n_HI vol_ph)
    if ((EXT & 2) && nflux_x > 0.0) {      // :133-137 the "P" source type: the same table positions, its own two tables and flux
        const double x_in = table_at(p.xthick, od_in);
        double dX, x_out;
        if (thick_cell) { x_out = table_at(p.xthick, od_out); dX = x_in - x_out; }
        else { dX = (tau_out - tau_in) * table_at(p.xthin, od_in); x_out = x_in - dX; }
        p_out = p_out + nflux_x * x_out;
        gamma = gamma + (nflux_x * dX) * r;
        if (EXT & 1) {                     // :165-171 its heating rate, from the X-ray heating tables at the same positions
            double dHx;
            if (fabs(tau_out - tau_in) > p.tau_heat_limit) dHx = table_at(p.xhthick, od_in) - table_at(p.xhthick, od_out);
            else dHx = ((cd_out - cd_in) * p.sigma) * table_at(p.xhthin, od_in);
            heat = heat + (nflux_x * dHx) * rcp1(vol_ph);
        }
    }
    return gamma;
}

// ---- buffer addressing (SRSRC descriptor + 32-bit byte offset) ---------------------------------------
// A descriptor built from block-uniform values lets every access use a 32-bit VGPR offset (no 64-bit
// address arithmetic per lane) and gives a free range check: an offset beyond the buffer reads 0,
// which is exactly the value of a zero-weight upstream corner.
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
template <int AUX = 0>      // AUX 1 = glc: read through to L2 (planes written by other waves of the same launch)
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, AUX));
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
// Cache policy of the sweep's streams (buffer instruction aux bits; 2 = nt, non-temporal): a source's shell
// planes are written once and read once, one launch later, after every other source's shell has gone by; the
// n_HI replica is read once per (cell, source).  Marking them non-temporal keeps the L2 for the Gamma
// atomics and the rate tables: +2.8 % in a same-box A/B (stores only +1 %, stores + planes +1 %, all three).
// Only in k_sweep_shell and only on large meshes (STREAM): at 128^3 the hint is neutral, at 64^3 -- everything
// fits the L2s -- it costs 3 %; the fused first sub-boxes re-read their planes within the same workgroup a
// shell later and never use it.
#ifndef C2R_STORE_AUX
#define C2R_STORE_AUX 2
#endif
#ifndef C2R_NHI_AUX
#define C2R_NHI_AUX 2
#endif
#ifndef C2R_PLANE_AUX
#define C2R_PLANE_AUX 2
#endif
template <int AUX = 0>
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, v), r, (int)byte_off, 0, AUX);
}

// periodic wrap (evolve_point.F90:122): srcw + d + n lies in [n/2, 5n/2); min(c, c-n) as unsigned folds
// [n, 2n) onto [0, n), twice
__device__ __forceinline__ unsigned wrap_pos(int srcw, int n, int d)
{
    unsigned c = (unsigned)(srcw + n + d);
    c = min(c, c - (unsigned)n);
    return min(c, c - (unsigned)n);
}

}  // namespace c2r
