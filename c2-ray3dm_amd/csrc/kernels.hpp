// Device kernels of the C2-Ray evolve hot path for gfx950 (MI355X, CDNA4).
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
    // non-isothermal runs (c2ray_parameters.f90:28 isothermal=.false.; HEAT kernels only): heating tables
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
// HEAT: also phi%heat of heat_lookuptable (:323-417) from the same two table positions.
template <bool HEAT = false>
__device__ __forceinline__ double photoion(const KParams &p, const v2f64 *__restrict__ ltab, double cd_in, double cd_out,
                                           double vol_ph, double nflux, double &p_out, double *heat = nullptr)
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
    if (HEAT) {
        const double h_in = nflux * read_table(p.hthick, pin);                         // :384
        if (fabs(tau_out - tau_in) > p.tau_heat_limit) {                               // :388
            if (!thick_cell) pout = tau_pos(tau_out, p, ltab);                         // (only if tau_heat_limit < tau_photo_limit)
            *heat = fdiv(h_in - nflux * read_table(p.hthick, pout), vol_ph);
        } else {
            const double tau_cell = (cd_out - cd_in) * p.sigma;                        // :104, :146
            *heat = fdiv(nflux * tau_cell * read_table(p.hthin, pin), vol_ph);         // :396-400
        }
    }
    return fdiv(p_cell, vol_ph);
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
// t_out: the thick-table value at the exit column times 1 (photo_out / NormFlux), for the photon loss.
template <bool HEAT>
__device__ __forceinline__ double rates_fast(const KParams &p, const v2f64 *__restrict__ ltab, const double *__restrict__ thick,
                                             const double cd_in, const double cd_out, const double nflux, const double volnhi,
                                             const double vol_ph, double &t_out, double &heat)
{
    const double tau_in = cd_in * p.sigma, tau_out = cd_out * p.sigma;
    const double od_in = tau_od(tau_in, p, ltab);
    const double t_in = table_at(thick, od_in);
    double dT, od_out = od_in;
    const bool thick_cell = fabs(tau_out - tau_in) > p.tau_limit;
    if (thick_cell) {
        od_out = tau_od(tau_out, p, ltab);
        t_out = table_at(thick, od_out);
        dT = t_in - t_out;
    } else {
        dT = (tau_out - tau_in) * table_at(p.thin, od_in);
        t_out = t_in - dT;
    }
    if (HEAT) {       // heat_lookuptable (radiation_photoionrates.F90:323-417) at the same table positions
        const double h_in = table_at(p.hthick, od_in);
        double dH;
        if (fabs(tau_out - tau_in) > p.tau_heat_limit) {
            if (!thick_cell) od_out = tau_od(tau_out, p, ltab);
            dH = h_in - table_at(p.hthick, od_out);
        } else dH = ((cd_out - cd_in) * p.sigma) * table_at(p.hthin, od_in);
        heat = (nflux * dH) * rcp1(vol_ph);                                // phi%heat = .../vol_ph
    }
    return (nflux * dT) * rcp1(volnhi);                                    // photo_cell_HI / (n_HI vol_ph)
}

// ---- source cells (q = 0) ------------------------------------------------------------------
// evolve_point.F90:151-160 (source cell) + the common tail of evolve0D, for source s; ltab: log10_tab's table (LDS or global).
// on_surface: degenerate meshes only, the source cell itself sits on the sub-box surface
template <bool HEAT>
__device__ __forceinline__ void source_cell(const KParams &p, const v2f64 *__restrict__ ltab, const int s, const bool on_surface,
                                            double *loss_acc, double *dbg_cdout)
{
    const int i = pmod(p.srcpos[3 * s + 0] - 1, p.n[0]);
    const int j = pmod(p.srcpos[3 * s + 1] - 1, p.n[1]);
    const int k = pmod(p.srcpos[3 * s + 2] - 1, p.n[2]);
    const size_t id = (size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k);
    const double nhi = p.nhi[id];
    const double cd_in = 0.0;
    const double path = 0.5 * step_of(p).dr[0];
    const double vol_ph = step_of(p).dr[0] * step_of(p).dr[1] * step_of(p).dr[2];
    const double cd_out = cd_in + nhi * path;
    // plane q=0 of every face is the single cell (0,0)
    for (int f = 0; f < 6; ++f)
        p.planes[((size_t)s * 2 + 0) * 6 * p.PP + (size_t)f * p.PP + (size_t)p.R * p.P + p.R] = cd_out;
    if (dbg_cdout) dbg_cdout[id] = cd_out;
    const double nflux = p.normflux[s];
    double p_out = 0.0, gamma = 0.0;
    if (nflux > 0.0) {      // cd_in = 0 is never above max_coldensh
        double heat = 0.0;
        gamma = photoion<HEAT>(p, ltab, cd_in, cd_out, vol_ph, nflux, p_out, &heat) / nhi;
        if (!p.gbox) atomicAdd(&p.phih[id], gamma);
        if (HEAT && !p.gbox && heat != 0.0) atomicAdd(&p.heat[id], heat);       // evolve_point.F90:285-286
        if (HEAT && p.gbox) p.gbox_h[(size_t)s * 2 * ((size_t)p.n[0] * p.n[1] * p.n[2]) + id] = heat;
    } else if (HEAT && p.gbox) p.gbox_h[(size_t)s * 2 * ((size_t)p.n[0] * p.n[1] * p.n[2]) + id] = 0.0;
    if (p.gbox) p.gbox[(size_t)s * 2 * ((size_t)p.n[0] * p.n[1] * p.n[2]) + id] = gamma;
    if (on_surface) loss_acc[s] += p_out * step_of(p).vol / vol_ph;
}

// One thread per source (the fused first sub-box does the same itself: BoxArgs::source_cell).
template <bool HEAT>
__global__ void k_source_cells(KParams p, int nsrc, const int *active, int boxR0, int boxR1, int boxR2,
                               int boxL0, int boxL1, int boxL2, double *loss_acc, double *dbg_cdout)
{
    __shared__ v2f64 s_log[kLogTab];                          // blocks of one wave
    const v2f64 *ltab = wave_log_table(p.logtab, s_log);
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= nsrc) return;
    source_cell<HEAT>(p, ltab, active[sl], boxR0 == 0 || boxR1 == 0 || boxR2 == 0 || boxL0 == 0 || boxL1 == 0 || boxL2 == 0,
                      loss_acc, dbg_cdout);
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

// ---- one Chebyshev shell of every active source ------------------------------------------------
// evolve0D (evolve_point.F90:83-299) + cinterp (column_density.f90:29-271) + photoion_rates.
// faces: 0:+z 1:-z 2:+y 3:-y 4:+x 5:-x.  Plane coordinates (a,b): z-face (x,y); y-face (x,z);
// x-face (y,z).  A cell on an edge/corner of the cube belongs to the face of highest priority
// (z over y over x: the branch order of cinterp); its owner also stores it into the other
// faces' planes, which read it in shell q+1.
// grid = (tiles, 6 faces, active sources); a wave runs along plane axis a, which is the
// unit-stride axis of the arrays the face reads (x for z/y faces, y in the transposed replicas
// for x faces).
// DET: deterministic_rates mode (per-source Gamma grids instead of atomics); LLS: type_of_LLS (1,2,3)
// ---- one cell, in two phases so that a thread can interleave two cells ---------------------------------
// CellState: everything evolve0D computes for cell (a,b) of face `face` in shell sa.q of source s before it
// touches memory for writing; cell_commit: the stores, the photo-ionization rate, the atomic and the
// photon loss.  The four upstream corners arrive as values (c*) with their weight reciprocals
// r* = 1/max(0.6, c*sigma) (weightf, column_density.f90:276-293), because neighbouring cells share them.
struct CellState {                 // kept small: it is live while the thread's other cell is worked on
    unsigned id, o8;
    double nhi, cd_in, cd_out, vol_ph;
    bool stop_far;
};

// mesh-axis deltas of plane cell (a,b) on a face normal to `axis` at signed distance pd (block-uniform selects)
struct Delta3 { int d0, d1, d2; };
__device__ __forceinline__ Delta3 mesh_delta(int axis, int pd, int a, int b)
{
    Delta3 d;
    d.d0 = (axis == 0) ? pd : a;
    d.d1 = (axis == 2) ? b : ((axis == 1) ? pd : a);
    d.d2 = (axis == 2) ? pd : b;
    return d;
}
// periodic wrap (evolve_point.F90:122): srcw + d + n lies in [n/2, 5n/2); min(c, c-n) as unsigned folds
// [n, 2n) onto [0, n), twice
__device__ __forceinline__ unsigned wrap_pos(int srcw, int n, int d)
{
    unsigned c = (unsigned)(srcw + n + d);
    c = min(c, c - (unsigned)n);
    return min(c, c - (unsigned)n);
}

// plane offset of (a,b) in bytes and the in-range test of a plane coordinate against shell q-1
__device__ __forceinline__ unsigned plane_off8(const KParams &p, int a, int b)
{
    return ((unsigned)((int)__umul24((unsigned)(b + p.R), (unsigned)p.P) + (a + p.R))) * 8u;   // factors in [0, 2^24)
}

// XONLY: read n_HI and the LLS grid from the x-fastest arrays whatever the face (same values; the look-ahead recompute
// calls this with a per-lane face and must not pick a buffer per lane)
template <int LLS, bool STREAM, bool XONLY = false>
__device__ __forceinline__ CellState cell_state(const KParams &p, const ShellArgs &sa, const int face, const int s,
                                                const int a, const int b, const double c1v, const double c2v,
                                                const double c3v, const double c4v, const double r1, const double r2,
                                                const double r3, const double r4)
{
    CellState cs;
    const int q = sa.q;
    const int axis = 2 - (face >> 1);            // 2:z 1:y 0:x
    const int pd = (face & 1) ? -q : q;
    // mesh-axis deltas and the source coordinates seen along (a,b)
    const Delta3 dl = mesh_delta(axis, pd, a, b);
    const int s0 = p.srcpos[3 * s + 0], s1 = p.srcpos[3 * s + 1], s2 = p.srcpos[3 * s + 2];
    const int su = (axis == 0) ? s1 : s0;
    const int sv = (axis == 2) ? s1 : s2;
    const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0);
    const unsigned c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1);
    const unsigned c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
    // cell index in the array this face reads: x-fastest, or y-fastest in the transposed replicas
    // (block-uniform choice; 24-bit multiplies: every factor is below 2^24)
    const bool xf = !XONLY && (axis == 0);
    const unsigned ca = xf ? c1 : c0, cb = xf ? c0 : c1;
    const unsigned na = xf ? (unsigned)p.n[1] : (unsigned)p.n[0], nb = xf ? (unsigned)p.n[0] : (unsigned)p.n[1];
    cs.id = ca + __umul24(na, cb + __umul24(nb, c2));
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(xf ? p.nhi_T : p.nhi, ncell * 8u);
    cs.nhi = buf_load_f64<STREAM ? C2R_NHI_AUX : 0>(r_x, cs.id * 8u);
    cs.o8 = plane_off8(p, a, b);

    // cinterp, generic in (a,b,pd): the three branches differ only by which axes play (u,v).
    // real(int) conversions of the reference are f32 but exact (|.| < 2^24): cvt i32->f64.
    const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1;
    const int am = a - sga, bm = b - sgb;
    const double du = (double)a, dv = (double)b;
    const double uc = sa.alam * du + (double)su;
    const double vc = sa.alam * dv + (double)sv;
    // real(im)+0.5*sgni (column_density.f90:117) is a half-integer: one conversion of 2*im+sgni
    const double ddu = 2.0 * fabs(uc - 0.5 * (double)(2 * (su + am) + sga));
    const double ddv = 2.0 * fabs(vc - 0.5 * (double)(2 * (sv + bm) + sgb));
    const double w1 = ((1. - ddu) * (1. - ddv)) * r1;
    const double w2 = ((1. - ddv) * ddu) * r2;
    const double w3 = ((1. - ddu) * ddv) * r3;
    const double w4 = (ddu * ddv) * r4;
    double cdi = fdiv(c1v * w1 + c2v * w2 + c3v * w3 + c4v * w4, w1 + w2 + w3 + w4);
    if (q == 1 && (abs(a) == 1 || abs(b) == 1))
        cdi = (abs(a) == 1 && abs(b) == 1) ? p.sqrt3 * cdi : p.sqrt2 * cdi;
    double path = fsqrt(udiv(du * du + dv * dv, sa.dp2, sa.inv_dp2, p.exact_udiv) + 1.0);

    // evolve0D
    path = path * step_of(p).dr[0];
    // dist2 = xs*xs + ys*ys + zs*zs (evolve_point.F90:171-174); the term of the face's own axis
    // is the same for the whole launch (ShellStep::d2axis = (dr_axis * q)^2)
    const double xs = step_of(p).dr[0] * (double)dl.d0;
    const double ys = step_of(p).dr[1] * (double)dl.d1;
    const double zs = step_of(p).dr[2] * (double)dl.d2;
    const C2R_AS4 ShellStep &ss = shell_step(p, q);
    const double xx = (axis == 0) ? ss.d2axis[0] : xs * xs;
    const double yy = (axis == 1) ? ss.d2axis[1] : ys * ys;
    const double zz = (axis == 2) ? ss.d2axis[2] : zs * zs;
    const double dist2 = xx + yy + zz;
    cs.vol_ph = p.fourpi * dist2 * path;
    // LLS opacity (evolve_point.F90:186-196): homogeneous column, per-cell column (LLS_point), or a
    // hard barrier at R_max that only stops the transfer
    cs.cd_in = cdi;
    cs.stop_far = false;
    if (LLS == 3) {
        cs.stop_far = dist2 > p.R_max2;
    } else {
        const double lls = (LLS == 2) ? (double)(xf ? p.lls_T : p.lls)[cs.id] : step_of(p).coldensh_LLS;
        cs.cd_in = cdi + udiv(lls * path, step_of(p).dr[0], step_of(p).inv_dr0, step_of(p).exact_udiv_dr0);
    }
    cs.cd_out = cs.cd_in + cs.nhi * path;
    return cs;
}

// Returns the cell's photon-loss contribution (0 unless it lies on the sub-box surface).
template <bool DET, int LLS, bool STREAM, bool HEAT, bool STORE = true>
__device__ __forceinline__ double cell_commit(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                              const int face, const int s, const int a, const int b, const CellState &cs)
{
    double loss = 0.0;
    const int q = sa.q;
    const int axis = 2 - (face >> 1);
    const int pd = (face & 1) ? -q : q;
    const bool xf = (axis == 0);
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const double cd_out = cs.cd_out;
    // store into this face's plane and into the planes of the faces sharing the cell
    const __amdgpu_buffer_rsrc_t r_cur = make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_cur) * 6 * p.PP, 6u * plane_bytes);
    constexpr int SA = STREAM ? C2R_STORE_AUX : 0;
    if (STORE) {                 // (off for the first shell of a look-ahead pair: nothing reads its planes)
    buf_store_f64<SA>(r_cur, (unsigned)face * plane_bytes + cs.o8, cd_out);
    if (axis == 2) {
        if (abs(a) == q)   // x-face (u=y=b, v=z=pd)
            buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (b + p.R)) * 8u, cd_out);
        if (abs(b) == q)   // y-face (u=x=a, v=z=pd)
            buf_store_f64<SA>(r_cur, (b > 0 ? 2u : 3u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (a + p.R)) * 8u, cd_out);
    } else if (axis == 1) {
        if (abs(a) == q)   // x-face (u=y=pd, v=z=b)
            buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((b + p.R) * p.P + (pd + p.R)) * 8u, cd_out);
    }
    }
    const Delta3 dl = mesh_delta(axis, pd, a, b);          // recomputed, not carried in CellState
    if (sa.dbg_cdout) {                                     // single-source test path: the N^3 coldensh_out
        const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0), c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1),
                       c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
        sa.dbg_cdout[c0 + (unsigned)p.n[0] * (c1 + (unsigned)p.n[1] * c2)] = cd_out;
    }

    const double nflux = p.normflux[s];
    double gamma = 0.0, heat = 0.0;
    if (!cs.stop_far && !(cs.cd_in > p.max_coldensh) && nflux > 0.0) {
        double t_out;
        gamma = rates_fast<HEAT>(p, ltab, p.thick, cs.cd_in, cd_out, nflux, cs.vol_ph * cs.nhi, cs.vol_ph, t_out, heat);
        if (!DET && gamma != 0.0) atomicAdd(&(xf ? p.phih_T : p.phih)[cs.id], gamma);
        if (HEAT && !DET && heat != 0.0) atomicAdd(&(xf ? p.heat_T : p.heat)[cs.id], heat);      // evolve_point.F90:285-286
        if (sa.has_boundary) {
            const bool bnd = dl.d0 == sa.boxR[0] || dl.d1 == sa.boxR[1] || dl.d2 == sa.boxR[2] ||
                             dl.d0 == -sa.boxL[0] || dl.d1 == -sa.boxL[1] || dl.d2 == -sa.boxL[2];
            if (bnd) loss = fdiv((nflux * t_out) * step_of(p).vol, cs.vol_ph);
        }
    }
    // deterministic mode: every visited cell records its rate (zero included) for k_gamma_reduce
    if (DET) p.gbox[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + cs.id] = gamma;
    if (DET && HEAT) p.gbox_h[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + cs.id] = heat;
    return loss;
}

__device__ __forceinline__ double weight_rcp(const KParams &p, double c) { return frcp(fmax(p.wfloor, c * p.sigma)); }

// One cell (a,b): four upstream corners of plane q-1 (zero weight and value outside |.| <= q-1: an
// out-of-range offset reads 0), state, commit.  Used by the fused first-sub-box kernel.
template <bool DET, int LLS, int GLC, bool HEAT>
__device__ __forceinline__ double shell_cell(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                             const int face, const int s, const int a, const int b)
{
    const int q = sa.q, qm = q - 1;
    const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1;
    const int am = a - sga, bm = b - sgb;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev =
        make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_prev) * 6 * p.PP + (size_t)face * p.PP, plane_bytes);
    const bool ina = abs(a) <= qm, inam = abs(am) <= qm, inb = abs(b) <= qm, inbm = abs(bm) <= qm;
    const unsigned p8 = (unsigned)p.P * 8u;
    const unsigned o8 = plane_off8(p, a, b), da8 = (unsigned)(sga * 8), db8 = b < 0 ? 0u - p8 : p8;
    const double c1v = buf_load_f64<GLC>(r_prev, (inam && inbm) ? o8 - db8 - da8 : kOOB);
    const double c2v = buf_load_f64<GLC>(r_prev, (ina && inbm) ? o8 - db8 : kOOB);
    const double c3v = buf_load_f64<GLC>(r_prev, (inam && inb) ? o8 - da8 : kOOB);
    const double c4v = buf_load_f64<GLC>(r_prev, (ina && inb) ? o8 : kOOB);
    const CellState cs = cell_state<LLS, false>(p, sa, face, s, a, b, c1v, c2v, c3v, c4v, weight_rcp(p, c1v), weight_rcp(p, c2v),
                                         weight_rcp(p, c3v), weight_rcp(p, c4v));
    return cell_commit<DET, LLS, false, HEAT>(p, sa, ltab, face, s, a, b, cs);
}

// kRows cells of one column: (a,b0), (a,b0+sgb), ... with sgb the sign class of all their rows (rows are
// grouped outward from 0 within each sign class, see FaceRect).  Cell k+1's upstream row is cell k's own
// row, so the group needs 2(kRows+1) corners instead of 4 kRows -- and as many of the seven-instruction
// weight reciprocals -- and everything that depends on `a` alone (its sign, the u-interpolation factor,
// the wrap of that mesh axis) is computed once; the cells' dependency chains interleave in one thread and
// a wave's start-up is paid once for 64 kRows cells.  Per-cell arithmetic is exactly shell_cell's.
// Same-box A/B at 256^3 x 1000: 1 row 231 ms, 2 rows 206, 3 rows 194, 4 rows 203 (86 VGPRs, occupancy 5).
#ifndef C2R_ROWS
#define C2R_ROWS 3
#endif
constexpr int kRows = C2R_ROWS;
#ifndef C2R_PAIR_LDS_TABLE
#define C2R_PAIR_LDS_TABLE 0        // 1: the pair kernels fill the per-wave LDS table like the single-shell kernels (experiments)
#endif
constexpr int kPairRows = 1;        // rows per thread of the second shell of a look-ahead pair (k_sweep_pair, k_sweep_pair_fast)
template <bool DET, int LLS, bool STREAM, bool HEAT, bool STORE = true>
__device__ __forceinline__ double shell_rows(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                             const int face, const int s, const int a, const int b0, const int sgb,
                                             const int nvalid)
{
    const int q = sa.q, qm = q - 1;
    const int sga = a < 0 ? -1 : 1;
    const int am = a - sga;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev =
        make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_prev) * 6 * p.PP + (size_t)face * p.PP, plane_bytes);
    const bool ina = abs(a) <= qm, inam = abs(am) <= qm;
    const unsigned p8 = (unsigned)p.P * 8u;
    const unsigned da8 = (unsigned)(sga * 8), db8 = sgb < 0 ? 0u - p8 : p8;
    unsigned o = plane_off8(p, a, b0) - db8;                 // row b0 - sgb
    double vm[kRows + 1], va[kRows + 1], rm[kRows + 1], ra[kRows + 1];
#pragma unroll
    for (int r = 0; r <= kRows; ++r) {                       // rows b0-sgb, b0, ..., b0+(kRows-1)sgb
        const bool inr = abs(b0 + (r - 1) * sgb) <= qm;
        vm[r] = buf_load_f64<STREAM ? C2R_PLANE_AUX : 0>(r_prev, (inam && inr) ? o - da8 : kOOB);
        va[r] = buf_load_f64<STREAM ? C2R_PLANE_AUX : 0>(r_prev, (ina && inr) ? o : kOOB);
        o += db8;
    }
#pragma unroll
    for (int r = 0; r <= kRows; ++r) { rm[r] = weight_rcp(p, vm[r]); ra[r] = weight_rcp(p, va[r]); }
    const CellState c0 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0, vm[0], va[0], vm[1], va[1], rm[0], ra[0], rm[1], ra[1]);
    const CellState c1 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0 + sgb, vm[1], va[1], vm[2], va[2], rm[1], ra[1], rm[2], ra[2]);
#if C2R_ROWS >= 3
    const CellState c2 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0 + 2 * sgb, vm[2], va[2], vm[3], va[3], rm[2], ra[2], rm[3], ra[3]);
#endif
#if C2R_ROWS >= 4
    const CellState c3 = cell_state<LLS, STREAM>(p, sa, face, s, a, b0 + 3 * sgb, vm[3], va[3], vm[4], va[4], rm[3], ra[3], rm[4], ra[4]);
#endif
    double loss = cell_commit<DET, LLS, STREAM, HEAT, STORE>(p, sa, ltab, face, s, a, b0, c0);
    if (nvalid > 1) loss = loss + cell_commit<DET, LLS, STREAM, HEAT, STORE>(p, sa, ltab, face, s, a, b0 + sgb, c1);
#if C2R_ROWS >= 3
    if (nvalid > 2) loss = loss + cell_commit<DET, LLS, STREAM, HEAT, STORE>(p, sa, ltab, face, s, a, b0 + 2 * sgb, c2);
#endif
#if C2R_ROWS >= 4
    if (nvalid > 3) loss = loss + cell_commit<DET, LLS, STREAM, HEAT, STORE>(p, sa, ltab, face, s, a, b0 + 3 * sgb, c3);
#endif
    return loss;
}

// Look-ahead of the exact mode (k_sweep_pair; see lookahead_cd_out of the fast mode below for the idea): the column density that
// shell sq.q leaves in plane `face` at (a, b), recomputed from the planes of shell sq.q - 1 with cell_state -- the arithmetic of
// the launch that owns the cell, in the owner's geometry.  Straight-line code: a thread's four recomputes share their waits.
template <int LLS, bool STREAM>
__device__ __forceinline__ double lookahead_cd_out_exact(const KParams &p, const ShellArgs &sq, const int face, const int s,
                                                         const int a, const int b)
{
    const int q = sq.q, qm = q - 1;
    const bool inside = abs(a) <= q && abs(b) <= q;           // else a zero-weight corner (an OOB load reads 0)
    const int axis = 2 - (face >> 1);
    const int pd = (face & 1) ? -q : q;
    int fo = face, ao = a, bo = b;                            // the owner of the cell and its coordinates there
    if (axis == 1) { if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = a; bo = pd; } }
    else if (axis == 0) {
        if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = pd; bo = a; }
        else if (abs(a) == q) { fo = a > 0 ? 2 : 3; ao = pd; bo = b; }
    }
    const int sga = ao < 0 ? -1 : 1, sgb = bo < 0 ? -1 : 1;
    const int am = ao - sga, bm = bo - sgb;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev = make_rsrc(p.planes + ((size_t)s * 2 + sq.buf_prev) * 6 * p.PP, 6u * plane_bytes);
    const bool ina = abs(ao) <= qm, inam = abs(am) <= qm, inb = abs(bo) <= qm, inbm = abs(bm) <= qm;
    const unsigned base = (unsigned)fo * plane_bytes;
    constexpr int PA = STREAM ? C2R_PLANE_AUX : 0;
    const double c1v = buf_load_f64<PA>(r_prev, (inam && inbm) ? base + plane_off8(p, am, bm) : kOOB);
    const double c2v = buf_load_f64<PA>(r_prev, (ina && inbm) ? base + plane_off8(p, ao, bm) : kOOB);
    const double c3v = buf_load_f64<PA>(r_prev, (inam && inb) ? base + plane_off8(p, am, bo) : kOOB);
    const double c4v = buf_load_f64<PA>(r_prev, (ina && inb) ? base + plane_off8(p, ao, bo) : kOOB);
    const CellState cs = cell_state<LLS, STREAM, true>(p, sq, fo, s, ao, bo, c1v, c2v, c3v, c4v, weight_rcp(p, c1v),
                                                       weight_rcp(p, c2v), weight_rcp(p, c3v), weight_rcp(p, c4v));
    return inside ? cs.cd_out : 0.0;
}
// one cell of the second shell of a pair: its four upstream corners recomputed
template <bool DET, int LLS, bool STREAM, bool HEAT>
__device__ __forceinline__ double shell_cell_look(const KParams &p, const ShellArgs &sa, const ShellArgs &sq,
                                                  const v2f64 *__restrict__ ltab, const int face, const int s, const int a, const int b)
{
    const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1;
    const int am = a - sga, bm = b - sgb;
    const double c1v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, am, bm);
    const double c2v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, a, bm);
    const double c3v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, am, b);
    const double c4v = lookahead_cd_out_exact<LLS, STREAM>(p, sq, face, s, a, b);
    const CellState cs = cell_state<LLS, STREAM>(p, sa, face, s, a, b, c1v, c2v, c3v, c4v, weight_rcp(p, c1v), weight_rcp(p, c2v),
                                                 weight_rcp(p, c3v), weight_rcp(p, c4v));
    return cell_commit<DET, LLS, STREAM, HEAT>(p, sa, ltab, face, s, a, b, cs);
}

// STREAM: the non-temporal cache policy above (the host turns it on for meshes whose n_HI array outgrows the L2s)
// LOOK (sq = the previous shell's arguments): one row per thread (FaceRect built for kPairRows), corners recomputed
template <bool DET, int LLS, bool STREAM, bool HEAT, int LOOK = 0, bool STORE = true>
__device__ __forceinline__ void sweep_tile(const KParams &p, const ShellArgs &sa, const FaceRect &fr, const v2f64 *ltab,
                                           double *sm, const int face, const int tile, const int sl, const ShellArgs &sq)
{
    double loss = 0.0;
    const unsigned t = (unsigned)tile * kBlock + threadIdx.x;
    const unsigned bi = fr.magic ? __umulhi(t, fr.magic) : t;       // row-group index
    if (tile < fr.ntiles && bi < (unsigned)fr.npr) {
        const int a = fr.a_lo + (int)(t - __umul24(bi, (unsigned)fr.wa));
        // rows are grouped outward from 0 within each sign class: (0..kRows-1), ... then (-1..-kRows), ...
        const bool pos = bi < (unsigned)fr.pp;
        constexpr int NR = LOOK ? kPairRows : kRows;
        const int k0 = NR * (pos ? (int)bi : (int)bi - fr.pp);
        const int sgb = pos ? 1 : -1;
        const int b0 = pos ? k0 : -1 - k0;
        const int left = pos ? (fr.b_lo + fr.wb - b0) : (b0 - fr.b_lo + 1);     // rows from b0 to the end of the class
        if (LOOK) loss = shell_cell_look<DET, LLS, STREAM, HEAT>(p, sa, sq, ltab, face, sa.active[sl], a, b0);
        else loss = shell_rows<DET, LLS, STREAM, HEAT, STORE>(p, sa, ltab, face, sa.active[sl], a, b0, sgb, min(left, kRows));
    }
    if (sa.has_boundary) {
        const double tot = block_sum_256(loss, sm);
        if (threadIdx.x == 0)
            sa.loss_partial[((size_t)sl * 6 + face) * sa.tiles_max + tile] = tot;
    }
}

template <bool DET, int LLS, bool STREAM, bool HEAT>
__global__ __launch_bounds__(kBlock) void k_sweep_shell(KParams p, ShellArgs sa)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];                      // kBlock/64 waves x 64 entries
    const int face = blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;         // block-uniform: this source retired after the launch was sized
    const FaceRect fr = sa.face[face];
    if (tile >= fr.ntiles && !sa.has_boundary) return;   // block-uniform: nothing to do and no partial to write
    const v2f64 *ltab = wave_log_table(p.odtab, s_log);      // (table positions: rates_fast)
    sweep_tile<DET, LLS, STREAM, HEAT>(p, sa, fr, ltab, sm, face, tile, (int)blockIdx.z, sa);
}

// The look-ahead pair of the exact mode (see k_sweep_pair_fast below): shell sa.q in blockIdx.y 0..5 (planes not stored),
// shell sb.q = sa.q + 1 in 6..11 (corners recomputed, one row per thread).
template <bool DET, int LLS, bool STREAM, bool HEAT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 4)))
void k_sweep_pair(KParams p, ShellArgs sa, ShellArgs sb)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];
    const bool second = blockIdx.y >= 6;
    const int face = second ? (int)blockIdx.y - 6 : (int)blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;
    const v2f64 *ltab = (C2R_PAIR_LDS_TABLE) ? wave_log_table(p.odtab, s_log) : p.odtab;     // see k_sweep_pair_fast
    if (second) {
        const FaceRect fr = sb.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile<DET, LLS, STREAM, HEAT, 1, true>(p, sb, fr, ltab, sm, face, tile, (int)blockIdx.z, sa);
    } else {
        const FaceRect fr = sa.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile<DET, LLS, STREAM, HEAT, 0, false>(p, sa, fr, ltab, sm, face, tile, (int)blockIdx.z, sa);
    }
}

// ==== tolerance ("fast") mode of the sweep =======================================================
// Same physics, same schedule, same integer decisions; the f64 arithmetic of one (cell, source) is
// re-associated where algebra allows, giving up bit-identity of the column densities with the
// Fortran (c2r_params.sweep_mode = 1; the exact kernel above stays the default).  Against the exact
// kernel: ~1e-13 relative on column densities and rates (stated and tested in tests/: integers exact,
// xh 1e-9).  What changes, with the reference lines each form restates:
//  * cinterp weights (column_density.f90:112-140): dx = 2|xc-(im+sgn/2)| is 1-|a|/q identically, so the
//    four weights factor into column and row parts; with t = c/max(0.6,c sigma), r = 1/max(0.6,c sigma)
//    per corner, cdensi = [omv(omu t1+ddu t2)+ddv(omu t3+ddu t4)] / [same in r]: the column sums of a row
//    are shared by the two cells above and below it (3 rows per thread: 4 rows of sums for 3 cells);
//  * reciprocals by v_rcp_f64 + one Newton step (2^-48) instead of the correctly rounded quotient;
//  * path = sqrt(q^2+a^2+b^2) dr/q from the exact integer (column_density.f90:168), dist2 by two FMAs
//    (evolve_point.F90:171-174), Gamma = (T_in-T_out) NormFlux / (4 pi dist2 n_HI path) with one
//    reciprocal (radiation_photoionrates.F90:262-263, evolve_point.F90:262);
//  * the table position 1+(log10 tau-minlogtau)/dlogtau (radiation_photoionrates.F90:195-199) comes
//    straight out of the log evaluation: the LDS table holds positions instead of logarithms.
__device__ __forceinline__ double sqrt_pos(double x)    // sqrt(x), x >= 1 normal: Goldschmidt step + one correction (~1 ulp)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = 0.5 * y;
    g = __builtin_fma(g, __builtin_fma(-h, g, 0.5), g);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
// Row sums over the two upstream columns am, a of one upstream row: R = omu r(am) + ddu r(a), T likewise with t = c r
// (r = 1/max(0.6, c sigma), weightf of column_density.f90:276-293)
__device__ __forceinline__ void row_sum_fast(const KParams &p, const double omu, const double ddu, const double vm, const double va,
                                             double &R, double &T)
{
    const double rm = rcp1(fmax(p.wfloor, vm * p.sigma)), ra = rcp1(fmax(p.wfloor, va * p.sigma));
    R = __builtin_fma(omu, rm, ddu * ra);
    T = __builtin_fma(omu, vm * rm, ddu * (va * ra));
}
// cinterp + path + the cell's own column for cell (a, b) of a face in shell sa.q, from the row sums of its upstream rows
// b - sgb (lo) and b (hi).  ONE piece of arithmetic for the shell kernel and for the look-ahead recompute of the same cell
// (lookahead_cd_out): both must produce the same bits.
struct CellCd { double cd_in, cd_out, pq, path, np; };
template <int LLS>
__device__ __forceinline__ CellCd cell_cd_fast(const KParams &p, const ShellArgs &sa, const int a, const int a2, const int b,
                                               const double Rlo, const double Tlo, const double Rhi, const double Thi,
                                               const double nhi, const double lls_cell /* LLS == 2: the cell's LLS_grid value */)
{
    CellCd c;
    const int q = sa.q;
    const double omv = (double)abs(b) * sa.inv_q, ddv = 1.0 - omv;      // weights of rows b-sgb and b
    const double den = __builtin_fma(omv, Rlo, ddv * Rhi);
    const double num = __builtin_fma(omv, Tlo, ddv * Thi);
    double cdi = num * rcp1(den);
    if (q == 1 && (abs(a) == 1 || abs(b) == 1))
        cdi = (abs(a) == 1 && abs(b) == 1) ? p.sqrt3 * cdi : p.sqrt2 * cdi;
    c.pq = sqrt_pos((double)(q * q + a2 + b * b));                      // |delta| in cells
    const C2R_AS4 ShellStep &ss = shell_step(p, q);
    c.path = c.pq * ss.path_scale;
    if (LLS == 3) c.cd_in = cdi;
    else if (LLS == 2) c.cd_in = __builtin_fma(lls_cell * sa.inv_q, c.pq, cdi);
    else c.cd_in = __builtin_fma(ss.lls_scale, c.pq, cdi);
    c.np = nhi * c.path;                                                // n_HI path: the cell's own column
    c.cd_out = c.cd_in + c.np;
    return c;
}

// Look-ahead (few sources, k_sweep_pair_fast): the column density that shell sq.q leaves in plane `face` at plane
// coordinates (a, b), recomputed from the planes of shell sq.q - 1 instead of read -- so that shell q+1 can run in the
// SAME launch as shell q (both depend on shell q-1 only): with few sources the GPU is idle and a launch is pure
// latency, the redundant arithmetic is free and the chain of dependent launches is halved.  A plane also holds the
// cells that a neighbouring face owns (the edge rows / columns the owner stores across, shell_rows_fast_core): those
// are recomputed in the owner's geometry.  Same arithmetic as the owner's own launch (row_sum_fast, cell_cd_fast).
template <int LLS, bool STREAM>
__device__ __forceinline__ double lookahead_cd_out(const KParams &p, const ShellArgs &sq, const int face, const int s,
                                                   const int a, const int b)
{
    const int q = sq.q, qm = q - 1;
    // straight-line code (no early exit): a thread recomputes 8 such cells and all their loads should be in flight together
    const bool inside = abs(a) <= q && abs(b) <= q;           // else: outside shell q's plane, a zero-weight corner (an OOB load reads 0)
    const int axis = 2 - (face >> 1);
    const int pd = (face & 1) ? -q : q;
    // the owner of the cell and the cell's coordinates on the owner's face
    int fo = face, ao = a, bo = b;
    if (axis == 1) { if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = a; bo = pd; } }
    else if (axis == 0) {
        if (abs(b) == q) { fo = b > 0 ? 0 : 1; ao = pd; bo = a; }
        else if (abs(a) == q) { fo = a > 0 ? 2 : 3; ao = pd; bo = b; }
    }
    const int sga = ao < 0 ? -1 : 1, sgb = bo < 0 ? -1 : 1;
    const int am = ao - sga, bm = bo - sgb;
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const __amdgpu_buffer_rsrc_t r_prev = make_rsrc(p.planes + ((size_t)s * 2 + sq.buf_prev) * 6 * p.PP, 6u * plane_bytes);
    const bool ina = abs(ao) <= qm, inam = abs(am) <= qm, inb = abs(bo) <= qm, inbm = abs(bm) <= qm;
    const unsigned base = (unsigned)fo * plane_bytes;
    constexpr int PA = STREAM ? C2R_PLANE_AUX : 0;
    const double v_mm = buf_load_f64<PA>(r_prev, (inam && inbm) ? base + plane_off8(p, am, bm) : kOOB);
    const double v_am = buf_load_f64<PA>(r_prev, (ina && inbm) ? base + plane_off8(p, ao, bm) : kOOB);
    const double v_mb = buf_load_f64<PA>(r_prev, (inam && inb) ? base + plane_off8(p, am, bo) : kOOB);
    const double v_ab = buf_load_f64<PA>(r_prev, (ina && inb) ? base + plane_off8(p, ao, bo) : kOOB);
    // the cell in the mesh: from its position on the plane it was asked for
    const Delta3 dl = mesh_delta(axis, pd, a, b);
    const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0), c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1),
                   c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
    const unsigned id = c0 + (unsigned)p.n[0] * (c1 + (unsigned)p.n[1] * c2);
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const double nhi = buf_load_f64<STREAM ? C2R_NHI_AUX : 0>(make_rsrc(p.nhi, ncell * 8u), id * 8u);
    const double omu = (double)abs(ao) * sq.inv_q, ddu = 1.0 - omu;
    double Rlo, Tlo, Rhi, Thi;
    row_sum_fast(p, omu, ddu, v_mm, v_am, Rlo, Tlo);
    row_sum_fast(p, omu, ddu, v_mb, v_ab, Rhi, Thi);
    const double cd_out = cell_cd_fast<LLS>(p, sq, ao, ao * ao, bo, Rlo, Tlo, Rhi, Thi, nhi, LLS == 2 ? (double)p.lls[id] : 0.0).cd_out;
    return inside ? cd_out : 0.0;
}

// Everything a thread does for its NR rows once the upstream values are known: vm[r], va_[r] = the previous shell's
// column densities at columns am = a - sga and a of rows b0 - sgb, b0, ..., b0 + (NR-1) sgb.
// STORE: write the column densities into the planes (off for the first shell of a look-ahead pair: nothing reads them)
template <bool DET, int LLS, bool STREAM, int NR, bool HEAT, bool STORE = true>
__device__ __forceinline__ double shell_rows_fast_core(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                                       const double *__restrict__ thick, const int face, const int s, const int a,
                                                       const int b0, const int sgb, const int nvalid,
                                                       const double (&vm)[NR + 1], const double (&va_)[NR + 1])
{
    const int q = sa.q;
    const int axis = 2 - (face >> 1);            // 2:z 1:y 0:x  (block-uniform)
    const int pd = (face & 1) ? -q : q;
    const bool xf = (axis == 0);
    const int ua = xf ? 1 : 0, va = (axis == 2) ? 1 : 2;      // mesh axes of the plane coordinates (a, b)
    const unsigned plane_bytes = (unsigned)p.PP * 8u;
    const unsigned p8 = (unsigned)p.P * 8u;
    const unsigned db8 = sgb < 0 ? 0u - p8 : p8;
    const unsigned o_first = plane_off8(p, a, b0);
    // cell indices: id = ca + base_p + stride_b * cb (x-fastest array, or the (x,y)-transposed replica for x faces)
    const unsigned na = (unsigned)p.n[ua], nmid = xf ? (unsigned)p.n[0] : (unsigned)p.n[1];
    const unsigned ca = wrap_pos(p.srcw[3 * s + ua], p.n[ua], a);
    const unsigned cp = wrap_pos(p.srcw[3 * s + axis], p.n[axis], pd);          // block-uniform
    const unsigned stride_b = (axis == 2) ? na : na * nmid;
    const unsigned base_p = (axis == 2) ? na * nmid * cp : na * cp;
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(xf ? p.nhi_T : p.nhi, ncell * 8u);
    unsigned id[NR];
    double nhi[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const unsigned cb = wrap_pos(p.srcw[3 * s + va], p.n[va], b0 + k * sgb);
        id[k] = ca + base_p + __umul24(stride_b, cb);
        nhi[k] = buf_load_f64<STREAM ? C2R_NHI_AUX : 0>(r_x, id[k] * 8u);
    }
    // column part of the interpolation and of the geometry
    const double omu = (double)abs(a) * sa.inv_q, ddu = 1.0 - omu;   // weights of columns am and a
    const int a2 = a * a;
    const double du2 = step_of(p).dr2[ua] * (double)a2;
    // row sums over the two columns
    double R[NR + 1], T[NR + 1];
#pragma unroll
    for (int r = 0; r <= NR; ++r) row_sum_fast(p, omu, ddu, vm[r], va_[r], R[r], T[r]);
    const double nflux = p.normflux[s];
    const __amdgpu_buffer_rsrc_t r_cur = make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_cur) * 6 * p.PP, 6u * plane_bytes);
    constexpr int SA = STREAM ? C2R_STORE_AUX : 0;
    const bool bnd_col = sa.has_boundary && (a == sa.boxR[ua] || a == -sa.boxL[ua] || pd == sa.boxR[axis] || pd == -sa.boxL[axis]);
    double loss = 0.0;
    unsigned o8 = o_first;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int b = b0 + k * sgb;
        if (k < nvalid) {
            const CellCd cc = cell_cd_fast<LLS>(p, sa, a, a2, b, R[k], T[k], R[k + 1], T[k + 1], nhi[k],
                                                LLS == 2 ? (double)(xf ? p.lls_T : p.lls)[id[k]] : 0.0);
            const double path = cc.path, cd_in = cc.cd_in, np = cc.np, cd_out = cc.cd_out;
            const double dist2 = __builtin_fma(step_of(p).dr2[va], (double)(b * b), du2 + shell_step(p, q).d2axis[axis]);
            const bool stop = (LLS == 3) && dist2 > p.R_max2;
            // the cell's column density, also into the planes of the faces sharing the cell
            if (STORE) {
            buf_store_f64<SA>(r_cur, (unsigned)face * plane_bytes + o8, cd_out);
            if (axis == 2) {
                if (abs(a) == q)
                    buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (b + p.R)) * 8u, cd_out);
                if (abs(b) == q)
                    buf_store_f64<SA>(r_cur, (b > 0 ? 2u : 3u) * plane_bytes + (unsigned)((pd + p.R) * p.P + (a + p.R)) * 8u, cd_out);
            } else if (axis == 1) {
                if (abs(a) == q)
                    buf_store_f64<SA>(r_cur, (a > 0 ? 4u : 5u) * plane_bytes + (unsigned)((b + p.R) * p.P + (pd + p.R)) * 8u, cd_out);
            }
            }
            if (sa.dbg_cdout) {
                const Delta3 dl = mesh_delta(axis, pd, a, b);
                const unsigned c0 = wrap_pos(p.srcw[3 * s + 0], p.n[0], dl.d0), c1 = wrap_pos(p.srcw[3 * s + 1], p.n[1], dl.d1),
                               c2 = wrap_pos(p.srcw[3 * s + 2], p.n[2], dl.d2);
                sa.dbg_cdout[c0 + (unsigned)p.n[0] * (c1 + (unsigned)p.n[1] * c2)] = cd_out;
            }
            double gamma = 0.0, heat = 0.0;
            if (!stop && !(cd_in > p.max_coldensh) && nflux > 0.0) {
                const double area = p.fourpi * dist2;                             // vol_ph = area path
                double t_out;
                gamma = rates_fast<HEAT>(p, ltab, thick, cd_in, cd_out, nflux, area * np, area * path, t_out, heat);
                if (!DET && gamma != 0.0) atomicAdd(&(xf ? p.phih_T : p.phih)[id[k]], gamma);
                if (HEAT && !DET && heat != 0.0) atomicAdd(&(xf ? p.heat_T : p.heat)[id[k]], heat);
                if (sa.has_boundary && (bnd_col || b == sa.boxR[va] || b == -sa.boxL[va]))
                    loss = loss + fdiv((nflux * t_out) * step_of(p).vol, area * path);
            }
            if (DET) p.gbox[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + id[k]] = gamma;
            if (DET && HEAT) p.gbox_h[((size_t)s * 2 + (xf ? 1 : 0)) * ncell + id[k]] = heat;
        }
        o8 += db8;
    }
    return loss;
}

// LOOK = 0: the upstream values are read from the previous shell's planes (sq unused: pass sa).  LOOK = 1 (sq = the previous
// shell's arguments, by reference -- a pointer to a kernel argument would force it into scratch): they are recomputed from
// the planes of the shell before it (lookahead_cd_out).
template <bool DET, int LLS, bool STREAM, int NR, bool HEAT, int LOOK = 0, bool STORE = true>
__device__ __forceinline__ double shell_rows_fast(const KParams &p, const ShellArgs &sa, const v2f64 *__restrict__ ltab,
                                                  const double *__restrict__ thick,   // p.thick, or the block's LDS copy of it
                                                  const int face, const int s, const int a, const int b0, const int sgb,
                                                  const int nvalid, const ShellArgs &sq)
{
    const int q = sa.q, qm = q - 1;
    const int sga = a < 0 ? -1 : 1;
    const int am = a - sga;
    double vm[NR + 1], va_[NR + 1];
    if (LOOK) {
#pragma unroll
        for (int r = 0; r <= NR; ++r) {                        // rows b0-sgb, b0, ..., b0+(NR-1)sgb
            // (rows beyond the thread's valid cells are computed too -- from periodic-wrapped, in-range addresses -- and unused)
            const int row = b0 + (r - 1) * sgb;
            vm[r] = lookahead_cd_out<LLS, STREAM>(p, sq, face, s, am, row);
            va_[r] = lookahead_cd_out<LLS, STREAM>(p, sq, face, s, a, row);
        }
    } else {
        const unsigned plane_bytes = (unsigned)p.PP * 8u;
        const __amdgpu_buffer_rsrc_t r_prev =
            make_rsrc(p.planes + ((size_t)s * 2 + sa.buf_prev) * 6 * p.PP + (size_t)face * p.PP, plane_bytes);
        const bool ina = abs(a) <= qm, inam = abs(am) <= qm;
        const unsigned p8 = (unsigned)p.P * 8u;
        const unsigned da8 = (unsigned)(sga * 8), db8 = sgb < 0 ? 0u - p8 : p8;
        unsigned o = plane_off8(p, a, b0) - db8;                   // row b0 - sgb
#pragma unroll
        for (int r = 0; r <= NR; ++r) {                        // rows b0-sgb, b0, ..., b0+(NR-1)sgb
            const bool inr = abs(b0 + (r - 1) * sgb) <= qm;
            vm[r] = buf_load_f64<STREAM ? C2R_PLANE_AUX : 0>(r_prev, (inam && inr) ? o - da8 : kOOB);
            va_[r] = buf_load_f64<STREAM ? C2R_PLANE_AUX : 0>(r_prev, (ina && inr) ? o : kOOB);
            o += db8;
        }
    }
    return shell_rows_fast_core<DET, LLS, STREAM, NR, HEAT, STORE>(p, sa, ltab, thick, face, s, a, b0, sgb, nvalid, vm, va_);
}

#ifndef C2R_FAST_ATTR
#define C2R_FAST_ATTR __launch_bounds__(kBlock)
#endif
// one (source, face, tile) block of work of k_sweep_shell_fast; sl = position in the active list
template <bool DET, int LLS, bool STREAM, bool HEAT, int LOOK = 0, bool STORE = true, int NR = kRows>
__device__ __forceinline__ void sweep_tile_fast(const KParams &p, const ShellArgs &sa, const FaceRect &fr, const v2f64 *ltab,
                                                const double *thick, double *sm, const int face, const int tile, const int sl,
                                                const ShellArgs &sq)
{
    double loss = 0.0;
    const unsigned t = (unsigned)tile * kBlock + threadIdx.x;
    const unsigned bi = fr.magic ? __umulhi(t, fr.magic) : t;
    if (tile < fr.ntiles && bi < (unsigned)fr.npr) {
        const int a = fr.a_lo + (int)(t - __umul24(bi, (unsigned)fr.wa));
        const bool pos = bi < (unsigned)fr.pp;
        const int k0 = NR * (pos ? (int)bi : (int)bi - fr.pp);
        const int sgb = pos ? 1 : -1;
        const int b0 = pos ? k0 : -1 - k0;
        const int left = pos ? (fr.b_lo + fr.wb - b0) : (b0 - fr.b_lo + 1);
        loss = shell_rows_fast<DET, LLS, STREAM, NR, HEAT, LOOK, STORE>(p, sa, ltab, thick, face, sa.active[sl], a, b0, sgb,
                                                                        min(left, NR), sq);
    }
    if (sa.has_boundary) {
        const double tot = block_sum_256(loss, sm);
        if (threadIdx.x == 0)
            sa.loss_partial[((size_t)sl * 6 + face) * sa.tiles_max + tile] = tot;
    }
}

template <bool DET, int LLS, bool STREAM, bool HEAT>
__global__ C2R_FAST_ATTR void k_sweep_shell_fast(KParams p, ShellArgs sa)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];                      // kBlock/64 waves x 64 entries
    const int face = blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;
    const FaceRect fr = sa.face[face];
    if (tile >= fr.ntiles && !sa.has_boundary) return;
    const v2f64 *ltab = wave_log_table(p.odtab, s_log);
    sweep_tile_fast<DET, LLS, STREAM, HEAT>(p, sa, fr, ltab, p.thick, sm, face, tile, (int)blockIdx.z, sa);
}

// Look-ahead pair (few sources, no cell of either shell on the sub-box surface): shell sa.q (blockIdx.y 0..5) and shell
// sb.q = sa.q + 1 (blockIdx.y 6..11) in ONE launch, both from the planes of shell sa.q - 1 -- the second shell recomputes
// the first shell's column densities where it needs them (lookahead_cd_out).  The first shell's planes are not stored
// (nobody reads them); the second shell's go to the other plane set, which the next launch reads.
template <bool DET, int LLS, bool STREAM, bool HEAT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 4)))     // latency-bound by design: registers before occupancy
void k_sweep_pair_fast(KParams p, ShellArgs sa, ShellArgs sb)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[kBlock];                      // kBlock/64 waves x 64 entries
    const bool second = blockIdx.y >= 6;
    const int face = second ? (int)blockIdx.y - 6 : (int)blockIdx.y;
    const int tile = blockIdx.x;
    const int nact = *sa.n_active;
    if ((int)blockIdx.z >= nact) return;
    // the position table straight from global memory (it is a few L1 lines): filling the per-wave LDS copy is a memory round
    // trip in front of everything else, and a launch of this kernel is made of round trips
    const v2f64 *ltab = (C2R_PAIR_LDS_TABLE) ? wave_log_table(p.odtab, s_log) : p.odtab;
    if (second) {
        const FaceRect fr = sb.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile_fast<DET, LLS, STREAM, HEAT, 1, true, kPairRows>(p, sb, fr, ltab, p.thick, sm, face, tile, (int)blockIdx.z, sa);
    } else {
        const FaceRect fr = sa.face[face];
        if (tile >= fr.ntiles) return;
        sweep_tile_fast<DET, LLS, STREAM, HEAT, 0, false>(p, sa, fr, ltab, p.thick, sm, face, tile, (int)blockIdx.z, sa);
    }
}

// ---- the first sub-boxes, fused: one workgroup per source, all shells of the sub-box in one launch -----
// Near the source a shell has few cells (24q^2+2: 26 ... 602 for q = 1..5) and a launch per shell is
// nothing but latency, with the six faces' 256-thread tiles mostly empty.  Here the cells of a shell are
// packed over the six faces (face_off = prefix sums of the owned rectangles) and a source's workgroup walks
// the shells itself, a barrier between them; its photon loss through the box surface is summed in a fixed
// order and added to loss_acc[source] (no k_loss_reduce).  Same per-cell code as k_sweep_shell.
constexpr int kMaxFused = 5;
struct BoxArgs {
    int source_cell;                 // 1: the workgroup first does its source's own cell (sub-box 1; else k_source_cells has)
    int nshell;
    int ncell[kMaxFused];            // packed cells of each shell
    int face_off[kMaxFused][8];      // [f]: first packed index of face f; [6] = ncell
    ShellArgs sh[kMaxFused];
    const int *active;
    const int *n_active;
    double *loss_acc;
};

template <bool DET, int LLS, bool FAST, bool HEAT>
__global__ __launch_bounds__(1024) void k_sweep_box_fused(KParams p, BoxArgs ba)
{
    __shared__ double sm[16];
    __shared__ v2f64 s_log[1024];
    const int sl = blockIdx.x;
    if (sl >= *ba.n_active) return;
    const int s = ba.active[sl];
    const v2f64 *ltab = wave_log_table(p.odtab, s_log);       // both modes: the table positions of rates_fast
    if (ba.source_cell) {
        if (threadIdx.x == 0) {
            const ShellArgs &s0 = ba.sh[0];
            source_cell<HEAT>(p, p.logtab, s, s0.boxR[0] == 0 || s0.boxR[1] == 0 || s0.boxR[2] == 0 || s0.boxL[0] == 0 ||
                              s0.boxL[1] == 0 || s0.boxL[2] == 0, ba.loss_acc, s0.dbg_cdout);
        }
        __syncthreads();             // shell 0's plane entries are visible to the waves that read them in shell 1
    }
    for (int k = 0; k < ba.nshell; ++k) {
        const ShellArgs &sa = ba.sh[k];
        double loss = 0.0;
        for (int t = threadIdx.x; t < ba.ncell[k]; t += blockDim.x) {
            int f = 0;
#pragma unroll
            for (int g = 1; g < 6; ++g) f += (t >= ba.face_off[k][g]) ? 1 : 0;
            const FaceRect fr = sa.face[f];
            const unsigned lt = (unsigned)(t - ba.face_off[k][f]);
            const unsigned bi = fr.magic ? __umulhi(lt, fr.magic) : lt;
            const int a = fr.a_lo + (int)(lt - __umul24(bi, (unsigned)fr.wa));
            const int b = fr.b_lo + (int)bi;
            if (FAST) loss = loss + shell_rows_fast<DET, LLS, false, 1, HEAT>(p, sa, ltab, p.thick, f, s, a, b, b < 0 ? -1 : 1, 1, sa);
            else loss = loss + shell_cell<DET, LLS, 0, HEAT>(p, sa, ltab, f, s, a, b);
        }
        if (sa.has_boundary) {
            const double tot = block_sum_256(loss, sm);       // contains a barrier
            if (threadIdx.x == 0) ba.loss_acc[s] += tot;      // what k_loss_reduce does for the per-shell launches
        }
        __syncthreads();        // workgroup-scope release/acquire: this shell's planes are visible to every wave of the group
    }
}

// Deterministic mode: phih(cell) += Gamma_s(cell) for the sources of a batch IN SOURCE ORDER (the order
// of the serial reference, evolve_point.F90:283 inside master_slave.F90:85's loop).  One thread per
// cell; a source contributes where the cell lies inside its final sub-box (evolve_source.F90:135-136).
__global__ __launch_bounds__(256) void k_gamma_reduce(KParams p, int nsrc, const int *final_nbox, int subbox,
                                                      double *phih, double *heat /* phiheat_grid, or null */,
                                                      const int *gate = nullptr /* see k_transpose_xy */)
{
    if (gate && *gate != 0) return;
    // block = 256 cells along x of one (y,z) row: the y and z parts of the box test are block-uniform
    const int c0 = blockIdx.x * 256 + threadIdx.x, c1 = blockIdx.y, c2 = blockIdx.z;
    const bool live = c0 < p.n[0];
    const unsigned ncell = (unsigned)p.n[0] * (unsigned)p.n[1] * (unsigned)p.n[2];
    const unsigned id = (unsigned)c0 + (unsigned)p.n[0] * ((unsigned)c1 + (unsigned)p.n[1] * (unsigned)c2);
    const unsigned id_t = (unsigned)c1 + (unsigned)p.n[1] * ((unsigned)c0 + (unsigned)p.n[0] * (unsigned)c2);
    double acc = live ? phih[id] : 0.0;
    double acc_h = (live && heat) ? heat[id] : 0.0;
    for (int s = 0; s < nsrc; ++s) {
        const int nb = final_nbox[s];                 // uniform
        if (nb <= 0) continue;
        const int ext = subbox * nb;
        int d1 = c1 - p.srcw[3 * s + 1], d2 = c2 - p.srcw[3 * s + 2];
        d1 -= (d1 > p.hr[1]) ? p.n[1] : 0;  d1 += (d1 < -p.hl[1]) ? p.n[1] : 0;
        d2 -= (d2 > p.hr[2]) ? p.n[2] : 0;  d2 += (d2 < -p.hl[2]) ? p.n[2] : 0;
        if (d1 < -min(ext, p.hl[1]) || d1 > min(ext, p.hr[1]) || d2 < -min(ext, p.hl[2]) || d2 > min(ext, p.hr[2]))
            continue;                                 // the whole row lies outside this source's box
        const int m12 = max(abs(d1), abs(d2));
        int d0 = c0 - p.srcw[3 * s + 0];
        d0 -= (d0 > p.hr[0]) ? p.n[0] : 0;  d0 += (d0 < -p.hl[0]) ? p.n[0] : 0;
        if (live && d0 >= -min(ext, p.hl[0]) && d0 <= min(ext, p.hr[0])) {
            const bool xf = abs(d0) > m12;            // cinterp branch priority z > y > x
            const double *g = p.gbox + ((size_t)s * 2 + (xf ? 1 : 0)) * ncell;
            acc = acc + g[xf ? id_t : id];
            if (heat) acc_h = acc_h + (p.gbox_h + ((size_t)s * 2 + (xf ? 1 : 0)) * ncell)[xf ? id_t : id];
        }
    }
    if (live) phih[id] = acc;
    if (live && heat) heat[id] = acc_h;
}

// nhi[i,j,k] = max(1-max(xh_av,eps),eps) * ndens (ion%h_av(0)*ndens_p of evolve0D) and its (x,y)-transposed
// replica, one z-plane tile at a time through LDS.
// zero (null or 4 pointers): arrays of the mesh's size to clear on the way -- phih_grid and phiheat_grid (x fastest) in [0], [2],
// their (x,y)-transposed accumulators in [1], [3] (any of them null) -- where launches are what an iteration costs (c2r_iterate)
struct ZeroGrids { double *g[4]; };
// copy (or null pointers): n 32-bit words that block (0,0,0) copies on the way -- the pristine image of a small batch's state
// block (active lists, counters, accumulators) over the working one, instead of a host-to-device copy node in the graph
struct WordCopy { const unsigned *src; unsigned *dst; unsigned n; };
__global__ __launch_bounds__(256) void k_prepare_nhi(int n0, int n1, int n2, double eps, const float *__restrict__ ndens,
                                                     const double *__restrict__ xh_av, double *__restrict__ nhi,
                                                     double *__restrict__ nhi_T, ZeroGrids zero, WordCopy copy)
{
    __shared__ double tile[32][33];
    const int k = blockIdx.z;
    if (copy.n && blockIdx.x == 0 && blockIdx.y == 0 && k == 0)
        for (unsigned w = threadIdx.x; w < copy.n; w += 256) copy.dst[w] = copy.src[w];
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + tx, j = j0 + r;
        if (i < n0 && j < n1) {
            const size_t id = (size_t)i + (size_t)n0 * ((size_t)j + (size_t)n1 * k);
            const double xav1 = fmax(xh_av[id], eps);
            const double xav0 = fmax(1.0 - xav1, eps);
            const double v = xav0 * (double)ndens[id];
            nhi[id] = v;
            tile[r][tx] = v;
            if (zero.g[0]) zero.g[0][id] = 0.0;
            if (zero.g[2]) zero.g[2][id] = 0.0;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + tx, i = i0 + r;
        if (i < n0 && j < n1) {
            const size_t idt = (size_t)j + (size_t)n1 * ((size_t)i + (size_t)n0 * k);
            nhi_T[idt] = tile[tx][r];
            if (zero.g[1]) zero.g[1][idt] = 0.0;
            if (zero.g[3]) zero.g[3][idt] = 0.0;
        }
    }
}

// out[j + N1*(i + N0*k)] = in[i + N0*(j + N1*k)]: (x,y) transpose of every z-plane through LDS.
// gate (or null): the launch does nothing unless *gate == 0 -- kernels enqueued behind a sweep before the host knows
// whether every source has retired (the fused outer iteration, c2ray_hip.hip iterate_impl)
template <typename T, bool ACCUM>
__global__ __launch_bounds__(256) void k_transpose_xy(int n0, int n1, int n2, const T *__restrict__ in, T *__restrict__ out,
                                                      const int *gate = nullptr)
{
    __shared__ T tile[32][33];
    if (gate && *gate != 0) return;
    const int k = blockIdx.z;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + tx, j = j0 + r;
        if (i < n0 && j < n1) tile[r][tx] = in[(size_t)i + (size_t)n0 * ((size_t)j + (size_t)n1 * k)];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + tx, i = i0 + r;
        if (i < n0 && j < n1) {
            const size_t o = (size_t)j + (size_t)n1 * ((size_t)i + (size_t)n0 * k);
            if (ACCUM) out[o] += tile[tx][r]; else out[o] = tile[tx][r];
        }
    }
}

// ---- evolve0D for ONE (cell, source): the reference's per-cell call surface (evolve_point.F90:83-299) ----------------
// Slow by construction (a launch and a handful of copies per cell): for hosts that drive the sweep themselves and for tests.
// The cell's geometry as the shell kernels see it (face, plane coordinates a, b, shell sa.q), the four upstream column
// densities as VALUES (the caller reads them from its coldensh_out), n_HI from the context's arrays; the arithmetic is
// cell_state's (bit-identical column densities) and rates_fast's, the source cell's is source_cell's.
// out[0..3] = coldensh_out(pos), the rate to add to phih_grid(pos), to phiheat_grid(pos), the photon loss through the box surface
template <int LLS, bool HEAT>
__global__ __launch_bounds__(64) void k_evolve0d_cell(KParams p, ShellArgs sa, int face, int a, int b, int is_source, int on_surface,
                                                      double c1, double c2, double c3, double c4, double *out)
{
    __shared__ v2f64 s_tab[kLogTab];
    const v2f64 *ltab = wave_log_table(is_source ? p.logtab : p.odtab, s_tab);
    if (threadIdx.x != 0) return;
    const int s = 0;
    const double nflux = p.normflux[s];
    double cd_out, gamma = 0.0, heat = 0.0, loss = 0.0;
    if (is_source) {                                             // evolve_point.F90:151-160
        const unsigned i = wrap_pos(p.srcw[0], p.n[0], 0), j = wrap_pos(p.srcw[1], p.n[1], 0), k = wrap_pos(p.srcw[2], p.n[2], 0);
        const double nhi = p.nhi[(size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k)];
        const double path = 0.5 * step_of(p).dr[0];
        const double vol_ph = step_of(p).dr[0] * step_of(p).dr[1] * step_of(p).dr[2];
        cd_out = 0.0 + nhi * path;
        if (nflux > 0.0) {
            double p_out = 0.0;
            gamma = photoion<HEAT>(p, ltab, 0.0, cd_out, vol_ph, nflux, p_out, &heat) / nhi;
            if (on_surface) loss = p_out * step_of(p).vol / vol_ph;
        }
    } else {
        const CellState cs = cell_state<LLS, false>(p, sa, face, s, a, b, c1, c2, c3, c4, weight_rcp(p, c1), weight_rcp(p, c2),
                                                    weight_rcp(p, c3), weight_rcp(p, c4));
        cd_out = cs.cd_out;
        if (!cs.stop_far && !(cs.cd_in > p.max_coldensh) && nflux > 0.0) {
            double t_out;
            gamma = rates_fast<HEAT>(p, ltab, p.thick, cs.cd_in, cd_out, nflux, cs.vol_ph * cs.nhi, cs.vol_ph, t_out, heat);
            if (on_surface) loss = fdiv((nflux * t_out) * step_of(p).vol, cs.vol_ph);
        }
    }
    out[0] = cd_out; out[1] = gamma; out[2] = heat; out[3] = loss;
}

// ---- sparse exchange of the rates (cold regime, big meshes) --------------------------------------------------------
// evolve.F90:599 all-reduces the whole N^3 phih_grid after every pass, also while the rates are non-zero only inside a few
// sub-boxes.  Every rank knows every source's final sub-box (one small all-reduce of the sub-box counts), so all ranks agree
// on the same list of boxes: pack the rates of those boxes (box after box, in source order), all-reduce the packed
// buffer, write it back.  A cell of overlapping boxes travels once per box; the collective may sum the copies in different
// orders (RCCL's ring / tree order depends on the element's offset), so they can come back differing in the last bit.  The
// write-back therefore takes the MAXIMUM of what the cell holds and every copy -- rates are non-negative, whose f64 bit
// patterns order like unsigned integers, and a sum over the ranks is never below this rank's own addend (rounding is
// monotone) -- one 64-bit atomic max per copy: the same value on every rank whichever copy lands last, and the all-reduce's
// own sum wherever the copies agree.
struct BoxDesc { int c[3]; int nbox; long long off; };      // wrapped source cell, final sub-box count, first packed element
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_pack_boxes(int n0, int n1, int n2, int hl0, int hl1, int hl2, int hr0, int hr1, int hr2,
                                                    int subbox, const BoxDesc *__restrict__ box, double *grid, double *packed)
{
    const BoxDesc b = box[blockIdx.y];
    if (b.nbox <= 0) return;
    const int ext = subbox * b.nbox;
    const int l0 = min(ext, hl0), l1 = min(ext, hl1), l2 = min(ext, hl2);
    const int e0 = l0 + min(ext, hr0) + 1, e1 = l1 + min(ext, hr1) + 1, e2 = l2 + min(ext, hr2) + 1;
    const long long vol = (long long)e0 * e1 * e2;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < vol; t += (long long)gridDim.x * 256) {
        const int i = (int)(t % e0), j = (int)((t / e0) % e1), k = (int)(t / ((long long)e0 * e1));
        const unsigned c0 = wrap_pos(b.c[0], n0, i - l0), c1 = wrap_pos(b.c[1], n1, j - l1), c2 = wrap_pos(b.c[2], n2, k - l2);
        const size_t id = (size_t)c0 + (size_t)n0 * ((size_t)c1 + (size_t)n1 * (size_t)c2);
        if (UNPACK) atomicMax(reinterpret_cast<unsigned long long *>(grid) + id, (unsigned long long)__double_as_longlong(packed[b.off + t]));
        else packed[b.off + t] = grid[id];
    }
}

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

// Adds the block partials of one shell launch to loss_acc[source], in a fixed order.
__global__ __launch_bounds__(256) void k_loss_reduce(const int *active, const int *n_active, const double *loss_partial,
                                                     int bps, double *loss_acc)
{
    __shared__ double sm[4];
    const int sl = blockIdx.x;
    if (sl >= *n_active) return;
    double v = 0.0;
    for (int i = threadIdx.x; i < bps; i += 256) v += loss_partial[(size_t)sl * bps + i];   // bps = 6*tiles_max
    const double tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) loss_acc[active[sl]] += tot;
}

// End of sub-box `nbox` (evolve_source.F90:128-131): keep a source active while more than
// loss_fraction of its photons leave the box and the box can still grow in z.  Compacts the
// active list (stable), finalises the others.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void k_box_decide(const int *active_in, const int *n_in_dev, int *active_out,
                                                     int *n_out, int *n_out_host, const double *normflux, double S_star,
                                                     double loss_fraction, int can_grow, int nbox,
                                                     double *loss_acc, double *final_loss, int *final_nbox,
                                                     const double *loss_partial, int bps)
{
    // loss_partial/bps: block partials of the sub-box's LAST shell launch (bps = 6 x tiles per source, 0: none) -- what
    // k_loss_reduce would add, folded in here to save a launch; four interleaved partial sums, then in a fixed order
    __shared__ int scan[1024];
    __shared__ int base;
    const int n_in = *n_in_dev;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n_in; start += 1024) {
        const int i = start + (int)threadIdx.x;
        int keep = 0, s = -1;
        if (i < n_in) {
            s = active_in[i];
            const double flux = normflux[s] * S_star;
            double loss = loss_acc[s];
            if (bps > 0) {
                const double *pp = loss_partial + (size_t)i * bps;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                int j = 0;
                for (; j + 3 < bps; j += 4) { a0 += pp[j]; a1 += pp[j + 1]; a2 += pp[j + 2]; a3 += pp[j + 3]; }
                for (; j < bps; ++j) a0 += pp[j];
                loss = loss + ((a0 + a1) + (a2 + a3));
                loss_acc[s] = loss;
            }
            keep = (loss > loss_fraction * flux) && can_grow;
            if (keep) loss_acc[s] = 0.0;                      // evolve_source.F90:133
            else { final_loss[s] = loss; final_nbox[s] = nbox; }
        }
        scan[threadIdx.x] = keep;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {            // inclusive Hillis-Steele scan
            int v = 0;
            if ((int)threadIdx.x >= off) v = scan[threadIdx.x - off];
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        if (keep) active_out[base + scan[threadIdx.x] - 1] = s;
        __syncthreads();
        if (threadIdx.x == 1023) base += scan[1023];
        __syncthreads();
    }
    // n_out_host: the schedule's pinned slot for this sub-box, written straight through the mapped pointer
    if (threadIdx.x == 0) { *n_out = base; *n_out_host = base; }
}

// k_box_decide for up to 64 active sources (one wave, no scan through LDS): the same sums in the same order, the same
// decisions.  tot.on: when the decision leaves no source active, the batch's totals (k_batch_totals for a pass of ONE batch:
// the running sums restart from zero) and the per-source results go out with it -- to the device scalars and, through
// mapped pointers, to the host -- so that the launches behind a small batch need neither k_batch_totals nor two copies.
struct SmallTotals {
    int on, nsrc;
    double *photon_loss; long long *sum_nbox;         // device running totals
    double *host_loss; long long *host_nbox;          // the host's pinned scalars
    int *host_final_nbox; double *host_final_loss;    // the host's pinned staging arrays [nsrc]
};
__global__ __launch_bounds__(64) void k_box_decide_small(const int *active_in, const int *n_in_dev, int *active_out,
                                                         int *n_out, int *n_out_host, const double *normflux, double S_star,
                                                         double loss_fraction, int can_grow, int nbox,
                                                         double *loss_acc, double *final_loss, int *final_nbox,
                                                         const double *loss_partial, int bps, SmallTotals tot)
{
#if defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "k_box_decide_small is one 64-lane wave (ballot, cross-lane reads of final_loss): build for a wave64 target (gfx950)"
#endif
    const int n_in = *n_in_dev;                       // <= 64
    const int i = threadIdx.x;
    int keep = 0, s = -1;
    if (i < n_in) {
        s = active_in[i];
        const double flux = normflux[s] * S_star;
        double loss = loss_acc[s];
        if (bps > 0) {
            const double *pp = loss_partial + (size_t)i * bps;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int j = 0;
            for (; j + 3 < bps; j += 4) { a0 += pp[j]; a1 += pp[j + 1]; a2 += pp[j + 2]; a3 += pp[j + 3]; }
            for (; j < bps; ++j) a0 += pp[j];
            loss = loss + ((a0 + a1) + (a2 + a3));
            loss_acc[s] = loss;
        }
        keep = (loss > loss_fraction * flux) && can_grow;
        if (keep) loss_acc[s] = 0.0;                      // evolve_source.F90:133
        else { final_loss[s] = loss; final_nbox[s] = nbox; }
    }
    const unsigned long long mask = __ballot(keep);
    if (keep) active_out[__popcll(mask & ((1ULL << i) - 1ULL))] = s;
    const int n_keep = __popcll(mask);
    if (i == 0) { *n_out = n_keep; *n_out_host = n_keep; }
    if (tot.on && n_keep == 0) {
        __threadfence_block();                            // (one wave: the stores above are ordered before the loads below)
        __syncthreads();
        for (int t = i; t < tot.nsrc; t += 64) { tot.host_final_nbox[t] = final_nbox[t]; tot.host_final_loss[t] = final_loss[t]; }
        if (i == 0) {
            double L = 0.0; long long NB = 0;
            for (int t = 0; t < tot.nsrc; ++t) { L = L + (0.0 + final_loss[t]); NB += final_nbox[t]; }
            *tot.photon_loss = L; *tot.sum_nbox = NB;
            *tot.host_loss = L; *tot.host_nbox = NB;
        }
    }
}

// photon_loss(1) += photon_loss_src, in source order (evolve_source.F90:216); sum_nbox (:219).
// first: first batch of a pass (the running totals restart from zero).  The totals so far are also
// written to the host's pinned scalars through their mapped pointers.
__global__ __launch_bounds__(1024) void k_batch_totals(int nsrc, const double *final_loss, const int *final_nbox,
                                                       double *photon_loss, long long *sum_nbox, int first,
                                                       double *host_loss, long long *host_nbox)
{
    // thread t sums the contiguous run [t*c, (t+1)*c) in order, thread 0 the runs in order: for up to 1024
    // sources (c = 1) that IS the sequential source-order sum of the reference; beyond, a fixed two-level order
    __shared__ double sl[1024];
    __shared__ long long sn[1024];
    const int c = (nsrc + 1023) / 1024;
    const int i0 = (int)threadIdx.x * c, i1 = min(nsrc, i0 + c);
    double l = 0.0; long long nb = 0;
    for (int i = i0; i < i1; ++i) { l = l + final_loss[i]; nb += final_nbox[i]; }
    sl[threadIdx.x] = l; sn[threadIdx.x] = nb;
    __syncthreads();
    if (threadIdx.x == 0) {
        double L = first ? 0.0 : *photon_loss; long long NB = first ? 0 : *sum_nbox;
        const int used = c > 0 ? (nsrc + c - 1) / c : 0;
        for (int t = 0; t < used; ++t) { L = L + sl[t]; NB += sn[t]; }
        *photon_loss = L; *sum_nbox = NB;
        *host_loss = L; *host_nbox = NB;
    }
}

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
        const double h_old0 = 1.0 - h_old1;
        double hav0 = 1.0 - hav1;
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
        const double yh0_old = 1.0 - fmax(c.eps, xav_in);           // evolve_point.F90:378-379
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
        lsum += h1;
        if (STATS) {                                           // k_photon_sums with xl = xh_intermed, xr = xh_av, same expressions
            st_h0 += nd * (1.0 - h1);
            st_h1 += nd * h1;
            const double y1 = hav1, y0 = 1.0 - y1;
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
                                                     double temph0)
{   // temper != null: non-isothermal run, the rate coefficients at every cell's temperature%average (:167)
    __shared__ double sm[4];
    double h0 = 0.0, h1 = 0.0, tr = 0.0, tc = 0.0;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < ncell; id += (size_t)gridDim.x * 256) {
        const double nd = (double)ndens[id];
        const double x = xl[id];
        h0 += nd * (1.0 - x);
        h1 += nd * x;
        const double y1 = xr[id], y0 = 1.0 - y1;
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
