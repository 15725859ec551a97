/* ORACLE -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * A plain-C, serial, f64 restatement of the algorithm of the C2-Ray evolve hot
 * path (reference: garrelt/C2-Ray3Dm).  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check (and time a
 * CPU baseline beside) the HIP path.  Nothing in the product may include, link
 * or call it.
 *
 * Parity status: PINNED.  Every function here is checked in tests/test_oracle.py
 * against fixtures under tests/golden/ that were produced by the UNMODIFIED
 * reference compiled with amdflang (oracle/ref_build.sh, oracle/ref_driver.F90,
 * tests/golden/make_golden.py).  The non-isothermal branches (heat_thick != NULL) are pinned the same way against the
 * reference rebuilt with isothermal=.false.; the cooling table that build reads is synthetic (the reference
 * repository lacks tables/corocool.tab) -- tests/test_oracle_thermal.py, DESIGN.md s8a.
 *
 * Each function cites the reference file:line it restates.  Operation order and
 * operand widths follow the reference statement by statement, because the
 * fixtures are compared at the 1e-13 level: build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/c2ray_constants.h"

typedef struct {
    int    n[3];              /* mesh(1:3)                       sizes.f90:33           */
    double dr[3];             /* cell size (cm)                  grid.F90:97-104        */
    double vol;               /* cell volume                     grid.F90               */
    double coldensh_LLS;      /* LLS column per cell             LLS.F90:178-182        */
    double clumping;          /* clumping (f32 in the reference) clumping_module.F90:17 */
    double temper;            /* isothermal temperature          c2ray_parameters.f90:112 */
    double S_star;            /* table normalisation             radiation_sed_parameters */
    const double *thick;      /* stellar_photo_thick_table(0:NumTau,1) */
    const double *thin;       /* stellar_photo_thin_table(0:NumTau,1)  */
    /* non-default physics switches of c2ray_parameters.f90:75-99 (0/NULL = shipped behaviour) */
    int    lls_type;          /* type_of_LLS: 1 homogeneous, 2 LLS_grid per cell, 3 hard barrier at R_max_LLS */
    double R_max_LLS;         /* LLS.F90:191  R_max/(1+z), proper cm (type 3)                  */
    const float *lls_grid;    /* LLS.F90:208  LLS_grid (f32), type 2                           */
    const float *clump_grid;  /* clumping_module.F90:116 clumping_grid (f32), type_of_clumping 3-5 */
    /* Checker-side diagnostic, not part of the reference's algorithm (NULL = off): per cell
     *   W = sum over sources of (1 + tau_in) * photo_in / (vol_ph * n_HI),
     * the rate the cell would have if it absorbed every arriving photon, weighted by the optical depth
     * in front of it.  The tests bound the rate error by RTOL*Gamma + WTOL*W: Gamma = (T(tau_in)-T(tau_out))
     * NormFlux/(vol_ph n_HI) is a difference of two table values whose own rounding error (log10 and the
     * table position, ~1e-16 od per unit of tau) does not shrink with the difference. */
    double *tolw;
    /* Non-isothermal run (c2ray_parameters.f90:28 isothermal=.false.; heat_thick == NULL: the shipped isothermal run).
     * Pinned against the reference rebuilt with that one parameter changed (oracle/ref_build.sh 32:thermal) and run with
     * a SYNTHETIC tables/corocool.tab (tests/golden/inputs.py cooling_table: the reference repository lacks the file). */
    const double *heat_thick; /* stellar_heat_thick_table(0:NumTau,1)  radiation_tables.F90:300 */
    const double *heat_thin;  /* stellar_heat_thin_table(0:NumTau,1)                            */
    const double *cie_cool;   /* cooling.f90:27 cie_cool(1:temppoints), linear (10**table)      */
    double cool_mintemp, cool_dtemp;   /* cooling.f90:78-79                                     */
    double zred;              /* cosmology.F90:42 zred at the middle of the step (cosmo_cool)   */
    float  *temper_grid;      /* temperature_module.F90:35 temperature_grid: (current, average, intermed) f32 per cell */
    double *phiheat;          /* evolve_data.F90:42 phiheat_grid                                */
    double *tolw_heat;        /* checker diagnostic like tolw, for the heating rate: sum_s (1+tau_in) heat_in / vol_ph */
    /* Second ("P") source type of photoion_rates (radiation_photoionrates.F90:133-137; builds with use_xray_SED=.true.,
     * sed_parameters.f90:56).  xray_thick == NULL: the shipped build.  Pinned against the reference rebuilt with that one
     * parameter (oracle/ref_build.sh 32:xray) whose X-ray tables the fixture driver sets (the reference integrates them
     * over an array it never fills, radiation_tables.F90:367): the tables are inputs of this path, like the stellar ones. */
    const double *xray_thick; /* xray_photo_thick_table(0:NumTau,1) */
    const double *xray_thin;  /* xray_photo_thin_table(0:NumTau,1)  */
    const double *xray_flux;  /* NormFlux_xray(1:NumSrc): column 5 of the source list / S_star_xray (sourceprops.F90:381, 631) */
    const double *xray_heat_thick, *xray_heat_thin;   /* xray_heat_thick/thin_table(0:NumTau,1): non-isothermal runs (heat_lookuptable "P", :165-171) */
    /* Builds of the reference with -DALLFRAC (ionfractions_module.F90:19-50; no shipped makefile defines it): xh, xh_av, xh_intermed are
     * (mesh,0:1) and the NEUTRAL fraction is stored, not derived as 1 - x.  xh0 == NULL: the shipped build.  Pinned against the reference
     * compiled with that one flag (oracle/ref_build.sh 32:allfrac).  The three arrays below are the (:,:,:,0) halves; the pointers every
     * routine already takes are the (:,:,:,1) halves. */
    double *xh0, *xh_av0, *xh_intermed0;
    long   *thermal_stats;    /* checker diagnostic (NULL = off): [0] thermal() calls, [1] of them left untouched because T_initial <=
                               * minitemp (thermal.f90:83), [2] of them ended by the sub-step cap i_heating > 10000 (:163), [3] sub-steps
                               * in all -- what the fixtures exercise; the reference has no such counters */
} oracle_cfg;

static inline int pmod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }
static inline int isign1(int d) { return d < 0 ? -1 : 1; }            /* sign(1,d): +1 for d==0 */
static inline size_t cidx(const int n[3], int i, int j, int k)        /* 1-based, column-major */
{ return (size_t)(i - 1) + (size_t)n[0] * ((size_t)(j - 1) + (size_t)n[1] * (size_t)(k - 1)); }
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* column_density.f90:276-293  weightf */
static inline double weightf(double cd) { return 1.0 / dmax(C2R_WEIGHT_FLOOR, cd * C2R_SIGMA_HI); }

/* column_density.f90:29-271  cinterp.  pos/srcpos are unwrapped 1-based mesh positions. */
void oracle_cinterp(const double *cd, const int n[3], const int pos[3], const int src[3],
                    double *cdensi, double *path)
{
    const int i = pos[0], j = pos[1], k = pos[2], i0 = src[0], j0 = src[1], k0 = src[2];
    const int idel = i - i0, jdel = j - j0, kdel = k - k0;
    const int idela = abs(idel), jdela = abs(jdel), kdela = abs(kdel);
    const int sgni = isign1(idel), sgnj = isign1(jdel), sgnk = isign1(kdel);
    const int im = i - sgni, jm = j - sgnj, km = k - sgnk;
    const double di = (double)(float)idel, dj = (double)(float)jdel, dk = (double)(float)kdel;
    double c1, c2, c3, c4, s1, s2, s3, s4, w1, w2, w3, w4, alam, cdi;

    if (kdela >= jdela && kdela >= idela) {                                   /* :108 */
        alam = (double)((float)(km - k0) + (float)sgnk * 0.5f) / dk;          /* :112 */
        const double xc = alam * di + (double)(float)i0;
        const double yc = alam * dj + (double)(float)j0;
        const double dx = 2.0 * fabs(xc - (double)((float)im + 0.5f * (float)sgni));
        const double dy = 2.0 * fabs(yc - (double)((float)jm + 0.5f * (float)sgnj));
        s1 = (1. - dx) * (1. - dy); s2 = (1. - dy) * dx; s3 = (1. - dx) * dy; s4 = dx * dy;
        const int ip = pmod(i - 1, n[0]) + 1, imp = pmod(im - 1, n[0]) + 1;
        const int jp = pmod(j - 1, n[1]) + 1, jmp = pmod(jm - 1, n[1]) + 1;
        const int kmp = pmod(km - 1, n[2]) + 1;
        c1 = cd[cidx(n, imp, jmp, kmp)]; c2 = cd[cidx(n, ip, jmp, kmp)];
        c3 = cd[cidx(n, imp, jp, kmp)];  c4 = cd[cidx(n, ip, jp, kmp)];
        w1 = s1 * weightf(c1); w2 = s2 * weightf(c2); w3 = s3 * weightf(c3); w4 = s4 * weightf(c4);
        cdi = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);   /* :140 */
        if (kdela == 1 && (idela == 1 || jdela == 1))                          /* :152-158 */
            cdi = (idela == 1 && jdela == 1) ? C2R_SQRT3 * cdi : C2R_SQRT2 * cdi;
        *path = sqrt((di * di + dj * dj) / (dk * dk) + 1.0);                   /* :168 */
    } else if (jdela >= idela && jdela >= kdela) {                             /* :173 */
        alam = (double)((float)(jm - j0) + (float)sgnj * 0.5f) / dj;
        const double zc = alam * dk + (double)(float)k0;
        const double xc = alam * di + (double)(float)i0;
        const double dz = 2.0 * fabs(zc - (double)((float)km + 0.5f * (float)sgnk));
        const double dx = 2.0 * fabs(xc - (double)((float)im + 0.5f * (float)sgni));
        s1 = (1. - dx) * (1. - dz); s2 = (1. - dz) * dx; s3 = (1. - dx) * dz; s4 = dx * dz;
        const int ip = pmod(i - 1, n[0]) + 1, imp = pmod(im - 1, n[0]) + 1;
        const int jmp = pmod(jm - 1, n[1]) + 1;
        const int kp = pmod(k - 1, n[2]) + 1, kmp = pmod(km - 1, n[2]) + 1;
        c1 = cd[cidx(n, imp, jmp, kmp)]; c2 = cd[cidx(n, ip, jmp, kmp)];
        c3 = cd[cidx(n, imp, jmp, kp)];  c4 = cd[cidx(n, ip, jmp, kp)];
        w1 = s1 * weightf(c1); w2 = s2 * weightf(c2); w3 = s3 * weightf(c3); w4 = s4 * weightf(c4);
        cdi = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
        if (jdela == 1 && (idela == 1 || kdela == 1))
            cdi = (idela == 1 && kdela == 1) ? C2R_SQRT3 * cdi : C2R_SQRT2 * cdi;
        *path = sqrt((di * di + dk * dk) / (dj * dj) + 1.0);                   /* :221 */
    } else {                                                                   /* :226 */
        alam = (double)((float)(im - i0) + (float)sgni * 0.5f) / di;
        const double zc = alam * dk + (double)(float)k0;
        const double yc = alam * dj + (double)(float)j0;
        const double dz = 2.0 * fabs(zc - (double)((float)km + 0.5f * (float)sgnk));
        const double dy = 2.0 * fabs(yc - (double)((float)jm + 0.5f * (float)sgnj));
        s1 = (1. - dz) * (1. - dy); s2 = (1. - dz) * dy; s3 = (1. - dy) * dz; s4 = dy * dz;
        const int imp = pmod(im - 1, n[0]) + 1;
        const int jp = pmod(j - 1, n[1]) + 1, jmp = pmod(jm - 1, n[1]) + 1;
        const int kp = pmod(k - 1, n[2]) + 1, kmp = pmod(km - 1, n[2]) + 1;
        c1 = cd[cidx(n, imp, jmp, kmp)]; c2 = cd[cidx(n, imp, jp, kmp)];
        c3 = cd[cidx(n, imp, jmp, kp)];  c4 = cd[cidx(n, imp, jp, kp)];
        w1 = s1 * weightf(c1); w2 = s2 * weightf(c2); w3 = s3 * weightf(c3); w4 = s4 * weightf(c4);
        cdi = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
        if (idela == 1 && (jdela == 1 || kdela == 1))
            cdi = (jdela == 1 && kdela == 1) ? C2R_SQRT3 * cdi : C2R_SQRT2 * cdi;
        *path = sqrt(1.0 + (dj * dj + dk * dk) / (di * di));                   /* :265 */
    }
    *cdensi = cdi;
}

/* radiation_photoionrates.F90:184-228  set_tau_table_positions + read_table */
static inline double table_lookup(const double *tab, double tau)
{
    const double lt = log10(dmax(1.0e-20, tau));
    const double od = dmin((double)C2R_NUMTAU, dmax(0.0, 1.0 + (lt - C2R_MINLOGTAU) / C2R_DLOGTAU));
    const int ip = (int)od;
    const double res = od - (double)ip;
    const int ip1 = ip + 1 < C2R_NUMTAU ? ip + 1 : C2R_NUMTAU;
    return tab[ip] + (tab[ip1] - tab[ip]) * res;
}

/* radiation_photoionrates.F90:71-179 photoion_rates + :233-317 photo_lookuptable (NumFreqBnd=1,
 * stellar table only).  out = { photo_cell_HI (= cell/vol), photo_in, photo_out } */
void oracle_photoion_rates(const double *thick, const double *thin, double cd_in, double cd_out,
                           double vol, double normflux, double out[3])
{
    out[0] = out[1] = out[2] = 0.0;
    if (!(normflux > 0.0)) return;                                             /* :126 */
    const double tau_in = cd_in * C2R_SIGMA_HI, tau_out = cd_out * C2R_SIGMA_HI;
    const double p_in = normflux * table_lookup(thick, tau_in);                /* :281 */
    double p_out, p_cell;
    if (fabs(tau_out - tau_in) > C2R_TAU_PHOTO_LIMIT) {                        /* :289 */
        p_out = normflux * table_lookup(thick, tau_out);
        p_cell = p_in - p_out;
    } else {                                                                   /* :299 */
        p_cell = normflux * (tau_out - tau_in) * table_lookup(thin, tau_in);
        p_out = p_in - p_cell;
    }
    out[0] = p_cell / vol; out[1] = p_in; out[2] = p_out;
}

/* radiation_photoionrates.F90:323-417 heat_lookuptable (stellar table, NumFreqBnd=1): phi%heat */
double oracle_heat_rate(const double *hthick, const double *hthin, double cd_in, double cd_out,
                        double vol, double normflux)
{
    if (!(normflux > 0.0)) return 0.0;                                         /* :157 */
    const double tau_in = cd_in * C2R_SIGMA_HI, tau_out = cd_out * C2R_SIGMA_HI;
    const double tau_cell = (cd_out - cd_in) * C2R_SIGMA_HI;                   /* :104, :146 */
    const double h_in = normflux * table_lookup(hthick, tau_in);               /* :384 */
    if (fabs(tau_out - tau_in) > C2R_TAU_HEAT_LIMIT) {                         /* :388 */
        const double h_out = normflux * table_lookup(hthick, tau_out);
        return (h_in - h_out) / vol;
    }
    return normflux * tau_cell * table_lookup(hthin, tau_in) / vol;            /* :396-400 */
}

/* cooling.f90:38-59 coolin (cie_cool is 1-based in the reference) */
double oracle_coolin(const oracle_cfg *c, double nucldens, double eldens, double temp0)
{
    const double tpos = (log10(temp0) - c->cool_mintemp) / c->cool_dtemp + 1.0;
    int itpos = (int)tpos;
    if (itpos < 1) itpos = 1;
    if (itpos > C2R_COOL_POINTS - 1) itpos = C2R_COOL_POINTS - 1;
    const double dtpos = tpos - (double)(float)itpos;
    const int itpos1 = itpos + 1 < C2R_COOL_POINTS ? itpos + 1 : C2R_COOL_POINTS;
    return nucldens * eldens * (c->cie_cool[itpos - 1] + (c->cie_cool[itpos1 - 1] - c->cie_cool[itpos - 1]) * dtpos);
}

/* tped.f90:32-62 */
static inline double temper2pressr(double temper, double ndens, double eldens) { return (ndens + eldens) * C2R_K_B * temper; }
static inline double pressr2temper(double pressr, double ndens, double eldens) { return pressr / (C2R_K_B * (ndens + eldens)); }
static inline double electrondens(double ndens, double x1) { return ndens * (x1 + C2R_ABU_C); }

/* cosmology.F90:198-225 cosmo_cool */
static inline double cosmo_cool(const oracle_cfg *c, double e_int)
{
    const double zp = 1.0 + c->zred;
    const double dzdt = C2R_H0 * zp * sqrt(C2R_OMEGA0 * (zp * zp * zp) + 1.0 - C2R_OMEGA0);
    return e_int * 2.0 / zp * dzdt;
}

/* thermal.f90:22-189.  Outputs are left untouched when T_initial <= minitemp (:83). */
void oracle_thermal(const oracle_cfg *c, double dt, double t_initial, double *t_final, double *t_average,
                    double ndens_electron, double ndens_atom, double h_old1, double h_av1, double h1, double heating)
{
    double e_int = temper2pressr(t_initial, ndens_atom, electrondens(ndens_atom, h_old1)) / C2R_GAMMA1;   /* :66 */
    const double cosmo_cool_rate = cosmo_cool(c, e_int);                       /* :73-76 (cosmological=.true.) */
    if (c->thermal_stats) c->thermal_stats[0]++;
    if (!(t_initial > C2R_MINITEMP)) { if (c->thermal_stats) c->thermal_stats[1]++; return; }
    double cumulative = 0.0, avg = 0.0, t_int = t_initial;
    int i_heating = 0;
    for (;;) {
        i_heating++;
        const double cooling = oracle_coolin(c, ndens_atom, ndens_electron, t_int) + cosmo_cool_rate;   /* :104 */
        const double rate = dmax(C2R_THERMAL_RATE_FLOOR, fabs(cooling - heating));
        const double timescale = e_int / fabs(rate);
        const double dt_thermal = C2R_RELATIVE_DENERGY * timescale;
        const double dt_ode = dmin(dt_thermal, dt - cumulative);               /* :127 */
        e_int = e_int + dt_ode * (heating - cooling);
        avg = avg + 0.5 * t_int * dt_ode;
        t_int = pressr2temper(e_int * C2R_GAMMA1, ndens_atom, electrondens(ndens_atom, h_av1));   /* :137 */
        avg = avg + 0.5 * t_int * dt_ode;
        if (t_int < C2R_MINITEMP) {                                            /* :147-153 (no division by gamma1 there) */
            e_int = temper2pressr(C2R_MINITEMP, ndens_atom, electrondens(ndens_atom, h_av1));
            t_int = C2R_MINITEMP;
        }
        cumulative = cumulative + dt_ode;
        if (cumulative >= dt || fabs(cumulative - dt) < C2R_THERMAL_TIME_TOL * dt) break;   /* :160 */
        if (i_heating > C2R_THERMAL_MAX_STEPS) { if (c->thermal_stats) c->thermal_stats[2]++; break; }   /* :163 */
    }
    if (c->thermal_stats) c->thermal_stats[3] += i_heating;
    *t_average = dt > 0.0 ? avg / dt : t_initial;                              /* :168-172 */
    *t_final = pressr2temper(e_int * C2R_GAMMA1, ndens_atom, electrondens(ndens_atom, h1));   /* :175 */
}

/* doric.f90:33-134.  xfh/xfh_av are (0:1) = {neutral, ionized}. */
void oracle_doric(double dt, double temp0, double rhe, double clumping,
                  double xfh[2], double xfh_av[2], double phih)
{
    const double brech0 = clumping * C2R_BH00 * pow(temp0 / 1e4, C2R_ALBPOW);  /* :73 */
    const double sqrtt0 = sqrt(temp0);
    const double acolh0 = C2R_COLH0 * sqrtt0 * exp(-C2R_TEMPH0 / temp0);       /* :78 */
    const double xfh1old = xfh[1], xfh0old = xfh[0];
    const double aih0 = phih + rhe * acolh0;
    const double delth = aih0 + rhe * brech0;
    const double eqxfh1 = aih0 / delth;
    const double eqxfh0 = rhe * brech0 / delth;
    const double deltht = delth * dt;
    const double ee = exp(-deltht);
    xfh[1] = (xfh1old - eqxfh1) * ee + eqxfh1;
    xfh[0] = (xfh0old - eqxfh0) * ee + eqxfh0;
    if (xfh[0] < C2R_EPSILON) { xfh[0] = C2R_EPSILON; xfh[1] = 1.0 - C2R_EPSILON; }   /* :107 */
    const double avg = deltht < C2R_DELTHT_SMALL ? 1.0 : (1.0 - ee) / deltht;  /* :119 */
    xfh_av[1] = eqxfh1 + (xfh1old - eqxfh1) * avg;
    xfh_av[0] = 1.0 - xfh_av[1];
    if (xfh_av[0] < C2R_EPSILON) xfh_av[0] = C2R_EPSILON;                      /* :130 */
}

/* ---- per-source sweep ---------------------------------------------------------------- */
typedef struct {
    const oracle_cfg *c;
    const float  *ndens;
    const double *xh_av;
    double *phih;
    double *cdout;             /* coldensh_out scratch, N^3 */
    const int *src;            /* unwrapped source position */
    double normflux;
    double normflux_x;         /* NormFlux_xray(ns) (0: none) */
    int last_l[3], last_r[3];
    double loss;               /* photon_loss_src_thread(1) */
    long   visited;
} sweep_t;

/* evolve_point.F90:83-299  evolve0D (niter /= -1, isothermal, type_of_LLS=1) */
static void evolve0d(sweep_t *s, const int rt[3])
{
    const oracle_cfg *c = s->c;
    const int *n = c->n;
    int pos[3];
    for (int d = 0; d < 3; ++d) pos[d] = pmod(rt[d] - 1, n[d]) + 1;            /* :122 */
    const size_t id = cidx(n, pos[0], pos[1], pos[2]);
    if (s->cdout[id] != 0.0) return;                                           /* :128 */
    s->visited++;
    const double xav1 = dmax(s->xh_av[id], C2R_EPSILON);                       /* :137 */
    const double xav0 = c->xh_av0 ? dmax(c->xh_av0[id], C2R_EPSILON)           /* ALLFRAC :131-132: the stored neutral fraction */
                                  : dmax(1.0 - xav1, C2R_EPSILON);             /* :140 */
    const double nd = (double)s->ndens[id];
    double cd_in, path, vol_ph;
    int stop_far = 0;
    if (rt[0] == s->src[0] && rt[1] == s->src[1] && rt[2] == s->src[2]) {      /* :151 */
        cd_in = 0.0;
        path = 0.5 * c->dr[0];
        vol_ph = c->dr[0] * c->dr[1] * c->dr[2];
    } else {
        oracle_cinterp(s->cdout, n, rt, s->src, &cd_in, &path);
        path = path * c->dr[0];
        const double xs = c->dr[0] * (double)(float)(rt[0] - s->src[0]);
        const double ys = c->dr[1] * (double)(float)(rt[1] - s->src[1]);
        const double zs = c->dr[2] * (double)(float)(rt[2] - s->src[2]);
        const double dist2 = xs * xs + ys * ys + zs * zs;
        vol_ph = 4.0 * C2R_PI * dist2 * path;                                  /* :177 */
        if (c->lls_type == 3) {                                                /* :187-191 */
            if (dist2 > c->R_max_LLS * c->R_max_LLS) stop_far = 1;
        } else {
            const double lls = c->lls_type == 2 ? (double)c->lls_grid[id] : c->coldensh_LLS;   /* :193 LLS_point */
            cd_in = cd_in + lls * path / c->dr[0];                             /* :194 */
        }
    }
    const int stop = stop_far || cd_in > C2R_MAX_COLDENSH;                     /* :201 */
    const double cd_out = cd_in + xav0 * nd * path;                            /* :247, doric.f90:153 */
    s->cdout[id] = cd_out;
    double phi[3] = {0.0, 0.0, 0.0}, heat = 0.0;
    if (!stop) {
        oracle_photoion_rates(c->thick, c->thin, cd_in, cd_out, vol_ph, s->normflux, phi);
        if (c->xray_thick && s->normflux_x > 0.0) {                             /* radiation_photoionrates.F90:133-137: phi = phi + "P" */
            double px[3];
            oracle_photoion_rates(c->xray_thick, c->xray_thin, cd_in, cd_out, vol_ph, s->normflux_x, px);
            phi[0] = phi[0] + px[0]; phi[1] = phi[1] + px[1]; phi[2] = phi[2] + px[2];
        }
        if (c->heat_thick) heat = oracle_heat_rate(c->heat_thick, c->heat_thin, cd_in, cd_out, vol_ph, s->normflux);   /* radiation_photoionrates.F90:142-172 */
        if (c->heat_thick && c->xray_heat_thick && s->normflux_x > 0.0) {                                               /* :165-171 phi = phi + heat "P" */
            heat = heat + oracle_heat_rate(c->xray_heat_thick, c->xray_heat_thin, cd_in, cd_out, vol_ph, s->normflux_x);
            if (c->tolw_heat) c->tolw_heat[id] += (1.0 + cd_in * C2R_SIGMA_HI) * s->normflux_x * table_lookup(c->xray_heat_thick, cd_in * C2R_SIGMA_HI) / vol_ph;
        }
        if (c->heat_thick && c->tolw_heat && s->normflux > 0.0)
            c->tolw_heat[id] += (1.0 + cd_in * C2R_SIGMA_HI) * s->normflux * table_lookup(c->heat_thick, cd_in * C2R_SIGMA_HI) / vol_ph;
        phi[0] = phi[0] / (xav0 * nd);                                         /* :262 */
        if (c->tolw) c->tolw[id] += (1.0 + cd_in * C2R_SIGMA_HI) * phi[1] / (vol_ph * (xav0 * nd));
    }
    s->phih[id] = s->phih[id] + phi[0];                                        /* :283 */
    if (c->heat_thick) c->phiheat[id] = c->phiheat[id] + heat;                 /* :285-286 */
    if (rt[0] == s->last_l[0] || rt[1] == s->last_l[1] || rt[2] == s->last_l[2] ||
        rt[0] == s->last_r[0] || rt[1] == s->last_r[1] || rt[2] == s->last_r[2])
        s->loss = s->loss + phi[2] * c->vol / vol_ph;                          /* :290-293 */
}

/* evolve_source.F90:227-267  evolve2D */
static void evolve2d(sweep_t *s, int k)
{
    int rt[3]; rt[2] = k;
    for (int j = s->src[1]; j <= s->last_r[1]; ++j) {
        rt[1] = j;
        for (int i = s->src[0]; i <= s->last_r[0]; ++i) { rt[0] = i; evolve0d(s, rt); }
        for (int i = s->src[0] - 1; i >= s->last_l[0]; --i) { rt[0] = i; evolve0d(s, rt); }
    }
    for (int j = s->src[1] - 1; j >= s->last_l[1]; --j) {
        rt[1] = j;
        for (int i = s->src[0]; i <= s->last_r[0]; ++i) { rt[0] = i; evolve0d(s, rt); }
        for (int i = s->src[0] - 1; i >= s->last_l[0]; --i) { rt[0] = i; evolve0d(s, rt); }
    }
}

/* evolve_source.F90:58-221  do_source, serial branch (:188-208).
 * Adds this source's rates into phih; returns nbox; *loss_out = final photon_loss_src.
 * cdout (N^3 scratch) holds coldensh_out of this source on return. */
int oracle_do_source_x(const oracle_cfg *c, const float *ndens, const double *xh_av, double *phih,
                       double *cdout, const int src[3], double normflux, double normflux_xray,
                       double *loss_out, long *visited_out)
{
    const size_t ncell = (size_t)c->n[0] * c->n[1] * c->n[2];
    memset(cdout, 0, ncell * sizeof(double));                                  /* :91 */
    sweep_t s = { c, ndens, xh_av, phih, cdout, src, normflux, normflux_xray, {0,0,0}, {0,0,0}, 0.0, 0 };
    int lastpos_l[3], lastpos_r[3];
    for (int d = 0; d < 3; ++d) {                                              /* :100-102 */
        const int hr = c->n[d] / 2 - 1 + c->n[d] % 2, hl = c->n[d] / 2;
        lastpos_r[d] = src[d] + (C2R_MAX_SUBBOX < hr ? C2R_MAX_SUBBOX : hr);
        lastpos_l[d] = src[d] - (C2R_MAX_SUBBOX < hl ? C2R_MAX_SUBBOX : hl);
        s.last_r[d] = src[d]; s.last_l[d] = src[d];
    }
    int nbox = 0;
    const double total_flux = normflux * c->S_star;                            /* :119 */
    double loss_src = total_flux;
    while (loss_src > C2R_LOSS_FRACTION * total_flux &&
           s.last_r[2] < lastpos_r[2] && s.last_l[2] > lastpos_l[2]) {         /* :128-131 */
        nbox++;
        s.loss = 0.0;
        for (int d = 0; d < 3; ++d) {
            const int r = src[d] + C2R_SUBBOXSIZE * nbox, l = src[d] - C2R_SUBBOXSIZE * nbox;
            s.last_r[d] = r < lastpos_r[d] ? r : lastpos_r[d];
            s.last_l[d] = l > lastpos_l[d] ? l : lastpos_l[d];
        }
        for (int k = src[2]; k <= s.last_r[2]; ++k) evolve2d(&s, k);           /* :192-195 */
        for (int k = src[2] - 1; k >= s.last_l[2]; --k) evolve2d(&s, k);       /* :198-201 */
        loss_src = s.loss;                                                     /* :208 */
    }
    *loss_out = loss_src;
    if (visited_out) *visited_out = s.visited;
    return nbox;
}

int oracle_do_source(const oracle_cfg *c, const float *ndens, const double *xh_av, double *phih,
                     double *cdout, const int src[3], double normflux,
                     double *loss_out, long *visited_out)
{
    return oracle_do_source_x(c, ndens, xh_av, phih, cdout, src, normflux, 0.0, loss_out, visited_out);
}

/* evolve.F90:444-495 pass_all_sources + master_slave.F90:74-96 do_grid_static for one rank:
 * sources rank+1, rank+1+npr, ... (1-based).  phih must be zeroed by the caller
 * (set_rates_to_zero, evolve.F90:430).  srcpos is 3 x S (column-major, 1-based positions). */
void oracle_pass_sources(const oracle_cfg *c, const float *ndens, const double *xh_av, double *phih,
                         const int *srcpos, const double *normflux, int nsrc, int rank, int npr,
                         double *photon_loss, long *sum_nbox, long *visited)
{
    const size_t ncell = (size_t)c->n[0] * c->n[1] * c->n[2];
    double *cdout = (double *)malloc(ncell * sizeof(double));
    double loss_total = 0.0; long nb = 0, vis = 0;
    for (int ns = rank; ns < nsrc; ns += npr) {
        double loss; long v;
        nb += oracle_do_source_x(c, ndens, xh_av, phih, cdout, srcpos + 3 * ns, normflux[ns],
                                 c->xray_flux ? c->xray_flux[ns] : 0.0, &loss, &v);
        loss_total = loss_total + loss;                                        /* evolve_source.F90:216 */
        vis += v;
    }
    free(cdout);
    *photon_loss = loss_total; *sum_nbox = nb; if (visited) *visited = vis;
}

/* evolve_point.F90:305-406 evolve0D_global + :410-555 do_chemistry(local=.false.) over the
 * whole mesh in the order of global_pass (evolve.F90:548-555).  Returns conv_flag. */
long oracle_global_pass(const oracle_cfg *c, double dt, const float *ndens, const double *xh,
                        double *xh_av, double *xh_intermed, const double *phih)
{
    const size_t ncell = (size_t)c->n[0] * c->n[1] * c->n[2];
    long conv_flag = 0;
    for (size_t id = 0; id < ncell; ++id) {
        double h_old[2], h[2], h_av[2];
        h_old[1] = dmax(C2R_EPSILON, xh[id]);
        h_av[1] = dmax(C2R_EPSILON, xh_av[id]);
        h_old[0] = 1.0 - h_old[1];
        h_av[0] = 1.0 - h_av[1];
        if (c->xh0) {                                                          /* ALLFRAC :341-346: both stored, both floored */
            h_old[0] = dmax(C2R_EPSILON, c->xh0[id]);
            h_av[0] = dmax(C2R_EPSILON, c->xh_av0[id]);
        }
        const double nd = (double)ndens[id];
        const double gamma = phih[id];
        const int thermal = c->heat_thick != NULL;
        /* get_temperature_point, temperature_module.F90:133-151; temperature_end = temperature_start (:436) */
        float *tg = thermal ? c->temper_grid + 3 * id : NULL;
        const double t_start_cur = thermal ? (double)tg[0] : c->temper, t_start_avg = thermal ? (double)tg[1] : c->temper;
        double t_end_avg = t_start_avg, t_end_int = thermal ? (double)tg[2] : c->temper;
        const double heat = thermal ? c->phiheat[id] : 0.0;                    /* :364 */
        int nit = 0;
        for (;;) {                                                             /* :442 */
            nit++;
            const double yh0_av_old = h_av[0];
            h[0] = h_old[0]; h[1] = h_old[1];                                  /* :463 */
            double de = nd * (h_av[1] + C2R_ABU_C);                            /* tped.f90:81 */
            oracle_doric(dt, t_end_avg, de, c->clump_grid ? (double)c->clump_grid[id] : c->clumping, h, h_av, gamma);   /* :443-445 clumping_point, :515 */
            de = nd * (h_av[1] + C2R_ABU_C);                                   /* :518 */
            if (thermal)                                                       /* :521-527 */
                oracle_thermal(c, dt, t_start_cur, &t_end_int, &t_end_avg, de, nd, h_old[1], h_av[1], h[1], heat);
            /* :531-538: the temperature clause compares temperature_end%current with its copy of the iteration
             * before; thermal never writes %current, so the clause is |0/T| < 1e-3: true for every finite T > 0 */
            if (fabs((h_av[0] - yh0_av_old) / h_av[0]) < C2R_MIN_FRACTIONAL_CHANGE ||
                h_av[0] < C2R_MIN_FRACTION_OF_ATOMS) break;
            if (nit > C2R_MAX_CHEM_ITER) break;                                /* :541 */
        }
        double t_new_avg = t_start_avg;
        if (thermal) {                                                         /* :553 set_temperature_point: f32 stores */
            tg[2] = (float)t_end_int;
            tg[1] = (float)t_end_avg;
            t_new_avg = (double)tg[1];                                         /* :381 get_temperature_point again */
        }
        const double yh1_av_old = dmax(C2R_EPSILON, xh_av[id]);                /* :378 */
        const double yh0_av_old = c->xh0 ? c->xh_av0[id] : 1.0 - yh1_av_old;   /* ALLFRAC :375: as stored, not floored */
        if ((fabs(h_av[0] - yh0_av_old) > C2R_MIN_FRACTIONAL_CHANGE &&
             fabs((h_av[0] - yh0_av_old) / h_av[0]) > C2R_MIN_FRACTIONAL_CHANGE &&
             h_av[0] > C2R_MIN_FRACTION_OF_ATOMS) ||
            (fabs((t_start_avg - t_new_avg) / t_new_avg) > C2R_TEMP_CONV_REL &&
             fabs(t_start_avg - t_new_avg) > C2R_TEMP_CONV_ABS)) conv_flag++;  /* :384-391 (.or. binds looser than .and.) */
        xh_intermed[id] = h[1];
        xh_av[id] = h_av[1];
        if (c->xh0) { c->xh_intermed0[id] = h[0]; c->xh_av0[id] = h_av[0]; }   /* ALLFRAC :395-398 */
    }
    return conv_flag;
}

/* SUM() of a real(8) array as the reference build evaluates it (evolve.F90:183): amdflang -O2
 * inlines the intrinsic as one sequential left-to-right accumulation (checked against the
 * "Intermediate result for mean H ionization fraction" lines of the reference log, which a
 * Kahan or pairwise sum does not reproduce in the last digits). */
double oracle_sum(const double *a, size_t n)
{
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s = s + a[i];
    return s;
}

/* photonstatistics.F90:104-217  state_before / state_after / total_rates: the four mesh sums
 * (sequential, k-j-i order) before scaling.  out = { h0, h1, totrec, totcollisions } where
 * h0,h1 use xh_l and the rate sums use xh_r. */
/* ALLFRAC (photonstatistics.F90:106-119, :158-160, :204-206): the stored neutral fractions of xh_l / xh_r (xl0 / xr0: the (:,:,:,0)
 * halves; NULL: derived as 1 - x) */
void oracle_photon_sums_x(const oracle_cfg *c, const float *ndens, const double *xh_l, const double *xh_r,
                          const double *xl0, const double *xr0, double out[4]);
void oracle_photon_sums(const oracle_cfg *c, const float *ndens, const double *xh_l, const double *xh_r,
                        double out[4])
{
    oracle_photon_sums_x(c, ndens, xh_l, xh_r, NULL, NULL, out);
}
void oracle_photon_sums_x(const oracle_cfg *c, const float *ndens, const double *xh_l, const double *xh_r,
                          const double *xl0, const double *xr0, double out[4])
{
    const size_t ncell = (size_t)c->n[0] * c->n[1] * c->n[2];
    double h0 = 0.0, h1 = 0.0, totrec = 0.0, totcoll = 0.0;
    const double rec = pow(c->temper / 1e4, C2R_ALBPOW), sq = sqrt(c->temper), ex = exp(-C2R_TEMPH0 / c->temper);
    for (size_t id = 0; id < ncell; ++id) {
        const double nd = (double)ndens[id];
        h0 = h0 + nd * (xl0 ? xl0[id] : 1.0 - xh_l[id]);
        h1 = h1 + nd * xh_l[id];
        const double y1 = xh_r[id], y0 = xr0 ? xr0[id] : 1.0 - xh_r[id];
        const double de = nd * (y1 + C2R_ABU_C);
        if (c->heat_thick) {                                                   /* :167 get_temperature_point: %average */
            const double t = (double)c->temper_grid[3 * id + 1];
            totrec = totrec + nd * y1 * de * (c->clump_grid ? (double)c->clump_grid[id] : c->clumping) * C2R_BH00 * pow(t / 1e4, C2R_ALBPOW);
            totcoll = totcoll + nd * y0 * de * C2R_COLH0 * sqrt(t) * exp(-C2R_TEMPH0 / t);
            continue;
        }
        totrec = totrec + nd * y1 * de * (c->clump_grid ? (double)c->clump_grid[id] : c->clumping) * C2R_BH00 * rec;   /* :163-168 */
        totcoll = totcoll + nd * y0 * de * C2R_COLH0 * sq * ex;                /* :169-172 */
    }
    out[0] = h0; out[1] = h1; out[2] = totrec; out[3] = totcoll;
}

typedef struct {
    int    niter;                   /* outer iterations done                       */
    int    converged;               /* 1 = xh updated (evolve.F90:218), 0 = gave up (:228) */
    long   conv_flag;               /* last global_pass count                      */
    double photon_loss_all;
    long   sum_nbox_all;
    long   visited;                 /* cell-source pairs executed, all iterations  */
    /* per outer iteration k (0-based, k < niter): values logged by the reference */
    long   it_conv_flag[128];       /* "Number of non-converged points"            */
    double it_rel1[128], it_rel0[128]; /* Test-2 values seen at the TOP of iteration k+1 */
    long   it_sum_nbox[128];
    double it_sum_xh1[128];         /* sum(xh_intermed) after global pass k        */
    /* photon statistics of the step (photonstatistics.F90, evolve.F90:136,277) */
    double totrec, totcollisions, dh0, total_ion;
} oracle_report;

/* evolve.F90:83-281  evolve3D, single rank.  restart_niter < 0: fresh start (restart=0).
 * restart_niter >= 0: the state (phih, xh_av, xh_intermed, niter) was loaded by start_from_dump
 * (evolve.F90:155-157): a global pass is run first, and the previous-sum variables are whatever the
 * module holds -- zero in a fresh process (they are saved module variables, evolve.F90:67-74). */
void oracle_evolve3d_x(const oracle_cfg *c, double dt, const float *ndens, double *xh,
                       double *xh_av, double *xh_intermed, double *phih,
                       const int *srcpos, const double *normflux, int nsrc, int restart_niter,
                       oracle_report *rep)
{
    const size_t ncell = (size_t)c->n[0] * c->n[1] * c->n[2];
    int niter = 0;
    long conv_flag = (long)ncell;
    double prev1 = (double)(((2.0f * (float)c->n[0]) * (float)c->n[1]) * (float)c->n[2]);   /* :150-151 */
    double prev0 = prev1;
    double rel1 = 1.0, rel0 = 1.0;
    if (restart_niter < 0) {
        memcpy(xh_av, xh, ncell * sizeof(double));                             /* :145-146 */
        memcpy(xh_intermed, xh, ncell * sizeof(double));
        if (c->xh0) {                                                          /* ALLFRAC :142-143: all of (:,:,:,:) */
            memcpy(c->xh_av0, c->xh0, ncell * sizeof(double));
            memcpy(c->xh_intermed0, c->xh0, ncell * sizeof(double));
        }
    } else {
        niter = restart_niter;
        prev1 = prev0 = 0.0;
        conv_flag = oracle_global_pass(c, dt, ndens, xh, xh_av, xh_intermed, phih);      /* :157 */
    }
    long c1 = (long)(C2R_CONVERGENCE_FRACTION * c->n[0] * c->n[1] * c->n[2]);  /* :162 */
    long c2 = (nsrc - 1) / 3;
    const long conv_criterion = c1 < c2 ? c1 : c2;
    memset(rep, 0, sizeof(*rep));
    double before[4], after[4];
    oracle_photon_sums_x(c, ndens, xh, xh, c->xh0, c->xh0, before);            /* state_before, evolve.F90:136 */
    for (;;) {
        const double sum1 = oracle_sum(xh_intermed, ncell);                    /* :183 */
        const double sum0 = c->xh0 ? oracle_sum(c->xh_intermed0, ncell)        /* ALLFRAC :180-181 */
                                   : (double)(float)ncell - sum1;              /* :184 */
        rel1 = sum1 > 0.0 ? fabs(sum1 - prev1) / sum1 : 1.0;
        rel0 = sum0 > 0.0 ? fabs(sum0 - prev0) / sum0 : 1.0;
        if (niter > 0 && niter <= 128) { rep->it_rel1[niter - 1] = rel1; rep->it_rel0[niter - 1] = rel0;
                                         rep->it_sum_xh1[niter - 1] = sum1; }
        if (conv_flag < conv_criterion ||
            (rel1 < C2R_CONVERGENCE_FRACTION && rel0 < C2R_CONVERGENCE_FRACTION)) {   /* :212 */
            memcpy(xh, xh_intermed, ncell * sizeof(double));
            if (c->xh0) memcpy(c->xh0, c->xh_intermed0, ncell * sizeof(double));   /* ALLFRAC :216 */
            if (c->heat_thick)                                                 /* :220 set_final_temperature_point */
                for (size_t id = 0; id < ncell; ++id) c->temper_grid[3 * id] = c->temper_grid[3 * id + 2];
            rep->converged = 1;
            break;
        } else if (niter > C2R_MAX_OUTER_ITER) {                               /* :228 */
            rep->converged = 0;
            break;
        }
        prev1 = sum1; prev0 = sum0;
        niter++;
        memset(phih, 0, ncell * sizeof(double));                               /* :243 */
        if (c->heat_thick) memset(c->phiheat, 0, ncell * sizeof(double));      /* :435 */
        if (c->tolw) memset(c->tolw, 0, ncell * sizeof(double));               /* checker diagnostic: last pass only */
        if (c->tolw_heat) memset(c->tolw_heat, 0, ncell * sizeof(double));
        double loss; long nb, vis;
        oracle_pass_sources(c, ndens, xh_av, phih, srcpos, normflux, nsrc, 0, 1, &loss, &nb, &vis);
        rep->photon_loss_all = loss; rep->sum_nbox_all = nb; rep->visited += vis;
        conv_flag = oracle_global_pass(c, dt, ndens, xh, xh_av, xh_intermed, phih);   /* :269 */
        if (niter <= 128) { rep->it_conv_flag[niter - 1] = conv_flag; rep->it_sum_nbox[niter - 1] = nb; }
    }
    rep->niter = niter; rep->conv_flag = conv_flag;
    oracle_photon_sums_x(c, ndens, xh, xh_av, c->xh0, c->xh_av0, after);       /* evolve.F90:277 */
    rep->totrec = after[2] * c->vol * dt;
    rep->totcollisions = after[3] * c->vol * dt;
    rep->dh0 = before[0] * c->vol - after[0] * c->vol;
    rep->total_ion = rep->totrec + rep->dh0;
}

void oracle_evolve3d(const oracle_cfg *c, double dt, const float *ndens, double *xh,
                     double *xh_av, double *xh_intermed, double *phih,
                     const int *srcpos, const double *normflux, int nsrc, oracle_report *rep)
{
    oracle_evolve3d_x(c, dt, ndens, xh, xh_av, xh_intermed, phih, srcpos, normflux, nsrc, -1, rep);
}
