// Photo-ionization rate tables: the one-time set-up the reference does in rad_ini
// (radiation_tables.F90:95-126) -- black-body photon SED, Romberg weights, and the optically
// thick / thin integrals over frequency at 2001 optical depths.  Host code (runs once per run,
// 2001 x 129 exponentials); it feeds c2r_set_tables.  Statement order and operand widths follow
// the reference so the tables agree with the Fortran to the last bit (tests/test_tables.py).
#include "../../include/c2ray_hip.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int kNumFreq = 128;            // radiation_sizes.f90:13
constexpr int kMaxPow = 14;              // romberg.f90:14

// romberg.f90:22-90  romberg_initialisation(nmax): weights romw(0:nmax) of the 2^pmax+1 point rule
std::vector<double> romberg_weights(int nmax)
{
    const int pmax = (int)std::lround(std::log((double)nmax) / (double)std::log(2.0f));   // :44
    std::vector<double> a(kMaxPow + 1, 0.0), b(kMaxPow + 1, 0.0);
    std::vector<std::vector<double>> s(kMaxPow + 1, std::vector<double>(kMaxPow + 1, 0.0));
    std::vector<std::vector<double>> romw(kMaxPow + 1, std::vector<double>(((size_t)1 << kMaxPow) + 1, 0.0));
    for (int k = 1; k <= pmax; ++k) {
        const float p4 = std::pow(4.0f, (float)k);                  // 4.0 ** k, default real (exact)
        b[k] = (double)(-1.0f / (p4 - 1.0f));                       // :51  evaluated in default real
        a[k] = -b[k] * (double)p4;                                  // :52
    }
    for (int k = 0; k <= pmax; ++k) {
        s[k][0] = 1.0;                                              // :65
        for (int j = 1; j <= pmax; ++j)
            for (int i = pmax; i >= j; --i)
                s[i][j] = a[j] * s[i][j - 1] + b[j] * s[i - 1][j - 1];              // :69
        for (int i = k; i <= pmax; ++i)
            for (int j = 0; j <= (1 << k); ++j) {
                const int idx = (1 << (i - k)) * j;
                romw[i][idx] = s[i][i] * (double)(1 << (i - k)) + romw[i][idx];     // :76
            }
        s[k][0] = 0.0;
    }
    for (int i = 0; i <= pmax; ++i) {                               // :84-87 edge weights halved
        romw[i][0] = 0.5 * romw[i][0];
        romw[i][(size_t)1 << i] = 0.5 * romw[i][(size_t)1 << i];
    }
    return std::vector<double>(romw[pmax].begin(), romw[pmax].begin() + nmax + 1);
}

}  // namespace

extern "C" {

int c2r_default_sed(c2r_sed_params *p)
{
    if (!p) return C2R_EINVAL;
    // values of the compiled reference modules ("Exact constant values")
    p->T_eff = 5.0e4;                                  // sed_parameters.f90  bb_Teff
    p->S_star = 1.00000000000000004e+48;               // bb_S_star
    p->min_freq = 3.28851300169676800e+15;             // bb_MinFreq = ion_freq_HI
    p->max_freq = 1.31598566206146560e+17;             // bb_MaxFreq = 10 * ion_freq_HeII
    p->pl_index_cross_section = 2.8;                   // radiation_sizes.f90:85
    p->hplanck = 6.62607550000000009e-27;              // cgsconstants.f90:30
    p->k_B = 1.38099999999999991e-16;                  // cgsconstants.f90:34
    p->two_pi_over_c_square = 6.99098471993956998e-21; // cgsphotoconstants.f90
    p->R_solar = 6.95990026240000000e+10;              // cgsastroconstants.f90
    p->pi = 3.14159274101257324;                       // mathconstants.f90:21
    p->minlogtau = -20.0; p->maxlogtau = 4.0;          // radiation_tables.F90:45-46
    p->numtau = 2000;
    p->sed_type = C2R_SED_BLACK_BODY;                  // sed_parameters.f90:26 stellar_SED_type=1
    p->pl_index = 3.0;                                 // sed_parameters.f90:40 (used by C2R_SED_POWER_LAW)
    p->grey = 0;                                       // c2ray_parameters.f90:43
    return C2R_OK;
}

int c2r_default_sed_power_law(c2r_sed_params *p)
{
    const int rc = c2r_default_sed(p);
    if (rc) return rc;
    p->sed_type = C2R_SED_POWER_LAW;                   // stellar_SED_type=2
    p->S_star = 1.00000000000000004e+48;               // sed_parameters.f90:42 pl_S_star
    p->min_freq = 3.28851300169676800e+15;             // pl_MinFreq = ion_freq_HI
    p->max_freq = 1.31598566206146560e+16;             // pl_MaxFreq = ion_freq_HeII
    return C2R_OK;
}

// rad_ini for the black-body or the power-law source (sourcetype "B" / "P"), with the frequency-dependent or the grey cross
// section; hthick/hthin (optional): the heating tables of a non-isothermal run
static int build_all(const c2r_sed_params *sp, double ion_freq_HI, double *thick, double *thin, double *hthick, double *hthin,
                     int32_t n, double *R_star_out)
{
    if (!sp || n != sp->numtau + 1 || sp->numtau < 1) return C2R_EINVAL;
    if (sp->sed_type != 0 && sp->sed_type != C2R_SED_BLACK_BODY && sp->sed_type != C2R_SED_POWER_LAW) return C2R_EINVAL;
    const bool power_law = sp->sed_type == C2R_SED_POWER_LAW;
    const int NF = kNumFreq, NT = sp->numtau;
    // radiation_sed_parameters.F90:82-163  spectrum_parms (black body)
    const double T_eff = std::fmax(std::fmin(sp->T_eff, (double)1e6f), (double)2000.f);
    double R_star = sp->R_solar;
    const double h_over_kT = sp->hplanck / (sp->k_B * T_eff);
    // radiation_sizes.f90:36-89  setup_scalingfactors
    const double freq_min = sp->min_freq, freq_max = sp->max_freq;
    const double delta_freq = (freq_max - freq_min) / (double)(float)NF;
    // romberg.f90:22
    const std::vector<double> romw = romberg_weights(NF);
    double S_scaling = 1.0;                                   // radiation_sed_parameters.F90:72
    if (power_law) {
        // radiation_sed_parameters.F90:204-222 spec_diag "P" + :272-279 integrate_sed("P","S"): photon-number power law
        const double freq_step = (freq_max - freq_min) / (double)(float)NF;
        double integral = 0.0;
        for (int i = 0; i <= NF; ++i) {
            const double f = freq_min + freq_step * (double)(float)i;
            const double integrand = std::pow(f, -sp->pl_index);                 // :276
            integral = integral + integrand * freq_step * romw[i] * 1.0;        // romberg.f90:139-140
        }
        const double S_unscaled = S_scaling * integral;                         // :279 (S_scaling still 1)
        S_scaling = sp->S_star / S_unscaled;                                    // :208
    } else
    // radiation_sed_parameters.F90:172-224 spec_diag + :226-283 integrate_sed("B","S")
    {
        const double freq_step = (freq_max - freq_min) / (double)(float)NF;
        double integral = 0.0;
        for (int i = 0; i <= NF; ++i) {
            const double f = freq_min + freq_step * (double)(float)i;
            double integrand;
            if (f * h_over_kT <= 709.0)
                integrand = sp->two_pi_over_c_square * f * f / (std::exp(f * h_over_kT) - 1.0);
            else
                integrand = sp->two_pi_over_c_square * f * f / (std::exp((f * h_over_kT) / 2.0)) /
                            (std::exp((f * h_over_kT) / 2.0));
            integral = integral + integrand * freq_step * romw[i] * 1.0;        // romberg.f90:139-140
        }
        const double S_unscaled = 4.0 * sp->pi * R_star * R_star * integral;    // :270
        const double S_scaling = sp->S_star / S_unscaled;                       // :188
        R_star = std::sqrt(S_scaling) * R_star;                                 // :189
    }
    const double R_star2 = R_star * R_star;
    if (R_star_out) *R_star_out = R_star;
    // radiation_tables.F90:130-236 spec_integration
    std::vector<double> tau(NT + 1), freq(NF + 1), cs(NF + 1), sed(NF + 1);
    const double dlogtau = (sp->maxlogtau - sp->minlogtau) / (double)(float)NT;     // :47
    for (int i = 1; i <= NT; ++i) tau[i] = std::pow(10.0, sp->minlogtau + dlogtau * (double)(float)(i - 1));
    tau[0] = 0.0;
    for (int i = 0; i <= NF; ++i) {
        freq[i] = freq_min + delta_freq * (double)(float)i;                          // :330
        cs[i] = sp->grey ? 1.0 : std::pow(freq[i] / freq_min, -sp->pl_index_cross_section);   // :345 / :352
        const double f = freq[i];
        if (power_law) sed[i] = S_scaling * std::pow(f, -sp->pl_index);              // PL_SED :455-466
        else
        sed[i] = (f * h_over_kT < 700.0)                                             // BB_SED :434-452
                     ? 4.0 * sp->pi * R_star2 * sp->two_pi_over_c_square * f * f / (std::exp(f * h_over_kT) - 1.0)
                     : 0.0;
    }
    for (int it = 0; it <= NT; ++it) {
        double sum_thick = 0.0, sum_thin = 0.0, sum_hthick = 0.0, sum_hthin = 0.0;
        for (int i = 0; i <= NF; ++i) {
            double fk = 0.0, fn = 0.0;
            if (tau[it] * cs[i] < 700.0) {                                           // :390
                fk = sed[i] * std::exp(-tau[it] * cs[i]);
                fn = sed[i] * cs[i] * std::exp(-tau[it] * cs[i]);
            }
            sum_thick = sum_thick + fk * delta_freq * romw[i];                      // romberg.f90:182-184
            sum_thin = sum_thin + fn * delta_freq * romw[i];
            if (hthick) {                                                            // :455-475 fill_heating_integrands_HI
                const double hk = sp->hplanck * (freq[i] - ion_freq_HI) * fk;
                const double hn = sp->hplanck * (freq[i] - ion_freq_HI) * fn;
                sum_hthick = sum_hthick + hk * delta_freq * romw[i];                 // :521-530 make_heat_tables_HI
                sum_hthin = sum_hthin + hn * delta_freq * romw[i];
            }
        }
        if (thick) { thick[it] = sum_thick; thin[it] = sum_thin; }
        if (hthick) { hthick[it] = sum_hthick; hthin[it] = sum_hthin; }
    }
    return C2R_OK;
}

int c2r_build_tables(const c2r_sed_params *sp, double *thick, double *thin, int32_t n, double *R_star_out)
{
    if (!thick || !thin) return C2R_EINVAL;
    return build_all(sp, 0.0, thick, thin, nullptr, nullptr, n, R_star_out);
}

int c2r_build_heat_tables(const c2r_sed_params *sp, double ion_freq_HI, double *heat_thick, double *heat_thin, int32_t n)
{
    if (!heat_thick || !heat_thin) return C2R_EINVAL;
    return build_all(sp, ion_freq_HI, nullptr, nullptr, heat_thick, heat_thin, n, nullptr);
}

}  // extern "C"
