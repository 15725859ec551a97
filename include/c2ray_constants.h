/* Physical and numerical constants of the C2-Ray evolve hot path, as the compiled
 * reference sees them.
 *
 * The reference writes most of its constants as DEFAULT-REAL (f32) literals that
 * are then widened to real(kind=dp); the widened f32 value -- not the decimal
 * literal -- is what every cell's arithmetic uses.  Values below are the
 * `ES25.17` prints of the compiled reference modules recorded in SURVEY.md s8
 * ("Exact constant values"); 17 significant digits round-trip an IEEE f64.
 * Each line cites the defining site in the reference.
 */
#ifndef C2RAY_CONSTANTS_H
#define C2RAY_CONSTANTS_H

/* mathconstants.f90:21  pi=3.141592654 (f32 literal) */
#define C2R_PI                        3.14159274101257324
/* cgsphotoconstants.f90:24  sigma_HI_at_ion_freq=freq_factor*6.30e-18 */
#define C2R_SIGMA_HI                  6.29999986469627735e-18
/* c2ray_parameters.f90:31  epsilon=1e-14_dp */
#define C2R_EPSILON                   1e-14
/* c2ray_parameters.f90:25  convergence_fraction=1.0e-4 */
#define C2R_CONVERGENCE_FRACTION      9.99999974737875164e-05
/* c2ray_parameters.f90:34  minimum_fractional_change=1.0e-3 */
#define C2R_MIN_FRACTIONAL_CHANGE     1.00000004749745131e-03
/* c2ray_parameters.f90:40  minimum_fraction_of_atoms=1.0e-8 */
#define C2R_MIN_FRACTION_OF_ATOMS     9.99999993922529029e-09
/* c2ray_parameters.f90:67  loss_fraction=1e-2_dp */
#define C2R_LOSS_FRACTION             1.0e-2
/* c2ray_parameters.f90:54,61 */
#define C2R_SUBBOXSIZE                5
#define C2R_MAX_SUBBOX                1000
/* evolve_point.F90:95  max_coldensh=2e19 */
#define C2R_MAX_COLDENSH              1.99999999610128957e+19
/* radiation_photoionrates.F90:244  tau_photo_limit=1.0e-7 */
#define C2R_TAU_PHOTO_LIMIT           1.00000001168609742e-07
/* column_density.f90:52-53  sqrt(3.0), sqrt(2.0) evaluated in f32 */
#define C2R_SQRT3                     1.73205077648162842
#define C2R_SQRT2                     1.41421353816986084
/* column_density.f90:289  weightf floor 0.6_dp */
#define C2R_WEIGHT_FLOOR              0.6
/* radiation_sizes.f90:14  NumTau; radiation_tables.F90:45-47  minlogtau, dlogtau */
#define C2R_NUMTAU                    2000
#define C2R_MINLOGTAU                 (-20.0)
#define C2R_DLOGTAU                   0.012 /* (4-(-20))/real(2000), a correctly rounded f64 division */
/* cgsconstants.f90:64,66,80,86 */
#define C2R_ALBPOW                    (-0.699999999999999956)
#define C2R_BH00                      2.59000000000000007e-13
#define C2R_TEMPH0                    1.57804333545729518e+05
#define C2R_COLH0                     5.83541027596890338e-11
/* abundances.f90:26  abu_c=7.1e-7 */
#define C2R_ABU_C                     7.09999994796817191e-07
/* radiation_sed_parameters: S_star (bb_S_star = 1e48, sed_parameters.f90) */
#define C2R_S_STAR                    1.00000000000000004e+48
/* evolve.F90:228  outer iteration cap; evolve_point.F90:541  chemistry iteration cap */
#define C2R_MAX_OUTER_ITER            100
#define C2R_MAX_CHEM_ITER             400
/* doric.f90:119  deltht threshold 1.0e-8 (f32 literal) */
#define C2R_DELTHT_SMALL              9.99999993922529029e-09


/* ---- non-isothermal runs (c2ray_parameters.f90:28 isothermal=.false.; shipped: .true.) ---- */
/* radiation_photoionrates.F90:333  tau_heat_limit=1.0e-4 (f32 literal) */
#define C2R_TAU_HEAT_LIMIT            9.99999974737875164e-05
/* cgsconstants.f90:30,34  hplanck, k_B */
#define C2R_HPLANCK                   6.62607550000000009e-27
#define C2R_K_B                       1.38099999999999991e-16
/* cgsphotoconstants.f90  ion_freq_HI=ev2fr*eth0 */
#define C2R_ION_FREQ_HI               3.28851300169676800e+15
/* atomic.f90:23-25  gamma1 = 5.0_dp/3.0_dp - 1.0_dp */
#define C2R_GAMMA1                    (5.0 / 3.0 - 1.0)
/* c2ray_parameters.f90:108,110  minitemp=1.0, relative_denergy=0.1 (f32 literal) */
#define C2R_MINITEMP                  1.0
#define C2R_RELATIVE_DENERGY          1.00000001490116119e-01
/* thermal.f90:117 floor of |cooling - heating| (1d-50); :160 exit tolerance 1e-6 (f32 literal); :163 sub-step cap */
#define C2R_THERMAL_RATE_FLOOR        1.0e-50
#define C2R_THERMAL_TIME_TOL          9.99999997475242708e-07
#define C2R_THERMAL_MAX_STEPS         10000
/* cooling.f90:26  temppoints */
#define C2R_COOL_POINTS               61
/* evolve_point.F90:387-388  temperature clause of the global convergence test */
#define C2R_TEMP_CONV_REL             1.0e-1
#define C2R_TEMP_CONV_ABS             100.0
/* cosmoparms.f90:28-31 (WMAP5+): H0 = 100 h km/s/Mpc in s^-1, Omega0 (f32 literals widened) */
#define C2R_H0                        2.26830837024227824e-18
#define C2R_OMEGA0                    2.70000010728836060e-01

#endif
