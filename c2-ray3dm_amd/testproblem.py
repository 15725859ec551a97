"""Inputs of the hot path for the reference's own test problem, for any mesh size.

The reference driver does not travel to the GPU box, so this module restates the ~20 lines of
set-up that `C2Ray.F90` + `nbody_test.F90` + `cosmology.F90` + `LLS.F90` + `density_module.F90` +
`time_module.F90` + `sourceprops.F90` execute before `evolve3D` is called:
a 100 Mpc/h box of uniform mean-IGM density at z=9, time steps of 1e7/10 yr, proper lengths and
densities taken at the mid-point of the step, a constant comoving LLS mean free path of 10 Mpc.
Checked against the scalars recorded from the reference in tests/golden (tests/test_host.py).
Constants are the compiled reference's values ("Exact constant values").
"""
import numpy as np

MPC = 3.08600011031262003e+24        # cgsastroconstants.f90:31
YEAR = 3.15576e+07                   # cgsastroconstants.f90:27
H_LITTLE = 0.699999988079071045      # cosmoparms.f90:28
OMEGA0 = 0.270000010728836060        # cosmoparms.f90:30
OMEGA_B = 4.39999997615814209e-02    # cosmoparms.f90:31
H0 = 2.26830837024227824e-18         # cosmoparms.f90:41
RHO_CRIT_0 = 9.20346643016612840e-30 # cosmoparms.f90:42
MU = 1.22200000286102295             # abundances.f90:32
M_P = 1.67266100000000007e-24        # cgsconstants.f90:26
SIGMA_HI = 6.29999986469627735e-18   # cgsphotoconstants.f90:24
S_STAR = 1.00000000000000004e+48     # sed_parameters.f90 (bb_S_star)
BOXSIZE = 100.0                      # nbody_test.F90:44, Mpc/h comoving
Z_START = 9.0                        # nbody_test.F90:228
SLICE_YEARS = 1e7                    # nbody_test.F90:225
STEPS_PER_SLICE = 10                 # inputs/input_example_test
LLS_CMFP_MPC = 10.0                  # LLS.F90 const_cmfp_LLS (LLS_model=5)
XH_INITIAL = 2e-4                    # ionfractions_module.F90:49


class TestProblem:
    """Per-step scalars + uniform fields of the reference test problem on an N^3 mesh."""
    __test__ = False     # not a pytest class

    def __init__(self, n, boxsize=BOXSIZE):
        self.n = int(n)
        self.boxsize = boxsize
        # cosmology.F90:63  t0 (Einstein-de Sitter high-z approximation)
        self.t0 = 2.0 * (1.0 + Z_START) ** (-1.5) / (3.0 * H0 * np.sqrt(OMEGA0))
        self.dt = float(np.float32(SLICE_YEARS)) * YEAR / STEPS_PER_SLICE     # time_module.F90:93
        self.dr_comoving = boxsize * MPC / H_LITTLE / self.n                  # grid.F90:97-104

    def zred_at(self, t):
        return -1.0 + (1.0 + Z_START) * ((self.t0 + t) / self.t0) ** (-2.0 / 3.0)   # cosmology.F90:147

    def slice_redshift(self, step):
        """Redshift of the density/source slice that time step `step` (1-based) belongs to."""
        nz = (step - 1) // STEPS_PER_SLICE
        return self.zred_at(nz * float(np.float32(SLICE_YEARS)) * YEAR) if nz else Z_START

    def step(self, step=1):
        """Scalars handed to evolve3D for time step `step`: proper values at mid-step."""
        t_mid = (step - 1) * self.dt + 0.5 * self.dt
        z_mid = self.zred_at(t_mid)                                   # C2Ray.F90:367
        z_slice = self.slice_redshift(step)
        dr = self.dr_comoving / (1.0 + z_mid)                         # cosmology.F90:186
        # density_module.F90:136 at the slice redshift (f32 storage), rescaled to mid-step
        n_slice = np.float32(RHO_CRIT_0 * OMEGA_B / (MU * M_P) * (1.0 + z_slice) ** 3)
        ndens = np.float32(float(n_slice) * ((1.0 + z_mid) / (1.0 + z_slice)) ** 3)
        # LLS.F90:178-179; the `zred` the driver passes (C2Ray.F90:376) is cosmology's module
        # variable, which redshift_evol has just moved to the mid-step value
        mfp_pmpc = max(LLS_CMFP_MPC / (1.0 + z_mid), 1.0 / (1.0 + z_mid))
        coldensh_lls = (1.0 / SIGMA_HI) * (dr / (mfp_pmpc * MPC))               # LLS.F90:181-182
        return dict(mesh=self.n, dt=self.dt, dr1=dr, dr2=dr, dr3=dr, vol=dr * dr * dr,
                    coldensh_LLS=coldensh_lls, clumping=1.0, S_star=S_STAR, zred=z_mid,
                    ndens=float(ndens), temper=1e4)

    def at_time(self, t):
        """Proper dr, vol, mean density and LLS column at simulation time t, i.e. what redshift_evol(t) +
        cosmo_evol leave behind at the end of a slice (C2Ray.F90:416-419).  Used for the output-time
        statistics only (the driver reaches the same values by incremental f32 rescaling; they agree
        to ~1e-7)."""
        z = self.zred_at(t)
        dr = self.dr_comoving / (1.0 + z)
        ndens = np.float32(RHO_CRIT_0 * OMEGA_B / (MU * M_P) * (1.0 + z) ** 3)
        mfp_pmpc = max(LLS_CMFP_MPC / (1.0 + z), 1.0 / (1.0 + z))
        return dict(dr1=dr, vol=dr * dr * dr, ndens=float(ndens), zred=z,
                    coldensh_LLS=(1.0 / SIGMA_HI) * (dr / (mfp_pmpc * MPC)))

    def fields(self, step=1, x_init=XH_INITIAL):
        """Uniform ndens (f32) and xh (f64) as flat Fortran-order arrays."""
        s = self.step(step)
        ncell = self.n ** 3
        return (np.full(ncell, s["ndens"], dtype=np.float32), np.full(ncell, x_init, dtype=np.float64))


def seeded_sources(n, nsrc, seed=20261003, flux_lo=1e54, flux_hi=1e57):
    """Synthetic source list of SURVEY.md s8d: distinct uniform positions in [1,N]^3, photon rates
    log-uniform in [flux_lo, flux_hi] s^-1.  Returns (srcpos (S,3) int32, NormFlux (S,) f64) with
    NormFlux = rate / S_star as the Test UV model does (sourceprops.F90:627-631)."""
    rng = np.random.default_rng(seed + nsrc)
    flat = rng.choice(n ** 3, size=nsrc, replace=False)
    pos = np.stack([flat % n, (flat // n) % n, flat // (n * n)], axis=1).astype(np.int32) + 1
    flux = 10.0 ** rng.uniform(np.log10(flux_lo), np.log10(flux_hi), size=nsrc)
    return pos, flux / S_STAR


def write_source_file(path, srcpos, normflux):
    """The reference's 5-column source list (sourceprops.F90:293-391): N, then `i j k rate 0.0`."""
    with open(path, "w") as f:
        f.write("%d\n" % len(normflux))
        for (i, j, k), nf in zip(srcpos, normflux):
            f.write("%d %d %d %.17e 0.0\n" % (i, j, k, nf * S_STAR))


def synthetic_cooling_table(kind="primordial"):
    """kind="steep": a SECOND synthetic curve, shaped like a collisional-ionization-equilibrium curve of enriched gas where it
    matters to thermal.f90: negligible below 8e3 K, a rise of four orders of magnitude between 1e4 and 1e5 K (H, He and metal
    lines), a slow decline beyond, free-free at the hot end -- and a cold-gas coolant below 100 K that DECLINES with
    temperature, so that coolin's linear extrapolation below the first table row (cooling.f90:47-58: T < 10 K) stays
    positive and large: dense cold cells are driven to minitemp, pinned there by thermal.f90:147-153 sub-step after
    sub-step, and leave through the cap i_heating > 10000 (:163).
    kind="primordial" (the default, the curve of the first fixtures):
    A SYNTHETIC 61-point cooling curve in the format of the reference's tables/corocool.tab (cooling.f90:71-76:
    rows of log10 T, log10 Lambda [erg cm^3 s^-1]).  The reference repository does not ship that file, so the
    non-isothermal fixtures are generated with this one: hydrogen excitation + collisional ionization +
    recombination + free-free cooling of a primordial gas (textbook fits), rounded to the 4 decimals written.
    Returns (text of the file, log10 T values, log10 Lambda values as the Fortran list-directed read sees them)."""
    lt = [float("%.2f" % (1.0 + 0.1 * i)) for i in range(61)]
    T = 10.0 ** np.array(lt)
    if kind == "steep":
        lam = (3.0e-22 * (10.0 / T) ** 1.5 / (1.0 + (T / 100.0) ** 4)
               + 6.6e-21 * np.exp(-118348.0 / T) * (1e4 / T) ** 0.2
               + 1.2e-20 * np.exp(-473638.0 / T) / (1.0 + (T / 2e5) ** 1.8)
               + 2.3e-27 * np.sqrt(T))
    else:
        lam = (7.5e-19 * np.exp(-118348.0 / T) / (1.0 + np.sqrt(T / 1e5))
               + 1.27e-21 * np.sqrt(T) * np.exp(-157809.1 / T) / (1.0 + np.sqrt(T / 1e5))
               + 8.7e-27 * np.sqrt(T) * (T / 1e3) ** -0.2 / (1.0 + (T / 1e6) ** 0.7)
               + 1.42e-27 * 1.3 * np.sqrt(T))
    ll = [float("%.4f" % v) for v in np.log10(lam)]
    text = "".join("%5.2f %9.4f\n" % (a, b) for a, b in zip(lt, ll))
    return text, np.array(lt), np.array(ll)
