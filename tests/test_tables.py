"""CPU tests of the rate-table builder (c2r_build_tables: the reference's rad_ini chain,
radiation_tables.F90 / radiation_sed_parameters.F90 / romberg.f90) against the tables dumped from
the compiled reference.  Host code: no GPU needed."""
import ctypes as C
import numpy as np
import pytest


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_tables_equal_reference_bit_for_bit(pkg, tables):
    thick, thin, r_star = pkg.build_tables()
    assert np.array_equal(thick, tables[0])
    assert np.array_equal(thin, tables[1])
    # known answers the reference logs (radiation_tables.F90:212-215)
    assert abs(thick[0] / 9.999999999999995e47 - 1) < 1e-15
    assert abs(thin[0] / 4.567323859246105e47 - 1) < 1e-15
    assert 1e11 < r_star < 1.2e11          # rescaled black-body radius (1.62 R_solar)


def test_power_law_and_grey_tables_equal_reference_bit_for_bit(pkg):
    """The other two switches of the table builder, each against the reference rebuilt with that one parameter
    (oracle/ref_build.sh 32:pl, 32:grey; tests/golden/make_golden.py seds): stellar_SED_type=2 -- a power law in photon
    number between the HI and HeII edges (sed_parameters.f90:38-45, radiation_tables.F90:455-466) -- and grey=.true.
    (c2ray_parameters.f90:43, radiation_tables.F90:338-347)."""
    import os
    from tests._util import GOLDEN
    lib = pkg.load_library()
    sed = pkg.SedParams()
    assert lib.c2r_default_sed_power_law(C.byref(sed)) == 0 and sed.sed_type == 2 and sed.pl_index == 3.0
    thick, thin, _ = pkg.build_tables(sed)
    ref = np.load(os.path.join(GOLDEN, "tables_pl.npz"))
    assert np.array_equal(thick, ref["thick"]) and np.array_equal(thin, ref["thin"])
    assert abs(thick[0] / 1e48 - 1) < 1e-14                       # normalised to S_star photons per second
    sed = pkg.SedParams()
    lib.c2r_default_sed(C.byref(sed)); sed.grey = 1
    thick, thin, _ = pkg.build_tables(sed)
    ref = np.load(os.path.join(GOLDEN, "tables_grey.npz"))
    assert np.array_equal(thick, ref["thick"]) and np.array_equal(thin, ref["thin"])
    assert np.array_equal(thick, thin)                            # sigma(nu) = sigma_0: both integrands are SED exp(-tau)
    bad = pkg.SedParams(); lib.c2r_default_sed(C.byref(bad)); bad.sed_type = 3
    assert lib.c2r_build_tables(C.byref(bad), thick.ctypes.data, thin.ctypes.data, 2001, None) != 0


def test_table_properties(pkg):
    thick, thin, _ = pkg.build_tables()
    # tau=0 entry integrates the bare SED to S_star; thick decreases monotonically with tau
    assert np.all(np.diff(thick[1:]) <= 0)
    assert thick[1] <= thick[0] and thick[-1] >= 0
    # d(thick)/d(tau) = -thin (definition of the two integrands): check by finite differences
    tau = np.concatenate([[0.0], 10.0 ** (-20.0 + 0.012 * np.arange(2000))])
    k = np.arange(1200, 1700)      # tau ~ 1e-5.6 .. 2.5
    num = -(thick[k + 1] - thick[k - 1]) / (tau[k + 1] - tau[k - 1])
    assert np.max(np.abs(num / thin[k] - 1)) < 2e-3


def test_other_sed_parameters(pkg):
    """A hotter black body and a different optical-depth grid: sizes and normalisation follow."""
    sed = pkg.SedParams()
    pkg.load_library().c2r_default_sed(C.byref(sed))
    sed.T_eff = 1.0e5
    thick, thin, _ = pkg.build_tables(sed)
    assert abs(thick[0] / sed.S_star - 1) < 1e-12
    ref = pkg.build_tables()[0]
    # harder spectrum: more photons survive a large optical depth
    assert thick[1700] > ref[1700]
    bad = pkg.SedParams()
    pkg.load_library().c2r_default_sed(C.byref(bad))
    assert pkg.load_library().c2r_build_tables(C.byref(bad), thick.ctypes.data, thin.ctypes.data, 17, None) != 0


def test_heating_tables_equal_reference_bit_for_bit(pkg):
    """c2r_build_heat_tables against the tables of the reference rebuilt with isothermal=.false.
    (radiation_tables.F90:455-543), and the two values that build logs (:222-226)."""
    from tests._util import load_thermal_tables
    tt = load_thermal_tables()
    hk, hn = pkg._capi.build_heat_tables()
    assert np.array_equal(hk, tt["heat_thick"]) and np.array_equal(hn, tt["heat_thin"])
    assert abs(hk[0] / 1.0706414469369616e37 - 1) < 1e-15 and abs(hn[0] / 2.621026881130597e36 - 1) < 1e-15
    # mean photon excess energy of the unattenuated spectrum: between 0 and a few ionization energies
    thick = pkg.build_tables()[0]
    assert 0.1 < hk[0] / thick[0] / (6.62607550000000009e-27 * pkg._capi.ION_FREQ_HI) < 2.0
    assert np.all(np.diff(hk[1:]) <= 0)
    lib = pkg.load_library()
    assert lib.c2r_build_heat_tables(None, 0.0, hk.ctypes.data, hn.ctypes.data, 2001) != 0
