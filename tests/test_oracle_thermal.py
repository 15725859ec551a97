"""The oracle's non-isothermal path (isothermal=.false.: heat_lookuptable, coolin, thermal, the temperature clause of the
global convergence test, phiheat_grid, set_final_temperature_point) against fixtures from the reference REBUILT with
that one parameter changed (oracle/ref_build.sh 32:thermal) and run with the synthetic cooling table of
tests/golden/inputs.cooling_table (the reference repository does not ship tables/corocool.tab).  Equality, as in
tests/test_oracle.py: the restatement follows the reference statement by statement."""
import numpy as np
from tests._util import F, load_case, load_tables, load_thermal_tables, thermal_oracle_for, oracle_for

TAB = load_tables()


def _point_oracle():
    from oracle.oracle import Oracle
    p = np.load("tests/golden/point_thermal.npz")
    tt = load_thermal_tables()
    o = Oracle(32, 1e24, 1e72, 7e16, *TAB)
    o.enable_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"], float(p["thermal_zred"]))
    return o, p


def test_heat_lookuptable_rows():
    o, p = _point_oracle()
    nf = float(p["photo_normflux"])
    got = np.array([o.heat_rate(a, b, v, nf) for a, b, v in p["photo_in"]])
    assert np.array_equal(got, p["heat_out"])
    assert np.count_nonzero(got) > 80 and got.max() > 0


def test_coolin_rows():
    o, p = _point_oracle()
    got = np.array([o.coolin(a, b, t) for a, b, t in p["cool_in"]])
    assert np.array_equal(got, p["cool_out"])


def test_thermal_rows():
    """168 rows: T_initial at and below minitemp (outputs untouched), heating on and off, two time steps."""
    o, p = _point_oracle()
    got = np.array([o.thermal(*r) for r in p["thermal_in"]])
    assert np.array_equal(got, p["thermal_out"])
    assert np.count_nonzero(got[:, 0] == -1.0) == 48          # T_initial = 0.5 and 1.0: thermal.f90:83 skips them


def test_sweep_heating_rates_vs_reference():
    m, a = load_case("sweep32_thermal")
    n = m["n"]
    tg = np.zeros((n ** 3, 3), dtype=np.float32)
    o = thermal_oracle_for(m, TAB, tg, n)
    nd, xh = F(a["ndens"]), F(a["xh"])
    phih = np.zeros(n ** 3)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert nb == m["sum_nbox"] and loss == m["photon_loss"]
    assert np.array_equal(phih, F(a["phih"]))
    assert np.array_equal(o.phiheat, F(a["phiheat"]))
    assert np.count_nonzero(o.phiheat) == np.count_nonzero(phih) > 5000


def test_sweep_with_xray_heating_vs_reference():
    """Both switches at once -- use_xray_SED=.true. and isothermal=.false. (ref_build.sh 32:xraythermal): the X-ray source type
    also heats (heat_lookuptable with the "P" tables, radiation_photoionrates.F90:165-171).  Rates AND heating rates of the
    reference's sweep, bit for bit."""
    m, a = load_case("sweep32_xraythermal")
    n = m["n"]
    o = thermal_oracle_for(m, TAB, np.zeros((n ** 3, 3), dtype=np.float32), n)
    o.enable_xray(a["xray_thick"], a["xray_thin"], m["normflux_xray"])
    o.enable_xray_heat(a["xray_heat_thick"], a["xray_heat_thin"])
    nd, xh = F(a["ndens"]), F(a["xh"])
    phih = np.zeros(n ** 3)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert nb == m["sum_nbox"] and loss == m["photon_loss"]
    assert np.array_equal(phih, F(a["phih"])) and np.array_equal(o.phiheat, F(a["phiheat"]))
    # ... and the X-ray heating is in there: the stellar-only heating rates of the same field differ
    o2 = thermal_oracle_for(m, TAB, np.zeros((n ** 3, 3), dtype=np.float32), n)
    o2.pass_sources(nd, xh, np.zeros(n ** 3), m["srcpos"], m["normflux"])
    assert np.max(np.abs(o2.phiheat - o.phiheat) / np.maximum(o.phiheat, 1e-300)) > 0.1


def test_evolve3d_nonisothermal_vs_reference():
    """Whole steps 1 and 3 of a non-isothermal run: iteration history, xh, Gamma, heating rates and the three temperature
    fields, all equal to the reference's."""
    m, a = load_case("evolve32_thermal")
    n = m["n"]
    for tag, s in m["steps"].items():
        tg = np.ascontiguousarray(a[tag + "_temper_before"]).copy()
        o = thermal_oracle_for(s, TAB, tg, n)
        xh = F(a[tag + "_xh_before"])
        rep, xav, xint, phih = o.evolve3d(s["dt"], F(a[tag + "_ndens"]), xh, s["srcpos"], s["normflux"])
        assert rep.converged and rep.niter == s["niter"]
        assert list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
        assert rep.sum_nbox_all == s["sum_nbox_all"]
        assert np.array_equal(xh, F(a[tag + "_xh_after"]))
        assert np.array_equal(phih, F(a[tag + "_phih_grid"]))
        assert np.array_equal(o.phiheat, F(a[tag + "_phiheat_grid"]))
        assert np.array_equal(tg, a[tag + "_temper_after"])
        assert np.array_equal(tg[:, 0], tg[:, 2])              # set_final_temperature_point
        assert np.max(np.abs(tg[:, 0] / a[tag + "_temper_before"][:, 0] - 1)) > 0.05       # the step did heat/cool cells
        for k in ("totrec", "totcollisions"):
            assert abs(getattr(rep, k) / s[k] - 1) < 1e-13


def test_isothermal_oracle_untouched_by_the_switch():
    """heat_thick == NULL keeps the shipped path: same results as before on an isothermal fixture."""
    m, a = load_case("sweep32_bubbles")
    o = oracle_for(m, TAB, m["n"])
    phih = np.zeros(m["n"] ** 3)
    o.pass_sources(F(a["ndens"]), F(a["xh"]), phih, m["srcpos"], m["normflux"])
    assert np.array_equal(phih, F(a["phih"]))


def _steep():
    p = np.load("tests/golden/point_thermal_steep.npz")
    return p, p["cool_logT"], p["cool_logL"]


def test_coolin_and_thermal_rows_with_the_steep_cooling_curve():
    """The second synthetic cooling table (tests/golden/inputs.cooling_table("steep"): a CIE-like rise of four orders of
    magnitude between 1e4 and 1e5 K, a cold-gas coolant whose extrapolation below the first table row stays large): coolin
    above, inside and BELOW the table's range, and 180 thermal rows of which 36 start at or below minitemp (thermal.f90:83:
    untouched) and 14 are driven to minitemp, pinned there by :147-153 and ended by the 10 000 sub-step cap (:163)
    -- all equal to the reference's values."""
    from oracle.oracle import Oracle
    p, lt, ll = _steep()
    tt = load_thermal_tables()
    o = Oracle(32, 1e24, 1e72, 7e16, *TAB)
    o.enable_thermal(tt["heat_thick"], tt["heat_thin"], lt, ll, float(p["thermal_zred"]))
    got = np.array([o.coolin(a, b, t) for a, b, t in p["cool_in"]])
    assert np.array_equal(got, p["cool_out"])
    assert np.count_nonzero(p["cool_in"][:, 2] < 10.0) >= 5                 # rows below the table's first temperature
    st = o.enable_thermal_stats()
    got = np.array([o.thermal(*r) for r in p["thermal_in"]])
    assert np.array_equal(got, p["thermal_out"])
    assert st[0] == len(got) == 180 and st[1] == 36 and np.count_nonzero(got[:, 0] == -1.0) == 36
    assert st[2] == 14, st                                                   # rows that left through i_heating > 10000
    assert st[3] > 10001 * st[2]
    # a row pinned at the floor ends near gamma1 * minitemp: the floor restores the pressure without dividing by
    # gamma - 1 (thermal.f90:148), so the temperature :175 derives from it is 2/3 K (times the electron-density ratio)
    assert np.count_nonzero(np.abs(got[:, 0] - 2.0 / 3.0) < 0.2) >= 14


def test_evolve3d_nonisothermal_with_the_steep_cooling_curve():
    """A whole non-isothermal step with that curve on a field with cells at and below minitemp, cold dense cells and hot
    cells on the steep part of the curve (inputs.temperature_field_cold): 35 outer iterations; the history, xh, Gamma, the
    heating rates and the three temperature fields equal the reference's -- and the counters say what the step exercised."""
    m, a = load_case("evolve32_thermal_steep")
    p, lt, ll = _steep()
    n, tag = m["n"], "step001"
    s = m["steps"][tag]
    tg = np.ascontiguousarray(a[tag + "_temper_before"]).copy()
    o = oracle_for(s, TAB, n)
    tt = load_thermal_tables()
    o.enable_thermal(tt["heat_thick"], tt["heat_thin"], lt, ll, s["zred"], tg)
    st = o.enable_thermal_stats()
    assert np.count_nonzero(tg[:, 0] <= 1.0) > 10
    xh = F(a[tag + "_xh_before"])
    rep, xav, xint, phih = o.evolve3d(s["dt"], F(a[tag + "_ndens"]), xh, s["srcpos"], s["normflux"])
    assert rep.converged and rep.niter == s["niter"] == 35
    assert list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
    assert np.array_equal(xh, F(a[tag + "_xh_after"]))
    assert np.array_equal(phih, F(a[tag + "_phih_grid"]))
    assert np.array_equal(o.phiheat, F(a[tag + "_phiheat_grid"]))
    assert np.array_equal(tg, a[tag + "_temper_after"])
    assert st[1] > 500 and st[2] > 10000, st          # calls skipped at T <= minitemp; calls ended by the sub-step cap
