"""CPU tests: the oracle (oracle/c2ray_oracle.c) against fixtures recorded from the compiled,
unmodified reference (tests/golden/make_golden.py).  The oracle is bit-exact with the reference
on every fixture, so these tests demand equality, not a tolerance."""
import numpy as np
import pytest
from tests._util import F, load_case, oracle_for, expand, relerr, GOLDEN


def test_tables_known_answers(tables):
    thick, thin = tables
    # values the reference logs at start-up (radiation_tables.F90:212-215)
    assert thick.shape == (2001,) and thin.shape == (2001,)
    assert abs(thick[0] - 9.999999999999995e47) < 1e33
    assert abs(thin[0] - 4.567323859246105e47) < 1e33
    assert np.all(np.diff(thick[1:]) <= 0) and np.all(thick >= 0)


def test_constants_match_reference_prints():
    """include/c2ray_constants.h against the widened-f32 semantics of the reference literals."""
    import re, os
    txt = open(os.path.join(GOLDEN, "..", "..", "include", "c2ray_constants.h")).read()
    val = {m.group(1): float(m.group(2).strip("()")) for m in
           re.finditer(r"#define\s+(C2R_\w+)\s+(\(?-?[0-9.]+(?:[eE][-+]?\d+)?\)?)", txt)}
    f32 = lambda x: float(np.float32(x))
    assert val["C2R_PI"] == f32(3.141592654)
    assert val["C2R_SIGMA_HI"] == f32(6.30e-18)
    assert val["C2R_CONVERGENCE_FRACTION"] == f32(1.0e-4)
    assert val["C2R_MIN_FRACTIONAL_CHANGE"] == f32(1.0e-3)
    assert val["C2R_MIN_FRACTION_OF_ATOMS"] == f32(1.0e-8) == val["C2R_DELTHT_SMALL"]
    assert val["C2R_MAX_COLDENSH"] == f32(2e19)
    assert val["C2R_TAU_PHOTO_LIMIT"] == f32(1.0e-7)
    assert val["C2R_SQRT3"] == float(np.sqrt(np.float32(3.0)))
    assert val["C2R_SQRT2"] == float(np.sqrt(np.float32(2.0)))
    assert val["C2R_ABU_C"] == f32(7.1e-7)
    assert val["C2R_DLOGTAU"] == 24.0 / 2000.0
    # cgsconstants.f90:39,76,80,86: temph0 = eth0*ev2k, colh0 = 1.3e-8*fh0*xih0/(eth0*eth0)
    eth0 = f32(13.598)
    assert val["C2R_TEMPH0"] == eth0 * float(np.float32(1.0) / np.float32(8.617e-05))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_cinterp_points(tables, tag):
    from oracle.oracle import Oracle
    pt = np.load(GOLDEN + "/point.npz")
    o = Oracle(32, 1.0, 1.0, 0.0, *tables)
    cd = F(pt["coldens"])
    sp = pt["cinterp_src_" + tag]; r = int(sp[3]); sp = [int(v) for v in sp[:3]]
    ref = pt["cinterp_out_" + tag]
    cnt = 0
    for kk in range(-r, r + 1):
        for jj in range(-r, r + 1):
            for ii in range(-r, r + 1):
                if ii == jj == kk == 0:
                    continue
                c, p = o.cinterp(cd, (sp[0] + ii, sp[1] + jj, sp[2] + kk), sp)
                assert c == ref[cnt, 0] and p == ref[cnt, 1], (ii, jj, kk)
                cnt += 1
    assert cnt == len(ref)


def test_photoion_rates_points(tables):
    from oracle.oracle import Oracle
    pt = np.load(GOLDEN + "/point.npz")
    o = Oracle(32, 1.0, 1.0, 0.0, *tables)
    nf = float(pt["photo_normflux"])
    for row, ref in zip(pt["photo_in"], pt["photo_out"]):
        assert o.photoion_rates(row[0], row[1], row[2], nf) == tuple(ref)
    # a source with zero flux contributes nothing (radiation_photoionrates.F90:126)
    assert o.photoion_rates(1e17, 2e17, 1e72, 0.0) == (0.0, 0.0, 0.0)


def test_doric_points():
    from oracle.oracle import Oracle
    pt = np.load(GOLDEN + "/point.npz")
    for row, ref in zip(pt["doric_in"], pt["doric_out"]):
        assert Oracle.doric(row[0], row[1], row[2], 1.0, row[4], row[5], row[6]) == tuple(ref)


@pytest.mark.parametrize("name", ["sweep32_std_x999", "sweep33_std_x999", "sweep32_bubbles"])
def test_sweep_full_grid(tables, name):
    m, a = load_case(name)
    n = m["n"]
    o = oracle_for(m, tables, n)
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    phih = np.zeros(o.ncell)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert nb == m["sum_nbox"]
    assert loss == m["photon_loss"]
    assert np.array_equal(phih, F(a["phih"]))
    assert np.count_nonzero(phih) == m["phih_nonzero"]
    ns = m["ns_dump"] - 1
    _, _, _, cdo = o.do_source(nd, xh, np.zeros(o.ncell), m["srcpos"][ns], m["normflux"][ns])
    assert np.array_equal(cdo, F(a["coldensh_out"]))


def test_xray_source_type_equals_reference(tables):
    """The second ("P") source type of photoion_rates (radiation_photoionrates.F90:133-137: phi = phi + the lookup in the
    X-ray tables with NormFlux_xray(ns), column 5 of the source list) against the reference rebuilt with
    use_xray_SED=.true. (ref_build.sh 32:xray).  The fixture driver hands the reference its X-ray tables -- the reference's
    own fill integrates an array it never sets (radiation_tables.F90:367) -- here: the reference's power-law tables."""
    m, a = load_case("sweep32_xray")
    n = m["n"]
    assert sum(1 for v in m["normflux_xray"] if v > 0) == 5 and len(m["normflux_xray"]) == len(m["normflux"])
    o = oracle_for(m, tables, n)
    o.enable_xray(a["xray_thick"], a["xray_thin"], m["normflux_xray"])
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    phih = np.zeros(o.ncell)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert nb == m["sum_nbox"] and loss == m["photon_loss"]
    assert np.array_equal(phih, F(a["phih"])) and np.count_nonzero(phih) == m["phih_nonzero"]
    ns = m["ns_dump"] - 1
    _, _, _, cdo = o.do_source(nd, xh, np.zeros(o.ncell), m["srcpos"][ns], m["normflux"][ns], m["normflux_xray"][ns])
    assert np.array_equal(cdo, F(a["coldensh_out"]))
    # ... and the X-ray component matters: without it the rates differ (the stellar-only oracle is NOT the fixture)
    o2 = oracle_for(m, tables, n)
    p2 = np.zeros(o2.ncell)
    o2.pass_sources(nd, xh, p2, m["srcpos"], m["normflux"])
    assert np.max(np.abs(p2 - phih) / np.maximum(phih, 1e-300)) > 0.1
    # a whole evolve3D step with both source types
    m, a = load_case("evolve32_xray")
    s = m["steps"]["step001"]
    o = oracle_for(s, tables, n)
    o.enable_xray(a["xray_thick"], a["xray_thin"], s["normflux_xray"])
    xh = F(a["step001_xh_before"]); nd = F(a["step001_ndens"])
    rep, xav, xint, phih = o.evolve3d(s["dt"], nd, xh, s["srcpos"], s["normflux"])
    assert rep.niter == s["niter"] and list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
    assert np.array_equal(xh, F(a["step001_xh_after"])) and np.array_equal(phih, F(a["step001_phih_grid"]))
    assert rep.sum_nbox_all == s["sum_nbox_all"] and rep.photon_loss_all == s["photon_loss_all"]


def test_sweep32_known_answers(tables):
    """The numbers SURVEY.md s8a records for this case (independent of our fixture files)."""
    m, _ = load_case("sweep32_std_x999")
    assert m["sum_nbox"] == 30 and m["phih_nonzero"] == 32705
    assert abs(m["photon_loss"] / 7.62318326526291e54 - 1) < 1e-14


def test_sweep64_planes(tables):
    m, a = load_case("sweep64_bubbles")
    n = m["n"]
    o = oracle_for(m, tables, n)
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    phih = np.zeros(o.ncell)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert nb == m["sum_nbox"] and loss == m["photon_loss"]
    p3 = phih.reshape((n, n, n), order="F")
    s = [(p - 1) % n for p in m["srcpos"][m["ns_dump"] - 1]]
    assert np.array_equal(p3[s[0]], a["phih_px"])
    assert np.array_equal(p3[:, s[1]], a["phih_py"])
    assert np.array_equal(p3[:, :, s[2]], a["phih_pz"])
    assert np.count_nonzero(phih) == m["phih_nonzero"]
    assert abs(float(np.sum(phih, dtype=np.longdouble)) / m["phih_sum"] - 1) < 1e-15


@pytest.mark.parametrize("name", ["evolve32_onesrc", "evolve32_std_bubbles", "evolve64_std_bubbles"])
def test_evolve3d_steps(tables, name):
    m, a = load_case(name)
    n = m["n"]
    for tag, s in m["steps"].items():
        o = oracle_for(s, tables, n)
        xh = F(a[tag + "_xh_before"]); nd = F(a[tag + "_ndens"])
        rep, xav, xint, phih = o.evolve3d(s["dt"], nd, xh, s["srcpos"], s["normflux"])
        assert rep.niter == s["niter"] and rep.converged == 1
        assert list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
        assert np.array_equal(xh, F(a[tag + "_xh_after"]))
        if tag + "_phih_grid" in a:
            assert np.array_equal(phih, F(a[tag + "_phih_grid"]))
            assert np.array_equal(xav, F(a[tag + "_xh_av"]))
        assert rep.sum_nbox_all == s["sum_nbox_all"]
        assert rep.photon_loss_all == s["photon_loss_all"]
        # photon statistics module variables after the step (photonstatistics.F90)
        for k in ("totrec", "totcollisions", "dh0", "total_ion"):
            assert getattr(rep, k) == s[k], k
        # logged Test-2 values and mean x (printed with 16-17 digits by the reference)
        t2 = np.array(s["log"]["test2"][1:])
        mine = np.array([rep.it_rel1[:rep.niter], rep.it_rel0[:rep.niter]]).T
        assert relerr(mine, t2) < 1e-13
        meanx = np.array(rep.it_sum_xh1[:rep.niter]) / float(np.float32(n ** 3))
        assert relerr(meanx, s["log"]["mean_x"]) < 1e-15
        # average sub-boxes per source line (evolve.F90:249)
        avg = np.array(rep.it_sum_nbox[:rep.niter]) / np.float32(len(s["normflux"]))
        assert relerr(avg, s["log"]["avg_nbox"]) < 1e-6


@pytest.mark.parametrize("name", ["restart32_std_bubbles", "restart32_onesrc"])
def test_evolve3d_restart_from_iteration_dump(tables, name):
    """evolve3D(restart=3): state loaded from an iteration dump, one global pass, then the loop
    (evolve.F90:153-157).  The fixture's dump was written by fileio.write_iteration_dump and read by
    the reference's own start_from_dump."""
    m, a = load_case(name)
    n = m["n"]
    o = oracle_for(m, tables, n)
    xh = F(a["xh_before"]); nd = F(a["ndens"])
    xav, xint, phih = F(a["dump_xh_av"]), F(a["dump_xh_intermed"]), F(a["dump_phih"])
    rep = o.evolve3d_restart(m["dt"], nd, xh, xav, xint, phih, m["srcpos"], m["normflux"], m["dump_niter"])
    assert rep.converged == 1
    k0 = m["dump_niter"]
    # the log holds the restart's own global pass first, then one entry per further iteration
    assert rep.niter - k0 == m["niter_after_restart"] - 1
    assert list(rep.it_conv_flag[k0:rep.niter]) == m["log"]["nonconv"][1:]
    assert np.array_equal(xh, F(a["xh_after"]))
    assert np.array_equal(phih, F(a["phih_grid"]))
    assert np.array_equal(xav, F(a["xh_av"]))


@pytest.mark.parametrize("name", ["evolve32_lls2", "evolve32_lls3", "evolve32_clump5"])
def test_evolve3d_physics_variants(tables, name):
    """Non-default switches of c2ray_parameters.f90 (reference rebuilt with type_of_LLS=2 / 3 or
    type_of_clumping=5): position-dependent LLS column (evolve_point.F90:193), hard barrier at R_max
    (:187-191), clumping grid in doric and the photon statistics (evolve_point.F90:443-445)."""
    m, a = load_case(name)
    n = m["n"]
    s = m["steps"]["step001"]
    o = oracle_for(s, tables, n, lls_grid=a["lls_grid"] if "lls_grid" in a else None,
                   clump_grid=a["clump_grid"] if "clump_grid" in a else None)
    xh = F(a["step001_xh_before"]); nd = F(a["step001_ndens"])
    rep, xav, xint, phih = o.evolve3d(s["dt"], nd, xh, s["srcpos"], s["normflux"])
    assert rep.niter == s["niter"] and list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
    assert np.array_equal(xh, F(a["step001_xh_after"]))
    assert np.array_equal(phih, F(a["step001_phih_grid"]))
    assert rep.sum_nbox_all == s["sum_nbox_all"] and rep.photon_loss_all == s["photon_loss_all"]
    for k in ("totrec", "totcollisions", "dh0", "total_ion"):
        assert getattr(rep, k) == s[k], k


def _check_planes(p3, a, m, prefix, key, exact=True, tol=0.0):
    n = m["n"]
    s = [(q - 1) % n for q in m["srcpos"][m.get("ns_dump", 1) - 1]]
    for tag, sl in (("px", p3[s[0]]), ("py", p3[:, s[1]]), ("pz", p3[:, :, s[2]])):
        ref = a[prefix + "_" + tag]
        if exact:
            assert np.array_equal(sl, ref), (key, tag)
        else:
            assert relerr(sl, ref, floor=1e-60) < tol, (key, tag)


@pytest.mark.parametrize("name", ["sweep128_std_x999", "sweep128_onesrc_x999", "sweep256_3src_x999"])
def test_sweep_at_baseline_grid_sizes(tables, name):
    """128^3 / 256^3 straight from the reference: planes through a source, checksums, and the known
    answers SURVEY.md records for the 128^3 case (sum_nbox = 110, 2 028 081 cells with a rate)."""
    m, a = load_case(name)
    n = m["n"]
    o = oracle_for(m, tables, n)
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    phih = np.zeros(o.ncell)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert nb == m["sum_nbox"] and loss == m["photon_loss"]
    assert np.count_nonzero(phih) == m["phih_nonzero"]
    _check_planes(phih.reshape((n, n, n), order="F"), a, m, "phih", name)
    if name == "sweep128_std_x999":
        assert m["sum_nbox"] == 110 and m["phih_nonzero"] == 2028081


def test_evolve3d_128(tables):
    m, a = load_case("evolve128_std")
    n = m["n"]
    o = oracle_for(m, tables, n)
    xh = F(expand(a["xh_before"], n)); nd = F(expand(a["ndens"], n))
    rep, xav, xint, phih = o.evolve3d(m["dt"], nd, xh, m["srcpos"], m["normflux"])
    assert rep.niter == m["niter"] and list(rep.it_conv_flag[:rep.niter]) == m["log"]["nonconv"]
    _check_planes(xh.reshape((n, n, n), order="F"), a, m, "xh", "xh")
    _check_planes(phih.reshape((n, n, n), order="F"), a, m, "phih", "phih")
    assert np.count_nonzero(phih) == m["phih_nonzero"]
    for k in ("totrec", "totcollisions", "dh0", "total_ion"):
        assert getattr(rep, k) == m[k]


def test_evolve3d_128_one_source(tables):
    """BASELINE configs[1] (128^3, ONE source) as a whole step from a field with a 30-cell ionized bubble: conv_criterion
    = 0, five outer iterations of six sub-boxes each -- bit for bit the reference's history, planes and statistics."""
    import hashlib
    from tests.golden.inputs import bubble_xfield
    m, a = load_case("evolve128_onesrc_bubble")
    n = m["n"]
    o = oracle_for(m, tables, n)
    xh = F(bubble_xfield(n, [(50, 50, 50)], 30.0)); nd = F(expand(a["ndens"], n))
    assert hashlib.sha256(xh.tobytes()).hexdigest() == m["xh_before_sha256"]
    rep, xav, xint, phih = o.evolve3d(m["dt"], nd, xh, m["srcpos"], m["normflux"])
    assert rep.niter == m["niter"] == 5 and list(rep.it_conv_flag[:rep.niter]) == m["log"]["nonconv"]
    assert rep.sum_nbox_all == m["sum_nbox_all"] == 6 and rep.photon_loss_all == m["photon_loss_all"]
    _check_planes(xh.reshape((n, n, n), order="F"), a, m, "xh", "xh")
    _check_planes(phih.reshape((n, n, n), order="F"), a, m, "phih", "phih")
    assert np.count_nonzero(phih) == m["phih_nonzero"]
    for k in ("totrec", "totcollisions", "dh0", "total_ion"):
        assert getattr(rep, k) == m[k]


def test_do_grid_then_global_pass_module_surface(tables):
    """master_slave_processing::do_grid (all sources) followed by evolve_point::evolve0D_global over the mesh, as the
    reference's driver mode 'grid' recorded them: the oracle's pass + global pass, bit for bit."""
    m, a = load_case("grid32_bubbles")
    n = m["n"]
    o = oracle_for(m, tables, n)
    nd, xh = F(a["ndens"]), F(a["xh"])
    phih = np.zeros(o.ncell)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert (loss, nb) == (m["photon_loss"], m["sum_nbox"]) and m["local_chemistry"] is False
    assert np.array_equal(phih, F(a["phih"]))
    xav, xint = xh.copy(), xh.copy()
    assert o.global_pass(m["dt"], nd, xh, xav, xint, phih) == m["conv_flag"]
    assert np.array_equal(xav, F(a["xh_av"])) and np.array_equal(xint, F(a["xh_intermed"]))


# ---- the reference compiled with -DALLFRAC (oracle/ref_build.sh 32:allfrac): both fractions stored ----------------------------

@pytest.mark.parametrize("name", ["sweep32_allfrac", "sweep32_allfrac_zeros"])
def test_allfrac_sweep_reads_the_stored_neutral_fraction(tables, name):
    """One pass on a field whose stored neutral fraction is NOT 1 - x (2e-3 of noise; `_zeros`: also 2 % stored zeros that evolve0D
    raises to epsilon, evolve_point.F90:131-134): column densities, rates and the photon loss of the -DALLFRAC reference, bit for
    bit -- and not what the shipped build's derived neutral fraction gives."""
    m, a = load_case(name)
    n = m["n"]
    nd, xh, xh0 = F(a["ndens"]), F(a["xh"]), F(a["xh0"])
    o = oracle_for(m, tables, n)
    o.enable_allfrac(xh0)
    phih = np.zeros(o.ncell)
    loss, nb, vis = o.pass_sources(nd, xh, phih, m["srcpos"], m["normflux"])
    assert nb == m["sum_nbox"] and loss == m["photon_loss"]
    assert np.array_equal(phih, F(a["phih"]))
    ns = m["ns_dump"]
    _, _, _, cd = o.do_source(nd, xh, np.zeros(o.ncell), m["srcpos"][ns - 1], m["normflux"][ns - 1])
    assert np.array_equal(cd, F(a["coldensh_out"]))
    plain = oracle_for(m, tables, n)
    g = np.zeros(o.ncell)
    plain.pass_sources(nd, xh, g, m["srcpos"], m["normflux"])
    assert relerr(g, phih, floor=1e-60) > 1e-4            # the fixture does tell the two builds apart


def test_allfrac_evolve3d_steps(tables):
    """Whole steps of the -DALLFRAC reference: evolve0D_global reads and writes both fractions (evolve_point.F90:341-346, :394-399),
    Test 2 sums the stored neutral fraction (evolve.F90:179-181), the photon statistics count it (photonstatistics.F90)."""
    m, a = load_case("evolve32_allfrac")
    n = m["n"]
    for tag, s in m["steps"].items():
        o = oracle_for(s, tables, n)
        xh, xh0, nd = F(a[tag + "_xh_before"]), F(a[tag + "_xh_before0"]), F(a[tag + "_ndens"])
        o.enable_allfrac(xh0)
        rep, xav, xint, phih = o.evolve3d(s["dt"], nd, xh, s["srcpos"], s["normflux"])
        assert rep.niter == s["niter"] and rep.converged == 1
        assert list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
        assert np.array_equal(xh, F(a[tag + "_xh_after"])) and np.array_equal(xh0, F(a[tag + "_xh_after0"]))
        assert np.array_equal(phih, F(a[tag + "_phih_grid"]))
        assert np.array_equal(xav, F(a[tag + "_xh_av"])) and np.array_equal(o.xh_av0, F(a[tag + "_xh_av0"]))
        assert np.array_equal(xint, F(a[tag + "_xh_intermed"])) and np.array_equal(o.xh_intermed0, F(a[tag + "_xh_intermed0"]))
        assert rep.sum_nbox_all == s["sum_nbox_all"] and rep.photon_loss_all == s["photon_loss_all"]
        for k in ("totrec", "totcollisions", "dh0", "total_ion"):
            assert getattr(rep, k) == s[k], k
        t2 = np.array(s["log"]["test2"][1:])
        assert relerr(np.array([rep.it_rel1[:rep.niter], rep.it_rel0[:rep.niter]]).T, t2) < 1e-13
