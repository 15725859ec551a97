"""CPU tests of the on-disk formats (c2-ray3dm_amd/fileio.py) against files the reference wrote
(sha256 of its xfrac3D / IonRates3D outputs, tests/golden/refrun32_onesrc.json) and its two
shipped source lists."""
import hashlib
import json
import os
import numpy as np
import pytest
from tests._util import GOLDEN


@pytest.fixture(scope="module")
def fio():
    import __graft_entry__ as g
    return g.load_package().fileio


@pytest.fixture(scope="module")
def refrun():
    return (json.load(open(os.path.join(GOLDEN, "refrun32_onesrc.json"))),
            np.load(os.path.join(GOLDEN, "refrun32_onesrc.npz")))


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_xfrac_and_ionrates_files_are_byte_identical_to_the_reference(fio, refrun, tmp_path):
    m, a = refrun
    for f in m["kept"]:
        z = float(f[len("xfrac3D_"):-4])
        p = fio.write_xfrac3D(str(tmp_path), z, a["xfrac_" + f[len("xfrac3D_"):-4]])
        assert os.path.basename(p) == f
        assert sha(p) == m["sha256"][f]
    zf = m["kept"][0][len("xfrac3D_"):-4]
    # the reference writes real(phih_grid, kind=si); the fixture holds those f32 values
    p = fio.write_IonRates3D(str(tmp_path), float(zf), a["ionrates_" + zf].astype(np.float64))
    assert sha(p) == m["sha256"]["IonRates3D_%s.bin" % zf]
    back = fio.read_sm3d(p)
    assert back.dtype == np.float32 and np.array_equal(back, a["ionrates_" + zf])
    x = fio.read_sm3d(str(tmp_path / m["kept"][0]))
    assert x.dtype == np.float64 and x.shape == (32, 32, 32)


def test_output_file_names_follow_the_slice_redshifts(fio, refrun):
    m, _ = refrun
    # output.F90:188: f6.3 of the redshift; the reference's run produced these names
    names = {"xfrac3D_%s.bin" % fio.zred_str(z) for z in m["slice_redshifts"]}
    assert names <= set(m["outputs"])


def test_source_list_round_trip_and_test_model(fio, tmp_path):
    pos = np.array([[50, 50, 50], [20, 10, 90], [3, 300, -2]], dtype=np.int32)
    nf = np.array([1e9, 1e6, 2.5e7])
    p = str(tmp_path / "test_sources.dat")
    fio.write_sources(p, pos, nf)
    pos2, nf2 = fio.read_sources(p)
    assert np.array_equal(pos, pos2) and np.allclose(nf, nf2, rtol=1e-15)
    # the reference's shipped lists (inputs/test_sources_*.dat), restated as text
    open(p, "w").write("1\n50 50 50 1e57 0.0\n")
    pos3, nf3 = fio.read_sources(p)
    assert pos3.tolist() == [[50, 50, 50]] and nf3[0] == 1e57 / 1.00000000000000004e+48
    # zero-luminosity lines are dropped (sourceprops.F90:363), Fortran d-exponents accepted
    open(p, "w").write("3\n1 2 3 0.0 0.0\n4 5 6 1d55 0.0\n7 8 9 0.0 1e50\n")
    pos4, nf4 = fio.read_sources(p)
    assert pos4.tolist() == [[4, 5, 6], [7, 8, 9]] and nf4[1] == 0.0


@pytest.mark.parametrize("access", ["stream", "sequential"])
def test_density_file_round_trip(fio, tmp_path, access):
    rng = np.random.default_rng(3)
    nd = rng.random((6, 5, 4)).astype(np.float32)
    p = str(tmp_path / "dens.bin")
    fio.write_density(p, nd, access=access)
    raw = open(p, "rb").read()
    if access == "stream":          # density_module.F90:213-243 with densityaccess="stream"
        assert len(raw) == 12 + nd.size * 4
        assert np.frombuffer(raw, dtype=np.int32, count=3).tolist() == [6, 5, 4]
        assert np.frombuffer(raw, dtype=np.float32, offset=12)[1] == nd[1, 0, 0]     # first index fastest
    back = fio.read_density(p, mesh=(6, 5, 4), access=access)
    assert np.array_equal(back, nd)
    with pytest.raises(ValueError):
        fio.read_density(p, mesh=(6, 5, 5), access=access)


def test_records_beyond_2GiB_are_gfortran_subrecords(tmp_path, monkeypatch):
    """A sequential record longer than 2^31-9 bytes (N^3 f64 from mesh 646^3 up) is split into subrecords by
    libgfortran: leading marker negative while another subrecord follows, trailing marker negative when one
    preceded.  Exercised with a small subrecord limit; the default limit is gfortran's."""
    import io
    import __graft_entry__ as g
    fio = g.load_package().fileio
    assert fio.MAX_SUBRECORD == 2 ** 31 - 9
    payload = np.arange(25, dtype=np.float64).tobytes()          # 200 bytes
    f = io.BytesIO()
    fio._rec(f, payload, max_sub=64)                             # 64 + 64 + 64 + 8
    raw = f.getvalue()
    marks = []
    off = 0
    while off < len(raw):
        m = int(np.frombuffer(raw, np.int32, 1, off)[0]); n = abs(m)
        t = int(np.frombuffer(raw, np.int32, 1, off + 4 + n)[0])
        marks.append((m, t)); off += 8 + n
    assert marks == [(-64, 64), (-64, -64), (-64, -64), (8, -8)]
    body, end = fio._read_rec(raw, 0)
    assert bytes(body) == payload and end == len(raw)
    # an ordinary record is unchanged: [n][payload][n]
    f = io.BytesIO(); fio._rec(f, payload)
    assert f.getvalue() == np.int32(200).tobytes() + payload + np.int32(200).tobytes()
    # whole files written with a small limit read back identically
    monkeypatch.setattr(fio, "MAX_SUBRECORD", 1000)
    a = np.random.default_rng(3).random((9, 8, 7))
    fio.write_sm3d(str(tmp_path / "x.bin"), a)
    assert np.array_equal(fio.read_sm3d(str(tmp_path / "x.bin")), a)


def test_cubep3m_density_scaling(tmp_path):
    """scale_density (density_module.F90:246-287) for the coarsened cubep3m fields (density_unit "grid",
    nbody_cubep3m.F90:115-127) and the slice file name (density_module.F90:159-163)."""
    import __graft_entry__ as g
    pkg = g.load_package()
    fio = pkg.fileio
    assert os.path.basename(fio.cubep3m_density_name("/d/", 9.0)) == " 9.000n_all.dat"
    assert os.path.basename(fio.cubep3m_density_name("/d/", 11.546)) == "11.546n_all.dat"
    raw = np.array([[[0.0, 1.0], [2.5, -1.0]], [[100.0, 1e-3], [7.0, 3.0]]], dtype=np.float32)
    nd = fio.scale_density(raw, 8.0, 300, 13824)
    mean0 = 9.20346643016612840e-30 * 4.39999997615814209e-02 / (1.22200000286102295 * 1.67266100000000007e-24)
    conv = mean0 * 300.0 ** 3 / 13824.0 ** 3 * 9.0 ** 3
    assert nd.dtype == np.float32
    assert np.allclose(nd[raw > 0], raw[raw > 0] * conv, rtol=2e-7)
    assert np.allclose(nd[raw <= 0], 0.1 * conv, rtol=2e-7)          # empty cells: 0.1 particles
    # a field whose mean is one coarse cell's share of the fine cells has the mean baryon density
    coarse = np.full((4, 4, 4), (13824.0 / 300.0) ** 3, dtype=np.float32)
    assert abs(float(fio.scale_density(coarse, 0.0, 300, 13824).mean()) / mean0 - 1) < 1e-6
    # through the file format of the slice
    p = str(tmp_path / fio.cubep3m_density_name("", 8.0).strip())
    fio.write_density(p, raw)
    assert np.array_equal(fio.read_density(p, mesh=2), raw)


def test_nonisothermal_dump_cooling_table_and_temperature_files(tmp_path):
    """The two extra records of a non-isothermal iteration dump (evolve.F90:314-317, :372-375), tables/corocool.tab
    (cooling.f90:64-87) and Temper3D / HeatRates3D (output.F90:314-329, :367-378; temperature_restart_init)."""
    import __graft_entry__ as g
    fio = g.load_package().fileio
    from tests.golden.inputs import cooling_table
    n = 6
    rng = np.random.default_rng(3)
    ph, xa, xi, he = (rng.random(n ** 3) for _ in range(4))
    tg = (1e4 * rng.random((n ** 3, 3))).astype(np.float32)
    p = str(tmp_path / "iterdump1.bin")
    fio.write_iteration_dump(p, 7, 1.5e50, ph, xa, xi, mesh=n, phiheat_grid=he, temperature_grid=tg)
    niter, loss, ph2, xa2, xi2, he2, tg2 = fio.read_iteration_dump(p, n, thermal=True)
    assert (niter, loss) == (7, 1.5e50)
    for a, b in ((ph, ph2), (xa, xa2), (xi, xi2), (he, he2), (tg, tg2)):
        assert np.array_equal(a, b)
    # record sizes: 4 | 8 | 3 x 8 N^3 | 8 N^3 | 12 N^3, each framed by two int32 markers
    assert len(open(p, "rb").read()) == 4 + 8 + 4 * 8 * n ** 3 + 12 * n ** 3 + 7 * 8
    assert len(fio.read_iteration_dump(p, n)) == 5                     # an isothermal reader stops after xh_intermed
    text, lt, ll = cooling_table()
    tab = str(tmp_path / "corocool.tab")
    fio.write_cooling_table(tab, lt, ll)
    assert open(tab).read() == text
    lt2, ll2 = fio.read_cooling_table(tab)
    assert np.array_equal(lt, lt2) and np.array_equal(ll, ll2) and len(lt) == 61
    t3 = fio.write_Temper3D(str(tmp_path), 8.515, tg, n)
    assert t3.endswith("Temper3D_8.515.bin")
    back = fio.read_Temper3D(t3, n)
    assert np.array_equal(back[:, 0], tg[:, 0]) and np.array_equal(back[:, 1], tg[:, 0]) and np.array_equal(back[:, 2], tg[:, 0])
    h3 = fio.write_HeatRates3D(str(tmp_path), 8.515, he, n)
    assert np.array_equal(fio.read_sm3d(h3).ravel(order="F"), he.astype(np.float32))
    with pytest.raises(ValueError):
        fio.read_Temper3D(t3, n + 1)
