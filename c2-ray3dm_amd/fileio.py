"""On-disk formats around the hot path, so the harness can consume the inputs
the reference consumes and post-processing tools can read what we write:

  * density input      `read_density_file`  density_module.F90:203-243  (nbody_*: densityformat/
                       densityaccess/densityheader): 3 x int32 header + N^3 float32, either a plain
                       stream or Fortran sequential records
  * source lists       `count_or_read_in_sources` sourceprops.F90:259-391: N, then one line per source
                       `i j k col4 [col5 ...]`; Test UV model: NormFlux = col4/S_star (:627-631),
                       sources with sum(col4:) <= 0 are dropped (:363)
  * 3-D outputs        `write_sm3d_dp/si_file_routine` read_sm3d.f90:63-103, used for xfrac3D_z.bin
                       (f64, output.F90:285-317) and IonRates3D_z.bin (f32, output.F90:342-360):
                       Fortran sequential records  [12][n1 n2 n3][12] [nbytes][data][nbytes]
All arrays are (n1,n2,n3) with the first index fastest on disk (Fortran order).
"""
import os
import numpy as np

S_STAR = 1.00000000000000004e+48


MAX_SUBRECORD = 2147483639        # libgfortran: records longer than 2^31-9 bytes are split into subrecords


def _rec(f, payload, max_sub=None):
    """One Fortran sequential record with gfortran's 4-byte markers.  A record longer than 2^31-9 bytes (an N^3
    f64 array from mesh 646^3 up: xfrac3D, iteration dumps) is written as subrecords, as libgfortran does
    (io/transfer.c next_record_w_unf): the LEADING marker of a subrecord is negative when another subrecord
    follows, the TRAILING marker is negative when one preceded it."""
    max_sub = max_sub or MAX_SUBRECORD
    payload = memoryview(payload)
    total, off, first = len(payload), 0, True
    while True:
        n = min(max_sub, total - off)
        more = off + n < total
        f.write(np.int32(-n if more else n).tobytes())
        f.write(payload[off:off + n])
        f.write(np.int32(n if first else -n).tobytes())
        off += n
        first = False
        if not more:
            break


def _read_rec(raw, off):
    """Returns (payload, offset behind the record); joins gfortran subrecords (see _rec)."""
    parts = []
    while True:
        m = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=off)[0])
        n = abs(m)
        parts.append(raw[off + 4: off + 4 + n])
        t = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=off + 4 + n)[0])
        if abs(t) != n or (t < 0) != (len(parts) > 1):
            raise ValueError("corrupt Fortran record")
        off += 8 + n
        if m >= 0:
            break
    return (parts[0] if len(parts) == 1 else b"".join(parts)), off


def write_sm3d(path, a):
    """read_sm3d.f90:63-103: header record shape(1:3), data record."""
    a = np.asarray(a)
    if a.ndim != 3 or a.dtype not in (np.float32, np.float64):
        raise ValueError("sm3d files hold 3-D float32/float64 arrays")
    with open(path, "wb") as f:
        _rec(f, np.asarray(a.shape, dtype=np.int32).tobytes())
        _rec(f, np.asfortranarray(a).tobytes(order="F"))


def read_sm3d(path, dtype=None):
    raw = open(path, "rb").read()
    hdr, off = _read_rec(raw, 0)
    shape = tuple(int(v) for v in np.frombuffer(hdr, dtype=np.int32))
    body, off = _read_rec(raw, off)
    ncell = shape[0] * shape[1] * shape[2]
    if dtype is None:
        dtype = {4: np.float32, 8: np.float64}[len(body) // ncell]
    return np.frombuffer(body, dtype=dtype).reshape(shape, order="F").copy()


def zred_str(zred):
    return "%.3f" % zred                      # output.F90:188  write(zred_str,"(f6.3)") zred_now


def write_xfrac3D(results_dir, zred, xh, mesh=None):
    """output.F90:285-317 stream 2: xh (f64)."""
    path = os.path.join(results_dir, "xfrac3D_%s.bin" % zred_str(zred))
    write_sm3d(path, _as3d(xh, mesh, np.float64))
    return path


def write_IonRates3D(results_dir, zred, phih_grid, mesh=None):
    """output.F90:342-360 stream 3: real(phih_grid, kind=si)."""
    path = os.path.join(results_dir, "IonRates3D_%s.bin" % zred_str(zred))
    write_sm3d(path, _as3d(phih_grid, mesh, np.float64).astype(np.float32))
    return path


def write_Temper3D(results_dir, zred, temperature_grid, mesh):
    """output.F90:314-329 (non-isothermal runs): temperature_grid%current, f32.  temperature_grid is (ncell, 3)."""
    path = os.path.join(results_dir, "Temper3D_%s.bin" % zred_str(zred))
    write_sm3d(path, _as3d(np.asarray(temperature_grid, dtype=np.float32).reshape(-1, 3)[:, 0], mesh, np.float32))
    return path


def read_Temper3D(path, mesh):
    """temperature_restart_init (temperature_module.F90:79-130): the file's field into all three components."""
    t = read_sm3d(path, np.float32)
    want = (mesh,) * 3 if np.isscalar(mesh) else tuple(mesh)
    if t.shape != want:
        raise ValueError("file with temperatures unusable: mesh found in file %r, expected %r" % (t.shape, want))
    return np.repeat(t.ravel(order="F")[:, None], 3, axis=1)


def write_HeatRates3D(results_dir, zred, phiheat_grid, mesh=None):
    """output.F90:367-378 (non-isothermal runs): real(phiheat_grid, kind=si)."""
    path = os.path.join(results_dir, "HeatRates3D_%s.bin" % zred_str(zred))
    write_sm3d(path, _as3d(phiheat_grid, mesh, np.float64).astype(np.float32))
    return path


def read_cooling_table(path):
    """setup_cool (cooling.f90:64-87): rows of (log10 T, log10 Lambda), list-directed.  Returns the two columns;
    the caller derives mintemp = logT[0], dtemp = logT[1] - logT[0] and cie_cool = 10**logL as the reference does."""
    lt, ll = [], []
    for line in open(path):
        cols = line.replace(",", " ").split()
        if len(cols) >= 2:
            lt.append(float(cols[0].replace("d", "e").replace("D", "e")))
            ll.append(float(cols[1].replace("d", "e").replace("D", "e")))
    return np.asarray(lt), np.asarray(ll)


def write_cooling_table(path, logT, logL):
    with open(path, "w") as f:
        for a, b in zip(logT, logL):
            f.write("%5.2f %9.4f\n" % (a, b))


def _as3d(a, mesh, dtype):
    a = np.asarray(a, dtype=dtype)
    if a.ndim == 1:
        mesh = (mesh,) * 3 if np.isscalar(mesh) else tuple(mesh)
        a = a.reshape(mesh, order="F")
    return a


def read_density(path, mesh=None, access="stream", header=True):
    """density_module.F90:203-243.  Returns float32 (n1,n2,n3)."""
    raw = open(path, "rb").read()
    if access == "stream":
        off = 0
        if header:
            shape = tuple(int(v) for v in np.frombuffer(raw, dtype=np.int32, count=3))
            off = 12
        else:
            shape = (mesh,) * 3 if np.isscalar(mesh) else tuple(mesh)
        data = np.frombuffer(raw, dtype=np.float32, count=shape[0] * shape[1] * shape[2], offset=off)
    else:                                      # "sequential": record markers around header and data
        off = 0
        if header:
            hdr, off = _read_rec(raw, 0)
            shape = tuple(int(v) for v in np.frombuffer(hdr, dtype=np.int32))
        else:
            shape = (mesh,) * 3 if np.isscalar(mesh) else tuple(mesh)
        body, off = _read_rec(raw, off)
        data = np.frombuffer(body, dtype=np.float32)
    if mesh is not None:
        want = (mesh,) * 3 if np.isscalar(mesh) else tuple(mesh)
        if shape != want:                      # density_module.F90:218-222
            raise ValueError("file with densities unusable: mesh found in file %r, expected %r" % (shape, want))
    return data.reshape(shape, order="F").copy()


def cubep3m_density_name(dir_dens, redshift):
    """nbody_cubep3m density file of a slice: <dir_dens><z, f6.3>n_all.dat (density_module.F90:159-163)."""
    return os.path.join(dir_dens, "%6.3fn_all.dat" % redshift)


def scale_density(raw, redshift, mesh, n_box, density_unit="grid", cosmological=True):
    """scale_density (density_module.F90:246-287): file values -> comoving... proper gas number density (cm^-3, f32 as
    the reference stores it).  density_unit "grid" (nbody_cubep3m.F90:115: the coarsened cubep3m density, in units of
    the mean mass of a FINE N-body cell): convert = rho_crit_0 Omega_B/(mu m_p) (mesh/n_box)^3 (nbody_cubep3m.F90:127,
    single-precision `real()` of the integers as there), times (1+z)^3 for cosmological runs; "particle" is 8 x that.
    Cells that are empty or negative get 0.1 of a fine cell's mean (:281)."""
    from .testproblem import RHO_CRIT_0, OMEGA_B, MU, M_P
    m32, n32 = np.float32(mesh if np.isscalar(mesh) else mesh[0]), np.float32(n_box)
    # rho_crit_0*Omega_B/(mu*m_p)*real(meshx)**3/(real(n_box)**3): the two cubes in single precision, then left to right
    convert = RHO_CRIT_0 * OMEGA_B / (MU * M_P) * float(m32 * m32 * m32) / float(n32 * n32 * n32)
    if density_unit == "particle":
        convert = 8.0 * convert
    elif density_unit != "grid":
        raise ValueError("density_unit %r not restated (grid, particle)" % density_unit)
    if cosmological:
        convert = convert * (1.0 + redshift) ** 3
    nd = (np.asarray(raw, dtype=np.float32).astype(np.float64) * convert).astype(np.float32)     # ndens is real(si)
    nd[nd <= 0.0] = np.float32(0.1 * convert)
    return nd


def write_density(path, ndens, access="stream", header=True):
    a = np.asarray(ndens, dtype=np.float32)
    with open(path, "wb") as f:
        hdr = np.asarray(a.shape, dtype=np.int32).tobytes()
        body = np.asfortranarray(a).tobytes(order="F")
        if access == "stream":
            if header:
                f.write(hdr)
            f.write(body)
        else:
            if header:
                _rec(f, hdr)
            _rec(f, body)


def read_sources(path, S_star=S_STAR):
    """Test UV model (sourceprops.F90:293-391, 627-631): (srcpos (S,3) int32, NormFlux_stellar (S,))."""
    with open(path) as f:
        n = int(f.readline().split()[0])
        pos, flux = [], []
        for _ in range(n):
            cols = [float(t.replace("d", "e").replace("D", "e")) for t in f.readline().split()]
            if sum(cols[3:]) > 0.0:                                    # :363 summed_weighted_mass > 0
                pos.append([int(cols[0]), int(cols[1]), int(cols[2])])  # :304 int(srclist(1:3))
                flux.append(cols[3] / S_star)                          # :380, :630
    return np.asarray(pos, dtype=np.int32).reshape(-1, 3), np.asarray(flux, dtype=np.float64)


def write_sources(path, srcpos, normflux, S_star=S_STAR):
    with open(path, "w") as f:
        f.write("%d\n" % len(normflux))
        for (i, j, k), nf in zip(srcpos, normflux):
            f.write("%d %d %d %.17e 0.0\n" % (i, j, k, nf * S_star))


def write_iteration_dump(path, niter, photon_loss_all, phih_grid, xh_av, xh_intermed, mesh=None, phiheat_grid=None,
                         temperature_grid=None):
    """write_iteration_dump (evolve.F90:285-324): Fortran sequential records
    niter (int32) | photon_loss_all(NumFreqBnd=1) f64 | phih_grid | xh_av | xh_intermed (N^3 f64 each); non-isothermal
    runs (:314-317) add phiheat_grid (f64) and temperature_grid (N^3 x (current, average, intermed) f32)."""
    with open(path, "wb") as f:
        _rec(f, np.int32(niter).tobytes())
        _rec(f, np.asarray([photon_loss_all], dtype=np.float64).tobytes())
        for a in (phih_grid, xh_av, xh_intermed):
            _rec(f, np.asfortranarray(_as3d(a, mesh, np.float64)).tobytes(order="F"))
        if temperature_grid is not None:
            _rec(f, np.asfortranarray(_as3d(phiheat_grid, mesh, np.float64)).tobytes(order="F"))
            _rec(f, np.ascontiguousarray(np.asarray(temperature_grid, dtype=np.float32).reshape(-1, 3)).tobytes())


def read_iteration_dump(path, mesh, thermal=False):
    """start_from_dump (evolve.F90:328-426).  Returns (niter, photon_loss_all, phih, xh_av, xh_intermed)
    with the arrays flat in Fortran order; thermal=True (non-isothermal dumps, :372-375) appends phiheat_grid and
    temperature_grid ((ncell, 3) f32)."""
    raw = open(path, "rb").read()
    body, off = _read_rec(raw, 0)
    niter = int(np.frombuffer(body, dtype=np.int32)[0])
    body, off = _read_rec(raw, off)
    loss = float(np.frombuffer(body, dtype=np.float64)[0])
    mesh = (mesh,) * 3 if np.isscalar(mesh) else tuple(mesh)
    arrs = []
    for _ in range(3):
        body, off = _read_rec(raw, off)
        a = np.frombuffer(body, dtype=np.float64)
        if a.size != mesh[0] * mesh[1] * mesh[2]:
            raise ValueError("iteration dump does not match the mesh")
        arrs.append(a.copy())
    if thermal:
        body, off = _read_rec(raw, off)
        arrs.append(np.frombuffer(body, dtype=np.float64).copy())
        body, off = _read_rec(raw, off)
        t = np.frombuffer(body, dtype=np.float32)
        if arrs[-1].size != mesh[0] * mesh[1] * mesh[2] or t.size != 3 * mesh[0] * mesh[1] * mesh[2]:
            raise ValueError("iteration dump does not match the mesh")
        arrs.append(t.reshape(-1, 3).copy())
    return (niter, loss) + tuple(arrs)


class PhotonCounts:
    """results/PhotonCounts.out and PhotonCounts2.out (output.F90:149-168, 504-606) and the grand totals
    of update_grandtotal_photonstatistics (photonstatistics.F90:286-293).  One line per output time,
    holding the statistics of the most recent time step."""

    def __init__(self, results_dir):
        self.f1 = open(os.path.join(results_dir, "PhotonCounts.out"), "a")
        self.f2 = open(os.path.join(results_dir, "PhotonCounts2.out"), "a")
        self.f1.write(" Columns: redshift, total number of photons used on the grid, total number of photons produced "
                      "on the grid, photon conservation number, fraction new ionization, fraction recombinations, "
                      "fraction LLS losses (seems to be wrong), fraction photon losses, fraction collisional "
                      "ionization, grand total photon conservation number\n")
        self.f2.write(" Columns: redshift, total number of ions, grand total ionizing photons, mean ionization "
                      "fraction (by volume and mass)\n")
        self.grtotal_ion = 0.0
        self.grtotal_src = 0.0
        self.last = None

    def update(self, phot, photon_loss_all, dt):
        """After every evolve3D: photonstatistics.F90:286-293.  phot: the step's statistics (dict with
        total_ion, totcollisions, totrec, dh0, totalsrc); photon_loss_all in photons/s."""
        self.grtotal_src += phot["totalsrc"]
        self.grtotal_ion += phot["total_ion"] - phot["totcollisions"]
        self.last = dict(phot, photon_loss=photon_loss_all * dt)

    @staticmethod
    def _es(v):
        return "%10.3E" % v

    def write(self, zred, time, totions, volfrac, massfrac):
        """output.F90:504-606 write_photonstatistics at an output time."""
        p = self.last
        if time > 0.0 and p is not None:
            cols = [p["total_ion"], p["totalsrc"], (p["total_ion"] - p["totcollisions"]) / p["totalsrc"],
                    p["dh0"] / p["total_ion"], p["totrec"] / p["total_ion"], 0.0, p["photon_loss"] / p["totalsrc"],
                    p["totcollisions"] / p["total_ion"], self.grtotal_ion / self.grtotal_src]
            self.f1.write("%6.3f" % zred + "".join(self._es(c) for c in cols) + "\n")
            self.f1.flush()
        self.f2.write("%6.3f" % zred + "".join(self._es(c) for c in (totions, self.grtotal_src, volfrac, massfrac)) + "\n")
        self.f2.flush()

    def close(self):
        self.f1.close(); self.f2.close()


def read_photon_counts(path):
    """Numeric rows of a PhotonCounts(2).out file (header lines skipped)."""
    rows = []
    for line in open(path):
        t = line.split()
        try:
            rows.append([float(v) for v in t])
        except ValueError:
            continue
    return rows
