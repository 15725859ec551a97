"""ORACLE bindings -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

ctypes wrapper over oracle/liboracle.so (the plain-C restatement in
oracle/c2ray_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


class Cfg(C.Structure):
    _fields_ = [("n", C.c_int * 3), ("dr", C.c_double * 3), ("vol", C.c_double),
                ("coldensh_LLS", C.c_double), ("clumping", C.c_double), ("temper", C.c_double),
                ("S_star", C.c_double), ("thick", C.c_void_p), ("thin", C.c_void_p),
                ("lls_type", C.c_int), ("R_max_LLS", C.c_double), ("lls_grid", C.c_void_p),
                ("clump_grid", C.c_void_p), ("tolw", C.c_void_p),
                ("heat_thick", C.c_void_p), ("heat_thin", C.c_void_p), ("cie_cool", C.c_void_p),
                ("cool_mintemp", C.c_double), ("cool_dtemp", C.c_double), ("zred", C.c_double),
                ("temper_grid", C.c_void_p), ("phiheat", C.c_void_p), ("tolw_heat", C.c_void_p),
                ("xray_thick", C.c_void_p), ("xray_thin", C.c_void_p), ("xray_flux", C.c_void_p),
                ("xray_heat_thick", C.c_void_p), ("xray_heat_thin", C.c_void_p),
                ("xh0", C.c_void_p), ("xh_av0", C.c_void_p), ("xh_intermed0", C.c_void_p),
                ("thermal_stats", C.c_void_p)]


class Report(C.Structure):
    _fields_ = [("niter", C.c_int), ("converged", C.c_int), ("conv_flag", C.c_long),
                ("photon_loss_all", C.c_double), ("sum_nbox_all", C.c_long), ("visited", C.c_long),
                ("it_conv_flag", C.c_long * 128), ("it_rel1", C.c_double * 128),
                ("it_rel0", C.c_double * 128), ("it_sum_nbox", C.c_long * 128),
                ("it_sum_xh1", C.c_double * 128),
                ("totrec", C.c_double), ("totcollisions", C.c_double), ("dh0", C.c_double),
                ("total_ion", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_sum.restype = C.c_double
        _LIB.oracle_global_pass.restype = C.c_long
        _LIB.oracle_do_source.restype = C.c_int
        _LIB.oracle_do_source_x.restype = C.c_int
        _LIB.oracle_heat_rate.restype = C.c_double
        _LIB.oracle_coolin.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Holds the per-step scalars and the tables; arrays are passed per call (Fortran order)."""

    def __init__(self, n, dr, vol, coldensh_LLS, thick, thin, clumping=1.0, temper=1e4,
                 S_star=1.00000000000000004e+48, lls_type=1, R_max_LLS=0.0, lls_grid=None, clump_grid=None):
        n = (n, n, n) if np.isscalar(n) else tuple(n)
        dr = (dr, dr, dr) if np.isscalar(dr) else tuple(dr)
        self.thick = np.ascontiguousarray(thick, dtype=np.float64)
        self.thin = np.ascontiguousarray(thin, dtype=np.float64)
        assert self.thick.size == 2001 and self.thin.size == 2001
        self.lls_grid = None if lls_grid is None else np.ascontiguousarray(lls_grid, dtype=np.float32)
        self.clump_grid = None if clump_grid is None else np.ascontiguousarray(clump_grid, dtype=np.float32)
        self.cfg = Cfg((C.c_int * 3)(*n), (C.c_double * 3)(*dr), vol, coldensh_LLS, clumping,
                       temper, S_star, _p(self.thick), _p(self.thin), lls_type, R_max_LLS,
                       None if self.lls_grid is None else _p(self.lls_grid),
                       None if self.clump_grid is None else _p(self.clump_grid), None)
        self.tolw = None
        self.n = n
        self.ncell = n[0] * n[1] * n[2]

    def clone(self):
        """A second oracle with the same configuration and its own diagnostics (tests trace disjoint source sets in threads:
        the C routines are serial and ctypes releases the GIL)."""
        import copy
        c = copy.copy(self)
        c.cfg = type(self.cfg).from_buffer_copy(self.cfg)
        c.tolw = None
        c.cfg.tolw = None
        return c

    def enable_xray_heat(self, heat_thick, heat_thin):
        """The X-ray type's heating tables (non-isothermal runs: heat_lookuptable "P")."""
        self.xray_heat_thick = np.ascontiguousarray(heat_thick, dtype=np.float64)
        self.xray_heat_thin = np.ascontiguousarray(heat_thin, dtype=np.float64)
        assert self.xray_heat_thick.size == 2001 and self.xray_heat_thin.size == 2001
        self.cfg.xray_heat_thick, self.cfg.xray_heat_thin = self.xray_heat_thick.ctypes.data, self.xray_heat_thin.ctypes.data

    def enable_xray(self, xray_thick, xray_thin, xray_flux):
        """The second source type of photoion_rates (use_xray_SED=.true.): its two tables and NormFlux_xray per source, in the
        order of the source lists later passed to pass_sources / evolve3d (do_source: source number ns -> xray_flux[ns-1])."""
        self.xray_thick = np.ascontiguousarray(xray_thick, dtype=np.float64)
        self.xray_thin = np.ascontiguousarray(xray_thin, dtype=np.float64)
        self.xray_flux = np.ascontiguousarray(xray_flux, dtype=np.float64)
        assert self.xray_thick.size == 2001 and self.xray_thin.size == 2001
        self.cfg.xray_thick, self.cfg.xray_thin = self.xray_thick.ctypes.data, self.xray_thin.ctypes.data
        self.cfg.xray_flux = self.xray_flux.ctypes.data

    def enable_allfrac(self, xh0):
        """A build of the reference with -DALLFRAC: the stored neutral fractions.  xh0 (ncell f64) is the (:,:,:,0) half of xh and
        is updated in place like xh; xh_av0 / xh_intermed0 are allocated here (evolve3d fills them; a lone pass reads xh_av0)."""
        assert xh0.dtype == np.float64 and xh0.size == self.ncell and xh0.flags.c_contiguous
        self.xh0 = xh0
        self.xh_av0 = xh0.copy()
        self.xh_intermed0 = xh0.copy()
        self.cfg.xh0, self.cfg.xh_av0, self.cfg.xh_intermed0 = xh0.ctypes.data, self.xh_av0.ctypes.data, self.xh_intermed0.ctypes.data

    def enable_thermal(self, heat_thick, heat_thin, cool_logT, cool_logL, zred, temper_grid=None):
        """Non-isothermal run (isothermal=.false.): heating tables, the cooling table as setup_cool (cooling.f90:64-87)
        holds it, the redshift of cosmo_cool, and the state arrays -- temper_grid (ncell x 3 f32: current, average,
        intermed; updated in place) and phiheat (allocated here)."""
        self.heat_thick = np.ascontiguousarray(heat_thick, dtype=np.float64)
        self.heat_thin = np.ascontiguousarray(heat_thin, dtype=np.float64)
        assert self.heat_thick.size == 2001 and self.heat_thin.size == 2001 and len(cool_logT) == 61
        # cooling.f90:83  10.0d0**cie_cool: libm's pow (python floats), not numpy's vector pow, which differs in the last bit
        self.cie_cool = np.array([10.0 ** float(v) for v in cool_logL], dtype=np.float64)
        self.cfg.heat_thick, self.cfg.heat_thin = self.heat_thick.ctypes.data, self.heat_thin.ctypes.data
        self.cfg.cie_cool = self.cie_cool.ctypes.data
        self.cfg.cool_mintemp = float(cool_logT[0]); self.cfg.cool_dtemp = float(cool_logT[1]) - float(cool_logT[0])
        self.cfg.zred = zred
        self.phiheat = np.zeros(self.ncell, dtype=np.float64)
        self.cfg.phiheat = self.phiheat.ctypes.data
        if temper_grid is not None:
            self.set_temperature(temper_grid)

    def set_temperature(self, temper_grid):
        assert temper_grid.dtype == np.float32 and temper_grid.size == 3 * self.ncell and temper_grid.flags.c_contiguous
        self.temper_grid = temper_grid
        self.cfg.temper_grid = temper_grid.ctypes.data

    def heat_rate(self, cd_in, cd_out, vol, normflux):
        return lib().oracle_heat_rate(_p(self.heat_thick), _p(self.heat_thin), C.c_double(cd_in), C.c_double(cd_out),
                                      C.c_double(vol), C.c_double(normflux))

    def coolin(self, nucldens, eldens, temp0):
        return lib().oracle_coolin(C.byref(self.cfg), C.c_double(nucldens), C.c_double(eldens), C.c_double(temp0))

    def thermal(self, dt, t_initial, ndens_electron, ndens_atom, h_old1, h_av1, h1, heat):
        """thermal.f90:22; returns (final, average), -1 where the reference leaves its output untouched."""
        tf, ta = C.c_double(-1.0), C.c_double(-1.0)
        lib().oracle_thermal(C.byref(self.cfg), C.c_double(dt), C.c_double(t_initial), C.byref(tf), C.byref(ta),
                             C.c_double(ndens_electron), C.c_double(ndens_atom), C.c_double(h_old1), C.c_double(h_av1),
                             C.c_double(h1), C.c_double(heat))
        return tf.value, ta.value

    def enable_tolerance_weight(self):
        """Checker diagnostic (see oracle_cfg.tolw): accumulate, from now on, W = sum_s (1+tau_in) photo_in /
        (vol_ph n_HI) per cell into self.tolw (zeroed here)."""
        self.tolw = np.zeros(self.ncell, dtype=np.float64)
        self.cfg.tolw = self.tolw.ctypes.data
        return self.tolw

    def enable_thermal_stats(self):
        """Checker diagnostic (oracle_cfg.thermal_stats): counts, from now on, [calls of thermal(), of them skipped because
        T_initial <= minitemp, of them ended by the 10 000 sub-step cap, sub-steps in all]."""
        self.thermal_stats = np.zeros(4, dtype=np.int64)
        self.cfg.thermal_stats = self.thermal_stats.ctypes.data
        return self.thermal_stats

    def enable_heat_tolerance_weight(self):
        """As enable_tolerance_weight, for the heating rate: W_heat = sum_s (1+tau_in) heat_in / vol_ph per cell."""
        self.tolw_heat = np.zeros(self.ncell, dtype=np.float64)
        self.cfg.tolw_heat = self.tolw_heat.ctypes.data
        return self.tolw_heat

    # -- point functions ------------------------------------------------------------------
    def cinterp(self, cdout, pos, src):
        cd, path = C.c_double(), C.c_double()
        lib().oracle_cinterp(_p(cdout), (C.c_int * 3)(*self.n), (C.c_int * 3)(*pos),
                             (C.c_int * 3)(*src), C.byref(cd), C.byref(path))
        return cd.value, path.value

    def photoion_rates(self, cd_in, cd_out, vol, normflux):
        out = (C.c_double * 3)()
        lib().oracle_photoion_rates(_p(self.thick), _p(self.thin), C.c_double(cd_in),
                                    C.c_double(cd_out), C.c_double(vol), C.c_double(normflux), out)
        return tuple(out)

    @staticmethod
    def doric(dt, temp0, rhe, clumping, x1_old, xav1, phih):
        xf = (C.c_double * 2)(1.0 - x1_old, x1_old)
        xa = (C.c_double * 2)(1.0 - xav1, xav1)
        lib().oracle_doric(C.c_double(dt), C.c_double(temp0), C.c_double(rhe), C.c_double(clumping),
                           xf, xa, C.c_double(phih))
        return xf[0], xf[1], xa[0], xa[1]

    # -- sweep ----------------------------------------------------------------------------
    def do_source(self, ndens, xh_av, phih, src, normflux, normflux_xray=0.0):
        """One source; adds into phih (in place).  Returns (nbox, loss, visited, coldensh_out)."""
        cdout = np.zeros(self.ncell, dtype=np.float64)
        loss, vis = C.c_double(), C.c_long()
        nbox = lib().oracle_do_source_x(C.byref(self.cfg), _p(ndens), _p(xh_av), _p(phih), _p(cdout),
                                        (C.c_int * 3)(*[int(v) for v in src]), C.c_double(normflux), C.c_double(normflux_xray),
                                        C.byref(loss), C.byref(vis))
        return nbox, loss.value, vis.value, cdout

    def pass_sources(self, ndens, xh_av, phih, srcpos, normflux, rank=0, npr=1):
        """All sources of one rank (static round-robin, master_slave.F90:85); adds into phih."""
        srcpos = np.ascontiguousarray(srcpos, dtype=np.int32)       # (S,3)
        normflux = np.ascontiguousarray(normflux, dtype=np.float64)
        loss, nb, vis = C.c_double(), C.c_long(), C.c_long()
        lib().oracle_pass_sources(C.byref(self.cfg), _p(ndens), _p(xh_av), _p(phih), _p(srcpos),
                                  _p(normflux), C.c_int(len(normflux)), C.c_int(rank), C.c_int(npr),
                                  C.byref(loss), C.byref(nb), C.byref(vis))
        return loss.value, nb.value, vis.value

    def global_pass(self, dt, ndens, xh, xh_av, xh_intermed, phih):
        return lib().oracle_global_pass(C.byref(self.cfg), C.c_double(dt), _p(ndens), _p(xh),
                                        _p(xh_av), _p(xh_intermed), _p(phih))

    def evolve3d(self, dt, ndens, xh, srcpos, normflux):
        """Full time step.  xh is updated in place.  Returns (report, xh_av, xh_intermed, phih)."""
        srcpos = np.ascontiguousarray(srcpos, dtype=np.int32)
        normflux = np.ascontiguousarray(normflux, dtype=np.float64)
        xh_av = np.empty(self.ncell); xh_int = np.empty(self.ncell); phih = np.zeros(self.ncell)
        rep = Report()
        lib().oracle_evolve3d(C.byref(self.cfg), C.c_double(dt), _p(ndens), _p(xh), _p(xh_av),
                              _p(xh_int), _p(phih), _p(srcpos), _p(normflux),
                              C.c_int(len(normflux)), C.byref(rep))
        return rep, xh_av, xh_int, phih

    def evolve3d_restart(self, dt, ndens, xh, xh_av, xh_intermed, phih, srcpos, normflux, niter0):
        """evolve3D(restart/=0) after start_from_dump loaded (niter0, phih, xh_av, xh_intermed)."""
        srcpos = np.ascontiguousarray(srcpos, dtype=np.int32)
        normflux = np.ascontiguousarray(normflux, dtype=np.float64)
        rep = Report()
        lib().oracle_evolve3d_x(C.byref(self.cfg), C.c_double(dt), _p(ndens), _p(xh), _p(xh_av),
                                _p(xh_intermed), _p(phih), _p(srcpos), _p(normflux),
                                C.c_int(len(normflux)), C.c_int(niter0), C.byref(rep))
        return rep

    def photon_sums(self, ndens, xh_l, xh_r):
        out = (C.c_double * 4)()
        lib().oracle_photon_sums(C.byref(self.cfg), _p(ndens), _p(xh_l), _p(xh_r), out)
        return tuple(out)

    @staticmethod
    def sum(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return lib().oracle_sum(_p(a), C.c_size_t(a.size))
