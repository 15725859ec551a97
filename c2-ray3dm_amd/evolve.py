"""Host-side mirror of the reference's evolve modules, one process per GPU.

Names and argument meaning follow the reference:

    evolve3D(time, dt, restart)         evolve.F90:83
    set_rates_to_zero()                 evolve.F90:430
    pass_all_sources(niter, dt)         evolve.F90:444
    do_grid(dt, niter)                  master_slave.F90:53   (static distribution, :74-96)
    do_source(dt, ns1, niter)           evolve_source.F90:58
    global_pass(dt) -> conv_flag        evolve.F90:499
    mpi_accumulate_grid_quantities()    evolve.F90:577        (RCCL all-reduce via torch.distributed)

State that the reference keeps in module-global arrays (xh, ndens, xh_av, xh_intermed,
phih_grid, srcpos, NormFlux_stellar, dr, vol, coldensh_LLS, clumping) lives in the backend:
`HipBackend` keeps it in HBM (torch tensors bound to the C-ABI context; torch is only the
allocator / stream / collective plumbing).  `Evolve` holds the outer convergence loop, written
once against the small backend interface so the multi-rank logic can be exercised on CPU with a
test double (tests/_cpu_backend.py) -- the product always runs it on `HipBackend`.
"""
import ctypes as C
import os
import time as _time
import numpy as np

from . import _capi
from ._capi import C2RayHipError



def static_source_share(nsrc, rank, npr):
    """0-based indices of the sources rank `rank` of `npr` traces: do ns1=1+rank,NumSrc,npr
    (master_slave.F90:85)."""
    return list(range(rank, nsrc, npr))


def box_cost(nbox, mesh, subbox=5):
    """Cells a source visits when it ends with `nbox` sub-boxes (evolve_source.F90:100-102,135-136)."""
    v = np.ones_like(np.asarray(nbox), dtype=np.int64)
    for n in mesh:
        hr, hl = n // 2 - 1 + n % 2, n // 2
        v = v * (np.minimum(subbox * np.asarray(nbox), hr) + np.minimum(subbox * np.asarray(nbox), hl) + 1)
    return np.where(np.asarray(nbox) > 0, v, 0)


def balanced_source_shares(cost, npr):
    """Longest-processing-time partition of the sources over npr ranks (deterministic: ties by source
    index, then by rank).  Each share is returned in ascending source order."""
    cost = np.asarray(cost, dtype=np.int64)
    order = sorted(range(len(cost)), key=lambda i: (-int(cost[i]), i))
    load = [0] * npr
    shares = [[] for _ in range(npr)]
    for i in order:
        r = min(range(npr), key=lambda k: (load[k], k))
        shares[r].append(i)
        load[r] += int(cost[i]) + 1          # +1: every source costs something, keeps counts even at zero cost
    return [sorted(sh) for sh in shares]


class HipBackend:
    """The HIP path: owns a c2r context on one GPU and the device-resident arrays."""

    def __init__(self, mesh, thick, thin, device=0, scratch_bytes=0, deterministic=False, fast=None, options=None, allfrac=False):
        import torch
        self.torch = torch
        self.lib = _capi.load_library()
        if not torch.cuda.is_available():
            raise C2RayHipError("no GPU visible: the c2ray_hip path has no CPU fallback")
        self.mesh = (mesh,) * 3 if isinstance(mesh, int) else tuple(mesh)
        self.ncell = self.mesh[0] * self.mesh[1] * self.mesh[2]
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        p = _capi.default_params(self.mesh, device)
        p.scratch_bytes = scratch_bytes
        p.deterministic_rates = 1 if deterministic else 0
        # allfrac: the driver this host stands in for was built with -DALLFRAC -- the neutral fractions are stored (arrays
        # xh0, xh_av0, xh_intermed0: context-owned, load()/fetch() by those names), not derived as 1 - x
        p.allfrac = 1 if allfrac else 0
        self.allfrac = bool(allfrac)
        # fast=None: this HOST's switch -- the environment variable C2R_SWEEP_MODE (0/1; how the GPU tests run every case
        # in both modes), else the library default (c2r_default_params: C2R_SWEEP_FAST).  The library itself reads no
        # environment variable: an explicit fast=True/False is what the context gets.
        if fast is None and os.environ.get("C2R_SWEEP_MODE") is not None:
            fast = os.environ["C2R_SWEEP_MODE"] not in ("0", "")
        if fast is not None:
            p.sweep_mode = 1 if fast else 0
        self.params = p
        self.ctx = C.c_void_p()
        rc = self.lib.c2r_create(C.byref(self.ctx), C.byref(p))
        if rc != 0:                      # the context exists even on failure (for its error text): free it
            try:
                self._check(rc, "c2r_create")
            finally:
                self.close()
        thick = np.ascontiguousarray(thick, dtype=np.float64)
        thin = np.ascontiguousarray(thin, dtype=np.float64)
        self._check(self.lib.c2r_set_tables(self.ctx, thick.ctypes.data, thin.ctypes.data, thick.size),
                    "c2r_set_tables")
        # device arrays: torch allocates, the context binds (so torch.distributed can reduce phih)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.ndens = torch.zeros(self.ncell, dtype=torch.float32, device=self.device)
        self.xh = torch.zeros(self.ncell, **f64)
        self.xh_av = torch.zeros(self.ncell, **f64)
        self.xh_intermed = torch.zeros(self.ncell, **f64)
        self.phih_grid = torch.zeros(self.ncell, **f64)
        self._check(self.lib.c2r_bind_device_buffers(self.ctx, self.ndens.data_ptr(), self.xh.data_ptr(),
                                                     self.xh_av.data_ptr(), self.xh_intermed.data_ptr(),
                                                     self.phih_grid.data_ptr()), "c2r_bind_device_buffers")
        self.stream = torch.cuda.current_stream(self.device)
        self._check(self.lib.c2r_set_stream(self.ctx, C.c_void_p(self.stream.cuda_stream)), "c2r_set_stream")
        self.nsrc = 0
        self.rank, self.npr = 0, 1
        self._cb = None
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name, value):
        """c2r_set_option: a switch of the launch schedule (include/c2ray_hip.h has the table), for A/B runs and tests."""
        self._check(self.lib.c2r_set_option(self.ctx, str(name).encode(), float(value)), "c2r_set_option(%s)" % name)

    # -- plumbing -----------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.c2r_last_error(self.ctx) if self.ctx else b""
            raise C2RayHipError("%s failed (%d): %s" % (what, rc, (msg or b"").decode()))

    def get_device(self):
        """c2r_get_device: the HIP device ordinal the context runs on (C2R_DEVICE_AUTO resolved)."""
        d = C.c_int32()
        self._check(self.lib.c2r_get_device(self.ctx, C.byref(d)), "c2r_get_device")
        return d.value

    def info(self):
        """c2r_info: device (and how C2R_DEVICE_AUTO resolved it), sweep mode, rate accumulation, rank."""
        return (self.lib.c2r_info(self.ctx) or b"").decode()

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.c2r_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs ---------------------------------------------------------------------------------
    def set_step(self, dr, vol, coldensh_LLS, clumping=1.0, temper=1e4):
        dr = (dr,) * 3 if np.isscalar(dr) else tuple(dr)
        self.vol = vol
        self._check(self.lib.c2r_set_step(self.ctx, (C.c_double * 3)(*dr), vol, coldensh_LLS, clumping, temper),
                    "c2r_set_step")

    def set_thermal(self, heat_thick, heat_thin, cool_logT, cool_logL, cosmological=True):
        """Non-isothermal run (c2ray_parameters.f90:28 isothermal=.false.; c2r_set_thermal): the heating tables
        (build_heat_tables / radiation_tables.F90:521-543) and the cooling table as the rows of tables/corocool.tab
        (log10 T, log10 Lambda; fileio.read_cooling_table).  The context then owns phiheat_grid and temperature_grid
        (load / fetch by those names; temperature_grid is (ncell, 3) f32: current, average, intermed)."""
        t = _capi.ThermalParams()
        self._check(self.lib.c2r_default_thermal(C.byref(t)), "c2r_default_thermal")
        t.cool_mintemp = float(cool_logT[0]); t.cool_dtemp = float(cool_logT[1]) - float(cool_logT[0])   # cooling.f90:78-79
        t.cool_points = len(cool_logL)
        t.cosmological = 1 if cosmological else 0
        hk = np.ascontiguousarray(heat_thick, dtype=np.float64); hn = np.ascontiguousarray(heat_thin, dtype=np.float64)
        cie = np.array([10.0 ** float(v) for v in cool_logL], dtype=np.float64)      # cooling.f90:83, libm pow as the reference
        self._check(self.lib.c2r_set_thermal(self.ctx, C.byref(t), hk.ctypes.data, hn.ctypes.data, hk.size, cie.ctypes.data),
                    "c2r_set_thermal")
        self.thermal = True

    def set_isothermal(self):
        self._check(self.lib.c2r_set_thermal(self.ctx, None, None, None, 0, None), "c2r_set_thermal")
        self.thermal = False

    def set_redshift(self, zred):
        """cosmology.F90:42 zred at the middle of the step (cosmo_cool); per step, non-isothermal runs."""
        self._check(self.lib.c2r_set_redshift(self.ctx, zred), "c2r_set_redshift")

    def set_lls(self, type_of_LLS=1, lls_grid=None, R_max_LLS=0.0):
        g = None if lls_grid is None else _flat(lls_grid, np.float32)
        self._check(self.lib.c2r_set_lls(self.ctx, type_of_LLS, None if g is None else g.ctypes.data, R_max_LLS),
                    "c2r_set_lls")

    def set_clumping_grid(self, clump_grid=None):
        g = None if clump_grid is None else _flat(clump_grid, np.float32)
        self._check(self.lib.c2r_set_clumping_grid(self.ctx, None if g is None else g.ctypes.data),
                    "c2r_set_clumping_grid")

    def set_sources(self, srcpos, normflux):
        srcpos = np.ascontiguousarray(srcpos, dtype=np.int32).reshape(-1, 3)
        normflux = np.ascontiguousarray(normflux, dtype=np.float64)
        assert len(srcpos) == len(normflux)
        self.nsrc = len(normflux)
        self.share = None
        self.normflux_sum = float(np.sum(normflux))
        self._check(self.lib.c2r_set_sources(self.ctx, srcpos.ctypes.data, normflux.ctypes.data, self.nsrc),
                    "c2r_set_sources")

    def set_xray(self, xray_thick=None, xray_thin=None, normflux_xray=None):
        """The second source type of photoion_rates (use_xray_SED=.true.): its two tables (None: off) and NormFlux_xray per
        source of the current list (after set_sources; None: leave as it is).  Isothermal runs only."""
        if xray_thick is None:
            self._check(self.lib.c2r_set_xray_tables(self.ctx, None, None, 0), "c2r_set_xray_tables")
            return
        if xray_thin is not None:
            k = np.ascontiguousarray(xray_thick, dtype=np.float64); t = np.ascontiguousarray(xray_thin, dtype=np.float64)
            self._check(self.lib.c2r_set_xray_tables(self.ctx, k.ctypes.data, t.ctypes.data, len(k)), "c2r_set_xray_tables")
        if normflux_xray is not None:
            f = np.ascontiguousarray(normflux_xray, dtype=np.float64)
            self._check(self.lib.c2r_set_xray_sources(self.ctx, f.ctypes.data, len(f)), "c2r_set_xray_sources")

    def set_xray_heat(self, heat_thick, heat_thin):
        """The X-ray source type's heating tables (c2r_set_xray_heat_tables): non-isothermal contexts need them before a pass."""
        k = np.ascontiguousarray(heat_thick, dtype=np.float64); t = np.ascontiguousarray(heat_thin, dtype=np.float64)
        self._check(self.lib.c2r_set_xray_heat_tables(self.ctx, k.ctypes.data, t.ctypes.data, len(k)), "c2r_set_xray_heat_tables")

    def set_source_share(self, indices=None):
        """Explicit list of this rank's sources (0-based) instead of the static stride; None resets."""
        if indices is None:
            self._check(self.lib.c2r_set_source_share(self.ctx, None, -1), "c2r_set_source_share")
            self.share = None
        else:
            idx = np.ascontiguousarray(indices, dtype=np.int32)
            self._check(self.lib.c2r_set_source_share(self.ctx, idx.ctypes.data, len(idx)), "c2r_set_source_share")
            self.share = idx

    def set_balance(self, on=True):
        """Cost-balanced source shares computed inside the library from the previous pass (c2r_set_balance)."""
        self._check(self.lib.c2r_set_balance(self.ctx, 1 if on else 0), "c2r_set_balance")

    def local_sources(self):
        """0-based global indices of the sources this rank sweeps (swept in the last pass): static stride,
        explicit share or the library's balanced share (c2r_source_share)."""
        n = C.c_int32()
        idx = np.zeros(max(1, self.nsrc), dtype=np.int32)
        self._check(self.lib.c2r_source_share(self.ctx, idx.ctypes.data, idx.size, C.byref(n)), "c2r_source_share")
        return idx[:n.value].astype(np.int64)

    def last_nbox(self):
        n = len(self.local_sources())
        out = np.zeros(n, dtype=np.int32)
        self._check(self.lib.c2r_last_nbox(self.ctx, out.ctypes.data, n), "c2r_last_nbox")
        return out

    def set_rank(self, rank, npr, allreduce=None):
        """allreduce(tensor): in-place SUM over ranks of a 1-D f64 device tensor."""
        self.rank, self.npr = rank, npr
        if npr > 1:
            torch = self.torch
            known = {self.phih_grid.data_ptr(): self.phih_grid}

            def _cb(user, ptr, count, stream):
                try:
                    t = known.get(ptr)
                    if t is None:
                        t = torch.as_tensor(_DevView(ptr, count), device=self.device)
                    # reduce ON the stream the library hands over (its kernels before and after are ordered on it)
                    if stream:
                        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                            allreduce(t)
                    else:
                        allreduce(t)
                    return 0
                except Exception as exc:      # never unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = _capi.ALLREDUCE_FN(_cb)
        else:
            self._cb = _capi.ALLREDUCE_FN(0)
        self._check(self.lib.c2r_set_rank(self.ctx, rank, npr, self._cb, None), "c2r_set_rank")

    def set_source_queue(self, next_sources=None, chunk=64):
        """c2r_set_source_queue (do_grid_master / do_grid_slave, master_slave.F90:124-330): the pass asks
        next_sources(user, pass_id, want, first, count) -- first[0], count[0] are the outputs -- for `chunk` more sources whenever it
        has swept what it had; None returns to the fixed rules."""
        if next_sources is None:
            self._queue_cb = None
            self._check(self.lib.c2r_set_source_queue(self.ctx, None, None, 0), "c2r_set_source_queue")
            return
        self._queue_cb = _capi.NEXT_SOURCES_FN(next_sources)
        self._check(self.lib.c2r_set_source_queue(self.ctx, C.cast(self._queue_cb, C.c_void_p), None, int(chunk)), "c2r_set_source_queue")

    def set_exchange_overlap(self, on=True):
        """c2r_set_exchange_overlap: passes that are followed by an all-reduce of the whole grid run as two halves, the first
        half's all-reduce travelling while the second is swept (allreduce_rates afterwards only refreshes the sub-box list)."""
        self._check(self.lib.c2r_set_exchange_overlap(self.ctx, 1 if on else 0), "c2r_set_exchange_overlap")

    def set_slab_chemistry(self, reduce_scatter=None, allgather=None):
        """Slab chemistry of the native loop (c2r_set_slab_chemistry, include/c2ray_hip.h): the rates are reduce-scattered
        by z-slabs, every rank runs the global pass on its slab, the pass's outputs are all-gathered.
        reduce_scatter(tensor_f64, offsets, counts): in place, afterwards rank r's slab [offsets[r], +counts[r]) holds
        the SUM over ranks (other parts undefined); allgather(tensor_u8, offsets, counts): in place, every rank's byte
        slab is valid in its own array on entry, all are valid everywhere on return.  None, None switches it off."""
        if reduce_scatter is None or allgather is None:
            self._rs = self._ag = None
            self._check(self.lib.c2r_set_slab_chemistry(self.ctx, None, None, None), "c2r_set_slab_chemistry")
            return
        torch = self.torch

        def _wrap(fn, typestr):
            def _cb(user, ptr, off, cnt, nranks, stream):
                try:
                    offs = [off[i] for i in range(nranks)]
                    cnts = [cnt[i] for i in range(nranks)]
                    t = torch.as_tensor(_DevView(ptr, offs[-1] + cnts[-1], typestr), device=self.device)
                    if stream:
                        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                            fn(t, offs, cnts)
                    else:
                        fn(t, offs, cnts)
                    return 0
                except Exception:                 # never unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1
            return _capi.SLAB_FN(_cb)
        self._rs, self._ag = _wrap(reduce_scatter, "<f8"), _wrap(allgather, "|u1")
        self._check(self.lib.c2r_set_slab_chemistry(self.ctx, C.cast(self._rs, C.c_void_p), C.cast(self._ag, C.c_void_p),
                                                    None), "c2r_set_slab_chemistry")

    _NEUTRAL = {"xh0": _capi.GRID_XH0, "xh_av0": _capi.GRID_XH_AV0, "xh_intermed0": _capi.GRID_XH_INTERMED0}

    def _neutral_tensor(self, name):
        """One of the stored neutral-fraction arrays (allfrac contexts; context-owned) as a torch tensor sharing the device memory."""
        ptr = C.c_void_p()
        self._check(self.lib.c2r_device_ptr(self.ctx, self._NEUTRAL[name], C.byref(ptr)), "c2r_device_ptr")
        return self.torch.as_tensor(_DevView(ptr.value, self.ncell), device=self.device)

    def load(self, ndens=None, xh=None, xh_av=None, xh_intermed=None, phih_grid=None, phiheat_grid=None, temperature_grid=None,
             xh0=None, xh_av0=None, xh_intermed0=None):
        """Host (numpy, Fortran-order flat or (N,N,N) with i fastest when ravelled 'F') -> HBM.
        temperature_grid: (ncell, 3) f32 as temperature_module.F90:35 lays it out, or one field (K) for all three.
        xh0 / xh_av0 / xh_intermed0: the stored neutral fractions of an allfrac context."""
        torch = self.torch
        for name, a in (("xh0", xh0), ("xh_av0", xh_av0), ("xh_intermed0", xh_intermed0)):
            if a is not None:
                h = _flat(a, np.float64)
                self._check(self.lib.c2r_upload(self.ctx, self._NEUTRAL[name], h.ctypes.data), "c2r_upload")
        if phiheat_grid is not None:
            a = _flat(phiheat_grid, np.float64)
            self._check(self.lib.c2r_upload(self.ctx, _capi.GRID_PHIHEAT, a.ctypes.data), "c2r_upload")
        if temperature_grid is not None:
            t = np.asarray(temperature_grid)
            if t.size == self.ncell:
                t = np.repeat(_flat(t, np.float32)[:, None], 3, axis=1)
            t = np.ascontiguousarray(t.reshape(self.ncell, 3), dtype=np.float32)
            self._check(self.lib.c2r_upload(self.ctx, _capi.GRID_TEMPER, t.ctypes.data), "c2r_upload")
        if ndens is not None:
            self.ndens.copy_(torch.from_numpy(_flat(ndens, np.float32)))
        for name, a in (("xh", xh), ("xh_av", xh_av), ("xh_intermed", xh_intermed), ("phih_grid", phih_grid)):
            if a is not None:
                getattr(self, name).copy_(torch.from_numpy(_flat(a, np.float64)))

    def fetch(self, name):
        if name == "phiheat_grid":
            a = np.empty(self.ncell, dtype=np.float64)
            self._check(self.lib.c2r_download(self.ctx, _capi.GRID_PHIHEAT, a.ctypes.data), "c2r_download")
            return a
        if name == "temperature_grid":
            a = np.empty((self.ncell, 3), dtype=np.float32)
            self._check(self.lib.c2r_download(self.ctx, _capi.GRID_TEMPER, a.ctypes.data), "c2r_download")
            return a
        if name in self._NEUTRAL:
            a = np.empty(self.ncell, dtype=np.float64)
            self._check(self.lib.c2r_download(self.ctx, self._NEUTRAL[name], a.ctypes.data), "c2r_download")
            return a
        return getattr(self, name).cpu().numpy()

    # -- backend interface used by Evolve ---------------------------------------------------------
    def begin_step(self):
        self.xh_av.copy_(self.xh)                 # evolve.F90:145
        self.xh_intermed.copy_(self.xh)           # evolve.F90:146
        if self.allfrac:                          # :142-143: all of (:,:,:,:)
            x0 = self._neutral_tensor("xh0")
            self._neutral_tensor("xh_av0").copy_(x0); self._neutral_tensor("xh_intermed0").copy_(x0)

    def accept(self):
        self.xh.copy_(self.xh_intermed)           # evolve.F90:218
        if self.allfrac:                          # :216
            self._neutral_tensor("xh0").copy_(self._neutral_tensor("xh_intermed0"))
        self._check(self.lib.c2r_set_final_temperature(self.ctx), "c2r_set_final_temperature")     # :220 (no-op when isothermal)

    def sum_xh_intermed(self):
        s = C.c_double()
        self._check(self.lib.c2r_sum(self.ctx, _capi.GRID_XH_INTERMED, C.byref(s)), "c2r_sum")
        return s.value

    def allreduce_rates(self):
        """mpi_accumulate_grid_quantities' array part through the library (c2r_allreduce_rates: the whole grids, or only the
        sources' sub-boxes while those are a small part of the mesh) -- over the all-reduce given to set_rank."""
        self._check(self.lib.c2r_allreduce_rates(self.ctx), "c2r_allreduce_rates")

    def exchange_stats(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.c2r_exchange_stats(self.ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "c2r_exchange_stats")
        return {"calls": a.value, "sparse_calls": b.value, "bytes_last": c.value, "bytes_total": d.value}

    def zero_rates(self):
        self._check(self.lib.c2r_zero_rates(self.ctx), "c2r_zero_rates")

    def pass_sources(self):
        loss, nb, vis = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self.lib.c2r_pass_sources(self.ctx, C.byref(loss), C.byref(nb), C.byref(vis)),
                    "c2r_pass_sources")
        return loss.value, nb.value, vis.value

    def do_source(self, ns1, want_coldens=False):
        cd = np.empty(self.ncell, dtype=np.float64) if want_coldens else None
        loss, nbox, vis = C.c_double(), C.c_int32(), C.c_int64()
        self._check(self.lib.c2r_do_source(self.ctx, ns1, cd.ctypes.data if want_coldens else None,
                                           C.byref(loss), C.byref(nbox), C.byref(vis)), "c2r_do_source")
        return nbox.value, loss.value, vis.value, cd

    def photon_sums(self, which_l, which_r):
        """photonstatistics.F90 mesh sums: arrays by name ('xh', 'xh_av', 'xh_intermed')."""
        ids = {"xh": _capi.GRID_XH, "xh_av": _capi.GRID_XH_AV, "xh_intermed": _capi.GRID_XH_INTERMED}
        out = (C.c_double * 4)()
        self._check(self.lib.c2r_photon_sums(self.ctx, ids[which_l], ids[which_r], C.byref(out)), "c2r_photon_sums")
        return tuple(out)

    def global_pass(self, dt):
        conv, s = C.c_int64(), C.c_double()
        self._check(self.lib.c2r_global_pass(self.ctx, dt, C.byref(conv), C.byref(s)), "c2r_global_pass")
        return conv.value, s.value

    def iterate(self, dt):
        """One outer iteration on a single rank (c2r_iterate): zero_rates + pass_sources + global_pass with one host wait
        where the sources are few.  Returns (photon_loss, sum_nbox, visited, conv_flag, sum_xh1)."""
        loss, nb, vis, conv, s = C.c_double(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        self._check(self.lib.c2r_iterate(self.ctx, dt, C.byref(loss), C.byref(nb), C.byref(vis), C.byref(conv), C.byref(s)),
                    "c2r_iterate")
        return loss.value, nb.value, vis.value, conv.value, s.value

    def rates_tensor(self):
        return self.phih_grid

    def heat_tensor(self):
        """phiheat_grid (context-owned) as a torch tensor sharing the device memory, or None when isothermal."""
        if not getattr(self, "thermal", False):
            return None
        ptr = C.c_void_p()
        self._check(self.lib.c2r_device_ptr(self.ctx, _capi.GRID_PHIHEAT, C.byref(ptr)), "c2r_device_ptr")
        return self.torch.as_tensor(_DevView(ptr.value, self.ncell), device=self.device)

    def scalars_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64, device=self.device)

    # -- whole-step entry point of the C ABI (the loop runs in C++) ---------------------------------
    def evolve3d_native(self, dt, restart_niter=None, restart_photon_loss=0.0):
        rep = _capi.Report()
        if restart_niter is None:
            self._check(self.lib.c2r_evolve3d_dev(self.ctx, dt, C.byref(rep)), "c2r_evolve3d_dev")
        else:
            self._check(self.lib.c2r_evolve3d_restart_dev(self.ctx, dt, restart_niter, restart_photon_loss,
                                                          C.byref(rep)), "c2r_evolve3d_restart_dev")
        return rep

    def set_iteration_hook(self, fn=None):
        """fn(niter, photon_loss_all) -> None, called by the native loop after every outer iteration at the point
        where the reference decides on an iteration dump (evolve.F90:271-275; c2r_set_iteration_hook).  The arrays
        xh_av, xh_intermed, phih_grid of that iteration are in place.  None removes the hook."""
        if fn is None:
            self._hook = None
            self._check(self.lib.c2r_set_iteration_hook(self.ctx, None, None), "c2r_set_iteration_hook")
            return

        def _cb(user, niter, loss):
            try:
                fn(niter, loss)
                return 0
            except Exception:                 # never unwind through C
                import traceback
                traceback.print_exc()
                return 1
        self._hook = _capi.ITERATION_FN(_cb)
        self._check(self.lib.c2r_set_iteration_hook(self.ctx, C.cast(self._hook, C.c_void_p), None), "c2r_set_iteration_hook")

    def selftest(self):
        bad = C.c_int64()
        self._check(self.lib.c2r_selftest(self.ctx, C.byref(bad)), "c2r_selftest")
        return bad.value

    def profile(self, mode=1):
        """0/False off, 1/True per-launch events, 2 one event pair per sub-box (see c2r_profile)."""
        self._check(self.lib.c2r_profile(self.ctx, int(mode)), "c2r_profile")

    def profile_read(self):
        a, b, c, d = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
        self._check(self.lib.c2r_profile_read(self.ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(d)),
                    "c2r_profile_read")
        return {"sweep_ms": a.value, "sweep_launches": b.value, "chem_ms": c.value, "chem_launches": d.value}


class _DevView:
    """__cuda_array_interface__ view of `count` elements (f64 unless typestr says otherwise) at a raw device pointer."""

    def __init__(self, ptr, count, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def _flat(a, dtype):
    a = np.asarray(a)
    if a.ndim == 3:
        a = np.asfortranarray(a).ravel(order="F")
    return np.ascontiguousarray(a, dtype=dtype)


def slab_collectives(comm):
    """(reduce_scatter, allgather) for HipBackend.set_slab_chemistry over a torch.distributed-like comm: one reduce per
    slab to its owner (RCCL: ncclReduce, as c2ray_rccl.cpp groups them), one broadcast per slab from its owner."""
    no_device_reduce = getattr(comm, "get_backend", lambda: "")() == "gloo"   # gloo: all_reduce / broadcast only on device tensors

    def reduce_scatter(t, offs, cnts):
        if no_device_reduce:
            comm.all_reduce(t)        # a superset of the contract: every slab summed everywhere
            return
        for r, (o, c) in enumerate(zip(offs, cnts)):
            if c:
                comm.reduce(t[o:o + c], dst=r)

    def allgather(t, offs, cnts):
        for r, (o, c) in enumerate(zip(offs, cnts)):
            if c:
                comm.broadcast(t[o:o + c], src=r)
    return reduce_scatter, allgather


class Evolve:
    """The outer loop of one time step over a backend, with the reference's procedure names."""

    def __init__(self, backend, comm=None, balance=False, slab=False):
        """comm: None (single process) or a torch.distributed-like module/object exposing
        get_rank(), get_world_size(), all_reduce(tensor) with SUM semantics.
        slab: slab chemistry for the NATIVE loop (backend.evolve3d_native): needs comm.reduce and
        comm.broadcast as torch.distributed has them (a backend without device-tensor reduce -- gloo -- falls back
        to an all-reduce of the whole array, which satisfies the reduce-scatter contract).
        balance: re-partition the sources over the ranks before every pass by the cost each had in the
        previous pass (volume of its final sub-box) instead of the static 1+rank,NumSrc,npr rule."""
        self.b = backend
        self.comm = comm
        self.balance = balance
        self.nbox_per_source = None
        self.rank = comm.get_rank() if comm is not None else 0
        self.npr = comm.get_world_size() if comm is not None else 1
        if hasattr(backend, "set_rank"):
            backend.set_rank(self.rank, self.npr, (lambda t: comm.all_reduce(t)) if comm is not None else None)
        # the HIP backend balances inside the library (c2r_set_balance: the same LPT rule, one small all-reduce
        # through the same callback), so the Fortran/C hosts get it too; do_grid's Python version below serves
        # backends without it (the CPU test double)
        self.slab = bool(slab) and self.npr > 1
        if self.slab:
            backend.set_slab_chemistry(*slab_collectives(comm))
        self._lib_balance = balance and hasattr(backend, "set_balance")
        if hasattr(backend, "set_balance"):
            backend.set_balance(self._lib_balance)
        self.sum_nbox = 0
        self.sum_nbox_all = 0
        self.photon_loss = 0.0
        self.photon_loss_all = 0.0
        self.visited = 0
        # wall time of this rank inside the three phases of `iteration` (several ranks): the sweep of its own sources, the
        # exchange (all-reduce of the rates + the scalar pair: waiting for the slowest rank shows up here), the global pass
        self.phase_seconds = {"sweep": 0.0, "exchange": 0.0, "chem": 0.0}
        self.log = []
        self.dump_dir = "./"               # file_admin.f90:23
        self.dump_interval_s = 15.0 * 60   # evolve.F90:260 (None: never)
        self._ndump = 0

    # evolve.F90:430
    def set_rates_to_zero(self):
        self.b.zero_rates()
        self.photon_loss = 0.0

    # master_slave.F90:53 -> :74 do_grid_static; the per-source loop and do_source live in the
    # backend (c2r_pass_sources), which traces this rank's share 1+rank, 1+rank+npr, ...
    def do_grid(self, dt, niter):
        if self._lib_balance:
            loss, nb, vis = self.b.pass_sources()
            self.photon_loss += loss
            self.sum_nbox += nb
            self.visited += vis
            return
        if self.nbox_per_source is not None and len(self.nbox_per_source) != self.b.nsrc:
            self.nbox_per_source = None        # the source list changed (new slice): back to the static rule for one pass
        if self.balance and self.npr > 1 and self.nbox_per_source is not None:
            cost = box_cost(self.nbox_per_source, self.b.mesh)
            shares = balanced_source_shares(cost, self.npr)
            assert sorted(i for sh in shares for i in sh) == list(range(self.b.nsrc))
            self.b.set_source_share(shares[self.rank])
        elif self.balance and getattr(self.b, "share", None) is not None:
            self.b.set_source_share(None)
        loss, nb, vis = self.b.pass_sources()
        if self.balance and self.npr > 1:
            # every rank learns every source's sub-box count: own entries, zero elsewhere, summed
            mine = np.zeros(self.b.nsrc, dtype=np.float64)
            mine[self.b.local_sources()] = self.b.last_nbox()
            t = self.b.scalars_tensor(mine.tolist())
            self.comm.all_reduce(t)
            self.nbox_per_source = np.rint(np.asarray(t.tolist())).astype(np.int64)
        self.photon_loss += loss
        self.sum_nbox += nb
        self.visited += vis

    # evolve_source.F90:58 (one source, mainly for tests)
    def do_source(self, dt, ns1, niter):
        nbox, loss, vis, _ = self.b.do_source(ns1)
        self.photon_loss += loss
        self.sum_nbox += nbox
        self.visited += vis
        return nbox

    # evolve.F90:577
    def mpi_accumulate_grid_quantities(self):
        if self.npr > 1:
            if hasattr(self.b, "allreduce_rates"):
                self.b.allreduce_rates()                                       # :599, :604-609 (packed sub-boxes while they are few)
            else:
                self.comm.all_reduce(self.b.rates_tensor())                    # :599 phih_grid
                heat = self.b.heat_tensor() if hasattr(self.b, "heat_tensor") else None
                if heat is not None:
                    self.comm.all_reduce(heat)                                 # :604-609 phiheat_grid
            t = self.b.scalars_tensor([self.photon_loss, float(self.sum_nbox)])  # :587, :612
            self.comm.all_reduce(t)
            vals = t.tolist()
            self.photon_loss_all, self.sum_nbox_all = vals[0], int(round(vals[1]))
        else:
            self.photon_loss_all, self.sum_nbox_all = self.photon_loss, self.sum_nbox

    # evolve.F90:444
    def pass_all_sources(self, niter, dt):
        self.sum_nbox = 0
        self.do_grid(dt, niter)
        self.mpi_accumulate_grid_quantities()

    # evolve.F90:499
    def global_pass(self, dt):
        return self.b.global_pass(dt)

    # evolve.F90:243-269 on one rank, nothing in between (no collective, no dump due): the backend's single call
    def iteration(self, niter, dt):
        """set_rates_to_zero + pass_all_sources + global_pass -> (conv_flag, sum_xh1).  A single rank with the HIP backend
        goes through c2r_iterate (one host wait per iteration where the sources are few); otherwise the three steps."""
        if self.npr == 1 and not self.balance and hasattr(self.b, "iterate"):
            loss, nb, vis, conv, s1 = self.b.iterate(dt)
            self.photon_loss = self.photon_loss_all = loss
            self.sum_nbox = self.sum_nbox_all = nb
            self.visited += vis
            self.nbox_per_source = None
            return conv, s1
        import time as _time
        t0 = _time.perf_counter()
        self.set_rates_to_zero()
        self.sum_nbox = 0
        self.do_grid(dt, niter)                       # (blocks until this rank's sweep has run)
        t1 = _time.perf_counter()
        self.mpi_accumulate_grid_quantities()         # (reads the reduced scalar pair back: blocks until the exchange has run)
        t2 = _time.perf_counter()
        out = self.global_pass(dt)
        t3 = _time.perf_counter()
        self.phase_seconds["sweep"] += t1 - t0; self.phase_seconds["exchange"] += t2 - t1; self.phase_seconds["chem"] += t3 - t2
        return out

    # evolve.F90:285-324
    def write_iteration_dump(self, niter):
        from . import fileio
        self._ndump += 1
        name = "iterdump2.bin" if self._ndump % 2 == 0 else "iterdump1.bin"
        b = self.b
        th = getattr(b, "thermal", False)
        fileio.write_iteration_dump(os.path.join(self.dump_dir, name), niter, self.photon_loss_all,
                                    b.fetch("phih_grid"), b.fetch("xh_av"), b.fetch("xh_intermed"), mesh=b.mesh,
                                    phiheat_grid=b.fetch("phiheat_grid") if th else None,
                                    temperature_grid=b.fetch("temperature_grid") if th else None)

    # evolve.F90:328-426
    def start_from_dump(self, restart):
        from . import fileio
        name = {1: "iterdump1.bin", 2: "iterdump2.bin", 3: "iterdump.bin"}[restart]
        th = getattr(self.b, "thermal", False)
        niter, loss, phih, xav, xint, *rest = fileio.read_iteration_dump(os.path.join(self.dump_dir, name), self.b.mesh, thermal=th)
        self.b.load(xh_av=xav, xh_intermed=xint, phih_grid=phih)      # every rank reads the same file (:393-416)
        if th:
            self.b.load(phiheat_grid=rest[0], temperature_grid=rest[1])
        self.photon_loss_all = loss
        return niter

    # evolve.F90:83
    def evolve3D(self, time, dt, restart=0):
        """evolve3D(time,dt,restart) (evolve.F90:83-281): a thin host of the ONE outer loop there is -- the library's
        (c2r_evolve3d_dev / c2r_evolve3d_restart_dev, csrc/evolve_loop.hip: the convergence tests, the all-reduce through
        this process's callback, the global pass, the photon statistics).  What stays here is what the reference does
        outside that arithmetic: restart=1/2/3 reads the iteration dump first (start_from_dump), and rank 0 writes
        iterdump1/2.bin alternately when dump_interval_s of wall clock have passed, from the loop's iteration hook.
        Returns the step's report as a dict (per-iteration history in `log`).  The piecewise entries above (iteration,
        pass_all_sources, global_pass, ...) remain for hosts that drive the steps themselves (bench.py, tests)."""
        b = self.b
        if not hasattr(b, "evolve3d_native"):
            raise TypeError("Evolve.evolve3D is a host of the library's loop: this backend has none (drive it piecewise)")
        self.visited = 0
        t_last = [_time.perf_counter()]

        def hook(niter, loss):                                           # evolve.F90:253-266
            if self.rank == 0 and self.dump_interval_s is not None and _time.perf_counter() - t_last[0] > self.dump_interval_s:
                self.photon_loss_all = loss
                self.write_iteration_dump(niter)
                t_last[0] = _time.perf_counter()
        b.set_iteration_hook(hook if self.dump_interval_s is not None else None)
        try:
            if restart == 0:
                rep = b.evolve3d_native(dt)
            else:
                niter0 = self.start_from_dump(restart)                   # :156
                rep = b.evolve3d_native(dt, restart_niter=niter0, restart_photon_loss=self.photon_loss_all)
        finally:
            b.set_iteration_hook(None)
        self.sum_nbox = self.sum_nbox_all = int(rep.sum_nbox_all)
        self.photon_loss = self.photon_loss_all = float(rep.photon_loss_all)
        self.visited = int(rep.visited)
        first = niter0 - 1 if restart != 0 and niter0 >= 1 else 0        # (a restart logs the repeated global pass in the dump's slot)
        self.log = [dict(conv_flag=int(rep.it_conv_flag[k]), sum_nbox=int(rep.it_sum_nbox[k]), rel_change_xh1=float(rep.it_rel_change_xh1[k]),
                         rel_change_xh0=float(rep.it_rel_change_xh0[k]), sum_xh1=float(rep.it_sum_xh1[k]), photcons=float(rep.it_photcons[k]))
                    for k in range(first, min(int(rep.niter), _capi.MAX_ITER_LOG))]
        phot = dict(totrec=rep.totrec, totcollisions=rep.totcollisions, dh0=rep.dh0, total_ion=rep.total_ion, totalsrc=rep.totalsrc,
                    photcons=rep.photcons, h1_before=rep.h1_before, h1_after=rep.h1_after)
        return dict(photon_statistics=phot, niter=int(rep.niter), converged=bool(rep.converged), conv_flag=int(rep.conv_flag),
                    conv_criterion=int(rep.conv_criterion), sum_nbox_all=self.sum_nbox_all, photon_loss_all=self.photon_loss_all,
                    visited=self.visited, seconds_sweep=float(rep.seconds_sweep), seconds_chem=float(rep.seconds_chem), log=self.log)
