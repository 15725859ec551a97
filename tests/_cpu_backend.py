"""TEST DOUBLE: a CPU backend with the interface Evolve expects, backed by the oracle.
Lives under tests/ on purpose -- the product (c2-ray3dm_amd/) never imports the oracle."""
import numpy as np


class _NpTensor:
    """Minimal tensor facade so the same all_reduce call works on numpy via torch CPU tensors."""


class OracleBackend:
    def __init__(self, oracle, ndens, xh, srcpos, normflux):
        self.o = oracle
        self.mesh = oracle.n
        self.ndens = ndens
        self.xh = xh.copy()
        self.xh_av = xh.copy()
        self.xh_intermed = xh.copy()
        self.phih_grid = np.zeros(oracle.ncell)
        self.srcpos = np.asarray(srcpos, dtype=np.int32)
        self.normflux = np.asarray(normflux, dtype=np.float64)
        self.nsrc = len(self.normflux)
        self.rank, self.npr = 0, 1

    def load(self, ndens=None, xh=None, xh_av=None, xh_intermed=None, phih_grid=None):
        for name, a in (("ndens", ndens), ("xh", xh), ("xh_av", xh_av), ("xh_intermed", xh_intermed),
                        ("phih_grid", phih_grid)):
            if a is not None:
                getattr(self, name)[:] = np.asarray(a).ravel(order="F") if np.asarray(a).ndim == 3 else a

    def fetch(self, name):
        return getattr(self, name).copy()

    def set_rank(self, rank, npr, allreduce=None):
        self.rank, self.npr = rank, npr

    def begin_step(self):
        self.xh_av[:] = self.xh
        self.xh_intermed[:] = self.xh

    def accept(self):
        self.xh[:] = self.xh_intermed
        if self.thermal:                                    # set_final_temperature_point, evolve.F90:220
            self.o.temper_grid[:, 0] = self.o.temper_grid[:, 2]

    def sum_xh_intermed(self):
        return self.o.sum(self.xh_intermed)

    def zero_rates(self):
        self.phih_grid[:] = 0.0
        if self.thermal:
            self.o.phiheat[:] = 0.0                         # evolve.F90:435

    @property
    def thermal(self):
        """Non-isothermal oracle (Oracle.enable_thermal): phiheat and temper_grid live in the oracle object."""
        return getattr(self.o, "heat_thick", None) is not None

    def heat_tensor(self):
        import torch
        return torch.from_numpy(self.o.phiheat) if self.thermal else None

    def set_source_share(self, indices=None):
        self.share = None if indices is None else np.asarray(indices, dtype=np.int64)

    def local_sources(self):
        sh = getattr(self, "share", None)
        return np.asarray(sh if sh is not None else list(range(self.rank, self.nsrc, self.npr)), dtype=np.int64)

    def last_nbox(self):
        return np.asarray(self._last_nbox, dtype=np.int32)

    def pass_sources(self):
        loss, nb, vis, self._last_nbox = 0.0, 0, 0, []
        for i in self.local_sources():
            n1, l1, v1, _ = self.o.do_source(self.ndens, self.xh_av, self.phih_grid, self.srcpos[i], self.normflux[i])
            loss = loss + l1; nb += n1; vis += v1; self._last_nbox.append(n1)
        return loss, nb, vis

    def global_pass(self, dt):
        conv = self.o.global_pass(dt, self.ndens, self.xh, self.xh_av, self.xh_intermed, self.phih_grid)
        return conv, self.o.sum(self.xh_intermed)

    # tensors handed to comm.all_reduce: torch CPU views sharing memory with the numpy arrays
    def rates_tensor(self):
        import torch
        return torch.from_numpy(self.phih_grid)

    def scalars_tensor(self, values):
        import torch
        return torch.tensor(values, dtype=torch.float64)


# ---- the piecewise outer loop of a time step, for THIS double (test infrastructure) -----------------------------------------
# The product has one outer loop: the library's (c2r_evolve3d_dev; Evolve.evolve3D is its thin host).  A backend without it --
# this CPU double -- is driven step by step through Evolve's piecewise entries (set_rates_to_zero, pass_all_sources / iteration,
# global_pass, the dump reader and writer) by the loop below, which restates evolve.F90:83-281 the way the oracle restates the
# kernels: the CPU tests of the host logic (source shares, the all-reduce over gloo, dumps and restarts) run through it.
CONVERGENCE_FRACTION = 9.99999974737875164e-05     # c2ray_parameters.f90:25 (f32 literal, widened)
MAX_OUTER_ITER = 100                               # evolve.F90:228


def evolve3d_piecewise(ev, time_, dt, restart=0):
    import time
    if ev.slab:
        raise ValueError("slab chemistry runs in the native loop: use backend.evolve3d_native(dt)")
    b = ev.b
    n = b.mesh
    ncell = n[0] * n[1] * n[2]
    niter = 0
    conv_flag = ncell
    prev1 = float(np.float32(2.0) * np.float32(n[0]) * np.float32(n[1]) * np.float32(n[2]))
    prev0 = prev1
    if restart == 0:
        b.begin_step()
    else:
        niter = ev.start_from_dump(restart)                        # :156
        prev1 = prev0 = 0.0                                          # saved module variables, zero in a new run
    conv_criterion = min(int(CONVERGENCE_FRACTION * n[0] * n[1] * n[2]), (b.nsrc - 1) // 3)
    ev.log = []
    ev.visited = 0
    t_sweep = t_chem = 0.0
    stats = hasattr(b, "photon_sums")
    before = b.photon_sums("xh", "xh") if stats else None          # evolve.F90:136 state_before
    if restart == 0:
        sum1 = b.sum_xh_intermed()
    else:
        conv_flag, sum1 = ev.global_pass(dt)                       # :157
        ev.log.append(dict(conv_flag=conv_flag, sum_nbox=0, photon_loss=ev.photon_loss_all))
    converged = False
    t_last_dump = time.perf_counter()
    while True:
        sum0 = float(np.float32(ncell)) - sum1
        rel1 = abs(sum1 - prev1) / sum1 if sum1 > 0.0 else 1.0
        rel0 = abs(sum0 - prev0) / sum0 if sum0 > 0.0 else 1.0
        if ev.log:
            ev.log[-1].update(rel_change_xh1=rel1, rel_change_xh0=rel0, sum_xh1=sum1)
        if conv_flag < conv_criterion or (rel1 < CONVERGENCE_FRACTION and rel0 < CONVERGENCE_FRACTION):
            b.accept()
            converged = True
            break
        if niter > MAX_OUTER_ITER:
            break
        prev1, prev0 = sum1, sum0
        niter += 1
        t0 = time.perf_counter()
        # evolve.F90:253-266: rank 0 writes iterdump1/2.bin alternately when the interval has passed -- between the
        # pass and the global pass, so an iteration in which a dump is due runs as its three steps
        dump_due = ev.rank == 0 and ev.dump_interval_s is not None and \
            time.perf_counter() - t_last_dump > ev.dump_interval_s
        if not dump_due and ev.npr == 1:
            conv_flag, sum1 = ev.iteration(niter, dt)
            t1 = t2 = time.perf_counter()
        else:
            ev.set_rates_to_zero()
            ev.pass_all_sources(niter, dt)
            t1 = time.perf_counter()
            if dump_due:
                ev.write_iteration_dump(niter)
                t_last_dump = time.perf_counter()
            conv_flag, sum1 = ev.global_pass(dt)
            t2 = time.perf_counter()
        t_sweep += t1 - t0
        t_chem += t2 - t1
        ev.log.append(dict(conv_flag=conv_flag, sum_nbox=ev.sum_nbox_all,
                             photon_loss=ev.photon_loss_all))
    phot = {}
    if stats:                                                        # evolve.F90:277-279
        after = b.photon_sums("xh", "xh_av")
        vol = b.vol
        totrec, totcol = after[2] * vol * dt, after[3] * vol * dt
        dh0 = before[0] * vol - after[0] * vol
        totalsrc = b.normflux_sum * b.params.S_star * dt
        phot = dict(totrec=totrec, totcollisions=totcol, dh0=dh0, total_ion=totrec + dh0, totalsrc=totalsrc,
                    photcons=(totrec + dh0 - totcol) / totalsrc if totalsrc > 0 else 0.0,
                    h1_before=before[1] * vol, h1_after=after[1] * vol)
    return dict(photon_statistics=phot,
                niter=niter, converged=converged, conv_flag=conv_flag, conv_criterion=conv_criterion,
                sum_nbox_all=ev.sum_nbox_all, photon_loss_all=ev.photon_loss_all,
                visited=ev.visited, seconds_sweep=t_sweep, seconds_chem=t_chem, log=ev.log)
