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
