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

    def sum_xh_intermed(self):
        return self.o.sum(self.xh_intermed)

    def zero_rates(self):
        self.phih_grid[:] = 0.0

    def pass_sources(self):
        return self.o.pass_sources(self.ndens, self.xh_av, self.phih_grid, self.srcpos, self.normflux,
                                   self.rank, self.npr)

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
