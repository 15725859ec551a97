"""Shared helpers for the tests: fixture loading (tests/golden/) and the oracle factory."""
import json
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def F(a):
    """3-D array (i,j,k) -> flat Fortran-order f-contiguous copy (how every C entry point sees it)."""
    return np.asfortranarray(a).ravel(order="F").copy()


def load_tables():
    t = np.load(os.path.join(GOLDEN, "tables.npz"))
    return t["thick"].copy(), t["thin"].copy()


def load_case(name):
    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
    arrays = np.load(os.path.join(GOLDEN, name + ".npz"))
    return meta, arrays


def oracle_for(meta, tables, n=None, lls_grid=None, clump_grid=None):
    """Oracle configured with the per-step scalars recorded from the reference."""
    from oracle.oracle import Oracle
    thick, thin = tables
    return Oracle(n or meta["mesh"], (meta["dr1"], meta["dr2"], meta["dr3"]), meta["vol"],
                  meta["coldensh_LLS"], thick, thin, clumping=meta["clumping"], S_star=meta["S_star"],
                  lls_type=meta.get("type_of_LLS", 1), R_max_LLS=meta.get("R_max_LLS", 0.0),
                  lls_grid=None if lls_grid is None else F(lls_grid),
                  clump_grid=None if clump_grid is None else F(clump_grid))


def expand(a, n):
    """Fixtures store uniform fields as one value."""
    a = np.asarray(a)
    if a.size == 1:
        return np.full((n, n, n), a.flat[0], dtype=a.dtype)
    return a


def relerr(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0
