"""Shared helpers for the tests: fixture loading (tests/golden/) and the oracle factory."""
import json
import os
import numpy as np
import pytest

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


def load_thermal_tables():
    """Heating tables and the (synthetic) cooling table the non-isothermal fixtures were generated with."""
    t = np.load(os.path.join(GOLDEN, "tables_thermal.npz"))
    return {k: t[k].copy() for k in ("heat_thick", "heat_thin", "cool_logT", "cool_logL")}


def thermal_oracle_for(meta, tables, temper_grid, n=None):
    """Oracle of a non-isothermal step: temper_grid (ncell x 3 f32: current, average, intermed) is updated in place."""
    o = oracle_for(meta, tables, n)
    tt = load_thermal_tables()
    o.enable_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"], meta["zred"], temper_grid)
    return o


def expand(a, n):
    """Fixtures store uniform fields as one value."""
    a = np.asarray(a)
    if a.size == 1:
        return np.full((n, n, n), a.flat[0], dtype=a.dtype)
    return a


def relerr(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


# ---- stated tolerances of the GPU parity tests, per sweep mode (include/c2ray_hip.h: c2r_params.sweep_mode) ----
# Integer results (nbox, visited cells, outer-iteration count, non-converged-cell sequence) are exact in both.
#   cd     column densities, relative           exact: differs from the Fortran only where the reference reads a
#                                               not-yet-computed neighbour with weight ~1e-16 (bit-identical otherwise)
#   loss   photon loss through the sub-box surface, relative
#   x      ionized fractions after a global pass / a whole step, absolute (north_star asks for 1e-5)
#   gamma  photo-ionization rates:  |dGamma| <= rtol * Gamma + wtol * W   with, per cell,
#          W = sum_s (1 + tau_in) photo_in / (vol_ph n_HI)   (oracle_cfg.tolw).
#          Gamma is NormFlux (T(tau_in) - T(tau_out)) / (vol_ph n_HI): the two table values carry the rounding error
#          of log10 and of the table position (an ulp of the position ~2000 is 2e-13 of a table step; d ln T / d position
#          = 0.028 tau deep in a column), so each is good to ~6e-15 (1 + tau) relative and their DIFFERENCE inherits
#          that absolute error however small it is.  rtol covers the rest of the arithmetic.  The reference has the
#          same sensitivity to its libm.
TOL = {
    # measured over 150 random cases per mode (tests/fuzz_gpu.py, profiles/r02_tolerance/): worst |dGamma|/W 3.9e-15
    # (exact) and 4.2e-15 (fast); worst plain relative error 1.8e-8 / 7.6e-8, in cells with W/Gamma ~ 1e7
    "exact": dict(cd=1e-11, loss=1e-10, x=1e-9, gamma_rtol=1e-13, gamma_wtol=2e-14),
    "fast":  dict(cd=1e-11, loss=1e-10, x=1e-9, gamma_rtol=1e-12, gamma_wtol=2e-14),
}


# ... and a PLAIN relative bound that no weight can hide a regression under: wherever a cell's own rate is at least
# GAMMA_PLAIN_FLOOR of the rate passing through it (W), |dGamma| <= GAMMA_PLAIN_RTOL * Gamma, in both modes (the weighted
# criterion allows 2e-8 there; below the floor a cell's rate is the difference of two table values ~1e6 times larger and only
# the weighted criterion is meaningful -- the reference's own libm moves those cells by as much).  evolve_point.F90:262.
GAMMA_PLAIN_RTOL = 2e-7
GAMMA_PLAIN_FLOOR = 1e-6


def gamma_plain_rel(dgamma, gamma_ref, w):
    """max |dGamma| / Gamma over the cells with Gamma >= GAMMA_PLAIN_FLOOR x W (0.0 when there is none)."""
    dgamma, gamma_ref, w = (np.asarray(v, dtype=np.float64) for v in (dgamma, gamma_ref, w))
    sig = (gamma_ref > 0) & (gamma_ref >= GAMMA_PLAIN_FLOOR * w)
    return float(np.max(np.abs(dgamma[sig]) / gamma_ref[sig])) if sig.any() else 0.0


def gamma_ok(dgamma, gamma_ref, w, fast):
    t = TOL["fast" if fast else "exact"]
    return bool(np.all(np.abs(dgamma) <= t["gamma_rtol"] * np.abs(gamma_ref) + t["gamma_wtol"] * w)) and \
        gamma_plain_rel(dgamma, gamma_ref, w) <= GAMMA_PLAIN_RTOL


def sweep_mode():
    """Mode the GPU tests run the sweep in: the `sweep_mode` fixture (conftest.py) sets C2R_SWEEP_MODE, which
    the HOSTS above the C ABI read (HipBackend(fast=None), the Fortran shim, the native harness), so every context a test creates follows it."""
    return "exact" if os.environ.get("C2R_SWEEP_MODE") == "0" else "fast"       # (unset: the library default, C2R_SWEEP_FAST)


def tol(key):
    return TOL[sweep_mode()][key]


# Whole time steps: the rates of the LAST pass are computed from the xh_av the previous iterations left, and the two
# codes' xh_av differ by then (device exp vs glibc exp in doric, a different rounding per iteration: |dx| ~ 1e-14,
# asserted < 1e-9).  n_HI = (1 - x) n turns that into a relative difference |dx| / (1 - x) of every column and
# rate -- ~1e-11 at x = 0.9995 -- before any sweep arithmetic runs.  Comparisons of a whole step's Gamma with the
# Fortran's therefore add this state term; same-state comparisons (a pass against the oracle) do not.
STATE_RTOL = 1e-9


def assert_gamma(got, ref, w, what="", state_rtol=0.0):
    """|got - ref| <= (rtol + state_rtol) |ref| + wtol W cell by cell (see TOL), and the same cells carry a rate."""
    got, ref, w = (np.asarray(v, dtype=np.float64) for v in (got, ref, w))
    assert np.array_equal(got == 0, ref == 0), what
    t = TOL[sweep_mode()]
    excess = np.abs(got - ref) - ((t["gamma_rtol"] + state_rtol) * np.abs(ref) + t["gamma_wtol"] * w)
    if excess.size and excess.max() > 0:
        i = int(np.argmax(excess))
        pytest.fail("%s Gamma out of tolerance at flat index %d: got %.17g ref %.17g W %.3g (rel %.2e, /W %.2e)" %
                    (what, i, got.flat[i], ref.flat[i], w.flat[i], abs(got.flat[i] / ref.flat[i] - 1),
                     abs(got.flat[i] - ref.flat[i]) / w.flat[i]))
    # the plain bound (GAMMA_PLAIN_RTOL wherever Gamma >= GAMMA_PLAIN_FLOOR x W), whatever the weight says
    plain = gamma_plain_rel(got - ref, ref, w)
    assert plain <= GAMMA_PLAIN_RTOL + state_rtol, "%s plain relative Gamma error %.2e in a cell with Gamma >= %.0e W (bound %.0e)" % (
        what, plain, GAMMA_PLAIN_FLOOR, GAMMA_PLAIN_RTOL + state_rtol)


def oracle_pass(o, nd, xh, srcpos, normflux):
    """One pass of the oracle over all sources with the tolerance weight: (loss, sum_nbox, visited, phih, W).
    On big meshes the sources are split into contiguous chunks traced by worker threads (the C oracle is serial: 0.15 us per
    visited cell, 20 s per source at 504^3), each chunk into its own rate and weight arrays, added in chunk order afterwards:
    integers are the serial pass's, rates / weight / loss re-associate at 1e-16 per add -- far inside every tolerance the
    checker applies (and bit-identical to the serial pass for up to two sources)."""
    srcpos = np.ascontiguousarray(srcpos, dtype=np.int32).reshape(-1, 3)
    normflux = np.ascontiguousarray(normflux, dtype=np.float64)
    nthr = min(len(normflux), os.cpu_count() or 1, 8)
    if nthr <= 1 or o.ncell < 96 ** 3 or getattr(o, "heat_thick", None) is not None:
        w = o.enable_tolerance_weight()
        phih = np.zeros(o.ncell)
        loss, nb, vis = o.pass_sources(nd, xh, phih, srcpos, normflux)
        return loss, nb, vis, phih, w.copy()
    import threading
    bounds = np.linspace(0, len(normflux), nthr + 1).astype(int)
    out = [None] * nthr

    def work(t):
        oc = o.clone()
        w = oc.enable_tolerance_weight()
        g = np.zeros(o.ncell)
        lo, hi = bounds[t], bounds[t + 1]
        out[t] = (oc.pass_sources(nd, xh, g, srcpos[lo:hi], normflux[lo:hi]), g, w, oc)
    th = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
    for t in th: t.start()
    for t in th: t.join()
    (loss, nb, vis), phih, w, _ = out[0]
    for (l, n_, v), g, wt, _ in out[1:]:
        loss += l; nb += n_; vis += v
        phih += g; w += wt
    return loss, nb, vis, phih, w


def oracle_step(o, dt, nd, xh0, srcpos, normflux):
    """A whole evolve3D step of the oracle: (report, xh_after, xh_av, phih of the last pass, its W)."""
    w = o.enable_tolerance_weight()
    xh = xh0.copy()
    rep, xh_av, xh_int, phih = o.evolve3d(dt, nd, xh, srcpos, normflux)
    return rep, xh, xh_av, phih, w.copy()
