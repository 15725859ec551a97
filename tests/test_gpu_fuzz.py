"""A 25-case cut of the randomized GPU-vs-oracle run (tests/_fuzz.py; the long run by hand is tests/fuzz_gpu.py),
once per sweep mode: non-cubic meshes of every remainder class, sources inside and outside the mesh, neutral and
highly ionized gas.  Integers and zero patterns exact (asserted in run_case), column densities, photon loss and
rates within the mode's stated tolerances (tests/_util.TOL).  Then 20 random WHOLE steps (tests/_fuzz_steps.py)."""
import pytest
from tests._util import tol, gamma_ok
from tests._fuzz import run_case

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.mark.parametrize("seed", range(1000, 1025))
def test_random_pass_vs_oracle(pkg, tables, sweep_mode, seed):
    r = run_case(seed, pkg, tables, sweep_mode == "fast")
    assert r["loss"] <= tol("loss"), (seed, r["mesh"], r["loss"])
    assert r["cd"] <= tol("cd"), (seed, r["mesh"], r["cd"])
    if sweep_mode == "exact":
        assert r["cd"] == 0.0          # column densities bit for bit
    assert gamma_ok(r["dgamma"], r["gamma_ref"], r["w"], sweep_mode == "fast"), (seed, r["mesh"], r["gamma_rel"], r["gamma_w"])


@pytest.mark.parametrize("seed", range(3000, 3020))
def test_random_whole_step_vs_oracle(pkg, tables, sweep_mode, seed):
    """Whole evolve3D steps to convergence on random small meshes (tests/_fuzz_steps.py: clumping grids, the three LLS
    types, cold / structured / highly ionized starts, one in three non-isothermal): every integer of the step, the
    ionized fractions and the temperatures."""
    from tests._fuzz_steps import run_step_case
    r = run_step_case(seed, pkg, tables, sweep_mode == "fast")
    assert r["niter"][0] == r["niter"][1] and r["converged"][0] == r["converged"][1], (seed, r)
    assert r["conv"][0] == r["conv"][1], (seed, r["mesh"])
    assert r["nbox"][0] == r["nbox"][1], (seed, r["mesh"])
    assert r["dx"] < tol("x"), (seed, r["mesh"], r["dx"])
    # a few units in the last place of the f32 temperature_grid: one per global pass at most, and a step that does not
    # converge repeats the pass 101 times (worst of 51 non-isothermal cases in 150: 2.1e-7, such a step)
    assert r["dtemp"] <= 5e-7, (seed, r["mesh"], r["dtemp"])


@pytest.mark.parametrize("seed", range(6))
def test_replayed_chains_fuzz(pkg, tables, sweep_mode, seed):
    """64 - 400 sources on random meshes, six outer iterations with the field changed in between: chains replayed (captured launch
    sequences, device-gated tail) against chains driven launch by launch -- photon loss (bits), sub-box sums, visited pairs and
    non-converged counts equal in every iteration (tests/_fuzz_chains.py; 200 cases per mode by hand: profiles/r06_chain_graph)."""
    from tests._fuzz_chains import run_chain_case
    r = run_chain_case(seed, pkg, tables, sweep_mode == "fast")
    assert r["counts"] is not None
