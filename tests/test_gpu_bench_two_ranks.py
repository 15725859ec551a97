"""bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per process), with
both ranks on the box's single GPU and gloo for the collective (C2R_BENCH_TEST_ONE_GPU=1; RCCL refuses two ranks on
one device): the N = 2 line must describe the same physics as the N = 1 line -- same sub-box counts, checksums of
Gamma and of the ionized fractions equal to the rounding of the summation order -- and count the same work."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--mesh", "128", "--sources", "64", "--no-cpu-baseline"]


def line(out):
    return json.loads([l for l in out.strip().split("\n") if l.startswith("{")][-1])


@pytest.mark.parametrize("balance", [False, True])
def test_bench_two_ranks_equal_one_rank(balance):
    extra = ["--balance"] if balance else []
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, C2R_BENCH_TEST_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2"] + ARGS + extra, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = line(one.stdout), line(two.stdout)
    assert (a["n_gpus"], b["n_gpus"]) == (1, 2) and b["scaling"] == "strong"
    assert a["config"]["mean_subboxes_per_source"] == b["config"]["mean_subboxes_per_source"]
    assert a["config"]["visited_cell_sources_per_step"] == b["config"]["visited_cell_sources_per_step"]
    assert a["check"]["sum_nbox_last_step"] == b["check"]["sum_nbox_last_step"]
    for k in ("phih_grid_sum", "xh_intermed_sum", "xh_av_sum"):
        assert abs(a["check"][k] / b["check"][k] - 1) < 1e-11, k
    assert b["config"]["sources_per_gpu"] == 32


def test_bench_plain_form_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE -- the shape of the driver's N = 1 command -- must
    work: the parent starts the two ranks itself (before it touches the GPU) and relays rank 0's single line."""
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["C2R_BENCH_TEST_ONE_GPU"] = "1"
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    json_lines = [l for l in two.stdout.strip().split("\n") if l.startswith("{")]
    assert len(json_lines) == 1
    a, b = line(one.stdout), json.loads(json_lines[0])
    assert (a["n_gpus"], b["n_gpus"]) == (1, 2) and b["config"]["ranks"] == 2 and b["config"]["collective"] == "gloo (C2R_BENCH_TEST_ONE_GPU)"
    assert a["check"]["sum_nbox_last_step"] == b["check"]["sum_nbox_last_step"]
    for k in ("phih_grid_sum", "xh_intermed_sum", "xh_av_sum"):
        assert abs(a["check"][k] / b["check"][k] - 1) < 1e-11, k
