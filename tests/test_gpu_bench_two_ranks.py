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


ARGS8 = ["--steps", "2", "--warmup", "1", "--mesh", "64", "--sources", "64", "--no-cpu-baseline", "--no-other-mode"]


@pytest.mark.parametrize("form,balance", [("launcher", False), ("plain", True)])
def test_bench_eight_ranks_equal_one_rank(form, balance):
    """The driver's scaling step runs 1 / 2 / 4 / 8 ranks; only two had ever been tried.  Eight ranks (all on the box's one GPU,
    gloo) through both launch forms: the line says 8 ranks, the checksums equal the one-rank line's to the order of the sums,
    the shares (static stride, or the library's LPT partition with --balance: master_slave.F90:74-96 / :124-330) partition the
    source list."""
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS8, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    extra = ["--balance"] if balance else []
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(C2R_BENCH_TEST_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    if form == "launcher":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "8"] + ARGS8 + extra
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + ARGS8 + extra
    eight = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert eight.returncode == 0, eight.stderr[-2000:]
    a, b = line(one.stdout), line(eight.stdout)
    assert (a["n_gpus"], b["n_gpus"]) == (1, 8) and b["config"]["ranks"] == 8 and b["scaling"] == "strong"
    assert a["check"]["sum_nbox_last_step"] == b["check"]["sum_nbox_last_step"]
    assert a["config"]["visited_cell_sources_per_step"] == b["config"]["visited_cell_sources_per_step"]
    for k in ("phih_grid_sum", "xh_intermed_sum", "xh_av_sum"):
        assert abs(a["check"][k] / b["check"][k] - 1) < 1e-11, k
    assert b["config"]["shares_partition_sources"] is True and sum(b["config"]["source_share_sizes"]) == 64
    assert len(b["config"]["source_share_sizes"]) == 8
    if not balance:
        assert b["config"]["source_share_sizes"] == [8] * 8
    # the diagnostic fields a first real multi-GPU run will be read by: where each rank's time went (min / max over the
    # ranks) and what the exchange moves against one ring over xGMI
    ph = b["config"]["rank_phases"]
    for k in ("sweep", "exchange", "chem"):
        assert 0.0 < ph[k]["min_s_per_step"] <= ph[k]["max_s_per_step"]
    assert sum(ph[k]["max_s_per_step"] for k in ph) >= 0.5 * b["ms_per_step"] * 1e-3
    assert b["config"]["gamma_exchange"]["ring_over_xgmi_s_per_step"] > 0.0 and a["config"]["rank_phases"] is None
    # first contact with a multi-GPU node: the communicator's own rank count, and the device every rank's context resolved to (all
    # eight on the one GPU here, on purpose -- which the line must SAY; without C2R_BENCH_TEST_ONE_GPU that fails the run, below)
    assert b["config"]["communicator_ranks"] == 8 and a["config"]["communicator_ranks"] == 1
    rd = b["config"]["rank_devices"]
    assert [d["rank"] for d in rd] == list(range(8)) and all(d["c2r_device"] == 0 and d["device_id"] for d in rd)
    assert b["config"]["ranks_on_distinct_devices"] is False and a["config"]["ranks_on_distinct_devices"] is True


def test_bench_more_ranks_than_devices_fails_loudly():
    """`bench.py --gpus 2` on a box with ONE GPU and no test override: every rank finds fewer visible devices than ranks before any
    rendezvous and exits non-zero with a message -- no hang in a collective, no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "C2R_BENCH_TEST_ONE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS8, capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the launch is legitimate")
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and not [l for l in r.stdout.split("\n") if l.startswith("{")]


def test_bench_two_ranks_with_the_exchange_overlapped():
    """bench.py --gpus 2 --overlap-exchange (two ranks on the one GPU, gloo): 256 sources, i.e. 128 per rank -- the pass runs as
    two halves of 64 (each as two chains in flight) with the first half's all-reduce issued before the second is swept; same
    checksums as one rank."""
    args = ["--steps", "2", "--warmup", "1", "--mesh", "64", "--sources", "256", "--no-cpu-baseline", "--no-other-mode"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(C2R_BENCH_TEST_ONE_GPU="1")
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--overlap-exchange", "--option", "sparse_exchange=0"] + args, capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = line(one.stdout), line(two.stdout)
    assert b["config"]["exchange_overlapped_with_sweep"] is True
    x = b["config"]["gamma_exchange"]
    assert x["bytes_per_step"] == 2 * x["full_grid_bytes"]          # two halves of a grid per pass: the overlapped path ran
    assert a["check"]["sum_nbox_last_step"] == b["check"]["sum_nbox_last_step"]
    for k in ("phih_grid_sum", "xh_intermed_sum", "xh_av_sum"):
        assert abs(a["check"][k] / b["check"][k] - 1) < 1e-11, k


def test_bench_a_failing_rank_fails_the_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(C2R_BENCH_TEST_ONE_GPU="1", C2R_BENCH_TEST_FAIL_RANK="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + ARGS8, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.split("\n") if l.startswith("{")]


def test_bench_cold_two_ranks_exchange_packed_sub_boxes():
    """Cold 256^3 x 1000 (x = 2e-4: every source stays inside its first sub-boxes) on two ranks: the rates travel as packed
    sub-boxes, at most a tenth of the N^3 x 8 bytes the plain all-reduce of evolve.F90:599 moves per iteration; same checksums
    as one rank."""
    args = ["--steps", "2", "--warmup", "1", "--mesh", "256", "--sources", "1000", "--x-init", "2e-4", "--no-cpu-baseline", "--no-other-mode",
            "--no-small-leg", "--no-dropin-leg", "--no-mix-ceiling"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["C2R_BENCH_TEST_ONE_GPU"] = "1"
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = line(one.stdout), line(two.stdout)
    x = b["config"]["gamma_exchange"]
    assert x["calls"] == 2 and x["packed_calls"] == 2
    assert x["bytes_per_step"] <= 0.10 * x["full_grid_bytes"], x
    assert a["check"]["sum_nbox_last_step"] == b["check"]["sum_nbox_last_step"]
    for k in ("phih_grid_sum", "xh_intermed_sum", "xh_av_sum"):
        assert abs(a["check"][k] / b["check"][k] - 1) < 1e-11, k
