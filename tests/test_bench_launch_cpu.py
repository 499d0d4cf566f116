"""`bench.py --gpus N` must BE N ranks: until round 6 the flag was parsed and never read -- the world came from WORLD_SIZE only --
so the N = 1 command with the number changed ran one rank and printed "n_gpus": 1 with exit code 0.

No GPU here: `--rendezvous-check` stops right behind the process-group rendezvous (gloo), which is the part under test; the same launcher in front of
the real measurement runs under -m gpu (tests/test_multigpu_gpu.py::test_bench_gpus_2_spawns_two_ranks_on_one_gpu)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TS2D_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", **extra)
    return env


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_gpus_2_alone_spawns_two_ranks():
    """No torchrun in front: the command the driver would plausibly issue for N = 2."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-check"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1] and line["distinct_processes"] == 2
    assert line["distributed"]["world"] == 2 and line["distributed"]["backend"] == "gloo"


def test_gpus_3_under_torchrun_with_matching_world():
    """The driver's documented form: torch.distributed.run --nproc-per-node N bench.py --gpus N."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", BENCH, "--gpus", "3", "--rendezvous-check"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 3 and line["ranks"] == [0, 1, 2] and line["distinct_processes"] == 3


def test_world_size_that_contradicts_gpus_is_refused():
    """WORLD_SIZE = 1 with --gpus 2 (and the reverse): exit code 2 and no JSON line -- never a line that reports fewer GPUs than were asked for."""
    for world, gpus in (("1", "2"), ("2", "1")):
        r = subprocess.run([sys.executable, BENCH, "--gpus", gpus, "--rendezvous-check"],
                           env=_env(WORLD_SIZE=world, RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29632"),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 2, (r.returncode, r.stdout, r.stderr[-500:])
        assert "WORLD_SIZE" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_rank_needs_no_rendezvous():
    r = subprocess.run([sys.executable, BENCH, "--rendezvous-check"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks"] == [0] and line["distributed"]["backend"] is None
