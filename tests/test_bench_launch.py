"""`python bench.py --gpus N` without a launcher must start N ranks itself (the flag was parsed and ignored).
Runs the rank plumbing over gloo on the CPU (`--dry-run`: stand-in step, no GPU call anywhere)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_flag_spawns_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2" and r["config"]["global_batch"] == 32
    assert r["outer_iteration"]["world_seen"] == 2 and r["outer_iteration"]["mean_ok"]
    assert r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["dry_run"] is True


def test_single_rank_default():
    r = _run(["--dry-run", "--steps", "2", "--warmup", "0"])
    assert r["n_gpus"] == 1 and r["config"]["parallelism"] == "dp1"


def test_under_a_launcher_nothing_is_spawned():
    # torchrun-style environment for a 1-rank world: the process is a rank, not a parent, whatever --gpus says
    r = _run(["--gpus", "4", "--dry-run", "--steps", "2", "--warmup", "0"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r["n_gpus"] == 1


import pytest


@pytest.mark.gpu
def test_real_two_rank_run_on_one_gpu():
    """The N > 1 path of bench.py on real kernels: two ranks share the one GPU of the test box (gloo moves the CUDA tensors; on a node the same code
    runs one rank per GPU over RCCL).  Timed loop per rank, max over ranks, and the outer-iteration leg with the flat all-reduce on every rank."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--oversubscribe", "--steps", "3", "--warmup", "1", "--steady-seconds", "0", "--batch", "4", "--size", "64"])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2" and r["config"]["world_seen"] == 2 and r["config"]["global_batch"] == 8
    assert r["value"] > 0 and abs(r["per_gpu_steps_s"] * 2 - r["value"]) < 1e-6 * r["value"]
    oi = r["outer_iteration"]
    assert oi["world_seen"] == 2 and oi["weights_max_abs_diff_across_ranks"] == 0.0 and oi["allreduce_bytes"] >= 1536325 * 4
    assert oi["loss_last"] < oi["loss_first"]
