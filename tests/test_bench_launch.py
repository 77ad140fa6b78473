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


def test_force_dist_goes_through_the_launcher_at_world_size_one():
    """--force-dist: also a 1-GPU run takes the launcher path (parent starts the rank, the rank creates a process group)."""
    r = _run(["--gpus", "1", "--force-dist", "--dry-run", "--steps", "2", "--warmup", "0"])
    assert r["n_gpus"] == 1 and r["outer_iteration"]["world_seen"] == 1 and r["outer_iteration"]["mean_ok"]


def test_gpu_count_without_hip(tmp_path):
    """The launcher parent counts GPUs from the KFD topology (no torch.cuda / HIP call): nodes with SIMDs whose render node is accessible,
    narrowed by the *_VISIBLE_DEVICES variables."""
    from maxstyle_amd.distributed import visible_gpu_count
    nodes = tmp_path / "nodes"
    dri = tmp_path / "dri"
    dri.mkdir()
    for i, (simd, minor) in enumerate([(0, -1), (0, -1), (1024, 128), (1024, 129), (1024, 130)]):     # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ndrm_render_minor {minor}\n")
        if minor >= 0 and minor != 130:                              # the third GPU's render node is not in this container
            (dri / f"renderD{minor}").write_text("")
    n = lambda env: visible_gpu_count(str(nodes), env, str(dri))
    assert n({}) == 2
    assert n({"HIP_VISIBLE_DEVICES": "0"}) == 1
    assert n({"ROCR_VISIBLE_DEVICES": "0,1", "HIP_VISIBLE_DEVICES": "1"}) == 1
    assert n({"CUDA_VISIBLE_DEVICES": "0,1,2,3"}) == 2
    assert n({"HIP_VISIBLE_DEVICES": "0,-1,1"}) == 1
    assert n({"HIP_VISIBLE_DEVICES": ""}) == 0
    assert visible_gpu_count(str(tmp_path / "absent"), {}) == -1


def test_launcher_parent_makes_no_gpu_call():
    """The parent branch of bench.py must not reach torch.cuda at all (a GPU-initialised process is never forked): no `.cuda` attribute access
    anywhere in launch_children's code."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "launch_children")
    attrs = {n.attr for n in ast.walk(fn) if isinstance(n, ast.Attribute)}
    assert "cuda" not in attrs and "device_count" not in attrs and "is_available" not in attrs, attrs
    assert "visible_gpu_count" in {n.id for n in ast.walk(fn) if isinstance(n, ast.Name)}


import pytest


@pytest.mark.gpu
def test_force_dist_world_size_one_runs_rccl():
    """The exact `--gpus N` launcher path at N = 1 WITHOUT --oversubscribe: parent counts GPUs through sysfs and starts the rank, the rank creates a
    world-size-1 `nccl` (RCCL) process group bound to its device, barriers bracket the timed graph replays, max-over-ranks and the flat outer-gradient
    all-reduce (6.1 MB and a 98 MB FCN_64-sized buffer) run through RCCL, and the outer-iteration leg broadcasts + all-reduces on the live group."""
    r = _run(["--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--steady-seconds", "0", "--batch", "4", "--size", "64", "--no-cpu-baseline"])
    assert r["n_gpus"] == 1 and r["config"]["world_seen"] == 1 and r["config"]["hip_graph"] is True
    rc = r["rccl"]
    assert rc["backend"] == "nccl" and rc["world_seen"] == 1 and rc["max_over_ranks_ok"]
    assert rc["fcn16_6MB"]["mean_ok"] and rc["fcn64_98MB"]["mean_ok"] and rc["fcn64_98MB"]["bytes"] == 98000000
    assert rc["fcn16_6MB"]["ms"] > 0 and rc["fcn16_6MB"]["ok_with_step_graph_replay_between"]
    oi = r["outer_iteration"]
    assert oi["world_seen"] == 1 and oi["weights_max_abs_diff_across_ranks"] == 0.0 and oi["allreduce_bytes"] >= 1536325 * 4


@pytest.mark.gpu
def test_plain_run_carries_the_rccl_selftest():
    r = _run(["--steps", "3", "--warmup", "1", "--steady-seconds", "0", "--batch", "4", "--size", "64", "--no-cpu-baseline", "--no-outer"])
    assert "error" not in r["rccl"], r["rccl"]
    assert r["rccl"]["backend"] == "nccl" and r["rccl"]["fcn16_6MB"]["mean_ok"]


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
