"""The one JSON line `python bench.py` prints on a GPU box: the fields the driver and the judge read (task contract), checked on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--steady-seconds", "0", "--cpu-steps", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1000.0 * d["steps"] / (d["ms_per_step"] * d["steps"])) < 1e-6 * d["value"]       # value = whole-job steps / s of the timed region
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] <= 1.0
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes"]
    # round 4: no fraction of any roofline block exceeds 1 (Winograd launches are priced on EXECUTED flop), the headline fraction uses the launch's
    # IN-STEP duration (measured live: two cut-off captures of the step), the isolated replay is a side field, and the kernel string names the instantiation that ran
    def blocks(o):
        if isinstance(o, dict):
            if "bound" in o and "frac" in o:
                yield o
            for v in o.values():
                yield from blocks(v)
    n_blocks = 0
    for b in blocks(d):
        n_blocks += 1
        for k, v in b.items():
            if "frac" in k and isinstance(v, float):
                assert 0.0 < v <= 1.0, (b.get("kernel"), k, v)
    assert n_blocks >= 8
    assert r["us_per_launch_source"].startswith("in-step") and r["us_per_launch"] >= 0.9 * r["us_per_launch_isolated"] and r["frac_isolated"] >= r["frac"] * 0.9
    assert "conv_wide_kernel<1,2,1,true,ms_f32w>" in r["kernel"] and "FX" in r["kernel"] and r["form"].startswith("winograd") and r["channel_blocks_per_tile"] == 1
    sr = d["step_roofline"]
    assert sr["launches"] >= 60 and 0.1 < sr["frac"] < 1.0 and abs(sr["frac"] - sr["sum_bound_us"] / sr["step_us"]) < 1e-9
    assert abs(sr["hbm_bound_us"] + sr["mfma_bound_us"] - sr["sum_bound_us"]) < 1e-6 * sr["sum_bound_us"]
    assert os.path.exists(os.path.join(ROOT, d["step_roofline_per_launch"].split(" ")[0]))
    # the committed traffic figures are not older than the kernel sources they price: profiles/r0N_traffic.json names the git blob of each source
    if r.get("traffic") is not None:
        assert r.get("traffic_fresh") is True, r.get("traffic_source")
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"] and isinstance(c["sample"], str)
    # round 3: the driver-visible line also carries parity at the benchmarked size against the reference's own run (both conv forms), the secondary
    # configurations and the RCCL self-test
    for form in ("winograd", "direct"):
        assert d["drift_full_size"][form]["ratio_to_reference_noise_max"] <= 2.0 and d["drift_full_size"][form]["ratio_to_reference_noise_rms"] <= 2.0
        assert d["dice_parity"][form]["dice_max_abs_diff_vs_reference"] <= 1e-3 and d["dice_parity"][form]["labels_equal_to_reference"] >= 0.9999
    sec = d["secondary"]
    assert sec["winograd_off"]["steps_s"] > 0 and sec["c4"]["steps_s"] > 0 and sec["c5_bf16"]["steps_s"] > 0
    assert "executed_mfma_frac" in sec["c4"]["roofline"] and sec["c4"]["roofline"]["form"].startswith("winograd") and sec["c4"]["roofline"]["channel_blocks_per_tile"] == 2
    assert sec["c4"]["step_roofline"]["launches"] >= 60 and 0.1 < sec["c4"]["step_roofline"]["frac"] < 1.0
    assert "error" not in d["rccl"] and d["rccl"]["backend"] == "nccl" and d["rccl"]["fcn16_6MB"]["mean_ok"]
    assert d["roofline"].get("traffic_source") is None or d["roofline"]["traffic_source"].startswith("profiles/")
    # round 6: the outer iteration carries the roofline of its training passes (tools/train_budget.py; a committed profile, named as such)
    osr = d["outer_iteration"]["step_roofline"]
    assert osr["source"].startswith("profiles/") and 0.05 < osr["frac"] < 1.0 and osr["training_passes_sum_bound_us"] < osr["training_passes_wall_us"]
