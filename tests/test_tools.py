"""CPU checks of the measurement tools' host logic: the per-launch step budget (tools/step_budget.py) - conv cost models, alignment of a launch ledger with a
rocprofv3 kernel trace (rotation to the traced step's order, auxiliary kernels merged into the call that launched them), bounds and executed fractions."""
import csv
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sb():
    spec = importlib.util.spec_from_file_location("ms_step_budget_t", os.path.join(ROOT, "tools", "step_budget.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_conv_cost_models_and_executed_fractions():
    sb = _sb()
    # ms_conv2d(in, in2, out, w, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm, pa, pb, pc, pn, pcs, slope, epi, stats, stream)
    a = (1, 0, 2, 3, 0, 16, 64, 320, 320, 64, 3, 1, 0x900, 1, 0, 0, 0, 0, 4, 0.2, 0, 5, 0)
    cv = sb.conv_cost("ms_conv2d", a)
    assert cv["flop"] == 2.0 * 16 * 320 * 320 * 64 * 64 * 9 and cv["epi"] == 0 and cv["pm"] == 1
    assert sb.executed_fraction("conv_wide_kernel<2, 1, 1, true, ms_f32w>", cv) == 16.0 / 36.0
    assert sb.executed_fraction("conv_wide_kernel<2, 1, 1, true, ms_f32wb>", cv) == 16.0 / 36.0
    assert sb.executed_fraction("conv_wide_kernel<1, 1, 1, true, float>", cv) == 1.0
    assert sb.executed_fraction("conv_mfma_kernel<3, 1, 0, 1, true, false, false, float>", cv) == 1.0
    s2 = sb.conv_cost("ms_conv2d", (1, 0, 2, 3, 0, 16, 16, 256, 256, 16, 3, 2, 0, 0, 0, 0, 0, 0, 4, 1.0, 0, 0, 0))
    assert s2["flop"] == 2.0 * 16 * 128 * 128 * 16 * 16 * 9
    up = sb.conv_cost("ms_conv_subpix", (1, 2, 3, 0, 16, 32, 64, 64, 16, 0, 0, 0, 0, 0, 0.2, 0, 0))
    assert up["flop"] == 2.0 * 16 * 4 * 64 * 64 * 16 * 32 * 9 and sb.executed_fraction("conv_subpix_kernel<0, float>", up) == 4.0 / 9.0
    assert sb.executed_fraction("conv_subpix_kernel<1, float>", up) == 0.25
    sc = sb.conv_cost("ms_conv3x3_small_cout", (1, 2, 3, 4, 16, 16, 256, 256, 1, 2, 0, 0, 0, 4, 0))
    assert sb.executed_fraction("conv3x3_small_cout_kernel<1, true, float>", sc) == 0.0      # vector-ALU kernel: priced by its bytes
    assert sb.conv_cost("ms_bn_finalize", (1, 2, 3)) is None


def test_merge_rotates_the_ledger_and_folds_auxiliary_kernels(tmp_path):
    sb = _sb()
    conv = dict(N=16, Cin=16, Hs=256, Ws=256, Cout=16, ks=3, stride=1, fetch=0x100, pm=0, epi=0, flop=2.0 * 16 * 256 * 256 * 16 * 16 * 9)
    ledger = [dict(fn="ms_conv2d", key="a", conv=conv, tensors=[], bytes=134217728, flop=conv["flop"]),
              dict(fn="ms_head_ce", key="", conv=None, tensors=[], bytes=67108864, flop=0.0),          # launches head_ce_kernel + ce_finalize_kernel
              dict(fn="ms_step_tail", key="", conv=None, tensors=[], bytes=0, flop=0.0),
              dict(fn="ms_style_fwd", key="3", conv=None, tensors=[], bytes=33554432, flop=0.0)]      # the re-decode behind the tail: FIRST in a traced step
    lp = tmp_path / "ledger.json"
    json.dump(dict(config="c2", batch=16, size=256, launches=4, ledger=ledger), open(lp, "w"))
    tdir = tmp_path / "trace" / "x"
    os.makedirs(tdir)
    names = ["void ms::step_tail_kernel(int)"]
    step = ["void ms::style_fused_kernel<512, 0, 0, float>(ms::FusedArgs)", "void ms::conv_wide_kernel<1, 0, 1, true, ms::ms_f32w>(ms::ConvArgs)",
            "void ms::head_ce_kernel<float>(int)", "ms::ce_finalize_kernel(int)", "void ms::step_tail_kernel(int)"]
    durs = {"style_fused": 15000, "conv_wide": 42000, "head_ce_kernel": 30000, "ce_finalize": 5000, "step_tail": 12000}
    rows, t = [], 1000
    for nm in names + step * 3:
        d = next(v for k, v in durs.items() if k in nm)
        rows.append({"Kernel_Name": nm, "Start_Timestamp": t, "End_Timestamp": t + d})
        t += d + 500
    with open(tdir / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader(); w.writerows(rows)
    stem = str(tmp_path / "out")
    sb.merge(str(lp), str(tmp_path / "trace"), stem)
    r = json.load(open(stem + ".json"))
    assert r["summary"]["alignment_mismatches"] == 0 and r["summary"]["launches"] == 4 and r["summary"]["steps_averaged"] == 3
    by_fn = {o["fn"]: o for o in r["launches"]}
    assert "style_fused" in by_fn["ms_style_fwd"]["kernel"] and r["launches"][0]["fn"] == "ms_style_fwd"            # rotated: the traced step starts behind the tail
    assert abs(by_fn["ms_head_ce"]["actual_us"] - 35.0) < 1e-9 and "ce_finalize" in by_fn["ms_head_ce"]["kernel"]       # both kernels of the call
    c = by_fn["ms_conv2d"]
    assert c["bound"] == "hbm" and abs(c["bound_us"] - 134217728 / 8.0e12 * 1e6) < 1e-9                                # Winograd: 16/36 of the flop -> HBM-bound
    assert abs(c["flop_executed"] - conv["flop"] * 16 / 36) < 1.0
    assert abs(r["summary"]["sum_actual_us"] - (15 + 42 + 35 + 12)) < 1e-6
    assert os.path.exists(stem + ".txt")
