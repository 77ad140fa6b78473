"""Measured parity numbers of round 3 against the reference's own fp32-vs-fp64 noise (run on the GPU box):
    python tools/parity_report.py > gpurun_out/r03_parity_report.txt
Prints, for every round-3 case of tests/r3_cases.py and the teacher-forced fixtures of tests/test_engine_gpu.py, the GPU's error against the
reference's fp64 run beside the reference's own fp32 error - the numbers the constants in tests/test_round3_gpu.py / test_engine_gpu.py come from."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import r3_cases as R
    import test_engine_gpu as TE
    dev = torch.device("cuda:0")
    golden = os.path.join(ROOT, "tests", "golden")
    print("== full size (BASELINE config 2, trained FCN_16, K=5 free-running) vs the reference's fp64 run")
    from maxstyle_amd.options import engine_defaults
    for wino in (True, False):
        with engine_defaults(winograd=wino):
            r = R.full_size_case(dev)
        print(json.dumps(r))
        print(f"   winograd={r['winograd']}: image max {r['image_max']:.3e} (reference noise {r['noise_image_max']:.3e}, ratio {r['image_max'] / r['noise_image_max']:.2f}), "
              f"rms {r['image_rms']:.3e} ({r['noise_image_rms']:.3e}, {r['image_rms'] / r['noise_image_rms']:.2f}), labels equal {r['labels_equal_f64']:.6f} "
              f"(reference {r['noise_labels_equal']:.6f}), Dice diff {r['dice_abs_diff']:.2e}")
    print("== arguments of the drop-in signature (trained FCN_16, 4x1x64x64) vs the reference's fp64 run")
    for case in R.ARG_CALLS:
        r = R.arg_case(dev, case)
        print(case, json.dumps(r))
    print("== teacher-forced gradients: err(GPU, fp64 twin) / reference fp32 noise of the step")
    orc = __import__("oracle.maxstyle_oracle", fromlist=["x"])
    TE.TF_C, TE.TF_FLOOR = 1e9, 1e9          # measure, do not assert
    for tag, fx, net, B, layers, K in (("c2small", "loop_c2small", (4, 1, 4), 4, [3, 4, 5], 5), ("c4small", "loop_c4small", (1, 3, 2), 4, [3, 4, 5], 3),
                                       ("all_layers", "loop_all_layers", (4, 1, 4), 3, [0, 1, 2, 3, 4, 5], 2)):
        g = np.load(os.path.join(golden, fx + ".npz")); tf = np.load(os.path.join(golden, fx + "_tf64.npz"))
        TE._teacher_forced(dev, g, None, orc.NetSpec(*net), B, 64, layers, K, tf=tf, tag=tag)
    for (tag, s, n), (err, noise) in sorted(TE.TF_TABLE.items()):
        print(f"   {tag:10s} step {s} {n:14s} err {err:.2e}  step noise {noise:.2e}  ratio {err / noise:.2f}")
    worst = max(err / max(noise, 1e-30) for err, noise in TE.TF_TABLE.values())
    print("   worst ratio", worst, " worst abs", max(err for err, _ in TE.TF_TABLE.values()))


def round4():
    """python tools/parity_report.py r4 > gpurun_out/r04_parity_report.txt - config 4 at size against the reference's own run (both conv forms) and config 5's two
    call shapes (fp32 and bf16 storage) against the reference's fp32 run with the bf16-storage oracle's distance beside them: the numbers behind tests/test_round4_gpu.py."""
    import r4_cases as R4
    dev = torch.device("cuda:0")
    print("== BASELINE config 4 at size (trained FCN_64, 16x3x320x320, K=10 free-running) vs the reference's fp64 run; noise_* = the reference's own fp32 run against it")
    from maxstyle_amd.options import engine_defaults
    for wino in (True, False):
        with engine_defaults(winograd=wino):
            r = R4.c4_full_case(dev)
        print(json.dumps(r))
        print(f"   winograd={r['winograd']}: image max {r['image_max']:.3e} (reference noise {r['noise_image_max']:.3e}, ratio {r['image_max'] / r['noise_image_max']:.2f}), "
              f"rms {max(r['image_rms_full'], r['image_rms_strided']):.3e} ({r['noise_image_rms']:.3e}, {max(r['image_rms_full'], r['image_rms_strided']) / r['noise_image_rms']:.2f}), "
              f"plane mean {r['mean_rel']:.2e} ({r['noise_plane_mean']:.2e}), plane rms {r['rms_rel']:.2e} ({r['noise_plane_rms']:.2e}), labels equal {r['labels_equal_f64']:.6f} "
              f"(reference {r['noise_labels_equal']:.6f}), Dice diff {r['dice_abs_diff']:.2e}, Dice {r['dice']} clean {r['dice_clean']}")
        print("   losses rel err per step:", ["%.1e" % e for e in r["losses_rel"]], " reference noise:", ["%.1e" % e for e in r["noise_losses_rel"]])
        print("   params rel err:", {k: "%.1e" % v for k, v in r["params_rel"].items()}, " worst reference noise: %.1e" % max(r["noise_params_rel"].values()))
    print("== BASELINE config 5, one call per shape (p = 0.5: the reference's own draw), vs the reference's fp32 run; oracle_bf16_* = the CPU oracle with bf16 storage emulation")
    for tag in ("acdc", "prostate"):
        for dt in (None, torch.bfloat16):
            r = R4.c5_call_case(dev, tag, dt)
            print(tag, r["storage"], json.dumps({k: v for k, v in r.items() if k not in ("losses", "losses_ref")}))
            print(f"   applied {r['applied']}: image max {r['image_max']:.2e} rms {r['image_rms']:.2e} (bf16 oracle {r['oracle_bf16_image_max']:.2e} / {r['oracle_bf16_image_rms']:.2e}), "
                  f"labels {r['labels_equal']:.5f} (oracle {r['oracle_bf16_labels_equal']:.5f}), Dice diff {r['dice_abs_diff']:.1e}, worst loss err {max(r['losses_rel']):.1e} "
                  f"(oracle {max(r['oracle_bf16_losses_rel']):.1e})")
    import r3_cases as R
    print("== arguments of the drop-in signature + all six layers (trained FCN_16, 4x1x64x64, K = 3) vs the reference's fp64 run; noise = the reference's fp32 run against it")
    for case in list(R.ARG_CALLS) + ["all6"]:
        r = R.arg_case(dev, case)
        pm = max(list(r["params_rel"].values()) + [0.0]); pn = max(list(r["noise_params_rel"].values()) + [0.0])
        print(f"   {case:14s} image {r['image_rel']:.2e} (noise {r['noise_image_rel']:.2e})  params {pm:.2e} ({pn:.2e})  losses {max(r['losses_rel'] or [0]):.1e} "
              f"({max(r['noise_losses_rel'] or [0]):.1e})  labels {r['labels_equal']:.6f}  Dice diff {r['dice_abs_diff']:.1e}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "r4":
    round4()
    sys.exit(0)
if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "free"):
    main()


def free_running():
    """Free-running K=5 on the small fixtures: the GPU's distance from the reference's fp64 run beside the reference's own fp32 distance."""
    import test_engine_gpu as TE
    from parity_util import rel
    orc = __import__("oracle.maxstyle_oracle", fromlist=["x"])
    dev = torch.device("cuda:0")
    golden = os.path.join(ROOT, "tests", "golden")
    print("== free-running loops: err(GPU, reference fp64) vs err(reference fp32, reference fp64)")
    for tag, fx, net, B, size, layers, K in (("c2small", "loop_c2small", (4, 1, 4), 4, 64, [3, 4, 5], 5), ("c4small", "loop_c4small", (1, 3, 2), 4, 64, [3, 4, 5], 3),
                                             ("c1", "loop_c1", (4, 1, 4), 4, 128, [3], 1)):
        g = np.load(os.path.join(golden, fx + ".npz")); g64 = np.load(os.path.join(golden, fx + "_f64.npz"))
        for graph in (False, True):
            eng, W, img, lab, styles = TE.build_engine(dev, orc.NetSpec(*net), B, size, layers)
            z_i = torch.from_numpy(g["z_i"]).to(dev)
            out = eng.run(z_i, lab.to(dev), K, use_graph=graph).clone()
            losses = eng.losses(K).cpu().numpy().astype(np.float64)
            ni, nl = rel(g["image"], g64["image"]), float(np.max(np.abs(g["losses"] - g64["losses"]) / np.abs(g64["losses"])))
            ei, el = rel(out, g64["image"]), float(np.max(np.abs(losses - g64["losses"]) / np.abs(g64["losses"])))
            print(f"   {tag:8s} graph={int(graph)} image err {ei:.2e} (reference noise {ni:.2e}, ratio {ei / ni:.2f})  losses err {el:.2e} (noise {nl:.2e}, ratio {el / max(nl, 1e-30):.2f})")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "free":
    free_running()
