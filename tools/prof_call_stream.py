"""Back-to-back generate_max_style_image calls at config 2 (no synchronise between them: the whole_call leg of bench.py) under rocprofv3 --kernel-trace --memory-copy-trace is
not allowed with counters, so: kernel trace only.  `python tools/prof_call_stream.py run` issues the calls; `... report <dir>` prints, per call, the time between the last
kernel of one call's graph (step_tail) and the first kernel of the next call's graph, and the kernels that ran in between."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import maxstyle_amd as M
    from maxstyle_amd import synthetic as syn
    dev = torch.device("cuda:0")
    spec = syn.NetSpec(4, 1, 4)
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True)
    W = syn.procedural_weights(spec, 0)
    for name, mod in S.model.items():
        mod.load_state_dict(W[name]); mod.train()
    S.loop_error_check = "deferred"
    img, lab = syn.synthetic_batch(16, 256, 1, 4, 1234)
    img, lab = img.to(dev), lab.to(dev)
    z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
    z_i = z_i.detach()
    for _ in range(16):
        S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, p=1.5, n_iter=5, lr=0.1, reference_image=img, reference_segmentation=lab)
    torch.cuda.synchronize()


def report(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    tails = [i for i, r in enumerate(rows) if "step_tail_kernel" in r[2]]
    ends = tails[4::5]                                   # a call = 5 step tails (+ the re-decode behind the last one)
    for a, b in zip(ends[-5:-1], ends[-4:]):
        seg = rows[a:b]
        wall = (seg[-1][1] - seg[0][1]) / 1e3
        lib = sum(e - s for s, e, n in seg[1:] if "ms::" in n) / 1e3
        oth = [(s_, e_, n_) for s_, e_, n_ in seg[1:] if "ms::" not in n_]
        gaps = sorted(((y[0] - x[1]) / 1e3, x[2][:50], y[2][:50]) for x, y in zip(seg, seg[1:]))[::-1][:6]
        print(f"one call period (tail of step 5 -> tail of the next call's step 5): {wall:.1f} us; library kernels {lib:.1f} us; {len(oth)} other kernels {sum(e - s for s, e, _ in oth) / 1e3:.1f} us; idle {wall - lib - sum(e - s for s, e, _ in oth) / 1e3:.1f} us")
        for s_, e_, n_ in oth:
            print(f"      other: +{(s_ - seg[0][1]) / 1e3:8.1f} us  {(e_ - s_) / 1e3:5.1f} us  {n_[:100]}")
        for g in gaps:
            print(f"      gap {g[0]:7.1f} us  after {g[1]:50s} before {g[2]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
