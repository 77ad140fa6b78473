"""Fidelity DISTRIBUTION of the full-size training pass's weight gradients (16x1x256x256, FCN_16): over N input seeds, per parameter
tensor, the distance from the fp64 CPU oracle (oracle/outer_oracle.py: autograd over the functional forward) of
    the fp32 CPU oracle (the reference's arithmetic on the host),
    the GPU pass with the direct conv form (the default of the training passes),
    the GPU pass with the Winograd form of the wide convolutions (EngineOptions.train_winograd),
in the max norm and in the L2 norm.

  python tools/train_fidelity.py prep [N]     CPU, build container (~40 s of host time per seed): writes .scratch/train_fid_<seed>.npz = inputs + the fp64 gradients (stored as fp32:
                                              6e-8, far below anything compared) + the fp32 oracle's own errors.  The files travel to the GPU box with the snapshot (not committed).
  python tools/train_fidelity.py fixture      CPU: tests/golden/train_fidelity_oracle32.npz from the prep files - per tensor the fp32 oracle's WORST error over the seeds (max norm, L2):
                                              the calibration of tests/test_train_gpu.py::test_training_pass_full_size_vs_oracle
  python tools/train_fidelity.py gpu          GPU box: the two GPU passes per seed -> the distribution table on stdout and gpurun_out/train_fidelity.json
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
SCR = os.path.join(ROOT, ".scratch")


def oracle_grads(dtype, batch_seed, noise_seed):
    from oracle import maxstyle_oracle as orc, outer_oracle as outer
    spec = orc.NetSpec(4, 1, 4)
    W = orc.procedural_weights(spec, 0, dtype=dtype)
    clean, lab = orc.synthetic_batch(16, 256, 1, 4, batch_seed)
    clean = clean.to(dtype)
    g = torch.Generator().manual_seed(noise_seed)
    noise = (0.05 * torch.randn(clean.shape, generator=g)).to(dtype)
    image_l = outer.noisy_input(clean, noise)
    names = [(n, k) for n in outer.NETS for k in outer.param_names(W[n])]
    for n, k in names:
        W[n][k].requires_grad_(True)
    seg, rec, z_i, z_s, recon, logits = outer.training_pass(W, image_l, clean, lab, track_bn=True)
    grads = torch.autograd.grad(seg + rec, [W[n][k] for n, k in names], allow_unused=True)
    keep = [(n, k) for (n, k), g_ in zip(names, grads) if g_ is not None and not outer.is_null_grad_bias(n, k)]
    return {f"{n}/{k}": g_.detach() for (n, k), g_ in zip(names, grads) if (n, k) in keep}, clean, lab, image_l


def errs(g, ref):
    g, ref = g.double().reshape(-1), ref.double().reshape(-1)
    return float((g - ref).abs().max() / ref.abs().max()), float((g - ref).norm() / ref.norm())


def prep(n):
    os.makedirs(SCR, exist_ok=True)
    torch.set_num_threads(int(os.environ.get("FID_THREADS", "8")))
    for s in range(n):
        path = os.path.join(SCR, f"train_fid_{s}.npz")
        if os.path.exists(path):
            continue
        t0 = time.time()
        bs, ns = (1234, 100) if s == 0 else (52000 + s, 53000 + s)          # seed 0 = the batch of tests/test_train_gpu.py::test_training_pass_full_size_vs_oracle
        g64, clean, lab, image_l = oracle_grads(torch.float64, bs, ns)
        g32, _, _, _ = oracle_grads(torch.float32, bs, ns)
        keys = sorted(g64)
        e = np.array([errs(g32[k], g64[k]) for k in keys])
        np.savez_compressed(path, keys=np.array(keys), clean=clean.float().numpy(), lab=lab.numpy().astype(np.uint8), image_l=image_l.float().numpy(),
                            oracle32_max=e[:, 0], oracle32_l2=e[:, 1], **{"g64/" + k: g64[k].float().numpy() for k in keys})
        print(f"seed {s}: {time.time() - t0:.0f} s; fp32 oracle vs fp64: worst max-norm {e[:, 0].max():.2e} ({keys[int(e[:, 0].argmax())]}), worst L2 {e[:, 1].max():.2e}", flush=True)


def gpu():
    import glob
    from test_train_gpu import make_solver
    from oracle import maxstyle_oracle as orc
    dev = torch.device("cuda:0")
    files = sorted(glob.glob(os.path.join(SCR, "train_fid_*.npz")), key=lambda f: int(f.split("_")[-1].split(".")[0]))
    assert files, "run `python tools/train_fidelity.py prep N` in the build container first"
    table = {"oracle32": [], "gpu_direct": [], "gpu_winograd": []}
    keys = None
    for f in files:
        z = np.load(f)
        keys = [str(k) for k in z["keys"]]
        table["oracle32"].append(np.stack([z["oracle32_max"], z["oracle32_l2"]], 1))
        clean, lab, image_l = torch.from_numpy(z["clean"]).to(dev), torch.from_numpy(z["lab"].astype(np.int64)).to(dev), torch.from_numpy(z["image_l"]).to(dev)
        for tag, wino in (("gpu_direct", False), ("gpu_winograd", True)):
            S, W = make_solver(dev, orc.NetSpec(4, 1, 4))
            S.train_options = {"train_winograd": wino}
            S.reset_all_optimizers()
            out = S.standard_training(clean, lab, perturbed_image=image_l, disable_track_bn_stats=False, return_output=True)
            (out[0] + out[1]).backward()
            torch.cuda.synchronize()
            got = {f"{n}/{k}": p.grad.detach().cpu() for n in S.model for k, p in S.model[n].named_parameters()}
            table[tag].append(np.array([errs(got[k], torch.from_numpy(z["g64/" + k])) for k in keys]))
            del S
    out = {"seeds": len(files), "tensors": len(keys), "what": "weight gradients of one standard_training pass at 16x1x256x256 against the fp64 CPU oracle, per tensor: max norm and L2, relative"}
    print(f"{len(files)} seeds x {len(keys)} tensors; per seed the WORST tensor, then over seeds: median / max   (and the mean over tensors, median over seeds)")
    for tag, rows in table.items():
        a = np.stack(rows)                      # [seed, tensor, 2]
        wm, wl = a[:, :, 0].max(1), a[:, :, 1].max(1)
        out[tag] = {"worst_tensor_max_norm_per_seed": wm.tolist(), "worst_tensor_l2_per_seed": wl.tolist(), "mean_over_tensors_max_norm_per_seed": a[:, :, 0].mean(1).tolist(),
                    "mean_over_tensors_l2_per_seed": a[:, :, 1].mean(1).tolist(),
                    "per_tensor_max_over_seeds_l2": {k: float(v) for k, v in zip(keys, a[:, :, 1].max(0))},
                    "per_tensor_max_over_seeds_max_norm": {k: float(v) for k, v in zip(keys, a[:, :, 0].max(0))}}
        print(f"  {tag:13s} max norm: median {np.median(wm):.2e}  max {wm.max():.2e}   L2: median {np.median(wl):.2e}  max {wl.max():.2e}   "
              f"mean over tensors: max norm {np.median(a[:, :, 0].mean(1)):.2e}  L2 {np.median(a[:, :, 1].mean(1)):.2e}")
    # per tensor: the GPU's worst L2 over seeds against the fp32 oracle's worst L2 over seeds
    o = np.stack(table["oracle32"])[:, :, 1].max(0)
    for tag in ("gpu_direct", "gpu_winograd"):
        r = np.stack(table[tag])[:, :, 1].max(0) / np.maximum(o, 1e-12)
        out[tag]["ratio_to_oracle32_worst_l2_per_tensor"] = {"median": float(np.median(r)), "max": float(r.max()), "argmax": keys[int(r.argmax())]}
        print(f"  {tag}: per tensor, worst L2 over seeds / the fp32 oracle's worst L2 over seeds: median {np.median(r):.2f}, max {r.max():.2f} ({keys[int(r.argmax())]})")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "train_fidelity.json"), "w"), indent=0)


def fixture():
    import glob
    files = sorted(glob.glob(os.path.join(SCR, "train_fid_*.npz")), key=lambda f: int(f.split("_")[-1].split(".")[0]))
    assert files
    keys = [str(k) for k in np.load(files[0])["keys"]]
    mx = np.stack([np.load(f)["oracle32_max"] for f in files]); l2 = np.stack([np.load(f)["oracle32_l2"] for f in files])
    path = os.path.join(ROOT, "tests", "golden", "train_fidelity_oracle32.npz")
    np.savez_compressed(path, keys=np.array(keys), seeds=np.array(len(files)), worst_max_norm=mx.max(0), worst_l2=l2.max(0), median_l2=np.median(l2, 0),
                        per_seed_worst_max_norm=mx.max(1), per_seed_worst_l2=l2.max(1), per_seed_mean_l2=l2.mean(1))
    print(path, os.path.getsize(path), "seeds", len(files), "worst tensor per seed: max norm median %.2e max %.2e; L2 median %.2e max %.2e" %
          (np.median(mx.max(1)), mx.max(), np.median(l2.max(1)), l2.max()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fixture":
        fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == "prep":
        prep(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    elif len(sys.argv) > 1 and sys.argv[1] == "gpu":
        gpu()
    else:
        raise SystemExit(__doc__)
