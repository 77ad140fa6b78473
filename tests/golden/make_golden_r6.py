"""Round-6 golden vectors, produced by running the REFERENCE here (needs /root/reference; older fixtures are left untouched).

    python tests/golden/make_golden_r6.py [image0] [c5_final]

VERDICT r5 missing 2 / next 3: a DETERMINISTIC bar on the augmented image at every full size.  The free-running K-step loop is chaotic (Adam's first steps are sign-like), so
the image it returns can only be compared with draw-calibrated bars; the decode itself - apply_max_style (encoder_decoder.py:598-631) through MaxStyle.forward
(maxstyle.py:157-188) - is a smooth map of (code, style parameters, frozen batch std).  Two points of that map per call:
  * the FINAL point: the parameters the reference's fp64 run holds after its K-th step and the batch std its first forward froze -> `f64.image` (already in
    loop_full_c2.npz / loop_full_c4.npz / loop_shipped_*.npz; for config 5's two calls the parameters were not stored:)
c5_final -> loop_c5_final_f64.npz : the fp64 twins of both config-5 calls once more (make_golden_r5.py c5f64's calls: same batch, same fix_seed draw), now keeping, per
            APPLIED layer, the parameters after the K-th step (`{tag}.final.param.{i}.{name}`), the frozen batch std (`{tag}.{i}.gamma_std / beta_std`) and the perm; the run
            is checked against loop_c5_calls_f64.npz (same losses to 1e-12, the same strided image bit for bit), whose `{tag}.image.strided / .mean / .rms` are its image.
  * the INITIAL point: generate_max_style_image(n_iter = 0) - one decode at the injected parameters, the first forward (which computes the batch std) -
image0   -> loop_image0_f64.npz : `{tag}.image0.strided{s}` (every s-th pixel: s = 2 for c2 16x1x256x256, acdc192 20x1x192x192, prostate224 20x1x224x224 - always_use_beta,
            lmda from the shipped fixture - and config 5's ACDC-shaped call; s = 4 for c4 16x3x320x320 and config 5's Prostate-shaped call) + the per-plane mean / rms of
            the WHOLE image, all in fp64 arithmetic, stored fp32; `{tag}.image_scale`, `{tag}.stride`.  (Whole images would be 19 MB.)
Fixtures are data only.  The reference is imported in place, never copied.
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from make_golden import inject  # noqa: E402
from make_golden_r3 import Spy, PNAMES, trained_reference  # noqa: E402
from make_golden_r5 import reference_p224, _c5_specs, SPEC_A, SPEC_P, NTHREADS  # noqa: E402
from oracle import maxstyle_oracle as orc  # noqa: E402


def _planes(res, key, out):
    res[f"{key}.mean"] = out.mean(dim=(2, 3)).numpy()
    res[f"{key}.rms"] = out.pow(2).mean(dim=(2, 3)).sqrt().numpy()


def image0(solver_mod):
    """One decode at the injected state (n_iter = 0), fp64."""
    from make_golden_r4 import trained_reference64, SPEC4
    torch.set_num_threads(NTHREADS)
    res = {}
    layers = [3, 4, 5]
    shipped_p = np.load(os.path.join(HERE, "loop_shipped_prostate.npz"))
    cases = (("c2", lambda dt: trained_reference(solver_mod, dt, "trained_fcn16_256.npz"), SPEC_A, 256, 16, None, 2),
             ("acdc192", lambda dt: trained_reference(solver_mod, dt, "trained_fcn16_192.npz"), SPEC_A, 192, 20, None, 2),
             ("prostate224", lambda dt: reference_p224(solver_mod, dt), SPEC_P, 224, 20, shipped_p, 2),
             ("c4", lambda dt: trained_reference64(solver_mod, dt), SPEC4, 320, 16, None, 4))
    for tag, mk, spec, size, B, lm_src, stride in cases:
        t0 = time.time()
        dtype = torch.float64
        img, lab = orc.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=1234)
        R = mk(dtype)
        x = img.to(dtype)
        states = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, dtype) for i in layers}
        if lm_src is not None:
            for i in layers:
                states[i].lmda = torch.from_numpy(lm_src[f"initial.{i}.lmda"]).to(dtype)
        with torch.no_grad():
            z_i, _ = R.encode_image(x, disable_track_bn_stats=True)
        Cpu = solver_mod.CpuMaxStyle
        Cpu.created = []
        Cpu.post_init_hook = staticmethod(lambda layer, idx: inject(layer, states[layers[idx]].clone(), dtype))
        torch.manual_seed(5000)
        with Spy(solver_mod), contextlib.redirect_stdout(io.StringIO()):
            out = R.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=1.5, n_iter=0, lr=0.1,
                                             always_use_beta=lm_src is not None, reference_image=x, reference_segmentation=lab)
        res[f"{tag}.image_scale"] = np.array(float(out.abs().max()))
        res[f"{tag}.stride"] = np.array(stride)
        res[f"{tag}.image0.strided"] = out[:, :, ::stride, ::stride].numpy().astype(np.float32)
        _planes(res, f"{tag}.image0", out)
        for i, layer in zip(layers, Cpu.created):
            res[f"{tag}.{i}.gamma_std"] = layer.gamma_std.numpy().astype(np.float64).reshape(-1)
            res[f"{tag}.{i}.beta_std"] = layer.beta_std.numpy().astype(np.float64).reshape(-1)
        print(tag, f"image0 ({time.time() - t0:.0f} s): scale", float(res[f"{tag}.image_scale"]), "mean", float(out.mean()), flush=True)
    # config 5's two calls: the applied subset is the reference's own draw under fix_seed
    g32 = np.load(os.path.join(HERE, "loop_c5_calls.npz"))
    for tag, mk, spec, size, K, seed, fix in _c5_specs(solver_mod):
        t0 = time.time()
        out, created = _c5_run(solver_mod, mk, spec, size, 0, seed, fix)
        applied = np.array([bool(l.rand_p < l.p) for l in created])
        assert np.array_equal(applied, g32[f"{tag}.applied"]), (applied, g32[f"{tag}.applied"])
        stride = 2 if size <= 256 else 4
        res[f"c5{tag}.image_scale"] = np.array(float(out.abs().max()))
        res[f"c5{tag}.stride"] = np.array(stride)
        res[f"c5{tag}.image0.strided"] = out[:, :, ::stride, ::stride].numpy().astype(np.float32)
        _planes(res, f"c5{tag}.image0", out)
        print("c5", tag, f"image0 ({time.time() - t0:.0f} s): applied", applied, flush=True)
    path = os.path.join(HERE, "loop_image0_f64.npz")
    np.savez_compressed(path, **res)
    print("loop_image0_f64.npz", os.path.getsize(path), flush=True)


def _c5_run(solver_mod, mk, spec, size, K, seed, fix):
    """make_golden_r5._c5_call in fp64, returning the created layers (their parameters after the K-th step, their frozen std)."""
    dtype = torch.float64
    B, layers = 16, [3, 4, 5]
    img, lab = orc.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=seed)
    R = mk(dtype)
    x = img.to(dtype)
    with torch.no_grad():
        z_i, _ = R.encode_image(x, disable_track_bn_stats=True)
    states = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, torch.float32) for i in layers}
    Cpu = solver_mod.CpuMaxStyle
    Cpu.created = []

    def hook(layer, idx):
        if "gamma_noise" in layer._parameters:
            st = states[layers[idx]]
            with torch.no_grad():
                layer.gamma_noise.data = st.gamma_noise.clone().to(dtype); layer.beta_noise.data = st.beta_noise.clone().to(dtype); layer.lmda.data = st.lmda.clone().to(dtype)
    Cpu.post_init_hook = staticmethod(hook)
    with Spy(solver_mod) as spy, contextlib.redirect_stdout(io.StringIO()):
        out = R.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=0.5, n_iter=K, lr=0.1,
                                         reference_image=x, reference_segmentation=lab, fix_seed=fix)
    _c5_run.losses = list(spy.losses)
    return out, list(Cpu.created)


def c5_final(solver_mod):
    torch.set_num_threads(NTHREADS)
    g64 = np.load(os.path.join(HERE, "loop_c5_calls_f64.npz"))
    res = {}
    for tag, mk, spec, size, K, seed, fix in _c5_specs(solver_mod):
        t0 = time.time()
        out, created = _c5_run(solver_mod, mk, spec, size, K, seed, fix)
        losses = np.array(_c5_run.losses, np.float64)
        assert np.allclose(losses, g64[f"{tag}.losses"], rtol=1e-12, atol=0.0), (losses, g64[f"{tag}.losses"])      # the same run as the committed fp64 twin
        assert np.array_equal(out[:, :, ::4, ::4].numpy().astype(np.float32), g64[f"{tag}.image.strided"])
        res[f"{tag}.applied"] = np.array([bool(l.rand_p < l.p) for l in created])
        for i, l in zip([3, 4, 5], created):
            res[f"{tag}.{i}.perm"] = l.perm.numpy()
            if bool(l.rand_p < l.p):
                for n in PNAMES:
                    res[f"{tag}.final.param.{i}.{n}"] = getattr(l, n).detach().numpy().astype(np.float64)
                res[f"{tag}.{i}.gamma_std"] = l.gamma_std.numpy().astype(np.float64).reshape(-1)
                res[f"{tag}.{i}.beta_std"] = l.beta_std.numpy().astype(np.float64).reshape(-1)
        print(tag, f"c5 final ({time.time() - t0:.0f} s): losses", losses, "applied", res[f"{tag}.applied"], flush=True)
        np.savez_compressed(os.path.join(HERE, "loop_c5_final_f64.npz"), **res)
    print("loop_c5_final_f64.npz", os.path.getsize(os.path.join(HERE, "loop_c5_final_f64.npz")), flush=True)


def main():
    solver_mod = ref_harness.load_solver_module()
    what = sys.argv[1:] or ["image0", "c5_final"]
    for w in what:
        {"image0": image0, "c5_final": c5_final}[w](solver_mod)


if __name__ == "__main__":
    main()
