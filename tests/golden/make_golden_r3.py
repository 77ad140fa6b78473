"""Round-3 golden vectors, produced by running the REFERENCE here (needs /root/reference; older fixtures are left untouched).

    python tests/golden/make_golden_r3.py [twins] [args] [train256] [full]      (full needs train256's output)

train256 -> trained_fcn16_256.npz : trained_fcn16.npz fine-tuned at 256x256 by the reference's own training step (240 iterations).

full  -> loop_full_c2.npz : the reference's generate_max_style_image (advanced_triplet...py:458-571) at the BENCHMARKED size - BASELINE config 2,
         16x1x256x256, layers [3,4,5], K = 5 - on the trained FCN_16 weights of trained_fcn16.npz (the networks are fully convolutional), in fp32
         and in fp64: losses, style parameters after every step, the fp64 image (stored as fp32), the fp32 run's distance from it (the reference's OWN
         noise: the calibration of the GPU bound), argmax labels of the segmentation of the stylised image (uint8), clean / stylised Dice.
args  -> loop_args.npz    : one case per non-default argument of the drop-in signature (advanced_triplet...py:458-466, maxstyle.py:75-117), each through
         generate_max_style_image at 4x1x64x64 on the trained weights, fp32 + fp64: mix_style=False; no_noise=True with noise_learnable=False;
         mix_learnable=False; noise_learnable=False; always_use_beta=True (lmda DRAWN by the reference's Beta(0.1,0.1) sampler under fix_seed, all
         draws on the CPU generator so the GPU product must reproduce them); loss_weights=[0.5]; two 'seg' terms; n_iter=0.
twins -> loop_*_tf64.npz  : TEACHER-FORCED fp64 twins of loop_c2small / loop_c4small / loop_all_layers: for every step s the reference is re-run in fp64
         for ONE step from the fp32 run's parameters after step s-1 (and its frozen gamma_std / beta_std), so every step's fp32 gradient has an fp64
         value at the same point: the per-step, per-tensor noise that calibrates the GPU bars.
Fixtures are data only.  The reference is imported in place, never copied.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from make_golden import inject, build_reference_solver  # noqa: E402
from oracle import maxstyle_oracle as orc  # noqa: E402

NETS = ("image_encoder", "segmentation_decoder", "image_decoder")
PNAMES = ("gamma_noise", "beta_noise", "lmda")


def trained_reference(solver_mod, dtype, weights="trained_fcn16.npz", **solver_kw):
    """A reference solver holding the weights of a trained_fcn16*.npz (fp16-stored; both sides load exactly these values), train mode."""
    store = np.load(os.path.join(HERE, weights))
    with contextlib.redirect_stdout(io.StringIO()):
        R = solver_mod.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=False, **solver_kw)
    for n in NETS:
        sd = {}
        for k in R.model[n].state_dict():
            a = store[f"{n}/{k}"]
            sd[k] = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
        R.model[n].load_state_dict(sd, strict=True)
        R.model[n].train()
        if dtype == torch.float64:
            R.model[n].double()
    return R


class Spy:
    """Captures losses (as -CE of every term), gradients before and parameters after every Adam step of a reference call."""

    def __init__(self, solver_mod):
        self.mod = solver_mod
        self.losses, self.params, self.grads, self.pnames = [], [], [], []

    def __enter__(self):
        self._step = torch.optim.Adam.step
        self._loss = self.mod.basic_loss_fn
        spy = self

        def step_spy(opt, *a, **k):
            ps = [p for g in opt.param_groups for p in g["params"]]
            spy.grads.append([None if p.grad is None else p.grad.detach().clone() for p in ps])
            r = spy._step(opt, *a, **k)
            spy.params.append([p.detach().clone() for p in ps])
            return r

        def loss_spy(*a, **k):
            l = spy._loss(*a, **k)
            spy.losses.append(-float(l))
            return l
        torch.optim.Adam.step = step_spy
        self.mod.basic_loss_fn = loss_spy
        return self

    def __exit__(self, *exc):
        torch.optim.Adam.step = self._step
        self.mod.basic_loss_fn = self._loss
        self.mod.CpuMaxStyle.post_init_hook = None


def segment(S, image):
    with torch.no_grad():
        _, zs = S.encode_image(image, disable_track_bn_stats=True)
        return S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs, disable_track_bn_stats=True)


def sample_idx(n, m=4096):
    return torch.linspace(0, n - 1, m).long()


# ----------------------------------------------------------------------------------------------------------------- full size
def train_256(solver_mod, iters=240, B=4, size=256, lr=5e-4):
    """trained_fcn16_256.npz: trained_fcn16.npz (trained at 64x64, where the synthetic anatomy is 4x smaller in pixels: clean Dice at 256x256 is ~0.1)
    fine-tuned at the BENCHMARKED resolution by the reference's OWN training step (standard_training -> backward -> optimize_all_params,
    train_adv...py:163-199,532-535) on the synthetic stream (seeds 7000+it: never the bench batch, seed 1234), stored fp16-rounded like its parent."""
    torch.set_num_threads(8)
    torch.manual_seed(0)
    S = trained_reference(solver_mod, torch.float32, optimizer_type="AdamW", learning_rate=lr)
    S.train()
    for it in range(iters):
        clean, lab = orc.synthetic_batch(B, size, 1, 4, seed=7000 + it)
        g = torch.Generator().manual_seed(11000 + it)
        noisy = torch.clamp(clean + 0.05 * torch.randn(clean.shape, generator=g), 0.0, 1.0)
        with contextlib.redirect_stdout(io.StringIO()):
            S.reset_all_optimizers()
            seg, rec, gt, sh = S.standard_training(clean, lab, perturbed_image=noisy)
            loss = seg + rec + gt + sh
            loss.backward()
            S.optimize_all_params()
        if it % 20 == 0 or it == iters - 1:
            print(f"train256 it {it}: seg {float(seg):.4f} rec {float(rec):.5f}", flush=True)
    store = {}
    for n in NETS:
        for k, v in S.model[n].state_dict().items():
            store[f"{n}/{k}"] = v.numpy().astype(np.float16) if v.is_floating_point() else v.numpy()
    np.savez_compressed(os.path.join(HERE, "trained_fcn16_256.npz"), **store)
    img, lab = orc.synthetic_batch(16, size, 1, 4, seed=1234)
    R = trained_reference(solver_mod, torch.float32, "trained_fcn16_256.npz")
    pred = segment(R, img).argmax(1)
    print("trained_fcn16_256.npz", os.path.getsize(os.path.join(HERE, "trained_fcn16_256.npz")), "clean Dice on the bench batch:",
          orc.dice_per_class(pred, lab, 4), flush=True)


def full_size(solver_mod):
    torch.set_num_threads(8)
    spec = orc.NetSpec(4, 1, 4)
    B, size, layers, K = 16, 256, [3, 4, 5], 5
    img, lab = orc.synthetic_batch(B, size, 1, 4, seed=1234)          # the bench workload's batch (bench.py, rank 0)
    res = {"layers": np.array(layers), "K": np.array(K)}
    images = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        R = trained_reference(solver_mod, dtype, "trained_fcn16_256.npz")
        x = img.to(dtype)
        states = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, dtype) for i in layers}
        with torch.no_grad():
            z_i, _ = R.encode_image(x, disable_track_bn_stats=True)
        Cpu = solver_mod.CpuMaxStyle
        Cpu.created = []
        Cpu.post_init_hook = staticmethod(lambda layer, idx: inject(layer, states[layers[idx]].clone(), dtype))
        with Spy(solver_mod) as spy, contextlib.redirect_stdout(io.StringIO()):
            out = R.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=1.5, n_iter=K, lr=0.1,
                                             reference_image=x, reference_segmentation=lab)
        images[tag] = out
        clean_pred = segment(R, x).argmax(1)
        sty_pred = segment(R, out).argmax(1)
        res[f"{tag}.losses"] = np.array(spy.losses, np.float64)
        res[f"{tag}.clean_dice"] = np.array(orc.dice_per_class(clean_pred, lab, 4))
        res[f"{tag}.final_dice"] = np.array(orc.dice_per_class(sty_pred, lab, 4))
        res[f"{tag}.final_pred"] = sty_pred.numpy().astype(np.uint8)
        res[f"{tag}.clean_pred"] = clean_pred.numpy().astype(np.uint8)
        names = [f"{i}.{n}" for i in layers for n in PNAMES]
        for s_, ps in enumerate(spy.params):
            for n, p_ in zip(names, ps):
                res[f"{tag}.step{s_ + 1}.param.{n}"] = p_.numpy().astype(np.float32 if tag == "f32" else np.float64)
        for i, layer in zip(layers, Cpu.created):
            res[f"{tag}.{i}.gamma_std"] = layer.gamma_std.numpy().astype(np.float64)
            res[f"{tag}.{i}.beta_std"] = layer.beta_std.numpy().astype(np.float64)
        zf = z_i.reshape(-1)
        res[f"{tag}.z_i.sample"] = zf[sample_idx(zf.numel())].numpy().astype(np.float64)
        res[f"{tag}.z_i.stats"] = np.array([float(zf.mean()), float(zf.std()), float(zf.abs().max())])
        print(tag, "losses", spy.losses, "clean dice", res[f"{tag}.clean_dice"], "stylised dice", res[f"{tag}.final_dice"], flush=True)
    i64 = images["f64"]
    i32 = images["f32"].double()
    res["f64.image"] = i64.numpy().astype(np.float32)                 # 4 MB: the value every run is measured against
    d = (i32 - i64)
    scale = float(i64.abs().max())
    res["image_scale"] = np.array(scale)
    res["ref_noise.image_max"] = np.array(float(d.abs().max()) / scale)
    res["ref_noise.image_rms"] = np.array(float(d.pow(2).mean().sqrt()) / scale)
    res["ref_noise.image_max_per_sample"] = (d.abs().amax(dim=(1, 2, 3)) / scale).numpy()
    res["ref_noise.losses_rel"] = np.abs(res["f32.losses"] - res["f64.losses"]) / np.abs(res["f64.losses"])
    res["ref_noise.labels_equal"] = np.array(float((res["f32.final_pred"] == res["f64.final_pred"]).mean()))
    res["f32.image.sample"] = images["f32"][:, :, ::4, ::4].numpy()   # a strided sample of the fp32 run's image (its full error is in ref_noise.*)
    print("reference fp32-vs-fp64 noise at full size: image max", float(res["ref_noise.image_max"]), "rms", float(res["ref_noise.image_rms"]),
          "losses", res["ref_noise.losses_rel"], "labels equal", float(res["ref_noise.labels_equal"]), flush=True)
    np.savez_compressed(os.path.join(HERE, "loop_full_c2.npz"), **res)
    print("loop_full_c2.npz", os.path.getsize(os.path.join(HERE, "loop_full_c2.npz")), flush=True)


# ----------------------------------------------------------------------------------------------------------------- arguments
ARG_CASES = {
    # tag: (kwargs of generate_max_style_image, inject perm/noise/lmda?)
    "nomix": (dict(mix_style=False), True),
    "nonoise": (dict(no_noise=True, noise_learnable=False), True),
    "mixfixed": (dict(mix_learnable=False), True),
    "noisefixed": (dict(noise_learnable=False), True),
    "lw05": (dict(loss_weights=[0.5]), True),
    "twoterms": (dict(loss_types=["seg", "seg"], loss_weights=[0.25, 0.5]), True),
    "k0": (dict(n_iter=0), True),
    "lr003": (dict(lr=0.03, n_iter=2), True),
    # everything drawn by the reference itself under fix_seed; with noise_learnable=False every draw comes from the CPU generator
    # (randperm, rand(1), Beta sample: maxstyle.py:54-61, 105-107), so a GPU run of the product must reproduce perm / rand_p / lmda
    "beta_drawn": (dict(always_use_beta=True, noise_learnable=False, fix_seed=11, p=0.8), False),
    # Beta(0.1, 0.1) values (mass at 0 and 1: the clamp boundary of d lmda) with injected noise, everything learnable
    "beta_injected": (dict(always_use_beta=True), "beta"),
}


# all six decoder layers on the TRAINED network (the random-weight `loop_all_layers` fixture is badly conditioned - the reference's own fp32 run is
# 7e-2 from its fp64 twin there); its own fixture so that loop_args.npz keeps its keys
ALL6_CASES = {"all6": (dict(), True)}


def arg_cases(solver_mod, cases=None, out_name="loop_args.npz", layers=(3, 4, 5), seed0=4000):
    torch.set_num_threads(1)
    spec = orc.NetSpec(4, 1, 4)
    B, size, layers = 4, 64, list(layers)
    cases = ARG_CASES if cases is None else cases
    img, lab = orc.synthetic_batch(B, size, 1, 4, seed=777)
    res = {}
    for case, (kw, inj) in cases.items():
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            R = trained_reference(solver_mod, dtype)
            x = img.to(dtype)
            states = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i, dtype) for i in layers}
            if inj == "beta":
                for i in layers:
                    # Beta(0.1,0.1) through its two-gamma form on an explicit generator is not available: draw with the global generator
                    torch.manual_seed(100 + i)
                    states[i].lmda = torch.distributions.Beta(0.1, 0.1).sample((B, 1, 1, 1)).to(dtype)
            if inj and tag == "f32":
                for i in layers:
                    res[f"{case}.initial.{i}.lmda"] = states[i].lmda.numpy().astype(np.float32)
            with torch.no_grad():
                z_i, _ = R.encode_image(x, disable_track_bn_stats=True)
            Cpu = solver_mod.CpuMaxStyle
            Cpu.created = []
            if inj:
                def hook(layer, idx, states=states):
                    st = states[layers[idx]].clone()
                    # keep the layer's own structure (which tensors are Parameters / learnable / present): overwrite values only
                    layer.perm = st.perm.clone()
                    with torch.no_grad():
                        if isinstance(layer.gamma_noise, torch.nn.Parameter):
                            layer.gamma_noise.data = st.gamma_noise.to(dtype)
                            layer.beta_noise.data = st.beta_noise.to(dtype)
                        elif layer.no_noise:
                            # fixed N(0,1) tensors the forward never uses (maxstyle.py:75-77, 181-182): leave the reference's draw in place
                            pass
                        if isinstance(layer.lmda, torch.nn.Parameter):
                            layer.lmda.data = st.lmda.to(dtype)
                Cpu.post_init_hook = staticmethod(hook)
            else:
                def hook64(layer, idx):
                    # drawn state is fp32 by construction (maxstyle.py:75-117); the fp64 twin casts it
                    if dtype == torch.float64:
                        for nm in PNAMES:
                            t = getattr(layer, nm)
                            if isinstance(t, torch.nn.Parameter):
                                t.data = t.data.double()
                            else:
                                object.__setattr__(layer, nm, t.double())
                Cpu.post_init_hook = staticmethod(hook64)
            call = dict(decoder_layers_indexes=list(layers), channel_num=spec.channel_num, p=1.5, n_iter=3, lr=0.1,
                        reference_image=x, reference_segmentation=lab)
            call.update(kw)
            # nuisance draws that no test reads (rand_p under p = 1.5, the N(0,1) tensors the `no_noise` forward never uses) come from the global generator: pin its
            # position so that every key of the fixture regenerates bit for bit (cases with `fix_seed` re-seed inside the call anyway)
            torch.manual_seed(seed0 + sorted(cases).index(case))
            with Spy(solver_mod) as spy, contextlib.redirect_stdout(io.StringIO()):
                out = R.generate_max_style_image(z_i, **call)
            pre = f"{case}.{tag}."
            res[pre + "image"] = out.numpy().astype(np.float32)        # (the fp64 run's image stored as fp32: 1e-7 is far below every bar)
            res[pre + "losses"] = np.array(spy.losses, np.float64)
            if tag == "f32":
                res["z_i"] = z_i.numpy()                               # the same code for every case (same weights, same batch)
                pred = segment(R, out).argmax(1)
                res[f"{case}.final_pred"] = pred.numpy().astype(np.uint8)
                res[f"{case}.final_dice"] = np.array(orc.dice_per_class(pred, lab, 4))
            # the optimiser's parameter list in the reference's order: every nn.Parameter of the ModuleDict (learnable or not)
            names = []
            for i, layer in zip(layers, Cpu.created):
                names += [f"{i}.{n}" for n, _ in layer.named_parameters()]
                res[pre + f"{i}.applied"] = np.array(bool(layer.rand_p < layer.p))
                res[pre + f"{i}.perm"] = layer.perm.numpy()
                res[pre + f"{i}.rand_p"] = layer.rand_p.numpy()
                if layer.gamma_std is not None:
                    res[pre + f"{i}.gamma_std"] = layer.gamma_std.numpy()
                    res[pre + f"{i}.beta_std"] = layer.beta_std.numpy()
                for nm in PNAMES:                              # final values of all three tensors, Parameter or not
                    res[pre + f"final.{i}.{nm}"] = getattr(layer, nm).detach().numpy()
            res[f"{case}.param_names"] = np.array(names)
            for s_, (ps, gs) in enumerate(zip(spy.params, spy.grads)):
                for n, p_, g_ in zip(names, ps, gs):
                    res[pre + f"step{s_ + 1}.param.{n}"] = p_.numpy()
                    if g_ is not None:
                        res[pre + f"step{s_ + 1}.grad.{n}"] = g_.numpy()
            if not inj and tag == "f32":
                # initial drawn values are not recoverable after the loop: re-draw them the way the call did (fix_seed, CPU generator)
                torch.manual_seed(kw["fix_seed"])
                for i in layers:
                    m = solver_mod.CpuMaxStyle.__mro__[1](B, spec.channel_num[i], p=kw["p"], mix_style=True, no_noise=False, mix_learnable=True,
                                                          noise_learnable=False, always_use_beta=True, use_gpu=False)
                    res[f"{case}.initial.{i}.lmda"] = m.lmda.detach().numpy()
                    assert torch.equal(m.perm, Cpu.created[layers.index(i)].perm)
            print(case, tag, "losses", spy.losses, "params", names, flush=True)
    np.savez_compressed(os.path.join(HERE, out_name), **res)
    print(out_name, os.path.getsize(os.path.join(HERE, out_name)), flush=True)


# ----------------------------------------------------------------------------------------------------------------- teacher-forced fp64 twins
def teacher_forced_twin(solver_mod, fixture, spec, B, size, layers, K):
    """For every step s: the reference in fp64, ONE step, from the fp32 fixture's parameters after step s-1 and its frozen gamma_std / beta_std."""
    torch.set_num_threads(1)
    g = np.load(os.path.join(HERE, fixture))
    S, W = build_reference_solver(solver_mod, spec, torch.float64)
    img, lab = orc.synthetic_batch(B, size, spec.image_ch, spec.num_classes, seed=1234)
    x = img.double()
    with torch.no_grad():
        z_i, _ = S.encode_image(x, disable_track_bn_stats=True)
    chn = spec.channel_num
    init = {i: orc.random_style_state(B, chn[i], 7 + i, torch.float64) for i in layers}
    res = {"losses": []}
    for s in range(1, K + 1):
        def hook(layer, idx, s=s):
            i = layers[idx]
            st = init[i].clone()
            if s > 1:
                for nm in PNAMES:
                    setattr(st, nm, torch.from_numpy(g[f"step{s - 1}.param.{i}.{nm}"]).double())
            inject(layer, st, torch.float64)
            layer.gamma_std = torch.from_numpy(g[f"{i}.gamma_std"]).double()
            layer.beta_std = torch.from_numpy(g[f"{i}.beta_std"]).double()
        Cpu = solver_mod.CpuMaxStyle
        Cpu.created = []
        Cpu.post_init_hook = staticmethod(hook)
        with Spy(solver_mod) as spy, contextlib.redirect_stdout(io.StringIO()):
            S.generate_max_style_image(z_i, decoder_layers_indexes=list(layers), channel_num=chn, p=1.5, n_iter=1, lr=0.1,
                                       reference_image=x, reference_segmentation=lab)
        names = [f"{i}.{n}" for i in layers for n in PNAMES]
        for n, gr in zip(names, spy.grads[0]):
            res[f"step{s}.grad.{n}"] = gr.numpy()
        res["losses"].append(spy.losses[0])
    res["losses"] = np.array(res["losses"], np.float64)
    out = fixture.replace(".npz", "_tf64.npz")
    np.savez_compressed(os.path.join(HERE, out), **res)
    noise = {n: [float(np.abs(g[f"step{s}.grad.{n}"] - res[f"step{s}.grad.{n}"]).max() / (np.abs(res[f"step{s}.grad.{n}"]).max() + 1e-30))
                 for s in range(1, K + 1)] for n in names}
    print(out, os.path.getsize(os.path.join(HERE, out)), "reference fp32 gradient noise per step:", flush=True)
    for n, v in noise.items():
        print("   ", n, ["%.1e" % e for e in v], flush=True)


def twins(solver_mod):
    teacher_forced_twin(solver_mod, "loop_c2small.npz", orc.NetSpec(4, 1, 4), 4, 64, [3, 4, 5], 5)
    teacher_forced_twin(solver_mod, "loop_c4small.npz", orc.NetSpec(1, 3, 2), 4, 64, [3, 4, 5], 3)
    teacher_forced_twin(solver_mod, "loop_all_layers.npz", orc.NetSpec(4, 1, 4), 3, 64, [0, 1, 2, 3, 4, 5], 2)


def main():
    what = sys.argv[1:] or ["twins", "args", "full"]
    solver_mod = ref_harness.load_solver_module()
    if "twins" in what:
        twins(solver_mod)
    if "args" in what:
        arg_cases(solver_mod)
    if "all6" in what:
        arg_cases(solver_mod, ALL6_CASES, "loop_args_all6.npz", (0, 1, 2, 3, 4, 5), 4100)
    if "train256" in what:
        train_256(solver_mod)
    if "full" in what:
        full_size(solver_mod)


if __name__ == "__main__":
    main()
