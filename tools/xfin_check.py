import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import r3_cases as R
from maxstyle_amd import engine as E, synthetic as syn
dev = torch.device("cuda:0")
def run(flag):
    os.environ["MS_XFIN"] = flag
    spec = E.NetSpec(4, 1, 4)
    W = R.load_trained("trained_fcn16.npz")
    to = lambda sd: {k: v.to(dev) for k, v in sd.items()}
    eng = E.InnerLoopEngine(spec, 4, 64, 64, dev, lr=0.1)
    eng.set_nets(E.PackedNets(spec, to(W["image_encoder"]), to(W["segmentation_decoder"]), to(W["image_decoder"])))
    img, lab = syn.synthetic_batch(4, 64, 1, 4, seed=777)
    layers = [3, 4, 5]
    chn = syn.NetSpec(4, 1, 4).channel_num
    eng.configure_styles(layers, {i: E.StyleSlot(i, 4, chn[i]) for i in layers})
    for i in layers:
        st = syn.random_style_state(4, chn[i], 7 + i)
        eng.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
    z = eng.encode_fwd(img.to(dev))[0].clone()
    out = eng.run(z, lab.to(dev), 3, use_graph=True).clone()
    eng.check_errors()
    return out, eng.losses(3).clone(), eng.flat_p.clone()
a = run("1"); b = run("0")
print("equal:", [bool(torch.equal(x, y)) for x, y in zip(a, b)], a[1])
