"""bf16 activation storage on the TRAINED FCN_16 fixture (tests/golden/trained_fcn16.npz, loop_trained.npz): K=5 loop vs the reference's fp64 run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import maxstyle_amd as M
from oracle import maxstyle_oracle as orc
from test_round2_gpu import load_trained
from test_solver_gpu import injector
from parity_util import rel
dev = torch.device("cuda:0")
gd = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gd, "loop_trained.npz"))
W = load_trained(gd)
spec = orc.NetSpec(4, 1, 4)
img, lab = orc.synthetic_batch(4, 64, 1, 4, 777)
layers = [3, 4, 5]
for name, dt in (("fp32", None), ("bf16", torch.bfloat16)):
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True)
    for n_, mod in S.model.items():
        mod.load_state_dict(W[n_], strict=True); mod.train()
    S.loop_act_dtype = dt
    styles = {i: orc.random_style_state(4, spec.channel_num[i], 7 + i) for i in layers}
    S.style_init_hook = injector(styles, dev)
    z_i, z_s = S.encode_image(img.to(dev), disable_track_bn_stats=True)
    out = S.generate_max_style_image(z_i, layers, spec.channel_num, p=1.5, n_iter=5, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
    losses = S.last_losses.cpu().numpy()
    _, zs2 = S.encode_image(out, disable_track_bn_stats=True)
    logits = S.decoder_inference(decoder=S.model["segmentation_decoder"], latent_code=zs2, disable_track_bn_stats=True)
    dice = orc.dice_per_class(logits.argmax(1).cpu(), lab, 4)
    agree = float((logits.argmax(1).cpu().numpy() == g["f32.final_pred"]).mean())
    print(name, "losses", losses, "ref", g["f32.losses"])
    print(name, "image rel err vs ref fp64", rel(out, g["f64.image"]), "ref fp32-vs-fp64", float(g["fp32_vs_fp64_image_rel"]), "dice", dice, "ref dice", g["f32.final_dice"], "agree", agree)
