# (round 3 diagnostic; see profiles/r03_experiments.txt 17-18 and DESIGN.md section 4)
import sys, os, hashlib
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch, numpy as np
import r3_cases as R
from maxstyle_amd import synthetic as syn
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "dirty":      # fill the caching allocator's pool with garbage first
    junk = [torch.full((1 << 24,), float(i + 1) * 1e3, device=dev) for i in range(40)]
    del junk
B, layers, K = 16, [3, 4, 5], 5
spec = syn.NetSpec(4, 1, 4)
S = R.trained_solver(dev, "trained_fcn16_256.npz")
img, lab = syn.synthetic_batch(B, 256, 1, 4, seed=1234)
img_d, lab_d = img.to(dev), lab.to(dev)
styles = {i: syn.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
def hook(mods):
    for k, m in mods.items():
        st = styles[int(k)]
        m.perm = st.perm.clone()
        with torch.no_grad():
            m.gamma_noise.data = st.gamma_noise.to(dev); m.beta_noise.data = st.beta_noise.to(dev); m.lmda.data = st.lmda.to(dev)
S.style_init_hook = hook
z_i, _ = S.encode_image(img_d, disable_track_bn_stats=True)
for rep in range(2):
    out = S.generate_max_style_image(z_i.detach(), layers, spec.channel_num, p=1.5, n_iter=K, lr=0.1, reference_image=img_d, reference_segmentation=lab_d)
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    print("rep", rep, "image sha", h, "losses", ["%.9f" % float(x) for x in S.last_losses.cpu()])
