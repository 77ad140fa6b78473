"""cProfile of the host side of generate_max_style_image (config 2, deferred error check)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maxstyle_amd as M
from maxstyle_amd import synthetic as syn
dev = torch.device("cuda:0")
S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
S.loop_error_check = "deferred"
clean, lab = syn.synthetic_batch(16, 256, 1, 4, 1234)
clean, lab = clean.to(dev), lab.to(dev)
cfg = {"mix_style": True, "no_noise": False, "lr": 0.1, "n_iter": 5, "mix_learnable": True, "noise_learnable": True,
       "decoder_layers_indexes": [3, 4, 5], "loss_types": ["seg"], "loss_weights": [1], "always_use_beta": False}
S.eval()
with torch.no_grad():
    z_i, _ = S.encode_image(clean, disable_track_bn_stats=True)
for it in range(5):
    out = S.generate_max_style_image_from_config(z_i, cfg, clean, lab, p=1.5)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(20):
    out = S.generate_max_style_image_from_config(z_i, cfg, clean, lab, p=1.5)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
