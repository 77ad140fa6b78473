import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from oracle import maxstyle_oracle as orc
from test_solver_gpu import make_solver, injector
dev = torch.device("cuda:0")
spec = orc.NetSpec(4, 1, 4)
S, W = make_solver(dev, spec)
B, size = 16, 256
img, lab = orc.synthetic_batch(B, size, 1, 4, 1234)
layers = [3, 4, 5]
styles = {i: orc.random_style_state(B, spec.channel_num[i], 7 + i) for i in layers}
S.style_init_hook = injector(styles, dev)
z_i, _ = S.encode_image(img.to(dev), disable_track_bn_stats=True)
kw = dict(image_code=z_i, decoder_layers_indexes=layers, channel_num=spec.channel_num, p=1.5, lr=0.1, reference_image=img.to(dev), reference_segmentation=lab.to(dev))
res = {}
for tag, graph in (("eager", False), ("graph1", True), ("graph2", True), ("eager2", False)):
    out = S.generate_max_style_image(n_iter=5, use_graph=graph, **kw)
    res[tag] = (out.clone(), S.last_losses.clone())
    print(tag, S.last_losses.tolist())
for tag in ("graph1", "graph2", "eager2"):
    print(tag, "max |d image| vs eager:", float((res[tag][0] - res["eager"][0]).abs().max()))
