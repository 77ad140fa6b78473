"""Style-parameter gradients of one inner-loop evaluation: bf16 activation storage against fp32 storage (relative norm error, cosine). python tools/bf16_grad_check.py B size"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bf16_loop_check import build

B, size = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
res = {}
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    eng, img, lab = build(dev, B, size, (4, 1, 4), dt)
    z = eng.encode_fwd(img.to(dt))[0].float().clone()
    eng.code = z.to(dt)
    _, loss = eng.step_grads(lab)
    res[name] = (float(loss), {(i, nm): eng.grad(i, nm).clone() for i in (3, 4, 5) for nm in ("gamma_noise", "beta_noise", "lmda")})
print("loss", res["fp32"][0], res["bf16"][0])
for k in res["fp32"][1]:
    a, b = res["fp32"][1][k].flatten(), res["bf16"][1][k].flatten()
    print(k, "rel", float((a - b).norm() / a.norm()), "cos", float(torch.dot(a, b) / (a.norm() * b.norm())))
