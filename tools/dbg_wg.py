import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from bench_kernels import timeit
from maxstyle_amd import ops
dev = torch.device("cuda:0")
for B in (1, 4, 16, 64):
    dy = torch.randn(B, 16, 256, 256, device=dev); x = torch.randn(B, 16, 256, 256, device=dev)
    t = timeit(lambda: ops.conv_wgrad(dy, x, 3), 30)
    print("B", B, "us", round(t * 1e6, 1))
