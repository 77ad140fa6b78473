"""In-kernel cycle stamps of the weight-gradient kernel (needs a library built with -DMS_WGRAD_TRACE_BUILD): who waits for whom, per tile."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
dev = torch.device("cuda:0")
trace = torch.zeros(256, dtype=torch.int64, device=dev)
from maxstyle_amd._lib import lib as _L
_L.ms_diag_set_trace(0, trace.data_ptr())
from maxstyle_amd import ops
B = 16
dy = torch.randn(B, 16, 256, 256, device=dev); x = torch.randn(B, 16, 256, 256, device=dev)
for _ in range(3):
    ops.conv_wgrad(dy, x, 3)
torch.cuda.synchronize()
t = trace.cpu().tolist()
c = [t[i * 3:(i + 1) * 3] for i in range(16)]
p = [t[64 + i * 4: 64 + (i + 1) * 4] for i in range(16)]
t0 = min(c[0][0], p[0][0])
print("consumer: tile  compute_start  compute_end  after_barrier   (cycles from start)")
for i in range(16):
    print(i, [v - t0 for v in c[i]], " compute", c[i][1] - c[i][0], " barrier wait", c[i][2] - c[i][1])
print("producer: tile  store_start store_end load_issued after_barrier")
for i in range(16):
    print(i, [v - t0 for v in p[i]], " store", p[i][1] - p[i][0], " set+load issue", p[i][2] - p[i][1], " barrier wait", p[i][3] - p[i][2])
