"""In-kernel cycle stamps of the wide conv kernel (needs a library built with -DMS_CONV_TRACE_BUILD)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
dev = torch.device("cuda:0")
trace = torch.zeros(512, dtype=torch.int64, device=dev)
from maxstyle_amd._lib import lib as _L
_L.ms_diag_set_trace(trace.data_ptr(), 0)
from maxstyle_amd import ops
# usage: trace_conv.py [plain|pro1|bwd|actbwd] [C] [size] [fetch bits: 0x100 = Winograd, 0x500 = Winograd one-block, +0x800 = weights from the appendix (0x900 / 0xD00)]
N = 16
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
FETCH = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
x = torch.randn(N, C, S, S, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.1; b = torch.randn(C, device=dev)
wp = ops.pack_conv_weight(w)
if FETCH & 0x800:                                  # MS_FETCH_WINO_U: the transformed weights staged from the packed tensor's appendix
    wp, _ = ops.with_wino_appendix(wp, C, C)
out = torch.empty_like(x)
stats, parts = ops.conv_stats_buffer(N, C, S, S, dev)
kw = dict(stats=stats)
if mode == "pro1":
    bc = torch.randn(C, 4, device=dev); pa, pb, pc = ops.coef_ptrs(bc)
    kw = dict(stats=stats, pro_mode=1, pro_a=pa, pro_b=pb, pro_cstride=4, slope=0.2)
if mode == "bwd":
    bc = torch.randn(C, 4, device=dev); pa, pb, pc = ops.coef_ptrs(bc)
    kw = dict(pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=torch.randn_like(x))
if mode == "actbwd":                                # the dominant launch of the C2 step: two-tensor prologue + activation-backward epilogue (ms_conv2d_actbwd)
    bc = torch.randn(C, 4, device=dev); pa, pb, pc = ops.coef_ptrs(bc)
    u = torch.randn_like(x); c4 = torch.randn(C, 4, device=dev); x2 = torch.randn_like(x)
    for _ in range(3):
        ops.conv2d_actbwd(x, wp, C, 3, u, c4, 0.2, pro_mode=2, pro_a=pa, pro_b=pb, pro_c=pc, pro_cstride=4, in2=x2, fetch=FETCH)
else:
    for _ in range(3):
        ops.conv2d(x, wp, b, C, 3, 1, out=out, fetch=FETCH, **kw)
torch.cuda.synchronize()
t = trace.cpu().tolist()
t0 = min(t[0], t[128])
print("consumer: p  [compute_start, compute_end, epilogue_end, after_barrier]   compute / epilogue / barrier wait")
for p in range(12):
    c = t[p * 4:(p + 1) * 4]
    print(p, [v - t0 for v in c], c[1] - c[0], c[2] - c[1], c[3] - c[2])
print("producer: p  [store_start, store_end, loads_issued, after_barrier]   store / set+load / barrier wait")
for p in range(12):
    c = t[128 + p * 4:128 + (p + 1) * 4]
    print(p, [v - t0 for v in c], c[1] - c[0], c[2] - c[1], c[3] - c[2])
dc, dr = t[502] - t[500], t[503] - t[501]
if dr > 0:
    print(f"workgroup 0 main loop: {dc} shader cycles in {dr * 10} ns (s_memrealtime, 100 MHz) -> shader clock {dc / (dr * 10):.3f} GHz")
