"""In-kernel cycle stamps of the wide conv kernel (needs a library built with -DMS_CONV_TRACE_BUILD)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
dev = torch.device("cuda:0")
trace = torch.zeros(16384, dtype=torch.int64, device=dev)
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
if t[256] != 0:
    print("producer, fine: p  wait-for-loads / prologue+LDS stores / U DMA issue / advance+set_tile / load issue + coefficient reads / DMA wait")
    for p in range(12):
        c = t[256 + p * 8:256 + p * 8 + 5]; c0 = t[128 + p * 4 + 1]
        print(p, c[1] - c[0], c0 - c[1], c[2] - c0, c[3] - c[2], t[128 + p * 4 + 2] - c[3], c[4] - t[128 + p * 4 + 2])
dc, dr = t[502] - t[500], t[503] - t[501]
if dr > 0:
    print(f"workgroup 0 main loop: {dc} shader cycles in {dr * 10} ns (s_memrealtime, 100 MHz) -> shader clock {dc / (dr * 10):.3f} GHz")

# per-workgroup wall-clock stamps (100 MHz): entry / first MFMA chunk / end of the item loop / exit, relative to the earliest entry
import numpy as np
w = np.array(t[1024:1024 + 4096], dtype=np.int64).reshape(-1, 4)
w = w[w[:, 0] != 0]
if len(w):
    t00 = w[:, 0].min()
    r = (w - t00) * 10 / 1000.0          # us
    q = lambda a: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print(f"{len(w)} workgroups, us since the first workgroup's entry:")
    print("  entry        ", q(r[:, 0]))
    print("  first MFMA   ", q(r[:, 1]))
    print("  loop end     ", q(r[:, 2]))
    print("  exit         ", q(r[:, 3]))
    print("  loop duration", q(r[:, 2] - r[:, 1]), "  entry -> first MFMA", q(r[:, 1] - r[:, 0]))
    if os.environ.get("MS_TRACE_DUMP"):
        np.save(os.environ["MS_TRACE_DUMP"], np.concatenate([r, np.array(t[1024 + 4096:1024 + 4096 + len(r)], dtype=np.float64).reshape(-1, 1)], axis=1))
    # by XCD (workgroup b runs on XCD b % 8) and by position inside the XCD: where do the slow workgroups sit?
    nb = len(r)
    for x in range(8):
        sel = r[np.arange(nb) % 8 == x]
        print(f"  XCD {x}: loop duration median {np.median(sel[:, 2] - sel[:, 1]):.1f}  min {np.min(sel[:, 2] - sel[:, 1]):.1f}  max {np.max(sel[:, 2] - sel[:, 1]):.1f}   exit max {sel[:, 3].max():.1f}")
    hw = np.array(t[1024 + 4096:1024 + 4096 + nb], dtype=np.int64)
    if hw.any():
        hw = hw & 0xFFFFFFFF; tg = (hw >> 16) & 15; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
        d = r[:, 2] - r[:, 1]
        for g in sorted(set(tg.tolist())):
            print(f"  thread-group slot {g}: {int((tg == g).sum())} workgroups, loop duration median {np.median(d[tg == g]):.1f}")
    pw = np.array(t[1024 + 4096 + 1024:1024 + 4096 + 1024 + 4 * nb], dtype=np.int64).reshape(-1, 4)
    if pw.any():
        e = (pw - (np.array(t[1024:1024 + 4 * nb], dtype=np.int64).reshape(-1, 4)[:, :1])) * 10 / 1000.0          # us since the workgroup's own entry
        print("  staging wave 0, us since its workgroup's entry (median over workgroups): set-up done %.2f | first loads issued %.2f | behind barrier #0 %.2f | first chunk stored %.2f | first MFMA %.2f"
              % (np.median(e[:, 0]), np.median(e[:, 1]), np.median(e[:, 2]), np.median(e[:, 3]), np.median(r[:, 1] - r[:, 0])))
