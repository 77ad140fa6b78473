"""ms_conv3x3_small_cout alone at the config-2 / config-4 shape (for tools/pmc_cmd.sh): python tools/one_small.py [c2|c4] [launches]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, Cin, Cout, H, W = (16, 16, 1, 256, 256) if cfg == "c2" else (16, 64, 3, 320, 320)
g = torch.Generator().manual_seed(3)
x = torch.randn(N, Cin, H, W, generator=g).to(dev); u = torch.randn(N, Cin, H, W, generator=g).to(dev)
wp = ops.pack_conv_weight_dgrad((torch.randn(Cin, Cout, 3, 3, generator=g) * 0.2).to(dev))
bc = torch.randn(Cin, 4, generator=g).to(dev).contiguous(); pa, pb, pc = ops.coef_ptrs(bc)
out = torch.empty(N, Cout, H, W, device=dev)
big = torch.empty(96 * 1024 * 1024, device=dev)            # 384 MB written between launches: the inputs come from HBM, not from the Infinity Cache
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
for i in range(reps):
    big.fill_(float(i))
    ev[2 * i].record()
    check(lib.ms_conv3x3_small_cout(x.data_ptr(), u.data_ptr(), out.data_ptr(), wp.data_ptr(), N, Cin, H, W, Cout, 2, pa, pb, pc, 4, st), "small_cout")
    ev[2 * i + 1].record()
torch.cuda.synchronize()
t = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) * 1e3 for i in range(1, reps))
mb = (2 * N * Cin * H * W + N * Cout * H * W) * 4 / 1e6
print(f"{cfg}: {mb:.1f} MB, median {t[len(t) // 2]:.1f} us (events, inputs evicted by a 384 MB fill between launches) = {mb / t[len(t) // 2]:.2f} TB/s")
