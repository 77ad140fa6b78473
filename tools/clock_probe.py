"""What clock64() and the fp32 MFMA pipe do under register-only MFMA load (ms_clock_probe), for different occupancies.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd._lib import lib, check
dev = torch.device("cuda:0")
probe = torch.zeros(2, dtype=torch.int64, device=dev); sink = torch.zeros(1, device=dev)
st = torch.cuda.current_stream().cuda_stream
iters = 2000
for wgs, thr in ((256, 256), (256, 512), (512, 512), (1024, 256), (1, 64), (1, 256)):
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        check(lib.ms_clock_probe(iters, wgs, thr, probe.data_ptr(), sink.data_ptr(), st), "ms_clock_probe")
        e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    cyc, ticks = int(probe[0]), int(probe[1])
    nm = iters * 64
    print(f"{wgs:5d} x {thr:3d} threads: kernel {us:8.1f} us = {wgs * (thr // 64) * nm * 2048 / (us * 1e-6) / 1e12:6.1f} TFLOP/s | workgroup 0: {ticks * 10 / 1e3:8.1f} us, "
          f"clock64 rate {cyc / (ticks * 10.0):.3f} GHz, {cyc / nm:.1f} clock64 cycles and {ticks * 10.0 / nm:.2f} ns per MFMA of one wave")
