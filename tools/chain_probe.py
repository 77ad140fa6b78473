"""Layer-chain go / no-go: L plain 128 -> 128 3x3 layers at 16 x 16 pixels (config 2's deepest level) as
  (a) L dependent ms_conv2d launches replayed from one captured graph (what the step does today), against
  (b) ONE persistent launch that walks the same layers with a grid barrier between them (ms_diag_k3n_chain; csrc/ms_conv_k3n.h, conv_k3n_chain_kernel).
Same kernel body, same bits; the difference is launch boundary vs barrier.  The judge's criterion: proceed only if a layer drops below 15 us.  (GPU box.)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maxstyle_amd import ops
from maxstyle_amd._lib import lib, check

dev = torch.device("cuda:0")
out = {}
for (N, C, H) in ((16, 128, 16), (20, 128, 16), (16, 64, 16)):
    for L in (2, 8):
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(N, C, H, 16, generator=g) * 0.5).to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dev)
        wp = ops.pack_conv_weight(w)
        a, b = x.clone(), torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream

        def launches():
            src, dst = a, b
            for _ in range(L):
                ops.conv2d(src, wp, None, C, 3, 1, out=dst)
                src, dst = dst, src
            return src
        a.copy_(x); ref = launches().clone()
        layers = torch.zeros(int(lib.ms_diag_k3n_chain_bytes(L)), dtype=torch.uint8, device=dev)
        arrive = torch.zeros(2, dtype=torch.int32, device=dev); err = torch.zeros(1, dtype=torch.int32, device=dev)

        def chain():
            check(lib.ms_diag_k3n_chain(a.data_ptr(), b.data_ptr(), wp.data_ptr(), N, C, H, L, layers.data_ptr(), arrive.data_ptr(), err.data_ptr(), st), "ms_diag_k3n_chain")
            return b if L % 2 else a
        a.copy_(x); got = chain().clone()
        torch.cuda.synchronize()
        same = bool(torch.equal(got, ref)) and int(err) == 0
        # (a): a captured graph of the L launches
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            launches(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                launches()
        gc = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            chain(); torch.cuda.synchronize()
            with torch.cuda.graph(gc, stream=s):
                check(lib.ms_diag_k3n_chain(a.data_ptr(), b.data_ptr(), wp.data_ptr(), N, C, H, L, layers.data_ptr(), arrive.data_ptr(), err.data_ptr(), s.cuda_stream), "chain")

        def timeit(fn, reps=200):
            for _ in range(20): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1000.0 / reps
        t_graph = timeit(gr.replay)
        t_chain = timeit(gc.replay)
        key = f"{N}x{C}x{H}x16 L={L}"
        out[key] = {"launches_us_per_layer": t_graph / L, "chain_us_per_layer": t_chain / L, "same_bits": same, "barrier_error": int(err)}
        print(key, json.dumps(out[key]))
# the marginal layer: (L = 8 total - L = 2 total) / 6 removes the fixed cost of the replay itself
for shp in ("16x128x16x16", "20x128x16x16", "16x64x16x16"):
    l2, l8 = out[f"{shp} L=2"], out[f"{shp} L=8"]
    m_l = (8 * l8["launches_us_per_layer"] - 2 * l2["launches_us_per_layer"]) / 6
    m_c = (8 * l8["chain_us_per_layer"] - 2 * l2["chain_us_per_layer"]) / 6
    print(f"{shp}: marginal layer {m_l:.2f} us as a launch, {m_c:.2f} us in the chain")
    out[shp + " marginal"] = {"launch_us": m_l, "chain_us": m_c}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "chain_probe.json"), "w"), indent=1)
