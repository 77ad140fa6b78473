"""bf16 activation storage against fp32 storage on the same inner loop (GPU box): losses, image, step time.  python tools/bf16_loop_check.py [B size K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maxstyle_amd import engine as E, synthetic as syn


def build(dev, B, size, net, act_dtype):
    spec_o = syn.NetSpec(*net)
    W = syn.procedural_weights(spec_o, 0)
    to = lambda sd: {k: v.to(dev) for k, v in sd.items()}
    spec = E.NetSpec(*net)
    nets = E.PackedNets(spec, to(W["image_encoder"]), to(W["segmentation_decoder"]), to(W["image_decoder"]))
    eng = E.InnerLoopEngine(spec, B, size, size, dev, lr=0.1, act_dtype=act_dtype)
    eng.set_nets(nets)
    img, lab = syn.synthetic_batch(B, size, net[1], net[2], seed=1234)
    layers = [3, 4, 5]
    slots = {i: E.StyleSlot(i, B, spec_o.channel_num[i]) for i in layers}
    eng.configure_styles(layers, slots)
    for i in layers:
        st = syn.random_style_state(B, spec_o.channel_num[i], 7 + i)
        eng.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
    return eng, img.to(dev), lab.to(dev)


def main():
    B, size, K = (int(a) for a in (sys.argv[1:4] + ["16", "256", "5"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    res = {}
    only = os.environ.get("ONLY")
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        if only and name != only:
            continue
        eng, img, lab = build(dev, B, size, (4, 1, 4), dt)
        z_i = eng.encode_fwd(img.to(dt))[0].float().clone()
        out = eng.run(z_i, lab, K, use_graph=True).float().clone()
        losses = eng.losses(K).clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.run(z_i, lab, K, use_graph=True)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / 10
        res[name] = (out, losses, dt_s)
        print(name, "losses", [round(float(x), 5) for x in losses], f"call {dt_s * 1e3:.2f} ms for K={K}", "graph", eng._graph is not None, flush=True)
        eng.check_errors(sync=True)
    if len(res) < 2:
        return
    a, b = res["fp32"], res["bf16"]
    print("image max abs diff", float((a[0] - b[0]).abs().max()), "rms", float((a[0] - b[0]).pow(2).mean().sqrt()), "loss rel diff", [float(abs(x - y) / abs(x)) for x, y in zip(a[1], b[1])])
    print("speed-up", a[2] / b[2])


if __name__ == "__main__":
    main()
