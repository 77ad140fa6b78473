#!/usr/bin/env python3
"""bench.py - inner adversarial style-optimisation steps/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "C2"): FCN_16 dual-branch encoder/decoder, per-GPU batch 16 x 1 x 256 x 256,
MaxStyle inserted after decoder blocks [3,4,5] (all applied), Adam(lr=0.1) on {lmda, gamma_noise, beta_noise}; synthetic
ACDC-shaped data, procedurally initialised weights (no network access).  One *step* = one iteration i>=1 of
advanced_triplet_recon_segmentation_model.py:539-566: encode(recon) -> segmentation decoder -> -CE -> backward to the style
parameters -> Adam -> re-decode.  All inputs are resident in HBM before the timed region.  Pure data parallel: every rank runs
the same loop on its own batch, there is no collective on the path; `value` is the whole-job step rate.

Prints ONE JSON line on rank 0 (contract fields + "roofline" + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_STEP_C2 = 154.1e9       # conv FLOPs of one inner step at C2 (fwd 83.7 + data-grad 70.4), un-cached figure
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak (== fp32 vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-outer", action="store_true", help="skip the auxiliary whole-training-iteration figure")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5, help="inner steps of the CPU-oracle sample")
    ap.add_argument("--config", default="c2", choices=["c2", "c4"], help="c2: FCN_16 1x256x256 (the quoted metric); c4: FCN_64 3x320x320 (Prostate-shaped)")
    return ap.parse_args()


def build(dev, B, size, rank, net=(4, 1, 4)):
    from maxstyle_amd import engine as E
    from maxstyle_amd import synthetic as syn      # procedural weights / images / style states (the GPU leg never imports oracle/)
    spec_o = syn.NetSpec(*net)
    W = syn.procedural_weights(spec_o, 0)
    to = lambda sd: {k: v.to(dev) for k, v in sd.items()}
    spec = E.NetSpec(*net)
    nets = E.PackedNets(spec, to(W["image_encoder"]), to(W["segmentation_decoder"]), to(W["image_decoder"]))
    eng = E.InnerLoopEngine(spec, B, size, size, dev, lr=0.1)
    eng.set_nets(nets)
    img, lab = syn.synthetic_batch(B, size, net[1], net[2], seed=1234 + rank)
    layers = [3, 4, 5]
    styles = {i: syn.random_style_state(B, spec_o.channel_num[i], 7 + i) for i in layers}
    slots = {i: E.StyleSlot(i, B, spec_o.channel_num[i]) for i in layers}
    eng.configure_styles(layers, slots)
    for i in layers:
        st = styles[i]
        eng.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
    img_d, lab_d = img.to(dev), lab.to(dev)
    # z_i: one clean encoder pass (train-mode batch statistics), as the trainer hands it over (train_adv...py:192-193)
    z_i = eng.encode_fwd(img_d)[0].clone()
    return eng, W, img, lab, styles, z_i, lab_d


def timed_steps(eng, z_i, lab_d, steps, warmup, use_graph, dist_on):
    import torch.distributed as dist
    eng.code, eng.labels = z_i, lab_d
    eng._prefix_valid = False
    eng.step_dev.zero_()
    img = eng.decode(z_i)
    img = eng.step(img)                       # eager warm-up step: allocates every buffer
    graph = None
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eng.step(img)
            eng._graph = graph
        except Exception as ex:               # noqa: BLE001 - report and fall back to eager launches
            print(f"[bench] HIP graph capture failed, running eager: {ex!r}", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    run_one = (graph.replay if graph is not None else (lambda: eng.step(img)))
    for _ in range(max(warmup - 1, 0)):
        run_one()
    eng.step_dev.zero_()                      # loss slots restart (the Adam moments keep evolving: same work per step)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        n = min(steps - done, 60)             # loss_buf holds 64 slots
        for _ in range(n):
            run_one()
        done += n
        if done < steps:
            eng.step_dev.zero_()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, graph is not None


def kernel_rooflines(eng, z_i, lab_d, dev):
    """Per-launch duration of the two kernels DESIGN.md prices, measured live with HIP events on the launch stream.

    conv3x3 16->16 @256^2 (the dominant kernel: decoder up4 / encoder inc / their data-gradients) against the fp32 MFMA peak,
    and the fused MaxStyle forward (moments+restyle) at layer 4 (16x16x256x256) against the HBM peak."""
    from maxstyle_amd import ops
    from maxstyle_amd._lib import lib, check
    out = {}
    B, H, W = eng.B, eng.H, eng.W
    x = eng.buf["d.u4.xu"]                     # [B,16,256,256] live activation of the last step
    cw = eng.nets.dec["u4.c0"]
    y = torch.empty_like(x)
    stats, parts = ops.conv_stats_buffer(B, cw.cout, H, W, dev)

    def conv():
        ops.conv2d(x, cw.wp, cw.b, cw.cout, 3, 1, out=y, stats=stats)

    def style():
        eng.style_fwd(4, eng.buf["d.u4.out"])

    for name, fn in (("conv3x3_c16_256", conv), ("maxstyle_fwd_l4", style)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = []
        for _ in range(20):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        out[name] = ts[len(ts) // 2] * 1e-3    # median seconds per launch (group of launches for the style op)
    n_elem = x.numel()
    # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; tools/pmc_traffic.py)
    traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if (B, H, W) == (16, 256, 256):
            traffic = tj
    except Exception:  # noqa: BLE001
        traffic = {}
    conv_flops = 2.0 * B * H * W * cw.cout * cw.cin * 9
    conv_bytes = 2.0 * n_elem * 4
    return {
        "conv": {"bound": "mfma", "achieved": conv_flops / out["conv3x3_c16_256"] / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": conv_flops / out["conv3x3_c16_256"] / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": traffic.get("conv3x3_c16_256"),
                 "kernel": "conv_wide_kernel<NT=1,PRO=0> (3x3 s1, +BN statistics epilogue) 16->16 @%dx%dx%d" % (B, H, W), "us_per_launch": out["conv3x3_c16_256"] * 1e6,
                 "hbm_GBps": conv_bytes / out["conv3x3_c16_256"] / 1e9},
        "style": {"bound": "hbm", "achieved": 8.0 * n_elem / out["maxstyle_fwd_l4"] / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                  "frac": 8.0 * n_elem / out["maxstyle_fwd_l4"] / 1e9 / HBM_PEAK_GBPS, "traffic": traffic.get("maxstyle_fwd_l4"),
                  "kernel": "ms_style_fwd -> style_fused_kernel<16,1024> (single read) 16x16x%dx%d" % (H, W), "us_per_launch": out["maxstyle_fwd_l4"] * 1e6},
    }


def dice_parity(dev):
    """Dice of the segmentation of the stylised image: HIP loop vs CPU oracle from the same seeds (K=5, 4x1x64x64, layers [3,4,5])."""
    from oracle import maxstyle_oracle as orc
    from maxstyle_amd.metrics import runningScore
    B, size, layers, K = 4, 64, [3, 4, 5], 5
    eng, W, img, lab, styles, z_i, lab_d = build(dev, B, size, 0)
    out = eng.run(z_i, lab_d, K, use_graph=False).clone()
    eng.seg_loss(out, lab_d, need_grad=False, need_logits=True)
    rs = runningScore(4, dev)
    rs.update(lab_d, eng.buf["s.logits"])
    gpu_dice = rs.dice()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    st = {i: orc.StyleState(s.perm.clone(), s.lmda.clone(), s.gamma_noise.clone(), s.beta_noise.clone()) for i, s in styles.items()}
    ref = orc.generate_max_style_image(W, z_i.cpu(), st, layers, lab, n_iter=K, lr=0.1)
    with torch.no_grad():
        _, zs = orc.encoder_forward(W["image_encoder"], ref)
        pred = orc.decoder_forward(W["segmentation_decoder"], zs, "NN").argmax(1)
    cpu_dice = orc.dice_per_class(pred, lab, 4)
    return {"gpu": gpu_dice, "cpu_oracle": cpu_dice, "max_abs_diff": max(abs(a - b) for a, b in zip(gpu_dice, cpu_dice)),
            "image_rel_err": float((out.cpu() - ref).abs().max() / ref.abs().max()), "case": "K=5, 4x1x64x64, layers [3,4,5], free-running"}


def physical_cores():
    """Physical cores this process may run on (SMT siblings counted once)."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    cores = set()
    try:
        cpu = phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip() and cpu is not None:
                if cpu in allowed:
                    cores.add((phys, core))
                cpu = phys = core = None
    except Exception:  # noqa: BLE001
        pass
    return max(1, len(cores)) if cores else max(1, len(allowed))


def cpu_baseline(W, img, lab, styles, steps):
    """The CPU oracle (plain PyTorch restatement of the reference path, oracle/) timed on this box's host cores."""
    from oracle import maxstyle_oracle as orc
    ncores = physical_cores()
    # eager PyTorch-CPU convolutions at batch 16 do not scale to 128 threads: calibrate the thread count on one encoder pass and keep the
    # fastest (the baseline should be the CPU path at its best, not at its most oversubscribed)
    best_t, best_dt = ncores, None
    for t in sorted({8, 16, 32, 64, ncores}):
        if t > ncores:
            continue
        torch.set_num_threads(t)
        with torch.no_grad():
            orc.encoder_forward(W["image_encoder"], img)
            t0 = time.perf_counter(); orc.encoder_forward(W["image_encoder"], img); dt_ = time.perf_counter() - t0
        if best_dt is None or dt_ < best_dt:
            best_t, best_dt = t, dt_
    torch.set_num_threads(best_t)
    with torch.no_grad():
        z_i, _ = orc.encoder_forward(W["image_encoder"], img)
    st = {i: orc.StyleState(s.perm.clone(), s.lmda.clone(), s.gamma_noise.clone(), s.beta_noise.clone()) for i, s in styles.items()}
    layers = sorted(st)
    t0 = time.perf_counter()
    orc.generate_max_style_image(W, z_i, st, layers, lab, n_iter=1, lr=0.1)          # warm-up (MKLDNN primitive creation)
    t1 = time.perf_counter()
    orc.generate_max_style_image(W, z_i, st, layers, lab, n_iter=steps, lr=0.1)
    t2 = time.perf_counter()
    with torch.no_grad():
        orc.apply_max_style(W["image_decoder"], z_i, st, layers)
    t3 = time.perf_counter()
    # a call of n steps = n x (loss+backward+Adam+decode) + one extra decode
    per_step = (t2 - t1 - (t3 - t2)) / steps
    return {"value": 1.0 / per_step, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle.generate_max_style_image, same C2 workload (B=16,1x256x256,layers[3,4,5]), {steps} inner steps after a 1-step warm-up "
                      f"({t2 - t1:.1f}s timed, warm-up {t1 - t0:.1f}s), fp32, torch CPU, {torch.get_num_threads()} threads (fastest of 8/16/32/64/{ncores} on this box)"}


def outer_iteration(dev, batch, size, iters=6):
    """Auxiliary figure (not the headline metric): whole training iterations/s around the inner loop at the same configuration -
    standard pass -> MaxStyle inner loop K=5 -> hard-example pass -> backward (weight gradients) -> AdamW x3
    (train_adv_supervised_segmentation_triplet.py:163-199, 251-287, 532-535; SURVEY.md 8(f) rows 1,3)."""
    import maxstyle_amd as M
    from maxstyle_amd import synthetic as syn
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
    clean, lab = syn.synthetic_batch(batch, size, 1, 4, 1234)
    clean, lab = clean.to(dev), lab.to(dev)
    cfg = {"mix_style": True, "no_noise": False, "lr": 0.1, "n_iter": 5, "mix_learnable": True, "noise_learnable": True,
           "decoder_layers_indexes": [3, 4, 5], "loss_types": ["seg"], "loss_weights": [1], "always_use_beta": False}

    def iteration():
        S.train()
        S.reset_all_optimizers()
        image_l = torch.clamp(clean + 0.05 * torch.randn_like(clean), clean.min(), clean.max())
        seg0, rec0, gt0, sh0, recon0, p0, _ = S.standard_training(clean, lab, perturbed_image=image_l, return_output=True)
        S.reset_all_optimizers()
        sty = S.generate_max_style_image_from_config(S.z_i, cfg, clean, lab, p=1.5).detach().clone()      # p > 1: all three layers applied (worst case)
        seg1, rec1, sh1, sh2 = S.hard_example_traininng(perturbed_image=sty, perturbed_seg=None, clean_image_l=clean, label_l=lab)
        loss = (seg0 + rec0 + sh0 + gt0) + (rec1 + seg1 + sh1 + sh2)
        S.reset_all_optimizers()
        loss.backward()
        S.optimize_all_params()
        return loss

    first = float(iteration().detach())
    for _ in range(2):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"what": "training iterations/s (standard pass + K=5 inner loop + hard-example pass + backward + AdamW), same batch", "value": 1.0 / dt,
            "ms_per_iteration": dt * 1e3, "loss_first": first, "loss_last": float(loss.detach())}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)     # RCCL on ROCm; used for the barriers and the max-over-ranks only
    net = (4, 1, 4)
    if args.config == "c4":
        net = (1, 3, 2)
        if args.size == 256:
            args.size = 320
    eng, W, img, lab, styles, z_i, lab_d = build(dev, args.batch, args.size, rank, net)
    dt, graphed = timed_steps(eng, z_i, lab_d, args.steps, args.warmup, not args.no_graph, dist_on)
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_gpus = world
    value = n_gpus * args.steps / dt
    res = None
    if rank == 0:
        loss_last = float(eng.loss_buf[0])
        roof = kernel_rooflines(eng, z_i, lab_d, dev) if args.config == "c2" else {"conv": None, "style": None}
        res = {
            "metric": "inner adversarial style-opt steps/sec (batch 16, 256x256)", "value": value, "unit": "steps/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"C2: FCN_16 dual-branch, per-GPU batch {args.batch}x1x{args.size}x{args.size}, MaxStyle layers [3,4,5], Adam lr 0.1" if args.config == "c2"
                                    else f"C4: FCN_64 dual-branch, per-GPU batch {args.batch}x3x{args.size}x{args.size}, MaxStyle layers [3,4,5], Adam lr 0.1"),
                       "global_batch": args.batch * n_gpus, "parallelism": f"dp{n_gpus}", "hip_graph": graphed},
            "conv_tflops_step": FLOP_PER_STEP_C2 / (dt / args.steps) / 1e12 if (args.config, args.batch, args.size) == ("c2", 16, 256) else None,
            "roofline": roof["conv"], "roofline_maxstyle": roof["style"], "loss_check": loss_last,
        }
        if world == 1 and not args.no_cpu_baseline and args.config == "c2":
            res["cpu_baseline"] = cpu_baseline(W, img, lab, styles, args.cpu_steps)
            res["speedup_vs_cpu"] = value / res["cpu_baseline"]["value"]
            res["dice_parity"] = dice_parity(dev)
        if world == 1 and not args.no_outer and args.config == "c2":
            res["outer_iteration"] = outer_iteration(dev, args.batch, args.size)
        print(json.dumps(res), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
