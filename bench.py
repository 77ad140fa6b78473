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

`python bench.py --gpus N` WITHOUT a launcher (no RANK in the environment) makes this process a parent that touches no GPU: it
starts N children of itself (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, RCCL rendezvous on 127.0.0.1) and waits for them;
under `torch.distributed.run` the ranks already exist and nothing is spawned.  A process that has initialised the GPU is never
re-exec'ed.  At N > 1 the auxiliary `outer_iteration` leg runs on every rank and exercises the ONE collective of the surrounding
training step (flat all-reduce of the outer gradients, train_adv_supervised_segmentation_triplet.py:532-535); its time is broken out.
`--dry-run` runs the same rank plumbing over gloo on the CPU with a stand-in step (tests/test_bench_launch.py): no number from it is a result.

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
FLOP_EXECUTED_C2 = 154.1e9 - 2 * 6.642e9   # what the engine launches: the style-independent decoder prefix (up1..up3 forward, SURVEY A.4: 3 x 2214 MMAC) is cached
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak (== fp32 vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--classes", type=int, default=4, help="segmentation classes of the c2 network (2 with --batch 20 --size 224 = the reference's shipped Prostate workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-outer", action="store_true", help="skip the auxiliary whole-training-iteration figure")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size parity legs against the reference fixture (drift_full_size / dice_parity)")
    ap.add_argument("--no-instep", action="store_true", help="skip the in-step timing of the priced launches (tools/prof_step.sh: their cut-off replays would pollute a kernel trace)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary blocks of the default line (winograd_off, c4, c5_bf16)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5, help="inner steps of the CPU-oracle sample")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="c2: FCN_16 1x256x256 (the quoted metric); c4: FCN_64 3x320x320 (Prostate-shaped); "
                         "c5: mixed ACDC+Prostate stream through the drop-in solver API with random-depth insertion (p=0.5), fp32 activation storage")
    ap.add_argument("--act-dtype", default="f32", choices=["f32", "bf16"],
                    help="storage type of the activation tensors of the conv stack (bf16: BASELINE config 5's 'bf16 activations'; statistics, parameters and the "
                         "matrix arithmetic stay fp32).  The headline metric is quoted on f32.")
    ap.add_argument("--mfma", default="f32", choices=["f32", "bf16"],
                    help="with --act-dtype bf16: matrix arithmetic of the 3x3 stride-1 convs (bf16: v_mfma_f32_16x16x16_bf16 on bf16-rounded operands, fp32 accumulation)")
    ap.add_argument("--stream-calls", type=int, default=8, help="c5: generate_max_style_image calls per pass of the stream (alternating ACDC / Prostate shaped)")
    ap.add_argument("--spinup-seconds", type=float, default=1.0, help="device spin-up (untimed replays of the step) in front of the W warm-up steps of the headline leg; 0 disables")
    ap.add_argument("--steady-seconds", type=float, default=2.0, help="length of the extra steady-state leg (graph replays, rank-local); 0 disables")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only: gloo on the CPU, stand-in step (CPU tests)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend of the real run (nccl = RCCL; gloo: validation of the N>1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than GPUs (rank r uses GPU r %% device_count): functional validation only, not a measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed path at ANY world size, also 1: the launcher parent starts the rank(s), the rank creates the RCCL process group with "
                         "device_id, barriers bracket the timed region, max-over-ranks and the flat outer-gradient all-reduce go through RCCL")
    ap.add_argument("--no-rccl-selftest", action="store_true", help="skip the world-size-1 RCCL self-test leg of a plain single-process run")
    return ap.parse_args()


def build(dev, B, size, rank, net=(4, 1, 4), act_dtype=None, mfma_bf16=False):
    from maxstyle_amd import engine as E
    from maxstyle_amd import synthetic as syn      # procedural weights / images / style states (the GPU leg never imports oracle/)
    spec_o = syn.NetSpec(*net)
    W = syn.procedural_weights(spec_o, 0)
    to = lambda sd: {k: v.to(dev) for k, v in sd.items()}
    spec = E.NetSpec(*net)
    nets = E.PackedNets(spec, to(W["image_encoder"]), to(W["segmentation_decoder"]), to(W["image_decoder"]))
    eng = E.InnerLoopEngine(spec, B, size, size, dev, lr=0.1, act_dtype=act_dtype, mfma_bf16=mfma_bf16)
    eng.set_nets(nets)
    img, lab = syn.synthetic_batch(B, size, net[1], net[2], seed=1234 + rank)
    layers = [3, 4, 5]
    styles = {i: syn.random_style_state(B, spec_o.channel_num[i], 7 + i) for i in layers}
    slots = {i: E.StyleSlot(i, B, spec_o.channel_num[i]) for i in layers}
    eng.configure_styles(layers, slots)
    for i in layers:
        st = styles[i]
        eng.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
    img_d, lab_d = img.to(dev).to(eng.act_dtype), lab.to(dev)
    # z_i: one clean encoder pass (train-mode batch statistics), as the trainer hands it over (train_adv...py:192-193)
    z_i = eng.encode_fwd(img_d)[0].clone()
    return eng, W, img, lab, styles, z_i, lab_d


def timed_steps(eng, z_i, lab_d, steps, warmup, use_graph, dist_on, spinup_s=0.0):
    """W untimed warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides.  spinup_s: before the warm-up steps the device is kept under the
    same load for that long (untimed replays): a fresh box's first ~100 ms of kernels run at whatever clock / power state the idle GPU was in - one driver-style run of
    round 5 measured 426.6 steps/s in its 50 timed steps and 455.2 in the 2 s steady-state leg right behind them (gpurun_out/r5_bench3.json)."""
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
    eng._bench_img = img                      # (the in-place image buffer of the step: in_step_times re-captures prefixes of the step on it)
    if spinup_s > 0:
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < spinup_s:
            for _ in range(40):
                run_one()
            eng.step_dev.zero_()
            torch.cuda.synchronize()
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
    eng.check_errors(sync=True)               # (outside the timed region) a timed-out bounded spin - single-read K1, `_xfin` - voids the run: raise, never report
    return dt, graph is not None, run_one


def _event_time(fn, reps=20, warm=3):
    """Seconds per call of fn(): `reps` back-to-back calls captured into ONE HIP graph on torch's current stream and replayed between two HIP events
    (a 25 us kernel launched call by call from Python is host-paced: the graph leaves only the dependent-launch boundaries between the launches).
    Median of 5 replays.  Falls back to per-call events when the capture fails."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / reps)
        ts.sort()
        return ts[len(ts) // 2] * 1e-3
    except Exception as ex:  # noqa: BLE001
        print(f"[bench] graph capture of a kernel-timing loop failed ({ex!r}); timing call by call", file=sys.stderr)
        torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e-3


def _step_budget():
    """tools/step_budget.py as a module (launch ledger of a step, per-launch bounds)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ms_step_budget", os.path.join(ROOT, "tools", "step_budget.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def in_step_times(eng, img, wanted, reps=15):
    """Duration of single launches INSIDE the replayed step, measured live (no profiler): the step is captured twice as a HIP graph, once cut off behind the launch
    and once in front of it (tools/step_budget.py::record_ledger(skip_after=i): the entry points behind the cut return without launching), both are replayed
    alternately between HIP events on the launch stream, and the launch's in-step time is the median difference.  It therefore contains what the launch pays in
    a step and not in isolation: inputs its predecessors left cold or still writing back, and its own dependent-launch boundary.
    wanted: {name: predicate(ledger entry)} - the first matching launch of the step.  -> ({name: seconds}, ledger, {name: index})"""
    sb = _step_budget()
    ledger, _ = sb.record_ledger(eng, img)
    torch.cuda.synchronize()
    idx = {}
    for name, pred in wanted.items():
        for i, e in enumerate(ledger):
            if pred(e):
                idx[name] = i
                break
    cuts = sorted({i for i in idx.values()} | {i - 1 for i in idx.values() if i > 0})
    graphs = {}
    for c in cuts:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sb.record_ledger(eng, img, skip_after=c)
        graphs[c] = g
        g.replay()
    torch.cuda.synchronize()

    def t_of(g):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record(); g.replay(); e_.record()
        torch.cuda.synchronize()
        return s_.elapsed_time(e_) * 1e-3
    out = {}
    for name, i in idx.items():
        diffs = []
        for _ in range(reps):
            a = t_of(graphs[i])
            b = t_of(graphs[i - 1]) if i > 0 else 0.0
            diffs.append(a - b)
        diffs.sort()
        out[name] = diffs[len(diffs) // 2]
    # leave the engine's buffers as a complete step leaves them
    eng.step(img)
    torch.cuda.synchronize()
    return out, ledger, idx


def step_roofline(ledger, step_s):
    """sum over the launches of one step of max(bytes / 8 TB/s, executed flop / 157.3 TFLOP/s) against the measured step time (accounting rules: tools/step_budget.py);
    the per-launch table with in-step durations from a kernel trace is committed as profiles/r05_step_budget_<config>.txt / .json."""
    sb = _step_budget()
    tot = hb = mf = 0.0
    for e in ledger:
        b, _ = sb.bound_us(e)
        tot += b
        if e["bytes"] / sb.HBM * 1e6 >= b:
            hb += b
        else:
            mf += b
    return {"launches": len(ledger), "sum_bound_us": tot, "hbm_bound_us": hb, "mfma_bound_us": mf, "step_us": step_s * 1e6, "frac": tot / (step_s * 1e6),
            "bound_rule": "per launch max(algorithmic bytes / 8.0 TB/s, EXECUTED flop / 157.3 TFLOP/s): Winograd launches 16/36 of the direct-form flop, sub-pixel forms 4/9 and 1/4",
            "algorithmic_GB_per_step": sum(e["bytes"] for e in ledger) / 1e9, "direct_form_GFLOP_per_step": sum(e["flop"] for e in ledger) / 1e9}


def conv_in_step(eng):
    """In-step durations of the three priced convolution launches of the engine's step (in_step_times) + the step's launch ledger; ({}, None) when it cannot be measured."""
    try:
        C, H = eng.nets.seg["u4.c3"].cout, eng.H
        top = lambda e: e["conv"] is not None and e["conv"]["ks"] == 3 and e["conv"]["stride"] == 1 and (e["conv"]["fetch"] & 0xFF) == 0 and \
            e["conv"]["Cin"] == C and e["conv"]["Cout"] == C and e["conv"]["Hs"] == H
        wanted = {"dgrad_actbwd": lambda e: top(e) and e["fn"].startswith("ms_conv2d_actbwd") and "xfin" not in e["fn"] and e["conv"]["pm"] == 2,
                  "dgrad_plain": lambda e: top(e) and not e["fn"].startswith("ms_conv2d_actbwd") and e["conv"]["pm"] == 2 and e["conv"]["epi"] == 0,
                  "conv_fwd": lambda e: top(e) and not e["fn"].startswith("ms_conv2d_actbwd") and e["conv"]["pm"] == 0 and e["conv"]["epi"] == 0}
        ts, ledger, idx = in_step_times(eng, eng._bench_img, wanted)
        for k, i in idx.items():
            print(f"[bench] in-step {k}: launch #{i} {ledger[i]['fn']}:{ledger[i]['key']} {ts[k] * 1e6:.1f} us", file=sys.stderr)
        return ts, ledger
    except Exception as ex:  # noqa: BLE001 - a measurement aid must not take the line down
        print(f"[bench] in-step timing unavailable: {ex!r}", file=sys.stderr)
        torch.cuda.synchronize()
        return {}, None


def _traffic_table(B, H, W):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE with the guide's gfx950 corrections,
    tools/pmc_traffic.py); the newest round's file wins."""
    if (B, H, W) != (16, 256, 256):
        return {}
    tab = {}
    for name in ("r01_traffic.json", "r02_traffic.json", "r03_traffic.json", "r04_traffic.json", "r05_traffic.json", "r06_traffic.json"):
        try:
            tab.update(json.load(open(os.path.join(ROOT, "profiles", name))))
            tab["_source"] = f"profiles/{name}: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel (tools/profile_round.sh), NOT collected in this run"
            # fresh = the kernel sources the passes priced are byte-identical to the tree's (sha256 recorded by tools/make_traffic.py); files of earlier rounds carry none
            import hashlib
            srcs = tab.get("_sources")
            tab["_fresh"] = bool(srcs) and all(os.path.exists(os.path.join(ROOT, f)) and hashlib.sha256(open(os.path.join(ROOT, f), "rb").read()).hexdigest() == h
                                               for f, h in srcs.items())
        except Exception:  # noqa: BLE001
            pass
    return tab


def kernel_rooflines(eng, dev, config, instep=None):
    """Per-launch duration of the kernels DESIGN.md prices, measured live with HIP events on the launch stream, on the live buffers of the
    last step.  `dominant` = the kernel with the largest per-step total in this round's rocprofv3 summary (profiles/r02_kernel_stats.txt):
    the data-gradient 3x3 conv with the two-tensor BatchNorm-backward prologue and the activation-backward epilogue at the top level
    (conv_wide_kernel<NT,2>); secondary blocks: the forward conv with the statistics epilogue and the MaxStyle K1/K2 kernels at layer 4."""
    from maxstyle_amd import ops
    from maxstyle_amd.engine import LEAKY
    b = eng.buf
    B, H, W = eng.B, eng.H, eng.W
    seg = eng.nets.seg
    C = seg["u4.c3"].cout
    x = b["d.u4.xu"]                           # [B,C,H,W] live activation of the last step
    cw = eng.nets.dec["u4.c0"]
    y = torch.empty_like(x)
    stats, parts = ops.conv_stats_buffer(B, cw.cout, H, W, dev)
    g2, u2, u1 = b["s.dh"], b["s.u4.u2"], b["s.u4.u1"]
    bc2, cf1 = b["s.u4.bw2.bcoef"], b["s.u4.bn1.coef"]
    c3, c0 = seg["u4.c3"], seg["u4.c0"]
    g2c = g2.clone()

    wino = ops.FETCH_WINOGRAD if getattr(eng, "winograd", False) else 0      # the form the engine's own launches take
    if wino and cw.wu:
        wino |= ops.FETCH_WINO_U                                              # ... with the transformed weights staged from the packed tensor's appendix

    def conv_fwd():
        ops.conv2d(x, cw.wp, cw.b, cw.cout, 3, 1, fetch=wino, out=y, stats=stats)

    def conv_dgrad_actbwd():                  # = the "s.u4.da1" launch of a step
        eng.conv_actbwd("bench.da1", "bench.bw1", g2c, c3, (bc2, u2), u1, cf1, LEAKY)

    def conv_dgrad_plain():                   # = the "s.u4.dhi" launch of a step
        eng.conv("bench.dhi", g2c, c0, bnbwd=(bc2, u2), dgrad=True)

    def style():
        eng.style_fwd(4, b["d.u4.out"])

    dy4 = torch.empty_like(b["d.u4.out"]).copy_(b["d.dh"]) if "d.dh" in b else torch.randn_like(b["d.u4.out"])

    def style_bwd():
        eng.style_bwd(4, dy4, need_dx=True)

    # bf16 activation storage of the MaxStyle layer (BASELINE config 5): the same layer-4 tensor, stored as bf16
    xb = b["d.u4.out"].to(torch.bfloat16)
    s4 = eng.styles[4]
    po = lambda nm: eng.flat_p[s4.off[nm][0]:s4.off[nm][0] + s4.off[nm][1]]
    gs16, bs16 = torch.empty(1, C, 1, 1, device=dev), torch.empty(1, C, 1, 1, device=dev)
    yb = torch.empty_like(xb)
    lm16, gn16, bn16 = po("lmda").view(B, 1, 1, 1), po("gamma_noise").view(B, C, 1, 1), po("beta_noise").view(B, C, 1, 1)
    ops.style_fwd(xb, s4.perm, lm16, gn16, bn16, gs16, bs16, True, out=yb)

    def style_bf16():
        ops.style_fwd(xb, s4.perm, lm16, gn16, bn16, gs16, bs16, False, out=yb)

    t = {name: _event_time(fn) for name, fn in (("conv_fwd", conv_fwd), ("dgrad_actbwd", conv_dgrad_actbwd), ("dgrad_plain", conv_dgrad_plain),
                                                ("style", style), ("style_bwd", style_bwd), ("style_bf16", style_bf16))}
    n_elem = x.numel()
    traffic = _traffic_table(B, H, W)
    flops = 2.0 * B * H * W * C * C * 9
    shape = "%d->%d @%dx%dx%d" % (C, C, B, H, W)

    instep = instep or {}
    from maxstyle_amd import _lib
    FORMS = {0: "conv_mfma_kernel (first generation, direct form)", 1: "conv_wide_kernel<NT,PRO,1,true,float> (wide, direct form)",
             2: "conv_wide_kernel<1,PRO,1,true,ms_f32w%s> (Winograd F(2x2,3x3), one 16-channel block per staged tile)" % ("32" if W < 64 else ""),
             3: "conv_wide_kernel<2,PRO,1,true,ms_f32w%s> (Winograd F(2x2,3x3), two 16-channel blocks per staged tile)" % ("32" if W < 64 else ""),
             4: "conv_wide_kernel<1,PRO,1,true,ms_f32wb> (Winograd F(2x2,3x3) on 8x8-pixel blocks, one 16-channel block)",
             5: "conv_wide_kernel<2,PRO,1,true,ms_f32wb> (Winograd F(2x2,3x3) on 8x8-pixel blocks, two 16-channel blocks)",
             6: "conv_k3n_kernel (narrow rows, direct form)",
             7: "conv_wide_kernel<2,PRO,1,true,ms_f32wf> (Winograd F(2x2,3x3) on the flattened tile list of 20-pixel images, two 16-channel blocks)"}

    def conv_block(key, what, pro, nbytes, tkey):
        # Both roofs are priced with what the launch EXECUTES: the Winograd form multiplies 16/36 of the direct form's products on the same fp32 matrix instruction, so its
        # matrix-pipe fraction is executed flop / peak (never above 1); the direct-form figure is a side field (`direct_form_equivalent_tflops`).  `bound` = the roof the
        # launch is closer to.  `frac` / `achieved` use the IN-STEP duration of the launch (in_step_times: measured live inside the replayed step) when the caller measured
        # it; the isolated back-to-back replay (warm caches, no neighbours) is `frac_isolated`.
        form = int(_lib.lib.ms_conv2d_form(B, C, H, W, C, pro, 0, wino))
        exf = flops * (16.0 / 36.0 if (form >= 2 and form != 6) else 1.0)
        t_iso = t[key]
        t_use = instep.get(key, t_iso)

        def fr(tt):
            return exf / tt / 1e12 / F32_MFMA_PEAK_TFLOPS, nbytes / tt / 1e9 / HBM_PEAK_GBPS
        mf, hf = fr(t_use)
        mfi, hfi = fr(t_iso)
        blk = {"bound": "mfma", "achieved": exf / t_use / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mf, "frac_isolated": mfi}
        if hf > mf:
            blk = {"bound": "hbm", "achieved": nbytes / t_use / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hf, "frac_isolated": hfi}
        blk.update({"traffic": traffic.get(tkey), "traffic_source": traffic.get("_source") if traffic.get(tkey) else None, "traffic_fresh": (traffic.get("_fresh") if traffic.get(tkey) else None),
                    "kernel": FORMS[form].replace("PRO", str(pro)) + (" [the hot combinations are instantiated with compile-time launch facts: template argument FX, ms_conv_wide.h] " if form >= 2 else " ") + what + " " + shape,
                    "us_per_launch": t_use * 1e6, "us_per_launch_source": ("in-step (two cut-off captures of the step, difference)" if key in instep else "isolated back-to-back replay"),
                    "us_per_launch_isolated": t_iso * 1e6, "algorithmic_bytes": nbytes, "hbm_GBps": nbytes / t_use / 1e9, "hbm_frac": hf, "hbm_frac_isolated": hfi,
                    "flop_per_launch_executed": exf, "executed_tflops": exf / t_use / 1e12, "executed_mfma_frac": mf, "executed_mfma_frac_isolated": mfi,
                    "form": "winograd F(2x2,3x3)" if (form >= 2 and form != 6) else "direct", "channel_blocks_per_tile": (2 if form == 7 else ((form - 2) % 2 + 1 if 2 <= form <= 5 else None))})
        if form >= 2 and form != 6:
            blk["direct_form_flop_per_launch"] = flops
            blk["direct_form_equivalent_tflops"] = flops / t_use / 1e12      # (work of the direct form per second: can pass the pipe's peak, NOT a roofline fraction)
        return blk

    def hbm_block(key, kernel, nbytes, tkey):
        t_use = instep.get(key, t[key])
        return {"bound": "hbm", "achieved": nbytes / t_use / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": nbytes / t_use / 1e9 / HBM_PEAK_GBPS,
                "frac_isolated": nbytes / t[key] / 1e9 / HBM_PEAK_GBPS,
                "traffic": traffic.get(tkey), "traffic_source": traffic.get("_source") if traffic.get(tkey) else None, "traffic_fresh": (traffic.get("_fresh") if traffic.get(tkey) else None), "kernel": kernel,
                "us_per_launch": t_use * 1e6, "us_per_launch_source": ("in-step (two cut-off captures of the step, difference)" if key in instep else "isolated back-to-back replay"),
                "us_per_launch_isolated": t[key] * 1e6, "algorithmic_bytes": nbytes}

    return {
        # reads g, u2 (prologue), u1 (mask), writes g1: 4 tensors
        "dominant": conv_block("dgrad_actbwd", "3x3 data-gradient, two-tensor BatchNorm-backward prologue, activation-backward epilogue", 2, 4.0 * n_elem * 4, "conv_dgrad_actbwd_c16_256"),
        "dgrad_plain": conv_block("dgrad_plain", "3x3 data-gradient, two-tensor prologue, plain epilogue", 2, 3.0 * n_elem * 4, "conv_dgrad_plain_c16_256"),
        "conv_fwd": conv_block("conv_fwd", "3x3 forward, +BN statistics epilogue", 0, 2.0 * n_elem * 4, "conv3x3_c16_256"),
        "style": hbm_block("style", "ms_style_fwd (K1: moments + restyle, single read) %dx%dx%dx%d" % (B, C, H, W), 8.0 * n_elem, "maxstyle_fwd_l4"),
        "style_bwd": hbm_block("style_bwd", "ms_style_bwd (K2: restyle backward with dx) %dx%dx%dx%d" % (B, C, H, W), 12.0 * n_elem, "maxstyle_bwd_l4"),
        "style_bf16": hbm_block("style_bf16", "ms_style_fwd_bf16 (K1 with bf16 activation storage, fp32 statistics; 4 B/element) %dx%dx%dx%d" % (B, C, H, W), 4.0 * n_elem, None),
    }


def steady_state(run_one, eng, seconds):
    """A seconds-long run of graph replays (the timed region of the contract is K steps = tens of ms): rules out clock / thermal drift."""
    if seconds <= 0:
        return None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(50):
            run_one()
        n += 50
        eng.step_dev.zero_()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"steps": n, "seconds": dt, "steps_s": n / dt}


def _load_trained():
    """Networks trained by the reference's own training step on the synthetic stream (tests/golden/make_golden_r2.py; fp16-rounded storage)."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "trained_fcn16.npz"))
    W = {"image_encoder": {}, "segmentation_decoder": {}, "image_decoder": {}}
    for key in z.files:
        net, name = key.split("/", 1)
        a = z[key]
        W[net][name] = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
    return W, np.load(os.path.join(ROOT, "tests", "golden", "loop_trained.npz"))


def dice_parity(dev):
    """Dice parity on TRAINED networks (a meaningful Dice: clean 0.65-0.80, stylised 0.39-0.55; random networks give ~0.07): K=5 free-running loop,
    4x1x64x64, layers [3,4,5]; HIP loop vs (a) the REFERENCE's own fp32 / fp64 runs (fixture tests/golden/loop_trained.npz) and (b) the CPU oracle live."""
    from oracle import maxstyle_oracle as orc
    from maxstyle_amd import engine as E, synthetic as syn
    from maxstyle_amd.metrics import runningScore
    W, g = _load_trained()
    B, size, layers, K = 4, 64, [3, 4, 5], 5
    spec = E.NetSpec(4, 1, 4)
    to = lambda sd: {k: v.to(dev) for k, v in sd.items()}
    eng = E.InnerLoopEngine(spec, B, size, size, dev, lr=0.1)
    eng.set_nets(E.PackedNets(spec, to(W["image_encoder"]), to(W["segmentation_decoder"]), to(W["image_decoder"])))
    img, lab = syn.synthetic_batch(B, size, 1, 4, seed=777)
    chn = syn.NetSpec(4, 1, 4).channel_num
    styles = {i: syn.random_style_state(B, chn[i], 7 + i) for i in layers}
    eng.configure_styles(layers, {i: E.StyleSlot(i, B, chn[i]) for i in layers})
    for i in layers:
        st = styles[i]
        eng.set_style_state(i, st.perm, st.lmda, st.gamma_noise, st.beta_noise)
    lab_d = lab.to(dev)
    z_i = eng.encode_fwd(img.to(dev))[0].clone()
    out = eng.run(z_i, lab_d, K, use_graph=False).clone()
    eng.seg_loss(out, lab_d, need_grad=False, need_logits=True)
    rs = runningScore(4, dev)
    rs.update(lab_d, logits=eng.buf["s.logits"])
    gpu_dice = rs.dice()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    st = {i: orc.StyleState(s.perm.clone(), s.lmda.clone(), s.gamma_noise.clone(), s.beta_noise.clone()) for i, s in styles.items()}
    ref = orc.generate_max_style_image(W, z_i.cpu(), st, layers, lab, n_iter=K, lr=0.1)
    with torch.no_grad():
        _, zs = orc.encoder_forward(W["image_encoder"], ref)
        pred = orc.decoder_forward(W["segmentation_decoder"], zs, "NN").argmax(1)
    cpu_dice = orc.dice_per_class(pred, lab, 4)
    o = out.cpu().double()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    ref64, ref32 = torch.from_numpy(g["f64.image"]), torch.from_numpy(g["f32.image"]).double()
    return {"case": "TRAINED FCN_16 (reference's own training step, tests/golden/trained_fcn16.npz), K=5 free-running, 4x1x64x64, layers [3,4,5]",
            "gpu": gpu_dice, "cpu_oracle": cpu_dice, "reference_fp32": [float(v) for v in g["f32.final_dice"]], "reference_clean": [float(v) for v in g["f32.clean_dice"]],
            "max_abs_diff_vs_reference": max(abs(a - float(b)) for a, b in zip(gpu_dice, g["f32.final_dice"])),
            "max_abs_diff_vs_oracle": max(abs(a - b) for a, b in zip(gpu_dice, cpu_dice)),
            "image_rel_err_vs_reference_fp64": rel(o, ref64), "image_rel_err_vs_oracle_fp32": rel(o, ref.double()),
            "reference_fp32_vs_fp64_image_rel": float(g["fp32_vs_fp64_image_rel"]), "oracle_fp32_vs_reference_fp64": rel(ref.double(), ref64),
            "pred_agreement_with_reference": float((eng.buf["s.logits"].argmax(1).cpu().numpy() == g["f32.final_pred"]).mean())}


def parity_full_size(dev):
    """Parity at the BENCHMARKED configuration against the reference itself: generate_max_style_image through the drop-in solver at
    16x1x256x256, layers [3,4,5], K=5 free-running, on the FCN_16 trained by the reference's own training step (fine-tuned at 256^2: clean Dice 0.92-0.95),
    against the REFERENCE's fp64 run of the same call (tests/golden/loop_full_c2.npz, made by tests/golden/make_golden_r3.py); `reference_noise_*` is the
    reference's own fp32 run against its fp64 run.  Both forms of the wide convolutions: Winograd F(2x2,3x3) (the timed default) and direct."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import r3_cases as R
    out = {"case": "C2 as benchmarked: trained FCN_16 (tests/golden/trained_fcn16_256.npz), 16x1x256x256, layers [3,4,5], K=5 free-running, drop-in API; "
                   "errors are against the reference's fp64 run, relative to max|image|"}
    from maxstyle_amd.options import engine_defaults
    if True:
        for form, flag in (("winograd", True), ("direct", False)):
            with engine_defaults(winograd=flag):
                r = R.full_size_case(dev)
            out[form] = {"image_max_err": r["image_max"], "image_rms_err": r["image_rms"],
                         "ratio_to_reference_noise_max": r["image_max"] / r["noise_image_max"], "ratio_to_reference_noise_rms": r["image_rms"] / r["noise_image_rms"],
                         "losses": r["losses"], "losses_rel_err": r["losses_rel"], "dice": r["dice"], "dice_clean": r["dice_clean"],
                         "dice_max_abs_diff_vs_reference": r["dice_abs_diff"], "labels_equal_to_reference": r["labels_equal_f64"]}
        out.update({"reference_noise_image_max": r["noise_image_max"], "reference_noise_image_rms": r["noise_image_rms"],
                    "reference_noise_losses_rel": r["noise_losses_rel"], "reference_noise_labels_equal": r["noise_labels_equal"],
                    "reference_dice_fp64": r["dice_ref_f64"], "reference_dice_fp32": r["dice_ref_f32"], "reference_dice_clean": r["dice_clean_ref"]})
    return out


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


def cpu_baseline(W, img, lab, styles, steps, z_gpu=None):
    """The CPU oracle (plain PyTorch restatement of the reference path, oracle/) timed on this box's host cores.  Returns (record, final image, losses)
    - the K-step result from the code `z_gpu` doubles as the full-size drift reference."""
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
    if z_gpu is not None:
        z_i = z_gpu
    else:
        with torch.no_grad():
            z_i, _ = orc.encoder_forward(W["image_encoder"], img)
    mk = lambda: {i: orc.StyleState(s.perm.clone(), s.lmda.clone(), s.gamma_noise.clone(), s.beta_noise.clone()) for i, s in styles.items()}
    st = mk()
    layers = sorted(st)
    t0 = time.perf_counter()
    orc.generate_max_style_image(W, z_i, st, layers, lab, n_iter=1, lr=0.1)          # warm-up (MKLDNN primitive creation)
    t1 = time.perf_counter()
    st = mk()                                                                        # the timed run starts from the initial state again
    tr = orc.InnerLoopTrace()
    final = orc.generate_max_style_image(W, z_i, st, layers, lab, n_iter=steps, lr=0.1, trace=tr)
    t2 = time.perf_counter()
    with torch.no_grad():
        orc.apply_max_style(W["image_decoder"], z_i, st, layers)
    t3 = time.perf_counter()
    # a call of n steps = n x (loss+backward+Adam+decode) + one extra decode
    per_step = (t2 - t1 - (t3 - t2)) / steps
    rec = {"value": 1.0 / per_step, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"oracle.generate_max_style_image, same C2 workload (B=16,1x256x256,layers[3,4,5]), {steps} inner steps after a 1-step warm-up "
                     f"({t2 - t1:.1f}s timed, warm-up {t1 - t0:.1f}s), fp32, torch CPU, {torch.get_num_threads()} threads (fastest of 8/16/32/64/{ncores} on this box)",
           "losses": [float(v) for v in tr.losses]}
    return rec, final


def outer_iteration(dev, batch, size, rank=0, world=1, iters=6, dist_on=None):
    """Auxiliary figure (not the headline metric): whole training iterations/s around the inner loop at the same configuration -
    standard pass -> MaxStyle inner loop K=5 -> hard-example pass -> backward (weight gradients) -> [N > 1: ONE flat RCCL all-reduce of the
    outer gradients] -> AdamW x3 (train_adv_supervised_segmentation_triplet.py:163-199, 251-287, 532-535; SURVEY.md 8(e), 8(f) rows 1,3).
    Every rank calls this (the all-reduce is a collective); rank r trains on its own batch (seed 1234 + r) from rank 0's weights."""
    import maxstyle_amd as M
    from maxstyle_amd import synthetic as syn
    import torch.distributed as dist
    S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True, optimizer_type="AdamW")
    # a trainer has a natural flush point - optimize_all_params() resolves the loop's error check before any weight moves - so it takes the deferred protocol
    # (no event wait inside generate_max_style_image: the host queues the hard-example pass while the GPU still runs the inner loop)
    S.loop_error_check = "deferred"
    dist_on = (world > 1) if dist_on is None else dist_on           # --force-dist: the collective path also at world size 1
    if dist_on:
        from maxstyle_amd import distributed as D
        D.broadcast_parameters(list(S.model.values()), src=0)
    clean, lab = syn.synthetic_batch(batch, size, 1, 4, 1234 + rank)
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
        S.optimize_all_params()             # all-reduces the flat gradient buffer first when torch.distributed is initialised
        return loss

    first = float(iteration().detach())
    for _ in range(2):
        iteration()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = iteration()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = (time.perf_counter() - t0) / iters
    out = {"what": "training iterations/s (standard pass + K=5 inner loop + hard-example pass + backward + flat all-reduce at N>1 + AdamW), per-GPU batch "
                   + str(batch), "ms_per_iteration": dt * 1e3, "loss_first": first, "loss_last": float(loss.detach())}
    if dist_on:
        from maxstyle_amd import distributed as D
        dt = D.max_over_ranks(dt, dev)
        bank = S._param_bank()
        flat = bank.flat_g
        for _ in range(3):
            bank.all_reduce_grads(force=True)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            bank.all_reduce_grads(force=True)
        torch.cuda.synchronize()
        ar = D.max_over_ranks((time.perf_counter() - t0) / 20, dev)
        # rank-equal weights after the exchanged steps: max |w_r - w_0| over ranks
        w0 = bank.flat_p.clone()
        dist.broadcast(w0, 0)
        dev_max = torch.tensor([float((bank.flat_p - w0).abs().max())], device=dev)
        dist.all_reduce(dev_max, op=dist.ReduceOp.MAX)
        out.update({"ms_per_iteration": dt * 1e3, "allreduce_ms": ar * 1e3, "allreduce_bytes": flat.numel() * 4,
                    "allreduce_algbw_GBps": flat.numel() * 4 / ar / 1e9, "world_seen": dist.get_world_size(),
                    "weights_max_abs_diff_across_ranks": float(dev_max.item())})
    out["value"] = world / dt
    out["per_gpu"] = 1.0 / dt
    # the roofline of the training passes around the inner loop (round 6): priced per launch family by tools/train_budget.py from a kernel trace of eager
    # iterations - a committed profile, NOT collected in this run (the live figure of this leg is `ms_per_iteration`)
    for name in ("r06_step_budget_train.json",):
        try:
            sm = json.load(open(os.path.join(ROOT, "profiles", name)))["summary"]
            tp = sm["training_passes"]
            out["step_roofline"] = {"source": f"profiles/{name} (tools/train_budget.py: rocprofv3 --kernel-trace of eager trainer iterations; committed, not collected in this run)",
                                    "training_passes_sum_bound_us": tp["sum_bound_us"], "training_passes_kernel_us": tp["lib_kernel_us"] + tp["torch_kernel_us"],
                                    "training_passes_wall_us": tp["wall_us"], "frac": tp["sum_bound_us"] / tp["wall_us"], "frac_of_kernel_time": tp["frac_of_kernel_time"],
                                    "inner_loop_wall_us": sm["inner_loop_wall_us"],
                                    "phases": {k: {kk: v.get(kk) for kk in ("sum_bound_us", "lib_us", "wall_us", "lib_launches")} for k, v in sm["phases"].items()}}
        except Exception:  # noqa: BLE001
            pass
    return out


def rccl_leg(dev, rank, run_one=None):
    """The collective path of the surrounding training step on the LIVE process group, at any world size (1 included): barrier, max-over-ranks, and the
    ONE exchange of an outer iteration - the in-place flat all-reduce (sum, then x 1/world) of the outer gradients (train_adv...py:532-535) -
    timed at the two sizes the path has: FCN_16's 1.54 M parameters (6.1 MB) and an FCN_64-sized buffer (24.5 M, 98 MB).  With `run_one` (the captured
    inner step) it also replays the step graph between collectives: graph replay and the process group coexist on the device."""
    import torch.distributed as dist
    from maxstyle_amd import distributed as D
    world = dist.get_world_size()
    out = {"backend": dist.get_backend(), "world_seen": world, "device_id_bound": True}
    dist.barrier()
    out["max_over_ranks_ok"] = D.max_over_ranks(float(rank + 1), dev) == float(world)
    for tag, n in (("fcn16_6MB", 1536325), ("fcn64_98MB", 24500000)):
        flat = torch.full((n,), float(rank + 1), device=dev)
        dist.all_reduce(flat); flat.mul_(1.0 / world)
        ok = bool(torch.all(flat == (world + 1) / 2.0))
        for _ in range(3):
            dist.all_reduce(flat)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        dt = D.max_over_ranks((time.perf_counter() - t0) / 20, dev)
        out[tag] = {"bytes": n * 4, "mean_ok": ok, "ms": dt * 1e3, "algbw_GBps": n * 4 / dt / 1e9}
        if run_one is not None and tag == "fcn16_6MB":
            flat.fill_(float(rank + 1))
            for _ in range(5):                              # the inner step's captured graph replayed between collectives on the same device
                run_one()
                dist.all_reduce(flat)
                flat.mul_(1.0 / world)
            torch.cuda.synchronize()
            out[tag]["ok_with_step_graph_replay_between"] = bool(torch.all(flat == (world + 1) / 2.0)) if world == 1 else bool(torch.isfinite(flat).all())
        del flat
    if world == 1:
        out["note"] = "world size 1: the collectives degenerate (nothing crosses a link); ms = latency of the RCCL call path, algbw is not a bandwidth"
    dist.barrier()
    return out


def rccl_selftest(dev, run_one=None):
    """A plain `python bench.py` (one process, no launcher) still executes the RCCL path: a world-size-1 `nccl` process group bound to the device
    (`device_id`), created AFTER the timed region so it cannot touch the headline, torn down afterwards.  Never fatal: a failure is reported in the line."""
    import torch.distributed as dist
    try:
        if dist.is_initialized():
            return {"skipped": "a process group is already live"}
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
        try:
            return rccl_leg(dev, 0, run_one)
        finally:
            dist.destroy_process_group()
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)[:400]}


def mixed_stream(dev, args, rank, world, dist_on):
    """BASELINE config 5 on this rank's share of the node: a stream of generate_max_style_image calls (the reference's own entry point,
    advanced_triplet...py:458-571) alternating ACDC-shaped (FCN_16, 16x1x256x256, K=5) and Prostate-shaped (FCN_64, 16x3x320x320, K=10) batches,
    every call drawing its own random subset of MaxStyle layers with the trainer's p=0.5 (train_adv...py:263).  One captured HIP graph per
    (shape, layer subset) signature is kept by the solver; the first pass over the stream captures them, the timed passes replay.  Activation storage is
    fp32 unless --act-dtype bf16 [--mfma bf16] ("bf16 conv stack"); a call whose layers all draw "not applied" runs 0 steps, as in the reference."""
    import maxstyle_amd
    from maxstyle_amd import synthetic as syn
    cfgs = []
    for tag, ntype, net, size, K in (("acdc", "FCN_16_standard_no_STN", (4, 1, 4), 256, 5), ("prostate", "FCN_64_standard_no_STN", (1, 3, 2), 320, 10)):
        spec = syn.NetSpec(*net)
        S = maxstyle_amd.AdvancedTripletReconSegmentationModel(network_type=ntype, image_ch=net[1], num_classes=net[2], use_gpu=True)
        if args.act_dtype == "bf16":
            S.loop_act_dtype = torch.bfloat16
            S.loop_mfma_bf16 = args.mfma == "bf16"
        Wt = syn.procedural_weights(spec, 0)
        for name, mod in S.model.items():
            mod.load_state_dict(Wt[name]); mod.train()
        img, lab = syn.synthetic_batch(args.batch, size, net[1], net[2], seed=1234 + rank)
        img, lab = img.to(dev), lab.to(dev)
        z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
        cfgs.append((tag, S, spec, img, lab, z_i.detach(), K))

    def one_pass(seed0):
        steps, subsets = 0, []
        for c in range(args.stream_calls):
            tag, S, spec, img, lab, z_i, K = cfgs[c % 2]
            S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, p=0.5, n_iter=K, lr=0.1, reference_image=img, reference_segmentation=lab,
                                       fix_seed=seed0 + c)
            applied = [int(k) for k, m in S.last_style_modules.items() if len(list(m.parameters())) > 0]
            subsets.append((tag, applied))
            steps += K if applied else 0
        return steps, subsets

    for _ in range(max(args.warmup, 1)):
        one_pass(100)                                   # same seeds as the timed passes: every signature's graph is captured here
    torch.cuda.synchronize()
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0
    passes = max(args.steps // 10, 1)
    for _ in range(passes):
        n, subsets = one_pass(100)
        steps += n
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        from maxstyle_amd import distributed as D
        dt = D.max_over_ranks(dt, dev)
    if rank != 0:
        return None
    return {"metric": "inner adversarial style-opt steps/sec (mixed ACDC 16x1x256x256 K=5 + Prostate 16x3x320x320 K=10 stream, random depth p=0.5)",
            "value": world * steps / dt, "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": dt / max(steps, 1) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (("bf16 matrix arithmetic (fp32 accumulation), bf16 activation storage" if args.mfma == "bf16" else "f32 arithmetic, bf16 activation storage")
                      if args.act_dtype == "bf16" else "f32"), "data": "synthetic",
            "config": {"workload": f"C5 (this rank's share): {args.stream_calls} generate_max_style_image calls per pass, alternating FCN_16 16x1x256x256 K=5 / FCN_64 16x3x320x320 K=10, "
                                   "MaxStyle layers drawn per call from [3,4,5] with p=0.5, " + ("bf16" if args.act_dtype == "bf16" else "fp32") + " activation storage", "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "hip_graph": True, "passes": passes, "calls_per_pass": args.stream_calls},
            "calls": [{"shape": t, "layers_applied": a} for t, a in subsets], "seconds": dt,
            "note": "whole-call rate through the drop-in solver API: includes MaxStyle construction, the initial and final decodes and the host side of every call"}


def whole_call(dev, args, rank):
    """the rate of the reference's own entry point, `generate_max_style_image` through the drop-in solver API at the C2 workload - whole-call
    steps/s K / t_call (MaxStyle construction, the clean decode, K steps, the final decode, the host side) and K / (t_call - t_decode_only), where
    t_decode_only is the same call with n_iter = 0 (everything but the K steps)."""
    import maxstyle_amd
    from maxstyle_amd import synthetic as syn
    net, K = (4, 1, 4), 5
    spec = syn.NetSpec(*net)
    S = maxstyle_amd.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=net[1], num_classes=net[2], use_gpu=True)
    Wt = syn.procedural_weights(spec, 0)
    for name, mod in S.model.items():
        mod.load_state_dict(Wt[name]); mod.train()
    img, lab = syn.synthetic_batch(args.batch, args.size, net[1], net[2], seed=1234 + rank)
    img, lab = img.to(dev), lab.to(dev)
    z_i, _ = S.encode_image(img, disable_track_bn_stats=True)
    z_i = z_i.detach()

    def call(n_iter):
        return S.generate_max_style_image(z_i, [3, 4, 5], spec.channel_num, p=1.5, n_iter=n_iter, lr=0.1, reference_image=img, reference_segmentation=lab, fix_seed=7)

    def timed(n_iter, reps):
        for _ in range(3):
            call(n_iter)
        best = None
        for _ in range(3):                    # three groups of `reps` calls, the fastest group: a 1-2 ms host-side call is sensitive to whatever else the box's cores do
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                call(n_iter)
            torch.cuda.synchronize()
            dt_ = (time.perf_counter() - t0) / reps
            best = dt_ if best is None else min(best, dt_)
        return best
    t_call, t_dec = timed(K, 20), timed(0, 20)
    # the same call with the SYNCHRONOUS error protocol (solver.loop_error_check = "sync"): the single-read kernel's error word is resolved by an event wait inside the
    # call that produced the image instead of by the next call / optimize_all_params (the default since round 5), so the host cannot prepare the next call meanwhile
    S.flush_loop_errors()
    S.loop_error_check = "sync"
    t_call_s, t_dec_s = timed(K, 20), timed(0, 20)
    S.loop_error_check = None
    return {"what": "generate_max_style_image through the drop-in API, C2 workload, K=5, every layer applied (p forced); from its second issue the whole call (decode + K steps) "
                    "replays as ONE captured graph; default error protocol = deferred (a spin time-out of the single-read kernel raises from the next call / the weight step at the latest)",
            "ms_per_call": t_call * 1e3, "ms_decode_only_call": t_dec * 1e3, "whole_call_steps_s": K / t_call,
            "steps_s_excluding_decode": K / max(t_call - t_dec, 1e-9),
            "sync_error_check": {"ms_per_call": t_call_s * 1e3, "ms_decode_only_call": t_dec_s * 1e3, "whole_call_steps_s": K / t_call_s,
                                 "steps_s_excluding_decode": K / max(t_call_s - t_dec_s, 1e-9)}}


def shipped_blocks(dev, rank):
    """The reference's SHIPPED workloads (VERDICT r4 missing 2): config/ACDC/1500_epoch/MICCAI2022_MaxStyle.json (crop 192x192, batch 20, 4 classes) and
    config/Prostate/MICCAI2022_MaxStyle.json (224x224, batch 20, 2 classes), FCN_16, layers [3,4,5]: steps/s, the step roofline, and WHICH kernel form every convolution
    launch of the step took (the library's own answer) - their 12- / 14- / 24- / 28-pixel levels are shapes the headline configuration never launches.
    Parity at these shapes: tests/test_round5_gpu.py::test_shipped_workload_vs_reference_run."""
    sb = _step_budget()
    out = {}
    for tag, net, size in (("acdc_192", (4, 1, 4), 192), ("prostate_224", (4, 1, 2), 224)):
        eng, _, _, _, _, z_i, lab_d = build(dev, 20, size, rank, net)
        dt, graphed, _ = timed_steps(eng, z_i, lab_d, 20, 3, True, False)
        blk = {"workload": f"FCN_16 dual-branch, batch 20x1x{size}x{size}, {net[2]} classes, MaxStyle layers [3,4,5], Adam lr 0.1, fp32", "steps_s": 20 / dt,
               "ms_per_step": dt / 20 * 1e3, "hip_graph": graphed}
        try:
            ledger, _ = sb.record_ledger(eng, eng._bench_img)
            torch.cuda.synchronize()
            blk["step_roofline"] = step_roofline(ledger, dt / 20)
            forms = {}
            names = {0: "first_generation", 1: "wide_direct", 2: "winograd_1block", 3: "winograd_2blocks", 4: "winograd_8x8_1block", 5: "winograd_8x8_2blocks",
                     6: "narrow_rows_second_generation"}
            for e in ledger:
                cv = e["conv"]
                if cv is None:
                    continue
                fn = e["fn"]
                if fn.startswith("ms_conv3x3_small_cin"):
                    kind = "k3_taps_as_k_first_conv"
                elif cv.get("vector_alu"):
                    kind = "vector_alu"
                elif fn.startswith("ms_conv_subpix"):
                    kind = "subpixel"
                elif cv["ks"] == 1:
                    kind = "k1_streaming" if lib_k1s(cv) else "k1_tiled_or_gemm"
                elif cv["ks"] == 3 and cv["stride"] == 1 and (cv["fetch"] & 0xFF) == 0:
                    kind = "k3_" + names.get(sb.conv_form(cv), "?")
                elif cv["ks"] == 3 and cv["stride"] == 2:
                    kind = "k3s2"
                elif (cv["fetch"] & 0xFF) != 0:
                    kind = "k3_fused_resample_first_generation"
                else:
                    kind = "k%ds%d" % (cv["ks"], cv["stride"])
                rows = "%dpx" % cv["Ws"]
                forms.setdefault(kind, {}).setdefault(rows, 0)
                forms[kind][rows] += 1
            blk["conv_forms_by_row_width"] = forms
            blk["launches"] = len(ledger)
        except Exception as ex:                              # noqa: BLE001 - a measurement aid must not take the line down
            blk["step_roofline_error"] = repr(ex)[:200]
        eng.check_errors()
        out[tag] = blk
        del eng
        torch.cuda.empty_cache()
    return out


def lib_k1s(cv):
    from maxstyle_amd import _lib
    return int(_lib.lib.ms_conv_k1s_would_run(cv["N"], cv["Cin"], cv["Hs"], cv["Ws"], cv["Cout"], cv["epi"] if cv["epi"] in (0, 2, 4, 5) else 0)) == 1


def secondary_blocks(dev, args, rank):
    """Driver-visible figures for what the headline line does not cover, a few hundred ms of GPU time each:
    `winograd_off` - the C2 workload with the direct form of the wide convolutions; `c4` - BASELINE config 4 (FCN_64, 16x3x320x320) with its dominant
    kernel priced on direct-form AND executed multiplications; `c5_bf16` - config 5's mixed stream (random depth, both shapes) with bf16 activation storage."""
    import copy
    out = {}
    from maxstyle_amd.options import engine_defaults
    with engine_defaults(winograd=False):
        eng, _, _, _, _, z_i, lab_d = build(dev, args.batch, args.size, rank)
        dt, graphed, _ = timed_steps(eng, z_i, lab_d, 20, 3, True, False)
        out["winograd_off"] = {"what": "the headline workload (C2) with the DIRECT form of the wide 3x3 convolutions (EngineOptions.winograd = False)", "steps_s": 20 / dt,
                               "ms_per_step": dt / 20 * 1e3, "hip_graph": graphed}
        del eng
    torch.cuda.empty_cache()
    eng, _, _, _, _, z_i, lab_d = build(dev, args.batch, 320, rank, (1, 3, 2))
    dt, graphed, _ = timed_steps(eng, z_i, lab_d, 10, 2, True, False)
    ins, ledger = conv_in_step(eng)
    roof = kernel_rooflines(eng, dev, "c4", ins)
    out["c4"] = {"workload": f"C4: FCN_64 dual-branch, batch {args.batch}x3x320x320, MaxStyle layers [3,4,5], Adam lr 0.1, fp32", "steps_s": 10 / dt, "ms_per_step": dt / 10 * 1e3,
                 "hip_graph": graphed, "roofline": roof["dominant"], "roofline_conv_fwd": roof["conv_fwd"], "roofline_maxstyle": roof["style"], "roofline_maxstyle_bwd": roof["style_bwd"],
                 "step_roofline": (step_roofline(ledger, dt / 10) if ledger else None), "step_roofline_per_launch": "profiles/r05_step_budget_c4.txt"}
    del eng
    torch.cuda.empty_cache()
    out["shipped"] = shipped_blocks(dev, rank)
    a5 = copy.copy(args)
    a5.act_dtype, a5.mfma, a5.steps, a5.warmup, a5.stream_calls = "bf16", "f32", 10, 1, 8
    r5 = mixed_stream(dev, a5, rank, 1, False)
    out["c5_bf16"] = {"workload": r5["config"]["workload"], "steps_s": r5["value"], "seconds": r5["seconds"], "calls": r5["calls"], "dtype": r5["dtype"], "note": r5["note"]}
    torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_children(args):
    """Parent of `python bench.py --gpus N` when no launcher set RANK: starts N ranks of this file (one per GPU) and waits.
    This process makes no GPU call of any kind - the GPUs are counted from the KFD topology in sysfs, not through torch.cuda / HIP
    (maxstyle_amd.distributed.visible_gpu_count) - and never exec's."""
    import subprocess
    from maxstyle_amd.distributed import visible_gpu_count
    n = args.gpus
    if not args.dry_run and not args.oversubscribe:
        have = visible_gpu_count()
        if 0 <= have < n:
            print(f"[bench] --gpus {n} but only {have} GPU(s) are visible (KFD topology, *_VISIBLE_DEVICES applied)", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = rc or code
                print(f"[bench] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()       # exact PIDs of our own children
        time.sleep(0.05)
    return rc


def dry_run(args, rank, world):
    """Rank plumbing without a GPU: gloo rendezvous, barrier-bracketed timed region, max over ranks, flat all-reduce of an FCN_16-sized buffer."""
    import torch.distributed as dist
    from maxstyle_amd import distributed as D
    dev = torch.device("cpu")
    multi = world > 1 or args.force_dist
    if multi:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    a = torch.randn(64, 64)
    step = lambda: (a @ a).sum().item()
    for _ in range(args.warmup):
        step()
    if multi:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    ar = None
    if multi:
        dt = D.max_over_ranks(dt, dev)
        flat = torch.full((1536325,), float(rank + 1))
        dist.all_reduce(flat); flat.mul_(1.0 / world)
        ar = {"world_seen": dist.get_world_size(), "mean_ok": bool(torch.allclose(flat, torch.full_like(flat, (world + 1) / 2.0)))}
    if rank == 0:
        emit({"metric": "inner adversarial style-opt steps/sec (batch 16, 256x256)", "value": world * args.steps / dt, "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "DRY RUN (CPU stand-in step, gloo): rank plumbing only, not a measurement", "global_batch": args.batch * world,
                                     "parallelism": f"dp{world}"}, "outer_iteration": ar})
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


_JSON_FD = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout, but libraries write to fd 1 themselves (RCCL prints a version banner when its first communicator is
    created).  Keep a private duplicate of stdout for the JSON line and point fd 1 at stderr for everything else."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    args = parse()
    if "RANK" not in os.environ and (args.gpus > 1 or args.force_dist):
        return launch_children(args)
    _claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: reporting n_gpus={world}", file=sys.stderr)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        os.environ.setdefault("MASTER_PORT", "29513")
        return dry_run(args, rank, world)
    dist_on = world > 1 or args.force_dist
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    ndev = torch.cuda.device_count()
    shared = args.oversubscribe or int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > ndev
    if shared:
        # several ranks on one GPU: the co-residency-dependent single-read MaxStyle kernel must not be selected; the engines read this switch
        os.environ["MS_SHARED_DEVICE"] = "1"
    if args.oversubscribe:
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29514")
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL on ROCm: barriers, max-over-ranks, the outer-gradient all-reduce
        else:
            dist.init_process_group("gloo")
    if args.config == "c5":
        res = mixed_stream(dev, args, rank, world, dist_on)
        if rank == 0:
            emit(res)
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return 0
    net = (4, 1, args.classes)
    if args.config == "c4":
        net = (1, 3, 2)
        if args.size == 256:
            args.size = 320
    bf16 = args.act_dtype == "bf16"
    eng, W, img, lab, styles, z_i, lab_d = build(dev, args.batch, args.size, rank, net, torch.bfloat16 if bf16 else None, bf16 and args.mfma == "bf16")
    dt, graphed, run_one = timed_steps(eng, z_i, lab_d, args.steps, args.warmup, not args.no_graph, dist_on, spinup_s=args.spinup_seconds)
    coll = {}
    if dist_on:
        from maxstyle_amd import distributed as D
        dt = D.max_over_ranks(dt, dev)
        # N > 1: every leg that contains a collective runs HERE, on all ranks, before rank 0 starts its rank-local extras (steady-state leg, in-step timing, kernel
        # rooflines: ~20-30 s) - otherwise ranks 1..N-1 would sit inside RCCL kernels waiting for rank 0 all that time
        coll["rccl"] = rccl_leg(dev, rank, run_one) if args.backend == "nccl" else None
        if not args.no_outer and args.config == "c2" and args.act_dtype != "bf16":
            coll["outer_iteration"] = outer_iteration(dev, args.batch, args.size, rank, world, dist_on=True)
    n_gpus = world
    value = n_gpus * args.steps / dt
    headline = (args.config, args.batch, args.size) == ("c2", 16, 256)
    res = None
    if rank == 0:
        loss_last = float(eng.loss_buf[0])
        steady = steady_state(run_one, eng, args.steady_seconds)
        none6 = {k: None for k in ("dominant", "dgrad_plain", "conv_fwd", "style", "style_bwd", "style_bf16")}
        ins, ledger = ({}, None) if (bf16 or args.no_instep) else conv_in_step(eng)
        roof = none6 if bf16 else kernel_rooflines(eng, dev, args.config, ins)     # (the priced kernels and their algorithmic bytes are the fp32-storage ones)
        step_s = dt / args.steps
        res = {
            "metric": ("inner adversarial style-opt steps/sec (batch 16, 256x256)" if headline else
                       f"inner adversarial style-opt steps/sec ({args.config.upper()}: batch {args.batch}, {net[1]}x{args.size}x{args.size}" + (", bf16 activation storage" if bf16 else "") + ")"),
            "value": value, "unit": "steps/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": (("bf16 matrix arithmetic (fp32 accumulation), bf16 activation storage" if args.mfma == "bf16" else "f32 arithmetic, bf16 activation storage") if bf16 else "f32"), "data": "synthetic",
            "config": {"workload": (f"C2: FCN_16 dual-branch, per-GPU batch {args.batch}x1x{args.size}x{args.size}, MaxStyle layers [3,4,5], Adam lr 0.1" if args.config == "c2"
                                    else f"C4: FCN_64 dual-branch, per-GPU batch {args.batch}x3x{args.size}x{args.size}, MaxStyle layers [3,4,5], Adam lr 0.1"),
                       "global_batch": args.batch * n_gpus, "parallelism": f"dp{n_gpus}", "hip_graph": graphed, "world_seen": world,
                       **({"oversubscribed": True, "backend": args.backend} if (args.oversubscribe or args.backend != "nccl") else {})},
            "per_gpu_steps_s": value / n_gpus,
            "device_spinup_s": args.spinup_seconds,
            "steady_state": steady,
            "conv_flops": ({"gflop_per_step_uncached": FLOP_PER_STEP_C2 / 1e9, "gflop_per_step_executed": FLOP_EXECUTED_C2 / 1e9,
                            "tflops_uncached_accounting": FLOP_PER_STEP_C2 / step_s / 1e12, "tflops_executed": FLOP_EXECUTED_C2 / step_s / 1e12,
                            "note": "executed = launched by the engine (decoder prefix up1..up3 cached per call); uncached = SURVEY 8(d) figure, comparable with the reference"}
                           if headline else None),
            "roofline": roof["dominant"], "roofline_dgrad_plain": roof["dgrad_plain"], "roofline_conv_fwd": roof["conv_fwd"],
            "roofline_maxstyle": roof["style"], "roofline_maxstyle_bwd": roof["style_bwd"], "roofline_maxstyle_bf16": roof["style_bf16"], "loss_check": loss_last,
            "step_roofline": (step_roofline(ledger, step_s) if ledger else None),
            "step_roofline_per_launch": f"profiles/r05_step_budget_{args.config}.txt (every launch of the step: bytes, executed flop, bound, in-step duration from a rocprofv3 kernel trace)",
        }
        if world == 1 and not args.no_cpu_baseline and args.config == "c2" and not bf16:
            res["cpu_baseline"], _ = cpu_baseline(W, img, lab, styles, args.cpu_steps, z_gpu=z_i.cpu())
            res["speedup_vs_cpu"] = value / res["cpu_baseline"]["value"]
        if world == 1 and headline and not bf16 and not args.no_parity:
            pf = parity_full_size(dev)
            res["drift_full_size"] = {"case": pf["case"], "reference_noise_image_max": pf["reference_noise_image_max"], "reference_noise_image_rms": pf["reference_noise_image_rms"],
                                      **{form: {k: pf[form][k] for k in ("image_max_err", "image_rms_err", "ratio_to_reference_noise_max", "ratio_to_reference_noise_rms", "losses_rel_err")}
                                         for form in ("winograd", "direct")}}
            res["dice_parity"] = {"case": pf["case"], "reference_fp64": pf["reference_dice_fp64"], "reference_fp32": pf["reference_dice_fp32"], "reference_clean": pf["reference_dice_clean"],
                                  **{form: {k: pf[form][k] for k in ("dice", "dice_clean", "dice_max_abs_diff_vs_reference", "labels_equal_to_reference")} for form in ("winograd", "direct")}}
            res["dice_parity_small"] = dice_parity(dev)
        if world == 1 and args.config == "c2" and not bf16 and not args.no_outer:
            res["whole_call"] = whole_call(dev, args, rank)
    if rank == 0 and world == 1 and headline and not bf16 and not args.no_secondary:
        res["secondary"] = secondary_blocks(dev, args, rank)
    if dist_on:
        if rank == 0:
            res.update(coll)
    elif rank == 0 and not args.no_rccl_selftest:
        res["rccl"] = rccl_selftest(dev, run_one)
    if not dist_on and not args.no_outer and args.config == "c2" and not bf16:
        del eng, run_one
        torch.cuda.empty_cache()
        oi = outer_iteration(dev, args.batch, args.size, rank, world, dist_on=False)
        if rank == 0:
            # side field: the same iteration with the training passes' forward / data-gradient convs in the DIRECT form (the default until round 4; since round 5 they take the
            # Winograd form: 20-seed weight-gradient fidelity equal, profiles/r05_train_fidelity.json)
            from maxstyle_amd.options import engine_defaults
            with engine_defaults(train_winograd=False):
                torch.cuda.empty_cache()
                ow = outer_iteration(dev, args.batch, args.size, rank, world, dist_on=False)
                oi["direct_form_training_passes"] = {"switch": "EngineOptions.train_winograd = False", "ms_per_iteration": ow["ms_per_iteration"], "value": ow["value"]}
        if rank == 0:
            res["outer_iteration"] = oi
    if rank == 0:
        emit(res)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
