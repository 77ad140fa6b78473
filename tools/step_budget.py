"""Per-launch roofline budget of ONE captured inner step (round 4; VERDICT r3 "next" item 3).

    python tools/step_budget.py record <c2|c4> <ledger.json>                 (GPU box)  the launch ledger of one eager step: entry point, engine key,
                                                                                        shapes, algorithmic bytes, direct-form flop
    python tools/step_budget.py merge <ledger.json> <rocprof dir> <out stem>  (anywhere) + the in-step duration of every launch from a rocprofv3 kernel trace
                                                                                        of the SAME step replayed as a graph -> <out stem>.txt / .json

Accounting rules ("step roofline"):
  bytes   every distinct activation-sized tensor (>= 64 KB, found by its address among the engine's buffers) the launch is handed, once; the output twice when
          the launch accumulates into it (conv epi_mode 1); tables, coefficients, weights and the small per-channel buffers are not counted.
  flop    convolutions only: 2*N*Ho*Wo*Cout*Cin*ks^2 in the direct form; EXECUTED flop = what the kernel that ran multiplies - 16/36 of that for the Winograd
          form (kernel name ...ms_f32w / ms_bf16w...), 4/9 for the sub-pixel up-sampling conv, 1/4 of the zero-inserted form for the sub-pixel stride-2
          data-gradient; everything else 0 (their arithmetic is far below their byte time).
  bound   max(bytes / 8.0 TB/s, executed flop / 157.3 TFLOP/s)   (MI355X_MICROARCH.md: HBM3E spec, dense fp32 MFMA peak).
  actual  median over the replayed steps of the trace of the launch at that position of the step.
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM = 8.0e12
MFMA = 157.3e12

# (index of N in the argument list, has in2, index of `out`, indices of further activation tensors) of the conv-family entry points
CONV_SIG = {
    "ms_conv2d": 5, "ms_conv2d_xfin": 5, "ms_conv2d_ride": 5, "ms_conv2d_actbwd": 4,
}
NO_LAUNCH = ("_bytes", "_parts", "_ok", "_eligible", "_capacity", "_slots", "_offset", "_threads", "ms_num_cus", "ms_version", "ms_last_error", "ms_set_option", "ms_get_option",
             "ms_option_default", "ms_option_count", "ms_option_name", "ms_diag_set_trace", "ms_conv2d_form", "ms_conv_k1s_would_run")


def conv_cost(fn, a):
    """-> dict(N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm, epi, flop) or None"""
    base = fn.replace("_bf16m", "").replace("_bf16", "")
    if base in CONV_SIG:
        i = CONV_SIG[base]
        N, Cin, Hs, Ws, Cout, ks, stride, fetch, pm = a[i:i + 9]
        epi = 3 if "actbwd" in base else (a[15] if base == "ms_conv2d_xfin" else (a[20] if base in ("ms_conv2d", "ms_conv2d_ride") else 0))
        up = 2 if (fetch & 0xFF) else 1
        Ho, Wo = (Hs * up + (2 if ks == 3 else 0) - ks) // stride + 1, (Ws * up + (2 if ks == 3 else 0) - ks) // stride + 1
        cols = 4 * Cout if epi == 2 else Cout
        return dict(N=N, Cin=Cin, Hs=Hs, Ws=Ws, Cout=Cout, ks=ks, stride=stride, fetch=fetch, pm=pm, epi=epi, flop=2.0 * N * Ho * Wo * cols * Cin * ks * ks)
    if base == "ms_conv2d_actbwd_xfin":      # (in, in2, out, w, N, Cin, Hs, Ws, Cout, ks, stride, fetch, ...): always the two-tensor prologue and the activation-backward epilogue
        N, Cin, Hs, Ws, Cout, ks, stride, fetch = a[4:12]
        return dict(N=N, Cin=Cin, Hs=Hs, Ws=Ws, Cout=Cout, ks=ks, stride=stride, fetch=fetch, pm=2, epi=3, flop=2.0 * N * Hs * Ws * Cout * Cin * ks * ks)
    if base in ("ms_conv1x1_bnres", "ms_conv1x1_bnres_xfin"):
        N, Cin, Hs, Ws, Cout = a[4:9]
        return dict(N=N, Cin=Cin, Hs=Hs, Ws=Ws, Cout=Cout, ks=1, stride=1, fetch=0, pm=0, epi=4, flop=2.0 * N * Hs * Ws * Cout * Cin)
    if base in ("ms_conv_subpix", "ms_conv_subpix2"):
        N, Cin, Hs, Ws, Cout, mode = a[4:10] if base == "ms_conv_subpix" else a[5:11]      # (ms_conv_subpix2: w_sums sits behind w_packed)
        # direct-form flop of the conv it replaces: 3x3 on the up-sampled / zero-inserted [2Hs, 2Ws] grid
        return dict(N=N, Cin=Cin, Hs=Hs, Ws=Ws, Cout=Cout, ks=3, stride=1, fetch=1 + mode, pm=0, epi=0, flop=2.0 * N * 4 * Hs * Ws * Cout * Cin * 9)
    if base == "ms_conv3x3_small_cout":
        N, Cin, H, W, Cout = a[4:9]
        return dict(N=N, Cin=Cin, Hs=H, Ws=W, Cout=Cout, ks=3, stride=1, fetch=0, pm=a[9], epi=0, flop=2.0 * N * H * W * Cout * Cin * 9, vector_alu=True)
    if base == "ms_conv3x3_small_cin":
        N, Cin, H, W, Cout = a[4:9]
        return dict(N=N, Cin=Cin, Hs=H, Ws=W, Cout=Cout, ks=3, stride=1, fetch=0, pm=0, epi=0, flop=2.0 * N * H * W * Cout * Cin * 9, vector_alu=True)
    return None


def record_ledger(eng, im, skip_after=None, skip_only=None):
    """One eager eng.step(im) with every library launch recorded -> list of dict(fn, key, conv, tensors, bytes, flop).
    skip_after = i: launches with index > i are NOT issued (their entry points return 0) - bench.py's in-step timing captures such prefixes of a step."""
    import torch
    from maxstyle_amd import _lib
    import maxstyle_amd.engine as E
    import maxstyle_amd.ops as O
    lib = _lib.lib
    calls = []

    class Proxy:
        def __getattr__(self, n):
            f = getattr(lib, n)
            if not n.startswith("ms_") or any(n.endswith(s) or n == s for s in NO_LAUNCH):
                return f

            def rec(*a):
                calls.append([n, a, None])
                if skip_after is not None and len(calls) - 1 > skip_after:
                    return 0
                return f(*a)
            return rec
    chk = E.check

    def check(rc, what=""):
        if calls and calls[-1][2] is None:
            calls[-1][2] = what
        return chk(rc, what)
    E.lib = Proxy(); O.lib = Proxy(); E.check = check
    try:
        out = eng.step(im)
    finally:
        E.lib = lib; O.lib = lib; E.check = chk
    bufs = {}
    for name, t in eng.buf.items():
        if torch.is_tensor(t) and t.numel() * t.element_size() >= 65536:
            bufs[t.data_ptr()] = (name, t.numel() * t.element_size())
    for extra_name in ("code", "labels"):
        t = getattr(eng, extra_name, None)
        if torch.is_tensor(t) and t.numel() * t.element_size() >= 65536:
            bufs.setdefault(t.data_ptr(), (extra_name, t.numel() * t.element_size()))
    bufs.setdefault(im.data_ptr(), ("image", im.numel() * im.element_size()))
    ledger = []
    for fn, a, what in calls:
        seen, tens = set(), []
        for v in a:
            if isinstance(v, int) and v in bufs and v not in seen:
                seen.add(v); tens.append(list(bufs[v]))
        cv = conv_cost(fn, a)
        nbytes = sum(t[1] for t in tens)
        if cv is not None and cv["epi"] == 1:       # accumulate epilogue: the output is read and written
            outp = a[2]
            if outp in bufs:
                nbytes += bufs[outp][1]
        key = what.split(":", 1)[-1] if (what and ":" in what) else ""
        ledger.append(dict(fn=fn, key=key, conv=cv, tensors=tens, bytes=nbytes, flop=(cv["flop"] if cv else 0.0)))
    return ledger, out


def conv_form(cv, bf16=0):
    """What ms_conv2d would launch for this call (the library's own answer: ms_conv2d_form) -> 0 first-generation kernel, 1 wide direct, 2 Winograd one block, 3 two blocks."""
    from maxstyle_amd import _lib
    if cv is None or cv["ks"] != 3 or cv["stride"] != 1 or (cv["fetch"] & 0xFF) != 0 or cv.get("vector_alu"):
        return 0
    return int(_lib.lib.ms_conv2d_form(cv["N"], cv["Cin"], cv["Hs"], cv["Ws"], cv["Cout"], cv["pm"] if cv["pm"] < 3 else 2, bf16, cv["fetch"] & 0xF00))


def bound_us(entry, kernel=None):
    """bound of one ledger entry; without a kernel name (no trace at hand: bench.py's live block) the executed flop follow the library's own dispatch answer."""
    cv = entry["conv"]
    if kernel is not None:
        fr = executed_fraction(kernel, cv)
    elif cv is None or cv.get("vector_alu"):
        fr = 0.0
    elif entry["fn"].startswith("ms_conv_subpix"):
        fr = 4.0 / 9.0 if cv["fetch"] == 1 else 0.25
    else:
        fr = 16.0 / 36.0 if conv_form(cv) >= 2 else 1.0
    ex = fr * entry["flop"]
    return max(entry["bytes"] / HBM, ex / MFMA) * 1e6, ex


def record(cfg, out):
    import torch
    import bench
    dev = torch.device("cuda:0")
    # c2 / c4: BASELINE configs 2 and 4; acdc192 / prostate224: the reference's shipped workloads (config/ACDC/1500_epoch/MICCAI2022_MaxStyle.json, config/Prostate/MICCAI2022_MaxStyle.json)
    net, size, B = {"c2": ((4, 1, 4), 256, 16), "c4": ((1, 3, 2), 320, 16), "acdc192": ((4, 1, 4), 192, 20), "prostate224": ((4, 1, 2), 224, 20)}[cfg]
    eng, W, img, lab, styles, z_i, lab_d = bench.build(dev, B, size, 0, net)
    eng.code, eng.labels = z_i, lab_d
    eng._prefix_valid = False
    im = eng.decode(z_i)
    eng.step(im)                                   # allocate everything
    torch.cuda.synchronize()
    ledger, _ = record_ledger(eng, im)
    torch.cuda.synchronize()
    json.dump(dict(config=cfg, batch=B, size=size, launches=len(ledger), ledger=ledger), open(out, "w"), indent=0)
    print(f"{len(ledger)} library calls recorded -> {out}")


def executed_fraction(kernel, cv):
    if cv is None:
        return 0.0
    if cv.get("vector_alu"):
        return 0.0                                  # (vector-ALU kernels: priced by their bytes)
    if "conv_subpix_kernel<0" in kernel or "conv_subpix2_kernel<0" in kernel:
        return 4.0 / 9.0
    if "conv_subpix_kernel<1" in kernel or "conv_subpix2_kernel<1" in kernel:
        return 0.25
    if "conv_wide_kernel" in kernel and any(t in kernel for t in ("ms_f32w", "ms_bf16w")):
        return 16.0 / 36.0
    return 1.0


def step_slices(rows):
    """rows of a kernel trace, time-ordered -> list of [start, end) index ranges of whole replayed steps (a step ends with step_tail / incr)."""
    ends = [i for i, r in enumerate(rows) if "step_tail_kernel" in r["Kernel_Name"] or "incr_kernel" in r["Kernel_Name"]]
    return [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]


def short(name):
    name = name.replace("void ms::", "").replace("ms::", "")
    return name[:name.index("(")] if "(" in name else name


# kernels that are the SECOND (third) launch of one library call: merged into the launch in front of them when a traced step has more launches than the ledger has calls
AUX = ("ce_finalize_kernel", "style_bwd_finalize_kernel", "style_finalize_kernel", "restyle_kernel<", "wgrad_reduce")


def align(names, durs, n):
    extra = len(names) - n
    on, od = [], []
    for nm, d in zip(names, durs):
        if extra > 0 and on and any(a in nm for a in AUX):
            on[-1] += " + " + nm; od[-1] += d; extra -= 1
        else:
            on.append(nm); od.append(d)
    return (on, od) if len(on) == n else (None, None)


def merge(ledger_path, trace_dir, stem):
    L = json.load(open(ledger_path))
    kt = glob.glob(os.path.join(trace_dir, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(kt)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sl = step_slices(rows)
    # a traced "step" runs from behind one step tail to the next: the re-decode that follows the tail inside eng.step() comes FIRST there - rotate the ledger to match
    tail = max(i for i, e in enumerate(L["ledger"]) if e["fn"] in ("ms_step_tail", "ms_counter_incr"))
    L["ledger"] = L["ledger"][tail + 1:] + L["ledger"][:tail + 1]
    n = len(L["ledger"])
    good, aligned = [], []
    for a, b in sl:
        if not (n <= b - a <= n + 6):
            continue
        nm = [short(rows[i]["Kernel_Name"]) for i in range(a, b)]
        du = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in range(a, b)]
        on, od = align(nm, du, n)
        if on is not None:
            good.append((a, b)); aligned.append((on, od))
    if not good:
        from collections import Counter
        raise SystemExit(f"no replayed step of {n} library calls in the trace (step lengths seen: {Counter(b - a for a, b in sl).most_common(5)})")
    good, aligned = good[-min(len(good), 12):], aligned[-min(len(good), 12):]
    names = aligned[-1][0]
    durs = []
    for i in range(n):
        d = sorted(od[i] for on, od in aligned)
        durs.append(d[len(d) // 2])
    walls = sorted((int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in good)
    out = []
    for i, e in enumerate(L["ledger"]):
        ex = executed_fraction(names[i], e["conv"]) * e["flop"]
        t_b, t_f = e["bytes"] / HBM * 1e6, ex / MFMA * 1e6
        out.append(dict(i=i, fn=e["fn"], key=e["key"], kernel=names[i], shape=(None if e["conv"] is None else [e["conv"][k] for k in ("N", "Cin", "Hs", "Ws", "Cout", "ks", "stride")]),
                        bytes=e["bytes"], flop_direct=e["flop"], flop_executed=ex, bound_us=max(t_b, t_f), bound=("hbm" if t_b >= t_f else "mfma"), actual_us=durs[i]))
    # alignment sanity: a conv-family call must sit on a conv kernel and vice versa
    bad = [o["i"] for o in out if (o["shape"] is not None) != ("conv" in o["kernel"].split(" + ")[0])]
    if bad:
        print(f"WARNING: {len(bad)} ledger entries sit on a kernel of the other family (positions {bad[:8]}...): the trace is not of this ledger's step", file=sys.stderr)
    sb, sa = sum(o["bound_us"] for o in out), sum(o["actual_us"] for o in out)
    wall = walls[len(walls) // 2]
    summary = dict(config=L["config"], alignment_mismatches=len(bad), launches=n, steps_averaged=len(good), sum_bound_us=sb, sum_actual_us=sa, step_wall_us=wall, frac_of_kernel_time=sb / sa, frac_of_wall=sb / wall,
                   gaps_us=wall - sa, peaks=dict(hbm_TBps=HBM / 1e12, mfma_f32_TFLOPs=MFMA / 1e12))
    json.dump(dict(summary=summary, launches=out), open(stem + ".json", "w"), indent=0)
    lines = [f"# step roofline, config {L['config']}: {n} launches, median over {len(good)} replayed steps of one rocprofv3 kernel trace",
             f"# sum(bound) {sb:.1f} us / sum(actual) {sa:.1f} us = {sb / sa:.3f} of the kernel time; step wall {wall:.1f} us (launch gaps {wall - sa:.1f} us) -> {sb / wall:.3f} of the wall",
             f"# bound = max(bytes / {HBM / 1e12:.1f} TB/s, executed flop / {MFMA / 1e12:.1f} TFLOP/s); sorted by (actual - bound)", "",
             f"{'#':>3s} {'entry point':24s} {'engine key':18s} {'kernel':44s} {'N,Cin,H,W,Cout,ks,s':>26s} {'MB':>8s} {'GF exec':>8s} {'bound':>5s} {'bound_us':>8s} {'actual':>8s} {'gap':>7s} {'frac':>5s}"]
    for o in sorted(out, key=lambda o: -(o["actual_us"] - o["bound_us"])):
        sh = "" if o["shape"] is None else ",".join(str(v) for v in o["shape"])
        lines.append(f"{o['i']:3d} {o['fn'][:24]:24s} {o['key'][:18]:18s} {o['kernel'][:44]:44s} {sh:>26s} {o['bytes'] / 1e6:8.1f} {o['flop_executed'] / 1e9:8.2f} {o['bound']:>5s} "
                     f"{o['bound_us']:8.1f} {o['actual_us']:8.1f} {o['actual_us'] - o['bound_us']:7.1f} {o['bound_us'] / max(o['actual_us'], 1e-9):5.2f}")
    open(stem + ".txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:3]))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "record":
        record(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "merge":
        merge(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
