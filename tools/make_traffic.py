"""Assemble profiles/rNN_traffic.json (HBM-side bytes per launch; what bench.py reports as roofline.traffic) from the PMC passes of a round.
    python tools/make_traffic.py <rNN_traffic_style.json> <gpurun_out/pmc_conv_rNN> <out.json>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_conv_summary import last_dispatches


TRAFFIC_SOURCES = ("maxstyle_amd/csrc/ms_conv_wide.h", "maxstyle_amd/csrc/ms_conv_kernel.h", "maxstyle_amd/csrc/ms_common.h", "maxstyle_amd/csrc/ms_style_fused.hip",
                   "maxstyle_amd/csrc/ms_style.hip", "maxstyle_amd/csrc/ms_conv_inst_wino.hip")


def main(style_json, conv_root, out):
    res, detail = {}, {}
    st = json.load(open(style_json))
    for key, kern in (("maxstyle_fwd_l4", "style_fused_kernel"), ("maxstyle_bwd_l4", "restyle_bwd_kernel")):
        for k, v in st.items():
            if kern in k:
                # K2 is launched with and without dx by tools/bench_kernels.py: price the with-dx launch (max over launches)
                res[key] = (v["read_bytes_max"] + v["write_bytes_max"]) if key == "maxstyle_bwd_l4" else v["total_bytes"]; detail[key] = v
    for key, w in (("conv_dgrad_actbwd_c16_256", "dgrad_actbwd"), ("conv_dgrad_plain_c16_256", "dgrad_plain"), ("conv3x3_c16_256", "fwd_pro0"),
                   ("conv_fwd_pro1_c16_256", "fwd_pro1"), ("conv_dgrad_acc_c16_256", "dgrad_acc"), ("conv_dgrad_c32_128", "dgrad_nt2")):
        f = last_dispatches(os.path.join(conv_root, "fetch_" + w))
        wr = last_dispatches(os.path.join(conv_root, "write_" + w))
        if f and wr:
            rd = f.get("FETCH_SIZE", 0.0) * 1024 * 2.0
            wb = wr.get("WRITE_SIZE", 0.0) * 1024
            res[key] = rd + wb; detail[key] = {"read_bytes": rd, "write_bytes": wb, "kernel": f.get("_kernel")}
    res["_detail"] = detail
    # the kernel sources these figures price: bench.py reports `traffic_fresh` = they are unchanged since (tests/test_bench_contract_gpu.py fails on stale figures)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res["_sources"] = {f: hashlib.sha256(open(os.path.join(root, f), "rb").read()).hexdigest() for f in TRAFFIC_SOURCES}
    res["_how"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/replay_conv.py (one launch of the C2 step replayed on its live "
                   "buffers) and tools/bench_kernels.py --only L4; KiB -> bytes, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts the 128-B requests of "
                   "16-B/lane streaming reads as 64 B), WRITE_SIZE as reported")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
