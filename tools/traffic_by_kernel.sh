#!/bin/bash
# Fabric traffic per kernel of an EAGER step against the step budget's algorithmic bytes (GPU box; needs gpurun_out/sb/step_budget_<cfg>.json from tools/step_budget_run.sh
# in the same call or a copy under profiles/):  bash tools/traffic_by_kernel.sh c2 <budget.json> > out.txt
# FETCH_SIZE is doubled (MI355X_MICROARCH.md: 128-byte requests tallied at 64 B for 16-byte-per-lane streaming reads); other access widths are uncalibrated - read the
# ratio column as a screen for over-fetch (>= 1.5), not as an absolute.
CFG=$1; BUDGET=$2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tbk_$CFG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python tools/one_step.py $CFG 3 > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python tools/one_step.py $CFG 3 > /dev/null 2> $O/write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- python tools/one_step.py $CFG 3 > /dev/null 2> $O/tcc.err
python - "$O" "$BUDGET" <<'PY'
import csv, glob, json, sys, collections, re
O, B = sys.argv[1], sys.argv[2]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"^ms::", "", n); n = re.sub(r"\(.*$", "", n)
    return n[:44]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
bud = json.load(open(B))
alg = collections.defaultdict(list)
for r in bud["launches"]:
    alg[short(r["kernel"])].append(r["bytes"] / 1e6)
print(f"{'kernel':52s} {'launches/step':>13s} {'algorithmic MB':>15s} {'2*FETCH+WRITE MB':>17s} {'ratio':>6s} {'L2 hit':>7s}   (per launch, mean)")
out = []
for k, c in agg.items():
    if k not in alg or not c.get("FETCH_SIZE"):
        continue
    f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 / 1e6 * 2
    w = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"])) * 1024 / 1e6
    a = sum(alg[k]) / len(alg[k])
    h, m = sum(c.get("TCC_HIT_sum", [0])), sum(c.get("TCC_MISS_sum", [0]))
    out.append(((f + w) / max(a, 1e-9), k, len(alg[k]), a, f + w, h / max(h + m, 1.0)))
for r, k, n, a, t, hr in sorted(out, reverse=True):
    print(f"{k:52s} {n:13d} {a:15.1f} {t:17.1f} {r:6.2f} {hr:7.2f}")
PY
rm -rf $O/fetch $O/write $O/tcc
