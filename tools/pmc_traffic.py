"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes: TCC has 4 slots, FETCH_SIZE takes 3).

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
bytes of a wide coalesced streaming read (16 B/lane) -> doubled here; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]
"""
import csv, glob, json, os, sys, collections


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"]
            name = name[:name.index("(")] if "(" in name else name
            acc[name.replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), max(v)) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fa, fm = fetch.get(k, (0.0, 0.0)); wa, wm = write.get(k, (0.0, 0.0))
        fr = fa * 1024 * 2.0          # gfx950: FETCH_SIZE = 1/2 of the streamed bytes
        wr = wa * 1024
        out[k] = {"read_bytes": fr, "write_bytes": wr, "total_bytes": fr + wr, "fetch_size_raw_KiB": fa, "write_size_raw_KiB": wa,
                  "read_bytes_max": fm * 1024 * 2.0, "write_bytes_max": wm * 1024}      # max over launches: a kernel launched with and without an optional output
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
