"""Instruction census of one kernel from `hipcc -S` output: per basic block, how many vector / scalar / matrix / LDS / memory instructions.

    python tools/isa_census.py <file.s> <substring of the mangled kernel name> [--blocks] [--dump LABEL]

Used with the PMC counters (SQ_INSTS_VALU / _SALU / _MFMA): trip counts x per-block counts should reproduce the measured totals; the blocks that
carry the difference between the measured count and the necessary arithmetic are the ones to rewrite."""
import re
import sys
import collections


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith("v_accvgpr"):
        return "VALU_ACC"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith("s_barrier"):
        return "BAR"
    if op.startswith("s_nop") or op.startswith("s_sleep"):
        return "NOP"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "BR"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") is False and re.match(r"^(_Z\S+):", l) and key in l:
            start = i
            break
        m = re.match(r"^(_Z\S+):", l)
        if m and key in m.group(1):
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print("kernel:", lines[start].split(":")[0], " lines", start, "-", end)
    blocks = []
    cur = ("entry", collections.Counter(), [])
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\S+):", s)
            if m:
                blocks.append(cur)
                cur = (m.group(1), collections.Counter(), [])
            continue
        op = s.split()[0]
        cur[1][classify(op)] += 1
        cur[2].append(s)
    blocks.append(cur)
    tot = collections.Counter()
    for name, c, body in blocks:
        tot.update(c)
    print("static totals:", dict(tot))
    if show_blocks:
        for name, c, body in blocks:
            n = sum(c.values())
            if n == 0:
                continue
            tail = body[-1] if body else ""
            print(f"{name:14s} n={n:5d} VALU={c['VALU']:4d} SALU={c['SALU']:4d} MFMA={c['MFMA']:3d} LDS={c['LDS']:3d} VMEM={c['VMEM']:3d} WAIT={c['WAIT']:3d} BAR={c['BAR']:2d} SMEM={c['SMEM']:2d} NOP={c['NOP']:2d} | {tail}")
    if dump:
        for name, c, body in blocks:
            if name == dump:
                print("\n".join(body))
    for l in lines[end:end + 60]:
        if any(k in l for k in (".num_vgpr", ".num_agpr", "numbered_sgpr", "private_seg_size")) and key in l:
            print(l.strip())


if __name__ == "__main__":
    main()
