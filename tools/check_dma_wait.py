"""Static check of the built ISA: the counted wait behind the weights' LDS-DMA of the Winograd kernels (ms_conv_wide.h, producer loop).

    python tools/check_dma_wait.py <file.s> [...]          (hipcc -S --cuda-device-only of a ms_conv_inst_wino*.hip)

The staging waves issue the chunk's weights as `buffer_load_dwordx4 ... lds` pieces, then the NEXT chunk's activation loads, then `s_waitcnt vmcnt(kDataLoads)` and the
chunk barrier: loads complete in order, so the wait covers the DMA iff at least kDataLoads loads were issued behind it.  kDataLoads is a source-level count; if the compiler
removes loads it can prove dead (the flat form's halo loads, round 6) the wait silently stops covering the last DMA pieces.  This script counts, per kernel, the vector-memory
instructions between the last DMA piece and every counted wait (N > 0) in front of the next s_barrier, and fails when N exceeds them."""
import re
import sys


def check(path):
    lines = open(path).read().split("\n")
    bad, seen = [], 0
    kernel = None
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(_Z\S+):", l)
        if m:
            kernel = m.group(1)
        s = l.strip()
        if s.startswith("buffer_load") and s.endswith("lds"):
            # the last piece of this DMA group
            j = i
            k = i + 1
            while k < len(lines) and not lines[k].strip().startswith("s_barrier") and not lines[k].startswith(".Lfunc_end"):
                t = lines[k].strip()
                if t.startswith("buffer_load") and t.endswith("lds"):
                    j = k
                k += 1
            issued = 0
            for q in range(j + 1, k):
                t = lines[q].strip()
                op = t.split()[0] if t else ""
                if op.startswith(("buffer_load", "buffer_store", "global_load", "global_store", "flat_load", "flat_store")):
                    issued += 1
                w = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", t)
                if w and int(w.group(1)) > 0 and lines[q - 1].strip().startswith(";;#ASMSTART"):      # the SOURCE's counted wait (inline asm), not the compiler's own
                    seen += 1
                    if int(w.group(1)) > issued:
                        bad.append((kernel, q + 1, int(w.group(1)), issued))
            i = k
        i += 1
    return seen, bad


def main():
    rc = 0
    for p in sys.argv[1:]:
        seen, bad = check(p)
        print(f"{p}: {seen} counted waits behind an LDS-DMA group, {len(bad)} of them wait for fewer loads than were issued")
        for k, ln, n, issued in bad:
            print(f"  {k} line {ln}: s_waitcnt vmcnt({n}) with {issued} loads issued behind the DMA")
            rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
