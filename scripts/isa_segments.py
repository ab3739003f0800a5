#!/usr/bin/env python3
"""Instruction mix of one kernel of a gfx950 assembly file, split at s_barrier (the fused kernel has one
barrier per FFT pass, so a segment ~ a pass). Usage: isa_segments.py FILE.s SUBSTRING_OF_MANGLED_NAME"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    segs, cur = [], collections.Counter()
    for ln in lines[start + 1:end]:
        t = ln.strip()
        if not t or t[0] in ";." or t.endswith(":") or re.match(r"^\.?LBB\S+:", t):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            segs.append(cur)
            cur = collections.Counter()
        else:
            cur[op] += 1
    segs.append(cur)
    tot = collections.Counter()
    for i, c in enumerate(segs):
        tot.update(c)
        v = sum(k for o, k in c.items() if o.startswith("v_"))
        pk = sum(k for o, k in c.items() if o.startswith("v_pk_"))
        ds = sum(k for o, k in c.items() if o.startswith("ds_"))
        vm = sum(k for o, k in c.items() if o.startswith(("global_", "buffer_", "scratch_", "flat_")))
        sc = sum(k for o, k in c.items() if o.startswith("s_"))
        top = ", ".join(f"{o}:{k}" for o, k in c.most_common(8))
        print(f"seg{i:2d} valu={v:5d} (pk {pk:4d}) ds={ds:4d} vmem={vm:4d} salu={sc:4d} | {top}")
    print("scratch ops:", sum(k for o, k in tot.items() if o.startswith("scratch_")),
          " total valu:", sum(k for o, k in tot.items() if o.startswith("v_")))


if __name__ == "__main__":
    main()
