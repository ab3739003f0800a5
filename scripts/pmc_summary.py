#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs written by scripts/gpu_pmc.sh: per kernel name,
the mean of every counter over its dispatches (skipping the first two, warm-up)."""
from __future__ import annotations

import csv
import sys
from collections import defaultdict
from pathlib import Path


def main(root: str) -> None:
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(Path(root).rglob("*counter_collection.csv")):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "?").split("(")[0]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kernel, counters in acc.items():
        print(f"== {kernel}")
        for cname in sorted(counters):
            vals = counters[cname]
            steady = vals[2:] if len(vals) > 4 else vals
            print(f"  {cname:28s} mean {sum(steady) / len(steady):16.1f}  (n={len(vals)})")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
