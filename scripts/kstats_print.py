#!/usr/bin/env python3
"""Print name, calls, average ns and share of every kernel in a rocprofv3 --stats output directory."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        print(r["Name"][:80].ljust(80), r["Calls"].rjust(6), r["AverageNs"].rjust(14), r["Percentage"])
