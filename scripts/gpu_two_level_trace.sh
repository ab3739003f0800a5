#!/usr/bin/env bash
# per-dispatch durations (in launch order) of the two-level kernels at one ratio
set -u
out=gpurun_out/two_level_trace
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
r=${1:-8}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 scripts/staged_rate.py $r > $out/rate.log 2>&1 || exit 1
f=$(find $out/raw -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-40:]:
    n = r["Kernel_Name"].split("(")[0].replace("void miups::", "")[:40]
    print(f'{n:42s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))}')
PY
rm -rf $out/raw
