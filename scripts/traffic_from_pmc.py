#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of scripts/gpu_r03_measure.sh (gpurun_out/<round>/pmc_<tag>.txt).
usage: traffic_from_pmc.py PREFIX [SOURCE_DIR]      e.g. traffic_from_pmc.py r03_m gpurun_out/r03m

HBM-side bytes per bench step = sum over the kernels of one mi_engine_process_device call of
(2 * FETCH_SIZE + WRITE_SIZE) KiB x launches of that kernel per call. Unit of both counters: KiB. gfx950 correction per
MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 128-B read requests at 64 B, so it is doubled; WRITE_SIZE as is."""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PREFIX = sys.argv[1] if len(sys.argv) > 1 else "r03_m"  # profiles/<PREFIX>_pmc_<tag>.txt
SRC = ROOT / (sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r03m")
TAGS = {"c2_256": ("2", 1, 2, 256), "c2_2048": ("2_2048blocks", 1, 2, 2048), "c3": ("3", 1, 8, 256), "c4": ("4", 32, 2, 32),
        "c5": ("5", 1, 32, 64)}


def parse(path):
    out, cur = {}, None
    for line in path.read_text().splitlines():
        if line.startswith("== "):
            cur = re.sub(r"^void ", "", line[3:].strip())
            out[cur] = {}
        elif cur and "mean" in line:
            name, _, rest = line.strip().partition(" ")
            m = re.search(r"mean\s+([0-9.]+)\s+\(n=(\d+)\)", line)
            out[cur][name] = (float(m.group(1)), int(m.group(2)))
    return out


def main():
    res = {"_comment": __doc__.strip().replace("\n", " ")}
    for tag, (key, streams, channels, blocks) in TAGS.items():
        k = parse(SRC / f"pmc_{tag}.txt")
        calls = k["miups::update_history_kernel"]["FETCH_SIZE"][1]  # one per bench step
        kernels, total = {}, 0.0
        for name, c in k.items():
            # "miups::" alone = a kernel of the anonymous namespace cut at its "(" by pmc_summary.py: the copy-ceiling kernel
            # of the bench line (mi_device_copy_rate), not part of a step
            if not name.startswith("miups::") or name == "miups::" or "update_history" in name or "anonymous" in name or "FETCH_SIZE" not in c:
                continue
            per_call = c["FETCH_SIZE"][1] / calls
            kib = (2 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * per_call
            hit, miss = c.get("TCC_HIT_sum", (0, 0))[0], c.get("TCC_MISS_sum", (0, 0))[0]
            kernels[name] = {"launches_per_step": per_call, "fetch_size_kib": c["FETCH_SIZE"][0], "write_size_kib": c["WRITE_SIZE"][0],
                             "bytes_per_step": int(kib * 1024), "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None}
            total += kib * 1024
        res[key] = {"streams": streams, "channels": channels, "blocks": blocks, "bytes": int(total), "kernels": kernels,
                    "source": f"profiles/{PREFIX}_pmc_{tag}.txt"}
        (ROOT / "profiles" / f"{PREFIX}_pmc_{tag}.txt").write_text((SRC / f"pmc_{tag}.txt").read_text())
    (ROOT / "profiles" / "traffic.json").write_text(json.dumps(res, indent=1) + "\n")
    for key, v in res.items():
        if key != "_comment":
            print(key, v["bytes"], {n.split("<")[0][7:]: x["bytes_per_step"] for n, x in v["kernels"].items()})


if __name__ == "__main__":
    main()
