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
        "c5": ("5", 1, 32, 64),
        # rows outside BASELINE's configs (bench.py --row): the 640k-tap filters through the two-level path, the 8x 80k-tap filter
        "2m_8x": ("2m_8x", 1, 8, 16), "2m_2x": ("2m_2x", 1, 8, 16), "8x80k_2ch": ("8x80k_2ch", 1, 2, 256),
        "8x80k_32ch": ("8x80k_32ch", 1, 32, 64)}


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
    try:  # tags this visit did not measure keep their previous record
        res.update({k: v for k, v in json.loads((ROOT / "profiles" / "traffic.json").read_text()).items() if k != "_comment"})
    except (OSError, ValueError):
        pass
    for tag, (key, streams, channels, blocks) in TAGS.items():
        if not (SRC / f"pmc_{tag}.txt").exists():
            continue
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
    # issue side of each config's TRANSFORM kernel (SQ counters, per launch; SQ cycle counters tick once per 4 clocks)
    for itag, key in (("issue_c2_256", "2"), ("issue_c3", "3"), ("issue_c4", "4"), ("issue_c5", "5")):
        issue = SRC / f"pmc_{itag}.txt"
        if not issue.exists() or key not in res:
            continue
        k = parse(issue)
        name = next(n for n in k if "fused_kernel" in n or "fused_split_kernel" in n)
        c = {n: v[0] for n, v in k[name].items()}
        waves = c["SQ_WAVES"]
        res[key]["issue"] = {
            "kernel": name, "waves": int(waves), "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / waves, 1),
            "lds_insts_per_wave": round(c["SQ_INSTS_LDS"] / waves, 1), "salu_insts_per_wave": round(c["SQ_INSTS_SALU"] / waves, 1),
            "vmem_rd_insts_per_wave": round(c["SQ_INSTS_VMEM_RD"] / waves, 1),
            "vmem_wr_insts_per_wave": round(c["SQ_INSTS_VMEM_WR"] / waves, 1),
            "wave_clocks": round(4 * c["SQ_WAVE_CYCLES"] / waves, 0),
            "wait_inst_any_share_of_wave_clocks": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
            "wait_inst_lds_share_of_wave_clocks": round(c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], 3),
            "lds_bank_conflict_share_of_lds_active": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 4),
            "source": f"profiles/{PREFIX}_pmc_{itag}.txt"}
        (ROOT / "profiles" / f"{PREFIX}_pmc_{itag}.txt").write_text(issue.read_text())
    (ROOT / "profiles" / "traffic.json").write_text(json.dumps(res, indent=1) + "\n")
    for key, v in res.items():
        if key != "_comment":
            print(key, v["bytes"], {n.split("<")[0][7:]: x["bytes_per_step"] for n, x in v["kernels"].items()})


if __name__ == "__main__":
    main()
