#!/usr/bin/env python3
"""Run the bench workload once with the MIUPS_STAMPS library variant and print the
per-stage cycle breakdown of a few workgroups (diagnostic; see MI_STAMP)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
os.environ.setdefault("MIUPS_LIB", str(ROOT / "totton-rasp-gpu-dsp_amd" / "lib_ablate" / "libmi_upsampler_STAMPS.so"))
BRIEF = os.environ.get("STAMPS_BRIEF") == "1"
sys.path.insert(0, str(ROOT))
import totton_rasp_gpu_dsp_amd as ups  # noqa: E402

filt = ups.Filter(ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json")
eng = ups.Engine(filt, 1, 2, ups.PCM_S32, ups.PCM_S32)
blocks = 256
x = (np.random.default_rng(0).standard_normal((blocks * eng.in_frames, 2)) * 0.2 * 2**31).astype("<i4")
for _ in range(3):
    eng.process_host(x, blocks)
ups.lib.mi_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
n = 32 * 8 * 192
buf = (C.c_ulonglong * n)()
assert ups.lib.mi_debug_read_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(32, 8, 192).astype(np.int64)
names = {0: "start", 1: "fwd_first", 2: "sync", 3: "fwd_mid256", 4: "sync", 5: "fwd_mid16", 6: "sync", 7: "fwd_last",
         8: "split"}
for p in range(4):
    for k, nm in enumerate(["phase_in", "inv_first", "sync", "inv_mid16", "sync", "inv_mid256", "sync", "inv_last",
                            "sync"]):
        names[9 + 10 * p + k] = f"p{p}.{nm}"
order = sorted(names)
for wg in (() if BRIEF else (0, 5, 17)):
    for wave in (0, 3, 7):
        t = st[wg, wave]
        print(f"--- wg {wg} wave {wave}: total {t[129] - t[0]} cycles; epilogue {t[129] - t[128]}")
        for cc in (0, 1):
            base = 64 * cc
            prev = t[base + 0]
            line = []
            for sid in order[1:]:
                cur = t[base + sid]
                line.append(f"{names[sid]}={cur - prev}")
                prev = cur
            print(f"  ch{cc}: " + " ".join(line))
# aggregate over all recorded workgroups/waves
agg = {}
for cc in (0, 1):
    prev = st[:, :, 64 * cc]
    for sid in order[1:]:
        cur = st[:, :, 64 * cc + sid]
        key = names[sid].split(".")[-1]
        agg.setdefault(key, []).append((cur - prev).mean())
        prev = cur
tot = (st[:, :, 129] - st[:, :, 0]).mean()
print("=== mean cycles per stage summed over 2 channels x (fwd + 4 phases):")
for k, v in agg.items():
    print(f"  {k:12s} {sum(v):10.0f}  ({100 * sum(v) / tot:5.1f}%)")
print(f"  epilogue     {(st[:, :, 129] - st[:, :, 128]).mean():10.0f}")
print(f"  total        {tot:10.0f}")
