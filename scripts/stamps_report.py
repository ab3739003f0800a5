#!/usr/bin/env python3
"""Run a bench workload once with a MIUPS_STAMPS library variant and print the
per-stage cycle breakdown of the fused kernel (diagnostic; see MI_STAMP).

env: MIUPS_LIB (default lib_ablate/libmi_upsampler_STAMPS.so), STAMPS_CONFIG = 2|3|4|5,
     STAMPS_BRIEF=1 (aggregate only). Stamp slots cover the LAST two channels and the
     LAST four phases a workgroup processed."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
os.environ.setdefault("MIUPS_LIB", str(ROOT / "totton-rasp-gpu-dsp_amd" / "lib_ablate" / "libmi_upsampler_STAMPS.so"))
BRIEF = os.environ.get("STAMPS_BRIEF") == "1"
os.environ.setdefault("MIUPS_EXP_HOST_SUBBATCHES", "1")  # one ProcessDevice call per process_host: the bench's kernel path
sys.path.insert(0, str(ROOT))
import totton_rasp_gpu_dsp_amd as ups  # noqa: E402

CONFIGS = {2: ("filter_44k_4x_80000_min_phase.json", 2, 256, None),
           3: ("filter_48k_16x_80000_min_phase.json", 8, 256, 768000.0),
           4: ("filter_44k_2x_80000_min_phase.json", 2, 1024, None),   # split kernel: fwd_first..fwd_last = both halves
           5: ("filter_48k_8x_160000_linear_phase.json", 32, 64, 768000.0)}
fname, channels, blocks, eq_fs = CONFIGS[int(os.environ.get("STAMPS_CONFIG", "2"))]
blocks = int(os.environ.get("STAMPS_BLOCKS", blocks))
filt = ups.Filter(ROOT / "data" / "coefficients" / fname)
if eq_fs:
    filt.set_eq(json.loads((ROOT / "tests" / "golden" / "g4_eq_profiles.json").read_text())["opra10"], eq_fs)
eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
x = np.clip(np.random.default_rng(0).standard_normal((blocks * eng.in_frames, channels)) * 0.2, -1, 1)
x = (x * (2**31 - 1)).astype("<i4")
for _ in range(3):
    eng.process_host(x, blocks)
ups.lib.mi_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
n = 32 * 8 * 192
buf = (C.c_ulonglong * n)()
assert ups.lib.mi_debug_read_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(32, 8, 192).astype(np.int64)
waves = max(1, min(8, (filt.config["fft_size"] // filt.config["upsample_factor"] // 2 // 32) // 64))
if filt.config["fft_size"] // filt.config["upsample_factor"] // 2 == 32768:
    waves = 8  # split kernel: 16384-point transforms
st = st[:, :waves]
names = {0: "start", 1: "fwd_first", 2: "sync", 3: "fwd_mid256", 4: "sync", 5: "fwd_mid16", 6: "sync", 7: "fwd_last",
         8: "split"}
for p in range(4):
    for k, nm in enumerate(["phase_in", "inv_first", "sync", "inv_mid16", "sync", "inv_mid256", "sync", "inv_last",
                            "sync"]):
        names[9 + 10 * p + k] = f"p{p}.{nm}"
order = sorted(names)
print(f"config {fname} channels {channels} blocks {blocks} path {eng.path} waves/wg {waves}")
for wg in (() if BRIEF else (0, 5, 17)):
    for wave in sorted({0, waves // 2, waves - 1}):
        t = st[wg, wave]
        print(f"--- wg {wg} wave {wave}: total {t[129] - t[0]} cycles; epilogue {t[129] - t[128]}")
        for cc in (0, 1):
            base = 64 * cc
            prev = t[base + 0]
            line = []
            for sid in order[1:]:
                cur = t[base + sid]
                if cur == 0 or prev == 0:
                    continue
                line.append(f"{names[sid]}={cur - prev}")
                prev = cur
            print(f"  ch{cc}: " + " ".join(line))
# aggregate over all recorded workgroups/waves: mean duration of each stage kind (per occurrence)
agg = {}
for cc in (0, 1):
    prev = st[:, :, 64 * cc]
    for sid in order[1:]:
        cur = st[:, :, 64 * cc + sid]
        if not cur.any():
            continue
        key = names[sid].split(".")[-1]
        d = (cur - prev)
        ok = (cur > 0) & (prev > 0) & (d >= 0)
        if ok.any():
            agg.setdefault(key, []).append(d[ok].mean())
        prev = cur
tot = (st[:, :, 129] - st[:, :, 0]).mean()
print("=== mean cycles per occurrence of each stage (last 2 channels x last 4 phases recorded):")
for k, v in agg.items():
    print(f"  {k:12s} {np.mean(v):10.0f}  x{len(v)} recorded = {sum(v):10.0f}  ({100 * sum(v) / tot:5.1f}% of wg)")
print(f"  epilogue     {(st[:, :, 129] - st[:, :, 128]).mean():10.0f}")
print(f"  total        {tot:10.0f}")
