#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary that hands over HOST buffers
(mi_engine_process_host: pageable host memory -> device -> kernels -> host), on
bench config 2. Reported in DESIGN.md / profiles/r01_summary.md; never bench.py's `value`."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import totton_rasp_gpu_dsp_amd as ups  # noqa: E402

blocks, channels = 256, 2  # a 74 s call; the PCIe-inclusive rate does not depend on the batch size
filt = ups.Filter(ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json")
eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
x = np.clip(np.random.default_rng(0).standard_normal((blocks * eng.in_frames, channels)) * 0.2, -1, 1)
x = (x * (2**31 - 1)).astype("<i4")
for _ in range(3):
    y = eng.process_host(x, blocks)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    y = eng.process_host(x, blocks)
dt = (time.perf_counter() - t0) / n
out_samples = y.size
print(json.dumps({"boundary": "mi_engine_process_host (pageable host buffers)", "ms_per_call": round(dt * 1e3, 3),
                  "Msamples_per_s": round(out_samples / dt / 1e6, 1),
                  "host_bytes_per_call": int(x.nbytes + y.nbytes),
                  "host_GB_per_s": round((x.nbytes + y.nbytes) / dt / 1e9, 2)}))
