#!/usr/bin/env python3
"""Diagnostic: the stereo 256-block call against the same stream in 64-block calls, per block and channel (s32)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import totton_rasp_gpu_dsp_amd as ups
path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
filt = ups.Filter(path)
eng = ups.Engine(filt, 1, 2, ups.PCM_S32, ups.PCM_S32)
nin, blocks, B = eng.in_frames, 256, filt.config["block_size"]
xf = np.clip(np.random.default_rng(11).standard_normal((blocks * nin, 2)) * 0.1, -1, 1)
raw = (xf * 2147483647).astype("<i4")
whole = eng.process_host(raw, blocks).copy().view("<i4").reshape(blocks, B, 2)
eng.reset()
step = int(sys.argv[1]) if len(sys.argv) > 1 else 64
parts = np.concatenate([eng.process_host(raw[i * step * nin:(i + 1) * step * nin], step).copy().view("<i4").reshape(step, B, 2)
                        for i in range(blocks // step)])
d = (whole.astype(np.int64) - parts.astype(np.int64))
for b in range(blocks):
    for c in range(2):
        n = np.count_nonzero(d[b, :, c])
        if n:
            idx = np.nonzero(d[b, :, c])[0]
            print(f"block {b} ch {c}: {n} differ, max |d| {np.abs(d[b,:,c]).max()}, first at {idx[0]}, last at {idx[-1]}")
print("done")
