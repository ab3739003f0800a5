#!/usr/bin/env python3
"""End-to-end rate of MultiEngine on ONE GPU listed several times: whole streams per slot against contiguous channel groups
per slot (pitched 2-D copies straight out of / into the caller's interleaved frames). BASELINE configs[4] shape:
32 channels, 160k-tap 8x linear filter, 16 blocks per call, pinned host buffers."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import totton_rasp_gpu_dsp_amd as ups  # noqa: E402

path = ROOT / "data" / "coefficients" / "filter_48k_8x_160000_linear_phase.json"
channels, blocks = 32, 16


def rate(multi, nbytes_in, nbytes_out, seconds=1.5):
    pin_in, pin_out = ups.PinnedBuffer(nbytes_in), ups.PinnedBuffer(nbytes_out)
    pin_in.array[:] = np.random.default_rng(1).integers(0, 255, nbytes_in, dtype=np.uint8)
    for _ in range(2):
        multi.process_host(pin_in.array, blocks, out=pin_out.array)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        multi.process_host(pin_in.array, blocks, out=pin_out.array)
        n += 1
    dt = (time.perf_counter() - t0) / n
    pin_in.close()
    pin_out.close()
    return dt


one = ups.MultiEngine(path, [0], 1, channels)
nin, nout = one.in_bytes(blocks), one.out_bytes(blocks)
samples = blocks * one.out_frames * channels
dt = rate(one, nin, nout)
print(f"1 slot, whole frames            {samples / dt / 1e9:7.2f} Gsamples/s  {(nin + nout) / dt / 1e9:6.1f} GB/s over the host link")
one.close()
for slots in (2, 4):
    m = ups.MultiEngine(path, [0] * slots, 1, channels, split_time=True)
    dt = rate(m, nin, nout)
    print(f"{slots} slots x block ranges (contiguous) {samples / dt / 1e9:7.2f} Gsamples/s  {(nin + nout) / dt / 1e9:6.1f} GB/s")
    m.close()
for slots in (2, 4, 8):
    m = ups.MultiEngine(path, [0] * slots, 1, channels, split_channels=True)
    dt = rate(m, nin, nout)
    print(f"{slots} slots x {channels // slots:2d} channels ({channels // slots * 4:3d}-byte rows) {samples / dt / 1e9:7.2f} Gsamples/s  "
          f"{(nin + nout) / dt / 1e9:6.1f} GB/s   worker cpus: {m.worker_cpus(0) or '(not pinned)'}")
    m.close()
