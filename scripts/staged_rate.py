"""Rate of the paths the '2m' (640 001-tap, N = 2^20) filters take: 16x -> split fused kernel (K = 32768), 8x / 4x / 2x ->
staged path (K = 65536 / 131072 / 262144). Device-resident s32 PCM, 8 channels, blocks chosen for ~100 M samples per call."""
import json, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "totton-rasp-gpu-dsp_amd"))
import bench
import filter_design as fd
import totton_rasp_gpu_dsp_amd as ups

hip = bench.Hip()
hip.check(hip.lib.hipSetDevice(0), "hipSetDevice")
tmp = Path(tempfile.mkdtemp())
ratios = [int(a) for a in sys.argv[1:]] or [16, 8, 4, 2]
for ratio in ratios:
    h = fd.design(640_000, ratio, "48k", "linear")
    p = fd.export(h, tmp, fd.base_name("48k", ratio, 640_000, "linear"), ratio)
    filt = ups.Filter(p, device=0)
    channels, blocks = 8, 32
    eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
    nin, B = eng.in_frames, eng.out_frames
    d_in, d_out = hip.malloc(eng.in_bytes(blocks)), hip.malloc(eng.out_bytes(blocks))
    x = (np.random.default_rng(ratio).standard_normal((blocks * nin, channels)) * 0.05 * 2147483647).astype("<i4")
    hip.h2d(d_in, x)
    for _ in range(2):
        eng.process_device(d_in, d_out, blocks)
    hip.sync()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        eng.process_device(d_in, d_out, blocks)
    hip.sync()
    dt = (time.perf_counter() - t0) / n
    samples = blocks * B * channels
    L = ratio
    alg = samples * 4 * (1 + 1 / L)
    print(json.dumps({"ratio": ratio, "path": eng.path, "two_level": eng.last_two_level, "K": (1 << 20) // (2 * L), "ms_per_call": round(dt * 1e3, 3),
                      "Gsamples_per_s": round(samples / dt / 1e9, 2), "frac_of_8TBps": round(alg / dt / 8e12, 4)}), flush=True)
    hip.free(d_in); hip.free(d_out); eng.close()
