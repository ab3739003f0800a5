"""Host-side stress of MultiEngine's worker threads (no GPU fault involved: the question is a silent SIGABRT inside the HIP
runtime seen once in tests/test_gpu_stream_host.py::test_time_split_is_bit_identical_to_one_engine[4 slots, 2 streams, 8 ch]).
Repeats that scenario N times in one process and checks every result against one engine.
usage: multi_stress.py [rounds]      env MIUPS_EXP_MULTI_THREE_STREAMS=1 puts the workers' engines on the three-stream host path without the small-call kernels"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import totton_rasp_gpu_dsp_amd as ups

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
filt = ups.Filter(path, device=0)
streams, channels = 2, 8
one = ups.Engine(filt, streams, channels, ups.PCM_S32, ups.PCM_S32)
multi = ups.MultiEngine(path, [0, 0, 0, 0], streams, channels, split_time=True)
rng = np.random.default_rng(1)
t0 = time.time()
for r in range(rounds):
    for blocks in (9, 3, 6, 1, 2):
        x = (np.clip(rng.standard_normal((streams, blocks * one.in_frames, channels)) * 0.1, -1, 1) * 2147483647).astype("<i4")
        a = multi.process_host(x, blocks)
        b = one.process_host(x, blocks)
        assert np.array_equal(a, b), (r, blocks)
    multi.reset()
    one.reset()
    print(f"round {r} ok ({time.time() - t0:.1f} s)", flush=True)
print("done", "three-stream workers" if os.environ.get("MIUPS_EXP_MULTI_THREE_STREAMS") else "default workers")
