"""Diagnostic twin of tests/test_gpu_stream_host.py::test_eq_swap_does_not_stall_a_running_stream: when do slow calls of the
streaming thread happen relative to the EQ swaps of the main thread? Prints every call slower than 2 ms with its start time,
and the swap intervals."""
import ctypes as C, json, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
import totton_rasp_gpu_dsp_amd as ups
hip = bench.Hip()
hip.check(hip.lib.hipSetDevice(0), "hipSetDevice")
F4X = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
PROFILES = json.loads((ROOT / "tests" / "golden" / "g4_eq_profiles.json").read_text())
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for r in range(rounds):
    filt = ups.Filter(F4X, device=0)
    eng = ups.Engine(filt, 1, 2, ups.PCM_S32, ups.PCM_S32)
    blocks = 64
    d_in, d_out = hip.malloc(eng.in_bytes(blocks)), hip.malloc(eng.out_bytes(blocks))
    x = (np.random.default_rng(1).standard_normal((blocks * eng.in_frames, 2)) * 0.1 * 2147483647).astype("<i4")
    hip.h2d(d_in, x)
    stream = hip.stream()
    calls, stop = [], threading.Event()
    def run():
        hip.lib.hipSetDevice(0)
        while not stop.is_set():
            t0 = time.perf_counter()
            eng.process_device(d_in, d_out, blocks, stream)
            t1 = time.perf_counter()
            hip.check(hip.lib.hipStreamSynchronize(C.c_void_p(stream)), "sync")
            calls.append((t0, t1 - t0, time.perf_counter() - t1))
    t = threading.Thread(target=run); t.start()
    time.sleep(0.3)
    swaps = []
    for k in range(4):
        t0 = time.perf_counter()
        filt.set_eq(PROFILES["opra10"] if k % 2 == 0 else "", 705600.0)
        swaps.append((t0, time.perf_counter()))
        time.sleep(0.05)
    time.sleep(0.2); stop.set(); t.join()
    base = calls[0][0]
    slow = [(round((c[0] - base) * 1e3, 2), round(c[1] * 1e3, 2), round(c[2] * 1e3, 2)) for c in calls[5:] if c[1] + c[2] > 2e-3]
    print(f"round {r}: {len(calls)} calls, median {np.median([c[1] + c[2] for c in calls]) * 1e6:.0f} us, parts {eng.last_phase_parts}; "
          f"swaps at ms {[(round((a - base) * 1e3, 1), round((b - a) * 1e3, 1)) for a, b in swaps]}; slow calls (start ms, enqueue ms, sync ms): {slow}", flush=True)
    hip.free(d_in); hip.free(d_out); eng.close(); filt.close()
