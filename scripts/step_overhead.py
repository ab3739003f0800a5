"""Where the gap between a bench step and its transform kernel goes (headline shape: config 2, 256 blocks).
Same call back to back, wall clock per call, with and without the engine's two timing events per launch."""
import sys, time, json
sys.path.insert(0, ".")
import bench
import totton_rasp_gpu_dsp_amd as ups
hip = bench.Hip()
hip.check(hip.lib.hipSetDevice(0), "hipSetDevice")
w = bench.Workload(ups, hip, 0, 2, 0)
w.prime(1.0)
out = {}
for label, timing in (("timing_events_on", 64), ("timing_events_off", 0), ("timing_events_on_again", 64)):
    w.eng.enable_kernel_timing(timing)
    hip.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(400):
            w.eng.process_device(w.d_in, w.d_out, w.blocks, w.stream)
        hip.sync()
        best = min(best, (time.perf_counter() - t0) / 400 * 1e3)
    out[label] = round(best, 5)
w.eng.enable_kernel_timing(64)
for _ in range(64):
    w.eng.process_device(w.d_in, w.d_out, w.blocks, w.stream)
hip.sync()
out["kernel_ms"] = w.eng.kernel_ms_stats()
out["per_kernel_ms"] = w.per_kernel_ms()
print(json.dumps(out))
