"""mi_ups_process_block latency (one channel-block per call) for every shipped filter."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
import totton_rasp_gpu_dsp_amd as ups
for f in sorted((ROOT / "data" / "coefficients").glob("*.json")):
    d = bench.process_block_latency(ups, 0, f, calls=500)
    print(f"{f.name:48s} p50 {d['p50_ms']:.4f} ms  p99 {d['p99_ms']:.4f}  {d['Msamples_per_s_at_p50']:.0f} Msamples/s", flush=True)
