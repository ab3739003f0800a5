"""mi_ups_process_block latency (the reference's call shape), A/B over engine switches given as env assignments."""
import json, os, subprocess, sys
ARMS = [a for a in sys.argv[1:]] or ["base:", "three_streams:MIUPS_EXP_HOST_THREE_STREAMS=1"]
CODE = """
import sys, json
sys.path.insert(0, '.')
import bench
import totton_rasp_gpu_dsp_amd as ups
print(json.dumps(bench.process_block_latency(ups, 0, bench.ROOT / 'data' / 'coefficients' / bench.CONFIGS[2][0], calls=2000)))
"""
for rnd in range(2):
    for arm in ARMS:
        name, _, envs = arm.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        try:
            d = json.loads(line)
            print(f"{name:16s} p50 {d['p50_ms']:.4f} p99 {d['p99_ms']:.4f} mean {d['mean_ms']:.4f}", flush=True)
        except Exception:
            print(name, "FAILED", line, flush=True)
