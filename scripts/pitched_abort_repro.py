#!/usr/bin/env python3
"""ONE run of the recorded hipMemcpy2DAsync abort (profiles/r03_g_split_park.txt, last paragraph): channel split, one s16
channel per slot = 2-byte rows at a 4-byte pitch out of hipHostRegister'ed memory, with the host-packing fallback switched
off (MIUPS_EXP_PITCHED_ANY_WIDTH=1). Runs in a child process with the abort-backtrace hook installed and stderr uncaptured,
so that the runtime's own message and the native stack reach the log. Diagnostic: run it once, read the log."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import totton_rasp_gpu_dsp_amd as ups
ups.lib.mi_debug_install_abort_backtrace()
path = %r
multi = ups.MultiEngine(path, [0, 0, 0], 1, 2, ups.PCM_S16, ups.PCM_S16, split_channels=True)
blocks = 2
x = (np.random.default_rng(1).standard_normal((1, blocks * multi.in_frames, 2)) * 3000).astype("<i2")
mode = sys.argv[1]
buf = x.copy()
if mode == "registered":
    reg = ups.RegisteredBuffer(buf.view(np.uint8).reshape(-1))
    y = multi.process_host(reg.array, blocks)
else:
    y = multi.process_host(buf, blocks)   # pageable: the call registers it itself
print(mode, "ok", int(np.abs(y.view("<i2")).max()), "unsafe copies", ups.unsafe_host_copies(), flush=True)
''' % (str(ROOT), str(ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"))
for mode in ("pageable", "registered"):
    env = dict(os.environ, MIUPS_EXP_PITCHED_ANY_WIDTH="1", MIUPS_ABORT_BACKTRACE="/dev/stderr", AMD_LOG_LEVEL="1")
    r = subprocess.run([sys.executable, "-c", CHILD, mode], env=env)
    print(f"== {mode}: exit code {r.returncode}", flush=True)
