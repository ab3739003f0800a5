"""Cost of a multi-stream mi_engine_process_host call on pageable memory (the call page-locks its buffers for its own duration)
against the same call on mi_host_alloc memory, by call size."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import totton_rasp_gpu_dsp_amd as ups
path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
filt = ups.Filter(path, device=0)
for streams, blocks in ((2, 1), (2, 8), (8, 8), (8, 64)):
    eng = ups.Engine(filt, streams, 2, ups.PCM_S32, ups.PCM_S32)
    x = (np.random.default_rng(1).standard_normal((streams, blocks * eng.in_frames, 2)) * 0.1 * 2147483647).astype("<i4")
    pin_in, pin_out = ups.PinnedBuffer(x.nbytes), ups.PinnedBuffer(eng.out_bytes(blocks) * streams)
    pin_in.array[:] = x.view(np.uint8).reshape(-1)
    out = np.empty(eng.out_bytes(blocks) * streams, np.uint8)
    res = {}
    for name, a, o in (("pageable", x, out), ("pinned", pin_in.array, pin_out.array)):
        for _ in range(3):
            eng.process_host(a, blocks, out=o)
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            eng.process_host(a, blocks, out=o)
        res[name] = (time.perf_counter() - t0) / n * 1e3
    mb = (x.nbytes + out.nbytes) / 1e6
    print(f"streams {streams} blocks {blocks}: {mb:7.1f} MB per call   pageable {res['pageable']:.3f} ms   pinned {res['pinned']:.3f} ms", flush=True)
    eng.close(); pin_in.close(); pin_out.close()
