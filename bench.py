#!/usr/bin/env python3
"""Headline benchmark: output Msamples/s of the 80k-tap overlap-save upsampler.

    python bench.py --gpus N --steps K --warmup W        (N = 1 by default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* = one pass of the hot path over one batch of synthetic PCM that is
already resident in HBM: `mi_engine_process_device` on interleaved s32 frames
(fused kernel: PCM load -> FFT -> spectral multiply -> P inverse FFTs ->
overlap-discard -> PCM store, plus the small history-carry kernel).

Workload at N = 1 is BASELINE.json configs[1]: 44.1k -> 176.4k (4x), stereo,
80 001-tap minimum-phase filter, 2048 blocks per channel per launch. The path shards by
independent streams: with N ranks every rank runs its own stereo
stream(s) of the same size on its own GPU -- no data-path collective -- so the
scaling is *weak*; torch.distributed (gloo) is used only for the rendezvous, the
barriers around the timed region and the max-over-ranks of the elapsed time.

Printed on rank 0 as ONE JSON line; `roofline` prices the fused kernel's
ALGORITHMIC bytes (4B(1+1/L) per channel-block + the filter
half-spectrum once per launch) against the 8 TB/s HBM peak using the kernel's
own hipEvent duration on the launching stream; `cpu_baseline` times the
reference's own C++ (oracle/_ref, built from /root/reference in the build
container) -- or the C restatement when that library is absent -- on one host
core over a bounded sample of the same workload.

`--dry-run` exercises everything except the GPU (rank/stream sharding,
rendezvous, barriers, max-reduce, JSON assembly) with a synthetic per-rank
time; tests/test_distributed.py uses it with world_size 2 on CPU.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CONFIGS = {
    # id: (filter file, streams per GPU, channels, blocks per channel, description)
    # 2048 blocks = 9.9 min of 44.1 kHz stereo per launch: 8 rounds of workgroups on 256 CUs (a one-round
    # launch of 256 blocks is 10-15 % slower: lockstep memory phases, launch gap; profiles/r01_summary.md)
    2: ("filter_44k_4x_80000_min_phase.json", 1, 2, 2048, "44.1k->176.4k 4x stereo, 80k-tap min-phase"),
    3: ("filter_48k_16x_80000_min_phase.json", 1, 8, 256, "48k->768k 16x 8ch, 80k-tap min-phase + EQ"),
    4: ("filter_44k_2x_80000_min_phase.json", 32, 2, 32, "32 stereo streams/GPU, 44.1k 2x 80k-tap"),
    5: ("filter_48k_8x_160000_linear_phase.json", 1, 32, 64, "48k 8x linear 160k-tap, 32ch + EQ"),
}

EQ_PROFILE = ROOT / "tests" / "golden" / "g4_eq_profiles.json"


class Hip:
    """The few HIP runtime calls the bench needs for buffers and sync (ctypes on
    the same libamdhip64 the product library is linked against)."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so.7")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipSetDevice.argtypes = [C.c_int]
        self.lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipStreamSynchronize.argtypes = [C.c_void_p]

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with hipError {rc}")

    def malloc(self, n):
        p = C.c_void_p()
        self.check(self.lib.hipMalloc(C.byref(p), n), "hipMalloc")
        return p.value

    def h2d(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.hipMemcpy(dst, arr.ctypes.data, arr.nbytes, 1), "hipMemcpy H2D")

    def d2h(self, arr, src):
        self.check(self.lib.hipMemcpy(arr.ctypes.data, src, arr.nbytes, 2), "hipMemcpy D2H")

    def sync(self):
        self.check(self.lib.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def stream(self):
        s = C.c_void_p()
        self.check(self.lib.hipStreamCreate(C.byref(s)), "hipStreamCreate")
        return s.value


def stream_ids(rank: int, streams_per_gpu: int) -> list[int]:
    """Global ids of the independent streams rank `rank` owns (static
    partition by stream, nothing shared between ranks but the read-only filter)."""
    return [rank * streams_per_gpu + s for s in range(streams_per_gpu)]


def synth_pcm(config_id: int, stream_id: int, frames: int, channels: int) -> np.ndarray:
    """round(clip(0.2*N(0,1), -1, 1) * 2^31), seed 1000*config + stream."""
    rng = np.random.default_rng(1000 * config_id + stream_id)
    x = np.clip(rng.standard_normal((frames, channels)) * 0.2, -1.0, 1.0)
    return np.clip(np.round(x * 2147483648.0), -2147483648, 2147483647).astype("<i4")


def cpu_baseline(filter_path: Path, budget_s: float) -> dict:
    """Reference C++ (or its C restatement) on ONE host core, one channel of the
    same workload, for about `budget_s` seconds."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O

    h, taps, fft, block, L = O.read_filter(filter_path)
    nin = block // L
    x = (np.random.default_rng(7).standard_normal(nin) * 0.2).astype(np.float32)
    if O.have_ref():
        u = O.RefUpsampler()
        ok, msg = u.load_filter(filter_path)
        if not ok:
            raise RuntimeError(msg)
        kind = "reference"
    else:
        u = O.OracleUpsampler(h, taps, fft, block, L)
        kind = "port"
    u.process_block(x)  # warm caches / page in
    n, t0 = 0, time.perf_counter()
    while True:
        y = u.process_block(x)
        assert y.size == block
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and n >= 16:
            break
    return {"value": round(n * block / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "host_cores_available": os.cpu_count(),
            "sample": f"{n} blocks x 1 channel of the same filter/geometry through ProcessBlock "
                      f"({'oracle/_ref = reference C++ CPU path' if kind == 'reference' else 'oracle C restatement'}, "
                      f"{dt:.1f} s, single thread)"}


def run_gpu(args, ups, cfg, fpath, rank, local_rank, streams, channels, blocks, barrier):
    """Timed region on this rank's GPU. Returns (elapsed_s, kernel stats, engine path)."""
    hip = Hip()
    device = local_rank % ups.device_count()
    hip.check(hip.lib.hipSetDevice(device), "hipSetDevice")
    filt = ups.Filter(fpath, device=device)
    use_eq = args.eq or args.config in (3, 5)
    if use_eq:
        text = json.loads(EQ_PROFILE.read_text())["opra10"]
        filt.set_eq(text, 768000.0 if "48k" in fpath.name else 705600.0)
    eng = ups.Engine(filt, streams, channels, ups.PCM_S32, ups.PCM_S32)
    # synthetic PCM, resident in HBM before anything is timed
    in_stride, out_stride = eng.in_bytes(blocks), eng.out_bytes(blocks)
    d_in = hip.malloc(in_stride * streams)
    d_out = hip.malloc(out_stride * streams)
    for s, sid in enumerate(stream_ids(rank, streams)):
        hip.h2d(d_in + s * in_stride, synth_pcm(args.config, sid, blocks * eng.in_frames, channels))
    stream = hip.stream()
    for _ in range(args.warmup):
        eng.process_device(d_in, d_out, blocks, stream)
    eng.enable_kernel_timing(max(args.steps, 1))
    hip.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.process_device(d_in, d_out, blocks, stream)
    hip.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    kstat = eng.kernel_ms_stats()
    # sanity on the last output (not timed): non-trivial
    tail = np.empty(min(cfg["block_size"] * channels, 65536), dtype="<i4")
    hip.d2h(tail, d_out)
    assert np.abs(tail.astype(np.int64)).max() > 0, "output is all zeros"
    return elapsed, kstat, eng.path, use_eq


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--blocks", type=int, default=0, help="override blocks per channel")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eq", action="store_true", help="fold the 10-band EQ profile into the filter")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: synthetic per-rank time (CPU tests)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} ...`",
                  file=sys.stderr)
            return 2
        args.gpus = world

    # product library first: it brings in the HIP runtime it was built against
    import totton_rasp_gpu_dsp_amd as ups

    if not args.dry_run and ups.device_count() < 1:
        print("bench.py: no HIP device (the upsampler has no CPU path)", file=sys.stderr)
        return 1

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist  # control plane only (gloo): rendezvous, barriers, max-reduce

        # gloo reports its connection state on stdout; keep stdout for the ONE json line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def barrier():
        if dist is not None:
            dist.barrier()

    fname, streams, channels, blocks, desc = CONFIGS[args.config]
    streams = args.streams or streams
    blocks = args.blocks or blocks
    fpath = ROOT / "data" / "coefficients" / fname
    ok, msg, cfg = ups.read_filter(fpath)
    if not ok:
        print(f"bench.py: {msg}", file=sys.stderr)
        return 1
    L, B, N = cfg["upsample_factor"], cfg["block_size"], cfg["fft_size"]

    if args.dry_run:
        barrier()
        elapsed = 0.010 * (rank + 1)  # deterministic, rank-dependent: the max-reduce must pick the last rank
        barrier()
        kstat = {"avg": elapsed * 1e3 / max(args.steps, 1), "min": 0.0, "count": 0}
        path, use_eq = "dry-run", False
    else:
        elapsed, kstat, path, use_eq = run_gpu(args, ups, cfg, fpath, rank, local_rank, streams, channels, blocks, barrier)

    owned = stream_ids(rank, streams)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ids = [None] * world
        dist.all_gather_object(ids, owned)
        all_ids = sorted(i for part in ids for i in part)
    else:
        all_ids = owned
    assert all_ids == list(range(world * streams)), "stream partition must be disjoint and complete"

    units_rank = blocks * streams * channels                 # channel-blocks per launch on this GPU
    samples_step = units_rank * B * world                    # whole job, all ranks
    value = samples_step * args.steps / elapsed / 1e6
    bytes_unit = 4.0 * B * (1.0 + 1.0 / L)
    bytes_launch = units_rank * bytes_unit + 8.0 * (N // 2 + 1)
    achieved = bytes_launch / (kstat["avg"] * 1e-3) / 1e9
    result = {
        "metric": "output Msamples/s, 80k-tap FIR upsample (overlap-save), interleaved s32 PCM in HBM",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"configs[{args.config - 1}]: {desc}", "filter": fname, "taps": cfg["taps"],
                   "fft_size": N, "block_size": B, "upsample_factor": L, "streams_per_gpu": streams,
                   "channels": channels, "blocks_per_channel": blocks, "pcm": "s32 interleaved in/out",
                   "eq": bool(use_eq), "kernel_path": path, "streams_total": world * streams,
                   "parallelism": f"streams sharded over {world} GPU(s), no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                     "kernel_ms_avg": round(kstat["avg"], 5), "kernel_ms_min": round(kstat["min"], 5),
                     "kernel_launches_timed": kstat["count"],
                     "algorithmic_bytes_per_launch": int(bytes_launch),
                     "note": "bytes = units*4B(1+1/L) + 8(N/2+1); duration = hipEvent pair around the "
                             f"{path} kernel(s) on the launching stream, rank 0"},
    }
    # HBM-side bytes per launch from the committed PMC passes (profiles/traffic.json), when
    # this run is the workload those passes measured
    try:
        rec = json.loads((ROOT / "profiles" / "traffic.json").read_text()).get(str(args.config))
    except (OSError, ValueError):
        rec = None
    if rec and (rec["streams"], rec["channels"], rec["blocks"]) == (streams, channels, blocks) and not args.dry_run:
        result["roofline"]["traffic"] = int(rec["bytes"])
        result["roofline"]["traffic_source"] = rec["source"] + " (separate rocprofv3 --pmc passes; 2*FETCH_SIZE + WRITE_SIZE)"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            result["cpu_baseline"] = cpu_baseline(fpath, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
