#!/usr/bin/env python3
"""Headline benchmark: output Msamples/s of the 80k-tap overlap-save upsampler.

    python bench.py --gpus N --steps K --warmup W        (N = 1 by default)

With N > 1 and no launcher the script starts N fresh child processes itself (one rank per GPU, before anything
touches HIP in the parent); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it
uses the ranks the launcher made (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment). Either way a run with
more ranks than visible devices fails loudly.

A *step* = one pass of the hot path over one batch of synthetic PCM that is already resident in HBM:
`mi_engine_process_device` on interleaved s32 frames (PCM load -> FFT -> spectral multiply -> P inverse FFTs ->
overlap-discard -> PCM store, plus the small history-carry kernel).

Workload at N = 1 is BASELINE.json configs[1]: 44.1k -> 176.4k (4x), stereo, 80 001-tap minimum-phase filter, 256 blocks
per channel per launch as BASELINE.md section 4 states it. The path shards by independent streams: with N
ranks every rank runs its own stream(s) of the same size on its own GPU -- no data-path collective -- so the scaling is
*weak* (`--split channels`: the channels of the SAME streams in N contiguous groups, *strong* scaling -- BASELINE
configs[4]'s 32-channel sweep). The control plane (rendezvous, the barriers around the timed region, max-over-ranks of the
elapsed and kernel times) is a standard-library Unix-socket hub on rank 0 (class Control): torch is not imported, so one
HIP runtime is mapped per process (`hip_runtimes_mapped` in the line).

ONE JSON line on rank 0. Besides the contract's fields:
  roofline      ALGORITHMIC bytes (4B(1+1/L) per channel-block + the filter half-spectrum once per launch)
                over the kernels' own hipEvent duration on the launching stream, against the 8 TB/s HBM peak; `traffic`
                = HBM-side bytes per launch from the committed PMC passes (profiles/traffic.json)
  variants      the same workload at 2048 blocks per launch, and a ~2 s sustained run of the headline launch
  configs       BASELINE configs[1..4] (ids 2-5), each with value / ms / kernel ms / roofline (N = 1 only)
  end_to_end    PCIe-inclusive rate of mi_engine_process_host on pinned host buffers (H2D, kernels, D2H overlapped);
                end_to_end_by_output_format: the same with s16 / packed-s24 output; process_block_latency: p50 / p99 of the
                reference-shaped one-channel-block call (mi_ups_process_block)
  roofline.copy_ceiling_GBps   measured device-to-device copy rate of this box, and the fraction against it
  cpu_baseline  the reference's own C++ (oracle/_ref) -- or the C restatement when that library is absent -- on one
                core AND on all cores of this host (one channel per process), plus the config-1 scipy.signal.fftconvolve
                leg (stereo 8192-frame block, zero-stuffed 2x, 80 001 taps)

`--dry-run` exercises everything except the GPU (rank/stream sharding, rendezvous, barriers, max-reduce, JSON
assembly) with a synthetic per-rank time; tests/test_distributed.py uses it with world_size 2 on CPU.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_VECTOR_PEAK_TF = 157.3  # MI355X_MICROARCH.md: 64 FLOP/clk/SIMD, i.e. packed v_pk_fma_f32 on every issue slot
FP32_SCALAR_ISSUE_PEAK_TF = FP32_VECTOR_PEAK_TF / 2  # one non-packed v_fma_f32 per issue slot: what scalar fp32 code can reach

CONFIGS = {
    # id: (filter file, streams per GPU, channels, blocks per channel, description)   [BASELINE.md section 4]
    2: ("filter_44k_4x_80000_min_phase.json", 1, 2, 256, "44.1k->176.4k 4x stereo, 80k-tap min-phase"),
    3: ("filter_48k_16x_80000_min_phase.json", 1, 8, 256, "48k->768k 16x 8ch, 80k-tap min-phase + EQ"),
    4: ("filter_44k_2x_80000_min_phase.json", 32, 2, 32, "32 stereo streams/GPU, 44.1k 2x 80k-tap"),
    5: ("filter_48k_8x_160000_linear_phase.json", 1, 32, 64, "48k 8x linear 160k-tap, 32ch + EQ"),
}
LONG_BLOCKS = 2048  # the headline workload as 8 rounds of workgroups per launch (9.9 min of audio)

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

    def free(self, p):
        self.check(self.lib.hipFree(C.c_void_p(p)), "hipFree")

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


def algorithmic_bytes(cfg: dict, units: int) -> float:
    """units * 4B(1 + 1/L) + the filter half-spectrum 8(N/2+1) once per launch."""
    return units * 4.0 * cfg["block_size"] * (1.0 + 1.0 / cfg["upsample_factor"]) + 8.0 * (cfg["fft_size"] // 2 + 1)


# ------------------------------------------------------------------------------------------------ CPU baselines --
def model_flops(cfg: dict, units: int) -> dict:
    """The arithmetic one launch does by the textbook count (what the path computes, whatever the kernel's instruction mix):
    per channel-block (1 + P) complex transforms of K = N/(2P) points at 5 K log2 K, plus the spectral stage on K/2 mirror
    pairs -- the real-FFT untangle once (14 flops: two complex adds, one complex multiply) and per phase two complex
    products, two sums, the conj(W) multiply and the re-tangle (26 flops). P = L when L | N, else 1."""
    n, L = cfg["fft_size"], max(1, cfg["upsample_factor"])
    P = L if n % L == 0 else 1
    K = n // P // 2
    fft = (1 + P) * 5.0 * K * np.log2(K)
    spectral = (K / 2.0) * (14.0 + 26.0 * P)
    return {"P": P, "K": K, "transform": units * fft, "spectral": units * spectral, "total": units * (fft + spectral)}


def host_cores() -> int:
    """Cores this process may really use: the affinity mask, cut by a cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_worker(filter_path: str, budget_s: float, seed: int):
    """One core, one channel of the headline workload through ProcessBlock for `budget_s` seconds.
    Returns (blocks, seconds, kind). Top-level so that multiprocessing (spawn) can import it."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O

    h, taps, fft, block, L = O.read_filter(filter_path)
    nin = block // L
    x = (np.random.default_rng(seed).standard_normal(nin) * 0.2).astype(np.float32)
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
        if dt >= budget_s and n >= 8:
            break
    return n, dt, kind, block


def cpu_baseline(filter_path: Path, budget_s: float) -> dict:
    """(1) one core, (2) all cores -- one channel per process, nothing shared --, (3) BASELINE configs[0]: the
    scipy.signal.fftconvolve plumbing leg. About budget_s seconds for each of (1) and (2)."""
    import multiprocessing as mp

    n1, dt1, kind, block = _cpu_worker(str(filter_path), budget_s, 7)
    one = n1 * block / dt1 / 1e6
    cores = host_cores()
    what = "oracle/_ref = reference C++ CPU path" if kind == "reference" else "oracle C restatement"
    res = {"value": round(one, 4), "unit": "Msamples/s", "cores": 1, "kind": kind,
           "sample": f"{n1} blocks x 1 channel of the headline filter/geometry through ProcessBlock ({what}, {dt1:.1f} s, "
                     "single thread)",
           "host_cores_available": cores, "host_cpu_count": os.cpu_count()}
    if cores > 1:
        ctx = mp.get_context("spawn")  # fresh interpreters: nothing of this process (HIP included) is inherited
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            parts = pool.starmap(_cpu_worker, [(str(filter_path), budget_s, 100 + i) for i in range(cores)])
        wall = time.perf_counter() - t0
        rate = sum(n * b / dt for n, dt, _, b in parts) / 1e6
        res["all_cores"] = {"value": round(rate, 3), "unit": "Msamples/s", "cores": cores, "kind": kind,
                            "sample": f"{sum(p[0] for p in parts)} blocks over {cores} processes, one channel each, "
                                      f"{budget_s:.0f} s of ProcessBlock per process ({wall:.1f} s wall incl. start-up)"}
    try:
        from scipy.signal import fftconvolve

        sys.path.insert(0, str(ROOT / "oracle"))
        import oracle as O

        p2 = ROOT / "data" / "coefficients" / "filter_44k_2x_80000_min_phase.json"
        h2 = O.read_filter(p2)[0]
        x = (np.random.default_rng(1000).standard_normal((8192, 2)) * 0.2)
        leg = {}
        for name, dt_ in (("f32", np.float32), ("f64", np.float64)):
            up = np.zeros((16384, 2), dt_)
            up[::2] = x.astype(dt_)
            hh = h2.astype(dt_)
            fftconvolve(up[:, 0], hh)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 1.0 or n < 5:
                for c in range(2):
                    y = fftconvolve(up[:, c], hh)
                n += 1
            dt = (time.perf_counter() - t0) / n
            leg[name] = {"ms_per_block": round(dt * 1e3, 3), "Msamples_per_s": round(16384 * 2 / dt / 1e6, 3)}
        res["fftconvolve_config1"] = {"workload": "configs[0]: 44.1k->88.2k 2x stereo, one 8192-frame block zero-stuffed 2x, "
                                                  "80 001 taps, scipy.signal.fftconvolve per channel, 1 thread", **leg}
    except ImportError:
        res["fftconvolve_config1"] = "scipy unavailable on this host"
    return res


# ------------------------------------------------------------------------------------------------------- GPU runs --
class Workload:
    """One config resident on this rank's GPU: filter (+EQ), engine, synthetic PCM in HBM."""

    def __init__(self, ups, hip, device, config_id, rank, streams=None, blocks=None, eq=None, channel_slice=None, custom=None):
        """channel_slice = (first, count): this GPU takes that contiguous channel group of every stream (--split channels);
        the synthetic frames are the full-width ones, cut -- every channel sees the samples it sees on one GPU.
        custom = (filter json path, streams, channels, blocks, description): a workload outside CONFIGS (config_id then only
        seeds the PCM)."""
        if custom:
            fpath, s, channels, b, desc = custom
            fname = Path(fpath).name
        else:
            fname, s, channels, b, desc = CONFIGS[config_id]
            fpath = ROOT / "data" / "coefficients" / fname
        self.config_id, self.desc, self.fname = config_id, desc, fname
        c0, nch = channel_slice or (0, channels)
        self.full_channels = channels
        self.streams, self.channels, self.blocks = streams or s, nch, blocks or b
        self.fpath = Path(fpath)
        ok, msg, cfg = ups.read_filter(self.fpath)
        if not ok:
            raise RuntimeError(msg)
        self.cfg = cfg
        self.hip = hip
        self.filt = ups.Filter(self.fpath, device=device)
        self.use_eq = (config_id in (3, 5)) if eq is None else eq
        if self.use_eq:
            self.eq_text = json.loads(EQ_PROFILE.read_text())["opra10"]
            self.eq_fs = 768000.0 if "48k" in fname else 705600.0
            self.filt.set_eq(self.eq_text, self.eq_fs)
        self.eng = ups.Engine(self.filt, self.streams, nch, ups.PCM_S32, ups.PCM_S32)
        self.in_stride, self.out_stride = self.eng.in_bytes(self.blocks), self.eng.out_bytes(self.blocks)
        self.d_in = hip.malloc(self.in_stride * self.streams)
        self.d_out = hip.malloc(self.out_stride * self.streams)
        self.host_pcm = []  # kept for check_output (the probe blocks' fp64 truth)
        for i, sid in enumerate(stream_ids(rank, self.streams)):
            full = synth_pcm(config_id, sid, self.blocks * self.eng.in_frames, channels)
            self.host_pcm.append(np.ascontiguousarray(full[:, c0:c0 + nch]))
            hip.h2d(self.d_in + i * self.in_stride, self.host_pcm[-1])
        self.stream = hip.stream()

    @property
    def units(self):
        return self.blocks * self.streams * self.channels

    def run(self, steps, warmup, barrier=lambda: None, min_seconds=0.0):
        """W untimed steps, then K (or at least min_seconds of) timed steps. Returns (steps done, elapsed s, kernel stats)."""
        for _ in range(warmup):
            self.eng.process_device(self.d_in, self.d_out, self.blocks, self.stream)
        # an event pair costs about 8 us of stream time (profiles/r03_n_step_overhead.txt: 131.3 us per headline step with a
        # pair on every launch, 122.9 without any): the timed region samples every 4th launch, never fewer than 5 of them
        self.timing_every = max(1, min(4, steps // 5))
        self.eng.enable_kernel_timing(64 if min_seconds else max(steps, 1), self.timing_every)
        self.hip.sync()
        barrier()
        t0 = time.perf_counter()
        done = 0
        while True:
            for _ in range(steps):
                self.eng.process_device(self.d_in, self.d_out, self.blocks, self.stream)
            done += steps
            if not min_seconds:
                break
            self.hip.check(self.hip.lib.hipStreamSynchronize(C.c_void_p(self.stream)), "hipStreamSynchronize")
            if time.perf_counter() - t0 >= min_seconds:
                break
        self.hip.sync()
        elapsed = time.perf_counter() - t0
        barrier()
        return done, elapsed, self.eng.kernel_ms_stats()

    def prime(self, seconds):
        """Untimed: the same call back to back for `seconds`. A device that has been idle runs its first milliseconds at lower
        clocks (variants.sustained vs variants.cold_start); the K timed steps that follow are the contract's."""
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                self.eng.process_device(self.d_in, self.d_out, self.blocks, self.stream)
            n += 8
            self.hip.check(self.hip.lib.hipStreamSynchronize(C.c_void_p(self.stream)), "hipStreamSynchronize")
        return n

    def per_kernel_ms(self, steps=3):
        """Untimed extra steps with one event pair per launch (mi_engine_last_class_ms): the split of a call over
        planarize / transform / frame assembly / history carry. Outside the timed region because the extra event records
        perturb the call; kernels that overlap on two streams (pipelined launches) add up to more than the call."""
        self.eng.enable_class_timing(True)
        acc = {}
        for _ in range(steps):
            self.eng.process_device(self.d_in, self.d_out, self.blocks, self.stream)
            for k, v in self.eng.last_class_ms().items():
                if v is not None:
                    acc.setdefault(k, []).append(v)
        self.eng.enable_class_timing(False)
        return {k: round(sum(v) / len(v), 5) for k, v in acc.items()}

    def eq_folded_fir(self, h: np.ndarray) -> np.ndarray:
        """The EQ folded into the FIR as include/mi_upsampler.h defines it, restated with scipy (nothing from oracle/):
        the cascade's recursion over the taps in fp64, cut to `taps` samples with a closing half-Hann over the last
        (taps-1)//64. Biquad coefficients from the library's host helper (pinned to the compiled reference at 1e-14 in
        tests/test_host_logic.py). The stream is then ONE linear convolution with this FIR: no block-periodic wrap."""
        from scipy.signal import lfilter

        import totton_rasp_gpu_dsp_amd as ups

        preamp, bands = ups.eq_parse(self.eq_text)
        y = h * (10.0 ** (preamp / 20.0) if preamp != 0.0 else 1.0)
        for en, typ, freq, gain, q in bands[:, :5]:  # rows: enabled, type, frequency, gain, q, ...
            if en:
                c = ups.eq_biquad(True, int(typ), freq, gain, q, self.eq_fs)
                y = lfilter(c[:3], [1.0, c[3], c[4]], y)
        W = (h.size - 1) // 64
        if W:
            y[h.size - W:] *= 0.5 * (1.0 + np.cos(np.pi * (np.arange(W) + 0.5) / W))
        return y

    def truth_block(self, stream_slot: int, channel: int, blk: int, x_stream: np.ndarray) -> np.ndarray:
        """fp64 statement of one output block of one channel (numpy only, nothing from oracle/): the reference's
        N-point overlap-save, Y = FFT_N([history | zero-stuffed input]) * FFT_N(h), y = Re IFFT_N(Y), keep the
        last B; with an EQ, h is the EQ-folded FIR of the same length (eq_folded_fir), so this equals the true streaming
        convolution (vulkan_streaming_upsampler.cpp:528-569). `x_stream` = this stream's PCM frames as float [frames][ch].
        The history of block `blk` is the O/L frames before it (the engine carries it; blocks >= 2 lie inside the call)."""
        n, B, L = self.cfg["fft_size"], self.cfg["block_size"], self.cfg["upsample_factor"]
        O, nin = n - B, B // L
        h = np.fromfile(self.cfg["coefficients_path"], "<f4").astype(np.float64)
        if self.use_eq:
            h = self.eq_folded_fir(h)
        H = np.fft.rfft(np.concatenate([h, np.zeros(n - h.size)]))
        t = np.zeros(n)
        first = blk * nin - O // L
        assert first >= 0, "probe block must lie far enough inside the call"
        t[: O + B : L] = x_stream[first:(blk + 1) * nin, channel]
        return np.fft.irfft(np.fft.rfft(t) * H, n)[O:]

    def check_output(self, probes=None):
        """Probe blocks of the output buffer against fp64 truth: 1 LSB + 1e-5 * max|y| (the parity bar of tests/), after the
        same clamp the PCM store applies. Default probes: last block of (first stream, first channel) and of (last stream,
        last channel), and a middle block of the last stream's first channel."""
        B, L = self.cfg["block_size"], self.cfg["upsample_factor"]
        nin = B // L
        probes = probes or [(0, 0, self.blocks - 1), (self.streams - 1, self.channels - 1, self.blocks - 1),
                            (self.streams - 1, 0, max(2, self.blocks // 2))]
        worst = 0.0
        for slot, ch, blk in sorted(set(probes)):
            got = np.empty(B * self.channels, dtype="<i4")
            self.hip.d2h(got, self.d_out + slot * self.out_stride + blk * B * self.channels * 4)
            y = got.reshape(B, self.channels)[:, ch].astype(np.float64) / 2147483648.0
            x = self.host_pcm[slot].astype(np.float64) / 2147483648.0
            want = np.clip(self.truth_block(slot, ch, blk, x), -1.0, float(np.float32(0.9999999)))
            err = float(np.abs(y - want).max())
            tol = 2.0 ** -31 + 1e-5 * float(np.abs(want).max())
            assert np.abs(want).max() > 1e-3, "truth is silence: bad probe"
            assert err <= tol, (f"config {self.config_id} stream {slot} channel {ch} block {blk}: max|d| {err:.3e} > {tol:.3e} "
                                "against fp64 truth")
            worst = max(worst, err / tol)
        return {"probes": len(set(probes)), "worst_err_over_tol": round(worst, 4),
                "bar": "1 LSB + 1e-5*max|y| vs fp64 overlap-save of the same PCM (numpy), after the PCM clamp"}

    def close(self):
        self.hip.sync()
        self.eng.close()
        self.filt.close()
        self.hip.free(self.d_in)
        self.hip.free(self.d_out)


def summary(w: Workload, steps, elapsed, kstat, world=1, traffic=None) -> dict:
    samples = w.units * w.cfg["block_size"] * world * steps
    bytes_launch = algorithmic_bytes(w.cfg, w.units)
    achieved = bytes_launch / (kstat["avg"] * 1e-3) / 1e9 if kstat["avg"] > 0 else 0.0
    roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel_ms_avg": round(kstat["avg"], 5),
            "kernel_ms_min": round(kstat["min"], 5), "kernel_launches_timed": kstat["count"],
            "timed_every_nth_launch": getattr(w, "timing_every", 1),
            "algorithmic_bytes_per_launch": int(bytes_launch)}
    # the roof that actually binds: fp32 vector arithmetic. Model flops per launch over the same kernel time.
    mf = model_flops(w.cfg, w.units)
    tf = mf["total"] / (kstat["avg"] * 1e-3) / 1e12 if kstat["avg"] > 0 else 0.0
    roof["compute"] = {"bound": "fp32 vector (VALU)", "model_flops_per_launch": int(mf["total"]),
                       "model": "units*((1+P)*5*K*log2(K) + (K/2)*(14+26*P)), K = N/(2P) = %d, P = %d" % (mf["K"], mf["P"]),
                       "achieved": round(tf, 2), "unit": "TFLOP/s", "peak": FP32_VECTOR_PEAK_TF,
                       "frac": round(tf / FP32_VECTOR_PEAK_TF, 4),
                       "frac_of_scalar_issue_peak": round(tf / FP32_SCALAR_ISSUE_PEAK_TF, 4),
                       "note": "157.3 TF needs packed v_pk_fma_f32 on every issue slot; scalar fp32 code (what measured fastest "
                               "here, profiles/r02_b_packed_math.txt) tops out at half of it, and an FFT butterfly is ~40 % FMA"}
    key = ((str(w.config_id) if w.blocks == CONFIGS[w.config_id][3] else f"{w.config_id}_{w.blocks}blocks")
           if w.config_id in CONFIGS else getattr(w, "traffic_key", ""))
    rec = (traffic or {}).get(key)
    if rec and (rec["streams"], rec["channels"], rec["blocks"]) == (w.streams, w.channels, w.blocks):
        roof["traffic"] = int(rec["bytes"])
        roof["traffic_over_algorithmic"] = round(rec["bytes"] / bytes_launch, 3)
        roof["traffic_source"] = rec["source"] + " (separate rocprofv3 --pmc passes; 2*FETCH_SIZE + WRITE_SIZE)"
        if "issue" in rec:
            # what bounds the kernel when HBM does not: a lone wave issues one VALU instruction per 4.94 clocks
            # (profiles/r02_a_ubench_valu_lds_rates.txt), so this share of every wave's lifetime is VALU issue alone
            iv = dict(rec["issue"])
            iv["valu_issue_share_of_wave_clocks"] = round(iv["valu_insts_per_wave"] * 4.94 / iv["wave_clocks"], 3)
            roof["compute"]["valu_busy_share"] = iv["valu_issue_share_of_wave_clocks"]
            iv["what"] = ("SQ counters of the same kernel, per launch (separate --pmc passes): the transform is VALU-issue / "
                          "LDS-write bound with its memory phases not overlapped (one workgroup owns the CU), not HBM-bound")
            roof["issue_view"] = iv
    return {"value": round(samples / elapsed / 1e6, 3), "ms_per_step": round(elapsed / steps * 1e3, 5), "roofline": roof}


def config_block(w: Workload) -> dict:
    name = f"configs[{w.config_id - 1}]: {w.desc}" if w.config_id in CONFIGS else w.desc
    return {"workload": name, "filter": w.fname, "taps": w.cfg["taps"],
            "fft_size": w.cfg["fft_size"], "block_size": w.cfg["block_size"], "upsample_factor": w.cfg["upsample_factor"],
            "streams_per_gpu": w.streams, "channels": w.channels, "blocks_per_channel": w.blocks,
            "pcm": "s32 interleaved in/out", "eq": bool(w.use_eq), "kernel_path": w.eng.path,
            "coop_frames": bool(w.eng.last_coop_frames)}


def end_to_end(ups, w: Workload, seconds=2.0) -> dict:
    """mi_engine_process_host on PINNED host buffers: sub-batches of the call overlap H2D, kernels and D2H."""
    pin_in = ups.PinnedBuffer(w.in_stride * w.streams)
    pin_out = ups.PinnedBuffer(w.out_stride * w.streams)
    x = np.concatenate([synth_pcm(w.config_id, sid, w.blocks * w.eng.in_frames, w.channels).reshape(-1)
                        for sid in range(w.streams)])
    pin_in.array[:] = x.view(np.uint8)
    for _ in range(2):
        w.eng.process_host(pin_in.array, w.blocks, out=pin_out.array)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or n < 3:
        w.eng.process_host(pin_in.array, w.blocks, out=pin_out.array)
        n += 1
    dt = (time.perf_counter() - t0) / n
    host_bytes = pin_in.array.nbytes + pin_out.array.nbytes
    res = {"value": round(w.units * w.cfg["block_size"] / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_call": round(dt * 1e3, 4),
           "host_link_GB_per_s": round(host_bytes / dt / 1e9, 2), "host_bytes_per_call": int(host_bytes), "calls_timed": n,
           "boundary": "mi_engine_process_host, pinned host buffers (mi_host_alloc), sub-batches pipelined over three "
                       "HIP streams with double-buffered device staging",
           "workload": f"configs[{w.config_id - 1}] x {w.blocks} blocks/channel per call"}
    pin_in.close()
    pin_out.close()
    return res


def end_to_end_format(ups, w: Workload, fmt: str, seconds=1.0) -> dict:
    """The same host-buffer call with s32 in and a narrower PCM format out (s16 / packed s24): the D2H side carries 2 or 3
    bytes per sample instead of 4, and the host link is what bounds the end-to-end rate."""
    out_fmt = ups.PCM_NAMES[fmt]
    eng = ups.Engine(w.filt, w.streams, w.channels, ups.PCM_S32, out_fmt)
    pin_in = ups.PinnedBuffer(eng.in_bytes(w.blocks) * w.streams)
    pin_out = ups.PinnedBuffer(eng.out_bytes(w.blocks) * w.streams)
    pin_in.array[:] = np.concatenate([x.reshape(-1) for x in w.host_pcm]).view(np.uint8)
    for _ in range(2):
        eng.process_host(pin_in.array, w.blocks, out=pin_out.array)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or n < 3:
        eng.process_host(pin_in.array, w.blocks, out=pin_out.array)
        n += 1
    dt = (time.perf_counter() - t0) / n
    host_bytes = pin_in.array.nbytes + pin_out.array.nbytes
    res = {"value": round(w.units * w.cfg["block_size"] / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_call": round(dt * 1e3, 4),
           "host_link_GB_per_s": round(host_bytes / dt / 1e9, 2), "host_bytes_per_call": int(host_bytes), "pcm": f"s32 in, {fmt} out"}
    pin_in.close()
    pin_out.close()
    eng.close()
    return res


def process_block_latency(ups, device, filter_path, calls=1000) -> dict:
    """The reference's own call shape (vulkan_streaming_upsampler.h:33, callers alsa_streamer_main.cpp:321,543): ONE block
    of ONE channel per call through mi_ups_process_block -- host float in, H2D, kernels, D2H, host float out, blocking."""
    u = ups.StreamingUpsampler(device)
    ok, msg = u.load_filter(filter_path)
    if not ok:
        raise RuntimeError(msg)
    cfg = u.config
    nin = cfg["block_size"] // cfg["upsample_factor"]
    x = (np.random.default_rng(5).standard_normal(nin) * 0.2).astype(np.float32)
    out = np.empty(cfg["block_size"], dtype=np.float32)
    fx, fo = x.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float))
    for _ in range(20):
        assert ups.lib.mi_ups_process_block(u._h, fx, nin, fo, out.size) == out.size
    t = np.empty(calls)
    for i in range(calls):
        t0 = time.perf_counter()
        ups.lib.mi_ups_process_block(u._h, fx, nin, fo, out.size)
        t[i] = time.perf_counter() - t0
    u.close()
    ms = np.sort(t) * 1e3
    return {"calls": calls, "p50_ms": round(float(ms[calls // 2]), 4), "p99_ms": round(float(ms[int(calls * 0.99)]), 4),
            "mean_ms": round(float(ms.mean()), 4), "Msamples_per_s_at_p50": round(cfg["block_size"] / ms[calls // 2] / 1e3, 2),
            "block_audio_ms_at_output_rate": round(cfg["block_size"] / (44100.0 * cfg["upsample_factor"]) * 1e3, 2),
            "what": "mi_ups_process_block, one channel-block per call (host float in/out, blocking), headline filter"}


def rows_8x_80k(ups, hip, device, args, ceiling, traffic, only=None) -> list:
    """north_star: "Msamples/s for 80 k-tap 2x/4x/8x/16x". The 8x 80k-tap filter (K = 8192: 64 KiB of LDS, TWO workgroups
    resident per CU) at the headline's stereo shape and at configs[4]'s 32-channel shape. Besides completing the ratio list
    these rows are the evidence for DESIGN 11.1: what a second resident workgroup buys over K = 16384 (one per CU)."""
    rows = []
    for cid, channels, blocks in ((8, 2, 256), (9, 32, 64)):
        key = f"8x80k_{channels}ch"
        if only and only != key:
            continue
        wl = Workload(ups, hip, device, cid, 0, custom=(ROOT / "data" / "coefficients" / "filter_48k_8x_80000_min_phase.json", 1,
                                                       channels, blocks, f"48k->384k 8x {channels}ch, 80k-tap min-phase (K = 8192)"))
        wl.traffic_key = key
        if args.prime_seconds > 0:
            wl.prime(args.prime_seconds)
        _, el, ks = wl.run(args.steps, args.warmup)
        sm = summary(wl, args.steps, el, ks, 1, traffic)
        if ceiling:
            sm["roofline"]["frac_of_copy_ceiling"] = round(sm["roofline"]["achieved"] / ceiling, 5)
        pk = wl.per_kernel_ms()
        row = {"config": config_block(wl), "value": sm["value"], "unit": "Msamples/s", "ms_per_step": sm["ms_per_step"],
               "roofline": sm["roofline"], "output_check": wl.check_output(), "per_kernel_ms": pk}
        if pk.get("transform"):
            row["transform_only_frac"] = round(algorithmic_bytes(wl.cfg, wl.units) / (pk["transform"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        rows.append(row)
        wl.close()
    return rows


def filters_2m(ups, hip, device, args, ceiling, traffic=None, only=None) -> list:
    import tempfile

    sys.path.insert(0, str(ROOT / "totton-rasp-gpu-dsp_amd"))
    import filter_design as fd

    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for cid, ratio in ((6, 8), (7, 2)):
            h = fd.design(640_000, ratio, "48k", "linear")
            path = fd.export(h, Path(tmp), fd.base_name("48k", ratio, 640_000, "linear"), ratio)
            wl = Workload(ups, hip, device, cid, 0, custom=(path, 1, 8, 16, f"48k {ratio}x linear 640k-tap ('2m'), 8ch"))
            wl.traffic_key = f"2m_{ratio}x"
            if only and only != wl.traffic_key:
                wl.close()
                continue
            if args.prime_seconds > 0:
                wl.prime(args.prime_seconds)
            _, el, ks = wl.run(max(5, args.steps // 4), 2)
            sm = summary(wl, max(5, args.steps // 4), el, ks, 1, traffic)
            if ceiling:
                sm["roofline"]["frac_of_copy_ceiling"] = round(sm["roofline"]["achieved"] / ceiling, 5)
            rows.append({"config": config_block(wl), "value": sm["value"], "unit": "Msamples/s", "ms_per_step": sm["ms_per_step"],
                         "roofline": sm["roofline"], "two_level": bool(wl.eng.last_two_level), "output_check": wl.check_output(),
                         "per_kernel_ms": wl.per_kernel_ms()})
            wl.close()
    return rows


def copy_ceiling(ups, device) -> dict:
    return {"read_plus_write_GBps": ups.device_copy_rate(device, 1 << 30, 5)}


def hip_runtimes_mapped() -> list:
    """Distinct libamdhip64 images in this process (must be exactly one: the one the product library links)."""
    seen = set()
    try:
        for line in Path("/proc/self/maps").read_text().splitlines():
            if "libamdhip64" in line:
                seen.add(line.split()[-1])
    except OSError:
        pass
    return sorted(seen)


# ----------------------------------------------------------------------------------------------------- launching --
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Control:
    """Control plane of an N-rank run on ONE node: rendezvous, barriers, max / sum reductions and an all-gather of small
    JSON objects. Rank 0 is the hub on a Unix-domain socket whose path is derived from MASTER_ADDR/MASTER_PORT (the same
    variables torchrun sets; the port itself is never bound here, so a launcher's own store on it is not disturbed).
    Standard library only: no second HIP runtime enters the process (torch ships its own ROCm libraries; round 2's
    gloo control plane needed an import-order rule because of it). Every wait has a deadline: a rank that dies takes the
    others down with an error instead of leaving them in a rendezvous."""

    def __init__(self, rank: int, world: int, timeout: float = 120.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        self.peers, self.sock = [], None
        if world == 1:
            return
        key = f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}_{os.environ.get('MASTER_PORT', '0')}_{world}"
        self.path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"miups_bench_{key}.sock".replace("/", "_"))
        deadline = time.monotonic() + timeout
        if rank == 0:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.path)
            srv.listen(world)
            srv.settimeout(timeout)
            by_rank = {}
            try:
                while len(by_rank) < world - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout)
                    hello = self._recv(conn)
                    if hello.get("world") != world or not 0 < hello.get("rank", 0) < world or hello["rank"] in by_rank:
                        conn.close()
                        raise RuntimeError(f"control plane: unexpected peer {hello}")
                    by_rank[hello["rank"]] = conn
            finally:
                srv.close()
                try:
                    os.unlink(self.path)
                except FileNotFoundError:
                    pass
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise RuntimeError(f"control plane: rank 0 did not open {self.path} within {timeout:.0f} s")
                    time.sleep(0.05)
            s.settimeout(timeout)
            self.sock = s
            self._send(s, {"rank": rank, "world": world})

    @staticmethod
    def _send(sock, obj):
        data = json.dumps(obj).encode()
        sock.sendall(len(data).to_bytes(8, "little") + data)

    @staticmethod
    def _recv(sock):
        def exact(n):
            buf = b""
            while len(buf) < n:
                part = sock.recv(n - len(buf))
                if not part:
                    raise RuntimeError("control plane: a peer closed its connection (did a rank fail?)")
                buf += part
            return buf
        return json.loads(exact(int.from_bytes(exact(8), "little")).decode())

    def all_gather(self, obj) -> list:
        """Everyone contributes one JSON-serialisable object and gets the list of all of them, in rank order."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            parts = [obj] + [self._recv(c) for c in self.peers]
            for c in self.peers:
                self._send(c, parts)
            return parts
        self._send(self.sock, obj)
        return self._recv(self.sock)

    def barrier(self):
        self.all_gather(None)

    def reduce_max(self, v: float) -> float:
        return max(self.all_gather(float(v)))

    def reduce_sum(self, v: float) -> float:
        return sum(self.all_gather(float(v)))

    def close(self):
        for c in self.peers + ([self.sock] if self.sock else []):
            try:
                c.close()
            except OSError:
                pass


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher: N children, one rank each, rendezvous on 127.0.0.1. The parent
    touches no HIP; rank 0's stdout (the ONE json line) is the parent's. The first child that fails ends the others:
    nobody is left waiting in a rendezvous or a barrier."""
    port = free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code)
                for q in live:  # exact PIDs of our own children
                    q.terminate()
    return rc


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="headline workload (default: configs[1])")
    ap.add_argument("--blocks", type=int, default=0, help="override blocks per channel")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--split", choices=("streams", "channels"), default="streams",
                    help="how N > 1 GPUs share the work: independent streams per GPU (weak scaling, default) or contiguous "
                         "channel groups of the SAME streams (strong scaling: BASELINE configs[4] '32-channel, 1->8 GPU sweep')")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no variants / configs / end_to_end")
    ap.add_argument("--prime-seconds", type=float, default=0.3,
                    help="untimed run of the same call before the W warmup steps, so that the K timed steps see the device at "
                         "its running clocks (0 = off; variants.cold_start is the same K steps without it)")
    ap.add_argument("--eq", action="store_true", help="fold the 10-band EQ profile into the headline filter")
    ap.add_argument("--row", default="", help="run ONE extra row instead of a config (for rocprofv3 passes): 2m_8x, 2m_2x, "
                                              "8x80k_2ch, 8x80k_32ch")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: synthetic per-rank time (CPU tests)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)  # nothing below has run yet in this process: no HIP
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2

    import totton_rasp_gpu_dsp_amd as ups

    if not args.dry_run:
        # BENCH_VISIBLE_DEVICES_FOR_TEST: CPU test hook for the over-subscription check only (tests/test_distributed.py)
        real_ndev = ups.device_count()
        ndev = int(os.environ.get("BENCH_VISIBLE_DEVICES_FOR_TEST") or real_ndev)
        if ndev < 1:
            print("bench.py: no HIP device (the upsampler has no CPU path)", file=sys.stderr)
            return 1
        if world > ndev:
            # every rank sees the same count and stops here, before the rendezvous: nobody waits for a peer
            print(f"bench.py: {world} ranks requested but only {ndev} HIP device(s) are visible; ranks are never "
                  "stacked on one GPU", file=sys.stderr)
            return 3
    if args.row:
        # one extra row alone (what the rocprofv3 kernel-trace / PMC passes of scripts/gpu_r04_measure.sh run)
        if args.dry_run or world != 1:
            print("bench.py: --row needs one real GPU", file=sys.stderr)
            return 2
        hip = Hip()
        hip.check(hip.lib.hipSetDevice(0), "hipSetDevice")
        try:
            traffic = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        except (OSError, ValueError):
            traffic = {}
        fn = filters_2m if args.row.startswith("2m_") else rows_8x_80k
        rows = fn(ups, hip, 0, args, 0.0, traffic, only=args.row)
        print(json.dumps({"row": args.row, "rows": rows}), flush=True)
        return 0 if rows else 2
    fname, streams, channels, blocks, desc = CONFIGS[args.config]
    streams = args.streams or streams
    blocks = args.blocks or blocks
    by_channels = args.split == "channels" and world > 1
    if by_channels and channels % world:
        print(f"bench.py: --split channels needs the channel count ({channels}) to be a multiple of --gpus ({world})",
              file=sys.stderr)
        return 2

    ctl = Control(rank, world)
    try:
        return run_rank(args, ups, ctl, rank, local_rank, world, fname, streams, channels, blocks, desc, by_channels)
    except Exception:
        # a failing rank closes its control connection on the way out: its peers' next wait raises instead of hanging
        import traceback

        traceback.print_exc()
        return 1
    finally:
        ctl.close()


def run_rank(args, ups, ctl, rank, local_rank, world, fname, streams, channels, blocks, desc, by_channels) -> int:
    try:
        traffic = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    except (OSError, ValueError):
        traffic = {}
    # this rank's share: its own streams (all channels), or channels [c0, c0 + nch) of the job's streams
    nch = channels // world if by_channels else channels
    c0 = rank * nch if by_channels else 0
    extras = {}
    if os.environ.get("BENCH_FAIL_RANK_FOR_TEST") == str(rank):  # CPU test hook: a rank that dies after the rendezvous
        raise RuntimeError(f"rank {rank}: failure injected by BENCH_FAIL_RANK_FOR_TEST")
    if args.dry_run:
        ok, msg, cfg = ups.read_filter(ROOT / "data" / "coefficients" / fname)
        if not ok:
            print(f"bench.py: {msg}", file=sys.stderr)
            return 1
        ctl.barrier()
        elapsed = 0.010 * (rank + 1)  # deterministic, rank-dependent: the max-reduce must pick the last rank
        ctl.barrier()
        kavg = elapsed * 1e3 / max(args.steps, 1)
        kstat = {"avg": kavg, "min": 0.0, "count": 0}
        units = blocks * streams * nch
        cblock = {"workload": f"configs[{args.config - 1}]: {desc}", "filter": fname, "taps": cfg["taps"],
                  "fft_size": cfg["fft_size"], "block_size": cfg["block_size"], "upsample_factor": cfg["upsample_factor"],
                  "streams_per_gpu": streams, "channels": channels, "blocks_per_channel": blocks,
                  "pcm": "s32 interleaved in/out", "eq": False, "kernel_path": "dry-run"}
    else:
        hip = Hip()
        # BENCH_STACK_RANKS_FOR_TEST=1 (with BENCH_VISIBLE_DEVICES_FOR_TEST >= ranks): every rank on device 0 -- only to
        # rehearse the multi-rank code path on a one-GPU box; its numbers mean nothing
        device = 0 if os.environ.get("BENCH_STACK_RANKS_FOR_TEST") == "1" else local_rank
        hip.check(hip.lib.hipSetDevice(device), "hipSetDevice")
        w = Workload(ups, hip, device, args.config, 0 if by_channels else rank, streams, blocks, eq=True if args.eq else None,
                     channel_slice=(c0, nch) if by_channels else None)
        cold = None
        if world == 1 and not args.no_extras and args.prime_seconds > 0:
            # the contract's W + K steps on a device that has just been idle (data synthesis, uploads): reported beside the
            # headline, which measures the same steps after the priming run below
            _, el, ks = w.run(args.steps, args.warmup)
            cold = summary(w, args.steps, el, ks, 1, None)
        primed = w.prime(args.prime_seconds) if args.prime_seconds > 0 else 0
        _, elapsed, kstat = w.run(args.steps, args.warmup, ctl.barrier)
        checked = w.check_output()
        units, cfg, cblock = w.units, w.cfg, config_block(w)
        cblock["channels"] = channels

    elapsed = ctl.reduce_max(elapsed)
    kstat = dict(kstat, avg=ctl.reduce_max(kstat["avg"]))  # the slowest rank's kernels price the roofline
    # the partition must be disjoint and complete: (stream, channel) units over all ranks
    owned = [[sid, c] for sid in stream_ids(0 if by_channels else rank, streams) for c in range(c0, c0 + nch)]
    all_units = sorted(tuple(u) for part in ctl.all_gather(owned) for u in part)
    total_streams = streams if by_channels else world * streams
    assert all_units == [(sid, c) for sid in range(total_streams) for c in range(channels)], \
        "the (stream, channel) partition must be disjoint and complete"

    if args.dry_run:
        samples = units * cfg["block_size"] * world * args.steps
        bytes_launch = algorithmic_bytes(cfg, units)
        achieved = bytes_launch / (kstat["avg"] * 1e-3) / 1e9
        head = {"value": round(samples / elapsed / 1e6, 3), "ms_per_step": round(elapsed / args.steps * 1e3, 5),
                "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                             "kernel_ms_avg": round(kstat["avg"], 5), "kernel_ms_min": 0.0, "kernel_launches_timed": 0,
                             "algorithmic_bytes_per_launch": int(bytes_launch)}}
    else:
        head = summary(w, args.steps, elapsed, kstat, world, traffic)
        # measured device-to-device copy ceiling of THIS box beside the 8 TB/s spec figure
        ceil = ctl.reduce_max(-copy_ceiling(ups, device)["read_plus_write_GBps"])  # min over ranks
        head["roofline"]["copy_ceiling_GBps"] = round(-ceil, 1)
        head["roofline"]["frac_of_copy_ceiling"] = round(head["roofline"]["achieved"] / -ceil, 5)
        head["roofline"]["copy_ceiling_what"] = ("mi_device_copy_rate: 16-byte-per-lane grid-stride copy kernel, 1 GiB, bytes "
                                                 "read + bytes written over the hipEvent time, best of 5 (min over ranks)")

    if not args.dry_run and world == 1 and not args.no_extras:
        # the same workload as 8 rounds of workgroups per launch, and a sustained run of the headline launch
        variants = {}
        if cold is not None:
            variants["cold_start"] = {"value": cold["value"], "ms_per_step": cold["ms_per_step"],
                                      "kernel_ms_avg": cold["roofline"]["kernel_ms_avg"], "frac": cold["roofline"]["frac"],
                                      "what": "the same W + K steps measured first, before the priming run"}
        done, el, ks = w.run(max(args.steps, 1), 1, min_seconds=2.0)
        s = summary(w, done, el, ks, 1, traffic)
        variants["sustained"] = {"seconds": round(el, 3), "steps": done, "value": s["value"], "ms_per_step": s["ms_per_step"],
                                 "kernel_ms_avg": s["roofline"]["kernel_ms_avg"], "frac": s["roofline"]["frac"]}
        if args.config == 2 and not args.blocks:
            wl = Workload(ups, hip, device, 2, rank, streams, LONG_BLOCKS, eq=True if args.eq else None)
            if args.prime_seconds > 0:
                wl.prime(args.prime_seconds)
            _, el, ks = wl.run(args.steps, args.warmup)
            s = summary(wl, args.steps, el, ks, 1, traffic)
            variants[f"blocks_{LONG_BLOCKS}"] = {"blocks_per_channel": LONG_BLOCKS, "value": s["value"],
                                                 "ms_per_step": s["ms_per_step"], "roofline": s["roofline"],
                                                 "output_check": wl.check_output()}
            wl.close()
        extras["variants"] = variants
        extras["end_to_end"] = end_to_end(ups, w)
        # what a caller with 16-bit / packed 24-bit output gets: fewer bytes per sample over the host link (D2H dominates)
        extras["end_to_end_by_output_format"] = {
            fmt: end_to_end_format(ups, w, fmt) for fmt in ("s16", "s24")}
        # the drop-in call itself: one block of one channel per call through the reference-shaped handle
        extras["process_block_latency"] = process_block_latency(ups, device, w.fpath)
        rows = []
        for cid in sorted(CONFIGS):
            wc = w if cid == args.config and not (args.blocks or args.streams) else Workload(ups, hip, device, cid, rank)
            if args.prime_seconds > 0:  # every row the same way: priming run, W warmup steps, K timed steps
                wc.prime(args.prime_seconds)
            _, el, ks = wc.run(args.steps, args.warmup)
            s = summary(wc, args.steps, el, ks, 1, traffic)
            s["roofline"]["frac_of_copy_ceiling"] = round(s["roofline"]["achieved"] / head["roofline"]["copy_ceiling_GBps"], 5)
            pk = wc.per_kernel_ms()
            rows.append({"id": cid, "config": config_block(wc), "value": s["value"], "unit": "Msamples/s",
                         "ms_per_step": s["ms_per_step"], "roofline": s["roofline"], "output_check": wc.check_output(),
                         "per_kernel_ms": pk})
            if pk.get("transform"):  # the transform kernel alone against the same algorithmic bytes (DESIGN 11.1's yardstick)
                rows[-1]["transform_only_frac"] = round(
                    algorithmic_bytes(wc.cfg, wc.units) / (pk["transform"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            if wc is not w:
                wc.close()
        extras["configs"] = rows
        # configs[2] as a LONG call (1024 blocks per channel: eight rounds of workgroups): where the transform kernel's own
        # workgroups assemble the frames of finished pairs while the rest of the launch computes (cooperative frames,
        # DESIGN 5.3b) -- the stated 256-block call is two rounds long and keeps the frame pass
        wl = Workload(ups, hip, device, 3, rank, blocks=1024)
        if args.prime_seconds > 0:
            wl.prime(args.prime_seconds)
        _, el, ks = wl.run(args.steps, args.warmup)
        s = summary(wl, args.steps, el, ks, 1, traffic)
        extras["config3_1024_blocks"] = {"config": config_block(wl), "value": s["value"], "unit": "Msamples/s",
                                         "ms_per_step": s["ms_per_step"], "roofline": s["roofline"],
                                         "output_check": wl.check_output(), "per_kernel_ms": wl.per_kernel_ms()}
        wl.close()
        # the 640 001-tap "2m" filters the selector prefers when present (alsa_filter_selector.cpp:74-96): transforms of
        # 2^16 / 2^18 points, past the fused kernels -- the two-level path. Designed here by the repo's generator.
        extras["filters_2m"] = filters_2m(ups, hip, device, args, head["roofline"]["copy_ceiling_GBps"], traffic)
        extras["filters_8x_80k"] = rows_8x_80k(ups, hip, device, args, head["roofline"]["copy_ceiling_GBps"], traffic)
        extras["per_kernel_ms"] = ("roofline.kernel_ms_* = one hipEvent pair around all kernels of a call; configs[].per_kernel_ms = "
                                   "untimed extra steps with one event pair per launch (planarize / transform / frames / history; "
                                   "launches that overlap on two streams add up to more than the call); rocprofv3 --kernel-trace "
                                   "--stats, one CSV per config: profiles/r03_*_kernel_stats_config*.csv")

    if not args.dry_run and world > 1 and not args.no_extras:
        # end to end at N GPUs: every rank drives its own GPU through pinned host buffers at the same time, the
        # ranks share the node's host links; aggregate = sum over ranks
        ctl.barrier()
        try:
            mine = end_to_end(ups, w, seconds=1.5)
        except Exception as exc:  # every rank still takes part in the gather below
            mine = {"error": f"{type(exc).__name__}: {exc}"}
        allr = ctl.all_gather(mine)
        if all("value" in r for r in allr):
            extras["end_to_end"] = dict(allr[0], value=round(sum(r["value"] for r in allr), 3),
                                        host_link_GB_per_s=round(sum(r["host_link_GB_per_s"] for r in allr), 2),
                                        per_rank=[r["value"] for r in allr],
                                        note="sum over ranks, all ranks running at the same time")
        else:
            extras["end_to_end"] = {"error": [r.get("error") for r in allr]}

    if by_channels:
        par = (f"channels of the same {channels}-channel stream(s) split into {world} contiguous groups of {nch}, one group "
               "per GPU, no collective")
    else:
        par = f"streams sharded over {world} GPU(s), no collective"
    result = {
        "metric": "output Msamples/s, 80k-tap FIR upsample (overlap-save), interleaved s32 PCM in HBM",
        "value": head["value"],
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong" if by_channels else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": dict(cblock, streams_total=total_streams, channels_per_gpu=nch, parallelism=par,
                       control_plane="unix-socket hub on this node (stdlib); no torch, one HIP runtime per process"),
        "roofline": dict(head["roofline"],
                         note="bytes = units*4B(1+1/L) + 8(N/2+1); duration = hipEvent pair around the call's kernels on the "
                              "launching stream, max over ranks"),
    }
    if not args.dry_run:
        result["output_check"] = checked
        result["priming"] = {"seconds": args.prime_seconds, "calls": primed,
                             "headline_is": "primed" if args.prime_seconds > 0 else "cold",
                             "what": "`value` is the PRIMED measurement: an untimed run of the headline call for `seconds` comes "
                                     "before the W warmup + K timed steps (a device that has just been idle runs its first "
                                     "milliseconds at lower clocks and K = 20 calls of 0.13 ms end before they have risen). The "
                                     "literal contract -- W + K steps on the just-idle device, nothing before -- is "
                                     "variants.cold_start; --prime-seconds 0 makes it the headline"}
        result["hip_runtimes_mapped"] = hip_runtimes_mapped()
    result.update(extras)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            result["cpu_baseline"] = cpu_baseline(ROOT / "data" / "coefficients" / fname, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    ctl.barrier()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
