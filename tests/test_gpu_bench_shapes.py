"""GPU parity of the launch paths bench.py times, at the shapes it times them.

Small parity cases (tests/test_gpu_parity.py) run 2-8 blocks per call: one launch, no chunking, no second stream. The
bench shapes do not: configs[2] (16x, 8 channels, 256 blocks) takes the PIPELINED route -- chunked transform kernels on
the caller's stream, interleave kernels on the engine's second stream, two plane buffers, four cross-stream events
(csrc/engine.hip, "Pipelined launches"); configs[3] runs 32 streams x 32 blocks through the split kernel's XCD item
mapping; configs[4] 64 blocks x 32 channels. Here each is built by bench.py's own Workload class (same filter, EQ,
synthetic PCM, device buffers, strides, stream) and checked three ways:

  * >= 4 (stream, channel) pairs, first and last included, every block, against fp64 truth: 1 LSB + 1e-5 * max|y|
    after the PCM clamp (the bench input is loud: sigma 0.2 through gain-L filters clips, as SURVEY 8d says);
  * bit-identity with the same streams fed in small calls (one launch each: the route the small tests cover);
  * bench.py's own probe check (Workload.check_output) passes on the result.

Forced cases: MIUPS_EXP_PIPELINE=1 with MIUPS_EXP_CHUNK_ROUNDS=1 (>= 5 chunks, both plane halves reused twice) and
MIUPS_EXP_CHUNK_MB small enough for >= 3 serial chunks, each bit-identical to the default route and to truth.
"""
from __future__ import annotations

import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, str(ROOT))
F32_HI = float(np.float32(0.9999999))


@pytest.fixture(scope="module")
def bench():
    import bench as B

    return B


@pytest.fixture(scope="module")
def hip(bench, gpu):
    h = bench.Hip()
    h.check(h.lib.hipSetDevice(gpu), "hipSetDevice")
    return h


def read_out(hip, w) -> np.ndarray:
    out = np.empty(w.out_stride * w.streams // 4, dtype="<i4")
    hip.sync()
    hip.d2h(out, w.d_out)
    return out.reshape(w.streams, w.blocks * w.cfg["block_size"], w.channels)


def truth_channel(O, ups, w, slot, ch) -> np.ndarray:
    """fp64 truth of every block of one (stream, channel), clamped like the PCM store."""
    h, taps, fft, block, L = O.read_filter(w.fpath)
    x = w.host_pcm[slot][:, ch].astype(np.float64) / 2147483648.0
    if w.use_eq:
        y = O.truth_stream(x, O.eq_fold_fir(h, w.eq_text, w.eq_fs), L, w.blocks, block).reshape(-1)
    else:
        y = O.truth_stream(x, h, L, w.blocks, block).reshape(-1)
    return np.clip(y, -1.0, F32_HI)


def run_in_small_calls(ups, hip, w, step) -> np.ndarray:
    """The same device input through a fresh engine, `step` blocks per call (every call a single launch)."""
    eng = ups.Engine(w.filt, w.streams, w.channels, ups.PCM_S32, ups.PCM_S32)
    d_out = hip.malloc(w.out_stride * w.streams)
    in_blk, out_blk = eng.in_bytes(1), eng.out_bytes(1)
    for b0 in range(0, w.blocks, step):
        nb = min(step, w.blocks - b0)
        eng.process_device(w.d_in + b0 * in_blk, d_out + b0 * out_blk, nb, w.stream, in_stride=w.in_stride,
                           out_stride=w.out_stride)
    hip.sync()
    out = np.empty(w.out_stride * w.streams // 4, dtype="<i4")
    hip.d2h(out, d_out)
    hip.free(d_out)
    eng.close()
    return out.reshape(w.streams, w.blocks * w.cfg["block_size"], w.channels)


@pytest.mark.parametrize("config_id,small_step", [(2, 64), (3, 4), (4, 4), (5, 2)])
def test_bench_workload_as_timed(ups, O, bench, hip, gpu, config_id, small_step):
    w = bench.Workload(ups, hip, gpu, config_id, rank=0)
    assert w.eng.path == "fused"
    # first call from zero history (what the truth assumes), exactly the call bench.py repeats
    w.eng.process_device(w.d_in, w.d_out, w.blocks, w.stream)
    y = read_out(hip, w)
    assert np.abs(y).max() > 1 << 24
    pairs = sorted({(0, 0), (0, min(1, w.channels - 1)), (0, w.channels - 1), (w.streams - 1, 0),
                    (w.streams - 1, w.channels - 1), (w.streams // 2, w.channels // 2)})
    assert len(pairs) >= 4 or w.streams * w.channels < 4
    for slot, ch in pairs:
        want = truth_channel(O, ups, w, slot, ch)
        got = y[slot, :, ch].astype(np.float64) / 2147483648.0
        tol = 2.0 ** -31 + 1e-5 * np.abs(want).max()
        assert np.abs(got - want).max() <= tol, (config_id, slot, ch)
    # the same streams in small calls: single-launch route, must agree bit for bit
    np.testing.assert_array_equal(y, run_in_small_calls(ups, hip, w, small_step))
    # second call of the bench loop: history carried from the end of the same buffer; bench's probes must hold
    w.eng.process_device(w.d_in, w.d_out, w.blocks, w.stream)
    hip.sync()
    res = w.check_output()
    assert res["probes"] >= 2 and res["worst_err_over_tol"] <= 1.0
    w.close()


def synthetic_filter(make_filter, fft, taps, L, seed):
    rng = np.random.default_rng(seed)
    h = (rng.standard_normal(taps) * 0.3 / np.sqrt(taps / L)).astype(np.float32)
    return h, make_filter(h, fft, fft - (taps - 1), L, name=f"syn{fft}_{L}")


def run_engine(ups, hip, filt, streams, channels, pcm, blocks, calls=2, want_parts=None, want_coop=None):
    """`calls` consecutive device calls over the same buffers (the second starts from carried history); returns both
    outputs, int32 [call][stream][frame][channel]. want_parts: what Engine.last_phase_parts must say after a call."""
    eng = ups.Engine(filt, streams, channels, ups.PCM_S32, ups.PCM_S32)
    in_stride, out_stride = eng.in_bytes(blocks), eng.out_bytes(blocks)
    d_in, d_out = hip.malloc(in_stride * streams), hip.malloc(out_stride * streams)
    outs = []
    for k in range(calls):
        hip.h2d(d_in, pcm[k])
        eng.process_device(d_in, d_out, blocks)
        hip.sync()
        if want_parts is not None:
            assert eng.last_phase_parts == want_parts
        if want_coop is not None:
            assert eng.last_coop_frames == want_coop
        o = np.empty(out_stride * streams // 4, dtype="<i4")
        hip.d2h(o, d_out)
        outs.append(o.reshape(streams, -1, channels))
    hip.free(d_in)
    hip.free(d_out)
    eng.close()
    return np.stack(outs)


def test_forced_pipelined_launches_reuse_both_plane_halves(ups, O, hip, gpu, make_filter, monkeypatch):
    """K = 4096, 4 channels (planes leave the kernel), 1400 blocks: with MIUPS_EXP_PIPELINE=1 and chunks of ONE full-chip
    round (1024 workgroups = 256 blocks) the call is 6 chunks -- each plane half is written three times, so chunk k+2's
    transform must wait for chunk k's frames (evFrames) and chunk k's interleave for its transform (evFused). Bit-identical
    to the serial single-buffer route (MIUPS_EXP_PIPELINE=0) over two calls, and channels 0 / 3 against fp64 truth."""
    fft, taps, L, channels, streams, blocks = 16384, 8193, 2, 4, 1, 1400
    h, path = synthetic_filter(make_filter, fft, taps, L, 7)
    filt = ups.Filter(path, device=gpu)
    nin, B = (fft - taps + 1) // L, fft - taps + 1
    rng = np.random.default_rng(17)
    pcm = [(np.clip(rng.standard_normal((streams, blocks * nin, channels)) * 0.2, -1, 1) * 2147483647).astype("<i4")
           for _ in range(2)]
    monkeypatch.setenv("MIUPS_EXP_PIPELINE", "0")
    serial = run_engine(ups, hip, filt, streams, channels, pcm, blocks)
    monkeypatch.setenv("MIUPS_EXP_PIPELINE", "1")
    monkeypatch.setenv("MIUPS_EXP_CHUNK_ROUNDS", "1")
    piped = run_engine(ups, hip, filt, streams, channels, pcm, blocks)
    np.testing.assert_array_equal(piped, serial)
    monkeypatch.delenv("MIUPS_EXP_PIPELINE")
    monkeypatch.delenv("MIUPS_EXP_CHUNK_ROUNDS")
    default = run_engine(ups, hip, filt, streams, channels, pcm, blocks)  # what an unforced caller gets at this shape
    np.testing.assert_array_equal(default, serial)
    x = np.concatenate(pcm, axis=1).astype(np.float64) / 2147483648.0
    y = np.concatenate(list(piped), axis=1).astype(np.float64) / 2147483648.0
    for c in (0, channels - 1):
        want = np.clip(O.truth_stream(x[0, :, c], h.astype(np.float64), L, 2 * blocks, B).reshape(-1), -1.0, F32_HI)
        assert np.abs(y[0, :, c] - want).max() <= 2.0 ** -31 + 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("fname,streams,channels,blocks,mb", [
    ("filter_44k_2x_80000_min_phase", 24, 2, 32, 128),   # split kernel: 768 pairs x 408 KB of planes = 300 MB -> 3 chunks
    ("filter_48k_16x_80000_min_phase", 1, 8, 96, 48),    # K = 4096: 96 pairs x 1.6 MB -> chunks of 29 pairs (4 chunks)
])
def test_forced_serial_chunks(ups, O, hip, gpu, monkeypatch, fname, streams, channels, blocks, mb):
    """MIUPS_EXP_CHUNK_MB: the > 1 GiB-of-planes loop (several transform + interleave launches over ONE plane buffer, in
    stream order) at a size a test can afford. Bit-identical to the single-launch route over two calls, plus fp64 truth on
    the first and last (stream, channel)."""
    path = ROOT / "data" / "coefficients" / f"{fname}.json"
    h, taps, fft, block, L = O.read_filter(path)
    filt = ups.Filter(path, device=gpu)
    nin = block // L
    rng = np.random.default_rng(blocks + mb)
    pcm = [(np.clip(rng.standard_normal((streams, blocks * nin, channels)) * 0.05, -1, 1) * 2147483647).astype("<i4")
           for _ in range(2)]
    monkeypatch.setenv("MIUPS_EXP_PIPELINE", "0")
    one = run_engine(ups, hip, filt, streams, channels, pcm, blocks)
    monkeypatch.setenv("MIUPS_EXP_CHUNK_MB", str(mb))
    chunked = run_engine(ups, hip, filt, streams, channels, pcm, blocks)
    np.testing.assert_array_equal(chunked, one)
    x = np.concatenate(pcm, axis=1).astype(np.float64) / 2147483648.0
    y = np.concatenate(list(chunked), axis=1).astype(np.float64) / 2147483648.0
    for s, c in ((0, 0), (streams - 1, channels - 1)):
        want = np.clip(O.truth_stream(x[s, :, c], h, L, 2 * blocks, block).reshape(-1), -1.0, F32_HI)
        assert np.abs(y[s, :, c] - want).max() <= 2.0 ** -31 + 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("fname,streams,channels,blocks,parts", [
    ("filter_44k_4x_80000_min_phase", 1, 1, 1, 4),    # the reference's call shape: 4 workgroups, one phase each (K = 16384)
    ("filter_44k_4x_80000_min_phase", 1, 2, 3, 4),    # stereo, 6 items x 4 parts: both channels out of 16-byte frame loads
    ("filter_48k_16x_80000_min_phase", 1, 8, 2, 16),  # P = 16, 16 items x 16 parts = 256 workgroups (K = 4096), planar input
    ("filter_48k_8x_160000_linear_phase", 2, 3, 5, 8),  # 30 items x 8 parts; odd channel count
    ("filter_48k_16x_80000_min_phase", 3, 2, 7, 4),   # 42 items: of P = 16's divisors 4 is the largest that fits (42 x 4 <= 256)
    ("filter_44k_2x_80000_min_phase", 1, 1, 1, 4),    # split form (K = 32768 as two halves): its 2P = 4 half transforms on 4 workgroups
    ("filter_44k_2x_80000_min_phase", 2, 2, 9, 4),    # ... 36 items x 4; stereo through the split-planar timeline
    ("filter_48k_2x_80000_min_phase", 1, 2, 40, 2),   # 80 items: two workgroups each (one phase = both halves)
])
def test_small_calls_split_their_phases_over_workgroups(ups, O, hip, gpu, monkeypatch, fname, streams, channels, blocks, parts):
    """Calls with far fewer work items than CUs take fused_parts_kernel (several workgroups per channel-block, P / parts
    phases each, frames by the interleave kernel). Bit-identical to the one-workgroup-per-item route
    (MIUPS_EXP_NO_PHASE_PARTS=1) over two calls, and against fp64 truth."""
    path = ROOT / "data" / "coefficients" / f"{fname}.json"
    h, taps, fft, block, L = O.read_filter(path)
    filt = ups.Filter(path, device=gpu)
    nin = block // L
    rng = np.random.default_rng(streams * 100 + channels * 10 + blocks)
    pcm = [(np.clip(rng.standard_normal((streams, blocks * nin, channels)) * 0.05, -1, 1) * 2147483647).astype("<i4")
           for _ in range(2)]
    monkeypatch.setenv("MIUPS_EXP_NO_PHASE_PARTS", "1")
    plain = run_engine(ups, hip, filt, streams, channels, pcm, blocks, want_parts=0)
    monkeypatch.delenv("MIUPS_EXP_NO_PHASE_PARTS")
    split = run_engine(ups, hip, filt, streams, channels, pcm, blocks, want_parts=parts)
    np.testing.assert_array_equal(split, plain)
    x = np.concatenate(pcm, axis=1).astype(np.float64) / 2147483648.0
    y = np.concatenate(list(split), axis=1).astype(np.float64) / 2147483648.0
    for s, c in ((0, 0), (streams - 1, channels - 1)):
        want = np.clip(O.truth_stream(x[s, :, c], h, L, 2 * blocks, block).reshape(-1), -1.0, F32_HI)
        assert np.abs(y[s, :, c] - want).max() <= 2.0 ** -31 + 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("fname,streams,channels,blocks", [
    ("filter_48k_16x_80000_min_phase", 1, 8, 48),       # BASELINE configs[2] shape: K = 4096, 128 planes, several co-resident workgroups
    ("filter_48k_8x_160000_linear_phase", 1, 32, 10),   # BASELINE configs[4] shape: K = 16384, 256 planes, one workgroup per CU
    ("filter_48k_8x_80000_min_phase", 2, 32, 6),        # K = 8192, two streams
    ("filter_48k_8x_80000_min_phase", 1, 2, 300),       # stereo that does not fill the chip: one channel per workgroup, 16 planes
])
def test_cooperative_frames_are_bit_identical_to_the_frame_pass(ups, O, hip, gpu, monkeypatch, fname, streams, channels, blocks):
    """Cooperative frames (device/frame_tile.h; taken by default where several workgroups share a CU and the launch is at
    least four rounds long, forced here at every shape the kernels cover): the transform kernel's own workgroups assemble
    the PCM frames of finished pairs, the frame pass behind it takes what nobody claimed.
    Same planes, same conversion: bit-identical to the frame pass alone (MIUPS_EXP_NO_COOP_FRAMES=1) over two calls, with a
    cap nobody reaches (MIUPS_EXP_COOP_CAP: the pairs' last workgroups take everything) as with the default one, and against
    fp64 truth."""
    path = ROOT / "data" / "coefficients" / f"{fname}.json"
    h, taps, fft, block, L = O.read_filter(path)
    filt = ups.Filter(path, device=gpu)
    nin = block // L
    rng = np.random.default_rng(streams * 100 + channels * 10 + blocks)
    pcm = [(np.clip(rng.standard_normal((streams, blocks * nin, channels)) * 0.05, -1, 1) * 2147483647).astype("<i4")
           for _ in range(2)]
    monkeypatch.setenv("MIUPS_EXP_NO_COOP_FRAMES", "1")
    plain = run_engine(ups, hip, filt, streams, channels, pcm, blocks, want_coop=False)
    monkeypatch.delenv("MIUPS_EXP_NO_COOP_FRAMES")
    monkeypatch.setenv("MIUPS_EXP_COOP_FRAMES", "1")   # whatever the launch shape (the default takes it only where it pays)
    coop = run_engine(ups, hip, filt, streams, channels, pcm, blocks, want_coop=True)
    np.testing.assert_array_equal(coop, plain)
    monkeypatch.setenv("MIUPS_EXP_COOP_CAP", "100000")
    np.testing.assert_array_equal(run_engine(ups, hip, filt, streams, channels, pcm, blocks, want_coop=True), plain)
    x = np.concatenate(pcm, axis=1).astype(np.float64) / 2147483648.0
    y = np.concatenate(list(coop), axis=1).astype(np.float64) / 2147483648.0
    for s, c in ((0, 0), (streams - 1, channels - 1)):
        want = np.clip(O.truth_stream(x[s, :, c], h, L, 2 * blocks, block).reshape(-1), -1.0, F32_HI)
        assert np.abs(y[s, :, c] - want).max() <= 2.0 ** -31 + 1e-5 * np.abs(want).max()


def test_cooperative_frames_default_policy(ups, hip, gpu):
    """The default takes cooperative frames only where they measured faster (profiles/r04_b_coop_frames.txt): K <= 4096 and
    at least four full-chip rounds of workgroups. configs[2]'s shape at 512 blocks: yes; at its stated 256 blocks: no (the
    two-stream pipelined route stays); K = 16384 (configs[4]): no."""
    f16 = ups.Filter(ROOT / "data" / "coefficients" / "filter_48k_16x_80000_min_phase.json", device=gpu)
    f8 = ups.Filter(ROOT / "data" / "coefficients" / "filter_48k_8x_160000_linear_phase.json", device=gpu)
    for filt, channels, blocks, want in ((f16, 8, 512, True), (f16, 8, 256, False), (f8, 32, 64, False)):
        eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
        d_in, d_out = hip.malloc(eng.in_bytes(blocks)), hip.malloc(eng.out_bytes(blocks))
        hip.h2d(d_in, np.zeros(eng.in_bytes(blocks), dtype=np.uint8))
        eng.process_device(d_in, d_out, blocks)
        hip.sync()
        assert eng.last_coop_frames == want, (channels, blocks)
        hip.free(d_in)
        hip.free(d_out)
        eng.close()
