"""GPU parity tests: the HIP path (through the C ABI) against the oracle.

Tolerances (fp32 path, stated per SURVEY §8c):
  * known-answer cases of tests/cpp/test_vulkan_upsampler.cpp: abs 1e-3 (its own eps)
  * vs the reference's arithmetic (oracle goldens / live oracle):  max|d| <= 2e-3 * max|y|
    (the reference's radix-2 recurrence FFT is itself ~5e-4*max|y| off fp64)
  * vs fp64 truth:                                               max|d| <= 1e-5 * max|y|
  * vs the Vulkan-path simulation with MI_LOAD_REF_COMPAT_SPECTRUM: max|d| <= 1e-5 * max|y|
"""
from __future__ import annotations

import json
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, real_input, rel_err

pytestmark = pytest.mark.gpu

TOL_REF = 2e-3
TOL_TRUTH = 1e-5
KNOWN = np.array([1, 2, 3, 2, 1], dtype=np.float32)


def stream_blocks(u, x, nin):
    return np.stack([u.process_block(x[i * nin:(i + 1) * nin]) for i in range(len(x) // nin)])


# ---- G1: the reference's own three known-answer checks ---------------------
def test_known_answer_impulse_and_stream(ups, gpu, make_filter):
    g = np.load(GOLDEN / "g1_known_answer.npz")
    for L in (1, 2):
        u = ups.StreamingUpsampler(gpu)
        ok, msg = u.load_filter(make_filter(KNOWN, 16, 12, L, name=f"k{L}"))
        assert ok, msg
        out = u.process_block(g[f"L{L}_impulse_in"])
        assert out.shape == (12,)
        np.testing.assert_allclose(out, g[f"L{L}_impulse_out"], atol=1e-3)
        # direct convolution, as the reference test computes it
        up = np.zeros(12)
        up[::L] = g[f"L{L}_impulse_in"]
        np.testing.assert_allclose(out, np.convolve(up, KNOWN)[:12], atol=1e-3)
        u.reset()
        a, b = g[f"L{L}_iota_in"]
        got = np.stack([u.process_block(a), u.process_block(b)])
        np.testing.assert_allclose(got, g[f"L{L}_iota_out"], atol=1e-3)


def test_process_block_guards_return_empty(ups, gpu, make_filter):
    u = ups.StreamingUpsampler(gpu)
    assert u.process_block(np.zeros(12, np.float32)).size == 0  # not initialised
    ok, _ = u.load_filter(make_filter(KNOWN, 16, 12, 2))
    assert ok
    assert u.process_block(np.zeros(0, np.float32)).size == 0   # count == 0
    assert u.process_block(np.zeros(5, np.float32)).size == 0   # count != block/L
    assert u.process_block(np.zeros(12, np.float32)).size == 0  # block, not block/L
    assert u.process_block(np.zeros(6, np.float32)).size == 12


# ---- G2: 257 taps, fft 1024, every ratio, full vectors ----------------------
@pytest.mark.parametrize("L", [1, 2, 4, 8, 16])
def test_mid_geometry_vs_golden(ups, gpu, make_filter, L):
    g = np.load(GOLDEN / "g2_mid.npz")
    u = ups.StreamingUpsampler(gpu)
    ok, msg = u.load_filter(make_filter(g["taps"], 1024, 768, L))
    assert ok, msg
    x = g[f"L{L}_in"]
    y = np.stack([u.process_block(x[b]) for b in range(4)])
    assert rel_err(y, g[f"L{L}_ref"]) <= TOL_REF
    assert rel_err(y, g[f"L{L}_truth"]) <= TOL_TRUTH


# ---- G3: the real 80k/160k-tap geometries -----------------------------------
G3 = json.loads((GOLDEN / "g3_real.json").read_text())


@pytest.mark.parametrize("name", sorted(G3))
def test_real_geometry_vs_golden(ups, gpu, name):
    meta = G3[name]
    g = np.load(GOLDEN / "g3_real.npz")
    nin = meta["block"] // meta["factor"]
    x = real_input(meta["seed"], meta["blocks"] * nin)
    np.testing.assert_array_equal(x[:8], g[f"{name}_x_head"])  # the seeded input is the one the golden saw
    u = ups.StreamingUpsampler(gpu)
    ok, msg = u.load_filter(ROOT / meta["filter"])
    assert ok, msg
    y = stream_blocks(u, x, nin)
    idx = g[f"{name}_idx"]
    scale = g[f"{name}_truth_max"].max()
    assert np.abs(y[:, idx] - g[f"{name}_ref"]).max() <= TOL_REF * scale
    assert np.abs(y[:, idx] - g[f"{name}_truth"]).max() <= TOL_TRUTH * scale
    # whole-block checksums against fp64 truth
    np.testing.assert_allclose(y.astype(np.float64).sum(axis=1), g[f"{name}_truth_sum"], atol=1e-3 * scale * 50)
    np.testing.assert_allclose(np.sqrt((y.astype(np.float64) ** 2).sum(axis=1)), g[f"{name}_truth_l2"], rtol=1e-5)


@pytest.mark.parametrize("name", sorted(G3))  # incl. the split kernel (44k_2x, K = 32768) and the 160 001-tap filter
def test_ref_compat_spectrum_matches_vulkan_path_simulation(ups, gpu, name):
    meta = G3[name]
    g = np.load(GOLDEN / "g3_real.npz")
    nin = meta["block"] // meta["factor"]
    x = real_input(meta["seed"], meta["blocks"] * nin)
    u = ups.StreamingUpsampler(gpu)
    ok, msg = u.load_filter(ROOT / meta["filter"], flags=ups.LOAD_REF_COMPAT_SPECTRUM)
    assert ok, msg
    y = stream_blocks(u, x, nin)
    idx = g[f"{name}_idx"]
    scale = g[f"{name}_truth_max"].max()
    assert np.abs(y[:, idx] - g[f"{name}_vksim"]).max() <= TOL_TRUTH * scale


# ---- live oracle, full vectors ------------------------------------------------
@pytest.mark.parametrize("fname", ["filter_44k_4x_80000_min_phase", "filter_48k_16x_80000_min_phase",
                                   "filter_44k_2x_80000_min_phase"])
def test_live_oracle_full_vectors(ups, O, gpu, fname):
    path = ROOT / "data" / "coefficients" / f"{fname}.json"
    h, taps, fft, block, L = O.read_filter(path)
    nin = block // L
    x = real_input(77, 3 * nin)
    orc = O.OracleUpsampler(h, taps, fft, block, L)
    ref = np.stack([orc.process_block(x[b * nin:(b + 1) * nin]) for b in range(3)])
    truth = O.truth_stream(x, h, L, 3, block)
    u = ups.StreamingUpsampler(gpu)
    assert u.load_filter(path)[0]
    y = stream_blocks(u, x, nin)
    assert rel_err(y, ref) <= TOL_REF
    assert rel_err(y, truth) <= TOL_TRUTH


# ---- batched engine == block-by-block operator --------------------------------
@pytest.mark.parametrize("fmt_in,fmt_out", [("s32", "s32"), ("s16", "s16"), ("s24", "s24"), ("f32", "f32"), ("s32", "f32")])
def test_engine_batched_interleaved_matches_oracle(ups, O, gpu, fmt_in, fmt_out):
    path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
    h, taps, fft, block, L = O.read_filter(path)
    streams, channels, blocks = 2, 2, 3
    filt = ups.Filter(path, device=gpu)
    eng = ups.Engine(filt, streams, channels, ups.PCM_NAMES[fmt_in], ups.PCM_NAMES[fmt_out])
    assert eng.path == "fused"
    nin = eng.in_frames
    rng = np.random.default_rng(5)
    xf = np.clip(rng.standard_normal((streams, blocks * nin, channels)) * 0.05, -1, 1).astype(np.float32)
    raw = xf if fmt_in == "f32" else O.float_to_pcm(xf.reshape(-1), fmt_in)
    xin = xf.reshape(-1) if fmt_in == "f32" else O.pcm_to_float(raw, fmt_in)
    xin = xin.reshape(streams, blocks * nin, channels)
    out = eng.process_host(raw, blocks)
    for s in range(streams):
        for c in range(channels):
            truth = O.truth_stream(xin[s, :, c], h, L, blocks, block).reshape(-1)
            if fmt_out == "f32":
                y = out.view(np.float32).reshape(streams, blocks * block, channels)[s, :, c]
                assert rel_err(y, truth) <= TOL_TRUTH
            else:
                got = O.pcm_to_float(out, fmt_out).reshape(streams, blocks * block, channels)[s, :, c]
                lsb = {"s16": 2.0**-15, "s24": 2.0**-23, "s32": 2.0**-31}[fmt_out]
                want = O.pcm_to_float(O.float_to_pcm(truth.astype(np.float32), fmt_out), fmt_out)
                # truncation of a value 1e-5-close to truth: at most one LSB plus the fp32 error
                assert np.abs(got - want).max() <= lsb + TOL_TRUTH * np.abs(truth).max()


# ---- wide frames: planarize -> one channel per workgroup -> interleave kernels -----
@pytest.mark.parametrize("fname,channels,streams,fmt_in,fmt_out", [
    ("filter_48k_16x_80000_min_phase", 8, 1, "s32", "s32"),     # BASELINE config 3 shape: vector interleave
    ("filter_44k_4x_80000_min_phase", 6, 2, "s24", "s24"),      # 5.1, packed 24-bit both ways: scalar interleave
    ("filter_44k_4x_80000_min_phase", 3, 1, "s16", "f32"),      # odd channel count
    ("filter_48k_8x_160000_linear_phase", 32, 1, "s32", "s32"), # BASELINE config 5 shape
    ("filter_44k_2x_80000_min_phase", 4, 1, "f32", "f32"),      # K = 32768: staged path, wide frame
])
def test_wide_frames_match_truth_across_calls(ups, O, gpu, fname, channels, streams, fmt_in, fmt_out):
    path = ROOT / "data" / "coefficients" / f"{fname}.json"
    h, taps, fft, block, L = O.read_filter(path)
    filt = ups.Filter(path, device=gpu)
    eng = ups.Engine(filt, streams, channels, ups.PCM_NAMES[fmt_in], ups.PCM_NAMES[fmt_out])
    nin, calls, blocks = eng.in_frames, 2, 2
    rng = np.random.default_rng(channels)
    xf = np.clip(rng.standard_normal((calls, streams, blocks * nin, channels)) * 0.05, -1, 1).astype(np.float32)
    outs, xin = [], []
    for k in range(calls):  # the second call reads real history (through the planar timelines when channels > 2)
        raw = xf[k] if fmt_in == "f32" else O.float_to_pcm(xf[k].reshape(-1), fmt_in)
        xin.append((xf[k].reshape(-1) if fmt_in == "f32" else O.pcm_to_float(raw, fmt_in)).reshape(streams, blocks * nin, channels))
        out = eng.process_host(raw, blocks)
        y = out.view(np.float32) if fmt_out == "f32" else O.pcm_to_float(out, fmt_out)
        outs.append(y.reshape(streams, blocks * block, channels))
    y = np.concatenate(outs, axis=1)
    x = np.concatenate(xin, axis=1)
    lsb = {"f32": 0.0, "s16": 2.0**-15, "s24": 2.0**-23, "s32": 2.0**-31}[fmt_out]
    for s in range(streams):
        for c in sorted({0, 1, channels // 2, channels - 1}):
            truth = O.truth_stream(x[s, :, c], h, L, calls * blocks, block).reshape(-1)
            assert np.abs(y[s, :, c] - truth).max() <= lsb + TOL_TRUTH * np.abs(truth).max()


@pytest.mark.parametrize("fft,taps,L,channels,streams,blocks,fmt_in,fmt_out", [
    # frame shapes that steer the helper kernels: planes per frame group R = L*channels -> rows (4, 8), quad (12), tiled
    # (16 .. 512, incl. non powers of two and a partial last tile), scalar (R % 4 != 0); planarize fast path (4-byte input,
    # whole tiles) and general loop (partial tiles, packed formats)
    (4096, 1025, 2, 2, 3, 5, "s32", "s32"),     # R = 4: stereo, whole-frame epilogue (rows form inside the kernel), three streams
    (4096, 1025, 4, 3, 1, 4, "s32", "s32"),     # R = 12: quad form, odd channel count, planar input
    (8192, 2113, 4, 5, 2, 3, "f32", "f32"),     # R = 20: quad form (20 % 16 != 0), Bc = 1520
    (8192, 2113, 8, 6, 1, 3, "s32", "f32"),     # R = 48: tiled, three 16-row groups, partial last tile (Bc = 760)
    (4096, 1025, 16, 12, 1, 2, "s32", "s32"),   # R = 192: tiled with 32-wide tiles
    (4096, 1025, 16, 32, 1, 2, "f32", "s32"),   # R = 512: tiled with 16-wide tiles
    (8192, 2049, 2, 7, 1, 3, "s24", "s16"),     # R = 14: scalar interleave, packed input through the general planarize loop
    (16384, 8193, 2, 2, 2, 3, "s32", "s32"),    # K = 4096, long history (8192 of 16384 samples), two stereo streams
])
def test_helper_kernel_shapes_match_truth(ups, O, gpu, make_filter, fft, taps, L, channels, streams, blocks, fmt_in, fmt_out):
    """Synthetic geometries through every frame-assembly and planarize variant, two calls each (the second starts from
    carried history), every channel against fp64 truth."""
    rng = np.random.default_rng(fft + taps + L + channels)
    h = (rng.standard_normal(taps) * 0.3 / np.sqrt(taps / L)).astype(np.float32)
    block = fft - (taps - 1)
    filt = ups.Filter(make_filter(h, fft, block, L), device=gpu)
    eng = ups.Engine(filt, streams, channels, ups.PCM_NAMES[fmt_in], ups.PCM_NAMES[fmt_out])
    nin, calls = eng.in_frames, 2
    xf = np.clip(rng.standard_normal((calls, streams, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    outs, xin = [], []
    for k in range(calls):
        raw = xf[k] if fmt_in == "f32" else O.float_to_pcm(xf[k].reshape(-1), fmt_in)
        xin.append((xf[k].reshape(-1) if fmt_in == "f32" else O.pcm_to_float(raw, fmt_in)).reshape(streams, blocks * nin, channels))
        out = eng.process_host(raw, blocks)
        y = out.view(np.float32) if fmt_out == "f32" else O.pcm_to_float(out, fmt_out)
        outs.append(y.reshape(streams, blocks * block, channels))
    y = np.concatenate(outs, axis=1)
    x = np.concatenate(xin, axis=1)
    lsb = {"f32": 0.0, "s16": 2.0**-15, "s24": 2.0**-23, "s32": 2.0**-31}[fmt_out]
    for s in range(streams):
        for c in range(channels):
            truth = O.truth_stream(x[s, :, c], h.astype(np.float64), L, calls * blocks, block).reshape(-1)
            assert np.abs(y[s, :, c] - truth).max() <= lsb + TOL_TRUTH * np.abs(truth).max(), (s, c)


# ---- float -> PCM on the device: clamp + truncate, bit for bit ---------------------
def _pcm_case_filter(make_filter, case):
    rng = np.random.default_rng(42)
    if case == "real2x":
        return ROOT / "data" / "coefficients" / "filter_44k_2x_80000_min_phase.json"
    fft, block, L = {"L4": (1024, 768, 4), "L16": (1024, 768, 16), "L16odd": (1024, 720, 16)}[case]
    n = fft - block + 1
    taps = (rng.standard_normal(n) * 0.55 / np.sqrt(n / L)).astype(np.float32)  # output sigma ~ 0.55 x the input's
    return make_filter(taps, fft, block, L, name=f"pcm_{case}")


@pytest.mark.parametrize("case,channels,blocks", [
    ("L4", 2, 2048),      # whole-frame stereo, 8 planes: epilogue_vec (s32) / epilogue_scalar (s16, s24)
    ("L16", 2, 2048),     # whole-frame stereo, 32 planes, Bc % 4 == 0: epilogue_quad
    ("L16odd", 2, 2048),  # Bc = 45, odd history: epilogue_tiled, plane_write without the even-Oc fast path
    ("L16", 8, 8),        # wide frames: interleave_quad_kernel (s32) / interleave_scalar_kernel (s16, s24)
    ("real2x", 2, 4),     # K = 32768 split kernel: interleave kernels reading split planes
])
@pytest.mark.parametrize("fmt", ["s16", "s24", "s32"])
def test_device_float_to_pcm_saturates_bit_exact(ups, O, gpu, make_filter, case, channels, blocks, fmt):
    """ConvertFloatToPcm (alsa_common.cpp:87-127): clamp to [-1, 0.9999695 | 0.9999999], scale in fp32,
    truncate toward zero; NaN -> the upper clamp (std::min/std::max comparison order). The engine is run
    twice on the same loud input -- once with f32 output, once with PCM output: the float values are the
    same (same kernels, same order), so the PCM bytes must equal the oracle's conversion of those floats
    EXACTLY. The input drives a large share of the outputs past +-1 and, in its last block, to +-inf / NaN."""
    path = _pcm_case_filter(make_filter, case)
    filt = ups.Filter(path, device=gpu)
    ef = ups.Engine(filt, 1, channels, ups.PCM_F32, ups.PCM_F32)
    ep = ups.Engine(filt, 1, channels, ups.PCM_F32, ups.PCM_NAMES[fmt])
    nin = ef.in_frames
    rng = np.random.default_rng(99)
    x = (rng.standard_normal((blocks * nin, channels)) * 1.5).astype(np.float32)
    if case != "real2x":            # (an 80k-tap tail would ring above full scale for the whole test)
        x[:nin:7, 0] = 1.0e9        # far beyond full scale
        x[1:nin:7, channels - 1] = -1.0e9
    x[(blocks - 1) * nin + 3, 0] = np.inf             # the whole last block of channel 0 becomes NaN / inf
    x[(blocks - 1) * nin + 5, channels - 1] = -np.inf
    yf = ef.process_host(x, blocks).view(np.float32)
    got = ep.process_host(x, blocks)
    B = filt.config["block_size"]
    y = yf.reshape(blocks * B, channels)
    assert np.isnan(y[-B:, 0]).any() and (np.abs(y[:-B]) > 1.0).mean() > 0.02 and (np.abs(y[:-B]) < 1.0).mean() > 0.02
    np.testing.assert_array_equal(got, O.float_to_pcm(yf, fmt))
    # and the host-side converter of the C ABI (mi_float_to_pcm) agrees on the same floats
    np.testing.assert_array_equal(ups.float_to_pcm(yf, ups.PCM_NAMES[fmt]), O.float_to_pcm(yf, fmt))


def test_wide_frame_channels_are_independent_and_ordered(ups, gpu):
    """Every channel of a 12-channel frame carries a different constant: each output channel
    must converge to its own DC level (catches any channel/phase permutation in the
    planarize / interleave index maps at full size)."""
    path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
    filt = ups.Filter(path, device=gpu)
    cfg = filt.config
    B, L = cfg["block_size"], cfg["upsample_factor"]
    channels, blocks = 12, 4
    eng = ups.Engine(filt, 1, channels, ups.PCM_F32, ups.PCM_F32)
    h = np.fromfile(cfg["coefficients_path"], "<f4").astype(np.float64)
    levels = (np.arange(channels) + 1) / 64.0
    x = np.broadcast_to(levels.astype(np.float32), (blocks * eng.in_frames, channels)).copy()
    y = eng.process_host(x, blocks).view(np.float32).reshape(blocks * B, channels)
    want = levels * h.sum() / L
    assert np.abs(y[-1000:].mean(axis=0) - want).max() <= 1e-5


@pytest.mark.parametrize("fmt", ["s32", "f32"])
def test_bench_sized_stereo_batch_whole_frame_epilogue(ups, O, gpu, fmt):
    """The bench's own shape: a stereo call big enough to give every CU a workgroup (256 blocks
    at K = 16384) takes the whole-frame path -- two channels per workgroup, frames written by
    the kernel's own epilogue. The same stream fed in 64-block calls takes the other path (one
    channel per workgroup + interleave kernel). Same arithmetic per channel: the outputs must
    be bit-identical, and channel 1 must match fp64 truth."""
    path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
    h, taps, fft, block, L = O.read_filter(path)
    filt = ups.Filter(path, device=gpu)
    pcm = ups.PCM_NAMES[fmt]
    eng = ups.Engine(filt, 1, 2, pcm, pcm)
    nin, blocks = eng.in_frames, 256
    xf = np.clip(np.random.default_rng(11).standard_normal((blocks * nin, 2)) * 0.1, -1, 1).astype(np.float32)
    raw = xf if fmt == "f32" else O.float_to_pcm(xf.reshape(-1), fmt)
    whole = eng.process_host(raw, blocks).copy()
    eng.reset()
    step = 64
    per_call = step * nin * 2 * 4  # bytes: both formats are 4-byte samples
    raw_b = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    parts = [eng.process_host(raw_b[i * per_call:(i + 1) * per_call].view(np.float32 if fmt == "f32" else np.uint8), step).copy()
             for i in range(blocks // step)]
    np.testing.assert_array_equal(whole.view(np.uint8).reshape(-1), np.concatenate([p.view(np.uint8).reshape(-1) for p in parts]))
    xin = xf if fmt == "f32" else O.pcm_to_float(raw, fmt).reshape(-1, 2)
    y = (whole.view(np.float32) if fmt == "f32" else O.pcm_to_float(whole, fmt)).reshape(blocks * block, 2)
    truth = O.truth_stream(xin[:, 1], h, L, blocks, block).reshape(-1)
    lsb = 0.0 if fmt == "f32" else 2.0**-31
    assert np.abs(y[:, 1] - truth).max() <= lsb + TOL_TRUTH * np.abs(truth).max()


@pytest.mark.parametrize("fname,channels,blocks,switch", [
    ("filter_48k_16x_80000_min_phase", 8, 4, "MIUPS_EXP_NO_TILED_INTERLEAVE"),  # 128 planes per frame group: tiled vs quad
    ("filter_48k_8x_80000_min_phase", 12, 3, "MIUPS_EXP_NO_TILED_INTERLEAVE"),  # 96 planes (not a power of two)
    ("filter_44k_2x_80000_min_phase", 2, 5, "MIUPS_EXP_NO_SPLIT_PLANAR"),       # split form: split-planar timeline vs stereo PCM
    ("filter_44k_2x_80000_min_phase", 1, 3, "MIUPS_EXP_NO_SPLIT_PLANAR"),       # ... mono
])
def test_alternative_data_paths_are_bit_identical(ups, gpu, monkeypatch, fname, channels, blocks, switch):
    """Two routes for the same numbers: frames assembled by interleave_tiled_kernel or interleave_quad_kernel, and the
    split form fed from the split-planar timeline (planarize_kernel, split_planes) or straight from the caller's PCM. Neither
    changes a single arithmetic operation, so the outputs must agree bit for bit (two calls: the second one starts from
    carried history)."""
    path = ROOT / "data" / "coefficients" / f"{fname}.json"
    filt = ups.Filter(path, device=gpu)
    rng = np.random.default_rng(channels * 100 + blocks)

    def run():
        eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
        outs = []
        for _ in range(2):
            x = (np.clip(rng_local.standard_normal((blocks * eng.in_frames, channels)) * 0.2, -1, 1) * 2147483647).astype("<i4")
            outs.append(eng.process_host(x, blocks).copy())
        return np.concatenate([o.view(np.uint8).reshape(-1) for o in outs])

    rng_local = np.random.default_rng(rng.integers(1 << 30))
    state = rng_local.bit_generator.state
    a = run()
    monkeypatch.setenv(switch, "1")
    rng_local.bit_generator.state = state
    b = run()
    assert a.size == 2 * blocks * filt.config["block_size"] * channels * 4
    np.testing.assert_array_equal(a, b)
    assert np.abs(a.view("<i4")).max() > 1 << 20  # not silence


def test_class_timing_reports_the_launches_of_a_call(ups, gpu):
    """mi_engine_last_class_ms: a wide-frame call has a planarize, a transform, a frame-assembly and a history launch; a
    stereo call on the whole-frame path has no planarize and no separate frame pass; the output is unchanged by the probes."""
    path = ROOT / "data" / "coefficients" / "filter_48k_8x_80000_min_phase.json"
    filt = ups.Filter(path, device=gpu)
    for channels, expect_none in ((8, ()), (2, ("planarize",))):
        eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
        blocks = 2 if channels == 8 else 300
        x = (np.clip(np.random.default_rng(3).standard_normal((blocks * eng.in_frames, channels)) * 0.2, -1, 1) * 2147483647).astype("<i4")
        plain = eng.process_host(x, blocks).copy()
        eng.reset()
        eng.enable_class_timing(True)
        probed = eng.process_host(x, blocks).copy()
        ms = eng.last_class_ms()
        np.testing.assert_array_equal(plain, probed)
        assert ms["transform"] > 0 and ms["history"] > 0
        for k in expect_none:
            assert ms[k] is None
        if channels == 8:
            assert ms["planarize"] > 0 and ms["frames"] > 0


@pytest.mark.parametrize("channels,blocks,in_off,out_off", [
    (2, 256, 4, 0),    # whole-frame path, input 4-byte aligned only: per-sample loads instead of 16-byte frame pairs
    (2, 256, 0, 4),    # whole-frame path, output 4-byte aligned only: scalar epilogue
    (2, 3, 4, 4),      # one channel per workgroup + scalar interleave kernel
    (8, 2, 8, 12),     # planar input + scalar interleave kernel
    (1, 2, 4, 8),      # mono: 8-byte complex-word loads need 8-byte alignment
])
def test_device_buffers_with_minimal_alignment(ups, O, gpu, channels, blocks, in_off, out_off):
    """A caller's device buffers need only be sample-aligned: every vector fast path has to fall
    back on its own. Same stream through aligned and deliberately offset device addresses:
    bit-identical output."""
    sys.path.insert(0, str(ROOT))
    from bench import Hip

    hip = Hip()
    hip.check(hip.lib.hipSetDevice(gpu), "hipSetDevice")
    path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
    filt = ups.Filter(path, device=gpu)
    eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
    x = (np.clip(np.random.default_rng(3).standard_normal((blocks * eng.in_frames, channels)) * 0.1, -1, 1)
         * (2**31 - 1)).astype("<i4")
    want = eng.process_host(x, blocks).copy()
    eng.reset()
    nin, nout = eng.in_bytes(blocks), eng.out_bytes(blocks)
    d_in, d_out = hip.malloc(nin + 64), hip.malloc(nout + 64)
    hip.h2d(d_in + in_off, x)
    eng.process_device(d_in + in_off, d_out + out_off, blocks)
    hip.sync()
    got = np.empty(nout, np.uint8)
    hip.d2h(got, d_out + out_off)
    hip.check(hip.lib.hipFree(d_in), "hipFree")
    hip.check(hip.lib.hipFree(d_out), "hipFree")
    np.testing.assert_array_equal(got, want)


def test_history_carries_across_calls_and_reset(ups, O, gpu):
    path = ROOT / "data" / "coefficients" / "filter_48k_16x_80000_min_phase.json"
    filt = ups.Filter(path, device=gpu)
    eng = ups.Engine(filt, 1, 1, ups.PCM_F32, ups.PCM_F32)
    nin = eng.in_frames
    x = real_input(9, 5 * nin)
    whole = eng.process_host(x, 5).view(np.float32)
    eng.reset()
    parts = np.concatenate([eng.process_host(x[:2 * nin], 2).view(np.float32),
                            eng.process_host(x[2 * nin:3 * nin], 1).view(np.float32),
                            eng.process_host(x[3 * nin:], 2).view(np.float32)])
    np.testing.assert_array_equal(whole, parts)  # same kernel, same inputs: bit-identical
    eng.reset()
    again = eng.process_host(x, 5).view(np.float32)
    np.testing.assert_array_equal(whole, again)


def test_clone_is_deep_and_independent(ups, gpu, make_filter):
    g = np.load(GOLDEN / "g2_mid.npz")
    path = make_filter(g["taps"], 1024, 768, 4)
    x = g["L4_in"]
    fresh = ups.StreamingUpsampler(gpu)
    assert fresh.load_filter(path)[0]
    expect = [fresh.process_block(x[i]) for i in range(3)]
    u = ups.StreamingUpsampler(gpu)
    assert u.load_filter(path)[0]
    np.testing.assert_array_equal(u.process_block(x[0]), expect[0])
    v = u.clone()  # carries the history after block 0, then lives on its own
    np.testing.assert_array_equal(u.process_block(x[1]), expect[1])
    np.testing.assert_array_equal(v.process_block(x[1]), expect[1])
    v.reset()
    np.testing.assert_array_equal(v.process_block(x[0]), expect[0])
    np.testing.assert_array_equal(u.process_block(x[2]), expect[2])  # untouched by the clone's reset
    assert v.config == u.config


# ---- staged (any-size) path agrees with the fused one --------------------------
def test_staged_path_small_and_odd_factors(ups, O, gpu, make_filter):
    rng = np.random.default_rng(3)
    for fft, T, L in [(16, 5, 1), (16, 5, 2), (16, 5, 3), (16, 5, 4), (64, 17, 3), (32, 9, 8)]:
        block = fft - (T - 1)
        if block % L:
            continue
        h = rng.standard_normal(T).astype(np.float32)
        u = ups.StreamingUpsampler(gpu)
        ok, msg = u.load_filter(make_filter(h, fft, block, L, name=f"s{fft}_{L}"))
        assert ok, msg
        nin = block // L
        x = rng.standard_normal(4 * nin).astype(np.float32)
        y = stream_blocks(u, x, nin)
        truth = O.truth_stream(x, h, L, 4, block)
        assert rel_err(y, truth) <= TOL_TRUTH


# ---- properties at full size ----------------------------------------------------
def test_full_size_properties(ups, gpu):
    path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
    filt = ups.Filter(path, device=gpu)
    cfg = filt.config
    B, L = cfg["block_size"], cfg["upsample_factor"]
    eng = ups.Engine(filt, 1, 2, ups.PCM_F32, ups.PCM_F32)
    nin, blocks = eng.in_frames, 8
    h = np.fromfile(cfg["coefficients_path"], "<f4")
    # impulse in channel 0, silence in channel 1: the output IS the filter
    x = np.zeros((blocks * nin, 2), np.float32)
    x[0, 0] = 1.0
    y = eng.process_host(x, blocks).view(np.float32).reshape(blocks * B, 2)
    assert np.abs(y[:h.size, 0] - h).max() <= 1e-6 * np.abs(h).max() * 4
    assert np.abs(y[h.size + L:, 0]).max() <= 1e-6
    assert np.abs(y[:, 1]).max() == 0.0
    # linearity: T(a + 2b) = T(a) + 2 T(b)
    a, b = real_input(1, blocks * nin * 2).reshape(-1, 2), real_input(2, blocks * nin * 2).reshape(-1, 2)
    ys = []
    for sig in (a, b, a + 2 * b):
        eng.reset()
        ys.append(eng.process_host(sig, blocks).view(np.float32).astype(np.float64))
    assert np.abs(ys[2] - (ys[0] + 2 * ys[1])).max() <= 3e-6 * np.abs(ys[2]).max()
    # DC gain: a constant input converges to L*0.99 * value (shipped design rule)
    eng.reset()
    c = np.full((blocks * nin, 2), 0.125, np.float32)
    y = eng.process_host(c, blocks).view(np.float32).reshape(blocks * B, 2)
    assert abs(y[-1000:, 0].mean() - 0.125 * h.astype(np.float64).sum() / L * 1.0) <= 1e-5


# ---- "2m" (640 001-tap) filters: the ones the selector prefers when they are present -------------------------------------
@pytest.mark.parametrize("ratio,phase,path_name", [
    (16, "min", "fused"),     # N = 2^20, K = 32768: the split fused kernel
    (8, "linear", "staged"),  # K = 65536: past the fused kernels, the staged (any-size) path
])
def test_640k_tap_filters_select_and_match_truth(ups, O, gpu, tmp_path, ratio, phase, path_name):
    """alsa_filter_selector.cpp:74-96: "2m" files (640 000 specified taps) outrank the 80 000-tap ones. Designed here with
    the repo's generator (filter_design.py, the reference recipe), selected through ResolveFilterPath next to a shorter
    filter of the same key, loaded, and two blocks of two channels checked against fp64 truth."""
    sys.path.insert(0, str(ROOT / "totton-rasp-gpu-dsp_amd"))
    import filter_design as fd

    h = fd.design(640_000, ratio, "48k", phase)
    assert h.size == 640_001
    name = fd.base_name("48k", ratio, 640_000, phase)
    assert name.endswith(f"_2m_{'min' if phase == 'min' else 'linear'}_phase")
    big = fd.export(h, tmp_path, name, ratio)
    fd.export(fd.design(8_000, ratio, "48k", phase), tmp_path, fd.base_name("48k", ratio, 8_000, phase), ratio)
    chosen, msg = ups.resolve_filter_path("", str(tmp_path), phase, ratio, 48000)
    assert chosen == str(big), (chosen, msg)
    _, taps, fft, block, L = O.read_filter(big)
    assert (taps, fft, L) == (640_001, 1 << 20, ratio) and block == fft - 640_000
    filt = ups.Filter(big, device=gpu)
    eng = ups.Engine(filt, 1, 2, ups.PCM_F32, ups.PCM_F32)
    assert eng.path == path_name
    blocks = 2
    x = real_input(640 + ratio, blocks * eng.in_frames * 2).reshape(-1, 2)
    y = eng.process_host(x, blocks).view(np.float32).reshape(blocks * block, 2)
    h32 = np.fromfile(str(big).replace(".json", ".bin"), "<f4")
    for c in range(2):
        truth = O.truth_stream(x[:, c], h32, L, blocks, block).reshape(-1)
        assert rel_err(y[:, c], truth) <= TOL_TRUTH


@pytest.mark.parametrize("ratio,channels,blocks,in_fmt,out_fmt", [
    (8, 2, 2, "f32", "f32"),   # K = 65536 = 16 x 4096
    (4, 3, 1, "s32", "s32"),   # K = 131072 = 16 x 8192, odd channel count (scalar frame assembly)
    (2, 8, 2, "s32", "s16"),   # K = 262144 = 32 x 8192 (radix-32 column passes), 8 channels (tiled frame assembly)
])
def test_640k_tap_filters_take_the_two_level_path(ups, O, gpu, tmp_path, monkeypatch, ratio, channels, blocks, in_fmt, out_fmt):
    """K = 2^16 .. 2^18 (the "2m" filters at 8x / 4x / 2x): the staged engine runs the two-level transforms
    (device/kernels_tiled.h). Against fp64 truth and against the pass-per-launch form (MIUPS_EXP_NO_TWO_LEVEL=1) on the same
    PCM over two calls (history carried)."""
    sys.path.insert(0, str(ROOT / "totton-rasp-gpu-dsp_amd"))
    import filter_design as fd

    h = fd.design(640_000, ratio, "48k", "linear")
    path = fd.export(h, tmp_path, fd.base_name("48k", ratio, 640_000, "linear"), ratio)
    _, taps, fft, block, L = O.read_filter(path)
    h32 = np.fromfile(str(path).replace(".json", ".bin"), "<f4")
    filt = ups.Filter(path, device=gpu)
    fmt = {"f32": ups.PCM_F32, "s32": ups.PCM_S32, "s16": ups.PCM_S16}
    nin = block // L
    rng = np.random.default_rng(ratio)
    xf = np.clip(rng.standard_normal((2, blocks * nin, channels)) * 0.1, -1, 1)
    if in_fmt == "s32":
        xi = (xf * 2147483647).astype("<i4")
        x, calls = xi.astype(np.float64) / 2147483648.0, [xi[0], xi[1]]
    else:
        x32 = xf.astype(np.float32)
        x, calls = x32.astype(np.float64), [x32[0], x32[1]]
    dt = {"f32": np.float32, "s32": "<i4", "s16": "<i2"}[out_fmt]
    scale = {"f32": 1.0, "s32": 2147483648.0, "s16": 32768.0}[out_fmt]

    def run(two_level):
        eng = ups.Engine(filt, 1, channels, fmt[in_fmt], fmt[out_fmt])
        assert eng.path == "staged"
        outs = []
        for k in range(2):
            outs.append(eng.process_host(calls[k], blocks).view(dt).reshape(blocks * block, channels).astype(np.float64) / scale)
            assert eng.last_two_level == two_level
        eng.close()
        return np.concatenate(outs)

    y = run(True)
    monkeypatch.setenv("MIUPS_EXP_NO_TWO_LEVEL", "1")
    y_old = run(False)
    lsb = 0.0 if out_fmt == "f32" else 1.0 / scale
    for c in (0, channels - 1):
        xs = np.concatenate([x[0, :, c], x[1, :, c]])
        truth = O.truth_stream(xs, h32, L, 2 * blocks, block).reshape(-1)
        if out_fmt != "f32":
            truth = np.clip(truth, -1.0, float(np.float32(0.9999999)))
        tol = lsb + 1e-5 * np.abs(truth).max()
        assert np.abs(y[:, c] - truth).max() <= tol
        assert np.abs(y[:, c] - y_old[:, c]).max() <= 2 * tol


def test_two_level_path_with_eq_folded_in(ups, O, gpu, tmp_path):
    """mi_filter_set_eq on a 640k-tap filter: the two-level path's own copies of the spectrum tables ([k1][k2] order) are
    rebuilt with the EQ folded into the FIR. 8x, two channels, two calls, against the fp64 true streaming convolution with
    the folded FIR (the same truth the fused path's EQ tests use); and back to the plain filter afterwards."""
    import json

    sys.path.insert(0, str(ROOT / "totton-rasp-gpu-dsp_amd"))
    import filter_design as fd

    ratio, channels, blocks = 8, 2, 2
    h = fd.design(640_000, ratio, "48k", "linear")
    path = fd.export(h, tmp_path, fd.base_name("48k", ratio, 640_000, "linear"), ratio)
    _, taps, fft, block, L = O.read_filter(path)
    h32 = np.fromfile(str(path).replace(".json", ".bin"), "<f4")
    eq_text = json.loads((ROOT / "tests" / "golden" / "g4_eq_profiles.json").read_text())["opra10"]
    fs_out = 48000.0 * ratio
    filt = ups.Filter(path, device=gpu)
    filt.set_eq(eq_text, fs_out)
    eng = ups.Engine(filt, 1, channels, ups.PCM_F32, ups.PCM_F32)
    nin = block // L
    x = (np.random.default_rng(77).standard_normal((2 * blocks * nin, channels)) * 0.1).astype(np.float32)
    y = np.concatenate([eng.process_host(x[k * blocks * nin:(k + 1) * blocks * nin], blocks).view(np.float32)
                        .reshape(blocks * block, channels) for k in range(2)])
    assert eng.path == "staged" and eng.last_two_level
    fir = O.eq_fold_fir(h32, eq_text, fs_out)
    assert filt.eq_residual()["over_limit"] == 0
    for c in range(channels):
        truth = O.truth_stream(x[:, c].astype(np.float64), fir, L, 2 * blocks, block).reshape(-1)
        assert rel_err(y[:, c], truth) <= TOL_TRUTH
    filt.set_eq("", fs_out)  # plain again: the next call takes the new tables (history restarted for a clean comparison)
    eng.reset()
    y0 = eng.process_host(x[:blocks * nin], blocks).view(np.float32).reshape(blocks * block, channels)
    truth0 = O.truth_stream(x[:blocks * nin, 0].astype(np.float64), h32, L, blocks, block).reshape(-1)
    assert rel_err(y0[:, 0], truth0) <= TOL_TRUTH
