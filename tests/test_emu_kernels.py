"""CPU pre-flight of the HIP kernels: the product's device headers compiled with
-DMIUPS_HOST_EMU (tests/emu/emu_driver.cpp: one OS thread per GPU thread, real
barriers, exact-size buffers) under AddressSanitizer + UBSan, compared with fp64
truth. This is how kernel indexing is checked before a kernel ever runs on a
GPU box; it is test infrastructure, not a product code path (the product library
contains no CPU compute path -- see test_host_logic.py)."""
from __future__ import annotations

import subprocess

import numpy as np
import pytest

from conftest import ROOT, rel_err

EMU_DIR = ROOT / "tests" / "emu"
EMU_BIN = ROOT / "tests" / "_emu" / "emu_driver"
FMT = {"f32": 0, "s16": 1, "s24": 2, "s32": 3}


@pytest.fixture(scope="session")
def emu():
    r = subprocess.run(["make", "-j8", "-C", str(EMU_DIR)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return EMU_BIN


def run_emu(emu, tmp_path, json_path, x_bytes, streams, channels, blocks, calls, path, in_fmt="f32", out_fmt="f32",
            flags=0):
    (tmp_path / "in.bin").write_bytes(x_bytes)
    r = subprocess.run([str(emu), str(json_path), str(flags), str(streams), str(channels), str(FMT[in_fmt]),
                        str(FMT[out_fmt]), str(blocks), str(calls), str(tmp_path / "in.bin"), str(tmp_path / "out.bin"),
                        path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]  # ASan/UBSan reports land here
    return (tmp_path / "out.bin").read_bytes()


CASES = [
    # fft, taps, L, path, streams, channels, blocks, calls
    (16, 5, 1, "staged", 1, 1, 2, 2),
    (16, 5, 2, "staged", 1, 2, 2, 2),
    (16, 5, 3, "staged", 1, 1, 2, 2),     # factor that does not divide fft: P = 1, S = 3
    (64, 17, 1, "fused", 1, 1, 3, 2),     # K = 32: one thread
    (128, 33, 1, "fused", 1, 2, 2, 2),    # K = 64
    (256, 65, 2, "fused", 2, 2, 2, 2),
    (512, 130, 1, "fused", 1, 2, 2, 2),   # odd history length
    (1024, 257, 1, "fused", 1, 1, 2, 2),
    (1024, 257, 4, "fused", 2, 3, 3, 2),  # odd channel count, chunked launch
    (1024, 257, 16, "fused", 1, 2, 4, 1),
    (1024, 257, 4, "staged", 2, 2, 2, 2),
    (4096, 1025, 2, "fused", 1, 8, 1, 2),  # 8 channels: planar input, one channel per workgroup, interleave kernel
    (4096, 1025, 8, "fused", 1, 12, 2, 1),  # 12 channels
    (8192, 2049, 1, "fused", 1, 1, 1, 2),  # K = 4096 (radices 16,16,16)
    (16384, 4097, 1, "fused", 1, 2, 1, 2),  # K = 8192 = 2 * 16^3, stereo, whole-frame epilogue
    (32768, 8193, 1, "fused", 1, 1, 1, 1),  # K = 16384 = 4 * 16^3
]
PRUNED_CASES = [  # history >= half the transform (O/N >= 1/2, like every shipped filter): pruned last inverse pass
    (1024, 641, 1, "fused", 1, 2, 2, 2),    # K = 512 = 2 * 16^2: radix-2 last pass
    (2048, 1281, 1, "fused", 1, 1, 2, 2),   # K = 1024 = 4 * 16^2: radix 4
    (1024, 641, 4, "fused", 1, 2, 2, 2),    # K = 128 = 8 * 16: radix 8, four phases
    (512, 321, 1, "fused", 2, 2, 2, 2),     # K = 256 = 16^2: radix 16, two butterflies per thread
    (8192, 5121, 1, "fused", 1, 1, 1, 2),   # K = 4096 = 16^3 (the 16x geometry's transform length)
    (1024, 642, 1, "fused", 1, 1, 2, 2),    # odd history length: the unpruned scalar plane path stays in charge
]
R32_CASES = [  # the radix-32 pass plan (experiment switch MIUPS_EXP_R32; kernel_fused.h FusedCfg)
    (16384, 4097, 1, "fused", 1, 2, 1, 2),  # K = 8192: passes 16, 32, 16
    (65536, 16385, 4, "fused", 1, 1, 2, 1),  # K = 8192 with four phases
    (32768, 8193, 1, "fused", 1, 1, 1, 2),  # K = 16384: passes 32, 32, 16
    (32768, 8192, 1, "fused", 1, 2, 1, 1),  # K = 16384, odd history length (plane_write without the even-Oc fast path)
]
NARROW_CASES = [
    (2048, 600, 1, 1, 2, 2, 2),   # K = 1024 = 4*16*16: one wave
    (8192, 2001, 2, 2, 2, 3, 1),  # K = 2048 = 8*16*16, stereo
    (4096, 1024, 1, 1, 1, 2, 2),  # K = 2048, odd history length (taps - 1 = 1023)
    (8192, 2049, 1, 1, 1, 1, 2),  # K = 4096 = 16^3
    (16384, 4097, 1, 1, 2, 1, 2), # K = 8192 = 2*16^3
]


@pytest.mark.parametrize("fft,taps,L,streams,channels,blocks,calls", NARROW_CASES)
def test_emulated_narrow_form(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, streams, channels, blocks, calls):
    """The one-butterfly-per-thread experiment form (K >= 1024: each mirror pair's two sets in lanes l and l ^ 32, traded
    through the lane exchange; kernel_fused.h "narrow form") against fp64 truth."""
    monkeypatch.setenv("EMU_NARROW", "1")
    rng = np.random.default_rng(fft + L)
    h = rng.standard_normal(taps).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin = block // L
    x = rng.standard_normal((calls, streams, blocks * nin, channels)).astype(np.float32)
    out = run_emu(emu, tmp_path, p, x.tobytes(), streams, channels, blocks, calls, "fused")
    y = np.frombuffer(out, np.float32).reshape(calls, streams, blocks * block, channels)
    for s_ in range(streams):
        for c in range(channels):
            xs = np.concatenate([x[k, s_, :, c] for k in range(calls)])
            truth = O.truth_stream(xs, h, L, calls * blocks, block).reshape(-1)
            got = np.concatenate([y[k, s_, :, c] for k in range(calls)])
            assert np.abs(got - truth).max() <= 1e-5 * np.abs(truth).max()


@pytest.mark.parametrize("fft,taps,L,path,streams,channels,blocks,calls", PRUNED_CASES)
def test_emulated_pruned_last_pass(emu, O, make_filter, tmp_path, fft, taps, L, path, streams, channels, blocks, calls):
    test_emulated_kernels_match_truth(emu, O, make_filter, tmp_path, fft, taps, L, path, streams, channels, blocks, calls)


TILED_CASES = [  # the two-level path (device/kernels_tiled.h): K = 2^15 .. 2^18 = K1 x M2, M2-point rows in LDS
    # fft, taps, L, streams, channels, blocks, calls, in_fmt, out_fmt
    (65536, 20001, 1, 1, 1, 1, 2, "f32", "f32"),    # K = 32768 = 16 x 2048 (rows 8.16.16), history carried to a second call
    (131072, 40003, 2, 1, 2, 2, 1, "s32", "s32"),   # K = 32768, two phases, stereo PCM, odd history length, chunked pairs
    (131072, 70001, 1, 2, 1, 1, 1, "f32", "s16"),   # K = 65536 = 16 x 4096 (rows 16.16.16), two streams
    (262144, 150001, 1, 1, 1, 1, 1, "f32", "f32"),  # K = 131072 = 16 x 8192 (rows 2.16.16.16, 512 threads)
    (1048576, 640001, 2, 1, 1, 1, 1, "s32", "f32"),  # K = 262144 = 32 x 8192 (radix-32 column passes): the 2x "2m" geometry
]


@pytest.mark.parametrize("fft,taps,L,streams,channels,blocks,calls,in_fmt,out_fmt", TILED_CASES)
def test_emulated_two_level_path(emu, O, make_filter, tmp_path, fft, taps, L, streams, channels, blocks, calls, in_fmt, out_fmt):
    """tiled_load / tiled_row_forward / tiled_row_inverse / tiled_store + interleave: against fp64 truth, and against the
    pass-per-launch staged kernels on the same input (same tables, same arithmetic up to the order of the butterflies)."""
    rng = np.random.default_rng(fft + taps)
    h = (rng.standard_normal(taps) * 0.5 / np.sqrt(taps / L)).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin = block // L
    xf = np.clip(rng.standard_normal((calls, streams, blocks * nin, channels)) * 0.1, -1, 1)
    if in_fmt == "s32":
        xi = (xf * 2147483647).astype("<i4")
        x, raw = xi.astype(np.float64) / 2147483648.0, xi.tobytes()
    else:
        x32 = xf.astype(np.float32)
        x, raw = x32.astype(np.float64), x32.tobytes()
    tiled = run_emu(emu, tmp_path, p, raw, streams, channels, blocks, calls, "tiled", in_fmt, out_fmt)
    staged = run_emu(emu, tmp_path, p, raw, streams, channels, blocks, calls, "staged", in_fmt, out_fmt)
    dt = {"f32": np.float32, "s32": "<i4", "s16": "<i2"}[out_fmt]
    scale = {"f32": 1.0, "s32": 2147483648.0, "s16": 32768.0}[out_fmt]
    y = np.frombuffer(tiled, dt).reshape(calls, streams, blocks * block, channels).astype(np.float64) / scale
    ys = np.frombuffer(staged, dt).reshape(calls, streams, blocks * block, channels).astype(np.float64) / scale
    lsb = 0.0 if out_fmt == "f32" else 1.0 / scale
    for s_ in range(streams):
        for c in range(channels):
            xs = np.concatenate([x[k, s_, :, c] for k in range(calls)])
            truth = np.clip(O.truth_stream(xs, h, L, calls * blocks, block).reshape(-1), -1.0, float(np.float32(0.9999999))
                            ) if out_fmt != "f32" else O.truth_stream(xs, h, L, calls * blocks, block).reshape(-1)
            got = np.concatenate([y[k, s_, :, c] for k in range(calls)])
            ref = np.concatenate([ys[k, s_, :, c] for k in range(calls)])
            tol = lsb + 1e-5 * np.abs(truth).max()
            assert np.abs(got - truth).max() <= tol
            assert np.abs(got - ref).max() <= 2 * tol


PARTS_CASES = [  # the engine's small-call form: EMU_PARTS workgroups per (block, stream, channel), P / parts phases each
    # fft, taps, L, parts, streams, channels, blocks, calls      (K = fft / 2L >= 1024: fused_parts_kernel exists)
    (8192, 2049, 4, 4, 1, 1, 1, 2),    # K = 1024, mono, one phase per workgroup (the reference's call shape)
    (8192, 5121, 4, 2, 1, 2, 2, 2),    # stereo out of 16-byte frame loads, two phases per workgroup, pruned last pass
    (32768, 8193, 16, 8, 2, 3, 1, 1),  # K = 1024, P = 16, planar input (3 channels), two streams
    (16384, 4099, 2, 2, 1, 2, 1, 2),   # K = 4096, odd history length (Oc = 2049)
]


@pytest.mark.parametrize("fft,taps,L,parts,streams,channels,blocks,calls", PARTS_CASES)
def test_emulated_phase_split_small_calls(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, parts, streams, channels,
                                          blocks, calls):
    """fused_parts_kernel (kernel_fused.h run<..., PARTS>): bit-identical to the plain form with the same channel group
    width and external epilogue (EMU_CG=1), and both against fp64 truth."""
    rng = np.random.default_rng(fft + L + parts)
    h = rng.standard_normal(taps).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin = block // L
    x = rng.standard_normal((calls, streams, blocks * nin, channels)).astype(np.float32)
    monkeypatch.setenv("EMU_PARTS", str(parts))
    split = run_emu(emu, tmp_path, p, x.tobytes(), streams, channels, blocks, calls, "fused")
    monkeypatch.delenv("EMU_PARTS")
    plain = run_emu(emu, tmp_path, p, x.tobytes(), streams, channels, blocks, calls, "fused")
    assert split == plain
    y = np.frombuffer(split, np.float32).reshape(calls, streams, blocks * block, channels)
    for s_ in range(streams):
        for c in range(channels):
            xs = np.concatenate([x[k, s_, :, c] for k in range(calls)])
            truth = O.truth_stream(xs, h, L, calls * blocks, block).reshape(-1)
            got = np.concatenate([y[k, s_, :, c] for k in range(calls)])
            assert np.abs(got - truth).max() <= 1e-5 * np.abs(truth).max()


@pytest.mark.parametrize("fft,taps,L,parts,channels", [
    (4096, 1025, 1, 2, 1),   # K = 2048 as two 1024-point halves; P = 1: the two halves of the one phase on two workgroups
    (8192, 2049, 2, 4, 2),   # P = 2: four workgroups, one half transform each; stereo
    (8192, 2049, 2, 2, 1),   # ... two workgroups, one phase (both halves) each
])
def test_emulated_split_form_small_calls(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, parts, channels):
    """fused_split_parts_kernel: the split form's 2P half transforms of a channel-block spread over `parts` workgroups,
    bit-identical to fused_split_kernel with the same group width, and against fp64 truth."""
    monkeypatch.setenv("EMU_SPLIT", "1")
    rng = np.random.default_rng(fft + parts)
    h = rng.standard_normal(taps).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin = block // L
    blocks, calls = 2, 2
    x = rng.standard_normal((calls, 1, blocks * nin, channels)).astype(np.float32)
    monkeypatch.setenv("EMU_PARTS", str(parts))
    split = run_emu(emu, tmp_path, p, x.tobytes(), 1, channels, blocks, calls, "fused")
    monkeypatch.delenv("EMU_PARTS")
    monkeypatch.setenv("EMU_CG", "1")
    plain = run_emu(emu, tmp_path, p, x.tobytes(), 1, channels, blocks, calls, "fused")
    assert split == plain
    y = np.frombuffer(split, np.float32).reshape(calls, 1, blocks * block, channels)
    for c in range(channels):
        xs = np.concatenate([x[k, 0, :, c] for k in range(calls)])
        truth = O.truth_stream(xs, h, L, calls * blocks, block).reshape(-1)
        got = np.concatenate([y[k, 0, :, c] for k in range(calls)])
        assert np.abs(got - truth).max() <= 1e-5 * np.abs(truth).max()


@pytest.mark.parametrize("fft,taps,L,path,streams,channels,blocks,calls", R32_CASES)
def test_emulated_radix32_plan(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, path, streams, channels, blocks,
                               calls):
    monkeypatch.setenv("EMU_R32", "1")
    test_emulated_kernels_match_truth(emu, O, make_filter, tmp_path, fft, taps, L, path, streams, channels, blocks, calls)


@pytest.mark.parametrize("fft,taps,L,path,streams,channels,blocks,calls", CASES)
def test_emulated_kernels_match_truth(emu, O, make_filter, tmp_path, fft, taps, L, path, streams, channels, blocks, calls):
    rng = np.random.default_rng(fft * 31 + L)
    h = rng.standard_normal(taps).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin = block // L
    x = rng.standard_normal((calls, streams, blocks * nin, channels)).astype(np.float32)
    y = np.frombuffer(run_emu(emu, tmp_path, p, x.tobytes(), streams, channels, blocks, calls, path), np.float32)
    y = y.reshape(calls, streams, blocks * block, channels)
    for s in range(streams):
        for c in range(channels):
            truth = O.truth_stream(x[:, s, :, c].reshape(-1), h, L, calls * blocks, block).reshape(-1)
            assert rel_err(y[:, s, :, c].reshape(-1), truth) <= 1e-5


@pytest.mark.parametrize("in_fmt,out_fmt", [("s32", "s32"), ("s16", "s16"), ("s24", "s24"), ("s16", "f32"), ("f32", "s24")])
def test_emulated_pcm_formats(emu, O, make_filter, tmp_path, in_fmt, out_fmt):
    rng = np.random.default_rng(3)
    fft, taps, L = 1024, 257, 4
    h = (rng.standard_normal(taps) * 0.05).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks, ch = block // L, 3, 2
    xf = np.clip(rng.standard_normal((blocks * nin, ch)) * 0.2, -1, 1).astype(np.float32)
    raw = xf.tobytes() if in_fmt == "f32" else O.float_to_pcm(xf.reshape(-1), in_fmt).tobytes()
    xin = xf if in_fmt == "f32" else O.pcm_to_float(np.frombuffer(raw, np.uint8), in_fmt).reshape(-1, ch)
    out = run_emu(emu, tmp_path, p, raw, 1, ch, blocks, 1, "fused", in_fmt, out_fmt)
    # same kernels with float output give the pre-conversion values: the PCM bytes
    # must equal the oracle's conversion of exactly those floats
    yf = np.frombuffer(run_emu(emu, tmp_path, p, raw, 1, ch, blocks, 1, "fused", in_fmt, "f32"), np.float32)
    if out_fmt == "f32":
        got = np.frombuffer(out, np.float32)
        np.testing.assert_array_equal(got, yf)
    else:
        np.testing.assert_array_equal(np.frombuffer(out, np.uint8), O.float_to_pcm(yf, out_fmt))
    for c in range(ch):
        truth = O.truth_stream(xin[:, c], h, L, blocks, block).reshape(-1)
        assert rel_err(yf.reshape(-1, ch)[:, c], truth) <= 1e-5


@pytest.mark.parametrize("in_fmt,channels,streams", [("s16", 3, 2), ("s24", 4, 1), ("s32", 8, 1), ("f32", 5, 2)])
def test_emulated_planarized_input(emu, O, make_filter, tmp_path, in_fmt, channels, streams):
    """More than two channels: planarize_kernel converts history ++ new frames to one fp32
    timeline per channel before the fused kernel runs; two calls so that the second one
    reads real history through it."""
    rng = np.random.default_rng(channels)
    fft, taps, L = 1024, 301, 4   # block 724 -> 181 frames per block: odd, so unaligned timelines too
    h = (rng.standard_normal(taps) * 0.05).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks, calls = block // L, 3, 2
    xf = np.clip(rng.standard_normal((calls, streams, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    if in_fmt == "f32":
        raw, xin = xf.tobytes(), xf
    else:
        pcm = O.float_to_pcm(xf.reshape(-1), in_fmt)
        raw = pcm.tobytes()
        xin = O.pcm_to_float(np.frombuffer(raw, np.uint8), in_fmt).reshape(xf.shape)
    y = np.frombuffer(run_emu(emu, tmp_path, p, raw, streams, channels, blocks, calls, "fused", in_fmt, "f32"), np.float32)
    y = y.reshape(calls, streams, blocks * block, channels)
    for s in range(streams):
        for c in range(channels):
            truth = O.truth_stream(xin[:, s, :, c].reshape(-1), h, L, calls * blocks, block).reshape(-1)
            assert rel_err(y[:, s, :, c].reshape(-1), truth) <= 1e-5


@pytest.mark.parametrize("channels,cg,out_fmt", [(8, 4, "f32"), (8, 4, "s32"), (8, 2, "f32"), (12, 4, "s32"), (2, 1, "s32")])
def test_emulated_narrow_channel_groups(emu, O, make_filter, tmp_path, monkeypatch, channels, cg, out_fmt):
    """The kernel's own epilogue also supports groups narrower than a frame (it then writes
    cg-sample runs into frames that other workgroups complete); the engine uses the
    external interleave kernels for those instead (next test)."""
    monkeypatch.setenv("EMU_CG", str(cg))
    monkeypatch.setenv("EMU_INKERNEL", "1")
    rng = np.random.default_rng(channels * 10 + cg)
    fft, taps, L = 1024, 257, 4
    h = (rng.standard_normal(taps) * 0.05).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks = block // L, 3
    x = np.clip(rng.standard_normal((1, 1, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    out = run_emu(emu, tmp_path, p, x.tobytes(), 1, channels, blocks, 1, "fused", "f32", out_fmt)
    y = (np.frombuffer(out, np.float32) if out_fmt == "f32" else O.pcm_to_float(np.frombuffer(out, np.uint8), out_fmt))
    y = y.reshape(blocks * block, channels)
    for c in range(channels):
        truth = O.truth_stream(x[0, 0, :, c], h, L, blocks, block).reshape(-1)
        assert np.abs(y[:, c] - truth).max() <= 1e-5 * np.abs(truth).max() + (0 if out_fmt == "f32" else 2.0**-31)


@pytest.mark.parametrize("channels,cg,L,taps,out_fmt", [
    (8, 1, 4, 257, "s32"),    # vector interleave (P*C % 4 == 0, 192 frames per phase)
    (6, 1, 2, 257, "f32"),    # 5.1: runs of four straddle phases
    (2, 1, 4, 257, "s32"),    # stereo call too small to fill the GPU: one channel per workgroup
    (3, 1, 4, 257, "f32"),    # P*C % 4 == 0 with odd channel count
    (3, 1, 2, 257, "f32"),    # P*C = 6: scalar interleave
    (8, 2, 4, 261, "s32"),    # 191 frames per phase: scalar interleave; two-channel groups
    (4, 1, 4, 257, "s16"),    # narrow PCM: scalar interleave
    (5, 1, 8, 257, "s24"),
])
def test_emulated_external_interleave(emu, O, make_filter, tmp_path, monkeypatch, channels, cg, L, taps, out_fmt):
    """Groups narrower than a frame: the fused kernel stops at the staging planes and
    interleave_quad_kernel / interleave_scalar_kernel write the PCM frames, chunk by chunk."""
    monkeypatch.setenv("EMU_CG", str(cg))
    rng = np.random.default_rng(channels * 100 + L)
    fft = 1024
    h = (rng.standard_normal(taps) * 0.05).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks, streams = block // L, 3, 2
    x = np.clip(rng.standard_normal((1, streams, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    out = run_emu(emu, tmp_path, p, x.tobytes(), streams, channels, blocks, 1, "fused", "f32", out_fmt)
    yf = np.frombuffer(run_emu(emu, tmp_path, p, x.tobytes(), streams, channels, blocks, 1, "fused", "f32", "f32"),
                       np.float32)
    if out_fmt == "f32":
        np.testing.assert_array_equal(np.frombuffer(out, np.float32), yf)
    else:  # the PCM bytes are the oracle's conversion of exactly the float results
        np.testing.assert_array_equal(np.frombuffer(out, np.uint8), O.float_to_pcm(yf, out_fmt))
    yf = yf.reshape(streams, blocks * block, channels)
    for s in range(streams):
        for c in range(channels):
            truth = O.truth_stream(x[0, s, :, c], h, L, blocks, block).reshape(-1)
            assert rel_err(yf[s, :, c], truth) <= 1e-5


@pytest.mark.parametrize("fft,taps,L,channels,cg,out_fmt", [
    # frames per phase % 4 == 0: register-transposed form (epilogue_quad)
    (16384, 4001, 4, 8, 8, "s32"),    # R = 32 planes
    (32768, 8001, 8, 8, 4, "f32"),    # narrow group: 16-byte runs inside wider frames
    (65536, 16001, 16, 2, 2, "s32"),  # cg == channels == 2: vectors straddle phases
    (16384, 4001, 4, 16, 8, "f32"),   # two groups per frame
    # 3095 frames per phase: LDS-tiled form (epilogue_tiled), last tile partial
    (16384, 4005, 4, 8, 8, "s32"),
    (32768, 8009, 8, 8, 4, "f32"),
    (65536, 16017, 16, 2, 2, "s32"),
    (16384, 4005, 4, 16, 8, "f32"),
])
def test_emulated_wide_epilogue(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, channels, cg, out_fmt):
    """More than 16 staging planes per workgroup: the epilogue transposes 4x4 blocks in
    registers (FusedKernel::epilogue_quad) or [planes][64] tiles through LDS
    (FusedKernel::epilogue_tiled)."""
    monkeypatch.setenv("EMU_CG", str(cg))
    monkeypatch.setenv("EMU_INKERNEL", "1")
    rng = np.random.default_rng(fft + channels)
    h = (rng.standard_normal(taps) * 0.02).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks = block // L, 2
    x = np.clip(rng.standard_normal((1, 1, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    out = run_emu(emu, tmp_path, p, x.tobytes(), 1, channels, blocks, 1, "fused", "f32", out_fmt)
    y = (np.frombuffer(out, np.float32) if out_fmt == "f32" else O.pcm_to_float(np.frombuffer(out, np.uint8), out_fmt))
    y = y.reshape(blocks * block, channels)
    for c in range(channels):
        truth = O.truth_stream(x[0, 0, :, c], h, L, blocks, block).reshape(-1)
        assert np.abs(y[:, c] - truth).max() <= 1e-5 * np.abs(truth).max() + (0 if out_fmt == "f32" else 2.0**-31)


@pytest.mark.parametrize("fft,taps,L,streams,channels,in_fmt,out_fmt", [
    (4096, 1025, 1, 1, 1, "f32", "f32"),    # half length 1024 (radices 4,16,16): 32 threads, 17 of them self lanes
    (8192, 2049, 2, 2, 2, "s32", "s32"),    # half length 1024, two phases, stereo vector loads
    (16384, 4097, 2, 1, 3, "s16", "s24"),   # half length 2048 (radices 8,16,16), planar input, scalar interleave
    (16384, 4097, 1, 1, 1, "f32", "s32"),   # half length 4096 (radices 16,16,16)
    (32768, 8193, 4, 1, 1, "f32", "f32"),   # half length 2048, four phases
    (32768, 8193, 1, 1, 2, "s32", "f32"),   # half length 8192 (radices 2,16,16,16): 256 threads
    # history >= half the transform (like the shipped 2x filters): pruned last inverse pass of every half transform
    (4096, 2561, 1, 1, 1, "f32", "f32"),    # half length 1024, radix-4 last pass
    (16384, 10241, 2, 1, 2, "s32", "s32"),  # half length 2048, radix 8, two phases, stereo
    (32768, 20481, 1, 1, 1, "f32", "f32"),  # half length 8192, radix 2
])
def test_emulated_split_form(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, streams, channels, in_fmt, out_fmt):
    """fused_split_kernel: the block transform is twice the LDS transform length (the product
    uses it for K = 32768, the 2x filters); forced here at the smaller sizes it covers
    (half length >= 1024), two calls."""
    monkeypatch.setenv("EMU_SPLIT", "1")
    rng = np.random.default_rng(fft + L)
    h = (rng.standard_normal(taps) * 0.01).astype(np.float32)  # keeps the PCM outputs inside [-1, 1)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks, calls = block // L, 2, 2
    xf = np.clip(rng.standard_normal((calls, streams, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    if in_fmt == "f32":
        raw, xin = xf.tobytes(), xf
    else:
        raw = O.float_to_pcm(xf.reshape(-1), in_fmt).tobytes()
        xin = O.pcm_to_float(np.frombuffer(raw, np.uint8), in_fmt).reshape(xf.shape)
    out = run_emu(emu, tmp_path, p, raw, streams, channels, blocks, calls, "fused", in_fmt, out_fmt)
    y = np.frombuffer(out, np.float32) if out_fmt == "f32" else O.pcm_to_float(np.frombuffer(out, np.uint8), out_fmt)
    y = y.reshape(calls, streams, blocks * block, channels)
    lsb = {"f32": 0.0, "s16": 2.0**-15, "s24": 2.0**-23, "s32": 2.0**-31}[out_fmt]
    for s in range(streams):
        for c in range(channels):
            truth = O.truth_stream(xin[:, s, :, c].reshape(-1), h, L, calls * blocks, block).reshape(-1)
            assert np.abs(y[:, s, :, c].reshape(-1) - truth).max() <= 1e-5 * np.abs(truth).max() + lsb


@pytest.mark.parametrize("fft,taps,L,channels,out_fmt,tiled", [
    (8192, 2113, 4, 8, "s32", True),     # 32 rows, Bc = 1520: the last 64-wide tile is partial
    (4096, 1025, 16, 16, "f32", True),   # 256 rows: 32-wide tiles
    (4096, 1025, 8, 12, "s32", True),    # 96 rows (not a power of two)
    (4096, 1025, 2, 8, "s32", False),    # the quad form where the tiled one would be chosen
])
def test_emulated_interleave_kernels_for_wide_frames(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, channels,
                                                     out_fmt, tiled):
    """interleave_tiled_kernel (LDS-tiled staging planes -> frames, the form the engine takes for 16 or more planes per
    frame group) and, with EMU_NO_TILED_INTERLEAVE (= the engine's MIUPS_EXP_NO_TILED_INTERLEAVE), the quad form on the
    same shapes: every channel against fp64 truth, which also pins the row order (phase-major, channel-minor)."""
    if not tiled:
        monkeypatch.setenv("EMU_NO_TILED_INTERLEAVE", "1")
    rng = np.random.default_rng(fft + L + channels)
    h = (rng.standard_normal(taps) * 0.01).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks = block // L, 2
    x = np.clip(rng.standard_normal((1, 1, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    out = run_emu(emu, tmp_path, p, x.tobytes(), 1, channels, blocks, 1, "fused", "f32", out_fmt)
    y = np.frombuffer(out, np.float32) if out_fmt == "f32" else O.pcm_to_float(np.frombuffer(out, np.uint8), out_fmt)
    y = y.reshape(blocks * block, channels)
    lsb = 0.0 if out_fmt == "f32" else 2.0**-31
    for c in range(channels):
        truth = O.truth_stream(x[0, 0, :, c], h, L, blocks, block).reshape(-1)
        assert np.abs(y[:, c] - truth).max() <= 1e-5 * np.abs(truth).max() + lsb


def test_emulated_split_form_with_parked_second_half(emu, O, make_filter, tmp_path, monkeypatch):
    """The split form's experiment switch MIUPS_EXP_PARK (here EMU_PARK): a phase's second half transform takes its
    first-pass inputs from a global-memory parking area instead of recomputing the spectral stage (measured slower on the
    GPU, profiles/r03_g_split_park.txt; kept behind the switch, and kept correct)."""
    monkeypatch.setenv("EMU_PARK", "1")
    test_emulated_split_form(emu, O, make_filter, tmp_path, monkeypatch, 16384, 10241, 2, 1, 2, "s32", "s32")
    test_emulated_split_form(emu, O, make_filter, tmp_path, monkeypatch, 4096, 1025, 1, 1, 1, "f32", "f32")


@pytest.mark.parametrize("fft,taps,L,channels,in_fmt", [(8192, 2049, 2, 2, "s32"), (16384, 4097, 2, 3, "s16")])
def test_emulated_split_form_from_interleaved_input(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, channels, in_fmt):
    """The split form's other input path (EMU_NO_SPLIT_PLANAR = the engine's MIUPS_EXP_NO_SPLIT_PLANAR, and what the
    engine falls back to when block windows do not start at multiples of four samples): stereo vector loads straight
    from the caller's PCM / the plain planar timeline, every second complex word per transform half."""
    monkeypatch.setenv("EMU_SPLIT", "1")
    monkeypatch.setenv("EMU_NO_SPLIT_PLANAR", "1")
    rng = np.random.default_rng(fft + L + 1)
    h = (rng.standard_normal(taps) * 0.01).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin, blocks, calls = block // L, 2, 2
    xf = np.clip(rng.standard_normal((calls, 1, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    raw = O.float_to_pcm(xf.reshape(-1), in_fmt).tobytes()
    xin = O.pcm_to_float(np.frombuffer(raw, np.uint8), in_fmt).reshape(xf.shape)
    out = run_emu(emu, tmp_path, p, raw, 1, channels, blocks, calls, "fused", in_fmt, "f32")
    y = np.frombuffer(out, np.float32).reshape(calls, 1, blocks * block, channels)
    for c in range(channels):
        truth = O.truth_stream(xin[:, 0, :, c].reshape(-1), h, L, calls * blocks, block).reshape(-1)
        assert np.abs(y[:, 0, :, c].reshape(-1) - truth).max() <= 1e-5 * np.abs(truth).max()


def test_emulated_fused_and_staged_agree_on_real_geometry(emu, O, tmp_path):
    """44k 4x shipped filter, one stereo block through both kernel families."""
    path = ROOT / "tests" / "golden" / "filters" / "filter_44k_4x_80000_min_phase.json"
    h, taps, fft, block, L = O.read_filter(path)
    nin = block // L
    x = (np.random.default_rng(8).standard_normal((nin, 2)) * 0.2).astype(np.float32)
    yf = np.frombuffer(run_emu(emu, tmp_path, path, x.tobytes(), 1, 2, 1, 1, "fused"), np.float32).reshape(-1, 2)
    ys = np.frombuffer(run_emu(emu, tmp_path, path, x.tobytes(), 1, 2, 1, 1, "staged"), np.float32).reshape(-1, 2)
    for c in range(2):
        truth = O.truth_stream(x[:, c], h, L, 1, block).reshape(-1)
        assert rel_err(yf[:, c], truth) <= 1e-5
        assert rel_err(ys[:, c], truth) <= 1e-5


@pytest.mark.parametrize("fft,taps,L,streams,channels,blocks,out_fmt,cap", [
    (16384, 4097, 4, 1, 8, 4, "s32", None),   # K = 2048 (64 threads), 32 rows, 64-wide tiles, 8 words per thread
    (16384, 4097, 4, 2, 8, 3, "f32", 4),      # two streams, odd pair count (chunks of 3 + 3 pairs), a small cap: most tiles
                                              # are left to the frame pass
    (32768, 8193, 4, 1, 8, 2, "s32", 1000),   # K = 4096 (128 threads), a cap nobody reaches: a pair's last workgroup takes all
    (32768, 8449, 2, 1, 8, 2, "s32", None),   # K = 8192 (256 threads), 16 rows: one word per thread; Bc = 12160: partial last tile
])
def test_emulated_cooperative_frames(emu, O, make_filter, tmp_path, monkeypatch, fft, taps, L, streams, channels, blocks,
                                     out_fmt, cap):
    """Cooperative frames (device/frame_tile.h, EMU_COOP = the engine's default for these shapes): the transform kernel's
    workgroups report their planes, assemble tiles of complete pairs up to their cap, and the frame pass behind the kernel
    takes the tiles nobody claimed. Whatever the split between the two, every output byte must be the byte of the plain
    route (frame pass alone) -- no tile lost, none written with stale planes -- and some tiles must really have been taken
    inside the kernel. (The emulation runs a launch's workgroups one after the other: the ordering rules that matter on the
    GPU -- planes visible before the count -- cannot fail here; the indexing, the claim arithmetic and the hand-over can.)"""
    rng = np.random.default_rng(fft + L + channels + blocks)
    h = (rng.standard_normal(taps) * 0.01).astype(np.float32)
    block = fft - (taps - 1)
    p = make_filter(h, fft, block, L)
    nin = block // L
    x = np.clip(rng.standard_normal((1, streams, blocks * nin, channels)) * 0.2, -1, 1).astype(np.float32)
    plain = run_emu(emu, tmp_path, p, x.tobytes(), streams, channels, blocks, 1, "fused", "f32", out_fmt)
    monkeypatch.setenv("EMU_COOP", "1")
    if cap is not None:
        monkeypatch.setenv("EMU_COOP_CAP", str(cap))
    (tmp_path / "in.bin").write_bytes(x.tobytes())
    r = subprocess.run([str(emu), str(p), "0", str(streams), str(channels), str(FMT["f32"]), str(FMT[out_fmt]), str(blocks), "1",
                        str(tmp_path / "in.bin"), str(tmp_path / "out2.bin"), "fused"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    coop = (tmp_path / "out2.bin").read_bytes()
    assert coop == plain
    took = [tuple(int(v) for v in line.split()[1:4:2]) for line in r.stderr.splitlines() if line.startswith("EMU_COOP:")]
    assert took and all(0 < a <= b for a, b in took), r.stderr[-500:]
    if cap == 4:
        assert all(a < b for a, b in took)       # the small cap leaves work for the frame pass
    y = np.frombuffer(coop, np.float32) if out_fmt == "f32" else O.pcm_to_float(np.frombuffer(coop, np.uint8), out_fmt)
    y = y.reshape(streams, blocks * block, channels)
    truth = O.truth_stream(x[0, streams - 1, :, channels - 1], h, L, blocks, block).reshape(-1)
    assert np.abs(y[streams - 1, :, channels - 1] - truth).max() <= 1e-5 * np.abs(truth).max() + 2.0**-31
