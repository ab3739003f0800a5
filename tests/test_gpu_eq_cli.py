"""GPU tests of the stages either side of the FFT path: the EQ cascade kernel and
its folding into the filter spectrum, and the alsa_streamer-compatible CLI in
file mode. Plus CPU-only checks of the CLI's argument handling."""
from __future__ import annotations

import json
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, real_input, rel_err

BIN = ROOT / "totton-rasp-gpu-dsp_amd" / "bin" / "alsa_streamer"
PROFILES = json.loads((GOLDEN / "g4_eq_profiles.json").read_text())
GRIDS = {"768k": (65537, 131072, 768000.0), "705k": (65537, 131072, 705600.0), "small": (513, 1024, 44100.0 * 16),
         "lin160k": (131073, 262144, 768000.0)}  # the N = 262144 grid of BASELINE configs[4]


# ------------------------------------------------------------------ EQ (GPU) --
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PROFILES))
def test_eq_cascade_kernel_vs_reference_golden(ups, gpu, name):
    """fp64 per-bin cascade on the device vs computeEqResponseForFft of the
    compiled reference (tests/golden/g4_eq.npz). Tolerance 1e-9 relative: both
    are fp64; they differ in sincos and complex-division rounding only."""
    g = np.load(GOLDEN / "g4_eq.npz")
    for tag, (bins, fft, fs) in GRIDS.items():
        dev = ups.eq_response_device(PROFILES[name], bins, fft, fs, device=gpu)
        idx = g[f"{name}_{tag}_idx"]
        np.testing.assert_allclose(dev[idx], g[f"{name}_{tag}_resp"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(dev, ups.eq_response_host(PROFILES[name], bins, fft, fs), rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_empty_eq_profile_is_unity(ups, gpu):
    r = ups.eq_response_device("", 513, 1024, 705600.0, device=gpu)
    np.testing.assert_allclose(r, 1.0 + 0j, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("fname,fs", [("filter_48k_16x_80000_min_phase.json", 768000.0),
                                      ("filter_44k_4x_80000_min_phase.json", 705600.0),
                                      ("filter_44k_2x_80000_min_phase.json", 705600.0)])  # K = 32768: split kernel tables
def test_eq_folded_into_filter_spectrum(ups, O, gpu, fname, fs):
    """The EQ is folded into the FIR (cascade recursion over the taps, cut to `taps` samples: this repo's definition of
    the fusion; the reference has no call site, so only the biquads and the per-bin EQ values are pinned to it). The
    stream is then ONE linear convolution: checked against the fp64 true streaming convolution with the folded FIR,
    1e-5*max|y| -- a truth with no block structure in it, so a wrap inside a block would show."""
    path = ROOT / "data" / "coefficients" / fname
    h, taps, fft, block, L = O.read_filter(path)
    text = PROFILES["opra10"]
    nin, nb = block // L, 3
    x = real_input(21, nb * nin)
    want = O.truth_stream(x, O.eq_fold_fir(h, text, fs), L, nb, block)
    u = ups.StreamingUpsampler(gpu)
    assert u.load_filter(path)[0]
    plain = np.stack([u.process_block(x[b * nin:(b + 1) * nin]) for b in range(nb)])
    u.reset()
    u.set_eq(text, fs)
    y = np.stack([u.process_block(x[b * nin:(b + 1) * nin]) for b in range(nb)])
    assert rel_err(y, want) <= 1e-5
    assert rel_err(y, plain) > 1e-2       # the EQ really changed the signal
    v = u.clone()                         # a clone keeps the EQ ...
    u.set_eq("", fs)                      # ... and removing it on the original does not touch the clone
    u.reset()
    v.reset()
    np.testing.assert_array_equal(np.stack([u.process_block(x[b * nin:(b + 1) * nin]) for b in range(nb)]), plain)
    assert rel_err(np.stack([v.process_block(x[b * nin:(b + 1) * nin]) for b in range(nb)]), want) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("fname,fs,channels,blocks", [
    ("filter_48k_16x_80000_min_phase.json", 768000.0, 8, 4),       # BASELINE configs[2] as benched: 8 ch + EQ
    ("filter_48k_8x_160000_linear_phase.json", 768000.0, 32, 3),   # BASELINE configs[4] as benched: 32 ch + EQ, N = 262144
])
def test_bench_configs_with_eq_as_benched(ups, O, gpu, fname, fs, channels, blocks):
    """bench.py --config 3 / --config 5 exactly as timed: mi_filter_set_eq(opra10) on the shared filter, then the
    batched engine on interleaved s32 frames of `channels` channels (planarize -> fused -> interleave kernels).
    Checked per channel against the fp64 true streaming convolution with the EQ-folded FIR: 1 LSB + 1e-5 * max|y|."""
    path = ROOT / "data" / "coefficients" / fname
    h, taps, fft, block, L = O.read_filter(path)
    text = PROFILES["opra10"]
    filt = ups.Filter(path, device=gpu)
    filt.set_eq(text, fs)
    eng = ups.Engine(filt, 1, channels, ups.PCM_S32, ups.PCM_S32)
    assert eng.path == "fused"
    nin = eng.in_frames
    xf = np.clip(np.random.default_rng(31).standard_normal((blocks * nin, channels)) * 0.02, -1, 1).astype(np.float32)
    raw = O.float_to_pcm(xf.reshape(-1), "s32")
    xin = O.pcm_to_float(raw, "s32").reshape(blocks * nin, channels)
    y = O.pcm_to_float(eng.process_host(raw, blocks), "s32").reshape(blocks * block, channels)
    fir = O.eq_fold_fir(h, text, fs)
    for c in sorted({0, 1, channels // 2, channels - 1}):
        want = O.truth_stream(xin[:, c], fir, L, blocks, block).reshape(-1)
        assert np.abs(want).max() < 0.9          # nothing clamps: the comparison is of the filter, not of the limiter
        assert np.abs(y[:, c] - want).max() <= 2.0**-31 + 1e-5 * np.abs(want).max()


@pytest.mark.gpu
def test_staged_and_fused_paths_agree_at_full_size(ups, O, gpu, make_filter):
    """2x filter -> K = 32768: the split fused kernel (two 16384-point halves); 4x -> fused;
    the same 2x geometry with a history length that is not a multiple of 4 -> the staged engine, whose K = 2^15
    transforms take the two-level path (rows of 2048 points in LDS: the smallest size that path covers, on the GPU
    here; the other sizes in test_640k_tap_filters_take_the_two_level_path). Same truth bar for all."""
    h2 = np.fromfile(ROOT / "data" / "coefficients" / "filter_44k_2x_80000_min_phase.bin", "<f4")
    odd = make_filter(np.concatenate([h2, np.zeros(2, np.float32)]), 131072, 131072 - 80002, 2, name="odd2x")
    for path, path_name in [(ROOT / "data" / "coefficients" / "filter_44k_2x_80000_min_phase.json", "fused"),
                            (ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json", "fused"),
                            (odd, "staged")]:
        h, taps, fft, block, L = O.read_filter(path)
        filt = ups.Filter(path, device=gpu)
        eng = ups.Engine(filt, 2, 2, ups.PCM_F32, ups.PCM_F32)
        assert eng.path == path_name
        nin, blocks = eng.in_frames, 2
        x = real_input(5, 2 * blocks * nin * 2).reshape(2, blocks * nin, 2)
        y = eng.process_host(x, blocks).view(np.float32).reshape(2, blocks * block, 2)
        assert eng.last_two_level == (path_name == "staged")
        for s in range(2):
            for c in range(2):
                truth = O.truth_stream(x[s, :, c], h, L, blocks, block).reshape(-1)
                assert rel_err(y[s, :, c], truth) <= 1e-5


# ------------------------------------------------------------------ CLI (GPU) --
@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["s32", "s16", "s24"])
def test_cli_file_mode_matches_engine_and_truth(ups, O, gpu, tmp_path, fmt):
    assert BIN.exists(), "build() must produce bin/alsa_streamer"
    path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
    h, taps, fft, block, L = O.read_filter(path)
    nin = block // L
    frames = 2 * nin + 1000                    # a short tail block
    xf = np.clip(np.random.default_rng(2).standard_normal((frames, 2)) * 0.05, -1, 1).astype(np.float32)
    raw = O.float_to_pcm(xf.reshape(-1), fmt)
    (tmp_path / "in.raw").write_bytes(raw.tobytes())
    r = subprocess.run([str(BIN), "--in-file", str(tmp_path / "in.raw"), "--out-file", str(tmp_path / "out.raw"),
                        "--rate", "44100", "--filter-dir", str(path.parent), "--ratio", "4", "--phase", "min",
                        "--channels", "2", "--format", fmt, "--blocks-per-call", "2"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    assert "File processing started: input 44100 Hz, period %d frames" % nin in r.stderr
    assert "File processing stopped" in r.stderr
    out = np.frombuffer((tmp_path / "out.raw").read_bytes(), np.uint8)
    bps = {"s16": 2, "s24": 3, "s32": 4}[fmt]
    assert out.size == frames * L * 2 * bps    # framesRead * ratio frames (documented deviation for L > 1)
    got = O.pcm_to_float(out, fmt).reshape(-1, 2)
    xin = O.pcm_to_float(raw, fmt).reshape(-1, 2)
    xin = np.concatenate([xin, np.zeros((3 * nin - frames, 2), np.float32)])
    lsb = 2.0 ** -(8 * bps - 1)
    for c in range(2):
        truth = O.truth_stream(xin[:, c], h, L, 3, block).reshape(-1)[:frames * L]
        want = O.pcm_to_float(O.float_to_pcm(truth.astype(np.float32), fmt), fmt)
        assert np.abs(got[:, c] - want).max() <= lsb + 1e-5 * np.abs(truth).max()


@pytest.mark.gpu
def test_cli_opra_record_equals_the_converted_apo_profile(ups, tmp_path):
    """--opra record.json [--modern-target] = the OPRA step before the EQ path (reference: web/routers/opra.py:140-160,
    record -> APO text -> profile file): the output must be byte-identical to --eq with the converted text."""
    g8 = json.loads((ROOT / "tests" / "golden" / "g8_opra.json").read_text())
    case = next(c for c in g8["cases"] if c["record"]["name"] == "ten bands")
    (tmp_path / "record.json").write_text(json.dumps(case["record"]))
    (tmp_path / "profile.txt").write_text(case["apo_modern_target"])
    path = ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json"
    x = (np.clip(np.random.default_rng(9).standard_normal((30000, 2)) * 0.1, -1, 1) * 2147483647).astype("<i4")
    (tmp_path / "in.raw").write_bytes(x.tobytes())
    common = ["--in-file", str(tmp_path / "in.raw"), "--rate", "44100", "--filter", str(path), "--channels", "2", "--format", "s32"]
    a = subprocess.run([str(BIN), *common, "--out-file", str(tmp_path / "a.raw"), "--opra", str(tmp_path / "record.json"),
                        "--modern-target"], capture_output=True, text=True, timeout=300)
    b = subprocess.run([str(BIN), *common, "--out-file", str(tmp_path / "b.raw"), "--eq", str(tmp_path / "profile.txt")],
                       capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr
    assert b.returncode == 0, b.stderr
    out_a, out_b = (tmp_path / "a.raw").read_bytes(), (tmp_path / "b.raw").read_bytes()
    assert len(out_a) == 30000 * 4 * 2 * 4 and out_a == out_b
    # and the EQ is really in the path: without it the output differs
    c = subprocess.run([str(BIN), *common, "--out-file", str(tmp_path / "c.raw")], capture_output=True, text=True, timeout=300)
    assert c.returncode == 0 and (tmp_path / "c.raw").read_bytes() != out_a
    bad = subprocess.run([str(BIN), *common, "--out-file", str(tmp_path / "d.raw"), "--opra", str(tmp_path / "profile.txt")],
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "OPRA: OPRA record:" in bad.stderr


@pytest.mark.gpu
def test_cli_without_filter_is_pcm_passthrough(O, tmp_path):
    x = (np.random.default_rng(1).integers(-2**31, 2**31 - 1, size=4096 * 2, dtype=np.int64)).astype("<i4")
    (tmp_path / "in.raw").write_bytes(x.tobytes())
    r = subprocess.run([str(BIN), "--in-file", str(tmp_path / "in.raw"), "--out-file", str(tmp_path / "out.raw"),
                        "--rate", "48000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = np.frombuffer((tmp_path / "out.raw").read_bytes(), "<i4")
    # the reference converts PCM -> float -> PCM even without a filter
    want = O.float_to_pcm(O.pcm_to_float(x.view(np.uint8), "s32"), "s32").view("<i4")
    np.testing.assert_array_equal(out, want)


# ------------------------------------------------------------------ CLI (CPU) --
def run_cli(*args):
    return subprocess.run([str(BIN), *args], capture_output=True, text=True, timeout=60)


def test_cli_usage_and_argument_errors():
    assert BIN.exists(), "build() must produce bin/alsa_streamer"
    r = run_cli("--help")
    assert r.returncode == 0
    for flag in ["--in-file", "--out-file", "--filter ", "--filter-dir", "--phase", "--ratio", "--rate", "--channels",
                 "--format", "--period", "--buffer", "--help"]:
        assert flag in r.stdout, flag
    r = run_cli()
    assert r.returncode == 1 and "--in and --out are required" in r.stderr
    r = run_cli("--in-file", "x.raw")
    assert r.returncode == 1 and "--in-file and --out-file must be specified together" in r.stderr
    r = run_cli("--bogus")
    assert r.returncode == 1 and "Unknown argument: --bogus" in r.stderr
    r = run_cli("--rate")
    assert r.returncode == 1 and "Missing value for --rate" in r.stderr
    r = run_cli("--in", "hw:0", "--out", "hw:1", "--format", "u8")
    assert r.returncode == 1 and "Unsupported format: u8" in r.stderr
    r = run_cli("--in", "hw:0", "--out", "hw:1")
    assert r.returncode == 1 and "ALSA support is not compiled" in r.stderr


def test_cli_filter_selection_errors(tmp_path):
    (tmp_path / "in.raw").write_bytes(b"\0" * 64)
    common = ["--in-file", str(tmp_path / "in.raw"), "--out-file", str(tmp_path / "out.raw")]
    r = run_cli(*common)
    assert r.returncode == 1 and "--rate is required for file processing" in r.stderr
    r = run_cli(*common, "--rate", "44100", "--filter", str(tmp_path / "missing.json"))
    assert r.returncode == 1 and f"Filter load failed: Filter file not found: {tmp_path / 'missing.json'}" in r.stderr
    # an explicit --filter-dir with the default ratio 1 finds nothing and is NOT fatal (reference quirk)
    r = run_cli(*common, "--rate", "44100", "--filter-dir", str(tmp_path))
    assert r.returncode == 0 and "Filter not available, continuing without filter" in r.stderr
    assert (tmp_path / "out.raw").read_bytes() == b"\0" * 64
    # a bad sidecar is fatal and reports the reference's message and the path
    bad = tmp_path / "bad.json"
    bad.write_text('{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 18, "block_size": 14}')
    r = run_cli(*common, "--rate", "44100", "--filter", str(bad))
    assert r.returncode == 1 and "Filter load failed: fft_size must be power of two" in r.stderr
    assert f"Filter path: {bad}" in r.stderr
