"""The EQ folded into the FIR (SURVEY 7 hard part D): host fold vs the oracle's scipy restatement, the committed golden
of the TRUE streaming result (tests/golden/g9_eq_stream.*: upsample, then run the real recursive cascade, fp64), the
stated residual as a real bound, the limit / warning / strict behaviour -- and on the GPU the stream itself.

Parity status of the fusion: UNPINNED (the reference never calls its EQ code from the data plane). Pinned pieces: the
biquad coefficients and the parser (compiled reference, g4), and here the golden's cascade, which was run with the
compiled reference's own coefficients."""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN, real_input

META = json.loads((GOLDEN / "g9_eq_stream.json").read_text())
CASES = sorted(META)
# cases whose free decay the golden followed to the end (64 N samples past the taps): the two Fc = 10 Hz corners ring on
COMPLETE = [k for k in CASES if META[k]["tail_followed_to"] < 1e-12]


def case_inputs(O, key):
    m = META[key]
    h, taps, fft, block, L = O.read_filter(GOLDEN / "filters" / m["filter"])
    x = real_input(m["seed"], m["blocks"] * (block // L))
    return m, h, taps, fft, block, L, x


@pytest.mark.parametrize("key", CASES)
def test_host_fold_matches_the_scipy_restatement(ups, O, key):
    """csrc/host/eq.cpp FoldCascadeIntoTaps == oracle.eq_fold_fir (scipy.signal.lfilter per section + taper), 1e-10 of
    the peak (both fp64 transposed direct form II; they differ in operation order only)."""
    m, h, taps, fft, block, L, _ = case_inputs(O, key)
    fir, r = ups.eq_fold_host(h, fft, m["profile"], m["fs_out"])
    want = O.eq_fold_fir(h, m["profile"], m["fs_out"])
    assert fir.size == taps == want.size and r["fir_taps"] == taps and r["taper"] == (taps - 1) // 64
    assert np.abs(fir - want).max() <= 1e-10 * np.abs(want).max()
    assert fir[-1] == pytest.approx(0.0, abs=1e-6 * np.abs(want).max())  # the roll-off closes the FIR


@pytest.mark.parametrize("key", CASES)
def test_residual_figures_match_the_golden(ups, O, key):
    """tail_l1 / tail_l2 of the library against the golden's (h_ideal followed 64 N samples past the taps). Where the
    golden reached the end of the decay the figures agree to 1e-6 relative (or are both below 1e-12); where it did not
    (Fc 10 Hz corners) the library, which follows further, may only report MORE."""
    m, h, taps, fft, block, L, _ = case_inputs(O, key)
    _, r = ups.eq_fold_host(h, fft, m["profile"], m["fs_out"])
    for name in ("tail_l1", "tail_l2"):
        if key in COMPLETE:
            assert r[name] == pytest.approx(m[name], rel=1e-6, abs=1e-12), name
        else:
            assert r[name] >= m[name] * (1 - 1e-3), name
    assert r["over_limit"] == int(r["tail_l1"] > 1e-3) and r["limit"] == 1e-3
    assert r["tail_l1_db"] == pytest.approx(20 * np.log10(max(r["tail_l1"], 1e-20)), abs=1e-6) or r["tail_l1"] < 1e-20
    # the response deviation is small exactly when the tail is
    assert (r["response_dev"] > 1e-4) == (r["tail_l2"] > 1e-4)


def test_which_profiles_are_over_the_limit(ups, O):
    """The table DESIGN 8 carries: the benched profile and the high corners fit an 80k-tap filter, everything whose
    ringing Q / (pi Fc) is comparable to the filter's 0.1 s does not -- and says so."""
    over = {}
    for key in CASES:
        m, h, taps, fft, block, L, _ = case_inputs(O, key)
        over[key] = ups.eq_fold_host(h, fft, m["profile"], m["fs_out"])[1]["over_limit"]
    for f in ("48k_16x", "44k_2x"):
        assert [over[f"{f}_{p}"] for p in ("opra10", "corner_hi_narrow", "corner_hi_wide")] == [0, 0, 0]
        assert all(over[f"{f}_{p}"] for p in ("pk20q8", "pk60q10", "pk30q4", "corner_lo_narrow", "corner_lo_wide", "shelf_lo"))


@pytest.mark.parametrize("key", CASES)
def test_oracle_fold_stream_matches_the_golden(O, key):
    """oracle.eq_fold_fir + truth_stream reproduce the golden's `folded` stream (1e-12 of the peak: same recipe, the
    golden's coefficients came from the compiled reference, the oracle's from its numpy restatement), and the golden's
    `ideal` stream (real recursive cascade on the upsampled signal) lies within the STATED bound
    tail_l1 * ||h_ideal||_1 * max|x| of it."""
    m, h, taps, fft, block, L, x = case_inputs(O, key)
    g = np.load(GOLDEN / "g9_eq_stream.npz")
    idx = g[m["filter"].replace("filter_", "").replace("_80000_min_phase.json", "") + "_idx"]
    y = O.truth_stream(x, O.eq_fold_fir(h, m["profile"], m["fs_out"]), L, m["blocks"], block)[:, idx]
    assert np.abs(y - g[f"{key}_folded"]).max() <= 1e-12 * max(m["max_ideal"], 1e-3) + 1e-15
    bound = m["tail_l1"] * m["l1_ideal"] * m["max_x"]
    assert np.abs(g[f"{key}_ideal"] - g[f"{key}_folded"]).max() <= bound + 1e-10 * m["max_ideal"]  # fp64 roundoff of a 200k-sample recursion
    assert m["max_ideal_minus_folded"] <= bound + 1e-10 * m["max_ideal"]  # fp64 roundoff of a 200k-sample recursion


def test_fold_without_bands_is_the_taps_times_preamp(ups, O):
    h, taps, fft, block, L = O.read_filter(GOLDEN / "filters" / "filter_44k_4x_80000_min_phase.json")
    fir, r = ups.eq_fold_host(h, fft, "Preamp: -6 dB\nFilter 1: ON PK Fc 1000 Hz Gain 0 dB Q 1\n", 705600.0)
    w = O.eq_fold_taper(taps)
    np.testing.assert_allclose(fir, h.astype(np.float64) * 10 ** (-6 / 20) * w, rtol=1e-15, atol=0)
    assert r["over_limit"] == 0 and r["tail_complete"] == 1


def test_fold_of_tiny_filters(ups, O):
    """taps < 65: no roll-off at all (W = 0); one tap: the cascade's first sample."""
    fir, r = ups.eq_fold_host(np.array([1, 2, 3, 2, 1], np.float32), 16, "Filter 1: ON PK Fc 1000 Hz Gain 6 dB Q 1\n", 48000.0)
    np.testing.assert_allclose(fir, O.eq_fold_fir(np.array([1, 2, 3, 2, 1.0]), "Filter 1: ON PK Fc 1000 Hz Gain 6 dB Q 1\n", 48000.0), rtol=1e-13)
    assert r["taper"] == 0 and r["over_limit"] == 1  # five taps cannot hold a biquad's ringing, and the report says so
    fir1, _ = ups.eq_fold_host(np.array([2.0], np.float32), 2, "Filter 1: ON PK Fc 1000 Hz Gain 6 dB Q 1\n", 48000.0)
    assert fir1.shape == (1,) and fir1[0] == pytest.approx(2.0 * O.eq_sections("Filter 1: ON PK Fc 1000 Hz Gain 6 dB Q 1\n", 48000.0)[1][0][0])


# ---------------------------------------------------------------------------------------------------------- GPU --
GPU_CASES = [k for k in CASES if k.split("_", 2)[2] in ("opra10", "pk20q8", "pk60q10", "corner_lo_narrow", "corner_hi_narrow", "shelf_lo")]


@pytest.mark.gpu
@pytest.mark.parametrize("key", GPU_CASES)
def test_gpu_stream_with_eq_is_the_true_streaming_convolution(ups, O, gpu, key):
    """mi_filter_set_eq + four blocks through the engine, against the committed golden of the TRUE streaming result:
    within 1e-5 * max|y| of the folded-FIR stream (nothing wraps inside a block: round 3's H*EQ-on-the-grid product was
    1.9e-3 off here for PK 20 Hz Q 8), and within that plus the STATED truncation bound of the real recursive cascade."""
    m, h, taps, fft, block, L, x = case_inputs(O, key)
    g = np.load(GOLDEN / "g9_eq_stream.npz")
    idx = g[m["filter"].replace("filter_", "").replace("_80000_min_phase.json", "") + "_idx"]
    filt = ups.Filter(GOLDEN / "filters" / m["filter"], device=gpu)
    warning = filt.set_eq(m["profile"], m["fs_out"])
    r = filt.eq_residual()
    assert r["active"] == 1 and r["fir_taps"] == taps
    assert bool(warning) == bool(r["over_limit"])
    if r["over_limit"]:
        assert warning.startswith(f"EQ cut to {taps} taps drops ") and "limit -60.0 dB" in warning
    eng = ups.Engine(filt, 1, 1, ups.PCM_F32, ups.PCM_F32)
    y = eng.process_host(x, m["blocks"]).view(np.float32).reshape(m["blocks"], block).astype(np.float64)[:, idx]
    scale = max(m["max_ideal"], np.abs(g[f"{key}_folded"]).max())
    assert np.abs(y - g[f"{key}_folded"]).max() <= 1e-5 * scale
    assert np.abs(y - g[f"{key}_ideal"]).max() <= 1e-5 * scale + r["tail_l1"] * m["l1_ideal"] * m["max_x"]
    # the device-evaluated cascade is the yardstick of response_dev: the host figure is the same number
    _, rh = ups.eq_fold_host(h, fft, m["profile"], m["fs_out"])
    assert r["response_dev"] == pytest.approx(rh["response_dev"], rel=1e-6, abs=1e-8)  # below that: fp64 noise of two FFT-size sums
    assert r["tail_l1"] == rh["tail_l1"]


@pytest.mark.gpu
def test_gpu_eq_limit_strict_refuses_and_keeps_the_old_spectrum(ups, O, gpu):
    path = GOLDEN / "filters" / "filter_48k_16x_80000_min_phase.json"
    h, taps, fft, block, L = O.read_filter(path)
    filt = ups.Filter(path, device=gpu)
    eng = ups.Engine(filt, 1, 1, ups.PCM_F32, ups.PCM_F32)
    x = real_input(5, 2 * (block // L))
    plain = eng.process_host(x, 2).view(np.float32).copy()
    filt.set_eq_limit(-1.0, strict=True)
    gen = filt.generation
    with pytest.raises(ups.UpsamplerError, match="EQ cut to 80001 taps drops -22.. dB"):
        filt.set_eq("Filter 1: ON PK Fc 20 Hz Gain 6 dB Q 8\n", 768000.0)
    assert filt.generation == gen and filt.eq_residual()["active"] == 0
    eng.reset()
    np.testing.assert_array_equal(eng.process_host(x, 2).view(np.float32), plain)
    assert filt.set_eq(json.loads((GOLDEN / "g4_eq_profiles.json").read_text())["opra10"], 768000.0) == ""  # fits: accepted
    filt.set_eq_limit(0.5, strict=True)   # a caller may widen the limit ...
    assert filt.set_eq("Filter 1: ON PK Fc 20 Hz Gain 6 dB Q 8\n", 768000.0) == ""
    assert filt.eq_residual()["over_limit"] == 0 and filt.eq_residual()["limit"] == 0.5
    filt.set_eq("", 768000.0)
    assert filt.eq_residual()["active"] == 0


@pytest.mark.gpu
def test_gpu_single_channel_handle_reports_the_residual(ups, O, gpu):
    u = ups.StreamingUpsampler(gpu)
    assert u.load_filter(GOLDEN / "filters" / "filter_44k_2x_80000_min_phase.json")[0]
    assert u.eq_residual()["active"] == 0
    w = u.set_eq("Filter 1: ON PK Fc 60 Hz Gain 6 dB Q 10\n", 705600.0)
    assert "EQ cut to 80001 taps" in w and u.eq_residual()["over_limit"] == 1
    v = u.clone()
    assert v.set_eq("Filter 1: ON PK Fc 6000 Hz Gain 6 dB Q 10\n", 705600.0) == ""   # forks a private filter
    assert v.eq_residual()["over_limit"] == 0 and u.eq_residual()["over_limit"] == 1
