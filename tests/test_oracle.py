"""Pins the CPU oracle (oracle/) before anything is checked against it.

* oracle_upsampler.c (plain-C restatement) vs the committed golden vectors that
  tests/golden/make_golden.py produced from the compiled reference: BIT-EXACT.
* the same vs oracle/_ref (the reference compiled from /root/reference) live,
  whenever that library is present: BIT-EXACT.
* numpy restatements of the EQ maths and the APO parser vs golden G4 / _ref.
* numpy restatement of the PCM conversions vs the reference's own round-trip
  eps (tests/cpp/test_alsa_common.cpp:58-83,153-161) and hand-derived known
  answers (ALSA headers are absent, so that file cannot be compiled here).
"""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, real_input

KNOWN = np.array([1, 2, 3, 2, 1], dtype=np.float32)


def test_oracle_known_answer_bit_exact(O):
    g = np.load(GOLDEN / "g1_known_answer.npz")
    for L in (1, 2):
        o = O.OracleUpsampler(KNOWN, 5, 16, 12, L)
        np.testing.assert_array_equal(o.process_block(g[f"L{L}_impulse_in"]), g[f"L{L}_impulse_out"])
        o.reset()
        a, b = g[f"L{L}_iota_in"]
        np.testing.assert_array_equal(np.stack([o.process_block(a), o.process_block(b)]), g[f"L{L}_iota_out"])
        # and the reference test's own expectation (direct convolution, abs 1e-3)
        up = np.zeros(2 * 12)
        up[::L] = np.concatenate([a, b])
        np.testing.assert_allclose(np.concatenate(g[f"L{L}_iota_out"]), np.convolve(up, KNOWN)[:24], atol=1e-3)


@pytest.mark.parametrize("L", [1, 2, 4, 8, 16])
def test_oracle_mid_geometry_bit_exact(O, L):
    g = np.load(GOLDEN / "g2_mid.npz")
    o = O.OracleUpsampler(g["taps"], 257, 1024, 768, L)
    y = np.stack([o.process_block(x) for x in g[f"L{L}_in"]])
    np.testing.assert_array_equal(y, g[f"L{L}_ref"])
    # the reference's arithmetic is ~1e-4-accurate at this size; the truth is fp64
    assert np.abs(y - g[f"L{L}_truth"]).max() <= 2e-3 * np.abs(g[f"L{L}_truth"]).max()


G3 = json.loads((GOLDEN / "g3_real.json").read_text())


@pytest.mark.parametrize("name", sorted(G3))
def test_oracle_real_geometry_probes_bit_exact(O, name):
    meta = G3[name]
    g = np.load(GOLDEN / "g3_real.npz")
    h, taps, fft, block, L = O.read_filter(ROOT / meta["filter"])
    assert (taps, fft, block, L) == (meta["taps"], meta["fft"], meta["block"], meta["factor"])
    nin = block // L
    x = real_input(meta["seed"], meta["blocks"] * nin)
    np.testing.assert_array_equal(x[:8], g[f"{name}_x_head"])
    assert abs(float(x.astype(np.float64).sum()) - float(g[f"{name}_x_sum"])) < 1e-9
    o = O.OracleUpsampler(h, taps, fft, block, L)
    y = np.stack([o.process_block(x[b * nin:(b + 1) * nin]) for b in range(meta["blocks"])])
    np.testing.assert_array_equal(y[:, g[f"{name}_idx"]], g[f"{name}_ref"])
    np.testing.assert_allclose(y.astype(np.float64).sum(axis=1), g[f"{name}_ref_sum"], rtol=0, atol=1e-9)
    # fp64 truth recomputed here agrees with the stored probes
    truth = O.truth_stream(x, h, L, meta["blocks"], block)
    np.testing.assert_allclose(truth[:, g[f"{name}_idx"]], g[f"{name}_truth"], rtol=0, atol=1e-12)


def test_oracle_guards_return_empty(O):
    o = O.OracleUpsampler(KNOWN, 5, 16, 12, 2)
    assert o.process_block(np.zeros(0, np.float32)).size == 0
    assert o.process_block(np.zeros(5, np.float32)).size == 0
    assert o.process_block(np.zeros(12, np.float32)).size == 0
    assert o.process_block(np.zeros(6, np.float32)).size == 12


def test_oracle_matches_compiled_reference_live(O, make_filter):
    if not O.have_ref():
        pytest.skip("oracle/_ref not built here (no /root/reference): golden vectors above are the pin")
    rng = np.random.default_rng(11)
    for fft, T, L in [(16, 5, 1), (16, 5, 2), (16, 5, 3), (256, 65, 4), (2048, 513, 8), (8192, 2049, 16)]:
        block = fft - (T - 1)
        h = rng.standard_normal(T).astype(np.float32)
        p = make_filter(h, fft, block, L, name=f"r{fft}_{L}")
        r = O.RefUpsampler()
        ok, msg = r.load_filter(p)
        assert ok, msg
        o = O.OracleUpsampler(h, T, fft, block, L)
        for _ in range(3):
            x = rng.standard_normal(block // L).astype(np.float32)
            np.testing.assert_array_equal(r.process_block(x), o.process_block(x))
        # FFT itself, both directions
        v = (rng.standard_normal(fft) + 1j * rng.standard_normal(fft)).astype(np.complex64)
        np.testing.assert_array_equal(O.ref_fft(v), O.oracle_fft(v))
        np.testing.assert_array_equal(O.ref_fft(v, True), O.oracle_fft(v, True))


# ---- EQ ------------------------------------------------------------------------
PROFILES = json.loads((GOLDEN / "g4_eq_profiles.json").read_text())
GRIDS = {"768k": (65537, 131072, 768000.0), "705k": (65537, 131072, 705600.0), "small": (513, 1024, 44100.0 * 16),
         "lin160k": (131073, 262144, 768000.0)}  # the N = 262144 grid of BASELINE configs[4]


@pytest.mark.parametrize("name", sorted(PROFILES))
def test_oracle_eq_parse_and_response_vs_golden(O, name):
    g = np.load(GOLDEN / "g4_eq.npz")
    ok, pre, bands = O.eq_parse(PROFILES[name])
    assert ok
    assert pre == float(g[f"{name}_preamp"])
    want = g[f"{name}_bands"]
    assert len(bands) == len(want)
    for b, w in zip(bands, want):
        got = [float(b["enabled"]), float(b["type"]), b["frequency"], b["gain"], b["q"], float(b["has_bw_hz"]),
               b["bw_hz"], float(b["has_bw_oct"]), b["bw_oct"]]
        np.testing.assert_allclose(got, w, rtol=1e-15, atol=0)
    for tag, (bins, fft, fs) in GRIDS.items():
        idx = g[f"{name}_{tag}_idx"]
        resp = O.eq_response(PROFILES[name], bins, fft, fs)[idx]
        np.testing.assert_allclose(resp, g[f"{name}_{tag}_resp"], rtol=1e-9, atol=1e-12)
        mag = O.eq_magnitude(PROFILES[name], bins, fft, fs)[idx]
        np.testing.assert_allclose(mag, g[f"{name}_{tag}_mag"], rtol=1e-9, atol=1e-12)


def test_oracle_eq_reference_smoke_expectations(O):
    # tests/cpp/test_eq_parser_smoke.cpp:28-45
    ok, pre, bands = O.eq_parse(PROFILES["smoke"])
    assert ok and pre == -6.0 and len(bands) == 3
    assert bands[0]["enabled"] and not bands[1]["enabled"]
    assert bands[0]["frequency"] == 1000.0 and bands[0]["gain"] == -3.0 and bands[0]["q"] == 1.41
    assert bands[2]["has_bw_hz"] and abs(bands[2]["q"] - 5.0) < 1e-9
    # tests/cpp/test_eq_to_fir_smoke.cpp:17-78
    off = dict(enabled=False, type=0, frequency=1000.0, gain=6.0, q=1.0)
    np.testing.assert_allclose(O.eq_biquad(off, 44100.0), [1, 0, 0, 0, 0], atol=1e-9)
    pk = dict(enabled=True, type=0, frequency=1000.0, gain=6.0, q=1.41)
    c = O.eq_biquad(pk, 44100.0)
    z = np.exp(-2j * np.pi * 1000.0 / 44100.0)
    mag_db = 20 * np.log10(abs((c[0] + c[1] * z + c[2] * z * z) / (1 + c[3] * z + c[4] * z * z)))
    assert abs(mag_db - 6.0) <= 0.6
    np.testing.assert_allclose(O.eq_magnitude("", 513, 1024, 44100.0 * 16), 1.0, atol=1e-6)
    m = O.eq_magnitude("Filter 1: ON PK Fc 1000 Hz Gain 6 dB Q 1.0\n", 513, 1024, 44100.0 * 16)
    assert abs(m.max() - 1.0) <= 1e-6 and m.min() < 0.95


def test_oracle_eq_type_table(O):
    g = np.load(GOLDEN / "g4_eq.npz")
    assert list(g["type_names"]) == O.FILTER_TYPES
    for s, want in [("PK", 0), ("modal", 1), ("LPQ", 4), ("HS 12DB", 19), ("unknown", 0), ("Peaking", 0), ("ls6db", 16)]:
        assert O.eq_parse_filter_type(s) == want


def test_oracle_eq_live_reference(O):
    if not O.have_ref():
        pytest.skip("oracle/_ref not built here")
    for name, text in PROFILES.items():
        pre, bands = O.ref_eq_parse(text)
        ok, pre2, bands2 = O.eq_parse(text)
        assert ok and pre == pre2 and len(bands) == len(bands2)
        for tag, (bins, fft, fs) in GRIDS.items():
            np.testing.assert_allclose(O.eq_response(text, bins, fft, fs), O.ref_eq_response(text, bins, fft, fs),
                                       rtol=1e-9, atol=1e-12)
    for t in range(20):
        for en in (0, 1):
            for gain in (0.0, -4.5, 7.25):
                band = dict(enabled=bool(en), type=t, frequency=1234.5, gain=gain, q=0.9)
                np.testing.assert_allclose(O.eq_biquad(band, 705600.0),
                                           O.ref_eq_biquad(en, t, 1234.5, gain, 0.9, 705600.0), rtol=1e-14, atol=0)


# ---- PCM -------------------------------------------------------------------------
def test_oracle_pcm_round_trip_eps_of_reference_test(O):
    x = np.array([-0.9, -0.5, 0.0, 0.5, 0.9], dtype=np.float32)
    for fmt, eps in (("s16", 1e-3), ("s24", 2e-5), ("s32", 1e-7)):
        back = O.pcm_to_float(O.float_to_pcm(x, fmt), fmt)
        assert np.abs(back - x).max() <= eps


def test_oracle_pcm_known_answers(O):
    x = np.array([-1.0, -2.0, 1.0, 2.0, 0.5, -0.5, 0.25, 3.0517578125e-05, 0.99996948, np.nan], dtype=np.float32)
    s16 = O.float_to_pcm(x, "s16").view("<i2")
    # clamp [-1, 0.9999695], *32768, truncate toward zero; NaN -> upper clamp (std::min picks it)
    assert list(s16) == [-32768, -32768, 32767, 32767, 16384, -16384, 8192, 1, 32767, 32767]
    s32 = O.float_to_pcm(x[:7], "s32").view("<i4")
    # 0.9999999f * 2^31 in fp32 = 2147483392
    assert list(s32) == [-2147483648, -2147483648, 2147483392, 2147483392, 1073741824, -1073741824, 536870912]
    s24 = O.float_to_pcm(x[:7], "s24").reshape(-1, 3)
    v = s24[:, 0].astype(np.int32) | (s24[:, 1].astype(np.int32) << 8) | (s24[:, 2].astype(np.int32) << 16)
    v = np.where(v & 0x800000, v - (1 << 24), v)
    assert list(v) == [-8388608, -8388608, 8388607, 8388607, 4194304, -4194304, 2097152]
    np.testing.assert_array_equal(O.pcm_to_float(np.array([-32768, 32767, 1], "<i2"), "s16"),
                                  np.array([-1.0, 32767 / 32768, 1 / 32768], np.float32))
    np.testing.assert_array_equal(O.pcm_to_float(np.array([-2**31, 2**31 - 1, 1 << 30], "<i4"), "s32"),
                                  np.array([-1.0, 1.0, 0.5], np.float32))  # int->float rounds 2^31-1 up
    assert O.parse_format("S24_3LE") == "s24" and O.parse_format("s32") == "s32" and O.parse_format("u8") is None
