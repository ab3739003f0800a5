"""CPU-only tests of the product's host side (no GPU, no compute calls):
the C-ABI library loads and exports every symbol include/mi_upsampler.h
declares; sidecar loading, filter selection, PCM conversion, EQ parsing/maths
and the load-time table construction agree with the oracle / the reference's
golden behaviour; and GPU entry points fail loudly without a device."""
from __future__ import annotations

import json
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

KNOWN = np.array([1, 2, 3, 2, 1], dtype=np.float32)


# ---- C ABI surface ------------------------------------------------------------
def header_symbols():
    text = (ROOT / "include" / "mi_upsampler.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ups):
    syms = header_symbols()
    assert len(syms) >= 40
    out = subprocess.run(["nm", "-D", "--defined-only", str(ups.LIB_PATH)], capture_output=True, text=True, check=True)
    exported = {line.split()[-1] for line in out.stdout.splitlines() if " T " in line}
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared in include/mi_upsampler.h but not exported: {missing}"
    # and the Python binding table names exactly the same set
    assert sorted(ups.EXPORTED_SYMBOLS) == syms
    assert ups.lib.mi_ups_abi_version() == 1


def test_gpu_entry_points_fail_loudly_without_a_device(ups):
    if ups.device_count() > 0:
        pytest.skip("a GPU is visible here")
    u = ups.StreamingUpsampler(0)
    ok, msg = u.load_filter(ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json")
    assert not ok and "no HIP device" in msg
    assert u.process_block(np.zeros(12768, np.float32)).size == 0
    with pytest.raises(ups.UpsamplerError):
        ups.Filter(ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json")
    with pytest.raises(ups.UpsamplerError):
        ups.eq_response_device("Preamp: -3 dB\n", 16, 32, 48000.0)


def test_product_does_not_reference_the_oracle():
    """The product tree must not import, link or call anything under oracle/."""
    bad = []
    for p in (ROOT / "totton-rasp-gpu-dsp_amd").rglob("*"):
        if p.is_file() and p.suffix in {".py", ".cpp", ".h", ".hip", ""} and "build" not in p.parts and p.name == p.name:
            if p.suffix == "" and p.name != "Makefile":
                continue
            t = p.read_text(errors="ignore")
            if re.search(r"liboracle|libref_oracle|orc_ups|import oracle|oracle_upsampler", t):
                bad.append(str(p))
    assert not bad, bad
    out = subprocess.run(["ldd", str(ROOT / "totton-rasp-gpu-dsp_amd" / "lib" / "libmi_upsampler.so")],
                         capture_output=True, text=True)
    assert "oracle" not in out.stdout


# ---- sidecar / coefficient loading --------------------------------------------
def test_read_filter_messages_match_reference(ups, tmp_path):
    cases = json.loads((GOLDEN / "g6_load_errors.json").read_text())
    KNOWN.astype("<f4").tofile(tmp_path / "c.bin")
    for name, case in cases.items():
        p = tmp_path / f"{name}.json"
        if case["body"] is not None:
            p.write_text(case["body"])
        ok, msg, cfg = ups.read_filter(p)
        assert ok == case["ok"], name
        assert msg == case["message"].replace("<DIR>", str(tmp_path)), name
        if ok:
            assert [cfg["taps"], cfg["fft_size"], cfg["block_size"], cfg["upsample_factor"]] == case["config"], name
            assert cfg["coefficients_path"] == str(tmp_path / "c.bin")


def test_read_filter_shipped_sidecars(ups):
    for p in sorted((GOLDEN / "filters").glob("*.json")) + sorted((ROOT / "data" / "coefficients").glob("*.json")):
        ok, msg, cfg = ups.read_filter(p)
        meta = json.loads(p.read_text())
        assert ok, (p, msg)
        assert cfg["taps"] == meta["taps"] and cfg["fft_size"] == meta["fft_size"]
        assert cfg["block_size"] == meta["block_size"] and cfg["upsample_factor"] == meta["upsample_factor"]
        assert cfg["fft_size"] - cfg["block_size"] == cfg["taps"] - 1


def test_absolute_coefficient_path_is_kept(ups, tmp_path):
    KNOWN.astype("<f4").tofile(tmp_path / "abs.bin")
    p = tmp_path / "a.json"
    p.write_text(json.dumps(dict(coefficients_bin=str(tmp_path / "abs.bin"), taps=5, fft_size=16, block_size=12)))
    ok, msg, cfg = ups.read_filter(p)
    assert ok and cfg["coefficients_path"] == str(tmp_path / "abs.bin")


# ---- filter selection (tests/cpp/test_alsa_filter_selector.cpp:33-77) ----------
def test_resolve_filter_path_reference_cases(ups, tmp_path):
    d = tmp_path / "filters"
    d.mkdir()
    direct = d / "direct.json"
    direct.write_text("{}\n")
    assert ups.resolve_filter_path(str(direct), "", "min", 1, 44100) == (str(direct), "")
    (d / "filter_44k_2x_80000_min_phase.json").write_text("{}\n")
    legacy = d / "filter_44k_2x_2m_min_phase.json"
    legacy.write_text("{}\n")
    path, err = ups.resolve_filter_path("", str(d), "min", 2, 44100)
    assert path == str(legacy)  # "2m" == 640000 taps beats 80000
    path, err = ups.resolve_filter_path("", str(d), "min", 2, 32000)
    assert path is None and err == "Unsupported input rate family: 32000"
    path, err = ups.resolve_filter_path("", str(d / "missing"), "min", 2, 44100)
    assert path is None and err == f"Filter directory not found: {d / 'missing'}"
    path, err = ups.resolve_filter_path(str(d / "nope.json"), str(d), "min", 2, 44100)
    assert path is None and err == f"Filter file not found: {d / 'nope.json'}"
    path, err = ups.resolve_filter_path("", str(d), "linear", 2, 88200)
    assert path is None and err == f"Filter file not found: {d}/filter_44k_2x_*_linear_phase.json"
    path, err = ups.resolve_filter_path("", "", "min", 2, 44100)
    assert path is None and err == ""


def test_resolve_filter_path_details(ups, tmp_path):
    d = tmp_path
    for n in ["filter_48k_8x_160000_linear_phase.json", "filter_48k_8x_80000_linear_phase.json",
              "filter_48k_8x_12x_linear_phase.json", "filter_48k_8x__linear_phase.json",
              "filter_48k_8x_999999_min_phase.json", "filter_48k_16x_80000_linear_phase.json"]:
        (d / n).write_text("{}")
    (d / "filter_48k_8x_500000_linear_phase.json").mkdir()  # directories are ignored
    path, _ = ups.resolve_filter_path("", str(d), "linear", 8, 96000)
    assert path == str(d / "filter_48k_8x_160000_linear_phase.json")
    # a custom phase string is used verbatim as the suffix
    (d / "filter_44k_4x_100_custom.json").write_text("{}")
    path, _ = ups.resolve_filter_path("", str(d), "custom", 4, 176400)
    assert path == str(d / "filter_44k_4x_100_custom.json")
    # this repo's data directory resolves the bench configs
    data = ROOT / "data" / "coefficients"
    assert ups.resolve_filter_path("", str(data), "min", 4, 44100)[0] == str(data / "filter_44k_4x_80000_min_phase.json")
    assert ups.resolve_filter_path("", str(data), "linear", 8, 48000)[0] == str(data / "filter_48k_8x_160000_linear_phase.json")


# ---- PCM ---------------------------------------------------------------------
def test_parse_format(ups, O):
    for name in ["s16", "S16_LE", "s24", "s24_3le", "S32", "s32_le", "u8", "", "f32"]:
        want = O.parse_format(name)
        got = ups.parse_format(name)
        assert got == (-1 if want is None else ups.PCM_NAMES[want]), name
    assert [ups.bytes_per_sample(f) for f in (ups.PCM_S16, ups.PCM_S24_3LE, ups.PCM_S32, ups.PCM_F32)] == [2, 3, 4, 4]


@pytest.mark.parametrize("fmt", ["s16", "s24", "s32"])
def test_pcm_conversion_bit_exact_vs_oracle(ups, O, fmt):
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 0.6,
                        np.array([-1.0, 1.0, -2.0, 2.0, 0.0, -0.0, 0.9999695, 0.9999999, 1e-9, -1e-9, np.nan, np.inf,
                                  -np.inf], np.float32)])
    pcm = ups.float_to_pcm(x, ups.PCM_NAMES[fmt])
    np.testing.assert_array_equal(pcm, O.float_to_pcm(x, fmt))
    np.testing.assert_array_equal(ups.pcm_to_float(pcm, ups.PCM_NAMES[fmt]), O.pcm_to_float(pcm, fmt))
    # every code of the format's range (s16) / random codes (wider formats)
    raw = (np.arange(-32768, 32768).astype("<i2").view(np.uint8) if fmt == "s16"
           else rng.integers(0, 256, size=3 * 4 * 5000, dtype=np.uint8))
    np.testing.assert_array_equal(ups.pcm_to_float(raw, ups.PCM_NAMES[fmt]), O.pcm_to_float(raw, fmt))
    # reference round-trip eps (tests/cpp/test_alsa_common.cpp:153-161)
    t = np.array([-0.9, -0.5, 0.0, 0.5, 0.9], np.float32)
    eps = {"s16": 1e-3, "s24": 2e-5, "s32": 1e-7}[fmt]
    assert np.abs(ups.pcm_to_float(ups.float_to_pcm(t, ups.PCM_NAMES[fmt]), ups.PCM_NAMES[fmt]) - t).max() <= eps


# ---- EQ -----------------------------------------------------------------------
PROFILES = json.loads((GOLDEN / "g4_eq_profiles.json").read_text())
GRIDS = {"768k": (65537, 131072, 768000.0), "705k": (65537, 131072, 705600.0), "small": (513, 1024, 44100.0 * 16),
         "lin160k": (131073, 262144, 768000.0)}  # the N = 262144 grid of BASELINE configs[4]


@pytest.mark.parametrize("name", sorted(PROFILES))
def test_eq_parser_and_host_response_vs_golden(ups, name):
    g = np.load(GOLDEN / "g4_eq.npz")
    pre, bands = ups.eq_parse(PROFILES[name])
    assert pre == float(g[f"{name}_preamp"])
    np.testing.assert_allclose(bands, g[f"{name}_bands"], rtol=1e-15, atol=0)
    for tag, (bins, fft, fs) in GRIDS.items():
        idx = g[f"{name}_{tag}_idx"]
        np.testing.assert_allclose(ups.eq_response_host(PROFILES[name], bins, fft, fs)[idx], g[f"{name}_{tag}_resp"],
                                   rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ups.eq_magnitude_host(PROFILES[name], bins, fft, fs)[idx], g[f"{name}_{tag}_mag"],
                                   rtol=1e-9, atol=1e-12)


def test_eq_parser_edge_cases_vs_oracle(ups, O):
    texts = [
        "", "# only a comment\n", "Preamp: 0 dB\n", "Preamp: -3.5dB\nFilter 1: ON PK Fc 100 Hz Gain 1 dB Q 2\n",
        "filter 7: on hs fc 8000 hz gain -2.5 db q 0.71\n", "Filter: ON LS 6dB Fc 120 Hz Gain 3 dB\n",
        "Filter 2: ON PK Fc 1000 Hz Gain 3 dB BW Oct 1.5\n", "Filter 3: ON PK Fc 1000 Hz Gain 3 dB BW Oct 0\n",
        "Filter 4: ON PK Fc 0 Hz Gain 3 dB BW 100\n", "Filter 5: ON NOTCH Fc 50 Hz\n",
        "Filter 6: ON PK Fc 2000 Hz Gain 2 dB Q 1.2 BW Oct 3 BW 40 Hz\n", "   \t\r\n;x\nFilter 8: OFF PEQ Fc 10.5 Hz Gain +4 dB Q 3\r\n",
        "garbage line\nFilter 9: ON PK Fc 300 Gain 1 dB Q 1\nPreamp: 2 dB\n",
    ]
    for t in texts:
        ok, pre, bands = O.eq_parse(t)
        got = ups.eq_parse(t)
        if not ok:
            assert got is None, repr(t)
            continue
        assert got is not None and got[0] == pre and len(got[1]) == len(bands), repr(t)
        for row, b in zip(got[1], bands):
            want = [float(b["enabled"]), float(b["type"]), b["frequency"], b["gain"], b["q"], float(b["has_bw_hz"]),
                    b["bw_hz"], float(b["has_bw_oct"]), b["bw_oct"]]
            np.testing.assert_allclose(row, want, rtol=1e-15, atol=0, err_msg=repr(t))


def test_eq_types_and_biquads_vs_oracle(ups, O):
    g = np.load(GOLDEN / "g4_eq.npz")
    assert [ups.eq_filter_type_name(i) for i in range(20)] == list(g["type_names"])
    for s in ["PK", "peak", "Peaking", "modal", "PEQ", "lp", "LOWPASS", "LPQ", "HP", "highpass", "HPQ", "BP", "bandpass",
              "NO", "notch", "AP", "allpass", "LS", "lowshelf", "HS", "HIGHSHELF", "LSC", "HSC", "LSQ", "HSQ", "LS 6dB",
              "ls6db", "LS 12DB", "LS12dB", "HS 6db", "HS6DB", "HS 12DB", "hs12db", "unknown", ""]:
        assert ups.eq_parse_filter_type(s) == O.eq_parse_filter_type(s), s
    for t in range(20):
        for en in (False, True):
            for gain in (0.0, -6.0, 3.3):
                for fs in (44100.0, 768000.0):
                    band = dict(enabled=en, type=t, frequency=997.0, gain=gain, q=1.3)
                    np.testing.assert_allclose(ups.eq_biquad(en, t, 997.0, gain, 1.3, fs), O.eq_biquad(band, fs),
                                               rtol=1e-14, atol=0)


# ---- load-time tables (spectrum.cpp) vs a numpy statement of the same maths ------
def numpy_tables(h, fft, block, L):
    N = fft
    P = L if N % L == 0 else 1
    M, K = N // P, N // P // 2
    ht = np.zeros(N)
    ht[: len(h)] = h
    Gs = np.empty((P, K), np.complex128)
    Gc = np.empty((P, K), np.complex128)
    for p in range(P):
        G = np.fft.fft(ht[p::P])
        Gs[p] = G[:K] / (2 * M)
        Gc[p] = np.conj(G[K - np.arange(K)]) / (2 * M)
    return dict(P=P, M=M, K=K, Gs=Gs.reshape(-1), Gc=Gc.reshape(-1), Wm=np.exp(-2j * np.pi * np.arange(K) / M))


@pytest.mark.parametrize("fft,T,L", [(16, 5, 1), (16, 5, 2), (16, 5, 3), (16, 5, 4), (1024, 257, 8), (4096, 1025, 16)])
def test_tables_match_numpy(ups, make_filter, fft, T, L):
    rng = np.random.default_rng(fft + L)
    h = rng.standard_normal(T).astype(np.float32)
    block = fft - (T - 1)
    t = ups.build_tables(make_filter(h, fft, block, L))
    want = numpy_tables(h, fft, block, L)
    g = t["geometry"]
    assert (g["P"], g["M"], g["K"]) == (want["P"], want["M"], want["K"])
    assert g["S"] * g["P"] == L and g["Oc"] == (fft - block) // g["P"] and g["Bc"] == block // g["P"]
    assert g["n_in"] == block // L and g["B"] == block and g["hist_frames"] == -(-g["Oc"] // g["S"])
    scale = np.abs(want["Gs"]).max()
    assert np.abs(t["Gs"] - want["Gs"]).max() <= 1e-7 * scale
    assert np.abs(t["Gc"] - want["Gc"]).max() <= 1e-7 * scale
    assert np.abs(t["Wm"] - want["Wm"]).max() <= 6e-8
    # twiddle table: q-th block holds exp(-2 pi i k / 2^q), k < 2^(q-1)
    for q in range(1, g["log2k"] + 1):
        k = np.arange(1 << (q - 1))
        blk = t["tw"][(1 << (q - 1)) - 1:(1 << q) - 1]
        assert np.abs(blk - np.exp(-2j * np.pi * k / (1 << q))).max() <= 6e-8


def test_tables_real_filter_with_eq_and_compat(ups, O):
    path = GOLDEN / "filters" / "filter_48k_16x_80000_min_phase.json"
    h, taps, fft, block, L = O.read_filter(path)
    text = PROFILES["opra10"]
    t = ups.build_tables(path, apo_text=text, fs_out=768000.0)
    want = numpy_tables(O.eq_fold_fir(h, text, 768000.0), fft, block, L)  # the EQ is part of the FIR: same length
    scale = np.abs(want["Gs"]).max()
    assert np.abs(t["Gs"] - want["Gs"]).max() <= 2e-7 * scale
    assert np.abs(t["Gc"] - want["Gc"]).max() <= 2e-7 * scale
    # reference-compatible spectrum: tables built from the oracle's own fp32 H
    tc = ups.build_tables(path, flags=ups.LOAD_REF_COMPAT_SPECTRUM)
    H = O.OracleUpsampler(h, taps, fft, block, L).spectrum().astype(np.complex128)
    he = np.fft.ifft(H).real
    P, M, K = L, fft // L, fft // L // 2
    Gs = np.concatenate([np.fft.fft(he[p::P])[:K] for p in range(P)]) / (2 * M)
    assert np.abs(tc["Gs"] - Gs).max() <= 2e-7 * np.abs(Gs).max()
    plain = ups.build_tables(path)
    assert np.abs(tc["Gs"] - plain["Gs"]).max() > 1e-5 * np.abs(Gs).max()  # the reference's H really is different
