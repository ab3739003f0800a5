#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json from the COMPILED REFERENCE (oracle/_ref).

Run in the build container only (needs /root/reference to have been compiled
by `make -C oracle`): the reference cannot travel to the GPU box, these small
fixtures can. Every vector here is an output of the reference's own code
(`VulkanStreamingUpsampler` CPU-fallback build, `eq_parser.cpp`,
`eq_to_fir.cpp`) or an fp64 truth computed with scipy, on inputs that are
stored next to it or reproducible from the stored seed.

  G1  known-answer geometry of tests/cpp/test_vulkan_upsampler.cpp:42-195
  G2  257 random taps, fft 1024, block 768, L in {1,2,4,8,16}, 4 blocks (full)
  G3  real 131072/262144-point geometry: three shipped 80k-tap filters
      (tests/golden/filters, byte copies of reference data files) and this
      repo's 160001-tap linear filter; 3 blocks of 0.2*N(0,1); sparse probes,
      per-block fp64 sums, fp64 truth and the Vulkan-path simulation
      (reference-computed H x fp64 signal FFTs) at the same probes
  G4  EQ: parser fields and per-bin responses from the reference's EQ code
  G6  LoadFilter error strings on malformed sidecars
"""
from __future__ import annotations

import json
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent
KNOWN_TAPS = np.array([1, 2, 3, 2, 1], dtype=np.float32)


def write_filter(d: Path, name: str, taps: np.ndarray, fft: int, block: int, factor: int | None) -> Path:
    taps.astype("<f4").tofile(d / f"{name}.bin")
    meta = dict(coefficients_bin=f"{name}.bin", taps=int(len(taps)), fft_size=fft, block_size=block)
    if factor is not None:
        meta["upsample_factor"] = factor
    (d / f"{name}.json").write_text(json.dumps(meta))
    return d / f"{name}.json"


def probe_indices(block: int) -> np.ndarray:
    idx = np.concatenate([np.arange(256), np.arange(0, block, 997), np.arange(block - 256, block)])
    return np.unique(idx)


def real_input(seed: int, n: int) -> np.ndarray:
    return (np.random.default_rng(seed).standard_normal(n) * 0.2).astype(np.float32)


def vulkan_sim(x, H_ref, fft, block, factor, nblocks):
    """Real Vulkan path = reference-computed (inaccurate) H times an accurate
    signal FFT: fp64 FFTs around the fp32 H."""
    ov = fft - block
    nin = block // factor
    overlap = np.zeros(ov)
    out = np.empty((nblocks, block))
    Hd = H_ref.astype(np.complex128)
    for b in range(nblocks):
        t = np.zeros(fft)
        t[:ov] = overlap
        t[ov::factor][:nin] = x[b * nin : (b + 1) * nin]
        out[b] = np.fft.ifft(np.fft.fft(t) * Hd).real[ov:]
        overlap = t[fft - ov :].copy()
    return out


def g1(tmp: Path):
    res = {}
    for L in (1, 2):
        p = write_filter(tmp, f"g1_{L}", KNOWN_TAPS, 16, 12, L)
        r = O.RefUpsampler()
        assert r.load_filter(p)[0]
        nin = 12 // L
        imp = np.zeros(nin, np.float32)
        imp[4 if L == 1 else 2] = 1.0
        res[f"L{L}_impulse_in"] = imp
        res[f"L{L}_impulse_out"] = r.process_block(imp)
        r.reset()
        a = np.arange(1, nin + 1, dtype=np.float32)
        b = np.arange(101, 101 + nin, dtype=np.float32)
        res[f"L{L}_iota_in"] = np.stack([a, b])
        res[f"L{L}_iota_out"] = np.stack([r.process_block(a), r.process_block(b)])
    np.savez(OUT / "g1_known_answer.npz", taps=KNOWN_TAPS, **res)


def g2(tmp: Path):
    rng = np.random.default_rng(0)
    taps = rng.standard_normal(257).astype(np.float32)
    res = dict(taps=taps)
    for L in (1, 2, 4, 8, 16):
        p = write_filter(tmp, f"g2_{L}", taps, 1024, 768, L)
        r = O.RefUpsampler()
        assert r.load_filter(p)[0]
        nin = 768 // L
        x = rng.standard_normal((4, nin)).astype(np.float32)
        y = np.stack([r.process_block(x[b]) for b in range(4)])
        res[f"L{L}_in"] = x
        res[f"L{L}_ref"] = y
        res[f"L{L}_truth"] = O.truth_stream(x.reshape(-1), taps, L, 4, 768)
    np.savez(OUT / "g2_mid.npz", **res)


def g3():
    cases = [
        ("44k_2x", OUT / "filters" / "filter_44k_2x_80000_min_phase.json", 3002),
        ("44k_4x", OUT / "filters" / "filter_44k_4x_80000_min_phase.json", 3004),
        ("48k_16x", OUT / "filters" / "filter_48k_16x_80000_min_phase.json", 3016),
        ("48k_8x_lin160k", ROOT / "data" / "coefficients" / "filter_48k_8x_160000_linear_phase.json", 3108),
    ]
    res = {}
    manifest = {}
    for name, path, seed in cases:
        h, taps, fft, block, L = O.read_filter(path)
        r = O.RefUpsampler()
        ok, msg = r.load_filter(path)
        assert ok, msg
        nb = 3
        nin = block // L
        x = real_input(seed, nb * nin)
        y = np.stack([r.process_block(x[b * nin : (b + 1) * nin]) for b in range(nb)])
        truth = O.truth_stream(x, h, L, nb, block)
        H_ref = O.OracleUpsampler(h, taps, fft, block, L).spectrum()
        sim = vulkan_sim(x.astype(np.float64), H_ref, fft, block, L, nb)
        idx = probe_indices(block)
        res[f"{name}_idx"] = idx
        res[f"{name}_ref"] = y[:, idx]
        res[f"{name}_truth"] = truth[:, idx]
        res[f"{name}_vksim"] = sim[:, idx]
        res[f"{name}_ref_sum"] = y.astype(np.float64).sum(axis=1)
        res[f"{name}_ref_l2"] = np.sqrt((y.astype(np.float64) ** 2).sum(axis=1))
        res[f"{name}_truth_sum"] = truth.sum(axis=1)
        res[f"{name}_truth_l2"] = np.sqrt((truth**2).sum(axis=1))
        res[f"{name}_truth_max"] = np.abs(truth).max(axis=1)
        res[f"{name}_x_head"] = x[:8]
        res[f"{name}_x_sum"] = np.float64(x.astype(np.float64).sum())
        manifest[name] = dict(filter=str(Path(path).relative_to(ROOT)), seed=seed, blocks=nb, taps=taps, fft=fft,
                              block=block, factor=L,
                              ref_vs_truth_max=float(np.abs(y - truth).max() / np.abs(truth).max()),
                              vksim_vs_truth_max=float(np.abs(sim - truth).max() / np.abs(truth).max()))
        print(name, manifest[name])
    np.savez(OUT / "g3_real.npz", **res)
    (OUT / "g3_real.json").write_text(json.dumps(manifest, indent=2) + "\n")


EQ_PROFILES = {
    # tests/cpp/test_eq_parser_smoke.cpp:30-33
    "smoke": "Preamp: -6 dB\nFilter 1: ON PK Fc 1000 Hz Gain -3 dB Q 1.41\n"
             "Filter: OFF LS Fc 80 Hz Gain 2 dB Q 0.7\nFilter 3: ON PK Fc 500 Hz Gain -2 dB BW 100 Hz\n",
    # 10-band OPRA-like profile (shape of scripts/integration/opra.py output)
    "opra10": "Preamp: -6.2 dB\n"
              "Filter 1: ON LS Fc 105 Hz Gain 5.5 dB Q 0.70\n"
              "Filter 2: ON PK Fc 210 Hz Gain -2.3 dB Q 0.90\n"
              "Filter 3: ON PK Fc 640 Hz Gain 1.1 dB Q 1.40\n"
              "Filter 4: ON PK Fc 1500 Hz Gain -1.8 dB Q 2.00\n"
              "Filter 5: ON PK Fc 2900 Hz Gain 3.2 dB Q 2.50\n"
              "Filter 6: ON PK Fc 4300 Hz Gain -4.0 dB Q 3.00\n"
              "Filter 7: ON PK Fc 6100 Hz Gain 2.6 dB Q 4.00\n"
              "Filter 8: ON PK Fc 8200 Hz Gain -3.1 dB Q 3.50\n"
              "Filter 9: ON HS Fc 10000 Hz Gain -2.5 dB Q 0.70\n"
              "Filter 10: ON PK Fc 13000 Hz Gain 1.9 dB BW Oct 0.5\n",
    # exercises bypass types, gain 0, OFF and every optional token
    "mixed": "# comment\n; another\nPreamp: +1.5 dB\n"
             "Filter 1: ON LP Fc 18000 Hz\n"
             "Filter 2: ON HSC Fc 9000 Hz Gain 3 dB Q 0.7\n"
             "Filter 3: ON PK Fc 250 Hz Gain 0 dB Q 2\n"
             "Filter 4: OFF PK Fc 900 Hz Gain 6 dB Q 1\n"
             "Filter 5: ON peaking Fc 3000 Gain -2.5 dB BW 600\n"
             "Filter 6: ON HS Fc 12000 Hz Gain 4 dB\n",
}


def g4():
    res = {}
    for name, text in EQ_PROFILES.items():
        pre, bands = O.ref_eq_parse(text)
        res[f"{name}_preamp"] = np.float64(pre)
        res[f"{name}_bands"] = bands
        # "lin160k": the N = 262144 grid of BASELINE configs[4] (160 001-tap linear filter, 768 kHz out)
        for tag, (bins, fft, fs) in {"768k": (65537, 131072, 768000.0), "705k": (65537, 131072, 705600.0),
                                     "small": (513, 1024, 44100.0 * 16),
                                     "lin160k": (131073, 262144, 768000.0)}.items():
            full = O.ref_eq_response(text, bins, fft, fs)
            idx = np.unique(np.round(np.logspace(0, np.log10(bins - 1), 64)).astype(int))
            idx = np.unique(np.concatenate([[0], idx, [bins - 1]]))
            res[f"{name}_{tag}_idx"] = idx
            res[f"{name}_{tag}_resp"] = full[idx]
            res[f"{name}_{tag}_mag"] = O.ref_eq_magnitude(text, bins, fft, fs)[idx]
    res["type_names"] = np.array([O.ref_lib().ref_eq_filter_type_name(i).decode() for i in range(20)])
    np.savez(OUT / "g4_eq.npz", **res)
    (OUT / "g4_eq_profiles.json").write_text(json.dumps(EQ_PROFILES, indent=2) + "\n")


def g6(tmp: Path):
    """LoadFilter failure strings (vulkan_streaming_upsampler.cpp:606-732)."""
    d = tmp / "g6"
    d.mkdir()
    KNOWN_TAPS.astype("<f4").tofile(d / "c.bin")
    cases = {
        "missing_file": None,
        "no_bin_key": '{"taps": 5, "fft_size": 16, "block_size": 12}',
        "zero_taps": '{"coefficients_bin": "c.bin", "taps": 0, "fft_size": 16, "block_size": 12}',
        "missing_block": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 16}',
        "fft_not_pow2": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 18, "block_size": 14}',
        "block_ge_fft": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 16, "block_size": 16}',
        "overlap_mismatch": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 16, "block_size": 11}',
        "block_not_div": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 16, "block_size": 12, "upsample_factor": 5}',
        "bin_missing": '{"coefficients_bin": "nope.bin", "taps": 5, "fft_size": 16, "block_size": 12}',
        "bin_wrong_size": '{"coefficients_bin": "c.bin", "taps": 9, "fft_size": 16, "block_size": 8}',
        "factor_zero_ok": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 16, "block_size": 12, "upsample_factor": 0}',
        "factor_missing_ok": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 16, "block_size": 12}',
        "factor3_ok": '{"coefficients_bin": "c.bin", "taps": 5, "fft_size": 16, "block_size": 12, "upsample_factor": 3}',
        "extra_keys_ok": '{"n_taps_actual": 7, "taps": 5, "x": {"fft_size": 99}, "fft_size": 16, "block_size": 12, '
                         '"coefficients_bin": "c.bin", "upsample_factor": 2}',
        "negative_number": '{"coefficients_bin": "c.bin", "taps": -5, "fft_size": 16, "block_size": 12}',
        "float_number": '{"coefficients_bin": "c.bin", "taps": 5.0, "fft_size": 16, "block_size": 12}',
    }
    out = {}
    for name, body in cases.items():
        p = d / f"{name}.json"
        if body is not None:
            p.write_text(body)
        r = O.RefUpsampler()
        ok, msg = r.load_filter(p)
        out[name] = dict(body=body, ok=ok, message=msg.replace(str(d), "<DIR>"),
                         config=list(r.config()) if ok else None)
    (OUT / "g6_load_errors.json").write_text(json.dumps(out, indent=2) + "\n")


def main():
    assert O.have_ref(), "run `make -C oracle` where /root/reference exists"
    only = set(sys.argv[1:])  # e.g. `make_golden.py g4` regenerates one family
    with tempfile.TemporaryDirectory() as t:
        tmp = Path(t)
        for name, fn in (("g1", g1), ("g2", g2), ("g6", g6)):
            if not only or name in only:
                fn(tmp)
    for name, fn in (("g3", g3), ("g4", g4)):
        if not only or name in only:
            fn()
    print("golden written to", OUT)


if __name__ == "__main__":
    main()
