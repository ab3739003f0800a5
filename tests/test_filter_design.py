"""The offline filter generator (totton-rasp-gpu-dsp_amd/filter_design.py) against the REFERENCE's generator
(scripts/filters/generate_filter.py, generate_minimum_phase.py, generate_linear_phase.py), through
tests/golden/g7_filter_design.json -- written in the build container by tests/golden/make_golden_filter_design.py, which
imports the reference's Python -- and against the files this repo ships under data/coefficients.

Cases follow the reference's tests/python/test_generate_filter.py: tap padding (:185-186), naming incl. "2m"
(:441-...), shipped-file inventory (:407-441), DC gain L x 0.99 (:389-400,700-718)."""
from __future__ import annotations

import importlib.util
import json

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

spec = importlib.util.spec_from_file_location("filter_design", ROOT / "totton-rasp-gpu-dsp_amd" / "filter_design.py")
fd = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fd)
G7 = json.loads((GOLDEN / "g7_filter_design.json").read_text())
DATA = ROOT / "data" / "coefficients"


def test_padded_taps_table():
    for n, r, want in G7["padded_taps"]:
        assert fd.padded_taps(n, r) == want, (n, r)
    assert fd.padded_taps(1600, 16) == 1601       # the reference's own example
    with pytest.raises(ValueError):
        fd.padded_taps(1600, 0)


def test_names_families_and_geometry():
    assert {k: (v["input_rate"], v["stopband"]) for k, v in G7["multi_rate"].items()} == fd.MULTI_RATE
    for e in G7["names"]:
        family, ratio = e["key"].split("_")
        ratio = int(ratio[:-1])
        phase = "min" if e["suffix"] == "min_phase" else "linear"
        assert family == e["family"]
        assert fd.base_name(family, ratio, e["taps"], phase) == e["base_name"]
        assert fd.taps_label(e["taps"]) == e["taps_label"]
        aligned = fd.padded_taps(e["taps"], ratio)
        assert aligned == e["aligned"] and fd.geometry(aligned) == (e["fft_size"], e["block_size"])
        assert fd.MULTI_RATE[e["key"]][0] * ratio == e["output_rate"]
    assert fd.base_name("44k", 16, 2_000_000, "min") == "filter_44k_16x_2m_min_phase"
    assert fd.base_name("48k", 8, 640_000, "linear") == "filter_48k_8x_2m_linear_phase"


@pytest.mark.parametrize("tag", sorted(G7["designs"]))
def test_small_designs_match_the_reference_generator(tag):
    """Same recipe, same numpy/scipy in this container: the linear prototype is the same firwin call, the
    minimum-phase taps come from this repo's own statement of the cepstral folding the reference calls
    (scipy.signal.minimum_phase, method "homomorphic", same n_fft, same floor, same window). Agreement to 1e-12 of the
    peak tap (observed: bit-identical). Against another scipy the reference itself moves by ~5e-4."""
    d = G7["designs"][tag]
    family, ratio = d["key"].split("_")
    ratio = int(ratio[:-1])
    h = fd.design(d["n_taps"], ratio, family, d["phase"], beta=d["kaiser_beta"])
    want = np.array(d["taps"])
    assert h.shape == want.shape and len(h) == fd.padded_taps(d["n_taps"], ratio)
    tol = 1e-12
    assert np.abs(h - want).max() <= tol * np.abs(want).max()
    assert abs(h.sum() - ratio * 0.99) <= 1e-9 and abs(d["normalization"]["normalized_dc_gain"] - ratio * 0.99) <= 1e-9
    input_rate, stop = fd.MULTI_RATE[d["key"]]
    v = fd.validate(h, ratio, input_rate, 20000, stop)
    r = d["validation"]
    for k in ("peak_position", "peak_threshold_samples", "is_minimum_phase", "is_symmetric", "actual_taps", "meets_stopband_spec"):
        assert v[k] == r[k], k
    assert abs(v["input_band_peak"] - r["input_band_peak"]) <= 1e-7 * r["input_band_peak"]
    assert abs(v["passband_ripple_db"] - r["passband_ripple_db"]) <= 1e-6 + 1e-3 * r["passband_ripple_db"]
    # the deepest stop-band point of a short filter is a -200..-300 dB null: compare on the dB scale, loosely
    assert abs(v["stopband_attenuation_db"] - r["stopband_attenuation_db"]) <= 3.0


@pytest.mark.parametrize("name", sorted(G7["shipped"]))
def test_validator_reproduces_the_shipped_sidecar_metrics(name):
    """The byte copies of the reference's shipped filters (tests/golden/filters) through this repo's validator: the
    numbers the reference wrote into their sidecars (data/coefficients/filter_44k_2x_80000_min_phase.json:10-19)."""
    meta = json.loads((GOLDEN / "filters" / f"{name}.json").read_text())
    h = np.fromfile(GOLDEN / "filters" / f"{name}.bin", "<f4")
    v = fd.validate(h, meta["upsample_ratio"], meta["sample_rate_input"], meta["passband_end_hz"], meta["stopband_start_hz"],
                    meta["target_stopband_attenuation_db"])
    ref = G7["shipped"][name]["validation"]            # reference validator on the same float32 file, run here
    side = G7["shipped"][name]["sidecar_validation"]   # what the reference recorded on its fp64 taps at generation time
    for k in ("peak_position", "peak_threshold_samples", "is_minimum_phase", "is_symmetric", "actual_taps", "meets_stopband_spec"):
        assert v[k] == ref[k] == side[k], k
    assert abs(v["input_band_peak"] - ref["input_band_peak"]) <= 1e-9 and abs(v["input_band_peak"] - side["input_band_peak"]) <= 1e-6
    assert abs(v["stopband_attenuation_db"] - ref["stopband_attenuation_db"]) <= 1e-6
    assert v["stopband_attenuation_db"] >= 140.0
    assert abs(v["passband_ripple_db"] - ref["passband_ripple_db"]) <= 1e-9


EXPECTED_80K = [f"filter_{fam}_{r}x_80000_min_phase" for fam in ("44k", "48k") for r in (16, 8, 4, 2)]


def test_shipped_inventory_metadata_and_dc_gain():
    """All eight multi-rate 80k-tap files exist with their sidecars (test_generate_filter.py:407-441), every file obeys
    the format rules the loader enforces, and its DC gain is L x 0.99 (:389-400)."""
    for base in EXPECTED_80K + ["filter_48k_8x_160000_linear_phase"]:
        meta = json.loads((DATA / f"{base}.json").read_text())
        h = np.fromfile(DATA / f"{base}.bin", "<f4")
        L = meta["upsample_factor"]
        specified = 160000 if "160000" in base else 80000
        assert meta["coefficients_bin"] == f"{base}.bin" and meta["output_basename"] == base
        assert len(h) == meta["taps"] == fd.padded_taps(specified, L) and (len(h) - 1) % L == 0
        assert (meta["fft_size"], meta["block_size"]) == fd.geometry(len(h))
        assert meta["block_size"] % L == 0 and meta["fft_size"] - meta["block_size"] == len(h) - 1
        assert np.isfinite(h).all() and np.isclose(h.astype(np.float64).sum(), L * 0.99, rtol=1e-6)
        key = base.split("_")[1] + "_" + base.split("_")[2]
        assert (meta["sample_rate_input"], meta["stopband_start_hz"]) == fd.MULTI_RATE[key]
        assert meta["sample_rate_output"] == meta["sample_rate_input"] * L
        v = meta["validation_results"]
        assert v["meets_stopband_spec"] and v["stopband_attenuation_db"] >= 140.0 and v["actual_taps"] == len(h)
        assert v["is_symmetric"] == ("linear" in base) and v["is_minimum_phase"] == ("min_phase" in base)
        assert v["passband_ripple_db"] < 1e-3


def test_shipped_filter_is_what_the_generator_produces(tmp_path):
    """Re-run the generator for one shipped file: identical float32 bytes (the data directory is reproducible)."""
    assert fd.main(["--taps", "80000", "--ratio", "4", "--family", "44k", "--phase", "min", "--out-dir", str(tmp_path)]) == 0
    base = "filter_44k_4x_80000_min_phase"
    assert (tmp_path / f"{base}.bin").read_bytes() == (DATA / f"{base}.bin").read_bytes()
    a, b = json.loads((tmp_path / f"{base}.json").read_text()), json.loads((DATA / f"{base}.json").read_text())
    for k in ("taps", "fft_size", "block_size", "upsample_factor", "coefficients_bin", "n_taps_actual", "stopband_start_hz"):
        assert a[k] == b[k]
