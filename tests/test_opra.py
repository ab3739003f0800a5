"""OPRA EQ record -> Equalizer APO text (csrc/host/opra.cpp, mi_opra_to_apo) against the reference's own converter
(golden: tests/golden/g8_opra.json, produced by tests/golden/make_golden_opra.py from scripts/integration/opra.py)."""
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
G8 = json.loads((ROOT / "tests" / "golden" / "g8_opra.json").read_text())


@pytest.mark.parametrize("case", G8["cases"], ids=[c["record"]["name"] for c in G8["cases"]])
def test_apo_text_equals_the_reference_converter(ups, case):
    assert ups.opra_to_apo(case["record"]) == case["apo"]
    assert ups.opra_to_apo(json.dumps(case["record"]), modern_target=True) == case["apo_modern_target"]


def test_converted_profile_parses_with_the_band_count_the_reference_keeps(ups):
    """The text goes straight into the EQ path: every converted band is a parsed band (PK/LS/HS evaluate, LP/HP parse and
    evaluate as bypass exactly as in the reference, eq_to_fir.cpp:60-66)."""
    case = next(c for c in G8["cases"] if c["record"]["name"] == "ten bands")
    text = ups.opra_to_apo(case["record"])
    lines = [ln for ln in text.split("\n") if ln.startswith("Filter ")]
    assert len(lines) == 8  # band_pass and band_stop are dropped
    parsed = ups.eq_parse(text)
    assert parsed is not None
    preamp, bands = parsed
    assert preamp == pytest.approx(-6.2) and bands.shape[0] == 8


def test_malformed_records_are_reported(ups):
    for bad in ("", "{", '{"parameters": {"bands": [}}', "[1, 2]", '{"a": 1} trailing'):
        with pytest.raises(ups.UpsamplerError):
            ups.opra_to_apo(bad)
    # wrong-typed fields fall back to the defaults instead of failing
    assert ups.opra_to_apo({"parameters": {"gain_db": None, "bands": [{"type": "peak_dip", "frequency": None, "q": None}]}}) == \
        "Filter 1: ON PK Fc 1000.0 Hz Gain 0.0 dB Q 1.00"
