#!/usr/bin/env python3
"""Golden vectors for the OPRA record -> APO text conversion (tests/golden/g8_opra.json).

Run in the build container only (needs /root/reference): imports the reference's own converter
(scripts/integration/opra.py: convert_opra_to_apo, apply_modern_target_correction, EqProfile.to_apo_format) and records
its output for a set of hand-written records (no OPRA catalogue data: the records below are synthetic)."""
import json
import sys
from pathlib import Path

REF = Path("/root/reference")
sys.path.insert(0, str(REF))
from scripts.integration.opra import apply_modern_target_correction, convert_opra_to_apo  # noqa: E402

RECORDS = [
    {"name": "flat", "author": "x", "details": "", "parameters": {"gain_db": 0.0, "bands": []}},
    {"name": "ten bands", "author": "someone", "details": "measured on a rig",
     "parameters": {"gain_db": -6.2, "bands": [
         {"type": "low_shelf", "frequency": 105.0, "gain_db": 5.5, "q": 0.7},
         {"type": "peak_dip", "frequency": 190, "gain_db": -2.95, "q": 0.55},
         {"type": "peak_dip", "frequency": 1234.56, "gain_db": 1.25, "q": 1.414},
         {"type": "peak_dip", "frequency": 2999.95, "gain_db": -0.05, "q": 2.005},
         {"type": "peak_dip", "frequency": 5400, "gain_db": 3.45, "q": 4},
         {"type": "high_shelf", "frequency": 10000, "gain_db": -4.0, "q": 0.7071},
         {"type": "low_pass", "frequency": 18000, "slope": 12},
         {"type": "high_pass", "frequency": 20.5, "slope": 24, "gain_db": 3.0},
         {"type": "band_pass", "frequency": 500, "q": 2.0},
         {"type": "band_stop", "frequency": 60, "q": 10.0},
     ]}},
    {"name": "defaults", "parameters": {"bands": [
        {"type": "peak_dip"},
        {"type": "peak_dip", "frequency": 800.0, "gain_db": 2.0},
        {"type": "low_pass", "frequency": 15000.0},
        {"type": "high_pass", "frequency": 30.0, "slope": 18},
        {"type": "low_pass", "frequency": 12000.0, "slope": 48},
        {"type": "high_pass", "frequency": 25.0, "slope": 36.0},
        {"type": "notch", "frequency": 50.0},
        {"frequency": 70.0, "gain_db": 1.0, "q": 1.0},
    ]}},
    {"name": "rounding", "parameters": {"gain_db": 0.05, "bands": [
        {"type": "peak_dip", "frequency": 0.25, "gain_db": 0.25, "q": 0.125},
        {"type": "peak_dip", "frequency": 19999.96, "gain_db": -0.04, "q": 0.005},
        {"type": "low_shelf", "frequency": 1e3, "gain_db": 12, "q": 1},
        {"type": "high_shelf", "frequency": 8.05e3, "gain_db": -11.95, "q": 0.995},
    ]}},
    {"name": "preamp cancels with the correction", "parameters": {"gain_db": 2.8, "bands": [
        {"type": "peak_dip", "frequency": 3000, "gain_db": -3, "q": 1.2}]}},
    {"name": "no parameters"},
]


def main():
    cases = []
    for rec in RECORDS:
        base = convert_opra_to_apo(rec)
        cases.append({"record": rec, "apo": base.to_apo_format(),
                      "apo_modern_target": apply_modern_target_correction(base).to_apo_format()})
    out = Path(__file__).with_name("g8_opra.json")
    out.write_text(json.dumps({"generator": "tests/golden/make_golden_opra.py (reference scripts/integration/opra.py)",
                               "cases": cases}, indent=1) + "\n")
    print(f"wrote {out} ({len(cases)} cases)")


if __name__ == "__main__":
    main()
