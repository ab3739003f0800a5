#!/usr/bin/env python3
"""Generate tests/golden/g7_filter_design.json from the REFERENCE's filter generator.

Run in the build container only: it imports /root/reference/scripts/filters/* (numpy/scipy, CPU; the CuPy branch is
inactive here) and records what pins this repo's restatement of the recipe (totton-rasp-gpu-dsp_amd/filter_design.py):

  * compute_padded_taps over a grid, FilterConfig.base_name / taps_label / family, the exporter's geometry rule
  * small complete designs through the reference's own designers + normalize_coefficients (full tap vectors)
  * FilterValidator metrics of those designs and of the byte copies of the shipped filters under tests/golden/filters

The reference cannot travel to the GPU box; this small JSON can.
"""
from __future__ import annotations

import contextlib
import io
import json
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REF))

with contextlib.redirect_stdout(io.StringIO()):
    from scripts.filters.generate_filter import (MULTI_RATE_CONFIGS, FilterConfig, FilterValidator,  # noqa: E402
                                                  compute_padded_taps, normalize_coefficients)
    from scripts.filters.generate_linear_phase import LinearPhaseDesigner  # noqa: E402
    from scripts.filters.generate_minimum_phase import MinimumPhaseDesigner  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def clean(v):
    if isinstance(v, dict):
        return {k: clean(x) for k, x in v.items()}
    if isinstance(v, (np.bool_, bool)):
        return bool(v)
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.floating):
        return float(v)
    return v


def main():
    g = {"multi_rate": MULTI_RATE_CONFIGS}
    g["padded_taps"] = [[n, r, int(compute_padded_taps(n, r))] for n in (1, 2, 15, 16, 17, 1000, 1600, 1601, 80000, 160000,
                                                                       640000, 2000000)
                        for r in (1, 2, 4, 8, 16)]
    names = []
    for key, c in MULTI_RATE_CONFIGS.items():
        for taps in (80000, 160000, 640000, 2000000, 1600):
            for suffix in ("min_phase", "linear_phase"):
                cfg = FilterConfig(n_taps=taps, input_rate=c["input_rate"], upsample_ratio=c["ratio"],
                                   stopband_start=c["stopband"], phase_suffix=suffix)
                aligned = cfg.aligned_taps
                fft = 2 ** int(np.ceil(np.log2(aligned)))   # FilterExporter._export_metadata
                names.append(dict(key=key, taps=taps, suffix=suffix, family=cfg.family, base_name=cfg.base_name,
                                  taps_label=cfg.taps_label, aligned=int(aligned), fft_size=int(fft),
                                  block_size=int(fft - (aligned - 1)), output_rate=int(cfg.output_rate)))
    g["names"] = names
    designs = {}
    for tag, taps, key, phase in [("min_1600_44k_16x", 1600, "44k_16x", "min"), ("min_4000_48k_4x", 4000, "48k_4x", "min"),
                                  ("lin_2000_48k_8x", 2000, "48k_8x", "linear"), ("min_801_44k_2x", 801, "44k_2x", "min")]:
        c = MULTI_RATE_CONFIGS[key]
        cfg = FilterConfig(n_taps=taps, input_rate=c["input_rate"], upsample_ratio=c["ratio"], stopband_start=c["stopband"],
                           kaiser_beta=25.0, stopband_attenuation_db=140,
                           phase_suffix="min_phase" if phase == "min" else "linear_phase")
        if phase == "min":
            h, h_lin = quiet(MinimumPhaseDesigner(cfg).design)
        else:
            h = quiet(LinearPhaseDesigner(cfg).design)
        hn, norm = quiet(normalize_coefficients, h, target_dc_gain=cfg.target_dc_gain, dc_gain_factor=cfg.dc_gain_factor)
        val = quiet(FilterValidator(cfg).validate, hn)
        designs[tag] = dict(n_taps=taps, key=key, phase=phase, kaiser_beta=25.0, taps=[float(v) for v in hn],
                            normalization=clean(norm), validation=clean(val))
    g["designs"] = designs
    shipped = {}
    for name in ("filter_44k_2x_80000_min_phase", "filter_44k_4x_80000_min_phase", "filter_48k_16x_80000_min_phase"):
        meta = json.loads((OUT / "filters" / f"{name}.json").read_text())
        h = np.fromfile(OUT / "filters" / f"{name}.bin", "<f4").astype(np.float64)
        cfg = FilterConfig(n_taps=meta["n_taps_specified"], input_rate=meta["sample_rate_input"],
                           upsample_ratio=meta["upsample_ratio"], stopband_start=meta["stopband_start_hz"],
                           kaiser_beta=meta["kaiser_beta"], stopband_attenuation_db=meta["target_stopband_attenuation_db"])
        shipped[name] = dict(validation=clean(quiet(FilterValidator(cfg).validate, h)),
                             sidecar_validation={k: v for k, v in meta["validation_results"].items() if k != "normalization"})
    g["shipped"] = shipped
    (OUT / "g7_filter_design.json").write_text(json.dumps(g) + "\n")
    print("wrote", OUT / "g7_filter_design.json", (OUT / "g7_filter_design.json").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
