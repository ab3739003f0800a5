"""Offline FIR design for the upsampler: produces the on-disk format the hot
path consumes (raw float32-LE taps + JSON sidecar with the geometry keys).

Host-only, CPU, numpy/scipy -- the reference's generator is Python too. This
is a compact restatement of the recipe, not of the reference's class tree:

* tap alignment so (taps-1) % L == 0   scripts/filters/generate_filter.py:462-470
* Kaiser-windowed sinc, cutoff midway between pass- and stop-band, odd length
                                        scripts/filters/generate_linear_phase.py:34-50
* homomorphic minimum phase, n_fft = 2^ceil(log2(8*len)), result truncated or
  zero-padded to the aligned tap count   scripts/filters/generate_minimum_phase.py:67-117
* DC gain normalised to L * 0.99         scripts/filters/generate_filter.py:473-519
* sidecar geometry fft = 2^ceil(log2 taps), block = fft - (taps-1)
                                        scripts/filters/generate_filter.py:207-238
* file name filter_{44k|48k}_{L}x_{taps|2m}_{min|linear}_phase
                                        scripts/filters/generate_filter.py:96-120

Usage:  python filter_design.py --taps 80000 --ratio 4 --family 44k --phase min \
            --out-dir data/coefficients
"""
from __future__ import annotations

import argparse
import json
import math
from datetime import datetime
from pathlib import Path

import numpy as np

# (input_rate, stop-band start) per family/ratio, as the reference ships them:
# every ratio lands on 705.6 kHz / 768 kHz output.
MULTI_RATE = {
    "44k_16x": (44100, 22050), "44k_8x": (88200, 44100), "44k_4x": (176400, 88200), "44k_2x": (352800, 176400),
    "48k_16x": (48000, 24000), "48k_8x": (96000, 48000), "48k_4x": (192000, 96000), "48k_2x": (384000, 192000),
}


def padded_taps(n_taps: int, ratio: int) -> int:
    if ratio <= 0:
        raise ValueError("ratio must be a positive integer")
    rem = (n_taps - 1) % ratio
    return n_taps if rem == 0 else n_taps + (ratio - rem)


def taps_label(n_taps: int) -> str:
    return "2m" if n_taps in (640_000, 2_000_000) else str(n_taps)


def base_name(family: str, ratio: int, n_taps: int, phase: str) -> str:
    return f"filter_{family}_{ratio}x_{taps_label(n_taps)}_{phase}_phase"


def geometry(actual_taps: int) -> tuple[int, int]:
    fft = 2 ** int(math.ceil(math.log2(actual_taps)))
    return fft, fft - (actual_taps - 1)


def design_linear(n_taps: int, ratio: int, input_rate: int, passband_end: float, stopband_start: float,
                  beta: float) -> np.ndarray:
    from scipy import signal

    aligned = padded_taps(n_taps, ratio)
    numtaps = aligned if aligned % 2 == 1 else aligned + 1
    cutoff = 0.5 * (passband_end + stopband_start) / (0.5 * input_rate * ratio)
    return signal.firwin(numtaps=numtaps, cutoff=cutoff, window=("kaiser", beta), fs=1.0, scale=True)


def minimum_phase_homomorphic(h_linear: np.ndarray) -> np.ndarray:
    """Cepstral folding, step for step what the reference's call computes
    (scipy.signal.minimum_phase(h, method='homomorphic', n_fft=2^ceil(log2(8 len))), generate_minimum_phase.py:67-86):
    |H| on the n_fft grid, a floor of 1e-7 x the smallest non-zero magnitude ADDED before the log, half the log
    (the square root of the magnitude response), the causal cepstrum window 1, 2, ..., 2, 0, ..., 0 (for an even
    n_fft the Nyquist term is dropped), exp of its transform; output length (len+1)//2."""
    n_fft = 2 ** int(math.ceil(math.log2(len(h_linear) * 8)))
    mag = np.abs(np.fft.fft(h_linear, n_fft))
    mag += 1e-7 * mag[mag > 0].min()
    ceps = np.fft.ifft(0.5 * np.log(mag)).real
    win = np.zeros(n_fft)
    win[0] = 1.0
    win[1 : n_fft // 2] = 2.0
    if n_fft % 2:
        win[n_fft // 2] = 1.0
    h_min = np.fft.ifft(np.exp(np.fft.fft(ceps * win))).real
    return h_min[: (len(h_linear) + 1) // 2]


def normalize_dc(h: np.ndarray, ratio: int, factor: float = 0.99) -> np.ndarray:
    dc = float(np.sum(h))
    if abs(dc) < 1e-12:
        raise ValueError("DC gain too close to zero")
    return h * (ratio * factor / dc)


def design(n_taps: int, ratio: int, family: str = "44k", phase: str = "min", passband_end: float = 20000.0,
           beta: float = 25.0) -> np.ndarray:
    input_rate, stop = MULTI_RATE[f"{family}_{ratio}x"]
    aligned = padded_taps(n_taps, ratio)
    if phase == "linear":
        h = design_linear(n_taps, ratio, input_rate, passband_end, stop, beta)
    else:
        # the reference feeds the *linear* prototype of aligned length into the
        # homomorphic transform, which halves it; then truncates/pads.
        h = minimum_phase_homomorphic(design_linear(n_taps, ratio, input_rate, passband_end, stop, beta))
    if len(h) > aligned:
        h = h[:aligned]
    elif len(h) < aligned:
        h = np.pad(h, (0, aligned - len(h)))
    return normalize_dc(h, ratio)


def validate(h: np.ndarray, ratio: int, input_rate: int, passband_end: float, stopband_start: float,
             target_stopband_db: float = 140.0) -> dict:
    """The reference validator's metrics (scripts/filters/generate_filter.py:373-417): 16384-point response on
    [0, fs_out/2); pass-band ripple = max - min dB up to passband_end; stop-band attenuation = |min dB| from
    stopband_start up (sic: the reference takes the MINIMUM of the stop-band, i.e. its deepest point); peak position,
    first/second-half energy ratio, minimum-phase and symmetry flags."""
    from scipy import signal

    h = np.asarray(h, dtype=np.float64)
    w, H = signal.freqz(h, worN=16384, fs=input_rate * ratio)
    H_db = 20 * np.log10(np.abs(H) + 1e-12)
    pb = w <= passband_end
    sb = w >= stopband_start
    peak_idx = int(np.argmax(np.abs(h)))
    mid = len(h) // 2
    energy_ratio = float(np.sum(h[:mid] ** 2) / (np.sum(h[mid:] ** 2) + 1e-12))
    peak_threshold = int(len(h) * 0.01)
    stop = float(np.min(H_db[sb]))
    peak = float(np.max(np.abs(H[pb]))) if pb.any() else 0.0
    return {
        "passband_ripple_db": float(np.max(H_db[pb]) - np.min(H_db[pb])),
        "input_band_peak": peak,
        "input_band_peak_normalized": peak / ratio if ratio else 0.0,
        "stopband_attenuation_db": abs(stop),
        "peak_position": peak_idx,
        "peak_threshold_samples": peak_threshold,
        "energy_ratio_first_to_second_half": energy_ratio,
        "meets_stopband_spec": abs(stop) >= target_stopband_db,
        "is_minimum_phase": bool(peak_idx < peak_threshold and energy_ratio > 10),
        "is_symmetric": bool(np.allclose(h, h[::-1], atol=1e-10)),
        "actual_taps": int(len(h)),
    }


def normalization_info(h_before: np.ndarray, h_after: np.ndarray, ratio: int, factor: float = 0.99) -> dict:
    """The reference's normalisation record (generate_filter.py:473-519)."""
    dc0 = float(np.sum(h_before))
    return {
        "original_dc_gain": dc0,
        "target_dc_gain": float(ratio),
        "dc_gain_factor": factor,
        "normalized_dc_gain": float(np.sum(h_after)),
        "applied_scale": float(ratio * factor / dc0),
        "l1_norm": float(np.sum(np.abs(h_after))),
        "l1_norm_ratio": float(np.sum(np.abs(h_after)) / ratio),
        "max_coefficient_amplitude": float(np.max(np.abs(h_after))),
        "normalization_applied": True,
    }


def export(h: np.ndarray, out_dir, name: str, ratio: int, extra: dict | None = None) -> Path:
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    h32 = np.asarray(h, dtype="<f4")
    h32.tofile(out / f"{name}.bin")
    fft, block = geometry(len(h32))
    meta = dict(extra or {})
    meta.update(coefficients_bin=f"{name}.bin", taps=int(len(h32)), fft_size=int(fft), block_size=int(block),
                upsample_factor=int(ratio))
    (out / f"{name}.json").write_text(json.dumps(meta, indent=2) + "\n")
    return out / f"{name}.json"


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--taps", type=int, default=80000)
    ap.add_argument("--ratio", type=int, default=2, choices=[2, 4, 8, 16])
    ap.add_argument("--family", default="44k", choices=["44k", "48k"])
    ap.add_argument("--phase", default="min", choices=["min", "linear"])
    ap.add_argument("--kaiser-beta", type=float, default=25.0)
    ap.add_argument("--out-dir", default="data/coefficients")
    a = ap.parse_args(argv)
    h = design(a.taps, a.ratio, a.family, a.phase, beta=a.kaiser_beta)
    input_rate, stop = MULTI_RATE[f"{a.family}_{a.ratio}x"]
    results = validate(h, a.ratio, input_rate, 20000, stop)
    meta = dict(generation_date=datetime.now().isoformat(), n_taps_specified=a.taps, n_taps_actual=int(len(h)),
                sample_rate_input=input_rate, sample_rate_output=input_rate * a.ratio, upsample_ratio=a.ratio,
                passband_end_hz=20000, stopband_start_hz=stop, target_stopband_attenuation_db=140,
                kaiser_beta=a.kaiser_beta, minimum_phase_method="homomorphic" if a.phase == "min" else None,
                target_dc_gain=float(a.ratio), output_basename=base_name(a.family, a.ratio, a.taps, a.phase),
                validation_results=results, dc_gain=float(np.sum(h.astype(np.float32))),
                generator="totton-rasp-gpu-dsp_amd/filter_design.py")
    p = export(h, a.out_dir, base_name(a.family, a.ratio, a.taps, a.phase), a.ratio, meta)
    print(p)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
