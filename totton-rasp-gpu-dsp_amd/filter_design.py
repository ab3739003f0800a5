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
    """Cepstral folding (same maths as scipy.signal.minimum_phase(method=
    'homomorphic') with the reference's n_fft); output length (len+1)//2."""
    n_fft = 2 ** int(math.ceil(math.log2(len(h_linear) * 8)))
    spec = np.abs(np.fft.fft(h_linear, n_fft))
    spec = np.maximum(spec, 1e-7 * spec[spec > 0].min())
    ceps = np.fft.ifft(0.5 * np.log(spec)).real
    fold = np.zeros_like(ceps)
    fold[0] = ceps[0]
    fold[1 : n_fft // 2] = 2.0 * ceps[1 : n_fft // 2]
    fold[n_fft // 2] = ceps[n_fft // 2]
    h_min = np.fft.ifft(np.exp(np.fft.fft(fold))).real
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
    meta = dict(generation_date=datetime.now().isoformat(), n_taps_specified=a.taps, n_taps_actual=int(len(h)),
                sample_rate_input=input_rate, sample_rate_output=input_rate * a.ratio, upsample_ratio=a.ratio,
                passband_end_hz=20000, stopband_start_hz=stop, kaiser_beta=a.kaiser_beta,
                dc_gain=float(np.sum(h.astype(np.float32))), generator="totton-rasp-gpu-dsp_amd/filter_design.py")
    p = export(h, a.out_dir, base_name(a.family, a.ratio, a.taps, a.phase), a.ratio, meta)
    print(p)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
