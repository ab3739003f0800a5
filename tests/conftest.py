from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def ups():
    import os

    import totton_rasp_gpu_dsp_amd as m

    # diagnostic: the native stack of a thread that calls abort() -- the HSA runtime does on a GPU memory fault, from a
    # thread Python's faulthandler does not know, and its message goes to the descriptor pytest has captured
    # (profiles/r03_r_multi_fault.txt). On a GPU box the record lands in gpurun_out/ unless MIUPS_ABORT_BACKTRACE says otherwise.
    out = ROOT / "gpurun_out"
    if not os.environ.get("MIUPS_ABORT_BACKTRACE") and m.device_count() > 0:
        try:
            out.mkdir(exist_ok=True)
            os.environ["MIUPS_ABORT_BACKTRACE"] = str(out / "abort_backtrace.txt")
        except OSError:
            pass
    if os.environ.get("MIUPS_ABORT_BACKTRACE"):
        m.lib.mi_debug_install_abort_backtrace()
    return m


@pytest.fixture(scope="session", autouse=True)
def host_copy_rule_held(request):
    """At the end of a session on a GPU box: no asynchronous host copy of the library broke the rule of DESIGN 4 (never two
    in flight on unpinned host ranges that may share a page). Counted by the library itself, process-wide."""
    yield
    if "totton_rasp_gpu_dsp_amd" in sys.modules:
        m = sys.modules["totton_rasp_gpu_dsp_amd"]
        if m.device_count() > 0:
            assert m.unsafe_host_copies() == 0, "a host copy broke the one-in-flight-per-unpinned-page rule"


@pytest.fixture(scope="session")
def O():
    import oracle

    return oracle


@pytest.fixture(scope="session")
def gpu(ups):
    if ups.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need a GPU (there is no CPU fallback to test)")
    return 0


@pytest.fixture()
def make_filter(tmp_path):
    """Write taps + sidecar; returns the json path."""

    def _make(taps, fft, block, factor=None, name="f", extra=None):
        taps = np.asarray(taps, dtype="<f4")
        taps.tofile(tmp_path / f"{name}.bin")
        meta = dict(coefficients_bin=f"{name}.bin", taps=int(taps.size), fft_size=int(fft), block_size=int(block))
        if factor is not None:
            meta["upsample_factor"] = int(factor)
        if extra:
            meta.update(extra)
        p = tmp_path / f"{name}.json"
        p.write_text(json.dumps(meta))
        return p

    return _make


def real_input(seed: int, n: int) -> np.ndarray:
    """Same recipe as tests/golden/make_golden.py (seeded 0.2*N(0,1) float32)."""
    return (np.random.default_rng(seed).standard_normal(n) * 0.2).astype(np.float32)


def rel_err(a, b) -> float:
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
