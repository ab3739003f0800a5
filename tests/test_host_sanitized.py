"""Host side of the product under AddressSanitizer + UndefinedBehaviorSanitizer (VERDICT r2 weak item 4,
ADVICE r2): csrc/host/*.cpp + csrc/capi_host.cpp compiled with -fsanitize=address,undefined into
tests/cpp/test_host_sanitized.cpp and fed, through the C ABI, the shipped sidecars / APO profiles / config.json / OPRA
records and several hundred damaged versions of each: truncated, bytes flipped, tokens swapped, brackets nested thousands
deep. Every reader must answer with a return code; a sanitizer report or a crash fails the test."""
from __future__ import annotations

import json
import random
import subprocess

import pytest

from conftest import GOLDEN, ROOT

CSRC = ROOT / "totton-rasp-gpu-dsp_amd" / "csrc"
HOST_SRCS = sorted((CSRC / "host").glob("*.cpp")) + [CSRC / "capi_host.cpp"]
CONFIG_EXAMPLE = json.dumps({
    "eqEnabled": True, "eqProfile": "hd650", "eqProfilePath": "/tmp/x.txt",
    "filter": {"ratio": 8, "phaseType": "minimum", "directory": "data/coefficients"},
    "alsa": {"inputDevice": "hw:0", "outputDevice": "hw:1", "sampleRate": 48000, "channels": 2, "format": "S32_LE",
             "periodFrames": 1024, "bufferFrames": 4096}}, indent=2)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("san") / "test_host_sanitized"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-ffp-contract=off", "-pthread", f"-I{ROOT / 'include'}", f"-I{CSRC}",
           "-DMIUPS_HOST_EMU",   # device/common.h without <hip/hip_runtime.h>: the host sources only use its plain structs
           str(ROOT / "tests" / "cpp" / "test_host_sanitized.cpp"), *map(str, HOST_SRCS), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


def damaged(text: bytes, rng: random.Random, n: int):
    """n damaged versions of one input: truncations, byte flips, duplicated / deleted slices, bracket storms."""
    out = []
    for k in range(n):
        b = bytearray(text)
        kind = k % 5
        if kind == 0 and len(b) > 1:
            b = b[: rng.randrange(1, len(b))]
        elif kind == 1:
            for _ in range(1 + rng.randrange(4)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        elif kind == 2 and len(b) > 8:
            i, j = sorted(rng.sample(range(len(b)), 2))
            b = b[:i] + b[j:]
        elif kind == 3 and len(b) > 8:
            i, j = sorted(rng.sample(range(len(b)), 2))
            b = b[:j] + b[i:j] * 3 + b[j:]
        else:
            i = rng.randrange(len(b))
            b = b[:i] + rng.choice([b"[", b"{", b'{"a":', b"\\u12", b'"', b"-", b"1e999", b"\x00"]) * rng.choice([1, 7, 300]) + b[i:]
        out.append(bytes(b))
    return out


def test_host_readers_under_asan_ubsan(exe, tmp_path):
    rng = random.Random(20251005)
    corpus = tmp_path / "corpus"
    corpus.mkdir()
    n = 0

    def put(kind: str, data: bytes, ext="txt"):
        nonlocal n
        (corpus / f"{kind}_{n}.{ext}").write_bytes(data)
        n += 1

    # config.json: the reference's example shape, valid and damaged, plus nesting bombs
    put("config", CONFIG_EXAMPLE.encode())
    for d in damaged(CONFIG_EXAMPLE.encode(), rng, 150):
        put("config", d)
    for depth in (65, 1000, 200000):
        put("config", b'{"a":' * depth + b"1" + b"}" * depth)
        put("config", b'{"a":' + b"[" * depth)
    # OPRA records (the golden inputs of the converter) and damaged ones; a deep one (its reader caps the depth too)
    opra = json.loads((GOLDEN / "g8_opra.json").read_text())
    records = [json.dumps(c["record"]).encode() for c in opra["cases"]] if "cases" in opra else []
    assert records, "golden OPRA records expected"
    for rec in records:
        put("opra", rec)
        for d in damaged(rec, rng, 25):
            put("opra", d)
    put("opra", b'{"parameters":' + b'{"bands":[' * 50000)
    # APO profiles
    for text in json.loads((GOLDEN / "g4_eq_profiles.json").read_text()).values():
        put("apo", text.encode())
        for d in damaged(text.encode(), rng, 40):
            put("apo", d)
    put("apo", ("Filter 1: ON PK Fc 1e309 Hz Gain nan dB Q -0\n" * 300).encode())
    # sidecars next to a real coefficient file: valid, damaged, and lying about their sizes
    side = (ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json").read_bytes()
    (corpus / "filter_44k_4x_80000_min_phase.bin").write_bytes(
        (ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.bin").read_bytes())
    put("sidecar", side, "json")
    for d in damaged(side, rng, 60):
        put("sidecar", d, "json")
    for bad in ('{"coefficients_bin": "filter_44k_4x_80000_min_phase.bin", "taps": 18446744073709551615, "fft_size": 131072, "block_size": 51072}',
                '{"coefficients_bin": "../../../../etc/passwd", "taps": 5, "fft_size": 16, "block_size": 12}',
                '{"coefficients_bin": "", "taps": 80001, "fft_size": 131072, "block_size": 51072, "upsample_factor": 0}'):
        put("sidecar", bad.encode(), "json")
    r = subprocess.run([str(exe), str(corpus)], capture_output=True, text=True, timeout=900,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-6000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    inputs, accepted = (int(t) for t in (r.stdout.split()[0], r.stdout.split()[2]))
    assert inputs == n and 5 <= accepted < inputs    # the intact inputs are accepted, most damaged ones refused
