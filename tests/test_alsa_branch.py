"""The HAVE_ALSA branch of the streamer (csrc/streamer_main.cpp: OpenAlsa, AlsaRead, AlsaWrite, the XRUN policy of the
reference's alsa_common.cpp:269-336) compiled and RUN on the CPU. Neither image carries alsa-lib, so the branch had
never been through a compiler (missing item 5): tests/cpp/alsa_stub/ holds a declaration stub of the slice
of <alsa/asoundlib.h> it uses and an in-memory fake device pair that misbehaves on purpose (short reads and writes,
-EAGAIN, XRUNs that need snd_pcm_recover, a capture that finally disappears). Test infrastructure only."""
from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PKG = ROOT / "totton-rasp-gpu-dsp_amd"
STUB = ROOT / "tests" / "cpp" / "alsa_stub"


@pytest.fixture(scope="module")
def alsa_streamer(tmp_path_factory):
    lib = PKG / "lib" / "libmi_upsampler.so"
    assert lib.exists(), "build() first"
    exe = tmp_path_factory.mktemp("alsa") / "alsa_streamer_with_alsa"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-Wno-unused-function", "-DHAVE_ALSA", f"-I{STUB}",
           str(PKG / "csrc" / "streamer_main.cpp"), str(STUB / "fake_alsa.cpp"), f"-L{lib.parent}", "-lmi_upsampler",
           f"-Wl,-rpath,{lib.parent}", "-pthread", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


@pytest.mark.parametrize("fmt,dtype,period", [("s16", "<i2", 256), ("s32", "<i4", 1000)])
def test_alsa_endpoints_pass_audio_through_an_awkward_device(alsa_streamer, tmp_path, fmt, dtype, period):
    """No filter requested (alsa_streamer_main.cpp:203-209: PCM passes through the float conversion): every complete
    period the fake capture delivers must reach the fake playback device, bit for bit, whatever mixture of short
    transfers, -EAGAIN and XRUN recoveries lies in between; the loop stops when the capture device goes away."""
    frames = period * 40 + 17   # the last 17 frames are less than a period: the reference stops at the short read
    rng = np.random.default_rng(period)
    info = np.iinfo(np.dtype(dtype))
    x = rng.integers(info.min, info.max, size=(frames, 2), dtype=np.int64).astype(dtype)
    cap, play, stats = tmp_path / "cap.raw", tmp_path / "play.raw", tmp_path / "stats.txt"
    cap.write_bytes(x.tobytes())
    env = dict(os.environ, FAKE_ALSA_CAPTURE=str(cap), FAKE_ALSA_PLAYBACK=str(play), FAKE_ALSA_STATS=str(stats))
    r = subprocess.run([str(alsa_streamer), "--in", "fake:capture", "--out", "fake:playback", "--rate", "48000", "--channels", "2",
                        "--format", fmt, "--period", str(period)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ALSA streaming started: input 48000 Hz" in r.stderr and "ALSA read failed" in r.stderr
    y = np.frombuffer(play.read_bytes(), dtype=dtype).reshape(-1, 2)
    assert y.shape[0] == period * 40
    if fmt == "s16":
        np.testing.assert_array_equal(y, x[:period * 40])            # 16 bits survive the float round trip exactly
    else:
        # s32 -> float32 -> s32 keeps 24 bits and clamps at the top (ConvertFloatToPcm): compare the float images
        want = np.clip(x[:period * 40].astype(np.float32) / np.float32(2**31), -1, np.float32(0.9999999))
        np.testing.assert_array_equal(y.astype(np.float32) / np.float32(2**31), (want * np.float32(2**31)).astype("<i4") / np.float32(2**31))
    s = stats.read_text()
    assert "capture" in s and "playback" in s
    for line in s.splitlines():
        fields = dict(kv.split("=") for kv in line.split()[1:])
        assert int(fields["recovered"]) > 0 and int(fields["waits"]) > 0   # XRUNs were recovered, -EAGAIN was waited out


def test_alsa_open_failure_is_reported(alsa_streamer, tmp_path):
    r = subprocess.run([str(alsa_streamer), "--in", "hw:9", "--out", "fake:playback", "--rate", "48000"], capture_output=True,
                       text=True, timeout=60, env=dict(os.environ, FAKE_ALSA_PLAYBACK=str(tmp_path / "p.raw")))
    assert r.returncode == 1 and "Failed to open ALSA device: hw:9" in r.stderr
