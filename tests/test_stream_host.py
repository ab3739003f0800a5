"""CPU tests of the host-side pieces around the hot path: the SPSC PCM ring, the streaming loop (against
a Python restatement of the reference's loop, with a stand-in processor), rate negotiation (the cases of the
reference's tests/cpp/audio/test_auto_negotiation.cpp), config.json parsing, the multi-GPU partition."""
from __future__ import annotations

import ctypes as C
import json
import subprocess

import numpy as np
import pytest

from conftest import ROOT

FULL = dict(min=44100, max=768000, rates=[44100, 48000, 88200, 96000, 176400, 192000, 352800, 384000, 705600, 768000])
LIMITED = dict(min=44100, max=192000, rates=[44100, 48000, 88200, 96000, 176400, 192000])
RANGE_ONLY = dict(min=44100, max=768000, rates=[])


# ---- negotiation (reference cases restated: test_auto_negotiation.cpp:70-318) ---------------------------------
def test_rate_family_and_ratio(ups):
    for r in (44100, 88200, 176400, 352800, 705600):
        assert ups.rate_family(r) == 1
    for r in (48000, 96000, 192000, 384000, 768000):
        assert ups.rate_family(r) == 2
    assert ups.rate_family(22050) == 1 and ups.rate_family(11025) == 1 and ups.rate_family(32000) == 2
    assert ups.same_family(44100, 88200) and ups.same_family(48000, 192000)
    assert not ups.same_family(44100, 48000) and not ups.same_family(176400, 192000)
    for fin, fout, want in [(44100, 705600, 16), (88200, 705600, 8), (176400, 705600, 4), (352800, 705600, 2),
                            (48000, 768000, 16), (96000, 768000, 8), (192000, 768000, 4), (384000, 768000, 2),
                            (0, 705600, 0), (44100, 0, 0), (44100, 100000, 0)]:
        assert ups.upsample_ratio(fin, fout) == want


def test_negotiation_cases_of_the_reference(ups):
    n = ups.negotiate(44100, FULL)
    assert n["valid"] and n["family"] == 1 and (n["output_rate"], n["ratio"]) == (705600, 16)
    assert n["requires_reconfiguration"]                        # first time
    n = ups.negotiate(88200, FULL)
    assert n["valid"] and (n["output_rate"], n["ratio"]) == (705600, 8)
    n = ups.negotiate(48000, FULL)
    assert n["valid"] and n["family"] == 2 and (n["output_rate"], n["ratio"]) == (768000, 16)
    # same family keeps the output rate: no reconfiguration; the other family needs one
    assert not ups.negotiate(88200, FULL, 705600)["requires_reconfiguration"]
    n = ups.negotiate(48000, FULL, 705600)
    assert n["requires_reconfiguration"] and n["output_rate"] == 768000
    # DAC limited to 192 kHz: falls back inside the family
    assert (ups.negotiate(44100, LIMITED)["output_rate"], ups.negotiate(44100, LIMITED)["ratio"]) == (176400, 4)
    assert (ups.negotiate(48000, LIMITED)["output_rate"], ups.negotiate(48000, LIMITED)["ratio"]) == (192000, 4)
    # a DAC that only reports a range
    n = ups.negotiate(44100, RANGE_ONLY)
    assert n["valid"] and (n["output_rate"], n["ratio"]) == (705600, 16)
    # errors
    n = ups.negotiate(44100, None)
    assert not n["valid"] and n["error"].startswith("Invalid DAC capability")
    assert not ups.negotiate(0, FULL)["valid"] and not ups.negotiate(-1, FULL)["valid"]
    n = ups.negotiate(11025, FULL)                              # would need 64x
    assert not n["valid"] and "ratio 64 not in {1, 2, 4, 8, 16}" in n["error"]
    n = ups.negotiate(768000, LIMITED)
    assert not n["valid"] and "is less than input rate" in n["error"]


# ---- config.json ------------------------------------------------------------------------------------------------
def test_runtime_config_parsing(ups, tmp_path):
    ok, err, c = ups.parse_runtime_config(json.dumps({
        "eqEnabled": True, "eqProfile": "hd650 \u00e9", "eqProfilePath": '/data/EQ/a "b".txt',
        "alsa": {"inputDevice": "hw:1,0", "outputDevice": "hw:2,0", "sampleRate": 48000, "channels": 2,
                 "format": "S24_3LE", "periodFrames": 2048, "bufferFrames": 8192},
        "filter": {"ratio": 8, "phaseType": "linear", "directory": "/opt/x"}, "extra": [1, {"a": None}, "s"]}))
    assert ok, err
    assert c["eq_enabled"] and c["eq_profile_path"] == '/data/EQ/a "b".txt' and c["eq_profile"].startswith("hd650")
    assert (c["ratio"], c["phase_type"], c["filter_directory"]) == (8, "linear", "/opt/x")
    assert (c["sample_rate"], c["channels"], c["period_frames"], c["buffer_frames"], c["format"]) == (48000, 2, 2048, 8192, "S24_3LE")
    # the shipped example shape: nulls are "not set"
    ok, err, c = ups.parse_runtime_config('{"eqEnabled": false, "eqProfile": null, "eqProfilePath": null, "filter": {"ratio": 2}}')
    assert ok and not c["eq_enabled"] and c["eq_profile_path"] == "" and c["ratio"] == 2
    for bad in ("", "[1,2]", '{"eqEnabled": tru}', '{"a": 1,}', '{"a": "unterminated}', '{"a": 1} x'):
        ok, err, _ = ups.parse_runtime_config(bad)
        assert not ok and err


# ---- ring -------------------------------------------------------------------------------------------------------
def test_ring_two_thread_stress_under_tsan(tmp_path):
    exe = tmp_path / "test_pcm_ring"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
                        f"-I{ROOT / 'totton-rasp-gpu-dsp_amd' / 'csrc'}", str(ROOT / "tests" / "cpp" / "test_pcm_ring.cpp"),
                        "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stderr[-2000:]


def test_ring_c_abi(ups):
    r = ups.lib.mi_ring_create(16)
    buf = (C.c_ubyte * 16)(*range(16))
    out = (C.c_ubyte * 16)()
    assert ups.lib.mi_ring_write(r, buf, 10) == 1 and ups.lib.mi_ring_write(r, buf, 7) == 0
    assert ups.lib.mi_ring_available_to_read(r) == 10 and ups.lib.mi_ring_available_to_write(r) == 6
    assert ups.lib.mi_ring_read(r, out, 11) == 0 and ups.lib.mi_ring_read(r, out, 4) == 1 and list(out[:4]) == [0, 1, 2, 3]
    ups.lib.mi_ring_clear(r)
    assert ups.lib.mi_ring_available_to_read(r) == 0
    ups.lib.mi_ring_destroy(r)
    assert not ups.lib.mi_ring_create(0)


# ---- streaming loop -----------------------------------------------------------------------------------------------
def reference_loop_model(x: bytes, fb: int, period: int, bin_: int, bout: int, process, max_blocks: int = 1,
                         drain: bool = False, write_capacity=None):
    """Restatement of the reference's loop (alsa_streamer_main.cpp:495-611) on interleaved frames: returns the list of
    chunks written to the sink and the number of silence frames."""
    L = bout // bin_
    in_cap = max(bin_ * max_blocks, period) * 3 * fb
    out_cap = max(bout * max_blocks, period * L) * 3 * fb
    inq, outq, written, silence, pos = bytearray(), bytearray(), [], 0, 0
    accepted = 0

    def run_blocks(pad_tail=False):
        nonlocal inq, outq
        while True:
            avail = len(inq) // (bin_ * fb)
            room = (out_cap - len(outq)) // (bout * fb)
            tail = pad_tail and avail == 0 and len(inq) > 0
            k = min(1 if tail else avail, room, max_blocks)
            if k == 0:
                break
            if tail:
                blk = bytes(inq) + bytes(bin_ * fb - len(inq))
                inq = bytearray()
            else:
                blk = bytes(inq[:k * bin_ * fb])
                del inq[:k * bin_ * fb]
            outq += process(blk, k)

    while True:
        chunk = x[pos:pos + period * fb]
        pos += len(chunk)
        if len(chunk) < period * fb:
            if drain and chunk:
                inq += chunk
                accepted += len(chunk) // fb
            break
        accepted += period
        if len(inq) + len(chunk) > in_cap:
            inq = bytearray()
        else:
            inq += chunk
        run_blocks()
        wrote = False
        while len(outq) >= period * L * fb:
            written.append(bytes(outq[:period * L * fb]))
            del outq[:period * L * fb]
            wrote = True
        if not wrote:
            written.append(bytes(period * L * fb))
            silence += period * L
    if drain:
        real_out = accepted * L
        already = sum(len(w) for w in written) // fb - silence
        while True:
            run_blocks(pad_tail=True)
            moved = False
            while outq and already < real_out:
                n = min(len(outq) // fb, period * L, real_out - already)
                written.append(bytes(outq[:n * fb]))
                del outq[:n * fb]
                already += n
                moved = True
            if not moved or not inq:
                break
    return written, silence


def fake_processor(fb: int, bin_: int, bout: int):
    """Stand-in for the engine: every input frame repeated L times with the repeat index added to its first byte."""
    L = bout // bin_

    def proc(blk: bytes, blocks: int) -> bytes:
        a = np.frombuffer(blk, np.uint8).reshape(blocks * bin_, fb)
        o = np.repeat(a, L, axis=0).copy()
        o[:, 0] += np.tile(np.arange(L, dtype=np.uint8), blocks * bin_)
        return o.tobytes()

    return proc


@pytest.mark.parametrize("period,bin_,bout,max_blocks,drain,nframes", [
    (64, 200, 800, 1, False, 64 * 37),       # period < block, 4x: bursts and silence exactly as the reference's greedy drain
    (64, 200, 800, 1, True, 64 * 37 + 13),   # + end-of-input drain of the zero-padded tail
    (200, 200, 400, 1, False, 200 * 9),      # period == block
    (128, 48, 768, 4, True, 128 * 21 + 5),   # several blocks per period, batched calls, 16x
    (96, 96, 96, 2, True, 96 * 7),           # ratio 1
])
def test_stream_loop_matches_reference_model(ups, period, bin_, bout, max_blocks, drain, nframes):
    channels, fmt = 2, ups.PCM_S16
    fb = 4
    x = np.random.default_rng(period).integers(0, 255, nframes * fb, dtype=np.uint8).tobytes()
    proc = fake_processor(fb, bin_, bout)
    want, silence = reference_loop_model(x, fb, period, bin_, bout, proc, max_blocks, drain)
    got, pos, calls = [], [0], []

    def read(n):
        c = x[pos[0]:pos[0] + n * fb]
        pos[0] += len(c)
        return c

    def process(blk, blocks):
        calls.append(blocks)
        return proc(blk, blocks)

    ok, st = ups.stream_loop_run(dict(channels=channels, format=fmt, period_frames=period, block_in_frames=bin_,
                                      block_out_frames=bout, max_blocks_per_call=max_blocks, drain_at_end=drain),
                                 read, lambda b: got.append(b) or True, process)
    assert ok
    assert got == want
    assert st["silence_frames_written"] == silence and st["frames_written"] == sum(len(w) for w in want) // fb
    assert st["blocks_processed"] == sum(calls) and st["process_calls"] == len(calls) and max(calls) <= max_blocks
    assert st["input_overflows"] == 0 and st["output_overflows"] == 0
    if drain:  # everything that came in went out, in order, behind the start-up silence
        real = b"".join(w for w in want if any(w))
        full = proc(x + bytes((-nframes % bin_) * fb), (nframes + bin_ - 1) // bin_)
        assert len(real) <= nframes * (bout // bin_) * fb and full.startswith(real[:len(real)])


def test_stream_loop_passthrough_and_stop(ups):
    fb, period = 8, 32     # s32 stereo, no filter: PCM -> float -> PCM per period
    x = (np.random.default_rng(0).integers(-2**31, 2**31 - 1, 2 * period * 5 + 6, dtype=np.int64)).astype("<i4")
    raw = x.tobytes()
    got, pos = [], [0]

    def read(n):
        c = raw[pos[0]:pos[0] + n * fb]
        pos[0] += len(c)
        return c

    ok, st = ups.stream_loop_run(dict(channels=2, format=ups.PCM_S32, period_frames=period), read,
                                 lambda b: got.append(b) or True)
    assert ok and st["periods_read"] == 5 and st["frames_written"] == 5 * period
    want = ups.float_to_pcm(ups.pcm_to_float(np.frombuffer(raw[:5 * period * fb], np.uint8), ups.PCM_S32), ups.PCM_S32)
    assert b"".join(got) == want.tobytes()
    # a sink that refuses stops the loop with an error; a cleared running flag stops it cleanly
    pos[0] = 0
    ok, _ = ups.stream_loop_run(dict(channels=2, format=ups.PCM_S32, period_frames=period), read, lambda b: False)
    assert not ok
    pos[0] = 0
    flag = C.c_int(0)
    ok, st = ups.stream_loop_run(dict(channels=2, format=ups.PCM_S32, period_frames=period), read,
                                 lambda b: True, running=flag)
    assert ok and st["periods_read"] == 0


def test_stream_loop_overflow_drops_and_continues(ups):
    """A processor that cannot keep up is modelled by an output staging that never drains fast enough: with a block
    much larger than three periods' worth of room the input staging overflows, the loop logs the reference's message,
    clears the staging and carries on."""
    fb, period, bin_, bout = 2, 100, 100, 100
    logs, got, pos = [], [], [0]
    x = bytes(range(256)) * 40

    def read(n):
        c = x[pos[0]:pos[0] + n * fb]
        pos[0] += len(c)
        return c

    def process(blk, blocks):
        return None if len(got) >= 6 else blk   # after a while the "engine" reports a wrong size

    ok, st = ups.stream_loop_run(dict(channels=1, format=ups.PCM_S16, period_frames=period, block_in_frames=bin_,
                                      block_out_frames=bout), read, lambda b: got.append(b) or True, process,
                                 log=logs.append)
    assert not ok and logs == ["Filter output size mismatch"]   # alsa_streamer_main.cpp:545-549
    assert b"".join(got) == x[:len(b"".join(got))]


# ---- multi-GPU partition ------------------------------------------------------------------------------------------
def test_static_stream_partition(ups):
    assert ups.multi_partition(8, 8) == list(range(8))
    assert ups.multi_partition(256, 8) == [s % 8 for s in range(256)]     # BASELINE configs[3]: 64 streams per GPU
    assert ups.multi_partition(5, 2) == [0, 1, 0, 1, 0] and ups.multi_partition(3, 8) == [0, 1, 2]
    assert ups.multi_partition(0, 4) == []
    counts = np.bincount(ups.multi_partition(257, 8), minlength=8)
    assert counts.max() - counts.min() == 1


# ---- the reference's streamer e2e scenario (tests/cpp/test_alsa_streamer_e2e.cpp:52-140) on the built-in null endpoints ----
def test_cli_null_endpoints_start_and_stop_on_sigint():
    """alsa_streamer --in null --out null ... : started log, SIGINT after 200 ms, stopped log, exit code 0 -- the reference
    test's exact command line and expectations (no filter: PCM pass-through, no GPU involved)."""
    import signal
    import time

    binary = ROOT / "totton-rasp-gpu-dsp_amd" / "bin" / "alsa_streamer"
    assert binary.exists(), "build() must produce bin/alsa_streamer"
    p = subprocess.Popen([str(binary), "--in", "null", "--out", "null", "--rate", "44100", "--period", "128", "--buffer", "512",
                          "--channels", "2", "--format", "s32"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    time.sleep(0.2)
    p.send_signal(signal.SIGINT)
    out, _ = p.communicate(timeout=3)
    assert p.returncode == 0, out
    assert "ALSA streaming started" in out and "ALSA streaming stopped" in out
    periods = int(out.split("ALSA streaming stopped: ")[1].split(" periods")[0])
    assert 3 <= periods <= 150  # 200 ms of 128-frame periods at 44.1 kHz is 69: the null capture is paced, not a busy loop
