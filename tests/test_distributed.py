"""The N > 1 path of bench.py on CPU: two ranks over gloo (no GPU). The data path
has no collective by design (independent streams per GPU); what
must hold across ranks is the launch contract: env-based rendezvous on
127.0.0.1, a disjoint and complete stream partition, barriers around the timed
region, MAX over ranks of the elapsed time, ONE JSON line from rank 0 whose
`value` is the whole-job aggregate."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world: int, extra: list[str]):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--dry-run",
                                       "--steps", "5", "--warmup", "1", *extra], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    return [o for o, _ in outs]


@pytest.mark.parametrize("config,streams", [(2, 1), (4, 3)])
def test_two_ranks_gloo_contract(config, streams):
    outs = run_ranks(2, ["--config", str(config), "--streams", str(streams)])
    lines = [ln for ln in outs[0].splitlines() if ln.strip()]
    assert len(lines) == 1, outs[0]          # ONE json line, rank 0 only
    assert outs[1].strip() == ""
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["vs_baseline"] is None and r["dtype"] == "f32" and r["data"] == "synthetic"
    assert "cpu_baseline" not in r            # N = 1 only
    cfg = r["config"]
    assert cfg["streams_total"] == 2 * streams and "no collective" in cfg["parallelism"]
    # dry-run time is 10 ms * (rank + 1): the reduce must have taken rank 1's 20 ms
    assert abs(r["ms_per_step"] - 20.0 / 5) < 1e-6
    samples = cfg["blocks_per_channel"] * cfg["channels"] * streams * cfg["block_size"] * 2 * 5
    assert abs(r["value"] - samples / 0.020 / 1e6) <= 1e-3 * r["value"]
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    per_launch = cfg["blocks_per_channel"] * cfg["channels"] * streams * 4 * cfg["block_size"] * (1 + 1 / cfg["upsample_factor"])
    assert roof["algorithmic_bytes_per_launch"] == int(per_launch + 8 * (cfg["fft_size"] // 2 + 1))


def test_single_process_dry_run_and_flag_mismatch():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip())
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["config"]["workload"].startswith("configs[1]")
    # asking for 2 GPUs without the launcher is an error, not a silent N = 1 run
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "torch.distributed.run" in r.stderr


def test_stream_partition_and_seeds():
    sys.path.insert(0, str(ROOT))
    import bench

    owned = [bench.stream_ids(r, 32) for r in range(8)]
    flat = [i for part in owned for i in part]
    assert flat == list(range(256))           # BASELINE config 4: 256 streams over 8 GPUs, 32 each
    a = bench.synth_pcm(4, 0, 64, 2)
    b = bench.synth_pcm(4, 1, 64, 2)
    assert a.dtype.str == "<i4" and a.shape == (64, 2) and (a != b).any()
    assert (bench.synth_pcm(4, 0, 64, 2) == a).all()
