"""The N > 1 path of bench.py on CPU: two ranks, no GPU. The data path has no
collective by design (independent streams -- or channel groups -- per GPU,
SURVEY §8e); what must hold across ranks is the launch contract: env-based
rendezvous (RANK / WORLD_SIZE / MASTER_*), a disjoint and complete partition,
barriers around the timed region, MAX over ranks of the elapsed time, ONE JSON
line from rank 0 whose `value` is the whole-job aggregate -- and no rank left
waiting when another one dies. The control plane is bench.Control (a
standard-library Unix-socket hub; torch is not imported by bench.py)."""
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


def run_ranks(world: int, extra: list[str], env_extra=None, check=True):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--dry-run",
                                       "--steps", "5", "--warmup", "1", *extra], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    if check:
        for p, (out, err) in zip(procs, outs):
            assert p.returncode == 0, err[-2000:]
        return [o for o, _ in outs]
    return [(p.returncode, out, err) for p, (out, err) in zip(procs, outs)]


def test_bench_does_not_import_torch():
    """One HIP runtime per process: the control plane is standard library (round 2's gloo control plane pulled in torch's
    own ROCm libraries beside the product library's)."""
    src = (ROOT / "bench.py").read_text()
    assert "import torch" not in src and "torch.distributed as" not in src


def test_a_dying_rank_takes_the_others_down_instead_of_hanging():
    import time

    t0 = time.monotonic()
    res = run_ranks(2, ["--config", "2"], env_extra={"BENCH_FAIL_RANK_FOR_TEST": "1"}, check=False)
    assert time.monotonic() - t0 < 60
    assert res[1][0] == 1 and "failure injected" in res[1][2]
    assert res[0][0] == 1 and "a peer closed its connection" in res[0][2]
    assert res[0][1].strip() == ""            # no JSON line from a failed job
    # launcher-free start: the parent ends the surviving rank and reports failure
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2"],
                       env=clean_env(BENCH_FAIL_RANK_FOR_TEST="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_channel_split_is_strong_scaling_over_one_stream():
    """BASELINE configs[4] "32-channel, 1->8 GPU sweep": --split channels gives every rank a contiguous channel group of
    the SAME stream; the job's value is the 32-channel rate."""
    outs = run_ranks(2, ["--config", "5", "--split", "channels"])
    r = json.loads(outs[0].strip())
    cfg = r["config"]
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert cfg["channels"] == 32 and cfg["channels_per_gpu"] == 16 and cfg["streams_total"] == 1
    assert "contiguous groups of 16" in cfg["parallelism"]
    samples = cfg["blocks_per_channel"] * 32 * cfg["block_size"] * 5   # the whole 32-channel stream, five steps
    assert abs(r["value"] - samples / 0.020 / 1e6) <= 1e-3 * r["value"]
    # a channel count that does not divide is refused in every rank before the rendezvous
    res = run_ranks(2, ["--config", "2", "--split", "channels", "--gpus", "2"], check=False)  # 2 channels / 2 ranks: fine
    assert all(rc == 0 for rc, _, _ in res)
    port_env = clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    bad = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--dry-run", "--config", "5", "--split",
                          "channels"], env=port_env, capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "multiple of --gpus" in bad.stderr


@pytest.mark.parametrize("config,streams", [(2, 1), (4, 3)])
def test_two_ranks_contract(config, streams):
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


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_single_process_dry_run():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run", "--steps", "2"], env=clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip())
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["config"]["workload"].startswith("configs[1]")
    assert d["config"]["blocks_per_channel"] == 256          # BASELINE.md section 4 states 256 blocks per channel


def test_plain_launch_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the script starts two fresh ranks itself, rank 0 prints the ONE line, the
    aggregate covers both ranks."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5"], env=clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["streams_total"] == 2
    assert abs(d["ms_per_step"] - 20.0 / 5) < 1e-6           # MAX over ranks: rank 1's synthetic 20 ms


def test_more_ranks_than_devices_fails_loudly_and_does_not_hang():
    """Ranks are never stacked on one GPU: with 1 visible device a 2-rank run stops in every rank before the
    rendezvous (exit 3, message), under the launcher-free start as well as under an external launcher."""
    env = clean_env(BENCH_VISIBLE_DEVICES_FOR_TEST="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 3 and "2 ranks requested but only 1 HIP device(s) are visible" in r.stderr
    assert r.stdout.strip() == ""
    env = clean_env(BENCH_VISIBLE_DEVICES_FOR_TEST="1", RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                    MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 3 and "never stacked" in r.stderr
    # --gpus that contradicts the launcher's WORLD_SIZE is an error, not a silent resize
    env = clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_cpu_baseline_legs_on_this_host():
    """The three CPU legs of the bench line, with a tiny budget: one core, all cores (one channel per process), and
    the config-1 fftconvolve block."""
    sys.path.insert(0, str(ROOT))
    import bench

    r = bench.cpu_baseline(ROOT / "data" / "coefficients" / "filter_44k_4x_80000_min_phase.json", 0.3)
    assert r["cores"] == 1 and r["value"] > 0 and r["kind"] in ("reference", "port")
    assert r["host_cores_available"] == bench.host_cores() >= 1
    if r["host_cores_available"] > 1:
        a = r["all_cores"]
        assert a["cores"] == r["host_cores_available"] and a["value"] > r["value"]
    f = r["fftconvolve_config1"]
    assert f["f32"]["Msamples_per_s"] > 0 and f["f64"]["ms_per_block"] > 0 and "8192-frame" in f["workload"]


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


@pytest.mark.gpu
def test_two_ranks_on_real_hardware_rehearsal():
    """The multi-rank bench path with real HIP work, rehearsed on ONE GPU (BENCH_STACK_RANKS_FOR_TEST puts both ranks on
    device 0; the numbers mean nothing): plain `--gpus 2` spawns its ranks, the ranks prime, step and reduce together over
    the socket control plane, exactly ONE HIP runtime is mapped in a rank, and the end-to-end rate is the sum over ranks.
    Then the same with --split channels on the 32-channel config."""
    env = dict(os.environ, BENCH_STACK_RANKS_FOR_TEST="1", BENCH_VISIBLE_DEVICES_FOR_TEST="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--prime-seconds", "0.1"], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["streams_total"] == 2 * d["config"]["streams_per_gpu"]
    assert len(d["end_to_end"]["per_rank"]) == 2 and d["end_to_end"]["value"] > 0
    assert len(d["hip_runtimes_mapped"]) == 1, d["hip_runtimes_mapped"]
    assert d["roofline"]["copy_ceiling_GBps"] > 1000 and d["output_check"]["worst_err_over_tol"] <= 1.0
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--prime-seconds", "0.1", "--config", "5", "--split", "channels", "--no-extras"], capture_output=True,
                       text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["channels_per_gpu"] == 16 and d["config"]["channels"] == 32
    assert d["output_check"]["worst_err_over_tol"] <= 1.0
