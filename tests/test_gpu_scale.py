"""Rehearsal of the N-GPU paths on the one GPU this box has (streams shard across the 8 GPUs with no data-path
collective; the driver's SCALE run launches `bench.py --gpus N` as one rank per GPU).  The first real 8-GPU run must not be the
first run of that code:
  * `bench.py --gpus 2 --share-device`: spawn_ranks -> torch.distributed.run with two ranks, both on device 0, rendezvous over gloo
    (RCCL refuses two ranks on one device) -- the rank-count check, the cost sharding of a mixed workload, barrier + max-over-ranks
    timing and the rank-0 JSON line are the N-GPU code, byte for byte;
  * `HostPipeBench --devices 0,0 --fan-in rccl --loopback`: the C++ multi-GPU host with two ranks through the library's loopback
    RCCL table (communicators, weight fan-out, gathered fan-in), bit-identical to one batch.
The numbers of such a run say nothing about scaling -- two ranks share one chip -- and the line says so (`share_device`)."""
import json
import os
import subprocess
import sys

import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHORT = ["--steps", "40", "--warmup", "10", "--ramp-ms", "20", "--no-cpu-baseline", "--no-host-path", "--rotate", "0", "--no-exact-f32"]


def _bench(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SHORT + extra, capture_output=True, text=True, timeout=100, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0
    return json.loads(lines[0])


@pytest.mark.watchdog(240)
def test_bench_with_two_ranks_on_one_device_runs_the_multi_rank_path():
    one = _bench(["--workload", "config3"])
    two = _bench(["--gpus", "2", "--share-device", "--workload", "config3"])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["share_device"] is True
    assert two["scaling"] == "weak" and two["steps"] == 40 and two["warmup"] == 10
    # the same line as N = 1 (the keys the driver reads), plus the sharding record
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in two and key in one, key
    assert two["metric"] == one["metric"] and two["unit"] == one["unit"] and two["dtype"] == one["dtype"]
    sh = two["config"]["shards"]
    assert sh is not None, sorted(two.keys())
    ranges = sh["ranges"]
    assert sh["global_streams"] == 2 * 4096 and ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 2 * 4096
    c0, c1 = sh["cost_per_rank"]
    assert abs(c0 - c1) / max(c0, c1) < 0.05  # cost-balanced: the rank with the Lite streams holds fewer of them
    assert ranges[0][1] != 4096               # ... so the cut is not the middle
    # whole-job value: all ranks' samples over the slowest rank's time
    assert abs(two["value"] - 2 * 4096 * 128 / (two["ms_per_step"] * 1e-3) / 1e6) / two["value"] < 1e-6
    assert two["parity_rms"] is not None and two["parity_rms"] < 1e-4 and two["output_finite"]


def test_bench_headline_workload_with_two_ranks_on_one_device():
    two = _bench(["--gpus", "2", "--share-device"])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["share_device"] is True
    assert "1024 batched streams per GPU" in two["config"]["workload"]
    assert abs(two["value"] - 2 * 1024 * 128 / (two["ms_per_step"] * 1e-3) / 1e6) / two["value"] < 1e-6
    assert two["parity_rms"] < 1e-4


def test_hostpipebench_two_ranks_over_the_loopback_rccl_table():
    exe = os.path.join(ROOT, "tools", "bin", "HostPipeBench")
    r = subprocess.run([exe, os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), "512", "128", "100", "--gpus", "2", "--devices", "0,0", "--fan-in", "rccl", "--loopback",
                        "--mix", os.path.join(O.MODELS_DIR, "BossLSTM-1x16.nam")], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["multi_gpu_host"] and j["fan_in"] == "rccl" and j["matches_single_batch"] and len(j["shards"]) == 2
    assert j["shards"][0]["end"] == j["shards"][1]["begin"] and j["shards"][1]["end"] == 512
