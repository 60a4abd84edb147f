"""The C++ multi-GPU host (csrc/multi_gpu.cpp, NA_Multi*): one batch + one host thread + one HIP stream per device, the global stream
list cut by cost.  On a one-GPU box the shards share the device (devices = [0, 0, ...]): that proves the per-device plumbing -- worker
threads, per-shard batches, row offsets, fan-in through the caller's arrays -- against a single batch holding the same global list."""
import json
import os
import subprocess

import numpy as np
import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def na():
    import neuralaudio_amd
    if neuralaudio_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the product path has no CPU fallback")
    return neuralaudio_amd


def _path(name):
    return os.path.join(O.MODELS_DIR, name)


@pytest.mark.parametrize("shards", [1, 2, 3])
def test_sharded_host_matches_one_batch_with_the_same_global_list(na, shards):
    loader = na.NeuralModelLoader()
    std = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    nano = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)
    lstm = loader.CreateFromFile(_path("BossLSTM-1x16.nam"), doPrewarm=False)
    a2 = loader.CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False)
    entries = [(std, 37, 1.0), (a2, 20, 0.2), (a2, 20, 1.0), (nano, 41, 1.0), (lstm, 55, 1.0)]  # architecture-sorted global list
    one = na.Batch(0)
    multi = na.MultiBatch([0] * shards)
    for m, c, q in entries:
        assert one.AddStreams(m, c, quality=q) == multi.AddStreams(m, c, quality=q)
    multi.Commit()
    ranges = multi.ShardRanges()
    S = one.NumStreams()
    assert len(ranges) == shards and ranges[0][0] == 0 and ranges[-1][1] == S and all(ranges[i][1] == ranges[i + 1][0] for i in range(shards - 1))
    if shards == 2:
        # cost-balanced, not count-balanced: the first shard holds the expensive WaveNets and therefore fewer streams
        assert (ranges[0][1] - ranges[0][0]) < (ranges[1][1] - ranges[1][0])
    rng = np.random.default_rng(2)
    for n in (128, 64, 100):
        x = (0.3 * rng.standard_normal((S, n))).clip(-1, 1).astype(np.float32)
        assert np.array_equal(multi.Process(x), one.Process(x))
    multi.SetQuality(40, 1.0)  # a stream of the second entry (A2, quality 0.2): global id -> (shard, local id)
    one.SetQuality(40, 1.0)
    x = (0.3 * rng.standard_normal((S, 128))).clip(-1, 1).astype(np.float32)
    ym = multi.Process(x)
    assert np.array_equal(ym, one.Process(x))
    assert np.all(np.isfinite(ym)) and O.rms(ym[0]) > 1e-3 and O.rms(ym[S - 1]) > 1e-4  # (the streams ran: not silence)
    multi.close()


def test_hostpipebench_multi_gpu_mode_two_threads_on_one_gpu():
    exe = os.path.join(ROOT, "tools", "bin", "HostPipeBench")
    r = subprocess.run([exe, _path("BossWN-standard.nam"), "512", "128", "200", "--devices", "0,0", "--mix", _path("BossLSTM-1x16.nam")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["multi_gpu_host"] and j["matches_single_batch"] and len(j["shards"]) == 2
    a, b = j["shards"]
    assert a["begin"] == 0 and a["end"] == b["begin"] and b["end"] == 512
    assert a["end"] - a["begin"] < 256  # Standard streams cost about twice an LSTM 1x16 stream: the cut lies inside the first half


def test_rccl_fan_in_with_one_rank_matches_the_host_row_path(na):
    """The RCCL mode of the multi-GPU host on what this box has: ONE rank (ncclCommInitAll over one device, the gather of one part, the
    single download from the gathered device buffer).  Bit-identical to the default host-row path; with two distinct GPUs the same code
    replicates the weights over xGMI and gathers both shards' rows (skipped below: one GPU)."""
    if not na.rccl_available():
        pytest.skip("no RCCL on this box")
    loader = na.NeuralModelLoader()
    std = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    lstm = loader.CreateFromFile(_path("BossLSTM-1x16.nam"), doPrewarm=False)
    ref, multi = na.MultiBatch([0]), na.MultiBatch([0])
    multi.SetFanIn("rccl")
    for mb in (ref, multi):
        mb.AddStreams(std, 9)
        mb.AddStreams(lstm, 7)
        mb.Commit()
    rng = np.random.default_rng(5)
    for n in (128, 128, 64):
        x = (0.3 * rng.standard_normal((16, n))).clip(-1, 1).astype(np.float32)
        assert np.array_equal(multi.Process(x), ref.Process(x))
    # one rank per DEVICE: two shards on one GPU cannot form a communicator
    dup = na.MultiBatch([0, 0])
    dup.SetFanIn("rccl")
    dup.AddStreams(std, 4)
    with pytest.raises(na.NeuralAudioError, match="distinct device"):
        dup.Commit()
    for mb in (ref, multi, dup):
        mb.close()


@pytest.fixture
def loopback(na):
    """The multi-GPU host bound to the in-library loopback table instead of librccl.so: ranks may share the one GPU of this box."""
    na.debug_set_rccl_api(1)
    yield na
    na.debug_set_rccl_api(0)


@pytest.mark.parametrize("shards", [2, 3])
def test_multi_rank_rccl_orchestration_runs_on_one_gpu_through_the_loopback_table(loopback, shards):
    """SURVEY 8(e) / VERDICT r04 item 6: the code paths that need MORE than one rank -- ncclCommInitAll over several ranks, the weight
    fan-out (only the first holder of a model uploads it; the others receive their weight images with ncclSend / ncclRecv and run
    their deferred prewarms), the all-gather of unequal parts (one ncclBroadcast per shard in a group) and the single download from
    shard 0 -- executed with 2 and 3 ranks on device 0.  Bit-identical to one batch holding the same global list: a receiver whose
    images had not arrived would run on uninitialised weights.  (No xGMI byte moves: numbers stay unmeasured.)"""
    na = loopback
    loader = na.NeuralModelLoader()
    std = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    nano = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)
    lstm = loader.CreateFromFile(_path("BossLSTM-1x16.nam"), doPrewarm=False)
    a2 = loader.CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False)
    # Standard is cut across the first shards and the LSTM across the last ones: both are replicated; A2 sits in the middle
    entries = [(std, 260, 1.0), (a2, 16, 0.2), (a2, 16, 1.0), (nano, 40, 1.0), (lstm, 120, 1.0)]
    one, multi = na.Batch(0), na.MultiBatch([0] * shards)
    multi.SetFanIn("rccl")
    for m, c, q in entries:
        assert one.AddStreams(m, c, quality=q) == multi.AddStreams(m, c, quality=q)
    multi.Commit()
    ranges = multi.ShardRanges()
    S = one.NumStreams()
    assert len(ranges) == shards and ranges[0][0] == 0 and ranges[-1][1] == S
    assert ranges[0][1] < 260, "the Standard entry must span two shards for the fan-out to have a receiver"
    rng = np.random.default_rng(11)
    for n in (128, 64, 128):
        x = (0.3 * rng.standard_normal((S, n))).clip(-1, 1).astype(np.float32)
        ym, yo = multi.Process(x), one.Process(x)
        assert np.array_equal(ym, yo)
        for s in range(shards):  # every rank holds the whole gathered array
            assert np.array_equal(multi.GatheredOutput(s, n), yo)
    assert O.rms(yo[0]) > 1e-3 and O.rms(yo[S - 1]) > 1e-4
    multi.close()


def test_a_failed_transfer_inside_a_group_tears_the_commit_down_instead_of_hanging(loopback):
    """a rank that fails between ncclGroupStart and ncclGroupEnd must still close its group, and the object must come back
    with an error rather than leave its peers blocked.  Fault injection: the first ncclSend of the weight fan-out fails; the receiving
    rank's rendezvous times out (0.3 s here; librccl would wait for ever, which is why the host does everything fallible before the
    group and compares the ranks' image lists first)."""
    na = loopback
    na.debug_set_rccl_api(1, fail_send_at=1, rendezvous_ms=300)
    loader = na.NeuralModelLoader()
    std = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    bad = na.MultiBatch([0, 0])
    bad.SetFanIn("rccl")
    bad.AddStreams(std, 32)
    with pytest.raises(na.NeuralAudioError):
        bad.Commit()
    bad.close()
    # the same process can go on: a fresh object on the healthy table commits and runs
    na.debug_set_rccl_api(1)
    good, one = na.MultiBatch([0, 0]), na.Batch(0)
    good.SetFanIn("rccl")
    good.AddStreams(std, 32)
    one.AddStreams(std, 32)
    x = (0.3 * np.random.default_rng(3).standard_normal((32, 128))).clip(-1, 1).astype(np.float32)
    assert np.array_equal(good.Process(x), one.Process(x))
    good.close()


def test_rccl_fan_in_across_two_gpus(na):
    if na.device_count() < 2 or not na.rccl_available():
        pytest.skip("needs two GPUs and RCCL")
    loader = na.NeuralModelLoader()
    std = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    nano = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)
    one, multi = na.Batch(0), na.MultiBatch([0, 1])
    multi.SetFanIn("rccl")
    for m, c in ((std, 40), (nano, 64)):
        assert one.AddStreams(m, c) == multi.AddStreams(m, c)
    multi.Commit()
    rng = np.random.default_rng(6)
    for _ in range(3):
        x = (0.3 * rng.standard_normal((104, 128))).clip(-1, 1).astype(np.float32)
        assert np.array_equal(multi.Process(x), one.Process(x))
    multi.close()
