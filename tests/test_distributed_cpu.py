"""world_size-2 gloo test of the multi-GPU path's host logic (runs on CPU).

The N>1 path is: shard streams by cost -> every rank processes its own shard with no data-path collective
-> (optionally) all_gather the output shards; timing = barrier + MAX over ranks.  Here each rank's "GPU" is
the CPU oracle on a tiny model, so the sharded result can be compared with the unsharded one.
"""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch
    from neuralaudio_amd import dist as nd
    from neuralaudio_amd.sharding import shard_ranges
    import na_oracle as O

    rank, _, world = nd.env_rank()
    assert nd.init(backend="gloo")
    S, n = 7, 96
    arrays = O.a1_arrays(4, 2)
    w = O.synth_wavenet_weights(arrays, seed=1)
    x = np.stack([O.signal_noise(n, seed=100 + s) for s in range(S)])
    costs = [1.0, 3.0, 1.0, 1.0, 2.0, 1.0, 1.0]
    ranges = shard_ranges(costs, world)
    a, b = ranges[rank]
    local = np.stack([O.OracleWaveNet(arrays, w).process(x[s]) for s in range(a, b)]) if b > a else np.zeros((0, n), np.float32)
    nd.barrier()
    full = nd.gather_shards(torch.from_numpy(local), ranges).numpy()
    t = nd.max_over_ranks(1.0 + rank)
    assert t == float(world), t
    if rank == 0:
        want = np.stack([O.OracleWaveNet(arrays, w).process(x[s]) for s in range(S)])
        assert full.shape == want.shape and np.array_equal(full, want)
        print("OK", ranges)
    nd.shutdown()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "OK" in outs[0]
