#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NeuralModel::Process hot path on MI355X.

Workload (BASELINE.json configs[1]): NAM A1 WaveNet 'Standard', 1024 batched streams per GPU,
128-sample buffers, FP32.  One "step" = one pass of the hot path over one buffer of every stream
(one WaveNetSplitKernel launch over 1024 streams x 128 samples), inputs already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU, launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; run
   WITHOUT that launcher, `bench.py --gpus N` spawns it itself.  Streams are independent, so ranks share nothing on the data
   path: weak scaling; RCCL carries the timing barrier / max-over-ranks and one all-reduce that counts the ranks.)

Prints ONE JSON line (rank 0).  PyTorch is plumbing only: device buffers, the stream/event used for
timing, and torch.distributed (RCCL) for the barrier + max-over-ranks.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

STREAMS_PER_GPU = 1024
BLOCK = 128
MODEL_FILE = os.path.join(ROOT, "tests", "golden", "models", "BossWN-standard.nam")
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 MFMA peak


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read())))
        except Exception:
            pass
    return max(1, n)


PORT_NOTE = ("kind 'port' = oracle/na_oracle.c, a scalar C restatement built -O3 -march=native: it has none of the reference's Eigen "
             "vectorisation / MULTIFRAME_8X8 conv tiling, so it understates the reference's CPU path")
SIMD_NOTE = ("kind 'port' = oracle/na_oracle_simd.c built -O3 -march=native: the oracle's WaveNet path with frames as the vector axis and "
             "8-frame x 8-channel register tiles (the idea of the reference's MULTIFRAME_8X8_CONVOLUTION, WaveNet.h:144-239, without Eigen), "
             "validated <= 1e-6 RMS against the scalar restatement (tests/test_oracle.py); `scalar` beside it is oracle/na_oracle.c")


def _timed_cpu_run(run, label, seconds_target):
    """run(blocks, threads) -> wall seconds of `threads` independent copies each processing `blocks` buffers of BLOCK zeros."""
    cores = usable_cores()
    # calibrate on a short all-core run, then size the timed run to ~seconds_target of wall time
    t1 = run(32, 1)
    single = 32 * BLOCK / t1
    tc = run(16, cores)
    per_thread = 16 * BLOCK / tc
    blocks = max(16, min(int(seconds_target * per_thread / BLOCK), 2000000))
    t = run(blocks, cores)
    total = cores * blocks * BLOCK
    return {
        "value": total / t / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%s: %d threads x %d buffers of %d zero samples each after prewarm (ModelTest protocol), %.1f s wall; "
                  "single-thread %.3f Msamples/s (%.1fx real-time); %s" % (label, cores, blocks, BLOCK, t, single / 1e6, single / 48000.0, PORT_NOTE),
    }


def cpu_baseline_synthetic_recurrent(kind, model_json, seconds_target):
    """config 4's seeded models (no such files ship with the reference): NAM LSTM 2x16 / keras GRU 1x16 through the oracle's bench loops."""
    import ctypes as C
    import numpy as np
    import na_oracle as O  # cpu_baseline leg only

    lib = O.load_native_lib()
    fp = C.POINTER(C.c_float)
    if kind == "lstm":
        w = np.ascontiguousarray(model_json["weights"], dtype=np.float32)
        nl, hid = int(model_json["config"]["num_layers"]), int(model_json["config"]["hidden_size"])
        return _timed_cpu_run(lambda blocks, threads: lib.na_oracle_lstm_bench(nl, hid, w.ctypes.data_as(fp), w.size, BLOCK, blocks, threads),
                              "synthetic NAM LSTM %dx%d" % (nl, hid), seconds_target)
    layers = model_json["layers"]
    nl, hid = len(layers) - 1, int(layers[0]["shape"][-1])
    keep = [[np.ascontiguousarray(np.array(layers[i]["weights"][k], dtype=np.float32).ravel()) for i in range(nl)] for k in range(3)]
    hw = np.ascontiguousarray(np.array(layers[-1]["weights"][0], dtype=np.float32).ravel())
    hb = float(layers[-1]["weights"][1][0])
    ptrs = [(fp * nl)(*[a.ctypes.data_as(fp) for a in keep[k]]) for k in range(3)]
    lib.na_oracle_gru_bench.restype = C.c_double
    lib.na_oracle_gru_bench.argtypes = [C.c_int, C.c_int, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp), fp, C.c_float, C.c_int, C.c_int, C.c_int]
    return _timed_cpu_run(lambda blocks, threads: lib.na_oracle_gru_bench(nl, hid, ptrs[0], ptrs[1], ptrs[2], hw.ctypes.data_as(fp), hb, BLOCK, blocks, threads),
                          "synthetic keras GRU %dx%d (RTNeural arithmetic: parity unpinned)" % (nl, hid), seconds_target)


def cpu_baseline(workload="standard", seconds_target=12.0):
    """The oracle ('port' of the reference's Internal CPU path) timed on this box's host cores,
    ModelTest protocol (blocks of zeros after prewarm, Utils/ModelTest/ModelTest.cpp:59-79)."""
    import ctypes as C
    import numpy as np
    import na_oracle as O  # cpu_baseline leg only

    lib = O.load_native_lib()
    files = {"standard": "BossWN-standard.nam", "feather": "BossWN-feather.nam", "nano": "BossWN-nano.nam", "a2full": "BossWN-a2.nam",
             "a2lite": "BossWN-a2.nam", "lstm1x16": "BossLSTM-1x16.nam", "lstm2x8": "BossLSTM-2x8.nam"}
    if workload == "lite":
        j = json.loads(synthetic_lite_nam())
        files = dict(files, lite="synthetic A1 Lite (12/6 channels, seeded weights)")
    elif workload not in files:
        raise ValueError("no CPU baseline for workload " + workload)
    else:
        j = O.load_json(files[workload])
    if j["architecture"] == "SlimmableContainer":
        j = j["config"]["submodels"][O.quality_to_submodel(j, 0.0 if workload == "a2lite" else 1.0)]["model"]
    w = np.ascontiguousarray(j["weights"], dtype=np.float32)
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    if j["architecture"] == "LSTM":
        nl, hid = int(j["config"]["num_layers"]), int(j["config"]["hidden_size"])
        lib.na_oracle_lstm_bench.restype = C.c_double
        lib.na_oracle_lstm_bench.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_int, C.c_int]

        def run(blocks, threads):
            return lib.na_oracle_lstm_bench(nl, hid, wp, w.size, BLOCK, blocks, threads)
    else:
        arrays = O.wavenet_arrays_from_nam(j)
        cfgs = O._cfgs(arrays)

        def run(blocks, threads):
            return lib.na_oracle_wavenet_bench(len(arrays), cfgs, wp, w.size, BLOCK, blocks, threads)
    scalar = _timed_cpu_run(run, files[workload], seconds_target if j["architecture"] == "LSTM" else seconds_target / 2)
    if j["architecture"] == "LSTM" or BLOCK % 8 != 0 or not hasattr(lib, "na_oracle_simd_wavenet_bench"):
        return scalar
    # WaveNets: the vectorised variant is the headline CPU figure (the fairer stand-in for the reference's Eigen path), the scalar port rides along
    def run_simd(blocks, threads):
        return lib.na_oracle_simd_wavenet_bench(len(arrays), cfgs, wp, w.size, BLOCK, blocks, threads)
    simd = _timed_cpu_run(run_simd, files[workload], seconds_target / 2)
    simd["sample"] = simd["sample"].replace(PORT_NOTE, SIMD_NOTE)
    simd["port_simd"] = {"value": simd["value"], "unit": simd["unit"], "cores": simd["cores"]}
    simd["port_scalar"] = {"value": scalar["value"], "unit": scalar["unit"], "cores": scalar["cores"], "sample": scalar["sample"]}
    return simd


def synthetic_lite_nam():
    """A1 'Lite' (channels 12 / head 6; dilation lists of InternalModel.h:12-17): no Lite file ships with the reference, so the
    weights are seeded U(-a, a) with a = 1/sqrt(fan_in), head scale 0.02."""
    import numpy as np
    rng = np.random.default_rng(126)
    d1, d2 = [1, 2, 4, 8, 16, 32, 64], [128, 256, 512, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    arrays = [dict(input_size=1, channels=12, head_size=6, dil=d1, head_bias=False), dict(input_size=12, channels=6, head_size=1, dil=d2, head_bias=True)]
    w = []

    def u(n, fan_in):
        a = 1.0 / np.sqrt(max(fan_in, 1))
        w.append(rng.uniform(-a, a, size=n))

    layers = []
    for a in arrays:
        c = a["channels"]
        u(c * a["input_size"], a["input_size"])
        for _ in a["dil"]:
            u(c * c * 3, c * 3); u(c, c * 3); u(c, 1); u(c * c, c); u(c, c)
        u(a["head_size"] * c, c)
        if a["head_bias"]:
            u(a["head_size"], c)
        layers.append({"input_size": a["input_size"], "condition_size": 1, "head_size": a["head_size"], "channels": c, "kernel_size": 3,
                       "dilations": a["dil"], "activation": "Tanh", "gated": False, "head_bias": a["head_bias"]})
    w.append(np.array([0.02]))
    return json.dumps({"version": "0.5.4", "architecture": "WaveNet", "metadata": {"loudness": -10.0},
                       "config": {"layers": layers, "head": None, "head_scale": 0.02},
                       "weights": [float(v) for v in np.concatenate(w).astype(np.float32)], "sample_rate": 48000})


def mixed_cpu_baseline(parts, seconds_target=12.0):
    """Stream-weighted CPU baseline of a mixed batch: equal stream counts per model -> harmonic mean of the per-model rates.
    `parts`: workload names, or ready-made (kind, model json) pairs for config 4's synthetic recurrent models."""
    res = [cpu_baseline(w, seconds_target / len(parts)) if isinstance(w, str) else cpu_baseline_synthetic_recurrent(w[0], w[1], seconds_target / len(parts))
           for w in parts]
    rate = len(res) / sum(1.0 / r["value"] for r in res)
    return {"value": rate, "unit": "Msamples/s", "cores": res[0]["cores"], "kind": "port",
            "sample": "harmonic mean over equal stream shares of: " + " | ".join(r["sample"] for r in res)}


def parity_spot_check(check_rows, x, y, issued, nbuf, mdir):
    """RMS difference between the kernel's output of the last issued step and the CPU oracle (checker only) for one stream per model."""
    import numpy as np
    import na_oracle as O  # checker only
    y_host = y.cpu().numpy()
    x_host = None
    worst, per_model = 0.0, []
    for row, (kind, spec), q in check_rows:
        if kind == "file":
            j = O.load_json(spec)
            recurrent = j.get("architecture") == "LSTM"
            ora = O.oracle_from_file(spec, quality=q, prewarm=False)
            label = spec
        elif kind == "wavenet_json":
            j = json.loads(spec)
            ora = O.OracleWaveNet(O.wavenet_arrays_from_nam(j), np.asarray(j["weights"], dtype=np.float32), prewarm=False)
            recurrent, label = False, "synthetic A1 Lite"
        elif kind == "lstm_json":
            j = json.loads(spec)
            ora = O.OracleLSTM.from_nam(int(j["config"]["num_layers"]), int(j["config"]["hidden_size"]), np.asarray(j["weights"], dtype=np.float32), prewarm=False)
            recurrent, label = True, "synthetic LSTM"
        else:
            ora = O.OracleGRU(json.loads(spec), prewarm=False)
            recurrent, label = True, "synthetic GRU"
        # a WaveNet's state IS its last receptive field of inputs (4092 frames A1, 6346 A2): replay that and a margin from zero history
        tail = min(issued, 600 if recurrent else (ora.receptive_field + BLOCK - 1) // BLOCK + 2)
        if x_host is None:
            x_host = x.cpu().numpy()
        xin = np.concatenate([x_host[k % nbuf][row] for k in range(issued - tail, issued)])
        ref = ora.process(xin)[-BLOCK:]
        err = float(np.sqrt(np.mean((y_host[row].astype(np.float64) - ref.astype(np.float64)) ** 2)))
        worst = max(worst, err)
        per_model.append({"model": label, "quality": q, "row": int(row), "rms": err, "replayed_buffers": int(tail), "output_rms": float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))})
    return {"rms": worst, "per_model": per_model}


def rotation_regime(na, model, local_rank, dev, n_rot, bytes_per_sample, steps=240, warmup=80, resident=False):
    """The metric's own regime: ONE batch of n_rot x 1024 A1 Standard streams, every step visits the whole
    n_rot x 249 MB of ring state once -- nothing survives in the 256 MB Infinity Cache from one step to the next.  Same timed-region
    rules as the headline (library HIP events around the steps, wall clock beside them)."""
    import torch
    S = STREAMS_PER_GPU * n_rot
    b = na.Batch(local_rank)
    b.AddStreams(model, S)
    if resident:
        b.SetResidentLaunch(True)
    g = torch.Generator(device="cpu").manual_seed(99)
    nbuf = 2
    x = torch.clamp(0.25 * torch.randn(nbuf, S, BLOCK, generator=g), -1.0, 1.0).to(dev)
    y = torch.empty(S, BLOCK, device=dev)
    torch.cuda.synchronize(dev)
    for i in range(warmup):
        b.ProcessDevice(x[i % nbuf].data_ptr(), y.data_ptr(), BLOCK, BLOCK, BLOCK)
    b.Synchronize()
    b.MarkTime(0)
    b.MarkTime(1)
    torch.cuda.synchronize(dev)
    b.MarkTime(0)
    t0 = time.perf_counter()
    for i in range(steps):
        b.ProcessDevice(x[i % nbuf].data_ptr(), y.data_ptr(), BLOCK, BLOCK, BLOCK)
    b.MarkTime(1)
    b.WaitMarks()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ms = b.ElapsedMs() / steps
    mode = "resident" if b.UsesResidentLaunch() else ("half-batch chains" if b.UsesHalfLaunches() else "ordered")
    b.Synchronize()
    finite = bool(torch.isfinite(y).all().item())
    state_mb = b.StateBytes() / 1e6
    b.close()
    alg = bytes_per_sample * S * BLOCK
    gbs = alg / (ms * 1e-3) / 1e9
    return {
        "what": "one batch of %d x 1024 = %d A1 Standard streams, 128-sample buffers: every step walks %.0f MB of stream state once "
                "(Infinity Cache: 256 MB)" % (n_rot, S, state_mb),
        "streams": S, "steps": steps, "warmup": warmup, "launch_mode": mode,
        "ms_per_step": ms, "ms_per_step_wall_clock": wall / steps * 1e3,
        "us_per_1024_step": ms * 1e3 / n_rot,
        "state_mb_total": state_mb,
        "achieved_gbs": gbs, "frac": gbs / HBM_PEAK_GBS,
        "msamples_per_s": S * BLOCK / (ms * 1e-3) / 1e6,
        # streams one GPU keeps up with in real time: samples per second / 48 000
        "realtime_streams_48k": S * BLOCK / (ms * 1e-3) / 48000.0,
        # the north star's per-buffer latency bound, in this regime: one buffer of ALL streams
        "latency_per_buffer_ms": ms,
        "output_finite": finite,
    }


def exact_f32_figure(streams):
    """The same step on the same box with the exact-f32 kernel (NA_WN_KERNEL=frame: v_mfma_f32_4x4x1_16b_f32, f32 values throughout):
    what strict f32 arithmetic costs on this hardware, one number beside `dtype`."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "300", "--warmup", "50", "--ramp-ms", "150", "--streams", str(streams),
           "--no-cpu-baseline", "--no-host-path", "--no-parity-check", "--rotate", "0", "--no-exact-f32"]
    env = dict(os.environ, NA_WN_KERNEL="frame")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        return {"kernel": j["roofline"]["kernel"], "dtype": j["dtype"], "ms_per_step": j["kernel_ms_avg"], "frac": j["roofline"]["frac"],
                "launch_mode": j.get("launch_mode"), "what": "bench.py under NA_WN_KERNEL=frame on the same box, 300 steps"}
    except Exception as e:
        return {"error": repr(e)}


def resident_launch_figure(streams):
    """The same step on the same box as commands to the opt-in RESIDENT launch (NA_BatchSetResidentLaunch): sustained
    step time and the latency of a lone buffer (host clock around post + wait).  Reported beside the default path, never as `value`."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "600", "--warmup", "100", "--ramp-ms", "200", "--streams", str(streams), "--resident",
           "--no-cpu-baseline", "--no-host-path", "--no-parity-check", "--rotate", "0", "--no-exact-f32"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        return {"launch_mode": j.get("launch_mode"), "ms_per_step": j["kernel_ms_avg"], "frac": j["roofline"]["frac"],
                "latency_per_buffer_ms": j.get("latency_per_buffer_ms"),
                "what": "bench.py --resident on the same box, 600 steps: one launch stays on the chip and walks the buffers (commands through a ring in "
                        "fine-grained device memory); latency = host clock around one NA_BatchProcessDevice + NA_BatchWaitOutputs"}
    except Exception as e:
        return {"error": repr(e)}


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...` (one rank per GPU)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and "--share-device" not in sys.argv:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    global BLOCK
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU (default: the BASELINE config)")
    ap.add_argument("--block", type=int, default=BLOCK, help="samples per buffer (default 128 = the BASELINE config; other sizes are exploration only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle spot check of the last timed step")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the host-buffer latency loop and tools/HostPipeBench after the timed region (profiling runs: their launches "
                         "would be averaged into the kernel statistics and counters)")
    ap.add_argument("--caller-stream", action="store_true",
                    help="create the batch on a stream of the caller (torch's) instead of its own: every launch is ordered on that stream, "
                         "so a step is ONE launch (the batch's own stream lets a step run as two free-running half-batch launches)")
    ap.add_argument("--rotate", type=int, default=8,
                    help="after the timed region (headline workload, one GPU): ONE batch of N x 1024 Standard streams -- N x 249 MB of stream "
                         "state, far outside the 256 MB Infinity Cache -- stepped the same way: the regime of the metric itself (10 k+ "
                         "concurrent real-time streams visit all of their state once per buffer).  Reported as `rotation`; "
                         "realtime_streams_48k is computed from it.  0: skip")
    ap.add_argument("--resident", action="store_true",
                    help="opt in to the resident launch (NA_BatchSetResidentLaunch) for the timed batch; default: free-running half-batch chains")
    ap.add_argument("--no-exact-f32", action="store_true",
                    help="skip the same-box run of the exact-f32 kernel (NA_WN_KERNEL=frame) that is reported beside `dtype`")
    ap.add_argument("--ramp-ms", type=float, default=400.0, help="untimed sustained load before warm-up so the shader clock reaches steady state")
    ap.add_argument("--share-device", action="store_true",
                    help="REHEARSAL of the multi-rank path on a box with one GPU: the N ranks of --gpus N all use device 0 and rendezvous over "
                         "gloo (RCCL refuses two ranks on one device).  Everything else is the N-GPU path: spawn_ranks, the rank-count check, "
                         "cost sharding of the mixed workloads, barrier + max-over-ranks, the rank-0 line.  The numbers say nothing about scaling "
                         "(the ranks share one chip) and the line says so: \"share_device\": true.")
    ap.add_argument("--workload", default="standard",
                    help="standard (default = the BASELINE metric's config) | lite | feather | nano | a2full | a2lite | lstm1x16 | lstm2x8 | "
                         "config3 (Lite+Feather+Nano, 4096 streams) | config4 (LSTM 2x16 + GRU) | config5 (A2 quality sweep, 2048 streams) "
                         "(other BASELINE configs, for DESIGN.md numbers; the driver uses the default)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    if args.workload == "mixed3":
        args.workload = "config3"

    import numpy as np
    import torch

    BLOCK = args.block
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.share_device:
        local_rank = 0  # every rank on device 0 (rehearsal of the multi-rank path on a one-GPU box)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    reduce_dev = "cpu" if args.share_device else dev  # where the timing all-reduce lives: gloo has no device tensors
    import neuralaudio_amd as na
    from neuralaudio_amd import dist as nd

    rccl_ranks = 1
    if distributed:
        nd.init(backend="gloo" if args.share_device else "nccl", device=dev)  # nccl == RCCL on ROCm; one rank per GPU
        import torch.distributed as tdist
        ones = torch.ones(1, device=reduce_dev)
        tdist.all_reduce(ones)  # every rank adds 1 over RCCL: proves N ranks on N GPUs are really in the job
        rccl_ranks = int(ones.item())
        if rccl_ranks != world or world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but RCCL sees %d ranks (WORLD_SIZE %d)" % (args.gpus, rccl_ranks, world))

    S = args.streams
    if args.workload == "config3" and S == STREAMS_PER_GPU:
        S = 4096  # BASELINE configs[2]
    if args.workload == "config5" and S == STREAMS_PER_GPU:
        S = 2048  # BASELINE configs[4]: 16384 streams over 8 GPUs
    loader = na.NeuralModelLoader()
    loader.SetDevice(local_rank)
    mdir = os.path.dirname(MODEL_FILE)
    files = {"standard": ["BossWN-standard.nam"], "feather": ["BossWN-feather.nam"], "nano": ["BossWN-nano.nam"],
             "a2full": ["BossWN-a2.nam"], "a2lite": ["BossWN-a2.nam"], "lstm1x16": ["BossLSTM-1x16.nam"], "lstm2x8": ["BossLSTM-2x8.nam"],
             "lite": [], "config3": ["BossWN-feather.nam", "BossWN-nano.nam"], "config4": [], "config5": ["BossWN-a2.nam"]}[args.workload]
    quality = 0.0 if args.workload == "a2lite" else 1.0
    models = [loader.CreateFromFile(os.path.join(mdir, f), doPrewarm=False) for f in files]
    # how the parity spot check builds the CPU oracle of each model: ("file", name) | ("wavenet_json", text) | ("lstm_json", text) | ("gru_json", text)
    oracle_specs = [("file", f) for f in files]
    synthetic_json = {}
    if args.workload in ("lite", "config3"):
        lite_text = synthetic_lite_nam()
        models.insert(0, loader.CreateFromString(lite_text, ".nam", doPrewarm=False))
        files = ["synthetic-A1-lite"] + files
        oracle_specs.insert(0, ("wavenet_json", lite_text))
    if args.workload == "config4":
        # BASELINE configs[3]: LSTM 2x16 + keras GRU (H=16), half/half; no such files ship with the reference -> seeded U(-a, a) weights
        rng = np.random.default_rng(4)
        H, a = 16, 0.25
        lw = np.concatenate([rng.uniform(-a, a, 4 * H * (1 + H) + 4 * H + 2 * H), rng.uniform(-a, a, 4 * H * 2 * H + 4 * H + 2 * H), rng.uniform(-a, a, H + 1)])
        lstm = json.dumps({"version": "0.5.4", "architecture": "LSTM", "config": {"input_size": 1, "hidden_size": H, "num_layers": 2},
                           "weights": [float(v) for v in lw]})
        gru = json.dumps({"in_shape": [None, None, 1], "layers": [
            {"type": "gru", "shape": [None, None, H], "weights": [rng.uniform(-a, a, (1, 3 * H)).tolist(), rng.uniform(-a, a, (H, 3 * H)).tolist(),
                                                                  rng.uniform(-a, a, (2, 3 * H)).tolist()]},
            {"type": "dense", "shape": [None, None, 1], "weights": [rng.uniform(-a, a, (H, 1)).tolist(), [0.0]]}]})
        models = [loader.CreateFromString(lstm, ".nam", doPrewarm=False), loader.CreateFromString(gru, ".json", doPrewarm=False)]
        oracle_specs = [("lstm_json", lstm), ("gru_json", gru)]
        synthetic_json = {"lstm": json.loads(lstm), "gru": json.loads(gru)}
    if any(m is None for m in models):
        raise SystemExit("could not load " + str(files))
    # The batch launches on its OWN streams (NA_BatchCreate with a null stream handle): one NA_BatchProcessDevice call = one step may then
    # run as two free-running launches of half the streams each.  The timed region is bracketed by the library's HIP events on EVERY
    # stream it launches on (NA_BatchMarkTime / NA_BatchElapsedMs: the longest span) -- torch.cuda.Event would only see torch's stream.
    # --caller-stream: the round 1-3 arrangement (torch's stream handed in, one launch per step, same events on that one stream).
    tstream = torch.cuda.Stream(device=dev)  # a real (non-null) HIP stream handle
    torch.cuda.set_stream(tstream)
    batch = na.Batch(local_rank, hip_stream=tstream.cuda_stream if args.caller_stream else None)
    if args.resident:
        batch.SetResidentLaunch(True)
    # The stream list of the workload as (model, quality, count) entries, architecture-sorted.  One GPU (or the headline workload):
    # every rank runs its own S streams (weak scaling).  Mixed workloads on several GPUs: the entries describe the GLOBAL list of
    # S x world streams, which is cut by cost into one contiguous range per rank -- NA_ShardByCost, the C++ host's partition
    # (csrc/multi_gpu.cpp) -- so a rank that gets the expensive models gets fewer streams; no rank sees the same mix.
    def entries_for(total):
        if args.workload == "config5":
            # quality sweep 0 -> 1 over the streams: half run the 3-channel submodel, half the 8-channel one
            qs = (0.0, 0.1, 0.25, 0.5, 0.6, 0.75, 0.9, 1.0)
            return [(models[0], q, total // 8 + (1 if k < total % 8 else 0)) for k, q in enumerate(qs)]
        return [(mdl, quality, total // len(models) + (1 if k < total % len(models) else 0)) for k, mdl in enumerate(models)]

    check_rows = []  # (row of this rank's batch, oracle spec, quality): the first stream of every entry this rank runs

    shard_info = None
    if distributed and len(entries_for(8)) > 1:
        from neuralaudio_amd import capi
        from neuralaudio_amd.sharding import shard_ranges
        lib = capi.load_library()
        entries = entries_for(S * world)
        costs = []
        for mdl, q, c in entries:
            costs += [float(lib.NA_ModelStreamCost(mdl._h, float(q)))] * c
        ranges = shard_ranges(costs, world)
        a, b = ranges[rank]
        first = 0
        for mdl, q, c in entries:
            lo, hi = max(first, a), min(first + c, b)
            if hi > lo:
                check_rows.append((batch.AddStreams(mdl, hi - lo, quality=q), oracle_specs[models.index(mdl)], q))
            first += c
        S_global = S * world
        S = b - a
        shard_info = {"global_streams": S_global, "ranges": [list(r) for r in ranges], "cost_per_rank": [round(sum(costs[r0:r1]), 1) for r0, r1 in ranges]}
    else:
        S_global = S * world
        for mdl, q, c in entries_for(S):
            check_rows.append((batch.AddStreams(mdl, c, quality=q), oracle_specs[models.index(mdl)], q))

    # synthetic 48 kHz buffers (bench-C of SURVEY 8d): clip(0.25*N(0,1), +-1), per-rank seed; a ring of 8 distinct buffers
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    nbuf = 8
    x = torch.clamp(0.25 * torch.randn(nbuf, S, BLOCK, generator=g), -1.0, 1.0).to(dev)
    y = torch.empty(S, BLOCK, device=dev)

    issued = [0]  # steps issued so far: step k of the process reads buffer k % nbuf, so every stream's whole input history is known

    def step(_i=None):
        batch.ProcessDevice(x[issued[0] % nbuf].data_ptr(), y.data_ptr(), BLOCK, BLOCK, BLOCK)
        issued[0] += 1

    # Clock ramp (untimed, before the W warm-up steps): a cold MI355X needs ~0.2 s of sustained load before the SMU raises the shader
    # clock to its steady state (measured: 68 us/step in the first 20 ms, 61 us/step after 0.2 s); a real-time audio server is
    # always in that steady state.  --ramp-ms 0 disables it.
    ramp_steps = 0
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for i in range(256):
            step(i)
        torch.cuda.synchronize(dev)
        ramp_steps += 256
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)

    # Average step duration from HIP events bracketing the K timed steps on every stream the batch launches on (an event per step would
    # put an extra timestamp packet between every two launches and stretch the very gaps it measures).  When a step is two free-running
    # half-batch launches (launches_per_step 2) the two chains overlap: rocprofv3's per-launch average is then the duration of ONE half
    # launch running beside the other chain (~ the step time), the step time is what the roofline is reported on.
    batch.MarkTime(0)  # (dry run: the marks' events are created on first use -- 90 us that do not belong to the K steps)
    batch.MarkTime(1)
    nd.barrier()
    torch.cuda.synchronize(dev)
    batch.MarkTime(0)  # on the idle streams: the span below starts here
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    batch.MarkTime(1)
    batch.WaitMarks()            # polls the closing marks of every launch stream, then ...
    torch.cuda.synchronize(dev)  # ... device-wide: nothing of the K steps is left anywhere
    if world > 1:
        nd.barrier()
        torch.cuda.synchronize(dev)
    elapsed = nd.max_over_ranks(time.perf_counter() - t0, device=reduce_dev)
    marks_ms = batch.ElapsedMs()
    batch.Synchronize()  # (outside the timed region: the batch's own bookkeeping of its chains)
    kernel_ms_avg = marks_ms / args.steps  # per STEP (one or two launches)
    launches_per_step = 2 if batch.UsesHalfLaunches() else 1
    # how a step reached the chip: one ordered launch | two free-running half-batch launches | a command to the resident launch (ONE
    # dispatch walks all K timed steps: rocprofv3 shows a single long dispatch per timed window, its duration / steps = the step time)
    launch_mode = "resident" if batch.UsesResidentLaunch() else ("half-batch chains" if batch.UsesHalfLaunches() else "ordered")

    # Parity spot check (outside the timed region, on what the timed region left behind): `y` holds the output of the LAST timed step.
    # The first stream of every model of the batch is replayed on the CPU oracle over the tail of its known input history -- one
    # receptive field and more for a WaveNet, whose state is exactly that; 600 buffers for a recurrent model, whose state forgets --
    # and the oracle's last buffer must be the kernel's: the launches that were timed did the work.
    parity = None
    if rank == 0 and not args.no_parity_check:
        parity = parity_spot_check(check_rows, x, y, issued[0], nbuf, mdir)

    # diagnostic only (outside the timed region): steps on their own, each bracketed and waited for
    nprobe = min(32, args.steps)
    kernel_ms = []
    if batch.UsesResidentLaunch():
        # the resident launch has no launch boundary to bracket (a mark would make it leave and come back): post one buffer, wait for its
        # output rows, host clock around both -- what a real-time caller with ONE buffer in flight sees, command post and completion included
        batch.WaitOutputs()
        for i in range(nprobe):
            t1 = time.perf_counter()
            step(i)
            batch.WaitOutputs()
            kernel_ms.append((time.perf_counter() - t1) * 1e3)
    else:
        for i in range(nprobe):
            batch.MarkTime(0)
            step(i)
            batch.MarkTime(1)
            kernel_ms.append(batch.ElapsedMs())
    batch.Synchronize()
    torch.cuda.synchronize(dev)
    kernel_ms.sort()

    finite = bool(torch.isfinite(y).all().item())

    if rank == 0:
        samples_per_step = S * BLOCK                      # this rank's launch (roofline bookkeeping)
        total_samples = S_global * BLOCK * args.steps     # whole job
        value = total_samples / elapsed / 1e6  # Msamples/s, whole job
        bytes_per_sample = batch.AlgorithmicBytesPerSample(BLOCK)
        flops_per_sample = 2.0 * batch.MacsPerSample()
        alg_bytes_per_launch = bytes_per_sample * samples_per_step
        wall_ms = elapsed / args.steps * 1e3                # the clock `value` is computed from (max over ranks)
        achieved_gbs = alg_bytes_per_launch / (kernel_ms_avg * 1e-3) / 1e9
        achieved_tflops = flops_per_sample * samples_per_step / (kernel_ms_avg * 1e-3) / 1e12
        # the roof of a workload is the one its algorithmic figures sit closer to -- max(bytes / HBM, flops / FP32).  The
        # WaveNets are HBM-bound (A1 Standard sits on the ridge: 1348.25 B and 26 512 FLOP per sample are both 0.1685 of their roofs; it
        # is reported on the HBM roof, the FLOP view rides along as roofline_mfma_f32), LSTM / GRU are FP32-bound (10-12 B per sample).
        frac_hbm, frac_fp32 = achieved_gbs / HBM_PEAK_GBS, achieved_tflops / FP32_MFMA_PEAK_TFLOPS
        on_fp32_roof = frac_fp32 > 1.01 * frac_hbm
        # HBM traffic per launch: PMC counters cannot be read from inside this process; the figure comes from the committed rocprofv3
        # --pmc passes of this same command (tools/pmc_passes.sh -> profiles/traffic_latest.json, one entry per workload), named in traffic_source
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and BLOCK == 128:
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                te = tj.get("workloads", {}).get(args.workload) if "workloads" in tj else (tj if args.workload == "standard" else None)
                if te and int(te.get("streams", S)) == S:
                    traffic, traffic_source = te.get("hbm_bytes_per_launch"), "profiles/traffic_latest.json: " + str(te.get("source"))
            except Exception:
                traffic = None
        out = {
            "metric": "concurrent 48 kHz real-time streams + Msamples/s/GPU, NAM WaveNet Standard",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            **({"share_device": True, "share_device_note": "rehearsal: %d ranks on ONE device over gloo -- exercises the multi-rank code path; "
                "not a scaling measurement" % world} if args.share_device else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # what the path computes in: f32 values and f32 accumulation; the WaveNet mat-muls of the f16-split kernels evaluate each f32
            # product as three f16 x f16 MFMA products (22 mantissa bits); the recurrent and frame kernels are plain f32
            "dtype": "f32 (f16x3-split MFMA, f32 accumulate)" if batch.StreamKernelName(0) in ("WaveNetSpecKernel", "WaveNetSplitKernel") else "f32",
            "data": "synthetic",
            "config": {
                "workload": ("NAM A1 WaveNet 'Standard' (BossWN-standard.nam weights), %d batched streams per GPU, "
                             "128-sample buffers, 48 kHz, inputs resident in HBM" % S) if args.workload == "standard" else
                            ("%s (%s, quality %.1f), %d batched streams per GPU, 128-sample buffers" % (args.workload, "+".join(files), quality, S)),
                "streams_per_gpu": S,
                "block": BLOCK,
                # address footprint of the streams' state: what decides whether the rings live in the 256 MB Infinity Cache between launches
                "state_mb_per_gpu": batch.StateBytes() / 1e6,
                "parallelism": "independent streams sharded across %d GPU(s), no data-path collective" % world,
                # mixed workloads on several GPUs: the global list cut by cost (NA_ShardByCost), [begin, end) per rank
                "shards": shard_info,
            },
            "realtime_streams_48k": value * 1e6 / 48000.0,
            "msamples_per_s_per_gpu": value / world,
            "clock_ramp": {"ms": args.ramp_ms, "untimed_steps": ramp_steps},
            "kernel_ms_avg": kernel_ms_avg,                       # per step, HIP events over the K timed steps on every launch stream
            "launches_per_step": launches_per_step,               # 2: the step ran as two free-running half-batch launches
            "launch_mode": launch_mode,
            "steps_per_dispatch": args.steps if launch_mode == "resident" else 1.0 / launches_per_step,
            "kernel_ms_median_isolated": kernel_ms[len(kernel_ms) // 2],  # a step on its own (bracketed and waited for)
            # the north star's "per-buffer latency < 1 ms": one buffer of all streams on its own, device pointers in, outputs valid
            "latency_per_buffer_ms": kernel_ms[len(kernel_ms) // 2],
            "frac_note": ("roofline.frac is from the library's HIP events around the K timed steps; roofline.frac_wall_clock from the clock `value` "
                          "is computed from (comparable with a driver's clock around the run; with few steps it carries the fixed cost of "
                          "opening and closing the timed window)"),
            "output_finite": finite,
            # oracle spot check of the last timed step (one stream per model; tolerance of the north star: 1e-4 RMS)
            "parity_rms": parity["rms"] if parity else None,
            "parity_check": parity,
            "roofline": {
                # "mfma" = the FLOP roof: FP32 vector peak == dense f32 MFMA peak on MI355X (157.3 TFLOP/s), one roof
                "bound": "mfma" if on_fp32_roof else "hbm",
                "achieved": achieved_tflops if on_fp32_roof else achieved_gbs,
                "peak": FP32_MFMA_PEAK_TFLOPS if on_fp32_roof else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if on_fp32_roof else "GB/s",
                # per STEP = one NA_BatchProcessDevice call over all streams: from the HIP-event average of the K timed steps (kernel_ms_avg) ...
                "frac": frac_fp32 if on_fp32_roof else frac_hbm,
                # ... and from the wall clock `value` is computed from (ms_per_step: launch gaps and the closing synchronisation included)
                "frac_wall_clock": (frac_fp32 if on_fp32_roof else frac_hbm) * kernel_ms_avg / wall_ms,
                "frac_hbm": frac_hbm,
                "frac_fp32": frac_fp32,
                "traffic": traffic,
                "traffic_source": traffic_source,
                # the same fraction on the bytes the counters saw (A2: the guard copy loads a block's leading 16 frames once for all taps of
                # a small-dilation layer where the formula counts every tap's history, so measured < algorithmic; padded narrow models: >)
                "frac_measured_bytes": (traffic / (kernel_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_bytes_per_sample": bytes_per_sample,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,  # per step (all launches of one NA_BatchProcessDevice call together)
                "launches_per_step": launches_per_step,
                "algorithmic_flops_per_sample": flops_per_sample,
                # the kernel that runs stream 0 of the batch (the dominant one of every workload here: the first group is the largest /
                # the only WaveNet one; FamilyFor(), PackFor(), PadFor() in gpu_batch.cpp decide per model)
                "kernel": batch.StreamKernelName(0),
                "stream_pack_factor": batch.StreamPackFactor(0),  # > 1: narrow model, several real streams per kernel-level stream
            },
            "roofline_mfma_f32": {
                "achieved": achieved_tflops,
                "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / FP32_MFMA_PEAK_TFLOPS,
                "algorithmic_flops_per_sample": flops_per_sample,
            },
        }
        if world == 1 and args.workload == "standard" and args.rotate > 0 and BLOCK == 128:
            try:
                out["rotation"] = rotation_regime(na, models[0], local_rank, dev, args.rotate, bytes_per_sample, resident=args.resident)
                # the stream count the metric asks for comes from THIS regime: all state visited once per buffer, none of it cache-resident
                out["realtime_streams_48k_cache_resident"] = out["realtime_streams_48k"]
                out["realtime_streams_48k"] = out["rotation"]["realtime_streams_48k"]
            except Exception as e:
                out["rotation"] = {"error": repr(e)}
        if world == 1 and args.workload == "standard" and not args.no_exact_f32 and BLOCK == 128:
            out["exact_f32"] = exact_f32_figure(S)
            if not args.resident and not args.caller_stream:
                out["resident_launch"] = resident_launch_figure(S)
        if world == 1 and not args.no_host_path:
            # per-buffer latency through the host-buffer entry point (pinned staging, H2D, kernel, D2H, stream sync) -- the path a
            # real-time host calls once per audio buffer; outside the timed region, reported next to the north star's "< 1 ms per buffer"
            xh = x[0].cpu().numpy()
            lat = []
            for i in range(300):
                t_a = time.perf_counter()
                batch.Process(xh)
                lat.append((time.perf_counter() - t_a) * 1e3)
            lat = sorted(lat[50:])
            out["host_buffer_latency_ms"] = {"p50": lat[len(lat) // 2], "p99": lat[int(len(lat) * 0.99)], "max": lat[-1], "calls": len(lat),
                                             "what": "NA_BatchProcess, %d streams x %d samples, host pointers" % (S, BLOCK)}
            # PCIe-inclusive throughput (SURVEY 8d metric 1): host buffers in, host buffers out, through the pipelined entry points
            # (NA_BatchSubmit / NA_BatchCollect: upload of buffer k+1 and download of k-1 overlap the kernels of k), driven by a plain C++
            # host (tools/HostPipeBench: a Python loop around 48 us kernels measures the interpreter).  Never `value`.
            hp = os.path.join(ROOT, "tools", "bin", "HostPipeBench")
            model_path = os.path.join(mdir, files[0]) if args.workload in ("standard", "feather", "nano", "a2full", "lstm1x16", "lstm2x8") else None
            if os.path.exists(hp) and model_path is not None:
                torch.cuda.synchronize(dev)
                r = subprocess.run([hp, model_path, str(S), str(BLOCK), "2000"], capture_output=True, text=True, timeout=300)
                if r.returncode == 0:
                    hj = json.loads(r.stdout.strip().splitlines()[-1])
                    out["pcie_inclusive"] = {"ms_per_buffer": hj["us_per_buffer_zero_copy"] * 1e-3,
                                             "Msamples/s": S * BLOCK / hj["us_per_buffer_zero_copy"],
                                             "what": "tools/HostPipeBench (C++ host over the C ABI, its own batch on the same GPU): pinned host buffers in / out "
                                                     "(NA_BatchNextInput + NA_BatchSubmit / NA_BatchCollect + NA_BatchOutputView), 2 buffers in flight, "
                                                     "%d streams x %d samples: the kernels read and write the pinned host blocks themselves (NA_HOST_DIRECT=0: copy engines)" % (S, BLOCK),
                                             "with_host_copies_ms_per_buffer": hj["us_per_buffer_copying"] * 1e-3,
                                             "blocking_call_latency_us": hj["blocking_latency_us"],
                                             "blocking_call_on_registered_blocks_us": hj.get("registered_blocking_latency_us"),
                                             "submit_collect_in_place_us": hj.get("in_place_latency_us")}
                else:
                    out["pcie_inclusive"] = {"ms_per_buffer": None, "what": "HostPipeBench failed: " + r.stderr.strip()[-300:]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                if args.workload == "config3":
                    out["cpu_baseline"] = mixed_cpu_baseline(["lite", "feather", "nano"])
                elif args.workload == "config4":
                    out["cpu_baseline"] = mixed_cpu_baseline([("lstm", synthetic_json["lstm"]), ("gru", synthetic_json["gru"])])
                elif args.workload == "config5":
                    out["cpu_baseline"] = mixed_cpu_baseline(["a2lite", "a2full"])  # the quality sweep: half the streams on each submodel
                else:
                    out["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))

    batch.close()
    nd.shutdown()


if __name__ == "__main__":
    main()
