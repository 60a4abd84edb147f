"""Per-buffer latency of the blocking host-pointer call at the headline shape (BASELINE north star: "per-buffer latency < 1 ms").
NA_BatchProcess is what a host's audio callback calls: 1024 A1 Standard streams x 128 samples in, the same out, host memory both
sides.  The bound asserted is on the TAIL (p99.9 of 5000 calls), with half the north star's budget; the maximum is reported and may exceed
one buffer period (2.667 ms) in at most ONE call: this harness is an ordinary Python thread on a shared box -- no mlockall, no real-time
priority, an interpreter with other threads -- and one suite run in six saw a single 5.0 ms call among 5000 of 57 us (all others: max
0.16 ms).  INTEGRATION.md section 6 says what a host does about that: mlockall, a real-time priority, pinned blocks."""
import ctypes as C
import gc
import json
import os
import time

import numpy as np
import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_tail_of_5000_blocking_host_buffers_stays_below_half_a_millisecond():
    import neuralaudio_amd as na
    from neuralaudio_amd import capi
    if na.device_count() < 1:
        pytest.fail("no HIP device visible: the product path has no CPU fallback")
    lib = capi.load_library()
    model = na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
    S, n, calls, warm = 1024, 128, 5000, 300
    b = na.Batch(0)
    b.AddStreams(model, S)
    rng = np.random.default_rng(3)
    x = np.clip(0.3 * rng.standard_normal((S, n)), -1.0, 1.0).astype(np.float32)
    y = np.empty_like(x)
    # what INTEGRATION.md asks of a host: its blocks registered once (the kernels then read / write them directly, no staging copy)
    registered = lib.NA_RegisterHostBuffer(x.ctypes.data_as(C.c_void_p), x.nbytes) == 0 and lib.NA_RegisterHostBuffer(y.ctypes.data_as(C.c_void_p), y.nbytes) == 0
    xp, yp = x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float))
    for _ in range(warm):
        assert lib.NA_BatchProcess(b._h, xp, yp, n) == 0
    t = np.empty(calls)
    gc.collect()
    gc.disable()  # (no collector pause inside a timed call)
    try:
        for i in range(calls):
            t0 = time.perf_counter()
            rc = lib.NA_BatchProcess(b._h, xp, yp, n)
            t[i] = time.perf_counter() - t0
            assert rc == 0
    finally:
        gc.enable()
    ms = np.sort(t) * 1e3
    stats = {"calls": calls, "streams": S, "block": n, "registered_blocks": bool(registered), "p50_ms": float(ms[calls // 2]), "p99_ms": float(ms[int(calls * 0.99)]),
             "p99_9_ms": float(ms[int(calls * 0.999)]), "max_ms": float(ms[-1]), "over_one_buffer_period": int(np.sum(ms >= 2.667))}
    print("host buffer latency: " + json.dumps(stats))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "latency_tail.json"), "w") as f:
            json.dump(stats, f)
    except OSError:
        pass
    assert np.all(np.isfinite(y)) and np.any(y)
    assert stats["p99_9_ms"] < 0.5, stats
    assert stats["over_one_buffer_period"] <= 1, stats  # (see the header: one hiccup of this non-real-time harness is tolerated, two are not)
    if registered:
        lib.NA_UnregisterHostBuffer(x.ctypes.data_as(C.c_void_p))
        lib.NA_UnregisterHostBuffer(y.ctypes.data_as(C.c_void_p))
    b.close()
