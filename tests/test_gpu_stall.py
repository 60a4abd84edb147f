"""Bounded waits (include/neuralaudio_amd.h "Bounded waits", csrc/gpu_batch.h): the reference's Process is called from a real-time
thread and cannot block (NeuralAudio/NeuralModel.h:127; README: one thread per model).  Here every processing entry point waits for
the device, so every such wait has a wall-clock limit; a device that does not answer inside it breaks the batch -- the call returns
an error and silence, later calls fail at once, destroying the batch returns.  NA_DebugStallDevice plays the wedged device: a
kernel that keeps the batch's streams busy for a given time, behind which the buffer's launches queue."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu

LIMIT_MS, STALL_MS = 100.0, 1200.0


@pytest.fixture(scope="module")
def na():
    import neuralaudio_amd
    if neuralaudio_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the product path has no CPU fallback")
    return neuralaudio_amd


@pytest.fixture(scope="module")
def std(na):
    return na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _expect_stall(na, call, what):
    t0 = time.monotonic()
    rc = call()
    dt = time.monotonic() - t0
    assert rc != 0, what + ": the call reported success on a device that does not answer"
    assert LIMIT_MS / 1000.0 * 0.8 <= dt < LIMIT_MS / 1000.0 + 0.4, (what, dt)
    from neuralaudio_amd import capi
    assert "did not answer within" in capi.last_error(), capi.last_error()
    return dt


def _let_the_device_come_back():
    import torch
    time.sleep(STALL_MS / 1000.0)
    torch.cuda.synchronize()


def test_a_blocking_host_buffer_gives_up_at_the_limit_returns_silence_and_breaks_the_batch(na, std):
    from neuralaudio_amd import capi
    lib = capi.load_library()
    b = na.Batch(0)
    S, n = 64, 128
    b.AddStreams(std, S)
    assert b.GetWaitLimitMs() == 2000.0  # (the default; NA_WAIT_LIMIT_MS unset)
    x = np.stack([O.signal_sine(n, start=977 * s) for s in range(S)]).astype(np.float32)
    y_ok = b.Process(x)  # a healthy buffer first
    assert np.any(y_ok)
    b.SetWaitLimitMs(LIMIT_MS)
    b.DebugStallDevice(STALL_MS)
    y = np.full_like(x, 7.0)
    _expect_stall(na, lambda: lib.NA_BatchProcess(b._h, _fp(x), _fp(y), n), "NA_BatchProcess")
    assert not np.any(y), "a failed buffer must be silence"
    assert b.IsBroken()
    # every later call fails at once, without waiting for anything
    y[:] = 7.0
    t0 = time.monotonic()
    assert lib.NA_BatchProcess(b._h, _fp(x), _fp(y), n) != 0
    assert lib.NA_BatchSynchronize(b._h) != 0
    assert lib.NA_BatchAddStreams(b._h, std._h, 1.0, 1, 1) < 0
    assert time.monotonic() - t0 < 0.05
    assert "broken" in capi.last_error()
    assert not np.any(y)
    # destroying the broken batch returns although the device is still busy
    t0 = time.monotonic()
    b.close()
    assert time.monotonic() - t0 < LIMIT_MS / 1000.0 + 0.4
    _let_the_device_come_back()
    # the process is fine: a new batch computes what the oracle computes
    b2 = na.Batch(0)
    b2.AddStreams(std, 2)
    y2 = b2.Process(x[:2])
    assert O.rms(y2[1] - O.oracle_from_file("BossWN-standard.nam").process(x[1])) < 1e-4
    b2.close()


@pytest.mark.parametrize("resident", [False, True], ids=["half-batch chains", "resident launch"])
def test_device_pointer_buffers_give_up_at_the_limit(na, std, resident):
    """Contract (b) of NA_BatchProcessDevice: the library schedules the buffer on its own streams (two free-running half-batch launches,
    or commands to the resident launch) and the caller waits with NA_BatchWaitOutputs / NA_BatchSynchronize -- polls of HIP events and
    of the resident launch's completion counter, which used to spin without a limit."""
    import torch
    from neuralaudio_amd import capi
    lib = capi.load_library()
    dev = torch.device("cuda", 0)
    S, n = 1024, 128
    b = na.Batch(0)
    b.AddStreams(std, S)
    if resident:
        b.SetResidentLaunch(True)
    x = torch.clamp(0.3 * torch.randn(S, n), -1.0, 1.0).to(dev)
    y = torch.zeros(S, n, device=dev)
    torch.cuda.synchronize(dev)
    for _ in range(3):
        b.ProcessDevice(x.data_ptr(), y.data_ptr(), n, n, n)
    b.WaitOutputs()
    if not any(os.environ.get(k) for k in ("NA_WN_KERNEL", "NA_WN_SPEC", "NA_HOST_HALVES", "NA_SP_T", "NA_SP_GEN")):
        assert b.UsesResidentLaunch() == resident and (resident or b.UsesHalfLaunches())
    b.SetWaitLimitMs(LIMIT_MS)
    b.DebugStallDevice(STALL_MS)
    for _ in range(2):
        assert lib.NA_BatchProcessDevice(b._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, n, n) == 0  # (posting does not wait)
    _expect_stall(na, lambda: lib.NA_BatchWaitOutputs(b._h), "NA_BatchWaitOutputs")
    assert b.IsBroken()
    t0 = time.monotonic()
    assert lib.NA_BatchProcessDevice(b._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, n, n) != 0
    assert lib.NA_BatchSynchronize(b._h) != 0
    assert lib.NA_BatchWaitOutputs(b._h) != 0
    assert time.monotonic() - t0 < 0.05
    t0 = time.monotonic()
    b.close()
    assert time.monotonic() - t0 < LIMIT_MS / 1000.0 + 0.4
    _let_the_device_come_back()


def test_the_pipelined_host_interface_gives_up_at_the_limit(na, std):
    from neuralaudio_amd import capi
    lib = capi.load_library()
    S, n = 64, 128
    b = na.Batch(0)
    b.AddStreams(std, S)
    x = np.stack([O.signal_sine(n, start=31 * s) for s in range(S)]).astype(np.float32)
    y = b.Collect(b.Submit(x))
    assert np.any(y)
    b.SetWaitLimitMs(LIMIT_MS)
    b.DebugStallDevice(STALL_MS)
    t = lib.NA_BatchSubmit(b._h, _fp(x), n)
    assert t >= 0
    out = np.full_like(x, 7.0)
    _expect_stall(na, lambda: lib.NA_BatchCollect(b._h, t, _fp(out)), "NA_BatchCollect")
    assert b.IsBroken()
    assert lib.NA_BatchSubmit(b._h, _fp(x), n) < 0
    b.close()
    _let_the_device_come_back()


def test_a_limit_of_zero_waits_as_long_as_it_takes(na, std):
    """<= 0 switches the limit off (the blocking HIP waits): a short stall is simply waited out and the buffer is right."""
    S, n = 4, 128
    b = na.Batch(0)
    b.AddStreams(std, S)
    b.SetWaitLimitMs(0.0)
    x = np.stack([O.signal_sine(n, start=5 * s) for s in range(S)]).astype(np.float32)
    b.DebugStallDevice(300.0)
    t0 = time.monotonic()
    y = b.Process(x)
    assert time.monotonic() - t0 >= 0.25
    assert not b.IsBroken()
    assert O.rms(y[2] - O.oracle_from_file("BossWN-standard.nam").process(x[2])) < 1e-4
    # ... and so does a limit longer than the stall
    b.SetWaitLimitMs(1500.0)
    b.DebugStallDevice(300.0)
    y2 = b.Process(x)
    assert not b.IsBroken() and np.any(y2)
    b.close()
