"""The resident launch (csrc/gpu_batch_chains.cpp, wavenet_spec_impl.h WaveNetSpecResidentKernel; opt-in: NA_BatchSetResidentLaunch): for a
large A1 Standard batch on the batch's own stream NA_BatchProcessDevice posts a command to ONE launch that stays on the chip and walks
consecutive buffers.  Reference
arithmetic: WaveNetModelT::Process (NeuralAudio/WaveNet.h:768-799) per stream and buffer -- the launch mechanics must not change a bit
of it: every test compares with ordered one-shot launches of the same chain (a batch on the caller's stream)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_KNOBS = ("NA_WN_KERNEL", "NA_WN_SPEC", "NA_WN_PACK", "NA_HOST_HALVES", "NA_HOST_DIRECT", "NA_SP_T", "NA_SP_GEN", "NA_RESIDENT")


@pytest.fixture(scope="module")
def na():
    import neuralaudio_amd
    if neuralaudio_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the product path has no CPU fallback")
    return neuralaudio_amd


@pytest.fixture(scope="module")
def std(na):
    return na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)


def _resident_expected(batch):
    if any(os.environ.get(k) for k in _KNOBS):
        return True  # (forced-family runs: whatever path runs, the results must agree)
    return batch.UsesResidentLaunch()


def _pair(na, std, S):
    import torch
    ts = torch.cuda.Stream(device=torch.device("cuda", 0))
    ref, b = na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0)
    ref.AddStreams(std, S)
    b.AddStreams(std, S)
    b.SetResidentLaunch(True)  # (opt-in: the default for such a batch is two free-running half-batch launches)
    return ts, ref, b


def test_resident_steps_are_bit_identical_to_ordered_launches_and_match_the_oracle(na, std):
    """40 steps queued back to back (the throughput regime: the launch never idles), every step with its own input and output rows;
    then the oracle on two streams over the whole history."""
    import torch
    S, n, steps = 1024, 128, 40
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(21)
    x = torch.clamp(0.3 * torch.randn(steps, S, n, generator=g), -1.0, 1.0).to(dev)
    ts, ref, b = _pair(na, std, S)
    want, got = torch.empty(steps, S, n, device=dev), torch.zeros(steps, S, n, device=dev)
    torch.cuda.synchronize(dev)
    for k in range(steps):
        ref.ProcessDevice(x[k].data_ptr(), want[k].data_ptr(), n)
    ref.Synchronize()
    for k in range(steps):
        b.ProcessDevice(x[k].data_ptr(), got[k].data_ptr(), n)
        assert _resident_expected(b)
    b.WaitOutputs()
    assert torch.equal(got, want)
    b.Synchronize()
    xs = x.cpu().numpy()
    for s in (0, S - 1):
        yo = O.oracle_from_file("BossWN-standard.nam").process(np.concatenate([xs[k][s] for k in range(steps)]))
        assert O.rms(np.concatenate([got[k][s].cpu().numpy() for k in range(steps)]) - yo) < 2e-6
    ref.close()
    b.close()


def test_a_producer_kernel_on_a_foreign_stream_rewrites_the_same_input_rows_before_every_step(na, std):
    """VERDICT r04 item 7, the own-stream contract of NA_BatchProcessDevice (include/neuralaudio_amd.h, contract (b)): the input rows
    are produced by a kernel on a stream the batch has never seen (torch's), INTO THE SAME BUFFER every step, and synchronised before the
    call; the outputs are read back after NA_BatchWaitOutputs by a kernel on that foreign stream -- while the resident launch is still
    on the chip (a long idle time-out keeps it there: no kernel boundary orders the caches, the rows must travel at system scope).  A
    stale input line would change the output; every step must be bit for bit the ordered launch."""
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
std = na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
S, n, steps = 1024, 128, 24
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev)          # the producer / consumer stream (foreign to the resident batch)
ref, b = na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0)
ref.AddStreams(std, S); b.AddStreams(std, S)
b.SetResidentLaunch(True)
g = torch.Generator(device="cpu").manual_seed(8)
src = torch.clamp(0.3 * torch.randn(steps, S, n, generator=g), -1.0, 1.0).to(dev)
x = torch.zeros(S, n, device=dev)            # ONE input buffer, rewritten on the device before every step
y, yref = torch.zeros(S, n, device=dev), torch.zeros(S, n, device=dev)
torch.cuda.synchronize(dev)
resident = 0
with torch.cuda.stream(ts):
    for k in range(steps):
        x.copy_(src[k] * 1.0)                # producer kernels on the foreign stream
        ts.synchronize()                     # contract (b): input rows complete before the call
        b.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
        resident += int(b.UsesResidentLaunch())
        b.WaitOutputs()                      # outputs valid; the launch stays resident (idle time-out 0.2 s)
        got = y.clone()                      # consumer kernel on the foreign stream
        ref.ProcessDevice(x.data_ptr(), yref.data_ptr(), n)   # ordered on ts: contract (a)
        ts.synchronize()
        assert torch.equal(got, yref), "step %%d differs" %% k
b.Synchronize(); ref.Synchronize()
print("RESIDENT_STEPS", resident)
''' % (ROOT, ROOT)
    env = dict(os.environ, NA_RESIDENT_IDLE_US="200000")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if not any(os.environ.get(k) for k in _KNOBS):
        assert "RESIDENT_STEPS 24" in r.stdout, r.stdout[-500:]


def test_the_launch_leaves_when_idle_comes_back_for_the_next_buffer_and_drains_for_every_state_change(na, std):
    """Real-time regime: one buffer, then nothing for longer than the idle time-out -- the launch must have left (a device-wide
    synchronisation returns) and the next buffer starts it again; joins, leaves, a re-prewarm, a host-buffer call and
    NA_BatchGetHipStream retire it first.  Bit for bit the ordered launches throughout."""
    import time
    import torch
    S, n = 700, 128
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(4)
    ts, ref, b = _pair(na, std, S)
    cap = S + 8

    def step(check_resident=True):
        rows = ref.NumStreams()
        x = torch.clamp(0.3 * torch.randn(cap, n, generator=g), -1.0, 1.0).to(dev)
        want, got = torch.zeros(cap, n, device=dev), torch.zeros(cap, n, device=dev)
        torch.cuda.synchronize(dev)
        ref.ProcessDevice(x.data_ptr(), want.data_ptr(), n)
        b.ProcessDevice(x.data_ptr(), got.data_ptr(), n)
        if check_resident:
            assert _resident_expected(b)
        ref.Synchronize()
        b.WaitOutputs()
        assert torch.equal(got[:rows], want[:rows])

    step()
    time.sleep(0.05)              # >> 200 us: every workgroup has idled out
    torch.cuda.synchronize(dev)   # (returns: nothing of the batch is left on the chip)
    step()                        # the launch comes back, resuming the command sequence
    for bb in (ref, b):
        bb.RemoveStreams(5, 3)
    step()
    for bb in (ref, b):
        assert bb.AddStreams(std, 2) == 5
    step()
    for bb in (ref, b):
        bb.Prewarm(9)
    step()
    xh = (0.3 * np.random.default_rng(1).standard_normal((ref.NumStreams(), n))).clip(-1, 1).astype(np.float32)
    assert np.array_equal(b.Process(xh), ref.Process(xh))
    step()
    # 256-frame buffers are two commands; 96 frames are not for the resident launch (the chains / ordered launches take them)
    for frames in (256, 96, 128):
        rows = ref.NumStreams()
        x = torch.clamp(0.3 * torch.randn(rows, frames, generator=g), -1.0, 1.0).to(dev)
        want, got = torch.zeros(rows, frames, device=dev), torch.zeros(rows, frames, device=dev)
        torch.cuda.synchronize(dev)
        ref.ProcessDevice(x.data_ptr(), want.data_ptr(), frames)
        b.ProcessDevice(x.data_ptr(), got.data_ptr(), frames)
        ref.Synchronize()
        b.Synchronize()
        assert torch.equal(got, want), frames
    assert b.GetHipStream() not in (None, 0)  # contract (a) from here on: ordered launches
    step(check_resident=False)
    assert not b.UsesResidentLaunch()
    ref.close()
    b.close()


def test_timing_marks_bracket_the_resident_steps(na, std):
    import torch
    S, n, steps = 1024, 128, 50
    dev = torch.device("cuda", 0)
    x = torch.zeros(S, n, device=dev)
    y = torch.zeros(S, n, device=dev)
    b = na.Batch(0)
    b.AddStreams(std, S)
    b.SetResidentLaunch(True)
    torch.cuda.synchronize(dev)
    for _ in range(2):
        b.MarkTime(0)
        for k in range(steps):
            b.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
        b.MarkTime(1)
        b.WaitMarks()
        ms = b.ElapsedMs()
        assert 0.0005 * steps < ms < 0.2 * steps, ms   # (between 0.5 us and 200 us per step: the events saw the work)
    b.Synchronize()
    assert bool(torch.isfinite(y).all().item())
    b.close()


def test_buffers_of_other_lengths_between_resident_steps_are_ordered_behind_them(na, std):
    """A buffer that is not a multiple of 128 frames cannot be a command to the resident launch: it runs as ordinary launches -- which must
    wait for every command still in flight (the same streams' state).  Bit for bit the ordered launches over a ragged sequence."""
    import torch
    S = 1024
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(12)
    ts, ref, b = _pair(na, std, S)
    lengths = [128, 128, 64, 128, 37, 256, 128, 5, 128, 128]
    x = torch.clamp(0.3 * torch.randn(S, sum(lengths), generator=g), -1.0, 1.0).to(dev)
    want, got = torch.zeros_like(x), torch.zeros_like(x)
    torch.cuda.synchronize(dev)
    total = x.shape[1]
    resident = 0
    for bb, y in ((ref, want), (b, got)):
        at = 0
        for n in lengths:
            bb.ProcessDevice(x[:, at:].data_ptr(), y[:, at:].data_ptr(), n, total, total)
            if bb is b:
                resident += int(b.UsesResidentLaunch())
            at += n
        bb.Synchronize()
    assert torch.equal(want, got)
    if not any(os.environ.get(k) for k in _KNOBS):
        assert resident == sum(1 for n in lengths if n % 128 == 0)
    ref.close()
    b.close()
