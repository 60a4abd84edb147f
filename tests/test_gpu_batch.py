"""GPU tests of the batch engine at the BASELINE configurations (through the C ABI), edge cases, and
size-independent properties at full batch size.  Tolerances: see test_gpu_parity.py."""
import os

import numpy as np
import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu

TOL_RMS = 2e-6


@pytest.fixture(scope="module")
def na():
    import neuralaudio_amd
    if neuralaudio_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the product path has no CPU fallback")
    return neuralaudio_amd


@pytest.fixture(scope="module")
def loader(na):
    return na.NeuralModelLoader()


def _path(name):
    return os.path.join(O.MODELS_DIR, name)


def _run_blocks(batch, x, n):
    return np.concatenate([batch.Process(x[:, i:i + n]) for i in range(0, x.shape[1], n)], axis=1)


def test_config3_mixed_lite_feather_nano_batch(na, loader):
    """BASELINE configs[2]: A1 Lite + Feather + Nano interleaved in one batch (scaled to 96 streams for the oracle)."""
    lite_arrays = O.a1_arrays(12, 6)
    lite_w = O.synth_wavenet_weights(lite_arrays, seed=33)
    lite = loader.CreateFromString(O.nam_json_wavenet_a1(12, 6, lite_w), ".nam", doPrewarm=False)
    feather = loader.CreateFromFile(_path("BossWN-feather.nam"), doPrewarm=False)
    nano = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)
    assert lite is not None and lite.IsStatic() and lite.GetReceptiveFieldSize() == 4092
    b = na.Batch(0)
    S, n, blocks = 96, 128, 6
    kinds = []
    for s in range(S):
        m = (lite, feather, nano)[s % 3]
        assert b.AddStreams(m, 1) == s
        kinds.append(s % 3)
    x = np.stack([O.signal_noise(n * blocks, 500 + s) for s in range(S)])
    y = _run_blocks(b, x, n)
    for s in (0, 1, 2, 45, 46, 47, 93, 94, 95):
        if kinds[s] == 0:
            ora = O.OracleWaveNet(lite_arrays, lite_w)
        else:
            ora = O.oracle_from_file(("", "BossWN-feather.nam", "BossWN-nano.nam")[kinds[s]])
        assert O.rms(y[s] - ora.process(x[s])) < TOL_RMS, s


def test_fused_launch_odd_group_sizes_two_streams_per_workgroup(na, loader):
    """>= 512 streams: one fused launch over three architectures with two streams per workgroup; odd group sizes leave a
    half-filled last workgroup per group (the shadow wave must not write), interleaved rows make the slot/row tables non-contiguous."""
    std = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    feather = loader.CreateFromFile(_path("BossWN-feather.nam"), doPrewarm=False)
    nano = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)
    b = na.Batch(0)
    first_f = b.AddStreams(feather, 201)     # rows 0..200 (contiguous group)
    first_s = b.AddStreams(std, 119)         # rows 201..319
    rows_n = []
    for k in range(100):                     # nano and extra feather streams interleaved: non-contiguous row tables
        rows_n.append(b.AddStreams(nano, 1))
        b.AddStreams(feather, 1)
    S = 201 + 119 + 200
    assert S >= 512
    n, blocks = 128, 3
    x = np.stack([O.signal_noise(n * blocks, 7000 + s) for s in range(S)])
    y = _run_blocks(b, x, n)
    assert np.all(np.isfinite(y))
    checks = [(first_f, "BossWN-feather.nam"), (first_f + 200, "BossWN-feather.nam"), (first_s, "BossWN-standard.nam"),
              (first_s + 118, "BossWN-standard.nam"), (rows_n[0], "BossWN-nano.nam"), (rows_n[-1], "BossWN-nano.nam"),
              (rows_n[-1] + 1, "BossWN-feather.nam")]
    for row, name in checks:
        assert O.rms(y[row] - O.oracle_from_file(name).process(x[row])) < TOL_RMS, (row, name)


def test_config5_a2_quality_sweep_and_midstream_switch(na, loader):
    """BASELINE configs[4]: A2 slimmable container, quality 0..1 per stream; a mid-stream quality change switches the
    active submodel and leaves the inactive one's state frozen (CompositeModel.h:94-100,176-181)."""
    a2 = loader.CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False)
    S, n = 12, 128
    b = na.Batch(0)
    qs = [s / (S - 1) for s in range(S)]
    for q in qs:
        b.AddStreams(a2, 1, quality=q)
    j = O.load_json("BossWN-a2.nam")
    for s, q in enumerate(qs):
        assert b.GetActiveSubModel(s) == O.quality_to_submodel(j, q)
    x = np.stack([O.signal_sine(n * 6, start=311 * s) for s in range(S)])
    y1 = _run_blocks(b, x[:, :n * 3], n)
    b.SetQuality(2, 1.0)   # Lite -> Full
    b.SetQuality(9, 0.0)   # Full -> Lite
    y2 = _run_blocks(b, x[:, n * 3:], n)
    y = np.concatenate([y1, y2], axis=1)
    for s in (0, 5, 6, 11):
        yo = O.oracle_from_file("BossWN-a2.nam", quality=qs[s]).process(x[s])
        assert O.rms(y[s] - yo) < TOL_RMS, s
    for s, q_before, q_after in ((2, qs[2], 1.0), (9, qs[9], 0.0)):
        first = O.oracle_from_file("BossWN-a2.nam", quality=q_before)   # sees only the first half
        second = O.oracle_from_file("BossWN-a2.nam", quality=q_after)   # prewarmed, sees only the second half
        yo = np.concatenate([first.process(x[s, :n * 3]), second.process(x[s, n * 3:])])
        assert O.rms(y[s] - yo) < TOL_RMS, s


def test_config4_lstm_2x16_many_streams_and_mixed_with_wavenet(na, loader):
    """BASELINE configs[3] (LSTM half): LSTM 2x16 (synthetic weights, 150 streams = 3 waves) + WaveNet streams in one batch."""
    w = O.synth_lstm_weights(2, 16, seed=8)
    lstm = loader.CreateFromString(O.nam_json_lstm(2, 16, w), ".nam", doPrewarm=False)
    nano = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)
    b = na.Batch(0)
    b.AddStreams(lstm, 150)
    b.AddStreams(nano, 10)
    n, blocks = 128, 3
    x = np.stack([O.signal_noise(n * blocks, 900 + s) for s in range(160)])
    y = _run_blocks(b, x, n)
    for s in (0, 63, 64, 149):
        yo = O.OracleLSTM.from_nam(2, 16, w).process(x[s])
        assert O.rms(y[s] - yo) < 5e-6, s
    for s in (150, 159):
        assert O.rms(y[s] - O.oracle_from_file("BossWN-nano.nam").process(x[s])) < TOL_RMS, s


def test_config4_keras_gru_next_to_lstm_2x16(na, loader):
    """BASELINE configs[3]: LSTM 2x16 + keras GRU (H=16) streams, half/half, in one batch.  GRU = RTNeural's arithmetic in the
    reference (parity unpinned): checked against the oracle restatement (which test_oracle.py checks against torch.nn.GRU)."""
    import json
    w = O.synth_lstm_weights(2, 16, seed=8)
    lstm = loader.CreateFromString(O.nam_json_lstm(2, 16, w), ".nam", doPrewarm=True)
    gj = O.synth_keras_gru(1, 16, seed=21)
    gru = loader.CreateFromString(json.dumps(gj), ".json", doPrewarm=True)
    assert gru is not None and gru.GetReceptiveFieldSize() == -1
    b = na.Batch(0)
    b.AddStreams(lstm, 70)
    b.AddStreams(gru, 70)
    n, blocks = 128, 3
    x = np.stack([O.signal_noise(n * blocks, 300 + s) for s in range(140)])
    y = _run_blocks(b, x, n)
    for s in (0, 69):
        assert O.rms(y[s] - O.OracleLSTM.from_nam(2, 16, w).process(x[s])) < 5e-6, s
    for s in (70, 100, 139):
        assert O.rms(y[s] - O.OracleGRU(gj).process(x[s])) < 5e-6, s


def test_one_launch_serves_every_recurrent_layout(na, loader):
    """Six recurrent models in one batch = one fused RecurrentDppKernel launch: LSTM 1x24 and GRU 1x24 (32-unit layouts, which also size the
    launch's LDS), LSTM 1x12 (padded into the 16-unit layout), LSTM 2x8 (two layers side by side), LSTM 2x16, GRU 1x16; ragged stream counts."""
    import json
    specs = [("lstm", 1, 24, 9), ("gru", 1, 24, 7), ("lstm", 1, 12, 5), ("lstm", 2, 8, 11), ("lstm", 2, 16, 3), ("gru", 1, 16, 6)]
    b = na.Batch(0)
    oracles = []
    for kind, layers, hidden, count in specs:
        if kind == "lstm":
            w = O.synth_lstm_weights(layers, hidden, seed=60 + hidden + layers)
            m = loader.CreateFromString(O.nam_json_lstm(layers, hidden, w), ".nam", doPrewarm=True)
            make = (lambda L=layers, H=hidden, W=w: O.OracleLSTM.from_nam(L, H, W))
        else:
            gj = O.synth_keras_gru(layers, hidden, seed=60 + hidden)
            m = loader.CreateFromString(json.dumps(gj), ".json", doPrewarm=True)
            make = (lambda G=gj: O.OracleGRU(G))
        assert m is not None
        b.AddStreams(m, count)
        oracles += [make] * count
    n, blocks = 128, 3
    x = np.stack([O.signal_noise(n * blocks, 700 + s) for s in range(len(oracles))])
    y = _run_blocks(b, x, n)
    for s in range(len(oracles)):
        if not os.environ.get("NA_REC_NO_DPP32") and not os.environ.get("NA_LSTM_NO_DPP") and not os.environ.get("NA_LSTM_LANE_KERNEL") and not os.environ.get("NA_REC_QUAD_MIN"):
            assert b.StreamKernelName(s) == "RecurrentDppKernel"
        assert O.rms(y[s] - oracles[s]().process(x[s])) < 5e-6, (s, specs)


@pytest.mark.parametrize("layers,hidden", [(1, 16), (2, 8), (1, 12), (2, 20), (1, 5), (2, 13), (1, 17), (1, 20), (1, 24), (1, 25), (1, 32)])
def test_keras_gru_single_stream_shapes(na, loader, layers, hidden):
    import json
    gj = O.synth_keras_gru(layers, hidden, seed=40 + hidden + layers)
    m = loader.CreateFromString(json.dumps(gj), ".aidax", doPrewarm=True)
    x = O.signal_sine(1000)
    y = np.concatenate([m.Process(x[i:i + 100]) for i in range(0, x.size, 100)])
    assert O.rms(y - O.OracleGRU(gj).process(x)) < 5e-6


@pytest.mark.parametrize("layers,hidden", [(3, 16), (1, 5), (2, 24), (1, 40)])
def test_runtime_shaped_keras_gru_matches_oracle(na, loader, layers, hidden):
    """GRU shapes without a shaped kernel run on the lane = stream runtime-shaped one (RTNeural's run-time model takes any)."""
    import json
    j = O.synth_keras_gru(layers, hidden, seed=11 + hidden)
    m = loader.CreateFromString(json.dumps(j), ".json")
    assert m is not None
    x = O.signal_noise(300, 8)
    y = np.concatenate([m.Process(x[i:i + 100]) for i in range(0, x.size, 100)])
    assert O.rms(y - O.OracleGRU(j).process(x)) < 5e-6


@pytest.mark.parametrize("sizes", [[1, 15, 16, 17, 63, 64, 65, 127, 128], [300, 5, 129, 128, 1]])
def test_ragged_buffer_sizes(na, loader, sizes):
    """Any n per call (the reference chunks at 64, InternalModel.h:104-117): 1, tile-straddling, > 128 (multi-launch)."""
    m = loader.CreateFromFile(_path("BossWN-feather.nam"))
    ora = O.oracle_from_file("BossWN-feather.nam")
    x = O.signal_sine(sum(sizes))
    out, pos = [], 0
    for n in sizes:
        out.append(m.Process(x[pos:pos + n]))
        pos += n
    assert O.rms(np.concatenate(out) - ora.process(x)) < TOL_RMS


def test_empty_and_inplace_and_no_prewarm(na, loader):
    m = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)   # fresh model: zero history (WaveNet.h:38-40)
    assert m.Process(np.zeros(0, np.float32)).size == 0
    ora = O.oracle_from_file("BossWN-nano.nam", prewarm=False)
    x = O.signal_sine(512)
    assert O.rms(m.Process(x) - ora.process(x)) < TOL_RMS
    # Prewarm() re-establishes the zero-input steady state at any time
    m.Prewarm()
    ora2 = O.oracle_from_file("BossWN-nano.nam", prewarm=True)
    assert O.rms(m.Process(x) - ora2.process(x)) < TOL_RMS
    # input == output through the raw C ABI (NeuralModel.h:127 contract)
    import ctypes as C
    from neuralaudio_amd import capi
    buf = x.copy()
    p = buf.ctypes.data_as(C.POINTER(C.c_float))
    m2 = loader.CreateFromFile(_path("BossWN-nano.nam"))
    capi.load_library().Process(m2._h, p, p, buf.size)
    assert O.rms(buf - O.oracle_from_file("BossWN-nano.nam").process(x)) < TOL_RMS


def test_lstm_prewarm_continues_from_current_state(na, loader):
    """NeuralModelImpl::Prewarm runs 2048 zeros from the CURRENT state (InternalModel.h:368-371)."""
    m = loader.CreateFromFile(_path("BossLSTM-1x16.nam"))
    ora = O.oracle_from_file("BossLSTM-1x16.nam")
    x = O.signal_sine(300)
    y0 = m.Process(x)
    m.Prewarm()
    y1 = m.Process(x)
    o0 = ora.process(x)
    ora.prewarm()
    o1 = ora.process(x)
    assert O.rms(y0 - o0) < 5e-6 and O.rms(y1 - o1) < 5e-6


def test_full_size_properties_1024_streams(na, loader):
    """BASELINE configs[1] at full size (A1 Standard, 1024 streams x 128): size-independent properties.
    (a) streams fed the same input produce bit-identical output wherever they sit in the batch;
    (b) processing 2 x 64 samples equals 1 x 128 bit-for-bit (chunk invariance);
    (c) spot-check 3 streams against the oracle."""
    m = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    S, n = 1024, 128
    base = np.stack([O.signal_noise(n * 2, 7000 + s) for s in range(8)])
    x = base[np.arange(S) % 8]
    b1 = na.Batch(0)
    b1.AddStreams(m, S)
    y = _run_blocks(b1, x, n)
    for s in range(8, S, 97):
        assert np.array_equal(y[s], y[s % 8]), s
    b2 = na.Batch(0)
    b2.AddStreams(m, S)
    y64 = _run_blocks(b2, x, 64)
    assert np.array_equal(y, y64)
    for s in (0, 511, 1023):
        assert O.rms(y[s] - O.oracle_from_file("BossWN-standard.nam").process(x[s])) < TOL_RMS, s


def test_zero_input_stays_at_steady_state(na, loader):
    """After prewarm, zero input must give the constant steady-state output forever (prewarm == fixed point)."""
    m = loader.CreateFromFile(_path("BossWN-standard.nam"))
    y = np.concatenate([m.Process(np.zeros(128, np.float32)) for _ in range(40)])   # > receptive field
    assert np.max(np.abs(y - y[0])) < 1e-6
    assert abs(y[0] - O.oracle_from_file("BossWN-standard.nam").process(np.zeros(4, np.float32))[0]) < 1e-6


# ------------------------------------------------------------------------------------------------ round 2 additions

def test_pipelined_submit_collect_matches_process(na, loader):
    """NA_BatchSubmit / NA_BatchCollect (upload, kernels and download of neighbouring buffers overlap) and the zero-copy pinned
    variants give bit-for-bit what the synchronous NA_BatchProcess gives."""
    m = loader.CreateFromFile(_path("BossWN-feather.nam"), doPrewarm=False)
    S, n, blocks = 48, 128, 9
    x = np.stack([O.signal_noise(n * blocks, 900 + s) for s in range(S)])
    ref = na.Batch(0)
    ref.AddStreams(m, S)
    want = _run_blocks(ref, x, n)
    b = na.Batch(0)
    b.AddStreams(m, S)
    tickets, got = [], []
    for i in range(blocks):
        tickets.append(b.Submit(x[:, i * n:(i + 1) * n]))
        if len(tickets) == 3:  # at most 3 in flight
            got.append(b.Collect(tickets.pop(0)))
    while tickets:
        got.append(b.Collect(tickets.pop(0)))
    assert np.array_equal(np.concatenate(got, axis=1), want)
    with pytest.raises(na.NeuralAudioError):
        b.Collect((0, (S, n)))  # nothing in flight any more
    z = na.Batch(0)
    z.AddStreams(m, S)
    tickets, got = [], []
    for i in range(blocks):
        z.NextInput(n)[:] = x[:, i * n:(i + 1) * n]
        tickets.append(z.SubmitInput(n))
        if len(tickets) == 2:
            got.append(z.CollectView(tickets.pop(0)).copy())
    while tickets:
        got.append(z.CollectView(tickets.pop(0)).copy())
    assert np.array_equal(np.concatenate(got, axis=1), want)
    for k in range(3):
        z.Submit(x[:, :n])
    with pytest.raises(na.NeuralAudioError, match="every pipeline slot in flight"):
        z.Submit(x[:, :n])


def test_pipelined_half_chains_match_process_and_mix_with_the_other_entry_points(na, loader):
    """From 512 streams of one contiguous WaveNet group a submitted buffer runs as TWO launches of half the streams on two free-running
    HIP streams (gpu_batch.h halfStream): bit for bit what NA_BatchProcess gives, also with an odd stream count, ragged buffer
    lengths (96 frames = two launches per chain), a synchronous call / a join / a leave between submissions, and three tickets in flight."""
    m = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    S, n = 1027, 128
    rng = np.random.default_rng(31)
    base = (0.3 * rng.standard_normal((11, n * 8))).clip(-1, 1).astype(np.float32)
    x = base[np.arange(S) % 11]
    ref, b = na.Batch(0), na.Batch(0)
    ref.AddStreams(m, S)
    b.AddStreams(m, S)

    def step(sl, frames=n):
        blk = np.ascontiguousarray(x[:, sl * n:sl * n + frames])
        return ref.Process(blk), blk

    def submit(blk):  # in place (a submission the library has to copy stays one launch: the host thread is the bottleneck there)
        b.NextInput(blk.shape[1])[:] = blk
        return b.SubmitInput(blk.shape[1])

    tickets, got, want = [], [], []
    for i in range(4):
        w, blk = step(i)
        want.append(w)
        tickets.append(submit(blk))
        if len(tickets) == 3:
            got.append(b.Collect(tickets.pop(0)))
        if i >= 1:
            assert _halves_as_expected(b)  # (from the second ticket in flight on)
    while tickets:
        got.append(b.Collect(tickets.pop(0)))
    assert np.array_equal(np.concatenate(got, axis=1), np.concatenate(want, axis=1))
    w, blk = step(4)  # a synchronous call between submissions (drains the half chains first)
    assert np.array_equal(b.Process(blk), w)
    w, blk = step(5, 96)  # 96 frames = a 64- and a 32-frame launch per chain
    assert np.array_equal(b.Collect(submit(blk)), w)
    w, blk = step(6)  # a copying submission between in-place ones
    t1 = submit(blk)
    w2, blk2 = step(7)
    t2 = b.Submit(blk2)
    assert np.array_equal(b.Collect(t1), w) and np.array_equal(b.Collect(t2), w2)
    assert ref.AddStreams(m, 2) == b.AddStreams(m, 2) == S  # streams join ...
    ref.RemoveStreams(3, 1)
    b.RemoveStreams(3, 1)  # ... and one leaves: the group is no longer contiguous -> the halves are cut out of its index lists
    x2 = np.concatenate([x, x[:2]], axis=0)
    for i in (6, 7):
        blk = np.ascontiguousarray(x2[:, i * n:(i + 1) * n])
        ta = submit(blk)
        blk_b = np.ascontiguousarray(x2[:, (i - 4) * n:(i - 3) * n])
        tb = submit(blk_b)
        assert np.array_equal(b.Collect(ta), ref.Process(blk))
        assert np.array_equal(b.Collect(tb), ref.Process(blk_b))
    ref.close()
    b.close()


def test_device_pointer_steps_on_the_batch_own_streams_run_as_free_running_halves(na, loader):
    """NA_BatchProcessDevice on a batch that created its own stream (and never handed it out): from 512 streams of one contiguous WaveNet
    group every step is two free-running launches of half the streams each.  Bit for bit the one-launch result (a batch on the
    caller's stream), through a synchronous host call in between, and ordered on the batch stream once NA_BatchGetHipStream was
    called.  The timing marks bracket every launch stream."""
    import torch
    m = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    S, n, steps = 1026, 128, 6
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.clamp(0.3 * torch.randn(steps, S, n, generator=g), -1.0, 1.0).to(dev)
    ts = torch.cuda.Stream(device=dev)
    ref, b = na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0)
    ref.AddStreams(m, S)
    b.AddStreams(m, S)
    want = torch.empty(steps, S, n, device=dev)
    got = torch.empty(steps, S, n, device=dev)
    torch.cuda.synchronize(dev)
    for k in range(steps):
        ref.ProcessDevice(x[k].data_ptr(), want[k].data_ptr(), n)
        assert not ref.UsesHalfLaunches()
    ref.Synchronize()
    b.MarkTime(0)
    for k in range(3):
        b.ProcessDevice(x[k].data_ptr(), got[k].data_ptr(), n)
        assert _halves_as_expected(b)
    b.MarkTime(1)
    ms = b.ElapsedMs()
    assert 0.0 < ms < 50.0, ms
    b.Synchronize()
    assert torch.equal(got[:3], want[:3])
    # a synchronous host-buffer call joins the chains, runs ordered, and the next device step starts new chains behind it
    assert np.array_equal(b.Process(x[3].cpu().numpy()), want[3].cpu().numpy())
    b.ProcessDevice(x[4].data_ptr(), got[4].data_ptr(), n)
    assert _halves_as_expected(b)
    # the stream handed out: from here on one ordered launch per step
    assert b.GetHipStream() not in (None, 0)
    b.ProcessDevice(x[5].data_ptr(), got[5].data_ptr(), n)
    assert not b.UsesHalfLaunches()
    b.Synchronize()
    assert torch.equal(got[4:], want[4:])
    ref.close()
    b.close()


def test_buffers_longer_than_a_launch_run_their_chunks_on_the_chains(na, loader):
    """A buffer of 352 frames is 128 + 128 + 64 + 32: each chain runs its four launches one after the other, the chains never wait for
    each other; 96 frames = 64 + 32, and the compact-ring models' safe lengths apply.  Bit for bit the ordered launches."""
    import torch
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device=dev)
    g = torch.Generator(device="cpu").manual_seed(9)
    for name, S in (("BossWN-standard.nam", 600), ("BossWN-feather.nam", 1100)):
        m = loader.CreateFromFile(_path(name), doPrewarm=False)
        ref, b = na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0)
        ref.AddStreams(m, S)
        b.AddStreams(m, S)
        for n in (352, 96, 128, 200, 17):
            x = torch.clamp(0.3 * torch.randn(S, n, generator=g), -1.0, 1.0).to(dev)
            want, got = torch.zeros(S, n, device=dev), torch.zeros(S, n, device=dev)
            torch.cuda.synchronize(dev)
            ref.ProcessDevice(x.data_ptr(), want.data_ptr(), n)
            b.ProcessDevice(x.data_ptr(), got.data_ptr(), n)
            assert _halves_as_expected(b) and not ref.UsesHalfLaunches()
            ref.Synchronize()
            b.Synchronize()
            assert torch.equal(want, got), (name, n)
        ref.close()
        b.close()


def _halves_as_expected(batch):
    """Outside a forced-family run (tests/test_gpu_families.py: fallback kernels run ordered launches, the results must still agree)
    the batch must have run its last step as two half-batch launches -- or, A1 Standard, as a command to the resident launch."""
    if any(os.environ.get(k) for k in ("NA_WN_KERNEL", "NA_WN_SPEC", "NA_WN_PACK", "NA_HOST_HALVES", "NA_HOST_DIRECT", "NA_SP_T", "NA_SP_GEN")):
        return True
    return batch.UsesHalfLaunches() or batch.UsesResidentLaunch()


def _device_steps(batch, x, out, n):
    for k in range(x.shape[0]):
        batch.ProcessDevice(x[k].data_ptr(), out[k].data_ptr(), n)


def test_mixed_and_packed_batches_run_as_free_running_halves_too(na, loader):
    """The half-batch launches cut EVERY group of a fused launch in two: a Lite + Feather + Nano batch (padded / packed 2 / packed 4
    streams per kernel-level stream, one packed launch) and an A2 batch whose streams change quality between steps (index lists
    re-uploaded while the chains are in flight) give bit for bit what the ordered launches on a caller's stream give."""
    import torch
    dev = torch.device("cuda", 0)
    lite_w = O.synth_wavenet_weights(O.a1_arrays(12, 6), seed=33)
    lite = loader.CreateFromString(O.nam_json_wavenet_a1(12, 6, lite_w), ".nam", doPrewarm=False)
    feather = loader.CreateFromFile(_path("BossWN-feather.nam"), doPrewarm=False)
    nano = loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)
    a2 = loader.CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False)
    n, steps = 128, 5
    ts = torch.cuda.Stream(device=dev)
    g = torch.Generator(device="cpu").manual_seed(6)

    def pair():
        return na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0)

    # (a) config-3 shaped: 601 Lite + 603 Feather + 806 Nano, two streams removed so that the lists are not contiguous
    ref, b = pair()
    for bb in (ref, b):
        bb.AddStreams(lite, 601)
        bb.AddStreams(feather, 603)
        bb.AddStreams(nano, 806)
        bb.RemoveStreams(5, 1)
        bb.RemoveStreams(1300, 2)
    S = 601 + 603 + 806
    x = torch.clamp(0.3 * torch.randn(steps, S, n, generator=g), -1.0, 1.0).to(dev)
    want, got = torch.zeros(steps, S, n, device=dev), torch.zeros(steps, S, n, device=dev)
    torch.cuda.synchronize(dev)
    _device_steps(ref, x, want, n)
    ref.Synchronize()
    _device_steps(b, x, got, n)
    assert _halves_as_expected(b) and not ref.UsesHalfLaunches()
    b.Synchronize()
    assert torch.equal(got, want)
    assert float(want[:, 0].abs().max()) > 0 and float(want[:, 5].abs().max()) == 0  # (a removed row stays untouched)
    ref.close()
    b.close()

    # (b) 1200 A2 streams, every third one on the small submodel; between the steps a few streams switch (both directions)
    ref, b = pair()
    S = 1200
    for bb in (ref, b):
        bb.AddStreams(a2, S, quality=1.0)
        for s in range(0, S, 3):
            bb.SetQuality(s, 0.0)
    x = torch.clamp(0.3 * torch.randn(steps, S, n, generator=g), -1.0, 1.0).to(dev)
    want, got = torch.zeros(steps, S, n, device=dev), torch.zeros(steps, S, n, device=dev)
    torch.cuda.synchronize(dev)
    for k in range(steps):
        for bb, out in ((ref, want), (b, got)):
            bb.SetQuality(7 * k + 1, 0.0)
            bb.SetQuality(3 * k, 1.0)
            bb.ProcessDevice(x[k].data_ptr(), out[k].data_ptr(), n)
        assert _halves_as_expected(b) and not ref.UsesHalfLaunches()
    ref.Synchronize()
    b.Synchronize()
    assert torch.equal(got, want)
    ref.close()
    b.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("NA_CHAIN_WALK_SEEDS", "1"))))  # (soak runs: NA_CHAIN_WALK_SEEDS=16)
@pytest.mark.parametrize("kind", ["standard", "a2", "narrow"])
def test_random_joins_leaves_switches_between_free_running_steps_match_ordered_launches(na, loader, kind, seed):
    """Seeded random walks over a batch large enough for the half-batch chains: streams leave and join (ids and state slots recycled,
    index lists re-uploaded), A2 streams switch quality, a stream is re-prewarmed, host-buffer calls and pipelined submissions are
    mixed in -- between device-pointer steps that run as two free-running chains.  Every step must be bit for bit what a batch on a
    caller's stream (ordered launches) produces under the same operations."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng({"standard": 11, "a2": 12, "narrow": 13}[kind] + 100 * seed)
    if kind == "standard":
        parts = [(loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False), 700)]
    elif kind == "a2":
        parts = [(loader.CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False), 1300)]
    else:
        parts = [(loader.CreateFromFile(_path("BossWN-feather.nam"), doPrewarm=False), 900), (loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False), 1500)]
    ts = torch.cuda.Stream(device=dev)
    ref, b = na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0)
    for bb in (ref, b):
        for m, count in parts:
            bb.AddStreams(m, count, quality=1.0)
    n, cap = 128, sum(c for _, c in parts) + 64
    g = torch.Generator(device="cpu").manual_seed(3)
    halves_seen = 0
    for step in range(14):
        op = int(rng.integers(0, 7))
        rows = ref.NumStreams()
        live = [s for s in range(rows) if ref.IsLive(s)]
        if op == 0 and len(live) > 600:
            first = int(rng.choice(live[:-4]))
            cnt = 1 + int(rng.integers(0, 3))
            if all(ref.IsLive(s) for s in range(first, first + cnt)):
                for bb in (ref, b):
                    bb.RemoveStreams(first, cnt)
        elif op == 1 and rows < cap - 8:
            m = parts[int(rng.integers(0, len(parts)))][0]
            cnt = 1 + int(rng.integers(0, 4))
            assert ref.AddStreams(m, cnt, quality=1.0) == b.AddStreams(m, cnt, quality=1.0)
        elif op == 2 and kind == "a2":
            for s_ in rng.choice(live, size=40, replace=False):
                q = float(rng.integers(0, 2))
                for bb in (ref, b):
                    bb.SetQuality(int(s_), q)
        elif op == 3:
            s_ = int(rng.choice(live))
            for bb in (ref, b):
                bb.Prewarm(s_)
        rows = ref.NumStreams()
        assert rows == b.NumStreams()
        x = torch.clamp(0.3 * torch.randn(rows, n, generator=g), -1.0, 1.0)
        if op == 4:  # a blocking host-buffer call
            want, got = ref.Process(x.numpy()), b.Process(x.numpy())
            assert np.array_equal(want, got), (kind, step, "Process")
            continue
        if op == 5:  # two pipelined submissions in place
            xa, xb = x.numpy(), np.ascontiguousarray(x.numpy()[:, ::-1])
            outs = []
            for bb in (ref, b):
                bb.NextInput(n)[:] = xa
                t1 = bb.SubmitInput(n)
                bb.NextInput(n)[:] = xb
                t2 = bb.SubmitInput(n)
                outs.append((bb.Collect(t1), bb.Collect(t2)))
            assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), (kind, step, "Submit")
            continue
        xd = x.to(dev)
        want, got = torch.zeros(rows, n, device=dev), torch.zeros(rows, n, device=dev)
        torch.cuda.synchronize(dev)
        ref.ProcessDevice(xd.data_ptr(), want.data_ptr(), n)
        b.ProcessDevice(xd.data_ptr(), got.data_ptr(), n)
        halves_seen += int(b.UsesHalfLaunches() or b.UsesResidentLaunch())  # (free-running either way)
        assert not ref.UsesHalfLaunches()  # (a device-pointer step on a caller's stream is always ordered)
        ref.Synchronize()
        b.Synchronize()
        assert torch.equal(want, got), (kind, step, op)
    if not any(os.environ.get(k) for k in ("NA_WN_KERNEL", "NA_WN_SPEC", "NA_WN_PACK", "NA_HOST_HALVES", "NA_HOST_DIRECT", "NA_SP_T", "NA_SP_GEN")):
        assert halves_seen >= (3 if seed == 0 else 1), halves_seen
    ref.close()
    b.close()


def test_quality_switch_every_buffer_2048_a2_streams_is_cheap(na, loader):
    """BASELINE configs[4] size on one GPU (2048 A2 streams): flipping the quality of HALF the streams before EVERY buffer must stay a
    real-time operation -- the active-stream lists travel from pinned memory asynchronously, nothing is allocated, synchronised or
    re-captured.  p99 per-buffer latency through the host-buffer entry point < 1 ms (the north star's bound), and the switch is
    reported real-time safe (LoadAll: every submodel was prewarmed, CompositeModel.h:44-50)."""
    import time
    if os.environ.get("NA_WN_KERNEL") == "generic":
        pytest.skip("the runtime-shaped kernel launches every model group on its own: a switch re-captures the batch's hipGraph")
    m = loader.CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False)
    S, n = 2048, 128
    b = na.Batch(0)
    b.AddStreams(m, S, quality=1.0)
    x = np.stack([O.signal_noise(n, 40 + (s % 16)) for s in range(S)])
    for _ in range(5):
        b.Process(x)
    assert b.IsQualityChangeRealtimeSafe(0, 0.0) and b.IsQualityChangeRealtimeSafe(0, 1.0)
    lat = []
    for it in range(120):
        q = 0.0 if it % 2 == 0 else 1.0
        for s in range(0, S, 2):
            b.SetQuality(s, q)
        t0 = time.perf_counter()
        y = b.Process(x)
        lat.append((time.perf_counter() - t0) * 1e3)
        assert b.GetActiveSubModel(0) == (0 if q == 0.0 else 1) and b.GetActiveSubModel(1) == 1
    lat = sorted(lat[10:])
    assert np.all(np.isfinite(y))
    # (a switch that re-captured or re-uploaded anything would cost several ms every time; a host hiccup on a shared box is not that)
    assert lat[len(lat) // 2] < 1.0 and lat[int(len(lat) * 0.99)] < 2.5, lat[-5:]


def test_a2_stream_with_switches_matches_two_oracles(na, loader):
    """A stream that alternates between the two A2 submodels: each submodel only advances while it is active (CompositeModel::Process,
    CompositeModel.h:94-100), so its output equals an oracle of that submodel fed only the buffers it was active for."""
    m = loader.CreateFromFile(_path("BossWN-a2.nam"))
    b = na.Batch(0)
    b.AddStreams(m, 3, quality=1.0)
    n, blocks = 128, 16
    x = O.signal_noise(n * blocks, 77)
    ora = {0: O.oracle_from_file("BossWN-a2.nam", quality=0.0), 1: O.oracle_from_file("BossWN-a2.nam", quality=1.0)}
    for i in range(blocks):
        q = 1.0 if (i // 3) % 2 == 0 else 0.2
        b.SetQuality(1, q)
        xb = x[i * n:(i + 1) * n]
        y = b.Process(np.stack([xb, xb, xb]))
        idx = b.GetActiveSubModel(1)
        assert idx == (1 if q == 1.0 else 0)
        assert O.rms(y[1] - ora[idx].process(xb)) < TOL_RMS, i


def test_ondemand_composite_load_mode(na):
    """ECompositeModelLoadMode::OnDemand (CompositeModel.h:52-60,104-109): only the current submodel is prewarmed at load; the first
    switch to another one prewarms it (not real-time safe, reported beforehand) and from then on switching is real-time safe.  The
    audio is the same as with LoadAll: a submodel that was never run sits at its prewarmed state either way."""
    ld = na.NeuralModelLoader()
    ld.SetCompositeModelLoadMode(na.ECompositeModelLoadMode.OnDemand)
    ld.SetDefaultQualityScaleFactor(1.0)
    m = ld.CreateFromFile(_path("BossWN-a2.nam"))
    x = O.signal_sine(512)
    y1 = np.concatenate([m.Process(x[i:i + 128]) for i in range(0, 256, 128)])
    assert m.IsQualityChangeRealtimeSafe(0.9) and not m.IsQualityChangeRealtimeSafe(0.1)
    m.SetQualityScaleFactor(0.1)
    y0 = np.concatenate([m.Process(x[i:i + 128]) for i in range(256, 512, 128)])
    assert m.IsQualityChangeRealtimeSafe(0.1) and m.IsQualityChangeRealtimeSafe(1.0)  # both had their prewarm now
    assert O.rms(y1 - O.oracle_from_file("BossWN-a2.nam", quality=1.0).process(x[:256])) < TOL_RMS
    assert O.rms(y0 - O.oracle_from_file("BossWN-a2.nam", quality=0.0).process(x[256:])) < TOL_RMS
    la = na.NeuralModelLoader().CreateFromFile(_path("BossWN-a2.nam"))
    assert la.IsQualityChangeRealtimeSafe(0.1)  # LoadAll: everything prewarmed at load -- also before the first Process()
    la.Process(x[:128])
    assert la.IsQualityChangeRealtimeSafe(0.1)
    # HadInitialPrewarm (CompositeModel.h:44-50) in LoadAll mode too: a model created WITHOUT prewarm has no prewarmed submodel to switch to
    cold = na.NeuralModelLoader().CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False)
    assert cold.IsQualityChangeRealtimeSafe(1.0) and not cold.IsQualityChangeRealtimeSafe(0.1)
    cold.Process(x[:128])
    assert not cold.IsQualityChangeRealtimeSafe(0.1)
    cold.Prewarm()
    assert cold.IsQualityChangeRealtimeSafe(0.1)
    # a mixed batch that takes several launches per buffer re-captures its hipGraph on a switch: reported as not real-time safe
    b = na.Batch(0)
    b.AddStreams(la, 2, quality=1.0)
    if os.environ.get("NA_WN_KERNEL") == "generic":
        return  # (every group of the runtime-shaped kernel is a launch of its own)
    assert b.IsQualityChangeRealtimeSafe(0, 0.1)  # both submodels ride in one frame-kernel launch
    b.AddStreams(na.NeuralModelLoader().CreateFromFile(_path("BossLSTM-1x16.nam")), 1)
    assert not b.IsQualityChangeRealtimeSafe(0, 0.1)  # WaveNet launch + recurrent launch: two units


def test_quality_setter_from_another_thread(na, loader):
    """SetQualityScaleFactor may come from a UI thread while the audio thread is inside Process (the reference keeps the index in
    atomics, CompositeModel.h:122,196-197).  Here the setter only stores atomically and the audio thread applies the change at the
    top of its next Process: hammering it from a second thread must never break a buffer."""
    import threading
    m = loader.CreateFromFile(_path("BossWN-a2.nam"))
    stop = threading.Event()

    def ui():
        k = 0
        while not stop.is_set():
            m.SetQualityScaleFactor(0.05 if k % 2 else 0.95)
            k += 1

    t = threading.Thread(target=ui)
    t.start()
    try:
        x = O.signal_noise(128, 5)
        outs = [m.Process(x) for _ in range(300)]
    finally:
        stop.set()
        t.join()
    assert all(np.all(np.isfinite(o)) for o in outs)
    assert m.GetQualityScaleFactor() in (pytest.approx(0.05), pytest.approx(0.95))


def test_oversampling_rewrite_matches_oracle_with_scaled_dilations(na):
    """SetExternalSampleRate(96000) on a 48 kHz WaveNet multiplies every dilation by 2 and runs the model as a runtime-shaped one
    (OversampleNAMConfig, NeuralModel.cpp:92-130).  The HIP path must equal an oracle built with those dilations."""
    ld = na.NeuralModelLoader()
    ld.SetExternalSampleRate(96000)
    for name in ("BossWN-nano.nam", "BossWN-standard.nam"):
        m = ld.CreateFromFile(_path(name))
        assert m is not None and not m.IsStatic() and m.GetReceptiveFieldSize() == 2 * 4092
        j = O.load_json(name)
        arrays = O.wavenet_arrays_from_nam(j)
        for a in arrays:
            a["dilations"] = [2 * d for d in a["dilations"]]
        x = O.signal_sine(2048)
        y = np.concatenate([m.Process(x[i:i + 128]) for i in range(0, x.size, 128)])
        assert O.rms(y - O.OracleWaveNet(arrays, j["weights"]).process(x)) < TOL_RMS, name
    a2 = ld.CreateFromFile(_path("BossWN-a2.nam"))  # A2: dilations x 2 and head dilation 2
    j = O.load_json("BossWN-a2.nam")["config"]["submodels"][1]["model"]
    arrays = O.wavenet_arrays_from_nam(j)
    for a in arrays:
        a["dilations"] = [2 * d for d in a["dilations"]]
        a["head_dilation"] = 2
    x = O.signal_sine(1024)
    y = np.concatenate([a2.Process(x[i:i + 128]) for i in range(0, x.size, 128)])
    assert O.rms(y - O.OracleWaveNet(arrays, j["weights"]).process(x)) < TOL_RMS


def _full_size_properties(na, add, S, n, spot, tol, chunk_exact=True):
    """Size-independent properties at a BASELINE batch size: (a) streams of one model fed the same input agree bit-for-bit wherever
    they sit, (b) 2 x 64 samples == 1 x 128 bit-for-bit, (c) a few streams against the oracle.  `add(batch)` adds the streams and
    returns kind[s] (streams with equal kind share a model + quality); spot: [(stream, oracle factory)]."""
    base = np.stack([O.signal_noise(n * 2, 3000 + s) for s in range(4)])
    b1 = na.Batch(0)
    kind = add(b1)
    assert len(kind) == S and b1.NumStreams() == S
    x = base[np.arange(S) % 4]
    y = _run_blocks(b1, x, n)
    assert np.all(np.isfinite(y))
    first = {}
    for s in range(S):
        key = (kind[s], s % 4)
        first.setdefault(key, s)
    for s in range(0, S, 61):
        assert np.array_equal(y[s], y[first[(kind[s], s % 4)]]), s
    b2 = na.Batch(0)
    add(b2)
    y64 = _run_blocks(b2, x, 64)
    if chunk_exact:
        assert np.array_equal(y64, y)
    else:  # recurrent kernels: the block size may change the compiler's contraction of the per-sample arithmetic by an ulp
        assert np.max(np.abs(y64 - y)) < 1e-6
    for s, make in spot:
        assert O.rms(y[s] - make().process(x[s])) < tol, s


def test_full_size_properties_config3_4096_mixed_streams(na, loader):
    lite_arrays = O.a1_arrays(12, 6)
    lite_w = O.synth_wavenet_weights(lite_arrays, seed=33)
    models = [loader.CreateFromString(O.nam_json_wavenet_a1(12, 6, lite_w), ".nam", doPrewarm=False),
              loader.CreateFromFile(_path("BossWN-feather.nam"), doPrewarm=False), loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False)]
    S = 4096

    def add(b):
        kind = []
        for k, m in enumerate(models):  # sorted by architecture, a third each
            cnt = S // 3 + (1 if k < S % 3 else 0)
            b.AddStreams(m, cnt)
            kind += [k] * cnt
        return kind

    _full_size_properties(na, add, S, 128, [(0, lambda: O.OracleWaveNet(lite_arrays, lite_w)), (2000, lambda: O.oracle_from_file("BossWN-feather.nam")),
                                            (4095, lambda: O.oracle_from_file("BossWN-nano.nam"))], TOL_RMS)


def test_full_size_properties_config4_1024_lstm_and_gru_streams(na, loader):
    w = O.synth_lstm_weights(2, 16, seed=4)
    gj = O.synth_keras_gru(1, 16, seed=9)
    import json
    models = [loader.CreateFromString(O.nam_json_lstm(2, 16, w), ".nam"), loader.CreateFromString(json.dumps(gj), ".json")]
    S = 1024

    def add(b):
        b.AddStreams(models[0], S // 2)
        b.AddStreams(models[1], S // 2)
        return [0] * (S // 2) + [1] * (S // 2)

    _full_size_properties(na, add, S, 128, [(0, lambda: O.OracleLSTM.from_nam(2, 16, w)), (511, lambda: O.OracleLSTM.from_nam(2, 16, w)),
                                            (1023, lambda: O.OracleGRU(gj))], 5e-6, chunk_exact=False)


def test_full_size_properties_config5_2048_a2_streams_quality_sweep(na, loader):
    m = loader.CreateFromFile(_path("BossWN-a2.nam"), doPrewarm=False)
    S = 2048
    qs = [s / (S - 1) for s in range(S)]  # q_s = s / (S - 1): half the streams on each submodel

    def add(b):
        kind = []
        # streams are added in runs of equal submodel (AddStreams is per quality value; 2 runs would do, 16 exercise the run merging)
        for r in range(16):
            q = qs[r * 128]
            b.AddStreams(m, 128, quality=q)
            kind += [0 if q <= 0.5 else 1] * 128
        return kind

    _full_size_properties(na, add, S, 128, [(0, lambda: O.oracle_from_file("BossWN-a2.nam", quality=0.0)),
                                            (1023, lambda: O.oracle_from_file("BossWN-a2.nam", quality=0.5)),
                                            (2047, lambda: O.oracle_from_file("BossWN-a2.nam", quality=1.0))], TOL_RMS)


@pytest.mark.parametrize("name,first,more", [("BossWN-nano.nam", 3077, 5), ("BossWN-feather.nam", 1539, 3)])
def test_packed_narrow_streams_match_oracle_stream_by_stream(na, loader, name, first, more):
    """Large batches of a narrow static model run PACKED: 4 (Nano) / 2 (Feather) real streams as the channel groups of one virtual
    stream of the f16-split kernel (gpu_batch.cpp PackFor, wavenet_plan.cpp PackWaveNetDesc).  Every stream has its own input; streams
    from all positions of a virtual stream, from the partially filled last one and from a later AddStreams call that completes it
    must match the oracle, across a ragged sequence of block sizes, and a re-prewarm of one stream must not disturb its neighbours."""
    m = loader.CreateFromFile(_path(name), doPrewarm=False)
    b = na.Batch(0)
    b.AddStreams(m, first)
    b.AddStreams(m, more)  # joins the existing group: fills the last virtual stream and opens another
    S = first + more
    assert b.NumStreams() == S
    forced_unpacked = os.environ.get("NA_WN_KERNEL") in ("frame", "generic") or os.environ.get("NA_WN_PACK") == "0"  # tests/test_gpu_families.py
    assert b.StreamPackFactor(0) == b.StreamPackFactor(S - 1) == (1 if forced_unpacked else (4 if "nano" in name else 2))  # the packed path really runs
    rng = np.random.default_rng(77)
    n_total = 128 + 128 + 37 + 128
    x = (0.3 * rng.standard_normal((S, n_total))).clip(-1, 1).astype(np.float32)
    y = np.concatenate([b.Process(np.ascontiguousarray(x[:, a:a + c])) for a, c in ((0, 128), (128, 128), (256, 37), (293, 128))], axis=1)
    assert np.all(np.isfinite(y))
    picks = sorted(set([0, 1, 2, 3, 4, 5, first - 2, first - 1, first, first + 1, S - 1, S // 2, S // 3]))
    for s in picks:
        yo = O.oracle_from_file(name).process(x[s])
        assert O.rms(y[s] - yo) < TOL_RMS, (s, O.rms(y[s] - yo))
    # re-prewarm stream 1 only: it restarts from the steady state, its neighbours in the same virtual stream carry on
    b.Prewarm(1)
    x2 = (0.3 * rng.standard_normal((S, 128))).clip(-1, 1).astype(np.float32)
    y2 = b.Process(x2)
    assert O.rms(y2[1] - O.oracle_from_file(name).process(x2[1])) < TOL_RMS
    for s in (0, 2, 3):
        yo = O.oracle_from_file(name).process(np.concatenate([x[s], x2[s]]))[n_total:]
        assert O.rms(y2[s] - yo) < TOL_RMS, s


def test_registered_host_blocks_are_processed_in_place(na, loader):
    """NA_RegisterHostBuffer: NA_BatchProcess on pointers inside a registered block runs the kernels on the caller's memory (no staging
    copies) -- same results as through the staging path, also for a sub-range of the block, retired rows read as silence, and an
    unregistered pointer goes back to the copies."""
    import ctypes as C
    from neuralaudio_amd import capi
    lib = capi.load_library()
    m = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=True)
    S, n = 37, 128
    x = np.stack([O.signal_noise(3 * n, 900 + s) for s in range(S)])

    def run(registered):
        b = na.Batch(0)
        b.AddStreams(m, S)
        b.RemoveStreams(5, 1)
        # one block holding [in | out] back to back: the call gets pointers INSIDE it
        block = np.zeros((2, S, n), dtype=np.float32)
        if registered:
            assert lib.NA_RegisterHostBuffer(block.ctypes.data_as(C.c_void_p), block.nbytes) == 0
        outs = []
        for k in range(3):
            block[0] = x[:, k * n:(k + 1) * n]
            block[1] = 7.0  # must be overwritten (retired rows: zeroed)
            rc = lib.NA_BatchProcess(b._h, block[0].ctypes.data_as(C.POINTER(C.c_float)), block[1].ctypes.data_as(C.POINTER(C.c_float)), n)
            assert rc == 0
            outs.append(block[1].copy())
        if registered:
            assert lib.NA_UnregisterHostBuffer(block.ctypes.data_as(C.c_void_p)) == 0
            assert lib.NA_UnregisterHostBuffer(block.ctypes.data_as(C.c_void_p)) != 0  # (already gone)
        b.close()
        return np.concatenate(outs, axis=1)

    y_reg, y_copy = run(True), run(False)
    assert np.array_equal(y_reg, y_copy)
    assert not np.any(y_reg[5])
    assert O.rms(y_reg[36] - O.oracle_from_file("BossWN-standard.nam").process(x[36])) < TOL_RMS


@pytest.mark.parametrize("name,models,per", [("BossWN-standard.nam", 24, 3), ("BossWN-standard.nam", 40, 1), ("BossWN-nano.nam", 20, 5), ("BossWN-feather.nam", 12, 2), ("BossWN-a2.nam", 22, 3), ("BossLSTM-1x16.nam", 30, 2), ("BossLSTM-2x8.nam", 17, 3)])
def test_many_distinct_models_run_as_one_table_launch_and_match_one_model_batch(na, loader, name, models, per):
    """A server's batch: many DIFFERENT model handles with a few streams each (here the same file loaded `models` times: as many model
    groups).  More groups than a launch's kernarg segment holds run as ONE launch whose group table lives in device memory
    (wavenet_spec_impl.h WaveNetSpecTableKernel) -- bit for bit what one model handle with all the streams computes, over a ragged
    buffer sequence (128-frame blocks: table launches; the rest: launches of eight groups)."""
    import torch
    dev = torch.device("cuda", 0)
    handles = [loader.CreateFromFile(_path(name), doPrewarm=False) for _ in range(models)]
    ts = torch.cuda.Stream(device=dev)
    many, one = na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0, hip_stream=ts.cuda_stream)
    a2 = "a2" in name
    for i, h in enumerate(handles):
        many.AddStreams(h, per, quality=(1.0 if i % 2 else 0.2) if a2 else 1.0)  # (A2: both submodels of the container, model by model)
    for i in range(models if a2 else 1):
        one.AddStreams(handles[0], per if a2 else models * per, quality=(1.0 if i % 2 else 0.2) if a2 else 1.0)
    S = models * per
    g = torch.Generator(device="cpu").manual_seed(5)
    lengths = [128, 128, 64, 128, 37, 256]
    total = sum(lengths)
    x = torch.clamp(0.3 * torch.randn(S, total, generator=g), -1.0, 1.0).to(dev)
    want, got = torch.zeros_like(x), torch.zeros_like(x)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(ts):
        for bb, y in ((one, want), (many, got)):
            at = 0
            for n in lengths:
                bb.ProcessDevice(x[:, at:].data_ptr(), y[:, at:].data_ptr(), n, total, total)
                at += n
            bb.Synchronize()
    if "LSTM" in name and os.environ.get("NA_REC_QUAD_MIN"):
        # (forced four-streams-per-wave runs, tests/test_gpu_families.py: the one-model batch is on that layout, the table launch on the
        # one-stream layout -- another summation order)
        assert float((want - got).abs().max()) < 2e-5
    else:
        assert torch.equal(want, got)
    yo = O.oracle_from_file(name, quality=1.0).process(x[S - 1].cpu().numpy())  # (the last handle runs at quality 1.0)
    assert O.rms(got[S - 1].cpu().numpy() - yo) < (5e-6 if "LSTM" in name else TOL_RMS)
    many.close()
    one.close()


def test_table_launches_inside_a_captured_multi_unit_batch_replay_their_own_tables(na, loader):
    """More than eight WaveNet model groups AND more than eight recurrent groups in ONE batch: several launch units, so the buffer is
    captured into a hipGraph and replayed (gpu_batch.cpp ProcessDeviceOn) -- with table launches inside the capture.  The group tables
    are uploaded in front of the capture, one immutable device copy per block length (wavenet_launch.h WnLaunchTable): a 192-frame
    call is a 128- and a 64-frame table launch in the same graph, each replaying its own table.  Same call signature over and over
    (replays), then other signatures (new captures), against one-model-handle batches of the same streams, bit for bit."""
    import torch
    dev = torch.device("cuda", 0)
    wn = [loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False) for _ in range(11)]
    nano = [loader.CreateFromFile(_path("BossWN-nano.nam"), doPrewarm=False) for _ in range(10)]
    rec = [loader.CreateFromFile(_path("BossLSTM-1x16.nam"), doPrewarm=False) for _ in range(12)]
    ts = torch.cuda.Stream(device=dev)
    many, one = na.Batch(0, hip_stream=ts.cuda_stream), na.Batch(0, hip_stream=ts.cuda_stream)
    per = 2
    for h in wn + nano + rec:
        many.AddStreams(h, per)
    for hs in (wn, nano, rec):
        one.AddStreams(hs[0], per * len(hs))
    S = per * (len(wn) + len(nano) + len(rec))
    lengths = [192] * 4 + [128] * 3 + [192] * 2 + [64, 100, 128]
    total = max(lengths)
    g = torch.Generator(device="cpu").manual_seed(17)
    x = torch.clamp(0.3 * torch.randn(len(lengths), S, total, generator=g), -1.0, 1.0).to(dev)
    want, got = torch.zeros_like(x), torch.zeros_like(x)
    xin = torch.zeros(S, total, device=dev)      # one fixed pair of buffers, as a real-time host uses: the same call signature replays
    yout = torch.zeros(S, total, device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(ts):
        for bb, y in ((one, want), (many, got)):
            for k, n in enumerate(lengths):
                xin.copy_(x[k])
                bb.ProcessDevice(xin.data_ptr(), yout.data_ptr(), n, total, total)
                y[k, :, :n].copy_(yout[:, :n])
            bb.Synchronize()
    nwn = per * (len(wn) + len(nano))
    assert torch.equal(want[:, :nwn], got[:, :nwn])
    if os.environ.get("NA_REC_QUAD_MIN"):
        # (forced four-streams-per-wave runs, tests/test_gpu_families.py: the one-model batch is on that layout, the table launch on the
        # one-stream layout -- another summation order)
        assert float((want[:, nwn:] - got[:, nwn:]).abs().max()) < 2e-5
    else:
        assert torch.equal(want[:, nwn:], got[:, nwn:])
    xs = np.concatenate([x[k, 0, :n].cpu().numpy() for k, n in enumerate(lengths)])
    ys = np.concatenate([got[k, 0, :n].cpu().numpy() for k, n in enumerate(lengths)])
    assert O.rms(ys - O.oracle_from_file("BossWN-standard.nam").process(xs)) < TOL_RMS
    many.close()
    one.close()


def test_batches_beyond_the_infinity_cache_mark_their_ring_traffic_non_temporal_and_compute_the_same(na, loader):
    """More A1 Standard state than the 256 MB Infinity Cache holds (1760 streams = 428 MB, beyond the 400 MB from which the variant is used): the chains of such a batch mark the ring traffic
    of the d >= 128 layers non-temporal (Cfg::NT, wavenet_spec_impl.h) -- a cache-policy bit, so two batches of 880 streams (inside the
    cache: the ordinary variant) must compute the same bits; both against the oracle on one stream."""
    import torch
    dev = torch.device("cuda", 0)
    m = loader.CreateFromFile(_path("BossWN-standard.nam"), doPrewarm=False)
    S, n, steps = 1760, 128, 5
    big = na.Batch(0)
    big.AddStreams(m, S)
    assert big.StateBytes() > 400 * 1024 * 1024
    halves = [na.Batch(0), na.Batch(0)]
    for h in halves:
        h.AddStreams(m, S // 2)
        assert h.StateBytes() < 400 * 1024 * 1024  # (below the threshold: the ordinary variant)
    g = torch.Generator(device="cpu").manual_seed(31)
    x = torch.clamp(0.3 * torch.randn(steps, S, n, generator=g), -1.0, 1.0).to(dev)
    got, want = torch.zeros(steps, S, n, device=dev), torch.zeros(steps, S, n, device=dev)
    torch.cuda.synchronize(dev)
    for k in range(steps):
        big.ProcessDevice(x[k].data_ptr(), got[k].data_ptr(), n)
        for i, h in enumerate(halves):
            h.ProcessDevice(x[k][i * (S // 2):].data_ptr(), want[k][i * (S // 2):].data_ptr(), n)
    big.Synchronize()
    for h in halves:
        h.Synchronize()
    assert torch.equal(got, want)
    yo = O.oracle_from_file("BossWN-standard.nam").process(np.concatenate([x[k][S - 1].cpu().numpy() for k in range(steps)]))
    assert O.rms(np.concatenate([got[k][S - 1].cpu().numpy() for k in range(steps)]) - yo) < TOL_RMS
    big.close()
    for h in halves:
        h.close()
