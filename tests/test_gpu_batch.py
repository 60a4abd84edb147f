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


@pytest.mark.parametrize("layers,hidden", [(1, 16), (2, 8), (1, 12), (2, 20)])
def test_keras_gru_single_stream_shapes(na, loader, layers, hidden):
    import json
    gj = O.synth_keras_gru(layers, hidden, seed=40 + hidden + layers)
    m = loader.CreateFromString(json.dumps(gj), ".aidax", doPrewarm=True)
    x = O.signal_sine(1000)
    y = np.concatenate([m.Process(x[i:i + 100]) for i in range(0, x.size, 100)])
    assert O.rms(y - O.OracleGRU(gj).process(x)) < 5e-6


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
