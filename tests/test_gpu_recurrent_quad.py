"""GPU tests of the four-streams-per-wave recurrent kernel (RecurrentQuadKernel: one-layer LSTMs of up to 16 units in launches of thousands
of streams).
Parity against the oracle with the tolerance of test_gpu_parity.py (5e-6 RMS); against the one-stream-per-wave kernel (which adds the
terms of a gate row in another order) to rounding."""
import numpy as np
import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def na():
    import neuralaudio_amd
    if neuralaudio_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the product path has no CPU fallback")
    return neuralaudio_amd


@pytest.fixture()
def quad(na):
    """sets the stream count from which the four-streams-per-wave layout is used; restores the default afterwards"""
    from neuralaudio_amd import capi
    lib = capi.load_library()
    before = lib.NA_DebugSetRecurrentQuadMin(3072)
    yield lib
    lib.NA_DebugSetRecurrentQuadMin(before)


def _inputs(S, N):
    base = np.stack([O.signal_noise(N, 40 + k) for k in range(7)])
    gain = (0.1 + 0.9 * ((np.arange(S) * 37) % 11) / 10.0).astype(np.float32)
    return (base[np.arange(S) % 7] * gain[:, None]).astype(np.float32)


@pytest.mark.parametrize("layers,hidden,std", [(1, 16, False), (1, 16, True), (1, 12, False), (1, 8, False), (1, 5, True), (1, 1, False)])
def test_quad_kernel_matches_oracle_and_the_one_stream_kernel(na, quad, layers, hidden, std):
    ld = na.NeuralModelLoader()
    if std:
        ld.SetLSTMMathMode(na.EMathMode.StdMath)
    w = O.synth_lstm_weights(layers, hidden, seed=700 + 10 * hidden + layers)
    m = ld.CreateFromString(O.nam_json_lstm(layers, hidden, w), ".nam", doPrewarm=True)
    S, sizes = 3075, [128, 37, 128, 3, 300]  # not a multiple of four streams; ragged blocks, one above the 128-sample chunk
    x = _inputs(S, sum(sizes))

    def run(min_streams):
        quad.NA_DebugSetRecurrentQuadMin(min_streams)
        b = na.Batch(0)
        b.AddStreams(m, S)
        out, pos = [], 0
        for n in sizes:
            out.append(b.Process(np.ascontiguousarray(x[:, pos:pos + n])))
            pos += n
        return np.concatenate(out, axis=1)

    before = quad.NA_DebugRecurrentQuadLaunches()
    yq = run(3072)
    assert quad.NA_DebugRecurrentQuadLaunches() - before >= len(sizes)  # the kernel under test really ran
    mid = quad.NA_DebugRecurrentQuadLaunches()
    y1 = run(0)
    assert quad.NA_DebugRecurrentQuadLaunches() == mid
    assert np.max(np.abs(yq - y1)) < 3e-6  # the two lane layouts add the terms of a row in different orders
    for s in (0, 1, 2, 3, 1500, S - 3, S - 2, S - 1):
        want = O.OracleLSTM.from_nam(layers, hidden, w, math_mode=O.MATH_STD if std else O.MATH_FAST).process(x[s])
        assert O.rms(yq[s] - want) < 5e-6, (s, O.rms(yq[s] - want))


def test_quad_kernel_with_two_models_removed_streams_and_recycled_slots(na, quad):
    """Several groups in one launch (index lists instead of contiguous ranges after streams left), a group of fewer than four streams."""
    ld = na.NeuralModelLoader()
    wa, wb = O.synth_lstm_weights(1, 16, seed=11), O.synth_lstm_weights(1, 12, seed=12)
    ma = ld.CreateFromString(O.nam_json_lstm(1, 16, wa), ".nam", doPrewarm=True)
    mb = ld.CreateFromString(O.nam_json_lstm(1, 12, wb), ".nam", doPrewarm=True)
    quad.NA_DebugSetRecurrentQuadMin(64)
    b = na.Batch(0)
    ia = b.AddStreams(ma, 70)
    ib = b.AddStreams(mb, 3)
    n, blocks = 96, 3
    x = _inputs(73, n * blocks)
    before = quad.NA_DebugRecurrentQuadLaunches()
    y0 = b.Process(np.ascontiguousarray(x[:, :n]))
    assert quad.NA_DebugRecurrentQuadLaunches() > before
    gone = [ia + 5, ia + 6, ia + 40, ib + 1]
    b.RemoveStreams(ia + 5, 2)
    b.RemoveStreams(ia + 40, 1)
    b.RemoveStreams(ib + 1, 1)
    y1 = b.Process(np.ascontiguousarray(x[:, n:2 * n]))
    back = b.AddStreams(ma, 2)  # recycled ids and state slots: fresh state
    assert back == ia + 5
    y2 = b.Process(np.ascontiguousarray(x[:, 2 * n:]))
    oa = lambda: O.OracleLSTM.from_nam(1, 16, wa)
    ob = lambda: O.OracleLSTM.from_nam(1, 12, wb)
    for s in range(73):
        o = oa() if s < 70 else ob()
        want = o.process(x[s])
        assert O.rms(y0[s] - want[:n]) < 5e-6
        if s in gone:
            continue
        assert O.rms(y1[s] - want[n:2 * n]) < 5e-6, s
        assert O.rms(y2[s] - want[2 * n:]) < 5e-6, s
    for k in range(2):  # the re-added streams took the ids of the first two leavers and start from a fresh (prewarmed) state
        s = back + k
        want = oa().process(x[s, 2 * n:])
        assert O.rms(y2[s] - want) < 5e-6, s


def test_two_layer_models_stay_on_the_one_stream_layout(na, quad):
    ld = na.NeuralModelLoader()
    m = ld.CreateFromFile(O.os.path.join(O.MODELS_DIR, "BossLSTM-2x8.nam"), doPrewarm=True)
    quad.NA_DebugSetRecurrentQuadMin(1)
    b = na.Batch(0)
    b.AddStreams(m, 9)
    x = _inputs(9, 64)
    before = quad.NA_DebugRecurrentQuadLaunches()
    y = b.Process(x)
    assert quad.NA_DebugRecurrentQuadLaunches() == before
    assert O.rms(y[8] - O.oracle_from_file("BossLSTM-2x8.nam").process(x[8])) < 5e-6


@pytest.mark.parametrize("hidden", [16, 12, 7, 1])
def test_quad_kernel_runs_keras_gru_layers(na, quad, hidden):
    """One keras GRU layer + dense(1) head (RTNeural semantics: parity unpinned) on the four-streams-per-wave layout."""
    import json
    gj = O.synth_keras_gru(1, hidden, seed=40 + hidden)
    m = na.NeuralModelLoader().CreateFromString(json.dumps(gj), ".json", doPrewarm=True)
    S, sizes = 3077, [128, 61, 128, 2, 200]
    x = _inputs(S, sum(sizes))

    def run(min_streams):
        quad.NA_DebugSetRecurrentQuadMin(min_streams)
        b = na.Batch(0)
        b.AddStreams(m, S)
        out, pos = [], 0
        for n in sizes:
            out.append(b.Process(np.ascontiguousarray(x[:, pos:pos + n])))
            pos += n
        return np.concatenate(out, axis=1)

    before = quad.NA_DebugRecurrentQuadLaunches()
    yq = run(3072)
    assert quad.NA_DebugRecurrentQuadLaunches() - before >= len(sizes)
    y1 = run(0)
    assert np.max(np.abs(yq - y1)) < 3e-6
    for s in (0, 3, 1700, S - 2, S - 1):
        assert O.rms(yq[s] - O.OracleGRU(gj).process(x[s])) < 5e-6, s


@pytest.mark.parametrize("hidden,std", [(16, False), (12, False), (16, True), (9, False)])
def test_two_layer_lstm_on_one_wave_per_layer_is_bit_identical_to_the_one_wave_body(na, quad, hidden, std):
    """LSTM 2x16 (BASELINE config 4; 2x12 is the reference's other static two-layer shape on this layout): launches whose second waves find
    half-empty SIMDs run TWO waves per stream, one per layer, a few samples apart through LDS (recurrent_dpp_kernels.hip LstmDppPipeBody,
    UsePipe: 2 S waves an odd number of times the 1024 SIMDs); the others keep one wave per stream (LstmDppBodyM).  Same lanes, same
    weights, same order of operations: the two must agree bit for bit -- the same streams inside a batch of 1900 (3800 waves would be
    the fourth wave of most SIMDs: one wave each; below the four-streams-per-wave threshold) and in a batch of 75 (pipelined), over
    ragged block lengths (a tail that is not a multiple of four, a block above the 128-sample chunk) -- and match the oracle."""
    ld = na.NeuralModelLoader()
    if std:
        ld.SetLSTMMathMode(na.EMathMode.StdMath)
    w = O.synth_lstm_weights(2, hidden, seed=900 + hidden)
    m = ld.CreateFromString(O.nam_json_lstm(2, hidden, w), ".nam", doPrewarm=True)
    sizes = [128, 37, 128, 3, 300, 64]
    S_small, S_big = 75, 1900
    x = _inputs(S_big, sum(sizes))
    quad.NA_DebugSetRecurrentQuadMin(0)  # (never the four-streams-per-wave kernel here)

    def run(S):
        b = na.Batch(0)
        b.AddStreams(m, S)
        out, pos = [], 0
        for n in sizes:
            out.append(b.Process(np.ascontiguousarray(x[:S, pos:pos + n])))
            pos += n
        b.close()
        return np.concatenate(out, axis=1)

    try:
        small, big = run(S_small), run(S_big)
    finally:
        quad.NA_DebugSetRecurrentQuadMin(3072)
    assert np.array_equal(small, big[:S_small])
    for s in (0, 1, 63, 64, S_small - 1):
        want = O.OracleLSTM.from_nam(2, hidden, w, math_mode=O.MATH_STD if std else O.MATH_FAST).process(x[s])
        assert O.rms(small[s] - want) < 5e-6, (s, O.rms(small[s] - want))


def test_many_two_layer_lstm_models_in_one_table_launch_use_the_layer_pipeline_too(na, quad):
    """More model groups than a launch's kernarg segment holds run as one table launch (RecurrentDppTableKernel); with two-layer 16-unit
    LSTMs among them that launch, too, is four-wave workgroups with one wave per layer -- 11 handles of the same LSTM 2x16 (3 streams each)
    next to 10 handles of an LSTM 1x16 (one wave per stream inside the same workgroups), against one handle each with all the streams:
    bit for bit, ragged blocks."""
    ld = na.NeuralModelLoader()
    w2, w1 = O.synth_lstm_weights(2, 16, seed=77), O.synth_lstm_weights(1, 16, seed=78)
    two = [ld.CreateFromString(O.nam_json_lstm(2, 16, w2), ".nam", doPrewarm=True) for _ in range(11)]
    one = [ld.CreateFromString(O.nam_json_lstm(1, 16, w1), ".nam", doPrewarm=True) for _ in range(10)]
    per = 3
    S = per * (len(two) + len(one))
    sizes = [128, 50, 128, 7]
    x = _inputs(S, sum(sizes))
    quad.NA_DebugSetRecurrentQuadMin(0)

    def run(many):
        b = na.Batch(0)
        if many:
            for h in two + one:
                b.AddStreams(h, per)
        else:
            b.AddStreams(two[0], per * len(two))
            b.AddStreams(one[0], per * len(one))
        out, pos = [], 0
        for n in sizes:
            out.append(b.Process(np.ascontiguousarray(x[:, pos:pos + n])))
            pos += n
        b.close()
        return np.concatenate(out, axis=1)

    try:
        ym, y1 = run(True), run(False)
    finally:
        quad.NA_DebugSetRecurrentQuadMin(3072)
    assert np.array_equal(ym, y1)
    assert O.rms(ym[1] - O.OracleLSTM.from_nam(2, 16, w2).process(x[1])) < 5e-6
    assert O.rms(ym[S - 1] - O.OracleLSTM.from_nam(1, 16, w1).process(x[S - 1])) < 5e-6


def test_four_streams_per_wave_kernel_is_right_beside_matrix_waves_of_another_kernel(na, quad):
    """Round 6 found the four-streams-per-wave kernel occasionally wrong in the fourth stream of a wave (lanes 48 .. 63, from some sample
    of a buffer on) whenever f16-split WaveNet launches shared the chip with it -- as another unit of the same batch or as another batch.
    Cause (tools/microbench/pk_lds_opsel.hip, profiles/r06_quad_race.txt): v_pk_fma_f32 with a non-default op_sel on a register that a
    ds_read has just delivered goes wrong in the last sixteen lanes while another wave issues f16 MFMAs on the SIMD.  The kernel's h
    entries therefore hold {h, h} pairs and the packed row sums read them as delivered.  Here: 64 LSTM (and GRU) streams on that
    kernel as one unit of a mixed batch beside 22 A1 Standard streams, 150 buffers, against the same streams on the one-stream kernel:
    rounding apart (two lane layouts), never the 1e-3 .. 1e-1 the fault produced in 10 - 40 of 400 buffers."""
    import os
    import torch
    ld = na.NeuralModelLoader()
    std = ld.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
    dev = torch.device("cuda", 0)
    nstd, nl, n, steps = 22, 64, 128, 150
    for name in ("BossLSTM-1x16.nam", "synthetic_gru_1x16.json"):
        rec = ld.CreateFromFile(os.path.join(O.MODELS_DIR, name), doPrewarm=False)
        batches = []
        for q in (1, 0):
            quad.NA_DebugSetRecurrentQuadMin(q)
            ts = torch.cuda.Stream(device=dev)
            b = na.Batch(0, hip_stream=ts.cuda_stream)
            b.AddStreams(std, nstd)
            b.AddStreams(rec, nl)
            batches.append((q, ts, b))
        g = torch.Generator(device="cpu").manual_seed(3)
        launches = quad.NA_DebugRecurrentQuadLaunches()
        worst = 0.0
        for k in range(steps):
            x = torch.clamp(0.3 * torch.randn(nstd + nl, n, generator=g), -1, 1).to(dev)
            ys = [torch.zeros(nstd + nl, n, device=dev) for _ in batches]
            torch.cuda.synchronize(dev)
            for (q, ts, b), y in zip(batches, ys):
                quad.NA_DebugSetRecurrentQuadMin(q)
                with torch.cuda.stream(ts):
                    b.ProcessDevice(x.data_ptr(), y.data_ptr(), n, n, n)
            torch.cuda.synchronize(dev)
            assert torch.equal(ys[0][:nstd], ys[1][:nstd]), (name, k)
            worst = max(worst, float((ys[0][nstd:] - ys[1][nstd:]).abs().max()))
        # the kernel under test ran as a unit of the mixed batch (the counter counts host-side launches: the captured graph's, not its replays)
        assert quad.NA_DebugRecurrentQuadLaunches() - launches >= 1
        assert worst < 2e-5, (name, worst)
        for _, _, b in batches:
            b.close()
