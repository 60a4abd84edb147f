"""CPU tests of the oracle (the parity checker) itself: what pins it, and its structural properties.

Pins (see oracle/na_oracle.h header):
  * keras LSTM known-answer vector shipped inside the reference's own sample model  -> test_keras_kat_*
  * the reference's MatMul.h outputs (tests/golden/matmul_ref.npz, made by oracle/_ref) -> test_tiny_matmuls_*
  * weight inventory of every reference sample model                                -> test_weight_counts_*
Self-consistency (not reference pins): independent float64 restatement, chunk invariance, prewarm == zero lead-in.
"""
import ctypes as C
import os

import numpy as np
import pytest

import na_oracle as O
import ref_np as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

WAVENET_FILES = [("BossWN-standard.nam", 1.0), ("BossWN-feather.nam", 1.0), ("BossWN-nano.nam", 1.0),
                 ("BossWN-a2.nam", 0.0), ("BossWN-a2.nam", 1.0)]


def _wavenet_json(name, q):
    j = O.load_json(name)
    if j["architecture"] == "SlimmableContainer":
        j = j["config"]["submodels"][O.quality_to_submodel(j, q)]["model"]
    return j


# ------------------------------------------------------------------------------------------ reference pins

def test_keras_kat_exact_math_pins_lstm_layout():
    """tw40 json input_batch -> output_batch: exact tanh/sigmoid, zero state, no prewarm."""
    j = O.load_json("tw40_blues_deluxe_deerinkstudios.json")
    x = np.array(j["input_batch"], dtype=np.float32).ravel()
    y_ref = np.array(j["output_batch"], dtype=np.float32).ravel()
    assert x.size == 2048 and y_ref.size == 2048
    y = O.OracleLSTM.from_keras(j, math_mode=O.MATH_STD, prewarm=False).process(x)
    assert O.rms(y - y_ref) < 1e-6
    # the default FastMath policy is a different function: the KAT must NOT be matched by it (4.8e-3 RMS)
    y_fast = O.OracleLSTM.from_keras(j, math_mode=O.MATH_FAST, prewarm=False).process(x)
    assert 1e-3 < O.rms(y_fast - y_ref) < 2e-2


def test_keras_kat_float64_restatement_agrees():
    j = O.load_json("tw40_blues_deluxe_deerinkstudios.json")
    x = np.array(j["input_batch"], dtype=np.float32).ravel()
    y_ref = np.array(j["output_batch"], dtype=np.float32).ravel()
    y = R.lstm_forward_keras(j, x, prewarm=0, tanh=np.tanh, sigmoid=R.std_sigmoid)
    assert O.rms(y - y_ref) < 1e-6


@pytest.mark.parametrize("cin,cout", [(3, 3), (8, 1), (3, 1), (1, 3)])
def test_tiny_matmuls_match_reference_matmul_h(cin, cout):
    """oracle dense / conv-tap arithmetic vs the reference's own MatMul.h outputs (committed vectors)."""
    g = np.load(os.path.join(GOLDEN, "matmul_ref.npz"))
    tag = "%d_%d" % (cin, cout)
    x, w, w2, init = g["x_" + tag], g["w_" + tag], g["w2_" + tag], g["init_" + tag]
    y0 = O.test_dense(w, None, x)
    assert np.max(np.abs(y0 - g["zero_" + tag])) < 1e-6
    if (cin, cout) != (1, 3):  # MatMul.h:169: the (1,3) InitColwise variant reads inData instead of initData (never instantiated)
        y1 = O.test_dense(w, init, x)
        assert np.max(np.abs(y1 - g["colwise_" + tag])) < 1e-6
    y2 = O.test_dense(w2, None, x, out=g["colwise_" + tag])
    assert np.max(np.abs(y2 - g["acc_" + tag])) < 1e-6


def test_tiny_matmuls_against_live_reference_build_if_present():
    """Where oracle/_ref was built (this container), call the reference code directly on fresh inputs."""
    so = os.path.join(O.ORACLE_DIR, "_ref", "libna_ref_matmul.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built here")
    ref = C.CDLL(so)
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, size=(64, 3)).astype(np.float32)
    w = rng.uniform(-1, 1, size=(9,)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(3,)).astype(np.float32)
    y = np.zeros((64, 3), np.float32)
    fp = C.POINTER(C.c_float)
    ref.na_ref_matmul_init_colwise_3_3(x.ctypes.data_as(fp), y.ctypes.data_as(fp), w.ctypes.data_as(fp), b.ctypes.data_as(fp),
                                       C.c_size_t(64))
    assert np.max(np.abs(O.test_dense(w, b, x) - y)) < 1e-6


@pytest.mark.parametrize("name,q,count", [("BossWN-standard.nam", 1.0, 13802), ("BossWN-feather.nam", 1.0, 3026),
                                          ("BossWN-nano.nam", 1.0, 842), ("BossWN-a2.nam", 0.0, 1871),
                                          ("BossWN-a2.nam", 1.0, 12146)])
def test_weight_counts_match_reference_sample_files(name, q, count):
    """WaveNet.h:704-709: the architecture consumes exactly weights.size() floats."""
    j = _wavenet_json(name, q)
    arrays = O.wavenet_arrays_from_nam(j)
    assert len(j["weights"]) == count
    assert O.wavenet_num_weights(arrays) == count
    with pytest.raises(ValueError):
        O.OracleWaveNet(arrays, j["weights"][:-1])


def test_official_architecture_tables():
    assert O.wavenet_num_weights(O.a1_arrays(16, 8)) == 13802
    assert O.wavenet_num_weights(O.a1_arrays(12, 6)) == 6554   # Lite (no sample file)
    assert O.wavenet_num_weights(O.a1_arrays(8, 4)) == 3026
    assert O.wavenet_num_weights(O.a1_arrays(4, 2)) == 842
    assert O.wavenet_num_weights(O.a2_arrays(3)) == 1871
    assert O.wavenet_num_weights(O.a2_arrays(8)) == 12146


@pytest.mark.parametrize("name,layers,hidden,count", [("BossLSTM-1x16.nam", 1, 16, 1201), ("BossLSTM-2x8.nam", 2, 8, 905)])
def test_lstm_weight_counts(name, layers, hidden, count):
    j = O.load_json(name)
    assert len(j["weights"]) == count
    assert j["config"]["num_layers"] == layers and j["config"]["hidden_size"] == hidden
    O.OracleLSTM.from_nam(layers, hidden, j["weights"])
    with pytest.raises(ValueError):
        O.OracleLSTM.from_nam(layers, hidden, j["weights"][:-1])


# ------------------------------------------------------------------------------------------ math policy

def test_fast_tanh_formula():
    """Activation.h:83-96 evaluated in float64 vs the C float implementation."""
    xs = np.linspace(-6, 6, 2001).astype(np.float32)
    got = np.array([O.lib().na_oracle_fast_tanh(float(x)) for x in xs])
    want = R.fast_tanh(xs.astype(np.float64))
    assert np.max(np.abs(got - want)) < 3e-7
    assert O.lib().na_oracle_fast_tanh(0.0) == 0.0
    sig = np.array([O.lib().na_oracle_fast_sigmoid(float(x)) for x in xs])
    assert np.max(np.abs(sig - R.fast_sigmoid(xs.astype(np.float64)))) < 3e-7
    # rational approximation, not libm: visibly different from tanh
    assert 1e-4 < np.max(np.abs(want - np.tanh(xs.astype(np.float64)))) < 5e-3
    assert O.lib().na_oracle_leaky_relu(-2.0) == pytest.approx(-0.02)
    assert O.lib().na_oracle_leaky_relu(3.0) == 3.0


# ------------------------------------------------------------------------------------------ self-consistency

@pytest.mark.parametrize("name,q", WAVENET_FILES)
def test_wavenet_oracle_vs_independent_float64(name, q):
    j = _wavenet_json(name, q)
    arrays = O.wavenet_arrays_from_nam(j)
    m = O.OracleWaveNet(arrays, j["weights"])
    x = O.signal_sine(3000)
    y = m.process(x)
    yr, rf = R.wavenet_forward(arrays, j["weights"], x)
    assert rf == m.receptive_field
    assert O.rms(y) > 0.05
    assert O.rms(y - yr) < 1e-6


def test_wavenet_synthetic_lite_and_noise_input():
    arrays = O.a1_arrays(12, 6)
    w = O.synth_wavenet_weights(arrays, seed=12)
    x = O.signal_noise(2500, seed=5)
    y = O.OracleWaveNet(arrays, w).process(x)
    yr, _ = R.wavenet_forward(arrays, w, x)
    assert O.rms(y - yr) < 1e-6


@pytest.mark.parametrize("name,q", WAVENET_FILES)
def test_chunk_size_invariance_is_bit_exact(name, q):
    """InternalModel.h:104-117 chunks at 64; results must not depend on the chunking."""
    j = _wavenet_json(name, q)
    arrays = O.wavenet_arrays_from_nam(j)
    x = O.signal_sine(1500)
    y64 = O.OracleWaveNet(arrays, j["weights"]).process(x)
    m = O.OracleWaveNet(arrays, j["weights"])
    m.set_max_frames(37)
    assert np.array_equal(y64, m.process(x))
    m1 = O.OracleWaveNet(arrays, j["weights"])
    m1.set_max_frames(1)
    assert np.array_equal(y64, m1.process(x))


def test_prewarm_equals_long_zero_lead_in():
    """WaveNet.h:746-766: the analytic prewarm is the steady state of zero input."""
    j = _wavenet_json("BossWN-feather.nam", 1.0)
    arrays = O.wavenet_arrays_from_nam(j)
    x = O.signal_sine(700)
    warm = O.OracleWaveNet(arrays, j["weights"], prewarm=True).process(x)
    cold = O.OracleWaveNet(arrays, j["weights"], prewarm=False)
    cold.process(np.zeros(4092 + 64, np.float32))
    assert np.max(np.abs(warm - cold.process(x))) < 1e-6


@pytest.mark.parametrize("name", ["BossLSTM-1x16.nam", "BossLSTM-2x8.nam"])
def test_lstm_oracle_vs_independent_float64(name):
    j = O.load_json(name)
    c = j["config"]
    x = O.signal_sine(1024)
    y = O.OracleLSTM.from_nam(c["num_layers"], c["hidden_size"], j["weights"]).process(x)
    yr = R.lstm_forward_nam(c["num_layers"], c["hidden_size"], j["weights"], x)
    assert O.rms(y - yr) < 5e-6


def test_lstm_synthetic_2x16():
    w = O.synth_lstm_weights(2, 16, seed=3)
    x = O.signal_noise(600, seed=9)
    y = O.OracleLSTM.from_nam(2, 16, w).process(x)
    yr = R.lstm_forward_nam(2, 16, w, x)
    assert O.rms(y - yr) < 5e-6


def test_quality_to_submodel_rule():
    """CompositeModel.h:200-213: first sorted level with q <= max_value, else last."""
    j = O.load_json("BossWN-a2.nam")
    mv = [s["max_value"] for s in j["config"]["submodels"]]
    assert sorted(mv) == [0.5, 1.0]
    lo, hi = mv.index(0.5), mv.index(1.0)
    assert O.quality_to_submodel(j, 0.0) == lo
    assert O.quality_to_submodel(j, 0.5) == lo
    assert O.quality_to_submodel(j, 0.5001) == hi
    assert O.quality_to_submodel(j, 1.0) == hi
    assert O.quality_to_submodel(j, 7.0) == hi


def test_oracle_regression_vectors():
    """Committed oracle outputs reproduce bit-for-bit on this machine (portable -ffp-contract=off build)."""
    g = np.load(os.path.join(GOLDEN, "oracle_outputs.npz"))
    x = g["input"]
    for key in g.files:
        if key == "input":
            continue
        name, q = key.split("@q")
        y = O.oracle_from_file(name, quality=float(q)).process(x)
        if name.endswith(".json") or "LSTM" in name:
            assert np.max(np.abs(y - g[key])) < 1e-6, key
        else:
            assert np.array_equal(y, g[key]), key


# ----------------------------------------------------------------------------- keras GRU (RTNeural arithmetic; parity unpinned)

@pytest.mark.parametrize("layers,hidden", [(1, 16), (2, 8), (1, 12)])
def test_gru_oracle_matches_float64_restatement_and_torch(layers, hidden):
    import torch
    j = O.synth_keras_gru(layers, hidden, seed=100 + 10 * layers + hidden)
    x = O.signal_noise(1024, seed=7)
    yo = O.OracleGRU(j, prewarm=True).process(x)
    yr = R.gru_forward_keras(j, x, prewarm=2048)
    assert O.rms(yo - yr) < 1e-6

    # independent implementation: torch.nn.GRU is the same reset-after cell with gate order (r, z, n) instead of keras' (z, r, c)
    H = hidden
    gru = torch.nn.GRU(1, H, num_layers=layers, batch_first=True).double()
    perm = np.concatenate([np.arange(H, 2 * H), np.arange(0, H), np.arange(2 * H, 3 * H)])
    with torch.no_grad():
        for l in range(layers):
            k, u, b = (np.array(j["layers"][l]["weights"][i], dtype=np.float64) for i in range(3))
            getattr(gru, "weight_ih_l%d" % l).copy_(torch.from_numpy(k.T[perm].copy()))
            getattr(gru, "weight_hh_l%d" % l).copy_(torch.from_numpy(u.T[perm].copy()))
            getattr(gru, "bias_ih_l%d" % l).copy_(torch.from_numpy(b[0][perm].copy()))
            getattr(gru, "bias_hh_l%d" % l).copy_(torch.from_numpy(b[1][perm].copy()))
        xs = torch.from_numpy(np.concatenate([np.zeros(2048), x.astype(np.float64)])).reshape(1, -1, 1)
        hs, _ = gru(xs)
        wh = torch.from_numpy(np.array(j["layers"][-1]["weights"][0], dtype=np.float64).ravel())
        yt = (hs[0] @ wh + float(j["layers"][-1]["weights"][1][0])).numpy()[2048:]
    assert O.rms(yo - yt) < 1e-6


def test_gru_oracle_chunking_and_prewarm():
    j = O.synth_keras_gru(1, 16, seed=5)
    x = O.signal_sine(700)
    a = O.OracleGRU(j, prewarm=True).process(x)
    m = O.OracleGRU(j, prewarm=True)
    b = np.concatenate([m.process(x[i:i + 37]) for i in range(0, x.size, 37)])
    assert np.array_equal(a, b)
    c = O.OracleGRU(j, prewarm=False)
    c.process(np.zeros(2048, dtype=np.float32))
    assert np.array_equal(a, c.process(x))


def test_gru_oracle_matches_committed_torch_vectors():
    """tests/golden/gru_torch.npz: torch.nn.GRU output for the committed synthetic keras GRU model (make_golden.py)."""
    g = np.load(os.path.join(GOLDEN, "gru_torch.npz"))
    y = O.oracle_from_file("synthetic_gru_1x16.json").process(g["input"])
    assert O.rms(y - g["output"]) < 1e-6


def test_oracle_matches_committed_fixture_matrix():
    """tests/golden/fixture_matrix.npz (SURVEY 8c matrix: 7 models x {sine, noise, zeros} x 4096 samples): the live-built oracle
    reproduces its own committed outputs bit-for-bit-ish (compiler flags may move the last ulp) and stays within f32 noise of the
    committed float64 restatement -- on the GPU box too, where /root/reference and the generating script's inputs do not exist."""
    from golden.make_golden import MATRIX_MODELS
    g = np.load(os.path.join(GOLDEN, "fixture_matrix.npz"))
    for tag, name, q in MATRIX_MODELS:
        for k in ("sine", "noise", "zeros"):
            y = O.oracle_from_file(name, quality=q).process(g["input/" + k])
            assert O.rms(y - g["oracle/%s/%s" % (tag, k)]) < 2e-7, (tag, k)
            assert O.rms(y - g["np64/%s/%s" % (tag, k)]) < (5e-6 if tag.startswith("lstm") else 1e-6), (tag, k)


def test_generic_keras_stack_restatement_matches_committed_torch_vectors():
    """tests/ref_np.keras_stack_forward (the checker of the generic keras stacks) against an independent implementation:
    torch.nn.LSTM / GRU / Linear outputs committed by tests/golden/make_golden.py (RTNeural, which evaluates these in the reference, is
    an absent submodule: parity unpinned)."""
    import json
    import ref_np as R
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = np.load(os.path.join(here, "keras_stacks_torch.npz"))
    names = [k for k in d.files if k != "input"]
    assert len(names) == 5  # (three lstm / gru / dense stacks, two with conv1d layers and softmax)
    for name in names:
        with open(os.path.join(here, "models", "synthetic_stack_%s.json" % name)) as f:
            j = json.load(f)
        y = R.keras_stack_forward(j, d["input"])
        assert O.rms(y - d[name]) < 1e-6, (name, O.rms(y - d[name]))


@pytest.mark.parametrize("name,q", [("BossWN-standard.nam", 1.0), ("BossWN-feather.nam", 1.0), ("BossWN-nano.nam", 1.0), ("BossWN-a2.nam", 1.0), ("BossWN-a2.nam", 0.0)])
def test_vectorised_bench_variant_matches_the_scalar_oracle(name, q):
    """oracle/na_oracle_simd.c (frames as the vector axis, 8 x 8 register tiles: what bench.py's cpu_baseline times as `port_simd`) computes
    what the scalar restatement computes -- same values, another summation order: <= 1e-6 RMS on every official WaveNet architecture,
    sine and noise, from the prewarmed state."""
    j = O.load_json(name)
    if j["architecture"] == "SlimmableContainer":
        j = j["config"]["submodels"][O.quality_to_submodel(j, q)]["model"]
    arrays = O.wavenet_arrays_from_nam(j)
    for x in (O.signal_sine(6400), O.signal_noise(6400, 11)):
        ref = O.OracleWaveNet(arrays, j["weights"]).process(x)
        got = O.OracleWaveNetSimd(arrays, j["weights"]).process(x)
        assert O.rms(ref) > 1e-3 and O.rms(got - ref) < 1e-6, (name, q, O.rms(got - ref))
    with pytest.raises(ValueError):
        O.OracleWaveNetSimd(arrays, j["weights"]).process(np.zeros(13, np.float32))

