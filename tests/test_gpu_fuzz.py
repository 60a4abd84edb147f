"""Randomised parity: runtime-shaped WaveNet architectures (what the reference's dynamic path accepts -- any channel count <= 16,
kernel size, dilation list, head shape) through the loader and the HIP path vs the oracle."""
import os

import numpy as np
import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu
TOL_RMS = 2e-6


@pytest.fixture(scope="module")
def na():
    import neuralaudio_amd
    return neuralaudio_amd


def _random_arrays(rng):
    kind = rng.integers(0, 3)
    if kind == 0:  # A1-style chain of two arrays, one kernel size per array, 1x1 heads
        c1 = int(rng.choice([2, 3, 4, 5, 7, 8, 11, 12, 16]))
        c2 = int(rng.choice([1, 2, 3, 4, 6, 8]))
        arrays = []
        for i, (cin, c, h, bias) in enumerate([(1, c1, c2, False), (c1, c2, 1, True)]):
            nl = int(rng.integers(1, 7))
            arrays.append(dict(input_size=cin, condition_size=1, head_size=h, head_kernel_size=1, head_dilation=1, channels=c,
                               has_head_bias=bias, activation=O.ACT_TANH, kernel_sizes=[int(rng.integers(2, 6))] * nl,
                               dilations=[int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 100, 128, 200, 256])) for _ in range(nl)]))
        return arrays
    # single array, per-layer kernel sizes, conv head (A2 style); tanh or LeakyReLU
    c = int(rng.choice([1, 2, 3, 5, 8, 13, 16]))
    nl = int(rng.integers(1, 9))
    return [dict(input_size=1, condition_size=1, head_size=1, head_kernel_size=int(rng.choice([1, 2, 7, 16])), head_dilation=1, channels=c,
                 has_head_bias=bool(rng.integers(0, 2)) or kind == 2, activation=O.ACT_LEAKYRELU if kind == 2 else O.ACT_TANH,
                 kernel_sizes=[int(rng.integers(1, 17)) for _ in range(nl)],
                 dilations=[int(rng.choice([1, 2, 3, 7, 13, 17, 41, 64, 101, 239])) for _ in range(nl)])]


@pytest.mark.parametrize("seed", range(int(os.environ.get("NA_FUZZ_ARCH_SEEDS", "64"))))  # (a longer campaign: NA_FUZZ_ARCH_SEEDS=200)
def test_random_architecture_matches_oracle(na, seed):
    rng = np.random.default_rng(1000 + seed)
    arrays = _random_arrays(rng)
    w = O.synth_wavenet_weights(arrays, seed=seed)
    loader = na.NeuralModelLoader()
    m = loader.CreateFromString(O.nam_json_wavenet_generic(arrays, w), ".nam", doPrewarm=True)
    assert m is not None, arrays
    ora = O.OracleWaveNet(arrays, w)
    assert m.GetReceptiveFieldSize() == ora.receptive_field
    x = O.signal_noise(1500, seed=seed)
    sizes, pos, out = [int(v) for v in rng.choice([1, 17, 64, 65, 128, 200, 333], size=12)], 0, []
    for n in sizes:
        n = min(n, x.size - pos)
        if n <= 0:
            break
        out.append(m.Process(x[pos:pos + n]))
        pos += n
    y = np.concatenate(out)
    err = O.rms(y - ora.process(x[:pos]))
    assert err < TOL_RMS, (arrays, err)


@pytest.mark.parametrize("hidden,layers", [(4, 1), (8, 3), (12, 2), (16, 3), (20, 1), (24, 2), (32, 1), (40, 1)])
def test_lstm_shapes_beyond_the_official_ones(na, hidden, layers):
    """NAM LSTM files of any supported shape (the reference's dynamic LSTM path, LSTMDynamic.h): every kernel family is hit --
    LDS-free DPP (hidden <= 16 padded into the 8 / 16 layouts, <= 2 layers), wave-per-stream (20..32, <= 2 layers; runtime-shaped beyond), lane-per-stream (the rest)."""
    w = O.synth_lstm_weights(layers, hidden, seed=hidden * 10 + layers)
    m = na.NeuralModelLoader().CreateFromString(O.nam_json_lstm(layers, hidden, w), ".nam", doPrewarm=True)
    assert m is not None
    x = O.signal_noise(700, seed=hidden)
    y = np.concatenate([m.Process(x[i:i + 100]) for i in range(0, x.size, 100)])
    assert O.rms(y - O.OracleLSTM.from_nam(layers, hidden, w).process(x)) < 5e-6


@pytest.mark.parametrize("seed", range(int(os.environ.get("NA_FUZZ_BATCH_SEEDS", "12"))))  # (a longer campaign: NA_FUZZ_BATCH_SEEDS=60)
def test_random_batch_operations_track_per_stream_oracles(na, seed):
    """A stateful walk over the batch API: streams of four model kinds join at random times (prewarmed or fresh), A2 streams switch
    quality mid-run, single streams are re-prewarmed, streams LEAVE and their ids / state slots are recycled by later joins (also
    inside packed virtual streams of the narrow models), buffer sizes are ragged -- every stream must keep matching its own oracle,
    which is driven through the same sequence.  (CompositeModel.h:94-100,176-181 semantics for the quality switch: the active
    submodel processes, the inactive one's state stays frozen.)"""
    import json
    import os
    rng = np.random.default_rng(500 + seed)
    loader = na.NeuralModelLoader()
    mdir = O.MODELS_DIR
    gj = O.synth_keras_gru(1, 8, seed=77)
    models = {
        "feather": loader.CreateFromFile(os.path.join(mdir, "BossWN-feather.nam"), doPrewarm=False),
        "nano": loader.CreateFromFile(os.path.join(mdir, "BossWN-nano.nam"), doPrewarm=False),
        "a2": loader.CreateFromFile(os.path.join(mdir, "BossWN-a2.nam"), doPrewarm=False),
        "lstm": loader.CreateFromFile(os.path.join(mdir, "BossLSTM-2x8.nam"), doPrewarm=False),
        "gru": loader.CreateFromString(json.dumps(gj), ".json", doPrewarm=False),
    }
    a2json = O.load_json("BossWN-a2.nam")

    class Ref:  # one reference-side stream: a set of oracle submodels + the active index
        def __init__(self, kind, quality, prewarm):
            self.kind = kind
            if kind == "a2":
                self.subs = [O.oracle_from_file("BossWN-a2.nam", quality=q, prewarm=prewarm) for q in (0.0, 1.0)]  # ch3, ch8 (LoadAll)
                self.active = O.quality_to_submodel(a2json, quality)
            elif kind == "gru":
                self.subs, self.active = [O.OracleGRU(gj, prewarm=prewarm)], 0
            else:
                name = {"feather": "BossWN-feather.nam", "nano": "BossWN-nano.nam", "lstm": "BossLSTM-2x8.nam"}[kind]
                self.subs, self.active = [O.oracle_from_file(name, prewarm=prewarm)], 0

        def process(self, x):
            return self.subs[self.active].process(x)

    b = na.Batch(0)
    refs = []  # row -> Ref, None for a retired id
    worst = 0.0
    for step in range(18):
        live = [i for i, r in enumerate(refs) if r is not None]
        op = rng.integers(0, 5) if live else 0
        if op == 0 or len(live) < 3:  # add 1-3 streams of a random kind: retired ids are recycled first, new rows are appended otherwise
            kind = str(rng.choice(list(models)))
            q = float(rng.choice([0.0, 0.3, 0.5, 0.51, 1.0]))
            pre = bool(rng.integers(0, 2))
            count = int(rng.integers(1, 4))
            holes = [i for i, r in enumerate(refs) if r is None]
            runs = [h for h in holes if all((h + k) in holes for k in range(count))]
            expect = runs[0] if runs else len(refs)
            first = b.AddStreams(models[kind], count, quality=q, doPrewarm=pre)
            assert first == expect, (first, expect, holes, count)
            for k in range(count):
                if first + k < len(refs):
                    refs[first + k] = Ref(kind, q, pre)
                else:
                    refs.append(Ref(kind, q, pre))
        elif op == 1:  # quality switch on a random A2 stream
            idx = [i for i in live if refs[i].kind == "a2"]
            if idx:
                i = int(rng.choice(idx))
                q = float(rng.choice([0.0, 0.5, 0.75, 1.0]))
                b.SetQuality(i, q)
                refs[i].active = O.quality_to_submodel(a2json, q)
                assert b.GetActiveSubModel(i) == refs[i].active
        elif op == 2:  # re-prewarm one stream (all of its submodels, like NeuralModel::Prewarm on a LoadAll composite)
            i = int(rng.choice(live))
            b.Prewarm(i)
            for sub in refs[i].subs:
                sub.prewarm()
        elif op == 3 and len(live) > 3:  # a stream leaves (NeuralAudioCApi.cpp:38-42 DeleteModel): its id is retired, its state slots are recycled
            i = int(rng.choice(live))
            b.RemoveStreams(i, 1)
            refs[i] = None
            while refs and refs[-1] is None:  # trailing retired rows leave the arrays
                refs.pop()
            assert b.NumStreams() == len(refs) and b.NumLiveStreams() == sum(r is not None for r in refs)
            assert i >= len(refs) or not b.IsLive(i)
        n = int(rng.choice([1, 31, 64, 100, 128, 129, 300]))
        x = np.stack([O.signal_noise(n, 10000 * seed + 100 * step + s) for s in range(len(refs))])
        y = b.Process(x)
        for s, r in enumerate(refs):
            if r is None:
                assert not np.any(y[s]), (step, s)  # a retired row reads as silence
                continue
            err = O.rms(y[s] - r.process(x[s]))
            worst = max(worst, err)
            assert err < 5e-6, (step, s, r.kind, err)
    assert sum(r is not None for r in refs) >= 3
