"""The HIP path against COMMITTED vectors (tests/golden/fixture_matrix.npz, written by tests/golden/make_golden.py): the SURVEY 8(c)
matrix -- 7 models x {sin(0.01 n), clipped noise, zeros} x 4096 samples after prewarm -- in two independent evaluations (float64 numpy
restatement and the C oracle), plus one A2 stream with mid-stream quality switches.  Unlike tests/test_gpu_parity.py nothing here
is computed by the live-built oracle, so kernel and oracle cannot drift together unnoticed.  Tolerances as in test_gpu_parity.py:
2e-6 RMS for WaveNets; LSTMs over these long runs 1e-5 against the f32 oracle vectors and 2e-5 against the float64 ones (north star: 1e-4)."""
import os

import numpy as np
import pytest

import na_oracle as O
from golden.make_golden import MATRIX_MODELS, SWITCH_PLAN

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def na():
    import neuralaudio_amd
    if neuralaudio_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the product path has no CPU fallback")
    return neuralaudio_amd


@pytest.fixture(scope="module")
def matrix():
    return np.load(os.path.join(GOLDEN, "fixture_matrix.npz"))


@pytest.mark.parametrize("tag,name,q", MATRIX_MODELS)
@pytest.mark.parametrize("block", [128, 100])
def test_hip_path_reproduces_the_committed_matrix(na, matrix, tag, name, q, block):
    # LSTMs: the recurrence runs 6144 samples (2048 prewarm + 4096); BossLSTM-1x16 is still on a slow zero-input transient then
    # (-0.0271 -> -0.0296 over the 4096 zeros), along which two f32 evaluations with different summation orders drift apart by 6e-6
    # and an f32 / float64 pair by a little more
    tol = {"oracle": 1e-5 if tag.startswith("lstm") else 2e-6, "np64": 2e-5 if tag.startswith("lstm") else 2e-6}
    for k in ("sine", "noise", "zeros"):
        ld = na.NeuralModelLoader()
        ld.SetDefaultQualityScaleFactor(q)
        m = ld.CreateFromFile(os.path.join(O.MODELS_DIR, name))
        x = matrix["input/" + k]
        y = np.concatenate([m.Process(x[i:i + block]) for i in range(0, x.size, block)])
        for src in ("np64", "oracle"):
            want = matrix["%s/%s/%s" % (src, tag, k)]
            assert O.rms(y - want) < tol[src], (tag, k, src, O.rms(y - want))
        if k != "zeros":
            assert O.rms(matrix["np64/%s/%s" % (tag, k)]) > 0.01


def test_hip_path_reproduces_the_committed_quality_switch_run(na, matrix):
    ld = na.NeuralModelLoader()
    m = ld.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-a2.nam"))
    b = na.Batch(0)
    b.AddStreams(m, 2, quality=SWITCH_PLAN[0][1])
    x = matrix["switch/input"].reshape(32, 128)
    plan = dict(SWITCH_PLAN)
    got = []
    for i in range(32):
        if i in plan:
            b.SetQuality(1, plan[i])
        got.append(b.Process(np.stack([x[i], x[i]]))[1])
        assert b.GetActiveSubModel(1) == int(matrix["switch/active"][i])
    y = np.concatenate(got)
    for src in ("np64", "oracle"):
        assert O.rms(y - matrix["switch/" + src]) < 2e-6, src
