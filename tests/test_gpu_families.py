"""Both WaveNet kernel families -- and every kernel variant that ships behind a tuning knob -- against the oracle on EVERY architecture.

By default a model runs on the family that is faster for it (FamilyFor() in gpu_batch.cpp: the f16-split kernel for Standard-like
models and padded A1 Lite, the f32 frame kernel for narrow / large-kernel ones, the runtime-shaped kernel for arrays wider than 16
channels; large batches of narrow models run packed).
NA_WN_KERNEL forces one family for all models it can run; it is read
once per process, so each forced run is a subprocess of the same parity + fuzz + batch test files."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"NA_WN_SPEC": "0"}, {"NA_WN_KERNEL": "split"}, {"NA_WN_KERNEL": "split", "NA_SP_T": "4"}, {"NA_WN_KERNEL": "split", "NA_SP_GEN": "1"},
                                 {"NA_WN_KERNEL": "frame"}, {"NA_WN_KERNEL": "frame", "NA_FR_PF": "2"}, {"NA_WN_KERNEL": "frame", "NA_FR_PF": "0"},
                                 {"NA_WN_KERNEL": "frame", "NA_FR_SPB": "4"}, {"NA_WN_KERNEL": "generic"}, {"NA_WN_PACK": "1"}, {"NA_LSTM_NO_DPP": "1", "NA_GRU_NO_DPP": "1"}, {"NA_LSTM_LANE_KERNEL": "1"}, {"NA_REC_NOSKEW": "1", "NA_REC_NO_DPP32": "1"}, {"NA_LSTM_NO_WAVE_RT": "1"},
                                 {"NA_HOST_DIRECT": "0"},    # host buffers through the copy engines instead of kernels on the pinned block
                                 {"NA_REC_QUAD_MIN": "1"},   # every recurrent launch that can on the four-streams-per-wave layout, whatever its size
                                 {"NA_HOST_HALVES": "0"},    # no free-running half-batch chains: every buffer as ordered launches on the batch stream
                                 {"NA_WN_DENSE": "0"},       # four Nano streams at 16 / 16 virtual channels (default: 16 / 8, two streams per channel group)
                                 {"NA_REC_RPL": "4"}],       # runtime-shaped recurrent kernel: four gate rows per lane (a quarter of the waves per stream)
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_forced_family_passes_parity_fuzz_and_batch_suites(env):
    if os.environ.get("NA_WN_KERNEL") or os.environ.get("NA_LSTM_NO_DPP") or os.environ.get("NA_LSTM_LANE_KERNEL") or os.environ.get("NA_REC_NOSKEW") or os.environ.get("NA_WN_PACK") or os.environ.get("NA_LSTM_NO_WAVE_RT") or os.environ.get("NA_WN_SPEC") or os.environ.get("NA_HOST_DIRECT") or os.environ.get("NA_REC_QUAD_MIN") or os.environ.get("NA_HOST_HALVES") or os.environ.get("NA_REC_RPL") or os.environ.get("NA_WN_DENSE"):
        pytest.skip("already inside a forced run")
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_fuzz.py"), os.path.join(ROOT, "tests", "test_gpu_batch.py")],
                       env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
