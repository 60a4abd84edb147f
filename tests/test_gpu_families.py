"""Both WaveNet kernel families -- and every kernel variant that ships behind a tuning knob -- against the oracle on EVERY architecture.

By default a model runs on the family that is faster for it (FamilyFor() in gpu_batch.cpp: the f16-split kernel for Standard-like
models and padded A1 Lite, the f32 frame kernel for narrow / large-kernel ones, the runtime-shaped kernel for arrays wider than 16
channels; large batches of narrow models run packed).
NA_WN_KERNEL forces one family for all models it can run; it is read
once per process, so each forced run is a subprocess.  Two tiers: four fallbacks over the direct parity file ride in `-m gpu`
(seconds each); the full knob matrix over parity + fuzz + batch (245 tests per run) is a soak test behind `-m gpu_soak`.
Every subprocess has its own limit below the per-test watchdog, so a stall is reported with the output it produced."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = os.path.join(ROOT, "tests")
KNOBS = ("NA_WN_KERNEL", "NA_LSTM_NO_DPP", "NA_LSTM_LANE_KERNEL", "NA_REC_NOSKEW", "NA_WN_PACK", "NA_LSTM_NO_WAVE_RT", "NA_WN_SPEC", "NA_HOST_DIRECT",
         "NA_REC_QUAD_MIN", "NA_HOST_HALVES", "NA_REC_RPL", "NA_WN_DENSE", "NA_BATCH_NO_GRAPH", "NA_REC_NOPIPE")

SOAK = [{"NA_WN_SPEC": "0"}, {"NA_WN_KERNEL": "split"}, {"NA_WN_KERNEL": "split", "NA_SP_T": "4"}, {"NA_WN_KERNEL": "split", "NA_SP_GEN": "1"},
        {"NA_WN_KERNEL": "frame"}, {"NA_WN_KERNEL": "frame", "NA_FR_PF": "2"}, {"NA_WN_KERNEL": "frame", "NA_FR_PF": "0"},
        {"NA_WN_KERNEL": "frame", "NA_FR_SPB": "4"}, {"NA_WN_KERNEL": "generic"}, {"NA_WN_PACK": "1"}, {"NA_LSTM_NO_DPP": "1", "NA_GRU_NO_DPP": "1"},
        {"NA_LSTM_LANE_KERNEL": "1"}, {"NA_REC_NOSKEW": "1", "NA_REC_NO_DPP32": "1"}, {"NA_LSTM_NO_WAVE_RT": "1"},
        {"NA_HOST_DIRECT": "0"},    # host buffers through the copy engines instead of kernels on the pinned block
        {"NA_REC_QUAD_MIN": "1"},   # every recurrent launch that can on the four-streams-per-wave layout, whatever its size
        {"NA_HOST_HALVES": "0"},    # no free-running half-batch chains: every buffer as ordered launches on the batch stream
        {"NA_WN_DENSE": "0"},       # four Nano streams at 16 / 16 virtual channels (default: 16 / 8, two streams per channel group)
        {"NA_REC_RPL": "4"},        # runtime-shaped recurrent kernel: four gate rows per lane (a quarter of the waves per stream)
        {"NA_BATCH_NO_GRAPH": "1"}, # multi-unit batches: fork / join issued directly every buffer (the path of a runtime older than the build's)
        {"NA_REC_NOPIPE": "1"}]     # two-layer 16-unit LSTMs on one wave per stream (default: one wave per layer below 1536 waves)

# the four fallbacks a deployment can actually land on, over the direct parity file only: part of -m gpu, a few seconds each
SHORT = [{"NA_WN_KERNEL": "frame"}, {"NA_WN_SPEC": "0"}, {"NA_LSTM_NO_DPP": "1", "NA_GRU_NO_DPP": "1"}, {"NA_HOST_HALVES": "0"}]


def ident(e):
    return ",".join("%s=%s" % kv for kv in e.items())


def forced_run(env, files, limit):
    if any(os.environ.get(k) for k in KNOBS):
        pytest.skip("already inside a forced run")
    e = dict(os.environ, **env)
    t0 = time.monotonic()
    try:    # --durations: a slow forced run names its slow tests in the failure text
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "--durations=8", "-p", "no:cacheprovider"] + files,
                           env=e, capture_output=True, text=True, timeout=limit)
    except subprocess.TimeoutExpired as ex:
        out = ex.stdout.decode(errors="replace") if isinstance(ex.stdout, bytes) else (ex.stdout or "")
        pytest.fail("forced run %s did not finish in %d s; output so far:\n%s" % (ident(env), limit, out[-3000:]))
    assert r.returncode == 0, "%.0f s\n" % (time.monotonic() - t0) + r.stdout[-3000:] + r.stderr[-1500:]


@pytest.mark.gpu
@pytest.mark.watchdog(260)
@pytest.mark.parametrize("env", SHORT, ids=ident)
def test_forced_fallback_passes_the_parity_suite(env):
    # (the parity file needs no torch: the child skips the warm import -- 16 s on a good box; one r06 run on a box with slow storage
    # spent more than 100 s before the child's first test, hence the generous limit)
    forced_run(dict(env, NA_TEST_NO_WARM="1"), [os.path.join(T, "test_gpu_parity.py")], 240)


@pytest.mark.gpu_soak
@pytest.mark.watchdog(330)
@pytest.mark.parametrize("env", SOAK, ids=ident)
def test_forced_family_passes_parity_fuzz_and_batch_suites(env):
    forced_run(env, [os.path.join(T, "test_gpu_parity.py"), os.path.join(T, "test_gpu_fuzz.py"), os.path.join(T, "test_gpu_batch.py")], 300)
