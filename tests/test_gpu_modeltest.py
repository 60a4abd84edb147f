"""ModelTest on the GPU box: the C++ host of the exported C++ API (tools/ModelTest, counterpart of the reference's Utils/ModelTest)
runs BASELINE config 1's workload (LSTM 1x16, single stream, 128-sample buffers) and the headline model, and its second-instance
RMS line (two instances driven with different call sizes) stays at float noise."""
import os
import re
import subprocess

import pytest

import na_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "bin", "ModelTest")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "ModelTest")], check=True)
    return EXE


@pytest.mark.parametrize("name,static", [("BossLSTM-1x16.nam", False), ("BossWN-standard.nam", True)])
def test_modeltest_runs_the_reference_protocol(exe, name, static):
    r = subprocess.run([exe, "-b", "128", os.path.join(O.MODELS_DIR, name)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "Block size: 128  Quality Scale: 1" in out
    m = re.search(r"^Internal: ([0-9.e+-]+) \(([0-9.e+-]+)xRT\)$", out, re.M)  # ModelTest.cpp:120-123 format
    assert m, out
    seconds, xrt = float(m.group(1)), float(m.group(2))
    assert seconds > 0 and abs(xrt - (4096 * 64 / 48000.0) / seconds) / xrt < 1e-3
    assert ("not using a static architecture" in out) == (not static)
    rms = float(re.search(r"RMS err: ([0-9.e+-]+)", out).group(1))
    assert rms < 2e-6, out


def test_modeltest_quality_and_batch_options(exe):
    r = subprocess.run([exe, "-b", "64", "-q", "0.0", "--streams", "256", os.path.join(O.MODELS_DIR, "BossWN-a2.nam")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Block size: 64  Quality Scale: 0" in r.stdout
    assert re.search(r"^Batch x256: [0-9.e+-]+ \([0-9.e+-]+xRT\)$", r.stdout, re.M), r.stdout


def test_modeltest_default_model_set(exe):
    """No model argument: the A2 Full / A2 Lite / A1 Standard / LSTM 1x16 set from a Models folder up the path (ModelTest.cpp:220-267)."""
    r = subprocess.run([exe, "-b", "128"], capture_output=True, text=True, timeout=900, cwd=os.path.join(ROOT, "tests"))
    assert r.returncode == 0, r.stdout + r.stderr
    for title in ("WaveNet (A2 Full) Test", "WaveNet (A2 Lite) Test", "WaveNet (A1 Standard) Test", "LSTM (1x16) Test"):
        assert title in r.stdout
    assert len(re.findall(r"^Internal: ", r.stdout, re.M)) == 4
