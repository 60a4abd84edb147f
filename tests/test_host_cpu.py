"""CPU-side tests of the product's host code through the C ABI: no GPU needed, no compute calls.

  * the shared library loads and exports every symbol include/*.h declares
  * the loader (C++) agrees with the independent Python reading of the same files (tests/na_oracle.py)
  * error behaviour of the boundary (missing file -> NULL, malformed / wrong weight count -> error, no exception leaks)
  * the product never touches oracle/ and fails loudly without a GPU
"""
import ctypes as C
import copy
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import na_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def na():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "neuralaudio_amd", "libNeuralAudioCAPI.so")):
        g.build()
    import neuralaudio_amd
    return neuralaudio_amd


def _declared_symbols():
    names = []
    for hdr in ("NeuralAudioCApi.h", "neuralaudio_amd.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
        names += re.findall(r"NA_EXTERN\s+[\w\s\*]+?\b(\w+)\s*\(", text)
    return names


def test_library_exports_every_declared_symbol(na):
    from neuralaudio_amd import capi
    lib = capi.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 15 + 20
    for legacy in capi.LEGACY_SYMBOLS:
        assert legacy in declared  # the reference's 15 symbols, NeuralAudioCApi.h:18-46
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    for name in declared:
        assert name in exported, name
        getattr(lib, name)
    assert sorted(capi.LEGACY_SYMBOLS + capi.NA_SYMBOLS) == sorted(declared)


def test_cpp_api_is_exported_and_a_cpp_host_links(na, tmp_path):
    """The C++ boundary (NeuralModel.h:33-153 of the reference): NeuralModel / NeuralModelLoader are exported from the shared
    library, and a C++ translation unit written against include/NeuralAudio/NeuralModel.h alone links and runs."""
    from neuralaudio_amd import capi
    out = subprocess.run(["nm", "-D", "-C", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for sym in ("NeuralAudio::NeuralModelLoader::CreateFromFile(", "NeuralAudio::NeuralModelLoader::CreateFromStream(",
                "NeuralAudio::NeuralModelLoader::CreateFromString(", "NeuralAudio::NeuralModelLoader::SupportsWaveNetLoadMode(",
                "NeuralAudio::NeuralModelLoader::SupportsLSTMLoadMode(", "vtable for NeuralAudio::NeuralModel",
                "typeinfo for NeuralAudio::NeuralModel"):
        assert sym in out, sym
    src = tmp_path / "host.cpp"
    src.write_text(
        '#include <NeuralAudio/NeuralModel.h>\n#include <cstdio>\n'
        'int main(int argc, char** argv) {\n'
        '  NeuralAudio::NeuralModelLoader loader;\n'
        '  loader.SetDefaultMaxAudioBufferSize(128);\n'
        '  NeuralAudio::NeuralModel* m = loader.CreateFromFile(argv[1], false);\n'
        '  if (!m) return 3;\n'
        '  std::printf("%d %g %d %d\\n", (int)m->GetLoadMode(), m->GetSampleRate(), m->GetReceptiveFieldSize(), (int)m->IsStatic());\n'
        '  delete m;\n  return loader.CreateFromFile("/nonexistent.nam") == nullptr ? 0 : 4;\n}\n')
    exe = tmp_path / "host"
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lNeuralAudioCAPI",
                    "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([str(exe), os.path.join(O.MODELS_DIR, "BossWN-standard.nam")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["0", "48000", "4092", "1"]


# declaration order of the virtuals of the reference's class NeuralModel (NeuralAudio/NeuralModel.h:40-134; the two destructor entries
# take slots 0 and 1)
REFERENCE_VIRTUAL_ORDER = ["GetLoadMode", "HasQualityScaling", "GetQualityScaleFactor", "IsQualityChangeRealtimeSafe", "SetQualityScaleFactor",
                           "IsStatic", "SetMaxAudioBufferSize", "SetAudioInputLevelDBu", "GetAudioInputLevelDBu",
                           "GetRecommendedInputDBAdjustment", "GetRecommendedOutputDBAdjustment", "GetSampleRate", "GetReceptiveFieldSize",
                           "GetModelVersion", "GetMetadata", "Process", "Prewarm"]


def test_vtable_slots_follow_the_reference_header(na, tmp_path):
    """Binary-level drop-in of the C++ class: every virtual sits in the vtable slot a host compiled against the reference's
    NeuralModel.h expects (slots follow declaration order), and the object has the reference's data members
    (5 floats, a string, a vector: NeuralModel.h:139-145)."""
    exe = tmp_path / "vtable_slots"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "vtable_slots.cpp"), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    slots = {l.split()[0]: int(l.split()[1]) for l in lines if l and not l.startswith("sizeof")}
    assert [n for n, _ in sorted(slots.items(), key=lambda kv: kv[1])] == REFERENCE_VIRTUAL_ORDER
    assert sorted(slots.values()) == list(range(2, 2 + len(REFERENCE_VIRTUAL_ORDER)))
    # vptr + 5 floats (padded to 8) + std::string + std::vector
    assert int([l for l in lines if l.startswith("sizeof")][0].split()[1]) == 8 + 24 + 32 + 24
    ref = "/root/reference/NeuralAudio/NeuralModel.h"
    if os.path.exists(ref):  # (this container only: the list above against the header itself)
        text = open(ref).read()
        body = text[text.index("class NeuralModel"):text.index("class NeuralModelLoader")]
        assert re.findall(r"virtual\s+[\w:<>]+\s+(\w+)\s*\(", body) == REFERENCE_VIRTUAL_ORDER


def test_range_proof_decides_the_wavenet_kernel_family(na):
    """the f16-split kernels run a model only when the plan builder proves that no value leaves the f16 range for inputs
    within a limit >= 8 and that the weights fit the operand format; everything else runs on the f32 frame kernel (host-side decision,
    NA_ModelKernelInfo -- the GPU tests check the kernel that really runs and its parity)."""
    loader = na.NeuralModelLoader()
    std = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
    info = std.KernelInfo(1.0, 1024)
    assert info["kernel"] == "f16-split" and info["range_proven"] and info["weights_ok"] and 8.0 <= info["input_limit"] <= 32752.0
    arrays = O.a1_arrays(16, 8)
    w = O.synth_wavenet_weights(arrays, seed=5)

    def info_of(weights, arr=arrays, streams=1024):
        m = loader.CreateFromString(O.nam_json_wavenet_generic(arr, weights), ".nam", doPrewarm=False)
        assert m is not None
        return m.KernelInfo(1.0, streams)

    assert info_of(w)["kernel"] == "f16-split"
    # tanh bounds what a layer adds to the residual stream by the row sums of its 1x1: x 4000 they sum past the f16 range
    big = info_of(O.scale_wavenet_tensors(arrays, w, {"1x1": 4000.0}))
    assert big["kernel"] == "frame" and not big["range_proven"] and big["weights_ok"] and big["input_limit"] == float("inf")
    # a moderate scale only lowers the input limit
    mid = info_of(O.scale_wavenet_tensors(arrays, w, {"1x1": 30.0}))
    assert mid["kernel"] == "f16-split" and mid["range_proven"] and 8.0 <= mid["input_limit"] < info_of(w)["input_limit"]
    # weights beyond half the f16 range / a weight matrix below 2^-12: not representable as (hi, lo) f16 pairs
    assert info_of(O.scale_wavenet_tensors(arrays, w, {"mixin": 1e5}))["weights_ok"] is False
    assert info_of(O.scale_wavenet_tensors(arrays, w, {"mixin": 1e5}))["kernel"] == "frame"
    tiny = info_of(O.scale_wavenet_tensors(arrays, w, {"rechannel": 1e-6}))
    assert tiny["kernel"] == "frame" and not tiny["weights_ok"]
    # narrow models: no packing without the proof (packing means the f16-split kernels)
    nano = O.a1_arrays(4, 2)
    wn = O.synth_wavenet_weights(nano, seed=6)
    assert info_of(wn, nano, 4096)["pack"] == 4 and info_of(wn, nano, 4096)["kernel"] == "f16-split"
    bad = info_of(O.scale_wavenet_tensors(nano, wn, {"1x1": 1e4}), nano, 4096)
    assert bad["pack"] == 1 and bad["kernel"] == "frame"
    # LeakyReLU: the worst case grows with the product of the layers' row sums.  The official A2 shapes stay on their chains
    # (saturating arithmetic + NA_BatchStreamRangeEvents); any other LeakyReLU model without a proof runs in f32
    a2 = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-a2.nam"), doPrewarm=False)
    for q in (0.0, 1.0):
        ia = a2.KernelInfo(q, 1024)
        assert ia["kernel"] == "f16-split" and not ia["range_proven"] and ia["weights_ok"] and 8.0 <= ia["input_limit"] <= 32752.0
    for layers, want in ((2, "f16-split"), (14, "frame")):
        leaky = [dict(input_size=1, condition_size=1, head_size=1, head_kernel_size=1, head_dilation=1, channels=16, has_head_bias=True,
                      activation=O.ACT_LEAKYRELU, kernel_sizes=[3] * layers, dilations=[1 << (i % 9) for i in range(layers)])]
        il = info_of(O.synth_wavenet_weights(leaky, seed=7), leaky)
        assert il["kernel"] == want and il["range_proven"] == (want == "f16-split"), (layers, il)


def test_rccl_is_bound_at_run_time_and_the_library_does_not_link_it(na):
    """The multi-GPU host's RCCL fan-out / fan-in (csrc/multi_gpu.cpp, rccl_dyn.cpp): librccl.so is loaded with dlopen when a multi
    batch asks for it -- every entry point resolves against the installed library (no GPU needed for that) -- and the product library
    itself still links only libamdhip64 (single-GPU hosts never load RCCL)."""
    from neuralaudio_amd import capi
    needed = subprocess.run(["readelf", "-d", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    libs = re.findall(r"Shared library: \[([^\]]+)\]", needed)
    assert any(l.startswith("libamdhip64") for l in libs) and not any("rccl" in l or "nccl" in l for l in libs), libs
    if not os.path.exists("/opt/rocm/lib/librccl.so.1"):
        pytest.skip("no RCCL in this image")
    assert na.rccl_available(), capi.last_error()
    # the fan-in mode is a property of a multi batch, fixed at Commit; without a device the batch itself cannot be created
    if na.device_count() == 0:
        with pytest.raises(na.NeuralAudioError):
            na.MultiBatch([0])


def test_modeltest_host_builds_and_reports_a_missing_gpu(na):
    """tools/ModelTest (the reference's Utils/ModelTest counterpart) is a C++ host of the exported API; without a device it must say so."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "ModelTest")], check=True, capture_output=True)
    exe = os.path.join(ROOT, "tools", "bin", "ModelTest")
    assert subprocess.run([exe, "--bogus"], capture_output=True).returncode == 1
    if na.device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_modeltest.py runs it for real")
    r = subprocess.run([exe, "-b", "128", os.path.join(O.MODELS_DIR, "BossLSTM-1x16.nam")], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stdout and "Block size: 128  Quality Scale: 1" in r.stdout


def test_release_library_exports_the_documented_surface_and_nothing_else(na):
    """make release -> dist/libNeuralAudioCAPI.so (-DNA_RELEASE -DNA_NO_TUNING, no loopback RCCL table): its dynamic symbol table is the 15
    legacy symbols of the reference's NeuralAudioCApi.h:18-46 plus the NA_* set include/neuralaudio_amd.h declares outside its
    `#ifndef NA_RELEASE` block -- no NA_Debug* hook, no loopback table, no tuning environment reader."""
    import re
    from neuralaudio_amd import capi
    subprocess.run(["make", "-C", os.path.join(ROOT, "neuralaudio_amd", "csrc"), "-j8", "release"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "dist", "libNeuralAudioCAPI.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "TWBDRV" and not l.split()[-1].startswith("_")}
    header = open(os.path.join(ROOT, "include", "neuralaudio_amd.h")).read()
    public, hooks = header.split("#ifndef NA_RELEASE")[0], header.split("#ifndef NA_RELEASE")[1].split("#endif /* NA_RELEASE */")[0]
    declared = set(re.findall(r"NA_EXTERN[^;(]*?\b(NA_[A-Za-z0-9]+)\(", public))
    debug = set(re.findall(r"NA_EXTERN[^;(]*?\b(NA_[A-Za-z0-9]+)\(", hooks))
    assert debug and all(n.startswith("NA_Debug") for n in debug)
    assert set(capi.NA_SYMBOLS) == declared | debug  # (the Python binding list and the header agree)
    assert exported == set(capi.LEGACY_SYMBOLS) | declared, (sorted(exported - set(capi.LEGACY_SYMBOLS) - declared), sorted((set(capi.LEGACY_SYMBOLS) | declared) - exported))
    data = open(lib, "rb").read()
    assert b"NA_WN_KERNEL" not in data and b"NA_LSTM_LANE_KERNEL" not in data  # no tuning knob is read
    assert b"gfx950" in data and b"WaveNetSpecKernel" in data


def test_no_kernel_feeds_a_loaded_register_to_a_packed_f32_instruction_through_op_sel(na):
    """gfx950 (profiles/r06_quad_race.txt, tools/microbench/pk_lds_opsel.hip): v_pk_fma_f32 with a non-default op_sel / op_sel_hi on a register
    that a 128-bit load (ds_read_b128, global_load_dwordx4) has just delivered is wrong in lanes 48 .. 63 now and then while another wave
    issues MFMAs on the SIMD.  The library's kernels have thousands of packed-f32 operands with op_sel -- on ALU / MFMA results -- and must
    have none on a loaded register: the hand-written ones go through pairs or copies, the compiler's own pairing (SLP) is off for the
    recurrent files (csrc/Makefile).  A scan of the disassembly of every code object in the built library."""
    import importlib.util
    from neuralaudio_amd import capi
    spec = importlib.util.spec_from_file_location("pk_opsel_sources", os.path.join(ROOT, "tools", "analysis", "pk_opsel_sources.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    if not os.path.exists(scan.OBJDUMP):
        pytest.skip("no llvm-objdump under /opt/rocm")
    n, total, found = scan.scan_library(capi.LIB_PATH)
    assert n >= 10 and sum(total.values()) > 1000  # (the scan saw the kernels and their packed instructions)
    assert any("RecurrentQuadKernel" in k for k in total)
    assert not found, sorted(found.items())


def test_library_embeds_gfx950_code_object(na):
    from neuralaudio_amd import capi
    data = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"WaveNetSplitKernel" in data and b"WaveNetFrameKernel" in data and b"LstmBlockKernel" in data


@pytest.mark.parametrize("name", ["BossWN-standard.nam", "BossWN-feather.nam", "BossWN-nano.nam", "BossWN-a2.nam",
                                  "BossLSTM-1x16.nam", "BossLSTM-2x8.nam", "tw40_blues_deluxe_deerinkstudios.json"])
def test_loader_reads_reference_sample_models(na, name):
    loader = na.NeuralModelLoader()
    for wide in (False, True):
        m = loader.CreateFromFile(os.path.join(O.MODELS_DIR, name), doPrewarm=False, use_wchar_entry=wide) if not wide else None
        if m is None:
            continue
        j = O.load_json(name)
        assert m.GetLoadMode() == na.EModelLoadMode.Internal
        assert m.GetSampleRate() == 48000.0
        if name.endswith(".nam"):
            assert m.GetModelVersion() == j["version"]
            loud = j["metadata"]["loudness"]
            assert m.GetRecommendedOutputDBAdjustment() == pytest.approx(-18.0 - loud, abs=1e-5)  # NeuralModel.h:92-95
            assert float(m.GetMetadata("loudness")) == pytest.approx(loud)
            assert m.GetMetadata("no_such_field") == ""
        else:
            assert m.GetRecommendedOutputDBAdjustment() == 0.0
        assert m.GetRecommendedInputDBAdjustment() == 0.0  # 12 dBu default on both sides
        arch = j.get("architecture")
        if arch == "WaveNet":
            assert m.GetReceptiveFieldSize() == 4092 and m.IsStatic() and not m.HasQualityScaling()
        elif arch == "SlimmableContainer":
            assert m.GetReceptiveFieldSize() == 6346 and m.IsStatic() and m.HasQualityScaling()
        else:
            assert m.GetReceptiveFieldSize() == -1 and not m.HasQualityScaling()


def test_quality_scaling_follows_reference_rule(na):
    loader = na.NeuralModelLoader()
    loader.SetDefaultQualityScaleFactor(0.25)
    m = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-a2.nam"), doPrewarm=False)
    assert m.HasQualityScaling() and m.GetQualityScaleFactor() == pytest.approx(0.25)
    m.SetQualityScaleFactor(0.9)
    assert m.GetQualityScaleFactor() == pytest.approx(0.9)
    plain = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-nano.nam"), doPrewarm=False)
    plain.SetQualityScaleFactor(0.1)
    assert plain.GetQualityScaleFactor() == 1.0  # NeuralModel.h:50-53 default


def test_input_level_calibration(na):
    loader = na.NeuralModelLoader()
    loader.SetAudioInputLevelDBu(18.0)
    m = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-nano.nam"), doPrewarm=False)
    assert m.GetRecommendedInputDBAdjustment() == pytest.approx(6.0)  # audioInputLevelDBu - modelInputLevelDBu(12)


def test_missing_and_malformed_files(na, tmp_path):
    loader = na.NeuralModelLoader()
    assert loader.CreateFromFile(str(tmp_path / "nope.nam")) is None  # NeuralModel.cpp:321-322
    bad = tmp_path / "bad.nam"
    bad.write_text("{ this is not json")
    with pytest.raises(na.NeuralAudioError):
        loader.CreateFromFile(str(bad))
    j = O.load_json("BossWN-nano.nam")
    j["weights"] = j["weights"][:-3]
    short = tmp_path / "short.nam"
    short.write_text(json.dumps(j))
    with pytest.raises(na.NeuralAudioError, match="Wrong number of weights. Expected 842 but got 839"):  # at LOAD, like WaveNet.h:704-709
        loader.CreateFromFile(str(short), doPrewarm=False)
    arr = [dict(O.a1_arrays(4, 2)[0], channels=20, head_size=1)]
    short.write_text(O.nam_json_wavenet_generic(arr, O.synth_wavenet_weights(arr, seed=2)))
    assert loader.CreateFromFile(str(short), doPrewarm=False) is not None  # > 16 channels: the runtime-shaped kernel (WaveNetDynamic.h)
    arr = [dict(O.a1_arrays(4, 2)[0], channels=80, head_size=1)]
    short.write_text(O.nam_json_wavenet_generic(arr, O.synth_wavenet_weights(arr, seed=2)))
    assert loader.CreateFromFile(str(short), doPrewarm=False) is not None  # 65 .. 128 channels: WaveNetWideKernel
    arr = [dict(O.a1_arrays(4, 2)[0], channels=144, head_size=1)]
    short.write_text(O.nam_json_wavenet_generic(arr, [0.0]))
    with pytest.raises(na.NeuralAudioError, match="channels > 128"):  # kernel limits are load-time errors too
        loader.CreateFromFile(str(short), doPrewarm=False)
    arr = [dict(O.a1_arrays(4, 2)[0], channels=80, head_size=1, head_kernel_size=4)]
    short.write_text(O.nam_json_wavenet_generic(arr, [0.0]))
    with pytest.raises(na.NeuralAudioError, match="conv head on a layer array wider than 64"):
        loader.CreateFromFile(str(short), doPrewarm=False)
    conv = tmp_path / "conv.json"
    conv.write_text(json.dumps({"in_shape": [None, None, 1], "layers": [{"type": "conv2d", "shape": [None, None, 8], "weights": []},
                                                                        {"type": "dense", "shape": [None, None, 1], "weights": []}]}))
    assert loader.CreateFromFile(str(conv)) is None  # keras layer types without a kernel: no model (the reference needs RTNeural for them)
    gru = tmp_path / "gru.json"
    gru.write_text(json.dumps(O.synth_keras_gru(1, 16, seed=3)))
    g = loader.CreateFromFile(str(gru), doPrewarm=False)  # keras GRU: RTNeural's arithmetic in the reference, restated here
    assert g is not None and g.GetSampleRate() == 48000.0 and g.GetReceptiveFieldSize() == -1
    wide = O.synth_keras_gru(1, 16, seed=3)
    wide["layers"][-1]["weights"] = [[[0.1, 0.2]] * 16, [0.0, 0.0]]  # dense head with 2 outputs: a generic stack (output = unit 0)
    wide["layers"][-1]["shape"] = [None, None, 2]
    gru.write_text(json.dumps(wide))
    assert loader.CreateFromFile(str(gru), doPrewarm=False) is not None
    wide["layers"][-1]["shape"] = [None, None, 1]  # ... but the declared shape must match the weights
    gru.write_text(json.dumps(wide))
    with pytest.raises(na.NeuralAudioError, match="unexpected weight shapes"):
        loader.CreateFromFile(str(gru), doPrewarm=False)


def test_lstm_shapes_accepted_or_rejected_at_load(na, tmp_path):
    """Any hidden size the reference's dynamic LSTM takes loads (runtime-shaped kernel); what has no kernel fails at load, with a reason."""
    loader = na.NeuralModelLoader()
    path = tmp_path / "m.nam"
    # (up to 1024 units whatever the weight size: streamed from L2, a workgroup of up to 16 waves per stream from 257 gate rows on)
    for layers, hidden in ((1, 3), (1, 18), (3, 16), (2, 64), (3, 128), (2, 192), (1, 512)):
        path.write_text(O.nam_json_lstm(layers, hidden, O.synth_lstm_weights(layers, hidden, seed=hidden)))
        assert loader.CreateFromFile(str(path), doPrewarm=False) is not None
    path.write_text(O.nam_json_lstm(1, 1030, O.synth_lstm_weights(1, 1030, seed=1)))
    with pytest.raises(na.NeuralAudioError, match="LSTM 1x1030 is not supported"):
        loader.CreateFromFile(str(path), doPrewarm=False)
    gru = tmp_path / "gru.json"
    for layers, hidden in ((3, 16), (1, 96), (2, 128), (1, 160)):
        gru.write_text(json.dumps(O.synth_keras_gru(layers, hidden, seed=3)))
        assert loader.CreateFromFile(str(gru), doPrewarm=False) is not None  # any shape up to 1024 units: runtime-shaped GRU kernel
    # a dense TAIL (more than the 1-unit head) behind a wide recurrent layer needs the [samples][units] buffer of the block in LDS:
    # 2 x 400 x 64 floats do not fit
    j = O.synth_keras_gru(1, 400, seed=3)
    j["layers"][-1] = {"type": "dense", "activation": "tanh", "shape": [None, None, 4], "weights": [[[0.01] * 4] * 400, [0.0] * 4]}
    j["layers"].append({"type": "dense", "activation": "", "shape": [None, None, 1], "weights": [[[0.5]] * 4, [0.0]]})
    gru.write_text(json.dumps(j))
    with pytest.raises(na.NeuralAudioError, match="GRU 1x400 is not supported"):
        loader.CreateFromFile(str(gru), doPrewarm=False)
    gru.write_text(json.dumps(O.synth_keras_gru(1, 1100, seed=3)))
    with pytest.raises(na.NeuralAudioError, match="GRU 1x1100 is not supported"):
        loader.CreateFromFile(str(gru), doPrewarm=False)


def test_generic_keras_stacks_accepted_or_rejected_at_load(na, tmp_path):
    """Stacks of lstm | gru and dense layers with activations -- what the reference hands to RTNeural (NeuralModel.cpp:565-572) -- load;
    layer types / activations / sizes without a kernel fail at load with a reason (no silent fallback)."""
    import ref_np as R
    loader = na.NeuralModelLoader()
    path = tmp_path / "m.json"
    for spec in ([("lstm", 8), ("dense", 6, "tanh"), ("dense", 1)], [("gru", 12), ("dense", 5, "relu"), ("dense", 3, "sigmoid"), ("dense", 1)],
                 [("dense", 8, "tanh"), ("dense", 4, "elu"), ("dense", 1)], [("gru", 8), ("dense", 2)], [("lstm", 16), ("dense", 1, "tanh")],
                 # the element-wise layers of RTNeural's parser, lowered to dense layers at load (model_loader.cpp AppendKerasTailLayer)
                 [("gru", 8), ("dense", 6), ("batchnorm", 6), ("activation", 6, "tanh"), ("dense", 1)],
                 [("lstm", 8), ("prelu", 8), ("dense", 4), ("prelu", 4, "scalar"), ("batchnorm", 4, "noaffine"), ("dense", 1)],
                 [("dense", 8), ("activation", 8, "relu"), ("batchnorm", 8), ("dense", 1)]):
        path.write_text(json.dumps(R.synth_keras_stack(spec, seed=5)))
        m = loader.CreateFromFile(str(path), doPrewarm=False)
        assert m is not None, spec
    bad = R.synth_keras_stack([("gru", 8), ("dense", 4, "tanh"), ("dense", 1)], seed=6)
    bad["layers"][1]["activation"] = "gelu"
    path.write_text(json.dumps(bad))
    with pytest.raises(na.NeuralAudioError, match="activation 'gelu' is not supported"):
        loader.CreateFromFile(str(path), doPrewarm=False)
    bad["layers"][1]["activation"] = "softmax"  # (round 5: across the units of its layer)
    path.write_text(json.dumps(bad))
    assert loader.CreateFromFile(str(path), doPrewarm=False) is not None
    path.write_text(json.dumps(R.synth_keras_stack([("gru", 8), ("dense", 100, "tanh"), ("dense", 1)], seed=6)))
    assert loader.CreateFromFile(str(path), doPrewarm=False) is not None  # (dense layers up to 256 units since round 4)
    path.write_text(json.dumps(R.synth_keras_stack([("gru", 8), ("dense", 300, "tanh"), ("dense", 1)], seed=6)))
    with pytest.raises(na.NeuralAudioError, match="wider than 256 units"):
        loader.CreateFromFile(str(path), doPrewarm=False)
    mixed = R.synth_keras_stack([("gru", 8), ("lstm", 8), ("dense", 1)], seed=7)  # two kinds of recurrent layers: no kernel
    path.write_text(json.dumps(mixed))
    assert loader.CreateFromFile(str(path), doPrewarm=False) is None
    wide = R.synth_keras_stack([("gru", 8), ("dense", 140), ("prelu", 140), ("dense", 1)], seed=9)  # relu(x) - alpha relu(-x): twice the width in between
    path.write_text(json.dumps(wide))
    with pytest.raises(na.NeuralAudioError, match="prelu layer wider than 128 units"):
        loader.CreateFromFile(str(path), doPrewarm=False)
    # conv1d layers (round 5): causal, dilated, behind the recurrent layers or in a stack without any; their limits are load errors
    for spec in ([("conv1d", 8, 3, 1, "tanh"), ("conv1d", 4, 2, 4, "relu"), ("dense", 1)], [("lstm", 8), ("conv1d", 6, 5, 3), ("prelu", 6), ("batchnorm", 6), ("dense", 1)],
                 [("gru", 12), ("conv1d", 16, 4, 64, "elu"), ("dense", 5, "softmax"), ("dense", 1)]):
        path.write_text(json.dumps(R.synth_keras_stack(spec, seed=8)))
        assert loader.CreateFromFile(str(path), doPrewarm=False) is not None, spec
    conv = R.synth_keras_stack([("conv1d", 4, 3, 1, "tanh"), ("dense", 1)], seed=8)
    conv["layers"][0]["strides"] = [2]
    path.write_text(json.dumps(conv))
    with pytest.raises(na.NeuralAudioError, match="stride or groups"):
        loader.CreateFromFile(str(path), doPrewarm=False)
    path.write_text(json.dumps(R.synth_keras_stack([("conv1d", 4, 3, 600), ("dense", 1)], seed=8)))
    with pytest.raises(na.NeuralAudioError, match="more than 1024 samples of history"):
        loader.CreateFromFile(str(path), doPrewarm=False)
    path.write_text(json.dumps(R.synth_keras_stack([("conv1d", 200, 3, 400), ("dense", 1)], seed=8)))  # two [200][800 + 128] buffers: beyond the LDS
    with pytest.raises(na.NeuralAudioError, match="conv1d layers with its two"):
        loader.CreateFromFile(str(path), doPrewarm=False)
    front = R.synth_keras_stack([("conv1d", 4, 3, 1, "tanh"), ("dense", 1)], seed=8)  # a conv1d layer in FRONT of a recurrent one: no kernel
    front["layers"].insert(1, R.synth_keras_stack([("dense", 4), ("gru", 8)], seed=8)["layers"][1])
    path.write_text(json.dumps(front))
    assert loader.CreateFromFile(str(path), doPrewarm=False) is None


def test_stream_packing_builds_a_block_diagonal_virtual_model(na):
    """Stream packing, host side (no GPU): Nano packs 4 streams, Feather 2, Standard / A2 none; the virtual model's flat weights hold
    the real model's tensors on the block diagonal, replicated vectors, zeros elsewhere, in the reference's weight order.  Nano's pack
    is dense: 16 / 8 virtual channels (wavenet_plan.cpp WaveNetPackCanBeDense)."""
    import ctypes as C
    from neuralaudio_amd import capi
    lib = capi.load_library()
    loader = na.NeuralModelLoader()
    mdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models")

    def packed(name):
        m = loader.CreateFromFile(os.path.join(mdir, name), doPrewarm=False)
        pf = C.c_int(0)
        n = lib.NA_DebugPackedWeights(m._h, C.byref(pf), None, 0)
        assert n >= 0, capi.last_error()
        out = np.zeros(n, np.float32)
        if n:
            assert lib.NA_DebugPackedWeights(m._h, C.byref(pf), out.ctypes.data_as(C.POINTER(C.c_float)), n) == n
        return pf.value, out

    assert packed("BossWN-standard.nam")[0] == 1 and packed("BossWN-a2.nam")[0] == 1
    for name, P in (("BossWN-nano.nam", 4), ("BossWN-feather.nam", 2)):
        pf, v = packed(name)
        assert pf == P
        j = O.load_json(name)
        w = np.array(j["weights"], np.float32)
        arrays = O.wavenet_arrays_from_nam(j)
        pad = [4 * ((a["channels"] + 3) // 4) for a in arrays]
        if P == 4 and [a["channels"] for a in arrays] == [4, 2] and os.environ.get("NA_WN_DENSE") != "0":
            pad[1] = 2  # dense pack (round 5): the 2-channel arrays of the four streams side by side, two streams per channel group
        pos = vpos = 0
        for ai, a in enumerate(arrays):
            C_, Cp = a["channels"], pad[ai]
            Cv = P * Cp
            In = a["input_size"]
            Inv = 1 if ai == 0 else P * pad[ai - 1]
            Inp = 1 if ai == 0 else pad[ai - 1]
            re_ = w[pos:pos + C_ * In].reshape(C_, In); pos += C_ * In
            vre = v[vpos:vpos + Cv * Inv].reshape(Cv, Inv); vpos += Cv * Inv
            for q in range(P):
                blk = vre[q * Cp:q * Cp + C_, (0 if ai == 0 else q * Inp):(1 if ai == 0 else q * Inp + In)]
                assert np.array_equal(blk, re_)
            assert np.count_nonzero(vre) == P * np.count_nonzero(re_)
            for K in a["kernel_sizes"]:
                conv = w[pos:pos + C_ * C_ * K].reshape(C_, C_, K); pos += C_ * C_ * K
                vec = [w[pos + i * C_:pos + (i + 1) * C_] for i in range(2)]; pos += 2 * C_
                w1 = w[pos:pos + C_ * C_].reshape(C_, C_); pos += C_ * C_
                b1 = w[pos:pos + C_]; pos += C_
                vconv = v[vpos:vpos + Cv * Cv * K].reshape(Cv, Cv, K); vpos += Cv * Cv * K
                vvec = [v[vpos + i * Cv:vpos + (i + 1) * Cv] for i in range(2)]; vpos += 2 * Cv
                vw1 = v[vpos:vpos + Cv * Cv].reshape(Cv, Cv); vpos += Cv * Cv
                vb1 = v[vpos:vpos + Cv]; vpos += Cv
                for q in range(P):
                    s_ = slice(q * Cp, q * Cp + C_)
                    assert np.array_equal(vconv[s_, s_], conv) and np.array_equal(vw1[s_, s_], w1) and np.array_equal(vb1[s_], b1)
                    assert np.array_equal(vvec[0][s_], vec[0]) and np.array_equal(vvec[1][s_], vec[1])
                assert np.count_nonzero(vconv) == P * np.count_nonzero(conv) and np.count_nonzero(vw1) == P * np.count_nonzero(w1)
            Hs = a["head_size"]
            last = ai == len(arrays) - 1
            Hp = 1 if last else pad[ai + 1]
            head = w[pos:pos + Hs * C_].reshape(Hs, C_); pos += Hs * C_
            vhead = v[vpos:vpos + P * Hp * Cv].reshape(P * Hp, Cv); vpos += P * Hp * Cv
            for q in range(P):
                assert np.array_equal(vhead[q * Hp:q * Hp + Hs, q * Cp:q * Cp + C_], head)
            assert np.count_nonzero(vhead) == P * np.count_nonzero(head)
            if a["has_head_bias"]:
                hb = w[pos:pos + Hs]; pos += Hs
                vhb = v[vpos:vpos + P * Hp]; vpos += P * Hp
                for q in range(P):
                    assert np.array_equal(vhb[q * Hp:q * Hp + Hs], hb)
        assert v[vpos] == w[pos] and vpos + 1 == v.size and pos + 1 == w.size  # head scale


def test_number_parsing_ignores_the_c_locale(na, tmp_path):
    """A host that called setlocale(LC_ALL, "") under a comma-decimal locale must still read "0.1234" as 0.1234."""
    import locale
    import subprocess
    import sys
    code = (
        "import locale, sys, os\n"
        "sys.path.insert(0, %r)\n"
        "ok = False\n"
        "for name in ('de_DE.UTF-8', 'fr_FR.UTF-8', 'de_DE', 'fr_FR', 'nl_NL.UTF-8', 'ru_RU.UTF-8'):\n"
        "    try:\n"
        "        locale.setlocale(locale.LC_ALL, name); ok = True; break\n"
        "    except locale.Error:\n"
        "        pass\n"
        "import neuralaudio_amd as na\n"
        "m = na.NeuralModelLoader().CreateFromFile(%r, doPrewarm=False)\n"
        "print('LOCALE' if ok else 'NOLOCALE', m.GetMetadata('loudness'), m.GetRecommendedOutputDBAdjustment())\n"
    ) % (ROOT, os.path.join(O.MODELS_DIR, "BossWN-nano.nam"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.split()
    ref = json.load(open(os.path.join(O.MODELS_DIR, "BossWN-nano.nam")))["metadata"]["loudness"]
    assert float(out[1]) == ref and abs(float(out[2]) - (-18 - ref)) < 1e-5
    # the image may carry no comma-decimal locale: the in-process check below covers the parser either way
    loader = na.NeuralModelLoader()
    j = O.load_json("BossWN-nano.nam")
    j["metadata"]["gain"] = 0.1234
    j["metadata"]["loudness"] = -20.5
    p = tmp_path / "m.nam"
    p.write_text(json.dumps(j))
    m = loader.CreateFromFile(str(p), doPrewarm=False)
    assert m.GetMetadata("gain") == "0.1234" and abs(m.GetRecommendedOutputDBAdjustment() - 2.5) < 1e-6


def test_metadata_values_are_dumped_like_nlohmann(na, tmp_path):
    """GetMetadata returns json.dump() text (NeuralModelImpl.h:85-94): shortest round-trip floats, "3.0" stays "3.0", sorted keys."""
    loader = na.NeuralModelLoader()
    j = O.load_json("BossWN-nano.nam")
    j["metadata"].update({"a_tenth": 0.1, "three": 3.0, "int": 3, "tiny": 1e-05, "big": 1.5e+20, "neg": -0.25,
                          "nested": {"zeta": 1, "alpha": [1.5, 2, "x"], "mid": None}})
    p = tmp_path / "m.nam"
    p.write_text(json.dumps(j))
    m = loader.CreateFromFile(str(p), doPrewarm=False)
    assert m.GetMetadata("a_tenth") == "0.1"
    assert m.GetMetadata("three") == "3.0"
    assert m.GetMetadata("int") == "3"
    assert m.GetMetadata("tiny") == "1e-05"
    assert m.GetMetadata("big") == "1.5e+20"
    assert m.GetMetadata("neg") == "-0.25"
    assert m.GetMetadata("nested") == '{"alpha":[1.5,2,"x"],"mid":null,"zeta":1}'


def test_numbers_beyond_the_double_range_follow_the_literal_not_its_spelling(na, tmp_path):
    """Out-of-range literals: underflow -> 0 whatever the spelling (plain decimals and positive exponents with many leading
    zeros included), overflow -> +-inf, subnormals keep their value (libstdc++'s from_chars reports them as out of range)."""
    loader = na.NeuralModelLoader()
    j = O.load_json("BossWN-nano.nam")
    text = json.dumps(j)
    extra = ('"u1": 1e-400, "u2": 0.' + "0" * 400 + '1, "u3": 0.' + "0" * 400 + '1e+20, "o1": 1e400, "o2": -1' + "0" * 400 + '.0, '
             '"sub": 4.9406564584124654e-324, "sub2": 2.5e-310, "edge": 1.7976931348623157e308, ')
    text = text.replace('"metadata": {', '"metadata": {' + extra, 1)
    p = tmp_path / "m.nam"
    p.write_text(text)
    m = loader.CreateFromFile(str(p), doPrewarm=False)
    assert m is not None
    assert float(m.GetMetadata("u1")) == 0.0 and float(m.GetMetadata("u2")) == 0.0 and float(m.GetMetadata("u3")) == 0.0
    assert m.GetMetadata("o1") in ("null", "inf", "Infinity") or float(m.GetMetadata("o1")) == float("inf")
    assert m.GetMetadata("o2") in ("null", "-inf", "-Infinity") or float(m.GetMetadata("o2")) == float("-inf")
    assert float(m.GetMetadata("sub")) == 5e-324 and float(m.GetMetadata("sub2")) == 2.5e-310
    assert float(m.GetMetadata("edge")) == 1.7976931348623157e308


def test_a2_features_outside_the_internal_path_are_rejected(na, tmp_path):
    """NAMIsA2Standard (NeuralModel.cpp:188-317) sends such files to NAM Core; without that back-end they must not load silently."""
    loader = na.NeuralModelLoader()
    base = O.load_json("BossWN-a2.nam")["config"]["submodels"][1]["model"]  # a plain A2 WaveNet (ch8)

    def variant(edit):
        j = copy.deepcopy(base)
        edit(j["config"]["layers"][0], j["config"])
        path = tmp_path / "v.nam"
        path.write_text(json.dumps(j))
        return str(path)

    assert loader.CreateFromFile(variant(lambda lc, c: None), doPrewarm=False) is not None
    # optional blocks that are simply absent are fine (they have no weights either)
    assert loader.CreateFromFile(variant(lambda lc, c: [lc.pop(k) for k in ("conv_pre_film", "head1x1", "slimmable")]), doPrewarm=False) is not None
    edits = {
        "head1x1": lambda lc, c: lc["head1x1"].update(active=True),
        "conv_post_film": lambda lc, c: lc["conv_post_film"].update(active=True),
        "gating_mode": lambda lc, c: lc.update(gating_mode=["gated"] * len(lc["dilations"])),
        "secondary_activation": lambda lc, c: lc.update(secondary_activation=[{"type": "Sigmoid"}] * len(lc["dilations"])),
        "bottleneck": lambda lc, c: lc.update(bottleneck=4),
        "layer1x1": lambda lc, c: lc["layer1x1"].update(active=False),
        "groups_input": lambda lc, c: lc.update(groups_input=2),
        "negative_slope": lambda lc, c: lc["activation"][3].update(negative_slope=0.2),
        "slimmable": lambda lc, c: lc.update(slimmable={"method": "slice_channels_uniform"}),
        "model-level head": lambda lc, c: c.update(head={"channels": 8}),
        "condition_dsp": lambda lc, c: c.update(condition_dsp={"architecture": "WaveNet"}),
        "in_channels": lambda lc, c: c.update(in_channels=2),
    }
    for what, edit in edits.items():
        with pytest.raises(na.NeuralAudioError, match="NAM Core"):
            loader.CreateFromFile(variant(edit), doPrewarm=False)


def test_unicode_path_through_wchar_entry(na, tmp_path):
    src = os.path.join(O.MODELS_DIR, "BossWN-nano.nam")
    dst = tmp_path / "mödel-音.nam"
    dst.write_bytes(open(src, "rb").read())
    loader = na.NeuralModelLoader()
    assert loader.CreateFromFile(str(tmp_path / "absent.nam"), use_wchar_entry=True) is None
    if na.device_count() == 0:
        # CreateModelFromFile prewarms (reference semantics); without a GPU the model stays a host-side template
        m = loader.CreateFromFile(str(dst), use_wchar_entry=True)
        assert m is not None and m.GetReceptiveFieldSize() == 4092


def test_load_modes_other_than_internal_are_rejected(na):
    from neuralaudio_amd import capi
    loader = na.NeuralModelLoader()
    loader.SetWaveNetLoadMode(na.EModelLoadMode.NAMCore)   # ignored like the reference without BUILD_NAMCORE
    loader.SetLSTMLoadMode(na.EModelLoadMode.RTNeural)
    m = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossLSTM-1x16.nam"), doPrewarm=False)
    assert m.GetLoadMode() == na.EModelLoadMode.Internal
    assert capi.load_library().NA_GetVersion().decode().startswith("neuralaudio_amd")


def test_process_without_gpu_fails_loudly(na):
    """No CPU fallback: on a box without a HIP device Process must report an error, not return numbers."""
    if na.device_count() > 0:
        pytest.skip("a GPU is present")
    from neuralaudio_amd import capi
    loader = na.NeuralModelLoader()
    m = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-nano.nam"), doPrewarm=False)
    x = np.ones(16, np.float32)
    y = np.full(16, 123.0, np.float32)
    fp = C.POINTER(C.c_float)
    capi.load_library().Process(m._h, x.ctypes.data_as(fp), y.ctypes.data_as(fp), 16)
    assert "no HIP device" in capi.last_error()
    assert np.all(y == 0.0)  # silence, never uninitialised memory (and never numbers from some fallback)
    # a second call must fail the same way (no half-built device state left behind by the first)
    y[:] = 123.0
    assert capi.load_library().NA_ProcessChecked(m._h, x.ctypes.data_as(fp), y.ctypes.data_as(fp), 16) != 0
    assert np.all(y == 0.0)
    with pytest.raises(na.NeuralAudioError):
        m.Process(x)
    with pytest.raises(na.NeuralAudioError):
        na.Batch(0)


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under neuralaudio_amd/ or include/ may mention it."""
    bad = []
    for base in ("neuralaudio_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath or "__pycache__" in dirpath:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if "na_oracle" in text or "oracle/" in text or "ref_np" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_shard_ranges_partition_and_balance(na):
    from neuralaudio_amd.sharding import shard_ranges
    for n, w in [(8192, 8), (16384, 8), (10, 3), (3, 8), (1, 1), (0, 2)]:
        r = shard_ranges([1.0] * n, w)
        assert len(r) == w and r[0][0] == 0 and r[-1][1] == n
        for (a, b), (c, d) in zip(r, r[1:]):
            assert b == c and a <= b
        if n >= w:
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    # config 5: half the streams cost 2434 B/sample (A2 Full), half 917.8 (A2 Lite), sorted by arch
    costs = [917.8] * 8192 + [2434.0] * 8192
    r = shard_ranges(costs, 8)
    loads = [sum(costs[a:b]) for a, b in r]
    assert max(loads) / (sum(loads) / 8) < 1.01


def test_a2_engine_selection_predicates(na):
    """NAMIsA2 (NeuralModel.cpp:159-168) and NAMIsA2Standard (:188-317), rule for rule, on accept / reject fixtures made from the
    reference's own A2 sample file."""
    from neuralaudio_amd import capi
    lib = capi.load_library()

    def classify(j):
        return lib.NA_DebugClassifyNam(json.dumps(j).encode())

    for v, want in (("0.5.4", 0), ("0.5.5", 1), ("0.6.0", 1), ("0.7.0", 1), ("1.0.0", 1), ("0.5.1", 0), ("0.4.9", 0)):
        assert classify({"version": v}) & 1 == want, v
    for sub in O.load_json("BossWN-a2.nam")["config"]["submodels"]:
        base = sub["model"]
        assert classify(base) == 3  # A2 version, standard architecture

        def variant(edit):
            j = copy.deepcopy(base)
            edit(j["config"]["layers"][0], j["config"], j)
            return classify(j) & 2

        assert variant(lambda lc, c, j: None) == 2
        rejects = [
            lambda lc, c, j: j.update(architecture="LSTM"),
            lambda lc, c, j: c.update(head={"x": 1}),
            lambda lc, c, j: c.update(condition_dsp=None),  # mere presence counts
            lambda lc, c, j: c.update(in_channels=2),
            lambda lc, c, j: c["layers"].append(copy.deepcopy(lc)),
            lambda lc, c, j: lc.update(input_size=2),
            lambda lc, c, j: lc.update(condition_size=2),
            lambda lc, c, j: lc.update(channels=4),
            lambda lc, c, j: lc.update(bottleneck=lc["channels"] + 1),
            lambda lc, c, j: lc["kernel_sizes"].__setitem__(3, 5),
            lambda lc, c, j: lc["dilations"].__setitem__(0, 2),
            lambda lc, c, j: lc["dilations"].append(1),
            lambda lc, c, j: lc["activation"].__setitem__(2, {"type": "Tanh"}),
            lambda lc, c, j: lc["activation"].__setitem__(2, {"type": "LeakyReLU", "negative_slope": 0.2}),
            lambda lc, c, j: lc["secondary_activation"].__setitem__(1, {"type": "Sigmoid"}),
            lambda lc, c, j: lc["gating_mode"].__setitem__(1, "gated"),
            lambda lc, c, j: lc["head"].update(out_channels=2),
            lambda lc, c, j: lc["head"].update(kernel_size=8),
            lambda lc, c, j: lc["head"].update(head_dilation=2),
            lambda lc, c, j: lc["head"].update(bias=False),
            lambda lc, c, j: lc["layer1x1"].update(active=False),
            lambda lc, c, j: lc["layer1x1"].update(groups=2),
            lambda lc, c, j: lc["head1x1"].update(active=True),
            lambda lc, c, j: lc["conv_pre_film"].update(active=True),
            lambda lc, c, j: lc["head1x1_post_film"].update(active=True),
            lambda lc, c, j: lc.pop("activation_pre_film"),  # the reference's IsActive(): a missing block counts as active
            lambda lc, c, j: lc.update(groups_input=2),
            lambda lc, c, j: lc.update(groups_input_mixin=4),
            lambda lc, c, j: lc.update(slimmable={"method": "x"}),
            lambda lc, c, j: lc.pop("head"),
        ]
        for k, edit in enumerate(rejects):
            assert variant(edit) == 0, k
        accepts = [
            lambda lc, c, j: lc["gating_mode"].__setitem__(1, None),
            lambda lc, c, j: lc["activation"].__setitem__(2, {"type": "LeakyReLU"}),  # default slope 0.01
            lambda lc, c, j: lc.pop("secondary_activation"),
            lambda lc, c, j: lc.pop("bottleneck"),
            lambda lc, c, j: lc.update(slimmable=None),
        ]
        for k, edit in enumerate(accepts):
            assert variant(edit) == 2, k
    for a1 in ("BossWN-standard.nam", "BossWN-nano.nam", "BossLSTM-1x16.nam"):
        assert classify(O.load_json(a1)) == 0


def _shard_ranges_in_python(costs, world_size):
    """NA_ShardByCost (csrc/multi_gpu.cpp) once more in plain Python: the test below compares the two."""
    n = len(costs)
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    total = float(sum(costs))
    ranges = []
    begin = 0
    acc = 0.0
    for rank in range(world_size):
        if rank == world_size - 1:
            end = n
        else:
            target = total * (rank + 1) / world_size
            end = begin
            while end < n and acc + costs[end] <= target + 1e-9:
                acc += costs[end]
                end += 1
            # at least one stream per rank while streams remain: take one even if it alone overshoots the share,
            # and leave one for each remaining rank
            cap = max(begin, n - (world_size - rank - 1))
            if end == begin and end < cap:
                end += 1
            end = min(end, cap)
            acc = float(sum(costs[:end]))
        ranges.append((begin, end))
        begin = end
    return ranges



def test_shard_by_cost_matches_the_python_restatement_and_covers_everything():
    """NA_ShardByCost (csrc/multi_gpu.cpp) is the partition both multi-GPU hosts use (the C++ NA_Multi* host and bench.py's
    one-process-per-GPU ranks through neuralaudio_amd.sharding): contiguous, ordered, disjoint, complete, near-equal cost, at least one
    item per range while items remain -- and identical to the plain-Python restatement."""
    import numpy as np
    from neuralaudio_amd.sharding import shard_ranges
    rng = np.random.default_rng(0)
    cases = [([1.0] * 8192, 8), ([41.6] * 4096 + [20.2] * 4096, 8), ([46.6] * 8192 + [71.4] * 8192, 8), ([1.0, 3.0, 1.0, 1.0, 2.0, 1.0, 1.0], 2),
             ([5.0], 4), ([], 3), ([0.0] * 10, 3), ([1.0] * 3, 8)]
    for _ in range(40):
        n = int(rng.integers(1, 400))
        cases.append((list(rng.choice([20.2, 33.7, 41.6, 71.4], size=n) * rng.uniform(0.5, 2.0)), int(rng.integers(1, 9))))
    for costs, parts in cases:
        got = shard_ranges(costs, parts)
        assert got == _shard_ranges_in_python(costs, parts), (costs[:8], parts, got)
        assert len(got) == parts and got[0][0] == 0 and got[-1][1] == len(costs)
        assert all(a <= b for a, b in got) and all(got[i][1] == got[i + 1][0] for i in range(parts - 1))
        if len(costs) >= parts:
            assert all(b > a for a, b in got), (parts, got)
    # balance: 8 ranks over config 5's global list (A2 ch3 then ch8): the ranks holding the expensive half get fewer streams
    costs = [46.6] * 8192 + [71.4] * 8192
    r = shard_ranges(costs, 8)
    loads = [sum(costs[a:b]) for a, b in r]
    assert max(loads) / min(loads) < 1.01 and (r[0][1] - r[0][0]) > (r[-1][1] - r[-1][0])


def test_stream_cost_estimates_rank_the_official_models(models_dir):
    import ctypes as C
    from neuralaudio_amd import capi
    lib = capi.load_library()
    loader = lib.CreateLoader()
    cost = {}
    for name in ("BossWN-standard.nam", "BossWN-feather.nam", "BossWN-nano.nam", "BossLSTM-1x16.nam", "BossLSTM-2x8.nam", "BossWN-a2.nam"):
        m = lib.NA_CreateModelFromFileUtf8(loader, os.path.join(models_dir, name).encode(), 0)
        assert m
        cost[name] = lib.NA_ModelStreamCost(m, C.c_float(1.0))
        if "a2" in name:
            cost["a2-lite"] = lib.NA_ModelStreamCost(m, C.c_float(0.0))
    assert cost["BossWN-standard.nam"] > cost["BossWN-feather.nam"] > cost["BossWN-nano.nam"] > 0
    assert cost["BossWN-a2.nam"] > cost["a2-lite"] > cost["BossLSTM-1x16.nam"] > 0
    assert 35.0 < cost["BossWN-standard.nam"] < 50.0 and 15.0 < cost["BossLSTM-1x16.nam"] < 25.0  # (us per 1024 streams x 128 frames)


def test_bench_cpu_baseline_legs_run(na):
    """bench.py's cpu_baseline leg (the oracle timed on the host cores: the only place outside tests / smoke() that may call it) for
    every kind of workload the bench line carries one for -- single models, the mixed batches, config 4's seeded recurrent models."""
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    r = bench.cpu_baseline("standard", 0.2)
    assert r["value"] > 0 and r["kind"] == "port" and r["cores"] >= 1 and "MULTIFRAME_8X8" in r["sample"]
    assert bench.mixed_cpu_baseline(["a2lite", "a2full"], 0.3)["value"] > 0
    rng = np.random.default_rng(4)
    H, a = 8, 0.25
    lw = np.concatenate([rng.uniform(-a, a, 4 * H * (1 + H) + 4 * H + 2 * H), rng.uniform(-a, a, H + 1)])
    lstm = {"architecture": "LSTM", "config": {"input_size": 1, "hidden_size": H, "num_layers": 1}, "weights": [float(v) for v in lw]}
    gru = {"layers": [{"type": "gru", "shape": [None, None, H], "weights": [rng.uniform(-a, a, (1, 3 * H)).tolist(), rng.uniform(-a, a, (H, 3 * H)).tolist(),
                                                                              rng.uniform(-a, a, (2, 3 * H)).tolist()]},
                      {"type": "dense", "shape": [None, None, 1], "weights": [rng.uniform(-a, a, (H, 1)).tolist(), [0.0]]}]}
    assert bench.mixed_cpu_baseline([("lstm", lstm), ("gru", gru)], 0.3)["value"] > 0
