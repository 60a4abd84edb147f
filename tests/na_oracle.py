"""ctypes binding for oracle/libna_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU checker (a restatement of the reference's Internal WaveNet/LSTM path,
see oracle/na_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product package never does.

Model-file reading here mirrors what the reference loader looks at
(NeuralAudio/NeuralModel.cpp:338-581) but is deliberately independent from the product's
C++ loader so the two can be checked against each other.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MODELS_DIR = os.path.join(ROOT, "tests", "golden", "models")

MAX_LAYERS = 64
ACT_TANH, ACT_LEAKYRELU = 0, 1
MATH_FAST, MATH_STD = 0, 1


class WnArrayCfg(C.Structure):
    _fields_ = [
        ("input_size", C.c_int),
        ("condition_size", C.c_int),
        ("head_size", C.c_int),
        ("head_kernel_size", C.c_int),
        ("head_dilation", C.c_int),
        ("channels", C.c_int),
        ("has_head_bias", C.c_int),
        ("activation", C.c_int),
        ("num_layers", C.c_int),
        ("kernel_sizes", C.c_int * MAX_LAYERS),
        ("dilations", C.c_int * MAX_LAYERS),
    ]


_lib = None


def build_oracle(native=False):
    target = "native" if native else "all"
    subprocess.run(["make", "-C", ORACLE_DIR, target], check=True, stdout=subprocess.DEVNULL)


def _bind(lib):
    fp = C.POINTER(C.c_float)
    lib.na_oracle_fast_tanh.restype = C.c_float
    lib.na_oracle_fast_tanh.argtypes = [C.c_float]
    lib.na_oracle_fast_sigmoid.restype = C.c_float
    lib.na_oracle_fast_sigmoid.argtypes = [C.c_float]
    lib.na_oracle_leaky_relu.restype = C.c_float
    lib.na_oracle_leaky_relu.argtypes = [C.c_float]
    lib.na_oracle_test_dense.restype = None
    lib.na_oracle_test_dense.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, C.c_int, C.c_int]
    lib.na_oracle_wavenet_num_weights.restype = C.c_size_t
    lib.na_oracle_wavenet_num_weights.argtypes = [C.c_int, C.POINTER(WnArrayCfg)]
    lib.na_oracle_wavenet_create.restype = C.c_void_p
    lib.na_oracle_wavenet_create.argtypes = [C.c_int, C.POINTER(WnArrayCfg), fp, C.c_size_t, C.c_int]
    lib.na_oracle_wavenet_free.argtypes = [C.c_void_p]
    lib.na_oracle_wavenet_receptive_field.restype = C.c_int
    lib.na_oracle_wavenet_receptive_field.argtypes = [C.c_void_p]
    lib.na_oracle_wavenet_reset.argtypes = [C.c_void_p]
    lib.na_oracle_wavenet_prewarm.argtypes = [C.c_void_p]
    lib.na_oracle_wavenet_process.argtypes = [C.c_void_p, fp, fp, C.c_size_t]
    lib.na_oracle_wavenet_set_max_frames.argtypes = [C.c_void_p, C.c_int]
    lib.na_oracle_lstm_create_nam.restype = C.c_void_p
    lib.na_oracle_lstm_create_nam.argtypes = [C.c_int, C.c_int, fp, C.c_size_t, C.c_int]
    lib.na_oracle_lstm_create_keras.restype = C.c_void_p
    lib.na_oracle_lstm_create_keras.argtypes = [C.c_int, C.c_int, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp), fp,
                                                C.c_float, C.c_int]
    lib.na_oracle_lstm_free.argtypes = [C.c_void_p]
    lib.na_oracle_lstm_prewarm.argtypes = [C.c_void_p]
    lib.na_oracle_lstm_process.argtypes = [C.c_void_p, fp, fp, C.c_size_t]
    lib.na_oracle_wavenet_bench.restype = C.c_double
    lib.na_oracle_wavenet_bench.argtypes = [C.c_int, C.POINTER(WnArrayCfg), fp, C.c_size_t, C.c_int, C.c_int, C.c_int]
    lib.na_oracle_lstm_bench.restype = C.c_double
    lib.na_oracle_lstm_bench.argtypes = [C.c_int, C.c_int, fp, C.c_size_t, C.c_int, C.c_int, C.c_int]
    # na_oracle_simd.c: the vectorised bench-only variant of the WaveNet path
    lib.na_oracle_simd_wavenet_create.restype = C.c_void_p
    lib.na_oracle_simd_wavenet_create.argtypes = [C.c_int, C.POINTER(WnArrayCfg), fp, C.c_size_t]
    lib.na_oracle_simd_wavenet_free.argtypes = [C.c_void_p]
    lib.na_oracle_simd_wavenet_process.restype = C.c_int
    lib.na_oracle_simd_wavenet_process.argtypes = [C.c_void_p, fp, fp, C.c_size_t]
    lib.na_oracle_simd_wavenet_bench.restype = C.c_double
    lib.na_oracle_simd_wavenet_bench.argtypes = [C.c_int, C.POINTER(WnArrayCfg), fp, C.c_size_t, C.c_int, C.c_int, C.c_int]
    return lib


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libna_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        _lib = _bind(C.CDLL(path))
    return _lib


def load_native_lib():
    """-O3 -march=native build, for cpu_baseline timing only (built on the box it is timed on)."""
    path = os.path.join(ORACLE_DIR, "libna_oracle_native.so")
    try:
        build_oracle(native=True)
        return _bind(C.CDLL(path))
    except Exception:
        return lib()


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_dense(w_colmajor, bias, x, out=None):
    """oracle 1x1: x [frames][cin] -> [frames][cout]; out given -> accumulate into it"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    frames, cin = x.shape
    w = np.ascontiguousarray(w_colmajor, dtype=np.float32)
    cout = w.size // cin
    acc = out is not None
    y = np.ascontiguousarray(out, dtype=np.float32).copy() if acc else np.zeros((frames, cout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    lib().na_oracle_test_dense(cin, cout, _fptr(w), None if b is None else _fptr(b), _fptr(x), _fptr(y), frames, 1 if acc else 0)
    return y


# ----------------------------------------------------------------------------- model-file reading

A1_STD = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
A2_KERNELS = [6] * 14 + [15, 15] + [6] * 7
A2_DILATIONS = [1, 3, 7, 17, 41, 101, 239, 1, 3, 7, 17, 41, 101, 239, 1, 13, 1, 3, 7, 17, 41, 101, 239]


def wavenet_arrays_from_nam(model_json):
    """Layer-array descriptions from a WaveNet .nam object (A1 'kernel_size' or A2 'kernel_sizes' form)."""
    arrays = []
    for lc in model_json["config"]["layers"]:
        dil = [int(d) for d in lc["dilations"]]
        if "kernel_sizes" in lc:  # A2 (NeuralModel.cpp:389-421; template args InternalModel.h:19-20)
            ks = [int(k) for k in lc["kernel_sizes"]]
            head = lc["head"]
            arrays.append(dict(input_size=int(lc["input_size"]), condition_size=int(lc["condition_size"]),
                               head_size=int(head["out_channels"]), head_kernel_size=int(head["kernel_size"]),
                               head_dilation=1, channels=int(lc["channels"]), has_head_bias=bool(head["bias"]),
                               activation=ACT_LEAKYRELU, kernel_sizes=ks, dilations=dil))
        else:
            ks = [int(lc["kernel_size"])] * len(dil)
            arrays.append(dict(input_size=int(lc["input_size"]), condition_size=int(lc["condition_size"]),
                               head_size=int(lc["head_size"]), head_kernel_size=1, head_dilation=1,
                               channels=int(lc["channels"]), has_head_bias=bool(lc["head_bias"]),
                               activation=ACT_TANH, kernel_sizes=ks, dilations=dil))
    return arrays


def a1_arrays(channels, head_size, lite=None):
    """Official A1 architectures (InternalModel.h:12-17,152-159)."""
    if lite is None:
        lite = channels != 16
    if not lite:
        d1, d2 = A1_STD, A1_STD
    else:
        d1, d2 = [1, 2, 4, 8, 16, 32, 64], [128, 256, 512, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    return [
        dict(input_size=1, condition_size=1, head_size=head_size, head_kernel_size=1, head_dilation=1,
             channels=channels, has_head_bias=False, activation=ACT_TANH, kernel_sizes=[3] * len(d1), dilations=d1),
        dict(input_size=channels, condition_size=1, head_size=1, head_kernel_size=1, head_dilation=1,
             channels=head_size, has_head_bias=True, activation=ACT_TANH, kernel_sizes=[3] * len(d2), dilations=d2),
    ]


def a2_arrays(channels):
    """A2 single-array architecture (NeuralModel.cpp:398,410)."""
    return [dict(input_size=1, condition_size=1, head_size=1, head_kernel_size=16, head_dilation=1, channels=channels,
                 has_head_bias=True, activation=ACT_LEAKYRELU, kernel_sizes=list(A2_KERNELS),
                 dilations=list(A2_DILATIONS))]


def _cfgs(arrays):
    arr = (WnArrayCfg * len(arrays))()
    for i, a in enumerate(arrays):
        c = arr[i]
        for k in ("input_size", "condition_size", "head_size", "head_kernel_size", "head_dilation", "channels"):
            setattr(c, k, int(a[k]))
        c.has_head_bias = int(bool(a["has_head_bias"]))
        c.activation = int(a["activation"])
        c.num_layers = len(a["dilations"])
        for j, (k, d) in enumerate(zip(a["kernel_sizes"], a["dilations"])):
            c.kernel_sizes[j] = int(k)
            c.dilations[j] = int(d)
    return arr


def wavenet_num_weights(arrays):
    return int(lib().na_oracle_wavenet_num_weights(len(arrays), _cfgs(arrays)))


class OracleWaveNet:
    def __init__(self, arrays, weights, math_mode=MATH_FAST, prewarm=True):
        self.arrays = arrays
        self.weights = np.ascontiguousarray(weights, dtype=np.float32)
        self._cfgs = _cfgs(arrays)
        self._h = lib().na_oracle_wavenet_create(len(arrays), self._cfgs, _fptr(self.weights), self.weights.size,
                                                 math_mode)
        if not self._h:
            raise ValueError("Wrong number of weights")
        if prewarm:
            self.prewarm()

    @property
    def receptive_field(self):
        return int(lib().na_oracle_wavenet_receptive_field(self._h))

    def reset(self):
        lib().na_oracle_wavenet_reset(self._h)

    def prewarm(self):
        lib().na_oracle_wavenet_prewarm(self._h)

    def set_max_frames(self, n):
        lib().na_oracle_wavenet_set_max_frames(self._h, n)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        lib().na_oracle_wavenet_process(self._h, _fptr(x), _fptr(y), x.size)
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            lib().na_oracle_wavenet_free(self._h)
            self._h = None


class OracleWaveNetSimd:
    """oracle/na_oracle_simd.c: the vectorised variant timed by bench.py's cpu_baseline leg (not the checker); always prewarmed."""
    def __init__(self, arrays, weights, native=False):
        self._lib = load_native_lib() if native else lib()
        self.weights = np.ascontiguousarray(weights, dtype=np.float32)
        self._cfgs = _cfgs(arrays)
        self._h = self._lib.na_oracle_simd_wavenet_create(len(arrays), self._cfgs, _fptr(self.weights), self.weights.size)
        if not self._h:
            raise ValueError("Wrong number of weights")

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        if self._lib.na_oracle_simd_wavenet_process(self._h, _fptr(x), _fptr(y), x.size) != 0:
            raise ValueError("the vectorised variant takes multiples of 8 samples")
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.na_oracle_simd_wavenet_free(self._h)
            self._h = None


class OracleLSTM:
    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_nam(cls, num_layers, hidden, weights, math_mode=MATH_FAST, prewarm=True):
        w = np.ascontiguousarray(weights, dtype=np.float32)
        h = lib().na_oracle_lstm_create_nam(num_layers, hidden, _fptr(w), w.size, math_mode)
        if not h:
            raise ValueError("Wrong number of weights")
        m = cls(h)
        if prewarm:
            m.prewarm()
        return m

    @classmethod
    def from_keras(cls, model_json, math_mode=MATH_FAST, prewarm=True):
        """keras / AIDA-X json (InternalModel.h:311-356)."""
        layers = model_json["layers"]
        nl = len(layers) - 1
        hidden = int(layers[0]["shape"][-1])
        fp = C.POINTER(C.c_float)
        ks = [np.ascontiguousarray(np.array(layers[i]["weights"][0], dtype=np.float32).ravel()) for i in range(nl)]
        rs = [np.ascontiguousarray(np.array(layers[i]["weights"][1], dtype=np.float32).ravel()) for i in range(nl)]
        bs = [np.ascontiguousarray(np.array(layers[i]["weights"][2], dtype=np.float32).ravel()) for i in range(nl)]
        hw = np.ascontiguousarray(np.array(layers[-1]["weights"][0], dtype=np.float32).ravel())
        hb = float(layers[-1]["weights"][1][0])
        kp = (fp * nl)(*[_fptr(a) for a in ks])
        rp = (fp * nl)(*[_fptr(a) for a in rs])
        bp = (fp * nl)(*[_fptr(a) for a in bs])
        h = lib().na_oracle_lstm_create_keras(nl, hidden, kp, rp, bp, _fptr(hw), hb, math_mode)
        m = cls(h)
        if prewarm:
            m.prewarm()
        return m

    def prewarm(self):
        lib().na_oracle_lstm_prewarm(self._h)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        lib().na_oracle_lstm_process(self._h, _fptr(x), _fptr(y), x.size)
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            lib().na_oracle_lstm_free(self._h)
            self._h = None


class OracleGRU:
    """keras GRU (reset_after form); RTNeural's arithmetic in the reference -- parity unpinned (see oracle/na_oracle.c)."""

    def __init__(self, model_json, prewarm=True):
        layers = model_json["layers"]
        nl = len(layers) - 1
        hidden = int(layers[0]["shape"][-1])
        fp = C.POINTER(C.c_float)
        self._keep = [[np.ascontiguousarray(np.array(layers[i]["weights"][k], dtype=np.float32).ravel()) for i in range(nl)] for k in range(3)]
        hw = np.ascontiguousarray(np.array(layers[-1]["weights"][0], dtype=np.float32).ravel())
        hb = float(layers[-1]["weights"][1][0])
        ptrs = [(fp * nl)(*[_fptr(a) for a in self._keep[k]]) for k in range(3)]
        L = lib()
        L.na_oracle_gru_create_keras.restype = C.c_void_p
        L.na_oracle_gru_create_keras.argtypes = [C.c_int, C.c_int, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp), fp, C.c_float]
        L.na_oracle_gru_process.argtypes = [C.c_void_p, fp, fp, C.c_size_t]
        L.na_oracle_gru_prewarm.argtypes = [C.c_void_p]
        L.na_oracle_gru_free.argtypes = [C.c_void_p]
        self._h = L.na_oracle_gru_create_keras(nl, hidden, ptrs[0], ptrs[1], ptrs[2], _fptr(hw), hb)
        if prewarm:
            self.prewarm()

    def prewarm(self):
        lib().na_oracle_gru_prewarm(self._h)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        lib().na_oracle_gru_process(self._h, _fptr(x), _fptr(y), x.size)
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            lib().na_oracle_gru_free(self._h)
            self._h = None


def synth_keras_gru(num_layers, hidden, seed):
    """A keras/AIDA-X style GRU model json with seeded U(-a, a) weights, a = 1/sqrt(hidden) (no GRU file ships with the reference)."""
    rng = np.random.default_rng(seed)
    a = 1.0 / np.sqrt(hidden)
    layers = []
    for l in range(num_layers):
        i = 1 if l == 0 else hidden
        layers.append({"type": "gru", "activation": "", "shape": [None, None, hidden],
                       "weights": [rng.uniform(-a, a, (i, 3 * hidden)).round(7).tolist(),
                                   rng.uniform(-a, a, (hidden, 3 * hidden)).round(7).tolist(),
                                   rng.uniform(-a, a, (2, 3 * hidden)).round(7).tolist()]})
    layers.append({"type": "dense", "activation": "", "shape": [None, None, 1],
                   "weights": [rng.uniform(-a, a, (hidden, 1)).round(7).tolist(), [float(np.round(rng.uniform(-a, a), 7))]]})
    return {"in_shape": [None, None, 1], "in_skip": 0, "samplerate": 48000.0, "layers": layers}


def load_json(name):
    path = name if os.path.isabs(name) else os.path.join(MODELS_DIR, name)
    with open(path) as f:
        return json.load(f)


def quality_to_submodel(container_json, quality):
    """CompositeModel.h:200-213: first sorted level with quality <= max_value, else the last."""
    levels = sorted(((float(s["max_value"]), i) for i, s in enumerate(container_json["config"]["submodels"])),
                    key=lambda t: t[0])
    idx = 0
    for mv, i in levels:
        idx = i
        if quality <= mv:
            break
    return idx


def oracle_from_file(name, quality=1.0, math_mode=MATH_FAST, prewarm=True):
    """Build the oracle model the reference loader would pick for this file (Internal path)."""
    j = load_json(name)
    if name.endswith(".nam"):
        if j["architecture"] == "SlimmableContainer":
            j = j["config"]["submodels"][quality_to_submodel(j, quality)]["model"]
        if j["architecture"] == "WaveNet":
            return OracleWaveNet(wavenet_arrays_from_nam(j), j["weights"], math_mode, prewarm)
        if j["architecture"] == "LSTM":
            c = j["config"]
            return OracleLSTM.from_nam(int(c["num_layers"]), int(c["hidden_size"]), j["weights"], math_mode, prewarm)
        raise ValueError("unsupported architecture " + j["architecture"])
    if j["layers"][0]["type"] == "gru":
        return OracleGRU(j, prewarm)
    return OracleLSTM.from_keras(j, math_mode, prewarm)


# ----------------------------------------------------------------------------- synthetic inputs / weights

def signal_sine(n, start=0):
    """ModelTest.cpp:103 -- (float)sin(pos * 0.01)"""
    return np.sin(np.arange(start, start + n, dtype=np.float64) * 0.01).astype(np.float32)


def signal_noise(n, seed):
    rng = np.random.default_rng(seed)
    return np.clip(0.25 * rng.standard_normal(n), -1.0, 1.0).astype(np.float32)


def synth_wavenet_weights(arrays, seed):
    """Seeded U(-a, a), a = 1/sqrt(fan_in) per tensor; last weight (head_scale) = 0.02."""
    rng = np.random.default_rng(seed)
    out = []

    def u(n, fan_in):
        a = 1.0 / np.sqrt(max(fan_in, 1))
        out.append(rng.uniform(-a, a, size=n))

    for a in arrays:
        c = a["channels"]
        u(c * a["input_size"], a["input_size"])
        for k in a["kernel_sizes"]:
            u(c * c * k, c * k)
            u(c, c * k)
            u(c * a["condition_size"], 1)
            u(c * c, c)
            u(c, c)
        u(a["head_size"] * c * a["head_kernel_size"], c * a["head_kernel_size"])
        if a["has_head_bias"]:
            u(a["head_size"], c)
    out.append(np.array([0.02]))
    return np.concatenate(out).astype(np.float32)


def wavenet_tensor_slices(arrays):
    """(name, array index, layer index or -1, slice into the flat weights) of every tensor, in the reference's order (WaveNet.h:700-719)."""
    out, pos = [], 0

    def take(name, a, l, n):
        nonlocal pos
        out.append((name, a, l, slice(pos, pos + n)))
        pos += n

    for ai, a in enumerate(arrays):
        c = a["channels"]
        take("rechannel", ai, -1, c * a["input_size"])
        for li, k in enumerate(a["kernel_sizes"]):
            take("conv", ai, li, c * c * k)
            take("conv_bias", ai, li, c)
            take("mixin", ai, li, c * a["condition_size"])
            take("1x1", ai, li, c * c)
            take("1x1_bias", ai, li, c)
        take("head", ai, -1, a["head_size"] * c * a["head_kernel_size"])
        if a["has_head_bias"]:
            take("head_bias", ai, -1, a["head_size"])
    take("head_scale", -1, -1, 1)
    return out


def scale_wavenet_tensors(arrays, weights, factors):
    """A copy of the flat weights with every tensor called `name` multiplied by factors[name]."""
    w = np.array(weights, dtype=np.float32, copy=True)
    for name, _, _, sl in wavenet_tensor_slices(arrays):
        if name in factors:
            w[sl] *= np.float32(factors[name])
    return w


def synth_lstm_weights(num_layers, hidden, seed):
    rng = np.random.default_rng(seed)
    out = []
    for l in range(num_layers):
        i = 1 if l == 0 else hidden
        a = 1.0 / np.sqrt(hidden)
        out.append(rng.uniform(-a, a, size=4 * hidden * (i + hidden)))
        out.append(rng.uniform(-a, a, size=4 * hidden))
        out.append(rng.uniform(-0.5, 0.5, size=hidden))  # initial hidden
        out.append(rng.uniform(-0.5, 0.5, size=hidden))  # initial cell
    out.append(rng.uniform(-0.5, 0.5, size=hidden))
    out.append(rng.uniform(-0.1, 0.1, size=1))
    return np.concatenate(out).astype(np.float32)


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))


# ----------------------------------------------------------------------------- synthetic .nam documents

def nam_json_wavenet_a1(channels, head_size, weights, lite=None):
    """A .nam document for an official A1 architecture with the given flat weights (loader input)."""
    arrays = a1_arrays(channels, head_size, lite)
    layers = []
    for a in arrays:
        layers.append({"input_size": a["input_size"], "condition_size": 1, "head_size": a["head_size"], "channels": a["channels"],
                       "kernel_size": 3, "dilations": list(a["dilations"]), "activation": "Tanh", "gated": False,
                       "head_bias": bool(a["has_head_bias"])})
    return json.dumps({"version": "0.5.4", "architecture": "WaveNet", "metadata": {"loudness": -10.0},
                       "config": {"layers": layers, "head": None, "head_scale": 0.02},
                       "weights": [float(w) for w in weights], "sample_rate": 48000})


def nam_json_wavenet_generic(arrays, weights):
    """A .nam document for ANY array list in the oracle's dict format: A1 form when every layer of an array shares one kernel size and
    the head is 1x1, A2 form ('kernel_sizes' + 'head' block) otherwise.  This is what the reference's dynamic path reads
    (WaveNetDynamic.h / InternalModel.h:177-248)."""
    layers = []
    for a in arrays:
        a2_form = len(set(a["kernel_sizes"])) > 1 or a["head_kernel_size"] != 1 or a["activation"] == ACT_LEAKYRELU
        if a2_form:
            act = {"type": "LeakyReLU", "negative_slope": 0.01} if a["activation"] == ACT_LEAKYRELU else {"type": "Tanh"}
            layers.append({"input_size": a["input_size"], "condition_size": 1, "channels": a["channels"], "kernel_sizes": list(a["kernel_sizes"]),
                           "dilations": list(a["dilations"]), "activation": [act] * len(a["dilations"]),
                           "head": {"out_channels": a["head_size"], "kernel_size": a["head_kernel_size"], "bias": bool(a["has_head_bias"])}})
        else:
            layers.append({"input_size": a["input_size"], "condition_size": 1, "head_size": a["head_size"], "channels": a["channels"],
                           "kernel_size": a["kernel_sizes"][0], "dilations": list(a["dilations"]), "activation": "Tanh", "gated": False,
                           "head_bias": bool(a["has_head_bias"])})
    return json.dumps({"version": "0.5.4", "architecture": "WaveNet", "metadata": {"loudness": -10.0},
                       "config": {"layers": layers, "head": None, "head_scale": 0.02},
                       "weights": [float(w) for w in weights], "sample_rate": 48000})


def nam_json_lstm(num_layers, hidden, weights):
    return json.dumps({"version": "0.5.4", "architecture": "LSTM", "metadata": {"loudness": -12.0},
                       "config": {"input_size": 1, "hidden_size": hidden, "num_layers": num_layers},
                       "weights": [float(w) for w in weights]})
