"""Independent float64 numpy restatement of the WaveNet / LSTM math -- TEST INFRASTRUCTURE ONLY.

Written from the algorithm description, NOT from oracle/na_oracle.c:
whole-signal (non-streaming, no rings, no chunks), vectorised over time, float64 accumulate.
It exists so the C oracle's ring / chunk / prewarm bookkeeping is checked by something that
has none of it.  The initial state is emulated by a zero lead-in longer than the receptive
field, which is what the reference's analytic Prewarm (WaveNet.h:746-766) is equivalent to.
"""
import numpy as np


def fast_tanh(x):
    """Activation.h:83-91 in float64 (rational approximation, not libm)."""
    ax = np.abs(x)
    x2 = x * x
    return (x * (2.45550750702956 + 2.45550750702956 * ax + (0.893229853513558 + 0.821226666969744 * ax) * x2)
            / (2.44506634652299 + (2.44506634652299 + x2) * np.abs(x + 0.814642734961073 * x * ax)))


def fast_sigmoid(x):
    return 0.5 * (fast_tanh(0.5 * x) + 1.0)


def leaky_relu(x):
    return np.where(x > 0, x, 0.01 * x)


def _shift(h, s):
    """h[:, t] -> h[:, t - s] with zeros flowing in (s >= 0)."""
    if s == 0:
        return h
    out = np.zeros_like(h)
    if s < h.shape[1]:
        out[:, s:] = h[:, :-s]
    return out


def wavenet_forward(arrays, weights, x, lead_in=None, tanh=fast_tanh):
    """Returns y for signal x assuming the model was prewarmed (zero-input steady state) before x."""
    w = np.asarray(weights, dtype=np.float64)
    pos = 0

    def take(n):
        nonlocal pos
        v = w[pos:pos + n]
        assert v.size == n, "ran out of weights"
        pos += n
        return v

    rf = sum((k - 1) * d for a in arrays for k, d in zip(a["kernel_sizes"], a["dilations"]))
    rf += sum((a["head_kernel_size"] - 1) * a["head_dilation"] for a in arrays)
    if lead_in is None:
        lead_in = rf + 8
    sig = np.concatenate([np.zeros(lead_in), np.asarray(x, dtype=np.float64)])
    T = sig.size
    cond = sig[None, :]
    layer_in = cond
    head = None
    for a in arrays:
        c = a["channels"]
        act = leaky_relu if a["activation"] == 1 else tanh
        w_re = take(c * a["input_size"]).reshape(c, a["input_size"])
        h = w_re @ layer_in
        if head is None:
            head = np.zeros((c, T))
        for k, d in zip(a["kernel_sizes"], a["dilations"]):
            wc = take(c * c * k).reshape(c, c, k)
            bc = take(c)
            wm = take(c * a["condition_size"]).reshape(c, a["condition_size"])
            w1 = take(c * c).reshape(c, c)
            b1 = take(c)
            z = bc[:, None] + wm @ cond
            for tap in range(k):
                z = z + wc[:, :, tap] @ _shift(h, d * (k - 1 - tap))
            z = act(z)
            head = head + z
            h = w1 @ z + b1[:, None] + h
        kh = a["head_kernel_size"]
        wh = take(a["head_size"] * c * kh).reshape(a["head_size"], c, kh)
        out = np.zeros((a["head_size"], T))
        for tap in range(kh):
            out = out + wh[:, :, tap] @ _shift(head, a["head_dilation"] * (kh - 1 - tap))
        if a["has_head_bias"]:
            out = out + take(a["head_size"])[:, None]
        head = out
        layer_in = h
    scale = take(1)[0]
    assert pos == w.size, "weights left over"
    return (scale * head[0])[lead_in:], rf


def lstm_forward_nam(num_layers, hidden, weights, x, prewarm=2048, tanh=fast_tanh, sigmoid=fast_sigmoid):
    w = np.asarray(weights, dtype=np.float64)
    pos = 0

    def take(n):
        nonlocal pos
        v = w[pos:pos + n]
        pos += n
        return v

    layers = []
    for l in range(num_layers):
        i = 1 if l == 0 else hidden
        W = take(4 * hidden * (i + hidden)).reshape(4 * hidden, i + hidden)
        b = take(4 * hidden)
        h0 = take(hidden).copy()
        c0 = take(hidden).copy()
        layers.append([W, b, h0, c0])
    wh = take(hidden)
    bh = take(1)[0]
    assert pos == w.size
    return _lstm_run(layers, wh, bh, hidden, x, prewarm, tanh, sigmoid)


def lstm_forward_keras(model_json, x, prewarm=2048, tanh=fast_tanh, sigmoid=fast_sigmoid):
    ls = model_json["layers"]
    hidden = int(ls[0]["shape"][-1])
    layers = []
    for l in ls[:-1]:
        kernel = np.array(l["weights"][0], dtype=np.float64)      # [I][4H]
        recurrent = np.array(l["weights"][1], dtype=np.float64)   # [H][4H]
        bias = np.array(l["weights"][2], dtype=np.float64).ravel()
        W = np.concatenate([kernel.T, recurrent.T], axis=1)
        layers.append([W, bias, np.zeros(hidden), np.zeros(hidden)])
    wh = np.array(ls[-1]["weights"][0], dtype=np.float64).ravel()
    bh = float(ls[-1]["weights"][1][0])
    return _lstm_run(layers, wh, bh, hidden, x, prewarm, tanh, sigmoid)


def _lstm_run(layers, wh, bh, H, x, prewarm, tanh, sigmoid):
    sig = np.concatenate([np.zeros(prewarm), np.asarray(x, dtype=np.float64)])
    y = np.empty(sig.size)
    for t in range(sig.size):
        inp = np.array([sig[t]])
        for L in layers:
            W, b, h, c = L
            g = W @ np.concatenate([inp, h]) + b
            c = sigmoid(g[H:2 * H]) * c + sigmoid(g[0:H]) * tanh(g[2 * H:3 * H])
            h = sigmoid(g[3 * H:4 * H]) * tanh(c)
            L[2], L[3] = h, c
            inp = h
        y[t] = wh @ inp + bh
    return y[prewarm:]


def std_sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_forward_keras(model_json, x, prewarm=2048):
    """keras GRU(reset_after=True) / RTNeural GRULayer in float64 with exact tanh; zero initial state, `prewarm` zeros first."""
    layers = model_json["layers"]
    nl = len(layers) - 1
    H = int(layers[0]["shape"][-1])
    K = [np.array(layers[i]["weights"][0], dtype=np.float64) for i in range(nl)]  # [I][3H]
    U = [np.array(layers[i]["weights"][1], dtype=np.float64) for i in range(nl)]  # [H][3H]
    B = [np.array(layers[i]["weights"][2], dtype=np.float64) for i in range(nl)]  # [2][3H]
    wh = np.array(layers[-1]["weights"][0], dtype=np.float64).ravel()
    bh = float(layers[-1]["weights"][1][0])
    h = [np.zeros(H) for _ in range(nl)]
    xs = np.concatenate([np.zeros(prewarm), np.asarray(x, dtype=np.float64)])
    out = np.empty(xs.size)
    sig = lambda v: 0.5 * (np.tanh(0.5 * v) + 1.0)
    for t, xv in enumerate(xs):
        inp = np.array([xv])
        for l in range(nl):
            ai = inp @ K[l] + B[l][0]
            ah = h[l] @ U[l] + B[l][1]
            z = sig(ai[:H] + ah[:H])
            r = sig(ai[H:2 * H] + ah[H:2 * H])
            c = np.tanh(ai[2 * H:] + r * ah[2 * H:])
            h[l] = (1.0 - z) * c + z * h[l]
            inp = h[l]
        out[t] = wh @ h[-1] + bh
    return out[prewarm:]


def keras_stack_forward(model_json, x, prewarm=2048):
    """Generic keras stack [lstm | gru]* + dense+ (what the reference hands to RTNeural's json_parser, RTNeuralModel.h:300), in float64
    with the Keras layer definitions and accurate tanh / sigmoid = (tanh(x/2)+1)/2 (the reference's FastMathsProvider,
    RTNeuralModel.h:10-31).  Output = unit 0 of the last layer; zero initial state, `prewarm` zeros first."""
    layers = model_json["layers"]
    sig = lambda v: 0.5 * (np.tanh(0.5 * v) + 1.0)
    def softmax(v):
        e = np.exp(v - np.max(v))
        return e / np.sum(e)
    acts = {"": lambda v: v, "linear": lambda v: v, "tanh": np.tanh, "relu": lambda v: np.maximum(v, 0.0), "sigmoid": sig,
            "elu": lambda v: np.where(v > 0, v, np.exp(np.minimum(v, 0.0)) - 1.0), "softmax": softmax}
    lastint = lambda v: int(v[-1]) if isinstance(v, (list, tuple)) else int(v)
    state = []
    width = 1
    for l in layers:
        H = int(l["shape"][-1])
        if l["type"] in ("lstm", "gru"):
            state.append([np.zeros(H), np.zeros(H)])
        elif l["type"] == "conv1d":  # the layer's input history, newest last: (kernel_size - 1) * dilation + 1 rows (zero after reset)
            state.append(np.zeros(((lastint(l["kernel_size"]) - 1) * lastint(l.get("dilation", 1)) + 1, width)))
        else:
            state.append(None)
        width = H
    W = [[np.array(w, dtype=np.float64) for w in l["weights"]] for l in layers]
    xs = np.concatenate([np.zeros(prewarm), np.asarray(x, dtype=np.float64)])
    out = np.empty(xs.size)
    for t, xv in enumerate(xs):
        v = np.array([xv])
        for i, l in enumerate(layers):
            H = int(l["shape"][-1])
            if l["type"] == "lstm":
                h, c = state[i]
                g = v @ W[i][0] + h @ W[i][1] + W[i][2].ravel()  # gate column blocks i, f, g(c), o
                c = sig(g[H:2 * H]) * c + sig(g[:H]) * np.tanh(g[2 * H:3 * H])
                h = sig(g[3 * H:]) * np.tanh(c)
                state[i] = [h, c]
                v = h
            elif l["type"] == "gru":
                h = state[i][0]
                ai = v @ W[i][0] + W[i][2][0]
                ah = h @ W[i][1] + W[i][2][1]
                z = sig(ai[:H] + ah[:H])
                r = sig(ai[H:2 * H] + ah[H:2 * H])
                c = np.tanh(ai[2 * H:] + r * ah[2 * H:])
                h = (1.0 - z) * c + z * h
                state[i][0] = h
                v = h
            elif l["type"] == "conv1d":
                # Keras Conv1D(padding="causal", strides=1): y[t] = b + sum_k x[t - (K - 1 - k) d] @ W[k]   (weights [K][in][out], [out])
                K, d = lastint(l["kernel_size"]), lastint(l.get("dilation", 1))
                state[i] = np.vstack([state[i][1:], v[None, :]])
                acc = W[i][1].ravel().copy()
                for k in range(K):
                    acc = acc + state[i][-1 - (K - 1 - k) * d] @ W[i][0][k]
                v = acts[l.get("activation", "") or ""](acc)
            elif l["type"] == "activation":
                v = acts[l.get("activation", "") or ""](v)
            elif l["type"] == "batchnorm":  # inference form; [gamma, beta, mean, var] or [mean, var]; keras default epsilon
                eps = float(l.get("epsilon", 1e-3))
                g, b, m, s2 = (W[i][0].ravel(), W[i][1].ravel(), W[i][2].ravel(), W[i][3].ravel()) if len(W[i]) >= 4 else (1.0, 0.0, W[i][0].ravel(), W[i][1].ravel())
                v = g * (v - m) / np.sqrt(s2 + eps) + b
            elif l["type"] == "prelu":
                alpha = W[i][0].ravel()
                v = np.maximum(v, 0.0) + alpha * np.minimum(v, 0.0)
            else:
                v = acts[l.get("activation", "") or ""](v @ W[i][0] + W[i][1].ravel())
        out[t] = v[0]
    return out[prewarm:]


def synth_keras_stack(spec, seed):
    """spec: list of ("lstm" | "gru" | "dense", units[, activation]) | ("conv1d", units, kernel_size, dilation[, activation]) and, behind a layer
    of `units` units, ("activation", units, name) | ("batchnorm", units[, "noaffine"]) | ("prelu", units[, "scalar"]); seeded U(-a, a) weights,
    a = 1/sqrt(fan-in-ish)."""
    rng = np.random.default_rng(seed)
    layers = []
    cur = 1
    for item in spec:
        kind, units = item[0], item[1]
        a = 1.0 / np.sqrt(max(units, cur))
        if kind in ("lstm", "gru"):
            g = 4 if kind == "lstm" else 3
            bias = rng.uniform(-a, a, (g * units,)) if kind == "lstm" else rng.uniform(-a, a, (2, g * units))
            layers.append({"type": kind, "activation": "tanh", "shape": [None, None, units],
                           "weights": [rng.uniform(-a, a, (cur, g * units)).round(7).tolist(), rng.uniform(-a, a, (units, g * units)).round(7).tolist(),
                                       bias.round(7).tolist()]})
        elif kind == "activation":
            layers.append({"type": "activation", "activation": item[2], "shape": [None, None, units], "weights": []})
        elif kind == "batchnorm":
            mean, var = rng.uniform(-0.3, 0.3, units).round(7).tolist(), rng.uniform(0.2, 1.5, units).round(7).tolist()
            w = [mean, var] if len(item) > 2 and item[2] == "noaffine" else [rng.uniform(0.5, 1.5, units).round(7).tolist(), rng.uniform(-0.2, 0.2, units).round(7).tolist(), mean, var]
            layers.append({"type": "batchnorm", "epsilon": 0.001, "shape": [None, None, units], "weights": w})
        elif kind == "prelu":
            alpha = [float(rng.uniform(0.05, 0.4))] if len(item) > 2 and item[2] == "scalar" else rng.uniform(0.05, 0.4, units).round(7).tolist()
            layers.append({"type": "prelu", "shape": [None, None, units], "weights": [alpha]})
        elif kind == "conv1d":  # ("conv1d", units, kernel_size, dilation[, activation])
            K, d = item[2], item[3]
            a = 1.0 / np.sqrt(max(units, cur * K))
            layers.append({"type": "conv1d", "activation": item[4] if len(item) > 4 else "", "shape": [None, None, units], "kernel_size": [K], "dilation": [d],
                           "weights": [rng.uniform(-a, a, (K, cur, units)).round(7).tolist(), rng.uniform(-a, a, (units,)).round(7).tolist()]})
        else:
            layers.append({"type": "dense", "activation": item[2] if len(item) > 2 else "", "shape": [None, None, units],
                           "weights": [rng.uniform(-a, a, (cur, units)).round(7).tolist(), rng.uniform(-a, a, (units,)).round(7).tolist()]})
        cur = units
    return {"in_shape": [None, None, 1], "in_skip": 0, "samplerate": 48000.0, "layers": layers}
