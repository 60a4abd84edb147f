#!/usr/bin/env python3
"""Regenerates the committed golden fixtures.  Run in the build container (where /root/reference exists).

  matmul_ref.npz      REFERENCE OUTPUTS: the reference's own NeuralAudio/MatMul.h (compiled unmodified into
                      oracle/_ref/libna_ref_matmul.so by `make -C oracle ref`) applied to seeded inputs.
                      These pin the oracle's tiny mat-mul / conv-tap arithmetic against the reference itself.
  oracle_outputs.npz  ORACLE REGRESSION VECTORS (not reference outputs): first 1024 output samples of the C
                      oracle for every sample model on sin(0.01 n) after prewarm.  They make the oracle's
                      behaviour reproducible on the GPU box and catch accidental edits; they do not pin parity
                      (see oracle/na_oracle.h for what does).
  fixture_matrix.npz  THE PARITY MATRIX OF SURVEY.md 8(c): {Standard, Feather, Nano, A2 q=0 (3 ch), A2 q=1 (8 ch), LSTM 1x16, LSTM 2x8}
                      x {sin(0.01 n), seeded clipped noise, zeros} x 4096 samples after prewarm, in two independent evaluations:
                      "np64/<model>/<input>" = tests/ref_np.py (float64 numpy restatement of SURVEY Appendix B, whole-signal, no rings /
                      chunks / prewarm code), "oracle/<model>/<input>" = the C oracle (f32, streaming).  Plus "switch/*": one A2 stream
                      whose quality flips mid-stream (each submodel only advances while active, CompositeModel.h:94-100).  The reference
                      cannot be built here (no Eigen), so these are NOT reference outputs; they are what the GPU tests hold the HIP path
                      to in addition to the live oracle, so that kernel and oracle cannot drift together unnoticed.
  keras_stacks_torch.npz  the same for five generic keras stacks (lstm / gru / dense / conv1d chains, softmax): torch.nn.LSTM / GRU / Linear / Conv1d, float64
  gru_torch.npz       INDEPENDENT-IMPLEMENTATION VECTORS for the keras GRU (RTNeural is absent from the reference tree, so there is no
                      reference output to record): output of torch.nn.GRU (float64, weights permuted from keras z|r|c to torch r|z|n order)
                      + dense head on the committed synthetic model models/synthetic_gru_1x16.json, after 2048 zeros of prewarm.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import na_oracle as O  # noqa: E402


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_matmul():
    ref = C.CDLL(os.path.join(O.ORACLE_DIR, "_ref", "libna_ref_matmul.so"))
    rng = np.random.default_rng(20260928)
    out = {}
    frames = 37
    for cin, cout in [(3, 3), (8, 1), (3, 1), (1, 3)]:
        x = rng.uniform(-1, 1, size=(frames, cin)).astype(np.float32)
        w = rng.uniform(-1, 1, size=(cin * cout,)).astype(np.float32)   # column-major W(i,j) = w[j*cout + i]
        w2 = rng.uniform(-1, 1, size=(cin * cout,)).astype(np.float32)
        init = rng.uniform(-1, 1, size=(cout,)).astype(np.float32)
        tag = "%d_%d" % (cin, cout)
        y0 = np.zeros((frames, cout), np.float32)
        getattr(ref, "na_ref_matmul_init_zero_" + tag)(fptr(x), fptr(y0), fptr(w), C.c_size_t(frames))
        y1 = np.zeros((frames, cout), np.float32)
        getattr(ref, "na_ref_matmul_init_colwise_" + tag)(fptr(x), fptr(y1), fptr(w), fptr(init), C.c_size_t(frames))
        y2 = y1.copy()
        getattr(ref, "na_ref_matmul_accumulate_" + tag)(fptr(x), fptr(y2), fptr(w2), C.c_size_t(frames))
        out.update({"x_" + tag: x, "w_" + tag: w, "w2_" + tag: w2, "init_" + tag: init, "zero_" + tag: y0,
                    "colwise_" + tag: y1, "acc_" + tag: y2})
    np.savez_compressed(os.path.join(HERE, "matmul_ref.npz"), **out)


def make_oracle_outputs():
    x = O.signal_sine(1024)
    out = {"input": x}
    for name, q in [("BossWN-standard.nam", 1.0), ("BossWN-feather.nam", 1.0), ("BossWN-nano.nam", 1.0),
                    ("BossWN-a2.nam", 0.0), ("BossWN-a2.nam", 1.0), ("BossLSTM-1x16.nam", 1.0), ("BossLSTM-2x8.nam", 1.0),
                    ("tw40_blues_deluxe_deerinkstudios.json", 1.0)]:
        m = O.oracle_from_file(name, quality=q)
        out["%s@q%.1f" % (name, q)] = m.process(x)
    np.savez_compressed(os.path.join(HERE, "oracle_outputs.npz"), **out)


def make_gru():
    import json
    import torch
    H = 16
    j = O.synth_keras_gru(1, H, seed=20260928)
    with open(os.path.join(HERE, "models", "synthetic_gru_1x16.json"), "w") as f:
        json.dump(j, f)
    x = O.signal_sine(1024)
    gru = torch.nn.GRU(1, H, num_layers=1, batch_first=True).double()
    perm = np.concatenate([np.arange(H, 2 * H), np.arange(0, H), np.arange(2 * H, 3 * H)])
    with torch.no_grad():
        k, u, b = (np.array(j["layers"][0]["weights"][i], dtype=np.float64) for i in range(3))
        gru.weight_ih_l0.copy_(torch.from_numpy(k.T[perm].copy()))
        gru.weight_hh_l0.copy_(torch.from_numpy(u.T[perm].copy()))
        gru.bias_ih_l0.copy_(torch.from_numpy(b[0][perm].copy()))
        gru.bias_hh_l0.copy_(torch.from_numpy(b[1][perm].copy()))
        xs = torch.from_numpy(np.concatenate([np.zeros(2048), x.astype(np.float64)])).reshape(1, -1, 1)
        hs, _ = gru(xs)
        wh = torch.from_numpy(np.array(j["layers"][-1]["weights"][0], dtype=np.float64).ravel())
        y = (hs[0] @ wh + float(j["layers"][-1]["weights"][1][0])).numpy()[2048:]
    np.savez_compressed(os.path.join(HERE, "gru_torch.npz"), input=x, output=y.astype(np.float32))


KERAS_STACKS = {"lstm8_dense6tanh_dense1": [("lstm", 8), ("dense", 6, "tanh"), ("dense", 1)],
                "gru12_dense5relu_dense3sigmoid_dense1": [("gru", 12), ("dense", 5, "relu"), ("dense", 3, "sigmoid"), ("dense", 1)],
                "dense8tanh_dense4elu_dense1": [("dense", 8, "tanh"), ("dense", 4, "elu"), ("dense", 1)],
                # round 5: causal dilated conv1d layers (torch.nn.Conv1d behind a left pad) and softmax
                "conv8k3d1tanh_conv8k3d2tanh_conv4k2d4relu_dense1": [("conv1d", 8, 3, 1, "tanh"), ("conv1d", 8, 3, 2, "tanh"), ("conv1d", 4, 2, 4, "relu"), ("dense", 1)],
                "gru12_conv16k4d64elu_dense5softmax_dense1": [("gru", 12), ("conv1d", 16, 4, 64, "elu"), ("dense", 5, "softmax"), ("dense", 1)]}


def make_keras_stacks():
    """INDEPENDENT-IMPLEMENTATION VECTORS for generic keras stacks (RTNeural is absent, so there is no reference output to
    record): torch.nn.LSTM / GRU / Linear in float64 on committed synthetic models, zero state, 2048 zeros of prewarm."""
    import json
    import torch
    import ref_np as R
    acts = {"": lambda v: v, "tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, "elu": torch.nn.functional.elu,
            "softmax": lambda v: torch.softmax(v, dim=-1)}
    x = O.signal_noise(1024, 20260929)
    out = {"input": x}
    for name, spec in KERAS_STACKS.items():
        j = R.synth_keras_stack(spec, seed=len(name))
        with open(os.path.join(HERE, "models", "synthetic_stack_%s.json" % name), "w") as f:
            json.dump(j, f)
        with torch.no_grad():
            v = torch.from_numpy(np.concatenate([np.zeros(2048), x.astype(np.float64)])).reshape(1, -1, 1)
            for l in j["layers"]:
                H = int(l["shape"][-1])
                W = [np.array(w, dtype=np.float64) for w in l["weights"]]
                if l["type"] == "lstm":
                    m = torch.nn.LSTM(v.shape[-1], H, batch_first=True).double()  # keras gate order i, f, c, o == torch's i, f, g, o
                    m.weight_ih_l0.copy_(torch.from_numpy(W[0].T.copy()))
                    m.weight_hh_l0.copy_(torch.from_numpy(W[1].T.copy()))
                    m.bias_ih_l0.copy_(torch.from_numpy(W[2].ravel().copy()))
                    m.bias_hh_l0.zero_()
                    v, _ = m(v)
                elif l["type"] == "gru":
                    m = torch.nn.GRU(v.shape[-1], H, batch_first=True).double()
                    perm = np.concatenate([np.arange(H, 2 * H), np.arange(0, H), np.arange(2 * H, 3 * H)])  # keras z | r | c -> torch r | z | n
                    m.weight_ih_l0.copy_(torch.from_numpy(W[0].T[perm].copy()))
                    m.weight_hh_l0.copy_(torch.from_numpy(W[1].T[perm].copy()))
                    m.bias_ih_l0.copy_(torch.from_numpy(W[2][0][perm].copy()))
                    m.bias_hh_l0.copy_(torch.from_numpy(W[2][1][perm].copy()))
                    v, _ = m(v)
                elif l["type"] == "conv1d":
                    K, d = int(l["kernel_size"][-1]), int(l["dilation"][-1])
                    m = torch.nn.Conv1d(v.shape[-1], H, K, dilation=d).double()
                    m.weight.copy_(torch.from_numpy(W[0]).permute(2, 1, 0))  # keras [k][in][out] -> torch [out][in][k]
                    m.bias.copy_(torch.from_numpy(W[1].ravel().copy()))
                    v = acts[l.get("activation", "") or ""](m(torch.nn.functional.pad(v.transpose(1, 2), ((K - 1) * d, 0))).transpose(1, 2))
                else:
                    v = acts[l.get("activation", "") or ""](v @ torch.from_numpy(W[0]) + torch.from_numpy(W[1].ravel()))
            out[name] = v[0, 2048:, 0].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "keras_stacks_torch.npz"), **out)


MATRIX_MODELS = [("standard", "BossWN-standard.nam", 1.0), ("feather", "BossWN-feather.nam", 1.0), ("nano", "BossWN-nano.nam", 1.0),
                 ("a2q0", "BossWN-a2.nam", 0.0), ("a2q1", "BossWN-a2.nam", 1.0), ("lstm1x16", "BossLSTM-1x16.nam", 1.0),
                 ("lstm2x8", "BossLSTM-2x8.nam", 1.0)]
MATRIX_N = 4096
SWITCH_PLAN = [(0, 1.0), (8, 0.2), (20, 0.9), (27, 0.0)]  # (first 128-sample block, quality) of the mid-stream switch run, 32 blocks


def matrix_inputs():
    return {"sine": O.signal_sine(MATRIX_N), "noise": O.signal_noise(MATRIX_N, 20260928), "zeros": np.zeros(MATRIX_N, np.float32)}


def _np64(name, q, x):
    import ref_np
    j = O.load_json(name)
    if j["architecture"] == "SlimmableContainer":
        j = j["config"]["submodels"][O.quality_to_submodel(j, q)]["model"]
    if j["architecture"] == "WaveNet":
        return ref_np.wavenet_forward(O.wavenet_arrays_from_nam(j), j["weights"], x)[0]
    c = j["config"]
    return ref_np.lstm_forward_nam(int(c["num_layers"]), int(c["hidden_size"]), j["weights"], x)


def make_fixture_matrix():
    out = {}
    inputs = matrix_inputs()
    for k, x in inputs.items():
        out["input/" + k] = x
    for tag, name, q in MATRIX_MODELS:
        for k, x in inputs.items():
            out["oracle/%s/%s" % (tag, k)] = O.oracle_from_file(name, quality=q).process(x)
            out["np64/%s/%s" % (tag, k)] = np.asarray(_np64(name, q, x), np.float32)
            err = O.rms(out["oracle/%s/%s" % (tag, k)] - out["np64/%s/%s" % (tag, k)])
            assert err < (5e-6 if tag.startswith("lstm") else 1e-6), (tag, k, err)  # f32 recurrence vs float64: ~1e-6 on the LSTMs
    # mid-stream quality switch on the A2 container: per-block active submodel, expected output from one evaluation per submodel over
    # the concatenation of the blocks it was active for
    x = O.signal_noise(32 * 128, 77)
    j = O.load_json("BossWN-a2.nam")
    active = np.zeros(32, np.int32)
    for b0, q in SWITCH_PLAN:
        active[b0:] = O.quality_to_submodel(j, q)
    blocks = x.reshape(32, 128)
    want_o, want_n = np.zeros_like(blocks), np.zeros_like(blocks)
    for idx in (0, 1):
        sel = np.flatnonzero(active == idx)
        xs = blocks[sel].ravel()
        qq = 0.0 if idx == O.quality_to_submodel(j, 0.0) else 1.0
        want_o[sel] = O.oracle_from_file("BossWN-a2.nam", quality=qq).process(xs).reshape(-1, 128)
        want_n[sel] = np.asarray(_np64("BossWN-a2.nam", qq, xs), np.float32).reshape(-1, 128)
    out["switch/input"] = x
    out["switch/active"] = active
    out["switch/oracle"] = want_o.ravel()
    out["switch/np64"] = want_n.ravel()
    np.savez_compressed(os.path.join(HERE, "fixture_matrix.npz"), **out)


if __name__ == "__main__":
    make_matmul()
    make_fixture_matrix()
    make_oracle_outputs()
    make_keras_stacks()
    make_gru()
    print("golden fixtures written")
