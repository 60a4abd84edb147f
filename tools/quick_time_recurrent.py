#!/usr/bin/env python3
"""us per 128-sample step of a recurrent model at several batch sizes, on both lane layouts of the LDS-free kernel:
tools/quick_time_recurrent.py [model file under tests/golden/models | lstm:<layers>:<hidden> | gru:<layers>:<hidden>] [stream counts ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
from neuralaudio_amd import capi

name = sys.argv[1] if len(sys.argv) > 1 else "BossLSTM-1x16.nam"
counts = [int(v) for v in sys.argv[2:]] or [1024, 2048, 4096, 8192, 16384]
lib = capi.load_library()
dev = torch.device("cuda", 0)
if name.startswith("gru:"):
    import json
    import na_oracle as O
    _, layers, hidden = name.split(":")
    m = na.NeuralModelLoader().CreateFromString(json.dumps(O.synth_keras_gru(int(layers), int(hidden), seed=3)), ".json", doPrewarm=False)
elif name.startswith("lstm:"):
    import na_oracle as O
    _, layers, hidden = name.split(":")
    m = na.NeuralModelLoader().CreateFromString(O.nam_json_lstm(int(layers), int(hidden), O.synth_lstm_weights(int(layers), int(hidden), seed=3)), ".nam", doPrewarm=False)
else:
    m = na.NeuralModelLoader().CreateFromFile(os.path.join(ROOT, "tests/golden/models", name), doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
for S in counts:
    x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty_like(x)
    line = []
    for quad_min in (0, 1):
        lib.NA_DebugSetRecurrentQuadMin(quad_min)
        b = na.Batch(0, hip_stream=ts.cuda_stream)
        b.AddStreams(m, S)
        for _ in range(20): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
        torch.cuda.synchronize()
        K = 200
        t0 = time.perf_counter()
        for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / K * 1e6
        line.append("%s %.1f us = %.2f Gsamples/s" % ("one stream per wave" if quad_min == 0 else "four per wave", us, S * 128 / us / 1e3))
        del b
    print("%s x %d: %s" % (name, S, " | ".join(line)))
