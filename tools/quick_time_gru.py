#!/usr/bin/env python3
"""Time one synthetic keras GRU on the GPU: tools/quick_time_gru.py <layers> <hidden> [streams]  (prints us per 128-sample step)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O

L, H = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda", 0)
m = na.NeuralModelLoader().CreateFromString(json.dumps(O.synth_keras_gru(L, H, seed=3)), ".json", doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
b = na.Batch(0, hip_stream=ts.cuda_stream)
b.AddStreams(m, S)
x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty_like(x)
for _ in range(20): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 100
for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
torch.cuda.synchronize()
print("GRU %dx%d streams %d (%s): %.1f us/step" % (L, H, S, b.StreamKernelName(0), (time.perf_counter() - t0) / K * 1e6))
