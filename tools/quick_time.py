#!/usr/bin/env python3
"""Time one synthetic A1-shaped WaveNet on the GPU: tools/quick_time.py <channels> <head_size> [streams] [lite=1]
(seeded random weights; prints us per 128-sample step).  Exploration helper."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O

ch, head = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
lite = (len(sys.argv) > 4 and sys.argv[4] == "1")
arrays = O.a1_arrays(ch, head, lite=lite)
w = O.synth_wavenet_weights(arrays, seed=7)
dev = torch.device("cuda", 0)
loader = na.NeuralModelLoader()
m = loader.CreateFromString(O.nam_json_wavenet_a1(ch, head, w, lite=lite), ".nam", doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
b = na.Batch(0, hip_stream=ts.cuda_stream)
b.AddStreams(m, S)
x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty_like(x)
for _ in range(300): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 1000
for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
torch.cuda.synchronize()
print("channels %d/%d lite=%d streams %d pack %d: %.2f us/step" % (ch, head, lite, S, b.StreamPackFactor(0), (time.perf_counter() - t0) / K * 1e6))
