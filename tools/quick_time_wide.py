#!/usr/bin/env python3
"""Time a Standard-SHAPED WaveNet with wider layer arrays on the runtime-shaped kernel: tools/quick_time_wide.py <channels> <head> [streams]
(us per 128-sample step; weights seeded; parity of a few streams against the oracle is printed too)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O

C, H = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
arrays = O.a1_arrays(C, H, lite=False)
w = O.synth_wavenet_weights(arrays, seed=5)
dev = torch.device("cuda", 0)
m = na.NeuralModelLoader().CreateFromString(O.nam_json_wavenet_generic(arrays, w), ".nam", doPrewarm=False)
assert m is not None
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
b = na.Batch(0, hip_stream=ts.cuda_stream)
b.AddStreams(m, S)
xh = np.stack([O.signal_noise(256, 50 + s % 7) for s in range(S)])
x = torch.from_numpy(xh).to(dev); y = torch.empty_like(x)
b.ProcessDevice(x[:, :128].contiguous().data_ptr(), y.data_ptr(), 128, 128, 256)
yy = torch.empty(S, 128, device=dev)
x1 = x[:, 128:].contiguous()
b.ProcessDevice(x1.data_ptr(), yy.data_ptr(), 128, 128, 128)
torch.cuda.synchronize()
got = np.concatenate([y[:, :128].cpu().numpy(), yy.cpu().numpy()], axis=1)
err = max(O.rms(got[s] - O.OracleWaveNet(arrays, w).process(xh[s])) for s in (0, S - 1))
xb = x[:, :128].contiguous()
for _ in range(10): b.ProcessDevice(xb.data_ptr(), yy.data_ptr(), 128)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K): b.ProcessDevice(xb.data_ptr(), yy.data_ptr(), 128)
torch.cuda.synchronize()
print("A1-shaped %d/%d channels, %d streams (%s): %.1f us/step, rms error vs oracle %.2e" % (C, H, S, b.StreamKernelName(0), (time.perf_counter() - t0) / K * 1e6, err))
