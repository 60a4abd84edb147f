#!/usr/bin/env python3
"""Time one model file on the GPU: tools/quick_time_model.py <file under tests/golden/models or path> [streams] [ENV=VAL ...]  (us per 128-sample step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
import torch
import neuralaudio_amd as na

path = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(ROOT, "tests/golden/models", sys.argv[1])
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
m = na.NeuralModelLoader().CreateFromFile(path, doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
b = na.Batch(0, hip_stream=ts.cuda_stream)
b.AddStreams(m, S)
x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty_like(x)
for _ in range(10): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
torch.cuda.synchronize()
print("%s streams %d %s (%s): %.1f us/step" % (os.path.basename(path), S, " ".join(sys.argv[3:]), b.StreamKernelName(0), (time.perf_counter() - t0) / K * 1e6))
