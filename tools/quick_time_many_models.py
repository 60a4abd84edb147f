#!/usr/bin/env python3
"""A batch of M DISTINCT model handles (the same file loaded M times: M model groups) x S streams each:
tools/quick_time_many_models.py <model> <M> <S>   -- us per 128-sample step (library timing marks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import neuralaudio_amd as na
path = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(ROOT, "tests/golden/models", sys.argv[1])
M, S = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
loader = na.NeuralModelLoader()
models = [loader.CreateFromFile(path, doPrewarm=False) for _ in range(M)]
b = na.Batch(0)
for m in models:
    b.AddStreams(m, S)
N = M * S
x = torch.clamp(0.25 * torch.randn(N, 128), -1, 1).to(dev); y = torch.empty_like(x)
torch.cuda.synchronize()
K = 200
for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
b.Synchronize()
res = []
for rep in range(3):
    b.MarkTime(0)
    for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
    b.MarkTime(1); b.WaitMarks(); res.append(b.ElapsedMs() / K * 1e3); b.Synchronize()
mode = "resident" if b.UsesResidentLaunch() else ("chains" if b.UsesHalfLaunches() else "ordered")
print("%s: %d models x %d streams = %d [%s]: %s us/step" % (os.path.basename(path), M, S, N, mode, " ".join("%.1f" % r for r in res)))
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
t1 = time.perf_counter()
b.Synchronize()
t2 = time.perf_counter()
print("   host: %.1f us per ProcessDevice call to enqueue, %.1f us per step with the final wait" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
