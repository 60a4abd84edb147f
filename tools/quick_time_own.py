#!/usr/bin/env python3
"""Time a batch on its OWN streams (free-running chains / resident launch): tools/quick_time_own.py <model> <streams> [ENV=VAL ...]
prints us per 128-sample step and per 1024 streams, from the library's timing marks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
import torch
import neuralaudio_amd as na

path = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(ROOT, "tests/golden/models", sys.argv[1])
S = int(sys.argv[2])
dev = torch.device("cuda", 0)
m = na.NeuralModelLoader().CreateFromFile(path, doPrewarm=False)
b = na.Batch(0)
b.AddStreams(m, S)
x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty_like(x)
torch.cuda.synchronize()
K = int(os.environ.get("K", "300"))
for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
b.Synchronize()
res = []
for rep in range(3):
    b.MarkTime(0)
    for _ in range(K): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128)
    b.MarkTime(1)
    b.WaitMarks()
    res.append(b.ElapsedMs() / K * 1e3)
    b.Synchronize()
mode = "resident" if b.UsesResidentLaunch() else ("chains" if b.UsesHalfLaunches() else "ordered")
print("%s streams %d %s [%s]: %s us/step = %.2f us per 1024 streams" % (os.path.basename(path), S, " ".join(sys.argv[3:]), mode, " ".join("%.2f" % r for r in res), min(res) * 1024 / S))
