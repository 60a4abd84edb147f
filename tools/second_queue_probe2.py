#!/usr/bin/env python3
"""Second half of the 40 -> 46 us question: a batch on its OWN stream, one launch per step (NA_HOST_HALVES=0), wall clock around K steps,
with and without the library's timing marks (which create and touch the two half-batch streams)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import neuralaudio_amd as na
import na_oracle as O

dev = torch.device("cuda", 0)
m = na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
b = na.Batch(0)
b.AddStreams(m, 1024)
x = torch.clamp(0.25 * torch.randn(8, 1024, 128), -1, 1).to(dev); y = torch.empty(1024, 128, device=dev)
torch.cuda.synchronize()
k = [0]

def run(K):
    for _ in range(K):
        b.ProcessDevice(x[k[0] % 8].data_ptr(), y.data_ptr(), 128); k[0] += 1

def measure(what, marks, K=1500):
    run(300); b.Synchronize()
    t0 = time.perf_counter()
    if marks: b.MarkTime(0)
    run(K)
    if marks: b.MarkTime(1)
    b.Synchronize()
    wall = (time.perf_counter() - t0) / K * 1e6
    print("%-40s wall %.2f us per step%s   (half launches: %s)" % (what, wall, ("   marks %.2f" % (b.ElapsedMs() / K * 1e3)) if marks else "", b.UsesHalfLaunches()), flush=True)

t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    run(256); b.Synchronize()
measure("own stream, no marks", False)
measure("own stream, no marks (again)", False)
measure("with marks", True)
measure("with marks (again)", True)
measure("no marks after marks", False)
