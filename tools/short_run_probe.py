#!/usr/bin/env python3
"""Where a 20-step timed region (the driver's bench.py --steps 20) spends its wall clock beyond the kernels: host stamps around the marks,
the enqueue loop and the closing synchronisation, next to the HIP-event span.  1024 x A1 Standard x 128 on the batch's own streams."""
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
t = time.perf_counter()
while time.perf_counter() - t < 0.4:
    run(256); torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
for rep in range(8):
    run(5)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    b.MarkTime(0)
    t1 = time.perf_counter()
    run(20)
    t2 = time.perf_counter()
    b.MarkTime(1)
    t3 = time.perf_counter()
    if mode == "torch": torch.cuda.synchronize(dev)
    else: b.Synchronize()
    t4 = time.perf_counter()
    ev = b.ElapsedMs() * 1e3
    print("%s: marks(0) %.1f us | enqueue 20 steps %.1f | marks(1) %.1f | closing sync %.1f | total wall %.1f = %.2f per step | event span %.1f = %.2f per step"
          % (mode, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t4 - t0) * 1e6, (t4 - t0) * 1e6 / 20, ev, ev / 20), flush=True)
