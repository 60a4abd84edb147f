#!/usr/bin/env python3
"""When does the one-launch 1024 x Standard step go from 40 to 46 us?  (round 4: it did as soon as the batch had recorded timing events on
two more idle streams; tools/microbench/second_queue_penalty.hip shows no such effect for a plain bandwidth / ALU kernel.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import neuralaudio_amd as na
import na_oracle as O

dev = torch.device("cuda", 0)
m = na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
b = na.Batch(0, hip_stream=ts.cuda_stream)
b.AddStreams(m, 1024)
x = torch.clamp(0.25 * torch.randn(8, 1024, 128), -1, 1).to(dev); y = torch.empty(1024, 128, device=dev)
k = [0]

def measure(what, K=1500):
    for _ in range(300):
        b.ProcessDevice(x[k[0] % 8].data_ptr(), y.data_ptr(), 128); k[0] += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ts)
    for _ in range(K):
        b.ProcessDevice(x[k[0] % 8].data_ptr(), y.data_ptr(), 128); k[0] += 1
    e1.record(ts)
    torch.cuda.synchronize()
    print("%-64s %.2f us per step" % (what, e0.elapsed_time(e1) / K * 1e3), flush=True)

t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(256): b.ProcessDevice(x[0].data_ptr(), y.data_ptr(), 128)
    torch.cuda.synchronize()
measure("torch stream alone")
measure("torch stream alone (again)")
s2 = torch.cuda.Stream(device=dev)
measure("second torch stream created")
ev = torch.cuda.Event(enable_timing=False); ev.record(s2); ev.synchronize()
measure("no-timing event recorded on it")
ev = torch.cuda.Event(enable_timing=True); ev.record(s2); ev.synchronize()
measure("timing event recorded on it")
with torch.cuda.stream(s2):
    z = torch.zeros(16, device=dev); z.add_(1.0)
torch.cuda.synchronize()
measure("a kernel ran on it")
s3 = torch.cuda.Stream(device=dev)
ev3 = torch.cuda.Event(enable_timing=True); ev3.record(s3); ev3.synchronize()
measure("timing event on a third stream")
if len(sys.argv) > 1:
    b2 = na.Batch(0)
    b2.AddStreams(m, 2)
    b2.MarkTime(0); b2.MarkTime(1); b2.ElapsedMs()
    measure("another batch marked its three own streams")
