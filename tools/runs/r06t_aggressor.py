"""Keeps the GPU busy with the f16-split WaveNet kernel in its one-stream-per-workgroup flavour (a batch of `S` A1 Standard streams stepped
back to back) for `seconds` -- the aggressor side of the co-residency experiments (profiles/r06_quad_race.txt)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import neuralaudio_amd as na
import na_oracle as O
import torch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
m = na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev)
b = na.Batch(0, hip_stream=ts.cuda_stream); b.AddStreams(m, S)
x = torch.clamp(0.3 * torch.randn(S, 128), -1, 1).to(dev); y = torch.zeros_like(x)
print("aggressor kernel:", b.StreamKernelName(0), flush=True)
t0 = time.time(); steps = 0
while time.time() - t0 < seconds:
    with torch.cuda.stream(ts):
        for _ in range(200): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    torch.cuda.synchronize(); steps += 200
print("aggressor done:", steps, "steps", flush=True)
