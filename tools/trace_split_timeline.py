#!/usr/bin/env python3
"""Per-layer timeline of one workgroup of the f16-split WaveNet kernel (trace build: make SUFFIX=_trace EXTRA=-DNA_SP_TRACE,
run with NA_LIB_SUFFIX=_trace).  Stamps per (stage, wave): 0 layer start, 1 conv issued, 2 activation done, 3 1x1 / publish issued,
4 weight DMA landed, 5 barrier passed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import neuralaudio_amd as na
from neuralaudio_amd import capi

S, n = int(os.environ.get("NA_TRACE_STREAMS", "1024")), 128
dev = torch.device("cuda", 0)
loader = na.NeuralModelLoader()
model = loader.CreateFromFile(os.path.join(ROOT, "tests/golden/models", os.environ.get("NA_TRACE_MODEL", "BossWN-standard.nam")), doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
# NA_TRACE_OWN=1: the batch on its own streams -- a step is two free-running half-batch launches, chain NA_TRACE_CHAIN (0) is stamped
batch = na.Batch(0) if os.environ.get("NA_TRACE_OWN") else na.Batch(0, hip_stream=ts.cuda_stream)
batch.AddStreams(model, S)
x = torch.clamp(0.25 * torch.randn(S, n), -1, 1).to(dev); y = torch.empty_like(x)
for _ in range(200):
    batch.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
torch.cuda.synchronize()
nst, waves = int(os.environ.get("NA_TRACE_STAGES", "23")), int(os.environ.get("NA_TRACE_WAVES", "8"))  # waves per workgroup of the traced instantiation (SPB x WPS)
trace = torch.zeros((nst + 1) * 8 * waves, dtype=torch.int64, device=dev)
capi.load_library().NA_DebugSetTraceBuffer(trace.data_ptr())
for _ in range(int(os.environ.get("NA_TRACE_STEPS", "3"))):
    batch.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
torch.cuda.synchronize()
capi.load_library().NA_DebugSetTraceBuffer(None)
t = trace.cpu().numpy().astype(np.float64).reshape(nst + 1, 8, waves)
t0, t1 = t[nst, 0].min(), t[nst, 1].max()
print("kernel entry -> exit of the traced workgroup: %.0f cycles" % (t1 - t0))
if t[nst, 3, 0] > 0:  # resident launch (stamps of the LAST block the traced workgroup ran; 2 = the closing barrier of the block before it)
    print("resident launch: command taken -> block entry %.0f cycles; block exit -> closing barrier %.0f; closing barrier of the previous block -> command taken %.0f"
          % (t0 - t[nst, 3, 0], t[nst, 2, 0] - t1, t[nst, 3, 0] - t[nst, 4, 0] if t[nst, 4, 0] > 0 else float("nan")))
    print("block period of the traced workgroup (closing barrier to closing barrier): %.0f cycles" % (t[nst, 2, 0] - t[nst, 4, 0]))
print("stage  start    conv   activ  1x1+pub  dma-wait  barrier | mean over waves (cycles), layer total = max over waves")
for s in range(nst):
    u = t[s]
    if u[5].max() <= 0:
        continue
    d = np.diff(u[:6], axis=0).mean(axis=1)
    print("%2d  %7.0f  %6.0f %6.0f %6.0f %6.0f %6.0f   total %6.0f" % ((s, u[0].min() - t0) + tuple(d) + (u[5].max() - u[0].min(),)))
for s in [int(v) for v in os.environ.get("NA_TRACE_DETAIL", "").split(",") if v]:
    u = t[s]
    print("stage %d, per wave, cycles since the first wave's layer start: start [dma-issued taps-done] conv activ 1x1+pub dma-wait barrier" % s)
    for w in range(waves):
        print("  wave %d: " % w + " ".join("%6.0f" % (u[k, w] - u[0].min()) for k in (0, 6, 7, 1, 2, 3, 4, 5)))
