#!/usr/bin/env python3
"""Per-stage timeline of workgroup 0 of the WaveNet kernel (shader-clock stamps, see NA_DebugSetTraceBuffer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import neuralaudio_amd as na
from neuralaudio_amd import capi

S, n = 1024, 128
dev = torch.device("cuda", 0)
loader = na.NeuralModelLoader()
model = loader.CreateFromFile(os.path.join(ROOT, "tests/golden/models/BossWN-standard.nam"), doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
batch = na.Batch(0, hip_stream=ts.cuda_stream)
batch.AddStreams(model, S)
x = torch.clamp(0.25 * torch.randn(S, n), -1, 1).to(dev); y = torch.empty_like(x)
for _ in range(5):
    batch.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
torch.cuda.synchronize()
nst, waves = 23, int(os.environ.get("NA_TRACE_WAVES", "16"))
trace = torch.zeros(nst * 4 * waves, dtype=torch.int64, device=dev)
capi.load_library().NA_DebugSetTraceBuffer(trace.data_ptr())
batch.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
torch.cuda.synchronize()
capi.load_library().NA_DebugSetTraceBuffer(None)
t = trace.cpu().numpy().reshape(nst, 4, waves).astype(np.float64)
t0 = t[0, 0].min()
print("stage  start(min..max)   conv   epi+pub  barrier-wait | per-wave mean cycles; total %.0f cycles" % (t[-1, 3].max() - t0))
for s in range(nst):
    st, cv, ep, br = t[s, 0], t[s, 1], t[s, 2], t[s, 3]
    conv = np.where(cv > 0, cv - st, 0)
    epi = np.where(cv > 0, ep - cv, ep - st)
    print("%2d  %7.0f..%7.0f  %6.0f  %6.0f  %6.0f   (wave spread at barrier exit %.0f)" % (s, st.min() - t0, st.max() - t0, conv.mean(), epi.mean(), (br - ep).mean(), br.max() - br.min()))
