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
model = loader.CreateFromFile(os.path.join(ROOT, "tests/golden/models", os.environ.get("NA_TRACE_MODEL", "BossWN-standard.nam")), doPrewarm=False)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
batch = na.Batch(0, hip_stream=ts.cuda_stream)
batch.AddStreams(model, S)
x = torch.clamp(0.25 * torch.randn(S, n), -1, 1).to(dev); y = torch.empty_like(x)
for _ in range(5):
    batch.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
torch.cuda.synchronize()
nst, waves = int(os.environ.get("NA_TRACE_STAGES", "23")) + 1, int(os.environ.get("NA_TRACE_WAVES", "4"))  # 23 stages of Standard + 1 slot for kernel entry / exit
trace = torch.zeros(nst * 4 * waves + nst * 8 * waves, dtype=torch.int64, device=dev)  # + 8 in-layer stamps per (stage, wave)
capi.load_library().NA_DebugSetTraceBuffer(trace.data_ptr())
batch.ProcessDevice(x.data_ptr(), y.data_ptr(), n)
torch.cuda.synchronize()
capi.load_library().NA_DebugSetTraceBuffer(None)
raw = trace.cpu().numpy().astype(np.float64)
t = raw[: nst * 4 * waves].reshape(nst, 4, waves)
sub = raw[nst * 4 * waves:].reshape(nst, waves, 8)
t0 = t[-1, 0].min()
print("kernel entry -> exit: %.0f cycles; entry -> stage 0: %.0f; last barrier -> exit: %.0f" % (t[-1, 1].max() - t0, t[0, 0].min() - t0, t[-1, 1].max() - t[-2, 3].max()))
print("stage  start(min..max)   conv   epi+pub  barrier-wait | per-wave mean cycles")
for s in range(nst - 1):
    st, cv, ep, br = t[s, 0], t[s, 1], t[s, 2], t[s, 3]
    conv = np.where(cv > 0, cv - st, 0)
    epi = np.where(cv > 0, ep - cv, ep - st)
    print("%2d  %7.0f..%7.0f  %6.0f  %6.0f  %6.0f   stage total %6.0f" % (s, st.min() - t0, st.max() - t0, conv.mean(), epi.mean(), (br - ep).mean(), br.max() - st.min()))

print("in-layer split (mean over waves, cycles): vec-read | shifted taps | last tap | activation | 1x1 | publish")
for s in range(nst - 1):
    u = sub[s]
    if u[:, 6].max() <= 0:
        continue
    d = np.diff(u[:, :7], axis=1).mean(axis=0)
    print("%2d  %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f   layer total %6.0f" % ((s,) + tuple(d) + ((u[:, 6] - u[:, 0]).mean(),)))
