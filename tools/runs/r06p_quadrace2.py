"""Hunt, part 2: an LSTM-only batch on the four-streams-per-wave kernel beside a SEPARATE batch of A1 Standard streams on another stream
(no multi-unit batch, no graph, no shared events): is plain co-residency enough?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
from neuralaudio_amd import capi
lib = capi.load_library()
ld = na.NeuralModelLoader()
P = lambda n: os.path.join(O.MODELS_DIR, n)
std = ld.CreateFromFile(P("BossWN-standard.nam"), doPrewarm=False)
lstm = ld.CreateFromFile(P("BossLSTM-1x16.nam"), doPrewarm=False)
dev = torch.device("cuda", 0)
nstd = int(sys.argv[1]) if len(sys.argv) > 1 else 22
OWN = len(sys.argv) > 2 and sys.argv[2] == "own"   # the batches on streams of their own (HIP's, not torch's)
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # Standard steps per LSTM step (more overlap)
nl, steps, n = 64, 400, 128
def lstm_batch(quad):
    lib.NA_DebugSetRecurrentQuadMin(1 if quad else 0)
    ts = torch.cuda.Stream(device=dev); b = (na.Batch(0) if OWN else na.Batch(0, hip_stream=ts.cuda_stream)); b.AddStreams(lstm, nl); return ts, b
tsq, bq = lstm_batch(True)
tsd, bd = lstm_batch(False)
tss = torch.cuda.Stream(device=dev); bs = (na.Batch(0) if OWN else na.Batch(0, hip_stream=tss.cuda_stream))
if nstd: bs.AddStreams(std, nstd)
g = torch.Generator(device="cpu").manual_seed(3)
xs = torch.clamp(0.3 * torch.randn(max(nstd, 1), n, generator=g), -1, 1).to(dev); ys = torch.zeros_like(xs)
bad = {}
for k in range(steps):
    x = torch.clamp(0.3 * torch.randn(nl, n, generator=g), -1, 1).to(dev)
    yq = torch.zeros(nl, n, device=dev); yd = torch.zeros(nl, n, device=dev)
    torch.cuda.synchronize()
    if nstd:
        with torch.cuda.stream(tss):
            for _ in range(REP): bs.ProcessDevice(xs.data_ptr(), ys.data_ptr(), n, n, n)
    lib.NA_DebugSetRecurrentQuadMin(1)
    with torch.cuda.stream(tsq):
        bq.ProcessDevice(x.data_ptr(), yq.data_ptr(), n, n, n)
    lib.NA_DebugSetRecurrentQuadMin(0)
    with torch.cuda.stream(tsd):
        bd.ProcessDevice(x.data_ptr(), yd.data_ptr(), n, n, n)
    if OWN:
        bs.Synchronize(); bq.Synchronize(); bd.Synchronize()
    torch.cuda.synchronize()
    d = (yq - yd).abs().amax(dim=1)
    for r in torch.nonzero(d > 1e-4).flatten().tolist():
        bad.setdefault(r, []).append(k)
print("separate batches:", "own streams" if OWN else "torch streams", "x%d" % REP, "std streams", nstd, "-> bad rows:", {r: v[:3] for r, v in sorted(bad.items())})
