"""Hunt: the four-streams-per-wave LSTM kernel inside a multi-unit batch is occasionally wrong in the 4th stream of a wave.
Compares a batch whose LSTM group runs on RecurrentQuadKernel with an identical batch on RecurrentDppKernel, step by step."""
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
nano = ld.CreateFromFile(P("BossWN-nano.nam"), doPrewarm=False)
lstm = ld.CreateFromFile(P("BossLSTM-1x16.nam"), doPrewarm=False)
dev = torch.device("cuda", 0)
mix = sys.argv[1] if len(sys.argv) > 1 else "std+nano"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 24
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 150
n = int(sys.argv[4]) if len(sys.argv) > 4 else 192
def build(quad):
    lib.NA_DebugSetRecurrentQuadMin(1 if quad else 0)
    ts = torch.cuda.Stream(device=dev)
    b = na.Batch(0, hip_stream=ts.cuda_stream)
    if "std" in mix: b.AddStreams(std, 22)
    if "nano" in mix: b.AddStreams(nano, 20)
    b.AddStreams(lstm, nl)
    return ts, b
bad = {}
tsq, bq = build(True)
tsd, bd = build(False)
S = bq.NumStreams(); first = S - nl
g = torch.Generator(device="cpu").manual_seed(3)
for k in range(steps):
    x = torch.clamp(0.3 * torch.randn(S, n, generator=g), -1, 1).to(dev)
    yq = torch.zeros(S, n, device=dev); yd = torch.zeros(S, n, device=dev)
    torch.cuda.synchronize()
    lib.NA_DebugSetRecurrentQuadMin(1)
    with torch.cuda.stream(tsq):
        bq.ProcessDevice(x.data_ptr(), yq.data_ptr(), n, n, n)
    lib.NA_DebugSetRecurrentQuadMin(0)
    with torch.cuda.stream(tsd):
        bd.ProcessDevice(x.data_ptr(), yd.data_ptr(), n, n, n)
    torch.cuda.synchronize()
    d = (yq[first:] - yd[first:]).abs().amax(dim=1)
    for r in torch.nonzero(d > 1e-4).flatten().tolist():
        e = (yq[first + r] - yd[first + r]).abs()
        f0 = int(torch.nonzero(e > 2e-5).flatten()[0])
        bad.setdefault(r, []).append((k, f0, "%.1e" % float(e[f0]), "%.1e" % float(e.max())))
print("mix", mix, "lstm streams", nl, "n", n, "steps", steps, "-> bad rows (row: first steps):", {r: v[:4] for r, v in sorted(bad.items())})
