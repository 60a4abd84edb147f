"""Every kernel family as a possible victim: a batch stepped ALONE and then again (fresh, same inputs) BESIDE a batch of A1 Standard streams on
the split kernel's one-stream-per-workgroup flavour (two Standard steps per victim step, own streams: they overlap).  Kernels are
deterministic, so the two runs must agree bit for bit."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
import ref_np as R
from neuralaudio_amd import capi
lib = capi.load_library()
ld = na.NeuralModelLoader()
P = lambda n: os.path.join(O.MODELS_DIR, n)
dev = torch.device("cuda", 0)
std = ld.CreateFromFile(P("BossWN-standard.nam"), doPrewarm=False)
def lstm(l, h): return ld.CreateFromString(O.nam_json_lstm(l, h, O.synth_lstm_weights(l, h, seed=10 * h + l)), ".nam", doPrewarm=False)
victims = {
    "lstm1x16 x64 (dpp)": (lambda: ld.CreateFromFile(P("BossLSTM-1x16.nam"), doPrewarm=False), 64, 0),
    "lstm1x16 x64 (quad)": (lambda: ld.CreateFromFile(P("BossLSTM-1x16.nam"), doPrewarm=False), 64, 1),
    "lstm2x8 x64 (skew)": (lambda: ld.CreateFromFile(P("BossLSTM-2x8.nam"), doPrewarm=False), 64, 0),
    "lstm2x16 x64 (pipe)": (lambda: lstm(2, 16), 64, 0),
    "lstm1x24 x64 (32-unit layout)": (lambda: lstm(1, 24), 64, 0),
    "lstm2x40 x32 (wave rt)": (lambda: lstm(2, 40), 32, 0),
    "lstm1x18 x512 (rt)": (lambda: lstm(1, 18), 512, 0),
    "lstm3x16 x256 (rt)": (lambda: lstm(3, 16), 256, 0),
    "lstm2x40 x256 (rt)": (lambda: lstm(2, 40), 256, 0),
    "lstm1x200 x64 (rt, waves share a stream)": (lambda: lstm(1, 200), 64, 0),
    "keras stack gru12+dense x512 (rt)": (lambda: ld.CreateFromFile(os.path.join(ROOT, "tests", "golden", "models", "synthetic_stack_gru12_dense5relu_dense3sigmoid_dense1.json"), doPrewarm=False), 512, 0),
    "gru1x16 x64": (lambda: ld.CreateFromFile(os.path.join(ROOT, "tests", "golden", "models", "synthetic_gru_1x16.json"), doPrewarm=False), 64, 0),
    "gru1x16 x64 (quad)": (lambda: ld.CreateFromFile(os.path.join(ROOT, "tests", "golden", "models", "synthetic_gru_1x16.json"), doPrewarm=False), 64, 1),
    "keras stack gru12+dense": (lambda: ld.CreateFromFile(os.path.join(ROOT, "tests", "golden", "models", "synthetic_stack_gru12_dense5relu_dense3sigmoid_dense1.json"), doPrewarm=False), 64, 0),
    "nano x20 (frame kernel)": (lambda: ld.CreateFromFile(P("BossWN-nano.nam"), doPrewarm=False), 20, 0),
    "feather x20": (lambda: ld.CreateFromFile(P("BossWN-feather.nam"), doPrewarm=False), 20, 0),
    "a2 x20": (lambda: ld.CreateFromFile(P("BossWN-a2.nam"), doPrewarm=False), 20, 0),
    "standard x20 (itself)": (lambda: ld.CreateFromFile(P("BossWN-standard.nam"), doPrewarm=False), 20, 0),
}
only = sys.argv[1] if len(sys.argv) > 1 else ""
steps, n = 300, 128
agg = na.Batch(0); agg.AddStreams(std, 64)
xa = torch.clamp(0.3 * torch.randn(64, n), -1, 1).to(dev); ya = torch.zeros_like(xa)
for name, (make, S, quad) in victims.items():
    if only and only not in name: continue
    m = make()
    g = torch.Generator(device="cpu").manual_seed(5)
    xs = torch.clamp(0.3 * torch.randn(steps, S, n, generator=g), -1, 1).to(dev)
    outs = []
    for beside in (False, True):
        lib.NA_DebugSetRecurrentQuadMin(1 if quad else 0)
        b = na.Batch(0); b.AddStreams(m, S)
        y = torch.zeros(steps, S, n, device=dev)
        torch.cuda.synchronize()
        for k in range(steps):
            if beside:
                agg.ProcessDevice(xa.data_ptr(), ya.data_ptr(), n, n, n); agg.ProcessDevice(xa.data_ptr(), ya.data_ptr(), n, n, n)
            b.ProcessDevice(xs[k].data_ptr(), y[k].data_ptr(), n, n, n)
            b.Synchronize()
            if beside: agg.Synchronize()
        kn = b.StreamKernelName(0)
        b.close(); outs.append(y)
    d = (outs[0] - outs[1]).abs()
    badsteps = torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten().tolist()
    rows = sorted(set(torch.nonzero(d.amax(dim=(0, 2)) > 0).flatten().tolist()))
    print("%-34s %-24s differing steps %d of %d, rows %s, max |diff| %.2e" % (name, kn.split(" /")[0], len(badsteps), steps, rows[:12], float(d.max())), flush=True)
lib.NA_DebugSetRecurrentQuadMin(3072)
