"""NA_REC_QUAD_MIN=1: which path of the captured-table test is off?  LSTM rows of (a) an LSTM-only batch, (b) the mixed one-handle batch,
(c) the many-handles batch, against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import neuralaudio_amd as na
import na_oracle as O
import torch
ld = na.NeuralModelLoader()
P = lambda n: os.path.join(O.MODELS_DIR, n)
wn = [ld.CreateFromFile(P("BossWN-standard.nam"), doPrewarm=False) for _ in range(11)]
nano = [ld.CreateFromFile(P("BossWN-nano.nam"), doPrewarm=False) for _ in range(10)]
rec = [ld.CreateFromFile(P("BossLSTM-1x16.nam"), doPrewarm=False) for _ in range(12)]
per = 2
dev = torch.device("cuda", 0)
for lengths in ([128] * 4, [192] * 4, [64, 100, 128], [192] * 4 + [128] * 3 + [192] * 2 + [64, 100, 128]):
    total = max(lengths)
    S = per * (len(wn) + len(nano) + len(rec))
    g = torch.Generator(device="cpu").manual_seed(17)
    x = torch.clamp(0.3 * torch.randn(len(lengths), S, total, generator=g), -1.0, 1.0)
    nwn = per * (len(wn) + len(nano))
    def run(kind):
        ts = torch.cuda.Stream(device=dev)
        b = na.Batch(0, hip_stream=ts.cuda_stream)
        if kind == "many":
            for h in wn + nano + rec: b.AddStreams(h, per)
            rows = slice(nwn, S)
        elif kind == "one":
            for hs in (wn, nano, rec): b.AddStreams(hs[0], per * len(hs))
            rows = slice(nwn, S)
        else:
            b.AddStreams(rec[0], per * len(rec))
            rows = slice(0, per * len(rec))
        Sb = b.NumStreams()
        xin = torch.zeros(Sb, total, device=dev); yout = torch.zeros(Sb, total, device=dev)
        out = []
        with torch.cuda.stream(ts):
            for k, n in enumerate(lengths):
                xin.copy_(x[k, :Sb] if kind != "lstm" else x[k, nwn:S])
                b.ProcessDevice(xin.data_ptr(), yout.data_ptr(), n, total, total)
                b.Synchronize()
                out.append(yout[rows, :n].cpu().numpy().copy())
        b.close()
        return np.concatenate(out, axis=1)
    ys = {k: run(k) for k in ("lstm", "one", "many")}
    xs = np.concatenate([x[k, nwn, :n].numpy() for k, n in enumerate(lengths)])
    yo = O.oracle_from_file("BossLSTM-1x16.nam").process(xs)
    print(lengths[:5], {k: float(np.max(np.abs(v[0] - yo))) for k, v in ys.items()}, "one-many", float(np.max(np.abs(ys["one"] - ys["many"]))), "lstm-many", float(np.max(np.abs(ys["lstm"] - ys["many"]))))
