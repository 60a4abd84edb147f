"""LSTM 2x16 + GRU 1x16 in one batch (config 4's mix) at several stream counts: us per 128-sample step (argv[1] = label; run under NA_REC_NOPIPE=1 / NA_REC_PIPE_MAX=100000 / nothing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
ld = na.NeuralModelLoader()
lstm = ld.CreateFromString(O.nam_json_lstm(2, 16, O.synth_lstm_weights(2, 16, seed=8)), ".nam", doPrewarm=False)
gru = ld.CreateFromFile(os.path.join(ROOT, "tests", "golden", "models", "synthetic_gru_1x16.json"), doPrewarm=False)
dev = torch.device("cuda", 0)
out = []
for L, G in ((256, 256), (512, 512), (256, 768), (768, 256), (512, 768), (512, 1024), (768, 768), (1024, 512), (256, 1536), (1024, 1024), (512, 1536)):
    b = na.Batch(0); b.AddStreams(lstm, L); b.AddStreams(gru, G)
    S = L + G
    x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty(S, 128, device=dev)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(2):
        for _ in range(100): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
        b.Synchronize(); b.MarkTime(0)
        for _ in range(300): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
        b.MarkTime(1); ms = b.ElapsedMs(); b.Synchronize()
        best = min(best, ms / 0.3)
    out.append("%d+%d:%.1f" % (L, G, best)); b.close()
print("%-10s" % (sys.argv[1] if len(sys.argv) > 1 else ""), " ".join(out))
