"""LSTM 1x16 / GRU 1x16 at 4096 / 8192 streams: us per 128-sample step (the four-streams-per-wave kernel's regime)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
import ref_np as R
ld = na.NeuralModelLoader()
lstm = ld.CreateFromFile(os.path.join(O.MODELS_DIR, "BossLSTM-1x16.nam"), doPrewarm=False)
gru = ld.CreateFromFile(os.path.join(ROOT, "tests", "golden", "models", "synthetic_gru_1x16.json"), doPrewarm=False)
dev = torch.device("cuda", 0)
for name, m in (("lstm1x16", lstm), ("gru1x16", gru)):
    for S in (3072, 4096, 8192):
        b = na.Batch(0); b.AddStreams(m, S)
        x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty(S, 128, device=dev)
        torch.cuda.synchronize()
        for _ in range(200): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
        b.Synchronize(); b.MarkTime(0)
        for _ in range(500): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
        b.MarkTime(1); ms = b.ElapsedMs(); b.Synchronize()
        print(name, S, "streams: %.2f us per step" % (ms * 2), b.StreamKernelName(0))
        b.close()
