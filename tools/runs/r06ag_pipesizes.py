"""LSTM 2x16 alone: one wave per stream against the two-wave layer pipeline over the batch size (run with NA_REC_NOPIPE=1 / NA_REC_PIPE_MAX=100000)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
ld = na.NeuralModelLoader()
m = ld.CreateFromString(O.nam_json_lstm(2, 16, O.synth_lstm_weights(2, 16, seed=8)), ".nam", doPrewarm=False)
dev = torch.device("cuda", 0)
out = []
for S in (384, 512, 640, 768, 896, 1024, 1152, 1280, 1408, 1536, 1792, 2048, 2304, 2560, 3072):
    b = na.Batch(0); b.AddStreams(m, S)
    x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty(S, 128, device=dev)
    torch.cuda.synchronize()
    for _ in range(200): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    b.Synchronize(); b.MarkTime(0)
    for _ in range(500): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    b.MarkTime(1); ms = b.ElapsedMs(); b.Synchronize()
    out.append("%d:%.1f" % (S, ms * 2)); b.close()
print(sys.argv[1] if len(sys.argv) > 1 else "", " ".join(out))
