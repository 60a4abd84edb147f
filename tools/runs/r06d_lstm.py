"""LSTM 2x16 alone at several batch sizes: us per 128-sample step (HIP-event marks of the batch)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
ld = na.NeuralModelLoader()
w = O.synth_lstm_weights(2, 16, seed=8)
m = ld.CreateFromString(O.nam_json_lstm(2, 16, w), ".nam", doPrewarm=False)
dev = torch.device("cuda", 0)
for S in (256, 512, 1024, 1536, 2048, 2560):
    b = na.Batch(0)
    b.AddStreams(m, S)
    x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev)
    y = torch.empty(S, 128, device=dev)
    torch.cuda.synchronize()
    for _ in range(300):
        b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    b.Synchronize()
    b.MarkTime(0)
    for _ in range(1000):
        b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    b.MarkTime(1)
    ms = b.ElapsedMs()
    b.Synchronize()
    print(S, "streams: %.2f us per step" % (ms))
    b.close()
