"""Step times of the recurrent kernels of lstm_kernels.hip / gru_kernels.hip (runtime-shaped wave kernel, wave kernels, lane = stream kernels):
what switching the SLP vectoriser off for these files costs (csrc/Makefile REC_SLP)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
import na_oracle as O
ld = na.NeuralModelLoader()
G = os.path.join(ROOT, "tests", "golden", "models")
def lstm(l, h): return ld.CreateFromString(O.nam_json_lstm(l, h, O.synth_lstm_weights(l, h, seed=10 * h + l)), ".nam", doPrewarm=False)
cases = [("lstm1x18", lambda: lstm(1, 18), 1024), ("lstm3x16", lambda: lstm(3, 16), 1024), ("lstm2x40", lambda: lstm(2, 40), 1024), ("lstm2x64 (weights from L2)", lambda: lstm(2, 64), 256),
         ("lstm1x200", lambda: lstm(1, 200), 64), ("lstm1x512", lambda: lstm(1, 512), 16),
         ("keras gru12+dense", lambda: ld.CreateFromFile(os.path.join(G, "synthetic_stack_gru12_dense5relu_dense3sigmoid_dense1.json"), doPrewarm=False), 1024),
         ("keras lstm8+dense", lambda: ld.CreateFromFile(os.path.join(G, "synthetic_stack_lstm8_dense6tanh_dense1.json"), doPrewarm=False), 1024),
         ("keras gru12+conv+dense", lambda: ld.CreateFromFile(os.path.join(G, "synthetic_stack_gru12_conv16k4d64elu_dense5softmax_dense1.json"), doPrewarm=False), 256),
         ("tw40 keras lstm", lambda: ld.CreateFromFile(os.path.join(G, "tw40_blues_deluxe_deerinkstudios.json"), doPrewarm=False), 1024)]
dev = torch.device("cuda", 0)
for name, make, S in cases:
    m = make()
    b = na.Batch(0); b.AddStreams(m, S)
    x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty(S, 128, device=dev)
    torch.cuda.synchronize()
    for _ in range(20): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    b.Synchronize(); b.MarkTime(0)
    for _ in range(100): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    b.MarkTime(1); ms = b.ElapsedMs(); b.Synchronize()
    print("%-28s x %4d: %8.1f us per step   %s" % (name, S, ms * 10, b.StreamKernelName(0).split(" /")[0]), flush=True)
    b.close()
