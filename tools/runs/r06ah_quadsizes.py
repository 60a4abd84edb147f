"""LSTM / GRU 1x16: one stream per wave against four streams per wave over the batch size (argv[1]: the NA_REC_QUAD_MIN to run with, 0 = never)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralaudio_amd as na
from neuralaudio_amd import capi
import na_oracle as O
lib = capi.load_library()
ld = na.NeuralModelLoader()
lstm = ld.CreateFromFile(os.path.join(O.MODELS_DIR, "BossLSTM-1x16.nam"), doPrewarm=False)
gru = ld.CreateFromFile(os.path.join(ROOT, "tests", "golden", "models", "synthetic_gru_1x16.json"), doPrewarm=False)
dev = torch.device("cuda", 0)
for q in (0, 1):
    lib.NA_DebugSetRecurrentQuadMin(q)
    for name, m in (("lstm1x16", lstm), ("gru1x16", gru)):
        out = []
        for S in (1024, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144):
            b = na.Batch(0); b.AddStreams(m, S)
            x = torch.clamp(0.25 * torch.randn(S, 128), -1, 1).to(dev); y = torch.empty(S, 128, device=dev)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(2):
                for _ in range(100): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
                b.Synchronize(); b.MarkTime(0)
                for _ in range(300): b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
                b.MarkTime(1); ms = b.ElapsedMs(); b.Synchronize()
                best = min(best, ms / 0.3)
            out.append("%d:%.1f" % (S, best)); b.close()
        print("%-9s %-22s %s" % (name, "four streams per wave" if q else "one stream per wave", " ".join(out)), flush=True)
lib.NA_DebugSetRecurrentQuadMin(3072)
