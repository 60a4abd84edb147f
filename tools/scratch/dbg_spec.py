import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import neuralaudio_amd as na
from neuralaudio_amd import capi
import na_oracle as O
lib = capi.load_library()
loader = na.NeuralModelLoader()
m = loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
rng = np.random.default_rng(1)
for streams in (1, 600):
    x = (0.3 * rng.standard_normal((streams, 256))).clip(-1, 1).astype(np.float32)
    res = {}
    for on in (1, 0):
        lib.NA_DebugSetWaveNetSpec(on)
        b = na.Batch(0)
        b.AddStreams(m, streams)
        print("streams", streams, "spec", on, b.StreamKernelName(0), "limit", b.StreamInputLimit(0))
        res[on] = np.concatenate([b.Process(x[:, :128]), b.Process(x[:, 128:])], axis=1)
        b.close()
    yo = O.oracle_from_file("BossWN-standard.nam").process(x[0])
    print(" spec  ", res[1][0][:6], res[1][0][128:132])
    print(" interp", res[0][0][:6], res[0][0][128:132])
    print(" oracle", yo[:6], yo[128:132])
    print(" maxdiff spec-interp", np.abs(res[1] - res[0]).max(), "interp-oracle rms", O.rms(res[0][0] - yo))
