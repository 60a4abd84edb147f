import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import neuralaudio_amd as na
from neuralaudio_amd import capi
import na_oracle as O
lib = capi.load_library()
loader = na.NeuralModelLoader()
name = sys.argv[1] if len(sys.argv) > 1 else "BossWN-feather.nam"
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 1030
m = loader.CreateFromFile(os.path.join(O.MODELS_DIR, name), doPrewarm=False)
rng = np.random.default_rng(1)
nb = 6
base = (0.3 * rng.standard_normal((9, 128 * nb))).clip(-1, 1).astype(np.float32)
x = base[np.arange(streams) % 9]
res = {}
for on in (1, 0):
    lib.NA_DebugSetWaveNetSpec(on)
    b = na.Batch(0)
    b.AddStreams(m, streams)
    print("spec", on, b.StreamKernelName(0), b.StreamPackFactor(0))
    res[on] = np.concatenate([b.Process(np.ascontiguousarray(x[:, i * 128:(i + 1) * 128])) for i in range(nb)], axis=1)
    b.close()
d = np.abs(res[1] - res[0])
print("max diff", d.max())
bad_streams = np.where(d.max(axis=1) > 0)[0]
print("bad streams", len(bad_streams), bad_streams[:20], bad_streams[-5:])
for blk in range(nb):
    seg = d[:, blk * 128:(blk + 1) * 128]
    fr = np.where(seg.max(axis=0) > 0)[0]
    print("block", blk, "max", seg.max(), "bad frames", (fr.min(), fr.max(), len(fr)) if len(fr) else None)
yo = O.oracle_from_file(name).process(x[0])
print("spec vs oracle rms", O.rms(res[1][0] - yo), "interp vs oracle", O.rms(res[0][0] - yo))
