import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import neuralaudio_amd as na
S, n = 1024, 128
loader = na.NeuralModelLoader()
m = loader.CreateFromFile(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/models/BossWN-standard.nam"), doPrewarm=False)
b = na.Batch(0)
b.AddStreams(m, S)
x = np.clip(0.25 * np.random.default_rng(0).standard_normal((S, n)), -1, 1).astype(np.float32)
for _ in range(3000):
    y = b.Process(x)
t0 = time.perf_counter()
K = 3000
for _ in range(K):
    y = b.Process(x)
dt = (time.perf_counter() - t0) / K
print("NA_BatchProcess (host buffers, H2D + kernel + D2H + sync): %.1f us per 1024x128 buffer -> %.0f Msamples/s" % (dt * 1e6, S * n / dt / 1e6))
