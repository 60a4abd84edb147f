import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import neuralaudio_amd as na
import na_oracle as O
loader = na.NeuralModelLoader()
for name in ("BossWN-standard.nam", "BossWN-nano.nam", "BossWN-a2.nam"):
    for amp in (1.0, 30.0, 1000.0, 30000.0):
        m = loader.CreateFromFile(os.path.join('/root/repo/tests/golden/models', name))
        x = (amp * np.sin(0.01 * np.arange(1024))).astype(np.float32)
        y = m.Process(x)
        yo = O.oracle_from_file(name).process(x)
        print(name, amp, 'rms err', float(np.sqrt(np.mean((y - yo) ** 2))), 'out rms', float(np.sqrt(np.mean(yo ** 2))), 'finite', bool(np.all(np.isfinite(y))))
