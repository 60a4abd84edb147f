import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import neuralaudio_amd as na
import na_oracle as O
import ref_np
loader = na.NeuralModelLoader()
def models(which):
    if which == "standard":
        return loader.CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False), O.oracle_from_file("BossWN-standard.nam")
    arrays = O.a1_arrays(12, 6); w = O.synth_wavenet_weights(arrays, seed=33)
    return loader.CreateFromString(O.nam_json_wavenet_a1(12, 6, w), ".nam", doPrewarm=False), O.OracleWaveNet(arrays, w)
print("kernel env", os.environ.get("NA_WN_KERNEL"))
for which in ("standard", "lite"):
    m, ora = models(which)
    n = 2048
    for amp in (0.3, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
        x = (amp * np.sin(0.013 * np.arange(n)) + 0.3 * amp * np.sin(0.31 * np.arange(n))).astype(np.float32)
        truth, _ = ref_np.wavenet_forward(ora.arrays, ora.weights, x)
        quiet, _ = ref_np.wavenet_forward(ora.arrays, ora.weights, np.zeros(n, dtype=np.float32))
        sig = O.rms(truth - quiet)
        b = na.Batch(0); b.AddStreams(m, 1)
        y = np.concatenate([b.Process(x[None, i:i + 128]) for i in range(0, n, 128)], axis=1)[0]
        kn = b.StreamKernelName(0); b.close()
        yo = ora.process(x)
        def parts(v):
            e = v.astype(np.float64) - truth
            return abs(e.mean()), O.rms(e - e.mean())
        gd, ga = parts(y); od, oa = parts(yo)
        print("%-8s amp %-7g signal %.3e | %s dc %.2e ac %.2e (ac/signal %.2e) | oracle dc %.2e ac %.2e (ac/signal %.2e)" % (which, amp, sig, kn, gd, ga, ga / sig, od, oa, oa / sig))
