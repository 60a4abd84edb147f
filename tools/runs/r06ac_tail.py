"""Where do the rare long calls of tests/test_gpu_latency.py come from?  The same loop, 4 x 20 000 calls: NA_BatchProcess (1024 x 128, registered
blocks) and, interleaved, a null ctypes call (NA_GetDeviceCount): outliers of both, as an ordinary thread and with mlockall + SCHED_FIFO."""
import ctypes as C, gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import neuralaudio_amd as na
from neuralaudio_amd import capi
import na_oracle as O
lib = capi.load_library()
model = na.NeuralModelLoader().CreateFromFile(os.path.join(O.MODELS_DIR, "BossWN-standard.nam"), doPrewarm=False)
S, n = 1024, 128
b = na.Batch(0); b.AddStreams(model, S)
x = np.clip(0.3 * np.random.default_rng(3).standard_normal((S, n)), -1, 1).astype(np.float32); y = np.empty_like(x)
lib.NA_RegisterHostBuffer(x.ctypes.data_as(C.c_void_p), x.nbytes); lib.NA_RegisterHostBuffer(y.ctypes.data_as(C.c_void_p), y.nbytes)
xp, yp = x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float))
for _ in range(500): lib.NA_BatchProcess(b._h, xp, yp, n)
null = lib.NA_RcclAvailable
def run(tag, calls=20000):
    t = np.empty(calls); u = np.empty(calls)
    gc.collect(); gc.disable()
    for i in range(calls):
        t0 = time.perf_counter(); lib.NA_BatchProcess(b._h, xp, yp, n); t1 = time.perf_counter(); null(); t2 = time.perf_counter()
        t[i] = t1 - t0; u[i] = t2 - t1
    gc.enable()
    ms, us = np.sort(t) * 1e3, np.sort(u) * 1e3
    print("%-26s process: p50 %.3f p99.9 %.3f max %.3f ms, calls > 0.5 ms: %s | null call: p50 %.4f max %.3f ms, > 0.5 ms: %d" % (
        tag, ms[calls // 2], ms[int(calls * 0.999)], ms[-1], np.round(t[t > 0.5e-3] * 1e3, 2).tolist(), us[calls // 2], us[-1], int((u > 0.5e-3).sum())), flush=True)
for k in range(4): run("ordinary thread #%d" % k)
libc = C.CDLL("libc.so.6", use_errno=True)
print("mlockall:", libc.mlockall(3), "errno", C.get_errno())
try:
    os.sched_setscheduler(0, os.SCHED_FIFO, os.sched_param(50)); print("SCHED_FIFO 50")
except Exception as e:
    print("sched_setscheduler:", e)
for k in range(4): run("mlockall + SCHED_FIFO #%d" % k)
