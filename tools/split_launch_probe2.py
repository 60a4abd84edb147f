"""Two free-running half-batches (512 Standard streams each, own HIP stream, own batch): does it matter whether the two sequences of
launches run IN phase or in ANTI-phase on the CUs (one workgroup of each per CU)?  Stream B is started `offset` launches of A late
(A runs `offset` extra steps first while B's stream is idle), then both free-run."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import neuralaudio_amd as na
dev = torch.device("cuda", 0)
loader = na.NeuralModelLoader()
m = loader.CreateFromFile(os.path.join(ROOT, "tests/golden/models/BossWN-standard.nam"), doPrewarm=False)


def run(skew_sleep_us, steps=4000):
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    batches = []
    for st in streams:
        b = na.Batch(0, hip_stream=st.cuda_stream)
        b.AddStreams(m, 512)
        batches.append(b)
    xs = [torch.clamp(0.25 * torch.randn(512, 128), -1, 1).to(dev) for _ in range(2)]
    ys = [torch.empty_like(x) for x in xs]
    for _ in range(3000):
        for b, x, y in zip(batches, xs, ys):
            b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    torch.cuda.synchronize()
    # phase offset: stream B sleeps on the device for ~skew_sleep_us before its first timed launch
    if skew_sleep_us > 0:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(skew_sleep_us * 1e-6 * 2.0e9))
    t0 = time.perf_counter()
    for _ in range(steps):
        for b, x, y in zip(batches, xs, ys):
            b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("two free-running half-batches, B started %5.1f us late: %.2f us per 1024-stream step" % (skew_sleep_us, dt * 1e6), flush=True)
    for b in batches:
        b.close()


for off in (0.0, 5.0, 10.0, 15.0, 20.0, 0.0):
    run(off)
