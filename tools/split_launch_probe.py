import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import neuralaudio_amd as na
dev = torch.device("cuda", 0)
loader = na.NeuralModelLoader()
m = loader.CreateFromFile(os.path.join(ROOT, "tests/golden/models/BossWN-standard.nam"), doPrewarm=False)
def run(nb, S_each, steps=3000):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nb)]
    batches = []
    for st in streams:
        b = na.Batch(0, hip_stream=st.cuda_stream)
        b.AddStreams(m, S_each)
        batches.append(b)
    xs = [torch.clamp(0.25 * torch.randn(S_each, 128), -1, 1).to(dev) for _ in range(nb)]
    ys = [torch.empty_like(x) for x in xs]
    def step():
        for b, x, y in zip(batches, xs, ys):
            b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
    for _ in range(8000): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%d batch(es) x %d streams: %.2f us per 1024-stream step, %.1f Msamples/s" % (nb, S_each, dt * 1e6, nb * S_each * 128 / dt / 1e6))
run(1, 1024)
run(2, 512)
run(4, 256)
run(1, 1024)
