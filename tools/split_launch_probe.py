"""Half-batches of the headline workload on separate HIP streams: free-running (no ordering between the halves: what round 3 measured,
39.4 vs 42.5 us) and JOINED per step -- fork from / join back into one stream around every step, which is what a single
NA_BatchProcessDevice call would have to do (its caller orders work on ONE stream).  1024 x A1 Standard x 128 frames."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import neuralaudio_amd as na
dev = torch.device("cuda", 0)
loader = na.NeuralModelLoader()
m = loader.CreateFromFile(os.path.join(ROOT, "tests/golden/models/BossWN-standard.nam"), doPrewarm=False)


def run(nb, S_each, steps=3000, joined=False, skew_us=0.0):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nb)]
    batches = []
    for st in streams:
        b = na.Batch(0, hip_stream=st.cuda_stream)
        b.AddStreams(m, S_each)
        batches.append(b)
    xs = [torch.clamp(0.25 * torch.randn(S_each, 128), -1, 1).to(dev) for _ in range(nb)]
    ys = [torch.empty_like(x) for x in xs]
    fork = [torch.cuda.Event() for _ in range(4)]
    joins = [[torch.cuda.Event() for _ in range(nb)] for _ in range(4)]
    k = [0]

    def step():
        if joined and nb > 1:
            e = fork[k[0] % 4]
            e.record(streams[0])
            for i in range(1, nb):
                streams[i].wait_event(e)
        for b, x, y in zip(batches, xs, ys):
            b.ProcessDevice(x.data_ptr(), y.data_ptr(), 128, 128, 128)
        if joined and nb > 1:
            for i in range(1, nb):
                j = joins[k[0] % 4][i]
                j.record(streams[i])
                streams[0].wait_event(j)
        k[0] += 1
    for _ in range(6000): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%d batch(es) x %d streams%s: %.2f us per 1024-stream step" % (nb, S_each, " joined per step" if joined else " free-running", dt * 1e6), flush=True)
    for b in batches:
        b.close()


run(1, 1024)
run(2, 512)
run(2, 512, joined=True)
run(4, 256, joined=True)
run(1, 1024)
