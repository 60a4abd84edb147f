"""Test-suite plumbing: markers, the order of the GPU files, and a per-test watchdog.

Order (`-m gpu`): the evidence first -- committed goldens, direct parity / known-answer tests, ModelTest (config 1), the batch API,
then the kernel-specific files, the fuzz, and last the short forced-knob runs.  The long forced-knob matrix is a soak test behind
its own marker (`-m gpu_soak`) and is not part of `-m gpu`.

Watchdog: every test has a wall-clock limit (180 s unless `@pytest.mark.watchdog(seconds)` says otherwise).  A hang inside native
code never returns to the interpreter, so the limit is enforced by a daemon thread: it writes the test id and every thread's Python
stack to the real stderr (behind pytest's capture) and to gpurun_out/watchdog.log, then ends the process with exit code 3 -- a stall
names itself inside the driver's window instead of running into it."""
import faulthandler
import os
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

DEFAULT_LIMIT_S = float(os.environ.get("NA_TEST_WATCHDOG_S", "180"))

GPU_FILE_ORDER = ["test_gpu_fixtures.py", "test_gpu_parity.py", "test_gpu_modeltest.py", "test_gpu_batch.py", "test_gpu_spec.py",
                  "test_gpu_recurrent_quad.py", "test_gpu_resident.py", "test_gpu_stall.py", "test_gpu_multi.py", "test_gpu_scale.py",
                  "test_gpu_latency.py", "test_gpu_fuzz.py", "test_gpu_families.py"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_soak: the long forced-knob matrix on a real MI355X (-m gpu_soak); not part of -m gpu")
    config.addinivalue_line("markers", "watchdog(seconds): wall-clock limit of this test (default %g s)" % DEFAULT_LIMIT_S)
    _Watchdog.instance = _Watchdog(config)


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(GPU_FILE_ORDER)}

    def key(indexed):
        i, item = indexed
        name = os.path.basename(str(item.fspath))
        # CPU files keep their place in front; GPU files follow GPU_FILE_ORDER; unknown GPU files go before the fuzz
        return (rank.get(name, -1 if not name.startswith("test_gpu_") else len(GPU_FILE_ORDER) - 2.5), i)

    ordered = [item for _, item in sorted(enumerate(items), key=key)]
    # the soak matrix runs only when asked for by name (-m gpu_soak); `-m "not gpu"` must not pick it up
    if "gpu_soak" not in (config.getoption("markexpr") or ""):
        soak = [item for item in ordered if item.get_closest_marker("gpu_soak")]
        if soak:
            config.hook.pytest_deselected(items=soak)
            ordered = [item for item in ordered if not item.get_closest_marker("gpu_soak")]
    items[:] = ordered
    if not os.environ.get("NA_TEST_NO_WARM") and any(item.get_closest_marker("gpu") or item.get_closest_marker("gpu_soak") for item in ordered):
        _warm_imports()


def _warm_imports():
    """The first `import torch` on a fresh GPU box pages ~2 GB of libraries in from cold storage: seconds on a good day, minutes on a bad one
    (r06 soak run: 301 s, 238 s and 135 s for the first three sub-processes of a session whose parent had not imported torch; the r05
    driver run lost its 1200 s window to the same thing).  That cost belongs to no test: it is paid here, once, in front of the first
    test and outside the per-test watchdog -- and with the libraries in the page cache every forced-run sub-process starts warm."""
    t0 = time.monotonic()
    try:
        # the product library FIRST: the process then runs on the ROCm runtime the library was built against (/opt/rocm, 7.2) and torch
        # joins it; the other way round everything runs on torch's bundled HIP 7.0 runtime (NA_TEST_TORCH_FIRST=1 tests that order)
        if not os.environ.get("NA_TEST_TORCH_FIRST"):
            from neuralaudio_amd import capi
            capi.load_library()
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda").item()
    except Exception:
        pass  # (a box without torch / without a GPU: the tests say so themselves)
    dt = time.monotonic() - t0
    if dt > 10.0:
        try:
            os.write(2, ("\n[conftest] importing torch and opening the device took %.0f s (cold box)\n" % dt).encode())
        except OSError:
            pass


class _Watchdog:
    instance = None

    def __init__(self, config):
        self.config = config
        self.lock = threading.Lock()
        self.deadline = None
        self.nodeid = None
        self.limit = None
        self.thread = threading.Thread(target=self._run, name="na-test-watchdog", daemon=True)
        self.thread.start()

    def arm(self, nodeid, limit):
        with self.lock:
            self.nodeid, self.limit, self.deadline = nodeid, limit, time.monotonic() + limit

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _real_stderr_fd(self):
        try:    # the descriptor pytest's fd capture saved before redirecting 2
            capman = self.config.pluginmanager.getplugin("capturemanager")
            return capman._global_capturing.err.targetfd_save
        except Exception:
            return 2

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                expired = self.deadline is not None and time.monotonic() > self.deadline
                nodeid, limit = self.nodeid, self.limit
            if not expired:
                continue
            msg = "\n\nWATCHDOG: %s exceeded its %g s limit; stacks follow, process exits with code 3\n" % (nodeid, limit)
            sinks = [self._real_stderr_fd()]
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                sinks.append(os.open(os.path.join(ROOT, "gpurun_out", "watchdog.log"), os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644))
            except OSError:
                pass
            for fd in sinks:
                try:
                    os.write(fd, msg.encode())
                    faulthandler.dump_traceback(file=fd, all_threads=True)
                except Exception:
                    pass
            os._exit(3)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    mark = item.get_closest_marker("watchdog")
    limit = float(mark.args[0]) if mark and mark.args else DEFAULT_LIMIT_S
    _Watchdog.instance.arm(item.nodeid, limit)
    try:
        yield
    finally:
        _Watchdog.instance.disarm()


@pytest.fixture(scope="session")
def models_dir():
    return os.path.join(ROOT, "tests", "golden", "models")
