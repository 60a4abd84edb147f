"""neuralaudio_amd -- MI355X-native NeuralAudio hot path (NAM WaveNet / LSTM per-sample inference).

Thin, pythonic mirror of the reference's host interface for this path:
  NeuralModelLoader / NeuralModel  <->  NeuralAudio/NeuralModel.h:33-231 (same method names and meaning)
  Batch                            <->  new: many independent streams per GPU (include/neuralaudio_amd.h)
All compute happens in libNeuralAudioCAPI.so (C++ host code + hand-written gfx950 HIP kernels) through
its C ABI; numpy arrays are only the host-side containers.
"""
import ctypes as C
import os

import numpy as np

from . import capi

__all__ = ["NeuralModelLoader", "NeuralModel", "Batch", "MultiBatch", "EModelLoadMode", "EMathMode", "ECompositeModelLoadMode", "device_count",
           "NeuralAudioError"]


class NeuralAudioError(RuntimeError):
    pass


class EModelLoadMode:
    Internal = 0
    RTNeural = 1
    NAMCore = 2


class EMathMode:
    FastMath = 0
    StdMath = 1


class ECompositeModelLoadMode:
    LoadAll = 0
    OnDemand = 1


def device_count():
    return capi.device_count()


def rccl_available():
    """librccl.so loads and exports what the multi-GPU host binds (no GPU needed)."""
    return bool(capi.load_library().NA_RcclAvailable())


def debug_set_rccl_api(mode, fail_send_at=0, rendezvous_ms=0):
    """Tests: 1 = the multi-GPU host binds the in-library loopback table (ranks may share a device) instead of librccl.so; 0 = back."""
    capi.load_library().NA_DebugSetRcclApi(int(mode), int(fail_send_at), int(rendezvous_ms))


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class NeuralModel:
    """One mono audio stream (NeuralAudio::NeuralModel). Created by NeuralModelLoader."""

    def __init__(self, handle):
        self._lib = capi.load_library()
        self._h = handle

    # -- reference API -------------------------------------------------------------------------
    def GetLoadMode(self):
        return self._lib.GetLoadMode(self._h)

    def IsStatic(self):
        return bool(self._lib.IsStatic(self._h))

    def SetMaxAudioBufferSize(self, max_size):
        self._lib.SetMaxAudioBufferSize(self._h, int(max_size))

    def GetRecommendedInputDBAdjustment(self):
        return float(self._lib.GetRecommendedInputDBAdjustment(self._h))

    def GetRecommendedOutputDBAdjustment(self):
        return float(self._lib.GetRecommendedOutputDBAdjustment(self._h))

    def GetSampleRate(self):
        return float(self._lib.GetSampleRate(self._h))

    def GetReceptiveFieldSize(self):
        return int(self._lib.NA_GetReceptiveFieldSize(self._h))

    def HasQualityScaling(self):
        return bool(self._lib.NA_HasQualityScaling(self._h))

    def GetQualityScaleFactor(self):
        return float(self._lib.NA_GetQualityScaleFactor(self._h))

    def SetQualityScaleFactor(self, q):
        self._lib.NA_SetQualityScaleFactor(self._h, float(q))

    def GetModelVersion(self):
        buf = C.create_string_buffer(256)
        self._lib.NA_GetModelVersion(self._h, buf, 256)
        return buf.value.decode()

    def GetMetadata(self, field):
        buf = C.create_string_buffer(65536)
        self._lib.NA_GetMetadata(self._h, field.encode(), buf, 65536)
        return buf.value.decode()

    def Prewarm(self):
        if self._lib.NA_Prewarm(self._h) != 0:
            raise NeuralAudioError(capi.last_error())

    def Process(self, x):
        """input -> output (same length), any number of samples; runs on the GPU."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        if self._lib.NA_ProcessChecked(self._h, _fptr(x), _fptr(y), x.size) != 0:
            raise NeuralAudioError(capi.last_error())
        return y

    def IsQualityChangeRealtimeSafe(self, q):
        return bool(self._lib.NA_IsQualityChangeRealtimeSafe(self._h, float(q)))

    def KernelInfo(self, quality=1.0, streams=1):
        """Host side only: the kernel family a batch of `streams` streams would run on and the f16-split range proof behind the choice."""
        name = C.create_string_buffer(64)
        lim = C.c_float(0.0)
        proven, wok, pack = C.c_int(0), C.c_int(0), C.c_int(0)
        if self._lib.NA_ModelKernelInfo(self._h, float(quality), int(streams), name, 64, C.byref(lim), C.byref(proven), C.byref(wok), C.byref(pack)) != 0:
            raise NeuralAudioError(capi.last_error())
        return {"kernel": name.value.decode(), "input_limit": float(lim.value), "range_proven": bool(proven.value), "weights_ok": bool(wok.value),
                "pack": int(pack.value)}

    def close(self):
        if self._h:
            self._lib.DeleteModel(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NeuralModelLoader:
    """NeuralAudio::NeuralModelLoader (settings + factories)."""

    def __init__(self):
        self._lib = capi.load_library()
        self._h = self._lib.CreateLoader()
        if not self._h:
            raise NeuralAudioError(capi.last_error())

    def SetLSTMLoadMode(self, mode):
        self._lib.SetLSTMLoadMode(self._h, int(mode))

    def SetWaveNetLoadMode(self, mode):
        self._lib.SetWaveNetLoadMode(self._h, int(mode))

    def SetAudioInputLevelDBu(self, dbu):
        self._lib.SetAudioInputLevelDBu(self._h, float(dbu))

    def SetDefaultMaxAudioBufferSize(self, n):
        self._lib.SetDefaultMaxAudioBufferSize(self._h, int(n))

    def SetDefaultQualityScaleFactor(self, q):
        self._lib.NA_SetDefaultQualityScaleFactor(self._h, float(q))

    def SetExternalSampleRate(self, sr):
        self._lib.NA_SetExternalSampleRate(self._h, int(sr))

    def SetDevice(self, device):
        self._lib.NA_SetDevice(self._h, int(device))

    def SetWaveNetMathMode(self, mode):
        self._lib.NA_SetWaveNetMathMode(self._h, int(mode))

    def SetLSTMMathMode(self, mode):
        self._lib.NA_SetLSTMMathMode(self._h, int(mode))

    def SetCompositeModelLoadMode(self, mode):
        self._lib.NA_SetCompositeModelLoadMode(self._h, int(mode))

    def CreateFromFile(self, path, doPrewarm=True, use_wchar_entry=False):
        """Returns None when the file is missing / unsupported (reference: nullptr); raises on malformed files."""
        if use_wchar_entry:
            h = self._lib.CreateModelFromFile(self._h, str(path))
        else:
            h = self._lib.NA_CreateModelFromFileUtf8(self._h, str(path).encode(), 1 if doPrewarm else 0)
        if not h:
            err = capi.last_error()
            if "not found or not supported" in err or "model not supported" in err:
                return None
            raise NeuralAudioError(err)
        return NeuralModel(h)

    def CreateFromString(self, text, extension, doPrewarm=True):
        h = self._lib.NA_CreateModelFromString(self._h, text.encode(), extension.encode(), 1 if doPrewarm else 0)
        if not h:
            err = capi.last_error()
            if "model not supported" in err:
                return None
            raise NeuralAudioError(err)
        return NeuralModel(h)

    def close(self):
        if self._h:
            self._lib.DeleteLoader(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """Many independent streams on one GPU; row s of the [streams][n] arrays is stream s."""

    def __init__(self, device=0, hip_stream=None):
        self._lib = capi.load_library()
        self._h = self._lib.NA_BatchCreate(int(device), C.c_void_p(hip_stream) if hip_stream else None)
        if not self._h:
            raise NeuralAudioError(capi.last_error())

    def AddStreams(self, model, count=1, quality=1.0, doPrewarm=True):
        first = self._lib.NA_BatchAddStreams(self._h, model._h, float(quality), int(count), 1 if doPrewarm else 0)
        if first < 0:
            raise NeuralAudioError(capi.last_error())
        return first

    def NumStreams(self):
        return int(self._lib.NA_BatchNumStreams(self._h))

    def NumLiveStreams(self):
        return int(self._lib.NA_BatchNumLiveStreams(self._h))

    def IsLive(self, stream):
        return bool(self._lib.NA_BatchIsLive(self._h, int(stream)))

    def RemoveStreams(self, first, count=1):
        if self._lib.NA_BatchRemoveStreams(self._h, int(first), int(count)) != 0:
            raise NeuralAudioError(capi.last_error())

    def SetQuality(self, stream, q):
        if self._lib.NA_BatchSetQuality(self._h, int(stream), float(q)) != 0:
            raise NeuralAudioError(capi.last_error())

    def IsQualityChangeRealtimeSafe(self, stream, q):
        return bool(self._lib.NA_BatchIsQualityChangeRealtimeSafe(self._h, int(stream), float(q)))

    def GetActiveSubModel(self, stream):
        return int(self._lib.NA_BatchGetActiveSubModel(self._h, int(stream)))

    def Prewarm(self, stream=-1):
        if self._lib.NA_BatchPrewarm(self._h, int(stream)) != 0:
            raise NeuralAudioError(capi.last_error())

    def Process(self, x):
        """x: host array [streams, n] -> host array [streams, n] (synchronous)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[0] == self.NumStreams(), "expected [streams, n]"
        y = np.empty_like(x)
        if self._lib.NA_BatchProcess(self._h, _fptr(x), _fptr(y), x.shape[1]) != 0:
            raise NeuralAudioError(capi.last_error())
        return y

    def Submit(self, x):
        """Pipelined variant of Process: returns a ticket for Collect(); up to 3 buffers may be in flight."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[0] == self.NumStreams(), "expected [streams, n]"
        t = self._lib.NA_BatchSubmit(self._h, _fptr(x), x.shape[1])
        if t < 0:
            raise NeuralAudioError(capi.last_error())
        return t, x.shape

    def Collect(self, ticket):
        t, shape = ticket
        y = np.empty(shape, np.float32)
        if self._lib.NA_BatchCollect(self._h, int(t), _fptr(y)) != 0:
            raise NeuralAudioError(capi.last_error())
        return y

    def NextInput(self, n):
        """Zero-copy: a numpy view [streams, n] of the pinned staging buffer of the next submission; fill it, then SubmitInput(n)."""
        p = self._lib.NA_BatchNextInput(self._h, int(n))
        if not p:
            raise NeuralAudioError(capi.last_error())
        return np.ctypeslib.as_array(p, shape=(self.NumStreams(), int(n)))

    def SubmitInput(self, n):
        t = self._lib.NA_BatchSubmit(self._h, None, int(n))
        if t < 0:
            raise NeuralAudioError(capi.last_error())
        return t, (self.NumStreams(), int(n))

    def CollectView(self, ticket):
        """Zero-copy: waits for the buffer and returns a numpy view of the pinned result (valid for the next 2 submissions)."""
        t, shape = ticket
        if self._lib.NA_BatchCollect(self._h, int(t), None) != 0:
            raise NeuralAudioError(capi.last_error())
        return np.ctypeslib.as_array(self._lib.NA_BatchOutputView(self._h, int(t)), shape=shape)

    def ProcessDevice(self, d_in, d_out, n, in_stride=None, out_stride=None):
        """d_in / d_out: raw device pointers (ints); asynchronous on the batch's HIP stream."""
        rc = self._lib.NA_BatchProcessDevice(self._h, C.c_void_p(d_in), C.c_void_p(d_out), int(n),
                                             int(in_stride if in_stride is not None else n),
                                             int(out_stride if out_stride is not None else n))
        if rc != 0:
            raise NeuralAudioError(capi.last_error())

    def Synchronize(self):
        if self._lib.NA_BatchSynchronize(self._h) != 0:
            raise NeuralAudioError(capi.last_error())

    def WaitOutputs(self):
        """Host-side wait until every buffer handed to ProcessDevice so far has been processed (the resident launch stays up)."""
        if self._lib.NA_BatchWaitOutputs(self._h) != 0:
            raise NeuralAudioError(capi.last_error())

    def SetWaitLimitMs(self, ms):
        """Wall-clock limit of every host-side wait of this batch (NA_BatchSetWaitLimitMs; <= 0: none).  A wait that runs into it breaks the batch."""
        self._lib.NA_BatchSetWaitLimitMs(self._h, float(ms))

    def GetWaitLimitMs(self):
        return float(self._lib.NA_BatchGetWaitLimitMs(self._h))

    def IsBroken(self):
        return bool(self._lib.NA_BatchIsBroken(self._h))

    def DebugStallDevice(self, ms):
        """Test hook: keep the batch's streams busy for `ms` milliseconds (NA_DebugStallDevice)."""
        if self._lib.NA_DebugStallDevice(self._h, float(ms)) != 0:
            raise NeuralAudioError(capi.last_error())

    def SetResidentLaunch(self, on=True):
        """Opt in to (or out of) the resident launch for device-pointer buffers of this batch (NA_BatchSetResidentLaunch)."""
        if self._lib.NA_BatchSetResidentLaunch(self._h, 1 if on else 0) != 0:
            raise NeuralAudioError(capi.last_error())

    def UsesResidentLaunch(self):
        """True when the last ProcessDevice call was a command to the resident launch (own stream, >= 512 A1 Standard streams)."""
        return bool(self._lib.NA_BatchUsesResidentLaunch(self._h))

    def GetHipStream(self):
        """The batch's HIP stream handle (int).  From the first call on every launch is ordered on it (see NA_BatchGetHipStream)."""
        return self._lib.NA_BatchGetHipStream(self._h)

    def MarkTime(self, which):
        """HIP events on every stream the batch launches on (0: start, 1: end); see ElapsedMs."""
        if self._lib.NA_BatchMarkTime(self._h, int(which)) != 0:
            raise NeuralAudioError(capi.last_error())

    def WaitMarks(self):
        """Polls until the marks of MarkTime(1) are reached on every stream the batch launches on."""
        if self._lib.NA_BatchWaitMarks(self._h) != 0:
            raise NeuralAudioError(capi.last_error())

    def ElapsedMs(self):
        ms = float(self._lib.NA_BatchElapsedMs(self._h))
        if ms < 0:
            raise NeuralAudioError(capi.last_error())
        return ms

    def UsesHalfLaunches(self):
        """True when the last ProcessDevice call ran as two free-running half-batch launches (own stream, one WaveNet group)."""
        return bool(self._lib.NA_BatchUsesHalfLaunches(self._h))

    def AlgorithmicBytesPerSample(self, block_frames=128):
        return float(self._lib.NA_BatchAlgorithmicBytesPerSample(self._h, int(block_frames)))

    def MacsPerSample(self):
        return float(self._lib.NA_BatchMacsPerSample(self._h))

    def StateBytes(self):
        return float(self._lib.NA_BatchStateBytes(self._h))

    def StreamKernelName(self, stream):
        return self._lib.NA_BatchStreamKernelName(self._h, int(stream)).decode()

    def StreamInputLimit(self, stream):
        return float(self._lib.NA_BatchStreamInputLimit(self._h, int(stream)))

    def StreamPackFactor(self, stream):
        return int(self._lib.NA_BatchStreamPackFactor(self._h, int(stream)))

    def StreamRangeEvents(self, stream):
        r = int(self._lib.NA_BatchStreamRangeEvents(self._h, int(stream)))
        if r < 0:
            raise NeuralAudioError(capi.last_error())
        return r

    def close(self):
        if self._h:
            self._lib.NA_BatchDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiBatch:
    """The C++ multi-GPU host (csrc/multi_gpu.cpp): one batch + one host thread per entry of `devices`, the global stream list sharded
    across them by cost.  Rows of the [streams][n] arrays are global stream ids."""

    def __init__(self, devices):
        self._lib = capi.load_library()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = self._lib.NA_MultiCreate(arr, len(devices))
        if not self._h:
            raise NeuralAudioError(capi.last_error())

    def AddStreams(self, model, count=1, quality=1.0, doPrewarm=True):
        first = self._lib.NA_MultiAddStreams(self._h, model._h, float(quality), int(count), 1 if doPrewarm else 0)
        if first < 0:
            raise NeuralAudioError(capi.last_error())
        return first

    def SetFanIn(self, mode):
        """"host" (default): every shard serves its own rows of the host arrays; "rccl": weights replicated and outputs gathered over RCCL."""
        if self._lib.NA_MultiSetFanIn(self._h, {"host": 0, "rccl": 1}[mode]) != 0:
            raise NeuralAudioError(capi.last_error())

    def Commit(self):
        if self._lib.NA_MultiCommit(self._h) != 0:
            raise NeuralAudioError(capi.last_error())

    def NumStreams(self):
        return int(self._lib.NA_MultiNumStreams(self._h))

    def ShardRanges(self):
        out = []
        for s in range(int(self._lib.NA_MultiNumShards(self._h))):
            b, e, d = C.c_int(), C.c_int(), C.c_int()
            if self._lib.NA_MultiShardRange(self._h, s, C.byref(b), C.byref(e), C.byref(d)) != 0:
                raise NeuralAudioError(capi.last_error())
            out.append((b.value, e.value, d.value))
        return out

    def Process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[0] == self.NumStreams(), "expected [streams, n]"
        y = np.empty_like(x)
        if self._lib.NA_MultiProcess(self._h, _fptr(x), _fptr(y), x.shape[1]) != 0:
            raise NeuralAudioError(capi.last_error())
        return y

    def GatheredOutput(self, shard, n):
        """RCCL fan-in: the [streams][n] device buffer of the last Process() on `shard`'s GPU, copied to the host (tests)."""
        ptr = self._lib.NA_MultiGatheredOutput(self._h, int(shard))
        if not ptr:
            raise NeuralAudioError("no gathered output (RCCL fan-in only, after Process)")
        out = np.empty((self.NumStreams(), int(n)), dtype=np.float32)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        if hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2) != 0:  # hipMemcpyDeviceToHost
            raise NeuralAudioError("hipMemcpy of the gathered output failed")
        return out

    def SetQuality(self, stream, q):
        if self._lib.NA_MultiSetQuality(self._h, int(stream), float(q)) != 0:
            raise NeuralAudioError(capi.last_error())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.NA_MultiDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
