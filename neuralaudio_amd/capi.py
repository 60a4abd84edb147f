"""ctypes binding of libNeuralAudioCAPI.so -- the reference-side binding a maintainer would write.

Everything here goes through the C ABI declared in include/NeuralAudioCApi.h (the 15 legacy symbols
of the reference's NeuralAudioCAPI) and include/neuralaudio_amd.h (the additive NA_* batch API).
There is no Python compute path and no CPU fallback: if the shared library (with its embedded gfx950
code objects) is missing, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NA_LIB_SUFFIX selects a tuning build (tools/ablate.sh); unset in normal use
_SUFFIX = os.environ.get("NA_LIB_SUFFIX", "")
LIB_PATH = (os.path.join(os.path.dirname(_HERE), "tools", "variants", "libNeuralAudioCAPI%s.so" % _SUFFIX) if _SUFFIX
            else os.path.join(_HERE, "libNeuralAudioCAPI.so"))

LEGACY_SYMBOLS = [
    "CreateLoader", "DeleteLoader", "CreateModelFromFile", "DeleteModel", "SetLSTMLoadMode", "SetWaveNetLoadMode",
    "SetAudioInputLevelDBu", "SetDefaultMaxAudioBufferSize", "GetLoadMode", "IsStatic", "SetMaxAudioBufferSize",
    "GetRecommendedInputDBAdjustment", "GetRecommendedOutputDBAdjustment", "GetSampleRate", "Process",
]

NA_SYMBOLS = [
    "NA_GetLastError", "NA_GetDeviceCount", "NA_GetVersion", "NA_CreateModelFromFileUtf8", "NA_CreateModelFromString",
    "NA_SetDevice", "NA_SetDefaultQualityScaleFactor", "NA_SetExternalSampleRate", "NA_HasQualityScaling",
    "NA_GetQualityScaleFactor", "NA_SetQualityScaleFactor", "NA_GetReceptiveFieldSize", "NA_Prewarm", "NA_GetMetadata",
    "NA_GetModelVersion", "NA_BatchCreate", "NA_BatchDestroy", "NA_BatchAddStreams", "NA_BatchNumStreams",
    "NA_BatchSetQuality", "NA_BatchGetActiveSubModel", "NA_BatchPrewarm", "NA_BatchProcess", "NA_BatchProcessDevice",
    "NA_BatchSynchronize", "NA_BatchGetHipStream", "NA_BatchAlgorithmicBytesPerSample", "NA_BatchMacsPerSample",
    "NA_BatchStateBytes", "NA_BatchStreamPackFactor", "NA_BatchStreamKernelName", "NA_DebugSetTraceBuffer", "NA_DebugSetWaveNetSpec", "NA_DebugSetRecurrentQuadMin", "NA_DebugRecurrentQuadLaunches", "NA_RegisterHostBuffer", "NA_UnregisterHostBuffer", "NA_BatchStreamInputLimit", "NA_BatchRemoveStreams", "NA_MultiCreate", "NA_MultiDestroy", "NA_MultiAddStreams", "NA_MultiCommit", "NA_MultiNumStreams", "NA_MultiNumShards", "NA_MultiShardRange", "NA_MultiProcess", "NA_MultiSubmit", "NA_MultiCollect", "NA_MultiSetQuality", "NA_ShardByCost", "NA_ModelStreamCost", "NA_BatchNumLiveStreams", "NA_BatchIsLive", "NA_SetWaveNetMathMode", "NA_SetLSTMMathMode", "NA_SetCompositeModelLoadMode",
    "NA_IsQualityChangeRealtimeSafe", "NA_ProcessChecked", "NA_BatchSubmit", "NA_BatchCollect", "NA_BatchNextInput", "NA_BatchOutputView", "NA_BatchIsQualityChangeRealtimeSafe", "NA_DebugClassifyNam", "NA_DebugPackedWeights", "NA_ModelKernelInfo", "NA_BatchStreamRangeEvents", "NA_MultiSetFanIn", "NA_MultiGatheredOutput", "NA_RcclAvailable",
    "NA_BatchMarkTime", "NA_BatchWaitMarks", "NA_BatchElapsedMs", "NA_BatchUsesHalfLaunches", "NA_DebugSetRcclApi", "NA_BatchWaitOutputs", "NA_BatchUsesResidentLaunch", "NA_BatchSetResidentLaunch",
    "NA_BatchSetWaitLimitMs", "NA_BatchGetWaitLimitMs", "NA_BatchIsBroken", "NA_DebugStallDevice",
]

_lib = None


def load_library():
    """Load the shared library; raises OSError if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libNeuralAudioCAPI.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                      "g.build()'` (there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    sig = {
        "CreateLoader": (vp, []),
        "DeleteLoader": (None, [vp]),
        "CreateModelFromFile": (vp, [vp, C.c_wchar_p]),
        "DeleteModel": (None, [vp]),
        "SetLSTMLoadMode": (None, [vp, C.c_int]),
        "SetWaveNetLoadMode": (None, [vp, C.c_int]),
        "SetAudioInputLevelDBu": (None, [vp, C.c_float]),
        "SetDefaultMaxAudioBufferSize": (None, [vp, C.c_int]),
        "GetLoadMode": (C.c_int, [vp]),
        "IsStatic": (C.c_bool, [vp]),
        "SetMaxAudioBufferSize": (None, [vp, C.c_int]),
        "GetRecommendedInputDBAdjustment": (C.c_float, [vp]),
        "GetRecommendedOutputDBAdjustment": (C.c_float, [vp]),
        "GetSampleRate": (C.c_float, [vp]),
        "Process": (None, [vp, fp, fp, C.c_size_t]),
        "NA_GetLastError": (C.c_char_p, []),
        "NA_GetDeviceCount": (C.c_int, []),
        "NA_GetVersion": (C.c_char_p, []),
        "NA_CreateModelFromFileUtf8": (vp, [vp, C.c_char_p, C.c_int]),
        "NA_CreateModelFromString": (vp, [vp, C.c_char_p, C.c_char_p, C.c_int]),
        "NA_SetDevice": (None, [vp, C.c_int]),
        "NA_SetDefaultQualityScaleFactor": (None, [vp, C.c_float]),
        "NA_SetExternalSampleRate": (None, [vp, C.c_int]),
        "NA_HasQualityScaling": (C.c_int, [vp]),
        "NA_GetQualityScaleFactor": (C.c_float, [vp]),
        "NA_SetQualityScaleFactor": (None, [vp, C.c_float]),
        "NA_GetReceptiveFieldSize": (C.c_int, [vp]),
        "NA_Prewarm": (C.c_int, [vp]),
        "NA_GetMetadata": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_int]),
        "NA_GetModelVersion": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "NA_BatchCreate": (vp, [C.c_int, vp]),
        "NA_BatchDestroy": (None, [vp]),
        "NA_BatchAddStreams": (C.c_int, [vp, vp, C.c_float, C.c_int, C.c_int]),
        "NA_BatchNumStreams": (C.c_int, [vp]),
        "NA_BatchSetQuality": (C.c_int, [vp, C.c_int, C.c_float]),
        "NA_BatchGetActiveSubModel": (C.c_int, [vp, C.c_int]),
        "NA_BatchPrewarm": (C.c_int, [vp, C.c_int]),
        "NA_BatchProcess": (C.c_int, [vp, fp, fp, C.c_size_t]),
        "NA_BatchProcessDevice": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_long, C.c_long]),
        "NA_BatchSynchronize": (C.c_int, [vp]),
        "NA_BatchGetHipStream": (vp, [vp]),
        "NA_BatchMarkTime": (C.c_int, [vp, C.c_int]),
        "NA_BatchWaitMarks": (C.c_int, [vp]),
        "NA_BatchElapsedMs": (C.c_float, [vp]),
        "NA_BatchUsesHalfLaunches": (C.c_int, [vp]),
        "NA_BatchUsesResidentLaunch": (C.c_int, [vp]),
        "NA_BatchSetResidentLaunch": (C.c_int, [vp, C.c_int]),
        "NA_BatchWaitOutputs": (C.c_int, [vp]),
        "NA_BatchSetWaitLimitMs": (C.c_int, [vp, C.c_double]),
        "NA_BatchGetWaitLimitMs": (C.c_double, [vp]),
        "NA_BatchIsBroken": (C.c_int, [vp]),
        "NA_DebugStallDevice": (C.c_int, [vp, C.c_double]),
        "NA_BatchAlgorithmicBytesPerSample": (C.c_double, [vp, C.c_int]),
        "NA_BatchMacsPerSample": (C.c_double, [vp]),
        "NA_BatchStateBytes": (C.c_double, [vp]),
        "NA_BatchStreamPackFactor": (C.c_int, [vp, C.c_int]),
        "NA_BatchStreamKernelName": (C.c_char_p, [vp, C.c_int]),
        "NA_DebugSetTraceBuffer": (None, [vp]),
        "NA_DebugSetWaveNetSpec": (None, [C.c_int]),
        "NA_DebugSetRcclApi": (None, [C.c_int, C.c_int, C.c_int]),
        "NA_DebugSetRecurrentQuadMin": (C.c_int, [C.c_int]),
        "NA_DebugRecurrentQuadLaunches": (C.c_longlong, []),
        "NA_RegisterHostBuffer": (C.c_int, [C.c_void_p, C.c_size_t]),
        "NA_UnregisterHostBuffer": (C.c_int, [C.c_void_p]),
        "NA_BatchStreamInputLimit": (C.c_float, [vp, C.c_int]),
        "NA_BatchRemoveStreams": (C.c_int, [vp, C.c_int, C.c_int]),
        "NA_MultiCreate": (vp, [C.POINTER(C.c_int), C.c_int]),
        "NA_MultiDestroy": (None, [vp]),
        "NA_MultiAddStreams": (C.c_int, [vp, vp, C.c_float, C.c_int, C.c_int]),
        "NA_MultiCommit": (C.c_int, [vp]),
        "NA_MultiNumStreams": (C.c_int, [vp]),
        "NA_MultiNumShards": (C.c_int, [vp]),
        "NA_MultiShardRange": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "NA_MultiProcess": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t]),
        "NA_MultiSubmit": (C.c_int, [vp, C.POINTER(C.c_float), C.c_size_t]),
        "NA_MultiCollect": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float)]),
        "NA_MultiSetQuality": (C.c_int, [vp, C.c_int, C.c_float]),
        "NA_ShardByCost": (C.c_int, [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_int)]),
        "NA_ModelStreamCost": (C.c_double, [vp, C.c_float]),
        "NA_BatchStreamRangeEvents": (C.c_int, [vp, C.c_int]),
        "NA_MultiSetFanIn": (C.c_int, [vp, C.c_int]),
        "NA_MultiGatheredOutput": (vp, [vp, C.c_int]),
        "NA_RcclAvailable": (C.c_int, []),
        "NA_ModelKernelInfo": (C.c_int, [vp, C.c_float, C.c_int, C.c_char_p, C.c_int, fp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "NA_BatchNumLiveStreams": (C.c_int, [vp]),
        "NA_BatchIsLive": (C.c_int, [vp, C.c_int]),
        "NA_SetWaveNetMathMode": (None, [vp, C.c_int]),
        "NA_SetLSTMMathMode": (None, [vp, C.c_int]),
        "NA_SetCompositeModelLoadMode": (None, [vp, C.c_int]),
        "NA_IsQualityChangeRealtimeSafe": (C.c_int, [vp, C.c_float]),
        "NA_ProcessChecked": (C.c_int, [vp, fp, fp, C.c_size_t]),
        "NA_BatchSubmit": (C.c_int, [vp, fp, C.c_size_t]),
        "NA_BatchCollect": (C.c_int, [vp, C.c_int, fp]),
        "NA_BatchNextInput": (fp, [vp, C.c_size_t]),
        "NA_BatchOutputView": (fp, [vp, C.c_int]),
        "NA_BatchIsQualityChangeRealtimeSafe": (C.c_int, [vp, C.c_int, C.c_float]),
        "NA_DebugClassifyNam": (C.c_int, [C.c_char_p]),
        "NA_DebugPackedWeights": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # a tuning build from an older source tree (NA_LIB_SUFFIX: A/B benches against it) may lack the newest additive entry
            # points; the default library must have every one of them
            if os.environ.get("NA_LIB_SUFFIX") and name.startswith("NA_"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load_library().NA_GetLastError().decode("utf-8", "replace")


def device_count():
    return int(load_library().NA_GetDeviceCount())
