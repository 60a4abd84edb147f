/*
 * neuralaudio_amd.h -- additive C ABI of the MI355X-native library: the many-stream batch engine the
 * reference lacks, plus C access to the NeuralModel virtuals the legacy C API never exported.
 *
 * Plain pointers and sizes only (no C++/torch types).  Every function returning int returns 0 on success
 * and a negative value on failure; the failure text is available from NA_GetLastError() (thread-local).
 *
 * The batch is the data-parallel drop-in for "N hosts each calling NeuralModel::Process"
 * (NeuralAudio/NeuralModel.h:127): stream s of the batch is bit-for-bit what a single NeuralModel created
 * from the same file computes, so row s of `in`/`out` replaces the s-th host's Process(input, output, n).
 */
#ifndef NEURALAUDIO_AMD_H
#define NEURALAUDIO_AMD_H

#include "NeuralAudioCApi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct NA_Batch NA_Batch;

/* ---- library / device ------------------------------------------------------------------------- */
NA_EXTERN const char* NA_GetLastError(void);
NA_EXTERN int NA_GetDeviceCount(void);                 /* 0 when no HIP device / driver is present */
NA_EXTERN const char* NA_GetVersion(void);

/* ---- loader / model extras (NeuralModelLoader setters NeuralModel.h:155-221, virtuals :40-134) -- */
NA_EXTERN NeuralModel* NA_CreateModelFromFileUtf8(NeuralModelLoader* loader, const char* utf8Path, int doPrewarm);
NA_EXTERN NeuralModel* NA_CreateModelFromString(NeuralModelLoader* loader, const char* jsonText, const char* extension, int doPrewarm);
NA_EXTERN void NA_SetDevice(NeuralModelLoader* loader, int device);
/* activation arithmetic of the models created next: 0 = FastMath (default), 1 = StdMath -- the reference's build options
 * WAVENET_MATH / LSTM_MATH (NeuralAudio/CMakeLists.txt:82-96, Activation.h:12-118) as load-time knobs */
NA_EXTERN void NA_SetWaveNetMathMode(NeuralModelLoader* loader, int mathMode);
NA_EXTERN void NA_SetLSTMMathMode(NeuralModelLoader* loader, int mathMode);
/* ECompositeModelLoadMode (NeuralModel.h:27-31,166-174): 0 = LoadAll (default), 1 = OnDemand */
NA_EXTERN void NA_SetCompositeModelLoadMode(NeuralModelLoader* loader, int loadMode);
/* NeuralModel::IsQualityChangeRealtimeSafe (NeuralModel.h:54-59) */
NA_EXTERN int NA_IsQualityChangeRealtimeSafe(NeuralModel* model, float newQuality);
/* NeuralModel::Process with a status: 0 ok; on failure `output` is zero-filled (silence) and NA_GetLastError() says why.
 * The legacy Process() symbol forwards here and drops the status. */
NA_EXTERN int NA_ProcessChecked(NeuralModel* model, float* input, float* output, size_t numSamples);
NA_EXTERN void NA_SetDefaultQualityScaleFactor(NeuralModelLoader* loader, float quality);
NA_EXTERN void NA_SetExternalSampleRate(NeuralModelLoader* loader, int sampleRate);
NA_EXTERN int NA_HasQualityScaling(NeuralModel* model);
NA_EXTERN float NA_GetQualityScaleFactor(NeuralModel* model);
NA_EXTERN void NA_SetQualityScaleFactor(NeuralModel* model, float quality);
NA_EXTERN int NA_GetReceptiveFieldSize(NeuralModel* model);
NA_EXTERN int NA_Prewarm(NeuralModel* model);
/* copies the JSON text of a metadata field into buf (NUL-terminated, truncated); returns its full length */
NA_EXTERN int NA_GetMetadata(NeuralModel* model, const char* fieldName, char* buf, int bufSize);
NA_EXTERN int NA_GetModelVersion(NeuralModel* model, char* buf, int bufSize);

/* ---- batch engine -------------------------------------------------------------------------------- */
/* One batch == one GPU.  hipStream: NULL -> the batch creates its own non-blocking HIP stream;
 * otherwise the caller's hipStream_t is borrowed (e.g. torch.cuda.current_stream().cuda_stream). */
NA_EXTERN NA_Batch* NA_BatchCreate(int device, void* hipStream);
NA_EXTERN void NA_BatchDestroy(NA_Batch* batch);
/* Adds `count` streams running `model` (weights are shared on the device); returns the id (= row) of the
 * first one, ids are consecutive; negative on failure.  quality is used by SlimmableContainer models.
 * Ids retired by NA_BatchRemoveStreams are recycled first: the lowest retired id for count == 1, a run of `count` consecutive retired
 * ids when there is one; otherwise new rows are appended.  The device layout of a model's streams (e.g. narrow WaveNet models run
 * several streams per kernel-level stream) does not depend on how the streams arrived: 4096 single adds == one add of 4096. */
NA_EXTERN int NA_BatchAddStreams(NA_Batch* batch, NeuralModel* model, float quality, int count, int doPrewarm);
/* Stream lifetime = the reference's model lifetime (NeuralAudioCApi.cpp:38-42 DeleteModel): frees the device state of streams
 * [first, first + count) for recycling and retires their ids.  Rows keep their place in the [streams][n] arrays -- input ignored, host
 * output zero -- except trailing retired rows, which leave the arrays (check NA_BatchNumStreams afterwards).  Waits for the batch's
 * stream; call it between buffers, not from the audio callback.  Fails (negative) on ids that are out of range or already removed. */
NA_EXTERN int NA_BatchRemoveStreams(NA_Batch* batch, int first, int count);
NA_EXTERN int NA_BatchNumStreams(NA_Batch* batch);     /* rows of the [streams][n] arrays, retired ids included */
NA_EXTERN int NA_BatchNumLiveStreams(NA_Batch* batch);
NA_EXTERN int NA_BatchIsLive(NA_Batch* batch, int stream);
NA_EXTERN int NA_BatchSetQuality(NA_Batch* batch, int stream, float quality);
NA_EXTERN int NA_BatchGetActiveSubModel(NA_Batch* batch, int stream);
/* 1 when NA_BatchSetQuality(stream, quality) costs the next NA_BatchProcess* call no allocation / synchronisation / prewarm */
NA_EXTERN int NA_BatchIsQualityChangeRealtimeSafe(NA_Batch* batch, int stream, float quality);
NA_EXTERN int NA_BatchPrewarm(NA_Batch* batch, int stream); /* stream < 0: all */
/* host pointers, layout [streams][n]; synchronous */
NA_EXTERN int NA_BatchProcess(NA_Batch* batch, const float* in, float* out, size_t n);
/* Optional, for hosts that reuse their buffers: a registered block (pinned and mapped: hipHostRegister) is read / written by the kernels
 * as it is when NA_BatchProcess is handed pointers inside it -- no staging copies (1024 x 128: 81 -> ~62 us per call).  Register once,
 * outside the audio path (it pins pages: milliseconds); the block must stay allocated until NA_UnregisterHostBuffer.  0 on success. */
NA_EXTERN int NA_RegisterHostBuffer(void* ptr, size_t bytes);
NA_EXTERN int NA_UnregisterHostBuffer(void* ptr);
/* Pipelined host-buffer interface: NA_BatchSubmit copies `in` ([streams][n]) and enqueues upload, kernels and download, returning a
 * ticket (>= 0; up to 3 may be in flight); NA_BatchCollect blocks until that buffer is done and copies its [streams][n] result to
 * `out`.  Uploads / downloads of neighbouring buffers overlap the kernels.  Buffers are processed in submission order. */
NA_EXTERN int NA_BatchSubmit(NA_Batch* batch, const float* in, size_t n);
NA_EXTERN int NA_BatchCollect(NA_Batch* batch, int ticket, float* out);
/* Zero-copy variants: NA_BatchNextInput returns the pinned [streams][n] staging buffer of the next submission -- fill it, then call
 * NA_BatchSubmit(batch, NULL, n); NA_BatchCollect(batch, ticket, NULL) only waits, and NA_BatchOutputView(batch, ticket) is the pinned
 * result, valid until that slot is submitted again (3 submissions later). */
NA_EXTERN float* NA_BatchNextInput(NA_Batch* batch, size_t n);
NA_EXTERN const float* NA_BatchOutputView(NA_Batch* batch, int ticket);
/* DEVICE pointers, row s = stream s, rows `stride` floats apart.  Two contracts, by who owns the stream:
 *
 * (a) the batch runs on a stream of the CALLER (NA_BatchCreate with a stream handle), or the caller has fetched the batch's own stream
 *     (NA_BatchGetHipStream): every launch is ordered on that stream, like any HIP kernel launch -- a producer kernel enqueued on it
 *     before the call and a consumer enqueued after it need no other synchronisation, and the call may be made from a device-side
 *     pipeline without any host wait.
 *
 * (b) the batch created its own stream (hipStream == NULL) and nobody has fetched it: the library schedules the buffer itself -- as
 *     two free-running launches of half the streams each on internal streams, or, where NA_BatchSetResidentLaunch asked for it, for
 *     large A1 Standard batches as a command to ONE resident launch that stays on the chip and walks consecutive buffers
 *     (csrc/gpu_batch_chains.cpp) -- none of which is ordered against any stream the caller knows.  The contract is then a HOST-side one:
 *       - the input rows must be COMPLETE in device memory when NA_BatchProcessDevice is called (synchronise their producer first:
 *         hipStreamSynchronize / hipEventSynchronize on its stream) and must stay untouched until the step's outputs are valid;
 *       - the output rows are valid after NA_BatchWaitOutputs (cheap: the resident launch stays up) or NA_BatchSynchronize (everything
 *         of the batch is idle, the resident launch has left the chip);
 *       - a caller that re-uses ONE output buffer for consecutive steps can only ever read the rows of the last step it waited for: give
 *         every step that is in flight its own output rows (up to 63 steps may be in flight; the call blocks beyond that);
 *       - a device-side producer / consumer that must not wait on the host cannot use (b): create the batch on its stream, contract (a).
 *     Inside the resident launch the rows are read and written at system scope, so a producer that ran on another stream / XCD between
 *     two steps is seen without a kernel boundary (tests/test_gpu_resident.py: a producer kernel on a foreign stream rewrites the same
 *     input buffer before every step). */
NA_EXTERN int NA_BatchProcessDevice(NA_Batch* batch, const float* dIn, float* dOut, size_t n, long inStride, long outStride);
/* every buffer handed to NA_BatchProcessDevice so far has been processed: its output rows are valid (host-side wait; contract (b)) */
NA_EXTERN int NA_BatchWaitOutputs(NA_Batch* batch);
NA_EXTERN int NA_BatchSynchronize(NA_Batch* batch);
/* Bounded waits.  Process is called from a real-time thread that must get its call back (NeuralAudio/NeuralModel.h:127): every host-side
 * wait of the processing entry points -- NA_BatchProcess, NA_BatchCollect, NA_BatchWaitOutputs, NA_BatchSynchronize, the timing marks,
 * the legacy Process / NA_ProcessChecked (a batch of one) -- gives up after a wall-clock limit: default 2000 ms, environment
 * NA_WAIT_LIMIT_MS for every batch of the process, NA_BatchSetWaitLimitMs for one batch (<= 0: wait without a limit).  A wait that runs
 * into the limit marks the batch BROKEN: the call returns non-zero with the reason in NA_GetLastError(), NA_BatchProcess / Process
 * hand back silence (zeros), and every later call on the batch fails at once without touching the device (the stream states are no
 * longer what the caller thinks they are).  A broken batch can only be destroyed; NA_BatchDestroy gives the device one more limit to
 * come back and otherwise leaves the device allocations alone instead of waiting in hipFree.  NA_BatchIsBroken: 1 / 0. */
NA_EXTERN int NA_BatchSetWaitLimitMs(NA_Batch* batch, double milliseconds);
NA_EXTERN double NA_BatchGetWaitLimitMs(NA_Batch* batch);
NA_EXTERN int NA_BatchIsBroken(NA_Batch* batch);
/* The batch's HIP stream.  Fetching it switches a batch that created its own stream from contract (b) to contract (a) for good: the
   internal launches are joined, and from then on every launch is ordered on this stream. */
NA_EXTERN void* NA_BatchGetHipStream(NA_Batch* batch);
/* Timing marks for benchmarks: HIP events recorded on EVERY stream the batch launches kernels on.  NA_BatchMarkTime(b, 0) ... launches ...
   NA_BatchMarkTime(b, 1); NA_BatchElapsedMs waits for the second mark and returns the longest mark-to-mark span over those streams (< 0: error). */
NA_EXTERN int NA_BatchMarkTime(NA_Batch* batch, int which);
NA_EXTERN int NA_BatchWaitMarks(NA_Batch* batch); /* polls until the second marks are reached on every stream */
NA_EXTERN float NA_BatchElapsedMs(NA_Batch* batch);
/* 1: the last NA_BatchProcessDevice call ran as two half-batch launches (see NA_BatchGetHipStream) */
NA_EXTERN int NA_BatchUsesHalfLaunches(NA_Batch* batch);
/* Opt-in (off unless the environment says NA_RESIDENT=1): device-pointer buffers of a batch on its own stream whose streams are >= 512
   A1 Standard models become commands to one resident launch (contract (b) above).  Measured on MI355X: per-buffer
   latency of a lone buffer 41 - 44 us instead of 43 - 48, sustained throughput 38.5 - 39.7 us per 1024 x 128 step instead of 36.6 -- the
   chip is power-limited on this kernel, so keeping every slot busy buys clock throttling, not throughput.  0 on success. */
NA_EXTERN int NA_BatchSetResidentLaunch(NA_Batch* batch, int on);
/* 1: the last NA_BatchProcessDevice call was a command to the resident launch */
NA_EXTERN int NA_BatchUsesResidentLaunch(NA_Batch* batch);
/* roofline bookkeeping (stream-weighted means): compulsory HBM bytes and multiply-accumulates per sample */
NA_EXTERN double NA_BatchAlgorithmicBytesPerSample(NA_Batch* batch, int blockFrames);
NA_EXTERN double NA_BatchMacsPerSample(NA_Batch* batch);
NA_EXTERN double NA_BatchStateBytes(NA_Batch* batch);
/* > 1 when the stream of a narrow static WaveNet model runs packed with others of its model into one kernel-level stream (0: bad argument) */
NA_EXTERN int NA_BatchStreamPackFactor(NA_Batch* batch, int stream);
/* the kernel that runs the stream (its rocprof name without template arguments; static string, "" on a bad argument) */
NA_EXTERN const char* NA_BatchStreamKernelName(NA_Batch* batch, int stream);
/* ---- multi-GPU host: one batch + one host thread + one HIP stream per device ------------------------------------------------
 * The GLOBAL stream list (order of the NA_MultiAddStreams calls: sort it by architecture) is cut into contiguous ranges of near-equal
 * cost, one per entry of `devices` (an index may repeat).  Streams are independent (the reference runs one NeuralModel per stream,
 * NeuralModel.h:127), so there is no data-path collective: every shard uploads, processes and downloads its own rows of the caller's
 * [streams][n] arrays. */
typedef struct NA_MultiBatch NA_MultiBatch;
NA_EXTERN NA_MultiBatch* NA_MultiCreate(const int* devices, int numDevices);
NA_EXTERN void NA_MultiDestroy(NA_MultiBatch* multi);
NA_EXTERN int NA_MultiAddStreams(NA_MultiBatch* multi, NeuralModel* model, float quality, int count, int doPrewarm); /* first global id */
NA_EXTERN int NA_MultiCommit(NA_MultiBatch* multi);   /* shard + create the device state (implied by the first Process / Submit) */
NA_EXTERN int NA_MultiNumStreams(NA_MultiBatch* multi);
NA_EXTERN int NA_MultiNumShards(NA_MultiBatch* multi);
NA_EXTERN int NA_MultiShardRange(NA_MultiBatch* multi, int shard, int* begin, int* end, int* device);
NA_EXTERN int NA_MultiProcess(NA_MultiBatch* multi, const float* in, float* out, size_t n);   /* host [streams][n]; synchronous */
NA_EXTERN int NA_MultiSubmit(NA_MultiBatch* multi, const float* in, size_t n);                /* pipelined, like NA_BatchSubmit */
NA_EXTERN int NA_MultiCollect(NA_MultiBatch* multi, int ticket, float* out);
NA_EXTERN int NA_MultiSetQuality(NA_MultiBatch* multi, int stream, float quality);
/* Fan-out / fan-in between the devices of a multi batch; call before NA_MultiCommit.  0 (default): host rows -- every shard uploads its
 * weights from the host and downloads its rows into the caller's array, no GPU talks to another.  1: RCCL over xGMI (librccl.so is
 * loaded with dlopen at this point; devices must be distinct) -- a model's weight images are replicated from the first shard that holds
 * it (ncclSend / ncclRecv), NA_MultiProcess gathers every shard's output rows into a [streams][n] device buffer on every GPU (an
 * all-gather of unequal parts) and serves the host array from shard 0 in one download; NA_MultiGatheredOutput(multi, shard) is that
 * buffer on the shard's GPU (valid until the next NA_MultiProcess).  The kernels' data path has no collective either way. */
NA_EXTERN int NA_MultiSetFanIn(NA_MultiBatch* multi, int mode);
NA_EXTERN const float* NA_MultiGatheredOutput(NA_MultiBatch* multi, int shard);
/* 1 when librccl.so loads and exports every entry point this library binds (no GPU needed), else 0 with NA_GetLastError().  The first
 * call is the dlopen (seconds on a cold page cache: measured 4.9 s): a set-up call, never one for the audio thread. */
NA_EXTERN int NA_RcclAvailable(void);
/* the partition itself: bounds[0 .. parts] of contiguous ranges of items [0, n) with near-equal total cost (every range keeps at least
 * one item while items remain).  One-process-per-GPU hosts (bench.py over torch.distributed / RCCL) call it with their rank. */
NA_EXTERN int NA_ShardByCost(const double* cost, int n, int parts, int* bounds);
/* relative cost of one stream of `model` at `quality` (what the sharder balances): estimated microseconds per 1024 streams x 128 frames */
NA_EXTERN double NA_ModelStreamCost(NeuralModel* model, float quality);

/* Host side only (no GPU needed): the kernel family a batch of `streams` streams of `model` would run on -- "f16-split", "frame" (f32),
 * "generic" (wide arrays) or "recurrent" -- with the facts behind the choice: the input limit of the f16-split range proof, whether the
 * proof holds (every value stays inside the f16 range for inputs within the limit, limit >= 8), whether the weights fit the operand
 * format, and the stream-packing factor.  A WaveNet model that fails the proof runs on the f32 frame kernel.  0 on success. */
NA_EXTERN int NA_ModelKernelInfo(NeuralModel* model, float quality, int streams, char* kernelBuf, int bufSize, float* inputLimit, int* rangeProven,
	int* weightsOk, int* packFactor);

/* Range contract of the kernel that runs the stream: input samples beyond +-limit are clamped, NaN reads as silence.  +inf for the f32
 * kernels (they follow the reference's f32 chain at any amplitude); a per-model bound <= 32752 for the f16-split WaveNet kernels, whose
 * values carry an f16 exponent -- far above any audio level (0 on a bad argument). */
NA_EXTERN float NA_BatchStreamInputLimit(NA_Batch* batch, int stream);
/* Models the f16-split kernels run WITHOUT a static range proof (LeakyReLU: the official A2 shapes) saturate a value that leaves the f16
 * range instead of overflowing, and count it: the number of (wave, block) pairs of this stream in which that happened since its last
 * reset / prewarm.  0 = the stream's output is the reference's to the usual tolerance; > 0 = some block was computed with clamped values
 * (the stream recovers one receptive field later).  Always 0 for proven models and the f32 kernels.  Synchronises the batch stream: a
 * diagnostic, not for the audio path.  Negative on a bad argument. */
NA_EXTERN int NA_BatchStreamRangeEvents(NA_Batch* batch, int stream);
#ifndef NA_RELEASE
/* ---- test / tuning hooks: exported by the test build only (csrc/Makefile default target; what tests/ loads).  The release library
 * (make RELEASE=1 -> dist/libNeuralAudioCAPI.so: -DNA_RELEASE -DNA_NO_TUNING, no loopback RCCL table) has none of the NA_Debug* symbols
 * and reads no tuning environment variable. ---- */
/* NAMIsA2 (bit 0) / NAMIsA2Standard (bit 1) of a .nam document (NeuralModel.cpp:159-168, 188-317); negative on a parse error */
NA_EXTERN int NA_DebugClassifyNam(const char* jsonText);
/* stream packing, host side only: pack factor of the model in a large batch (1: none); flat weights of the packed virtual model into
 * out[capacity] when given; returns their count (0: model does not pack), -1 on failure */
NA_EXTERN int NA_DebugPackedWeights(NeuralModel* model, int* packFactor, float* out, int capacity);
/* tests / tuning: 0 = WaveNet models with a compile-time specialised layer chain run on the stage interpreter instead (same stream state,
 * bit-identical results); process-wide, set it only while no other thread is processing */
NA_EXTERN void NA_DebugSetWaveNetSpec(int on);
/* Tests / tuning: the stream count of one recurrent launch from which the four-streams-per-wave kernel is used (0: never; default 3072,
 * environment NA_REC_QUAD_MIN); returns the previous value.  NA_DebugRecurrentQuadLaunches: launches of that kernel so far. */
NA_EXTERN int NA_DebugSetRecurrentQuadMin(int streams);
NA_EXTERN long long NA_DebugRecurrentQuadLaunches(void);
/* Tests: which implementation of the NCCL entry points the multi-GPU host binds.  0 = librccl.so (the product).  1 = a loopback table
 * inside this library (csrc/rccl_loopback.cpp): every rank may sit on the SAME device and a transfer is a device-to-device copy, so the
 * multi-rank orchestration (communicators, weight fan-out, gathered fan-in, failure teardown) executes on a one-GPU box; it moves no
 * byte over xGMI.  failSendAt > 0 makes the failSendAt-th ncclSend of the loopback table fail (fault injection), rendezvousMs > 0 is how
 * long a loopback rank waits for a peer that never posts.  Set it while no multi batch is being committed. */
NA_EXTERN void NA_DebugSetRcclApi(int mode, int failSendAt, int rendezvousMs);
/* Tests: a kernel that keeps the batch's streams busy for `milliseconds` (at most 10 000) behind whatever they hold -- a device that
 * does not answer, as far as the waits of this batch can tell (tests/test_gpu_stall.py drives the wait limit with it). */
NA_EXTERN int NA_DebugStallDevice(NA_Batch* batch, double milliseconds);
/* tuning aid: device buffer (long long[stages*4*waves]) that workgroup 0 of the WaveNet kernel stamps with the shader clock; NULL = off */
NA_EXTERN void NA_DebugSetTraceBuffer(void* deviceBuffer);
#endif /* NA_RELEASE */

#ifdef __cplusplus
}
#endif
#endif
