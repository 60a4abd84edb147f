// gpu_batch.h -- the batched multi-stream engine: owns device state for many independent audio
// streams on ONE GPU and runs the WaveNet / LSTM kernels over them, one launch per model group
// and block of <= 128 frames.
//
// The reference has no counterpart (it processes one stream per NeuralModel instance on the caller's
// thread, NeuralAudio/NeuralModel.h:127); this is the data-parallel axis the GPU path adds.  A
// single-stream NeuralModel (neural_model.cpp) is a GpuBatch with one stream.
#pragma once

#include <chrono>
#include <cmath>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "model_loader.h"
#include "tuning.h"
#include "wavenet_launch.h"

namespace na
{
	// Host blocks the caller registered (NA_RegisterHostBuffer: hipHostRegister, mapped): ProcessHost runs the kernels directly on them --
	// no staging copies.  RegisteredDevicePointer: the device address of `p` when [p, p + bytes) lies inside a registered block of
	// `device`'s process, else nullptr.
	bool RegisterHostBuffer(void* p, size_t bytes, std::string& error);
	bool UnregisterHostBuffer(void* p);
	void* RegisteredDevicePointer(const void* p, size_t bytes);

	class HipError : public std::runtime_error
	{
	public:
		HipError(hipError_t e, const char* what) : std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)), code(e) {}
		hipError_t code;
	};

	void CheckHip(hipError_t e, const char* what);

	// returns the number of visible HIP devices (0 when there is no GPU / no driver); never throws
	int VisibleDeviceCount();

	// Relative cost of one stream of a model (what the multi-GPU sharder balances, multi_gpu.h): microseconds per 1024 streams x 128
	// frames of the kernel family that runs it, fitted to the round-3 measurements; host-side only (no device needed)
	double EstimateStreamCost(const LoadedModel& model, float quality);

	// Host-side prediction of the kernel family for a model (no device needed): "f16-split" | "frame" | "generic" | "recurrent", the
	// pack factor, and the static range proof of the f16-split kernels (wavenet_plan.h: splitRangeProven / splitWeightsOk / condLimit)
	struct ModelKernelInfo
	{
		const char* kernel = "";
		int pack = 1;
		float inputLimit = INFINITY;
		bool rangeProven = true, weightsOk = true;
	};
	ModelKernelInfo PredictModelKernel(const LoadedModel& model, float quality, int streams);

	class ModelGroup; // one per distinct ModelDesc: packed weights + state of all its streams

	class GpuBatch
	{
	public:
		// throws std::runtime_error if `device` is not a usable HIP device -- there is no CPU fallback
		// `borrowedStream` != nullptr: launch on the caller's HIP stream instead of creating one
		explicit GpuBatch(int device, hipStream_t borrowedStream = nullptr);
		~GpuBatch();

		GpuBatch(const GpuBatch&) = delete;
		GpuBatch& operator=(const GpuBatch&) = delete;

		// Adds a stream running `model`; its row in the [streams][n] input/output arrays is the returned id.
		// For a SlimmableContainer every submodel gets state, `quality` picks the active one.
		// onDemand = ECompositeModelLoadMode::OnDemand (CompositeModel.h:104-109): only the active submodel is prewarmed now, the
		// others when a quality change first selects them.
		int AddStream(const std::shared_ptr<const LoadedModel>& model, float quality, bool prewarm, bool onDemand = false);
		// `count` streams of the same model at once (one state reset / prewarm launch per model group); returns the first id
		int AddStreams(const std::shared_ptr<const LoadedModel>& model, float quality, int count, bool prewarm, bool onDemand = false);

		// Stream lifecycle (the reference's unit of lifetime is the model instance: NeuralAudioCApi.cpp:38-42 DeleteModel, CompositeModel.h:17-23):
		// RemoveStreams frees the state slots of streams [first, first + count) in every model group (also inside packed virtual streams)
		// and retires their ids; the rows keep their place in the [streams][n] arrays (input ignored, host-buffer output zero).  A later
		// AddStreams recycles retired ids -- the lowest retired id for count == 1, a run of `count` consecutive retired ids if there is
		// one -- before it appends new rows, and recycled state slots start from the same fresh / prewarmed state as new ones.
		// Not real-time safe (like AddStreams): call between buffers.
		void RemoveStreams(int first, int count);
		bool IsLive(int stream) const { return stream >= 0 && stream < (int)streams.size() && streams[(size_t)stream].live; }

		int NumStreams() const { return (int)streams.size(); } // rows of the [streams][n] arrays (retired ids included)
		int NumLiveStreams() const { return (int)streams.size() - (int)retired.size(); }
		unsigned StreamPrewarmedMask(int stream) const; // bit k: submodel k of the stream had its prewarm

		// ScalableCompositeModel::SetQualityScaleFactor (CompositeModel.h:176-181): switches the active submodel,
		// the inactive one keeps its state untouched.
		void SetQuality(int stream, float quality);
		// CompositeModel::IsModelChangeRealtimeSafe (CompositeModel.h:44-50) for this engine: false when the switch would prewarm a
		// submodel (OnDemand) or force a hipGraph re-capture; true when it only re-uploads the pinned index lists asynchronously
		bool IsQualityChangeRealtimeSafe(int stream, float quality) const;
		float GetQuality(int stream) const;
		int GetActiveSubModel(int stream) const;

		// Re-establish the zero-input steady state of every submodel of `stream` (NeuralModel::Prewarm)
		void Prewarm(int stream);

		// in/out are DEVICE pointers, row s = stream s, `n` samples per row, rows `stride` floats apart.
		// Asynchronous on GetStream(). in == out is allowed.
		void ProcessDevice(const float* dIn, float* dOut, size_t n, long inStride, long outStride);

		// in/out are HOST pointers laid out [streams][n]; stages through pinned buffers; synchronous.
		void ProcessHost(const float* in, float* out, size_t n);
		// `in`: HOST rows [streams][n] (staged through the pinned block); `dOut`: DEVICE rows, `outStride` floats apart.  Asynchronous on
		// GetStream() once the input is staged -- the multi-GPU host's RCCL fan-in gathers the shards' device rows (multi_gpu.cpp).
		void ProcessHostToDevice(const float* in, float* dOut, size_t n, long outStride);
		// device copies of a model's weight tables (every group of the batch that runs one of `model`'s submodels, in submodel order):
		// what the multi-GPU host replicates from the first device that holds the model (RCCL fan-out)
		void WeightImages(const LoadedModel& model, std::vector<std::pair<void*, size_t>>& out) const;
		// Weight fan-out (multi-GPU host): while SetPeerWeights(true), a model group created by AddStreams allocates its weight images
		// without uploading them -- a peer device that holds the model sends them -- and the prewarm of its streams waits; WeightsArrived()
		// says the images are in place (on this batch's stream): device-side derivations and the deferred prewarms run, the mode ends.
		// Processing before WeightsArrived() is an error of the caller.
		void SetPeerWeights(bool on) { peerWeights = on; }
		bool AwaitsWeights() const { return !awaitingWeights.empty(); }
		void WeightsArrived();

		// Pipelined host-buffer interface: Submit() copies `in` ([streams][n], host) into a pinned slot and enqueues H2D (copy-in
		// stream), the kernels (batch stream) and D2H (copy-out stream); Collect() waits for that slot and copies the result out.
		// With two slots in flight the upload of buffer k+1 and the download of buffer k-1 overlap the kernels of buffer k.
		// Buffers are processed strictly in submission order (the streams' state depends on it).
		static constexpr int kPipelineSlots = 3;
		int Submit(const float* in, size_t n);       // returns a ticket; throws if all slots are in flight
		void Collect(int ticket, float* out);        // blocks until that buffer is done
		// Zero-copy variants: the caller writes the next buffer straight into the pinned staging memory the upload reads
		// (NextInput, then Submit(nullptr, n)) and reads results in place (Collect(ticket, nullptr), then OutputView: valid until
		// that slot is submitted again, kPipelineSlots submissions later).
		float* NextInput(size_t n);
		const float* OutputView(int ticket) const;
		size_t SlotFrames(int ticket) const { return (ticket >= 0 && ticket < kPipelineSlots) ? pipe[ticket].n : 0; } // frames per row of that submission
		size_t SlotRows(int ticket) const { return (ticket >= 0 && ticket < kPipelineSlots) ? pipe[ticket].rows : 0; } // rows of that submission

		// ---- bounded waits ------------------------------------------------------------------------------------------------------------
		// Process is called from a real-time thread that must get its call back (NeuralAudio/NeuralModel.h:127, README "one thread per
		// model"): every host-side wait of the processing paths -- stream and event waits, the polls of the resident launch's counters,
		// the timing marks -- has a wall-clock limit (default 2000 ms: NA_WAIT_LIMIT_MS; <= 0: none).  A wait that runs into it marks the
		// batch BROKEN and throws: the call returns (the C ABI: non-zero, NA_GetLastError(), output rows zeroed), and every later call on
		// the batch fails at once without touching the device -- the stream states are no longer what the caller thinks they are.  A
		// broken batch can only be destroyed; its destructor gives the device one more limit to come back and otherwise leaves the
		// device allocations alone (a kernel may still be writing them) instead of hanging in hipFree.
		void SetWaitLimitMs(double ms) { waitLimitMs = ms; }
		double GetWaitLimitMs() const { return waitLimitMs; }
		bool IsBroken() const { return broken; }
		const std::string& BrokenReason() const { return brokenWhy; }
		void WaitStreamBounded(hipStream_t s, const char* what); // hipStreamSynchronize with the limit
		void WaitEventBounded(hipEvent_t e, const char* what);   // hipEventSynchronize with the limit
		// test hook (NA_DebugStallDevice): a kernel that keeps the batch stream busy for `ms` milliseconds (at most 10 s), behind whatever
		// the stream holds -- a wedged device as far as every wait on this batch can tell
		void DebugStallDevice(double ms);

		void Synchronize();
		// every buffer handed to ProcessDevice so far has been processed (host-side wait).  Unlike Synchronize() it leaves the resident
		// launch on the chip; on a caller's / observed stream it is a synchronisation of that stream.
		void WaitOutputs();
		// Timing marks for bench.py (HIP events on EVERY stream this batch launches kernels on -- the batch stream and the two half-batch
		// streams): MarkTime(0) ... launches ... MarkTime(1); ElapsedMs() = the longest mark-to-mark span over those streams (after a
		// Synchronize()).  The caller's own events only see the stream it handed in.
		bool UsesHalfLaunches() const { return lastStepHalves; }
		// Opt-in (default: Tuning residentOn = NA_RESIDENT=1): large A1 Standard batches on the batch's own stream run as commands to ONE
		// launch that stays on the chip (gpu_batch_chains.cpp) instead of two free-running half-batch launches per buffer.
		void SetResidentLaunch(bool on);
		bool UsesResidentLaunch() const { return lastStepResident; } // the last device-pointer buffer went through the resident launch
		void MarkTime(int which);
		void WaitMarks(); // polls until the marks of MarkTime(1) are reached on every stream
		float ElapsedMs();
		// The batch's HIP stream.  Handing it out makes the batch order every launch on it from then on: until then a batch that created
		// its own stream may run a buffer as two free-running half-batch launches on internal streams (see halfStream below) -- nobody
		// outside can observe the order of work on a stream they never saw; NA_BatchSynchronize / the host-buffer entry points wait for all.
		hipStream_t GetStream();
		int GetDevice() const { return device; }

		// roofline bookkeeping for bench.py: stream-weighted algorithmic bytes / MACs per sample
		double AlgorithmicBytesPerSample(int blockFrames) const;
		double MacsPerSample() const;
		size_t StateBytes() const;
		const char* StreamKernelName(int stream) const; // which kernel runs the stream (rocprof name without template arguments)
		// range contract: samples beyond +-limit are clamped (NaN reads as silence) by the kernel that runs the stream; +inf for the f32 kernels
		float StreamInputLimit(int stream) const;
		// f16-split kernels on a model without a static range proof (the official A2 shapes): how often a value of the stream left the f16
		// range and was saturated -- (wave, block) pairs since the stream's last reset / prewarm; 0 means the output is the reference's to
		// the usual tolerance.  Synchronises the batch stream (a diagnostic, not for the audio path).
		int StreamRangeEvents(int stream);
		int StreamPackFactor(int stream) const; // > 1: the stream shares a kernel-level stream with others of its model (stream packing)

	private:
		struct StreamRef
		{
			std::shared_ptr<const LoadedModel> model;
			std::vector<std::pair<ModelGroup*, int>> members; // per submodel: (group, member index)
			int active = 0;
			float quality = 1.0f;
			bool onDemand = false;
			bool live = true;
			std::vector<char> prewarmed; // per submodel: had its initial prewarm
		};
		std::vector<int> retired; // sorted ids of removed streams
		int AllocateIds(int count);
		void DropTrailingRetiredRows();
		int LaunchUnitsAfterSwitch(const ModelGroup* leaving, const ModelGroup* entering) const;
		void ZeroRetiredRows(float* hostRows, size_t n, size_t rows) const;

		ModelGroup* GroupFor(const std::shared_ptr<const ModelDesc>& desc, int packHint = 0);
		void EnsureStaging(size_t floats);

		int device;
		hipStream_t stream = nullptr;
		bool ownsStream = true;
		double waitLimitMs = DefaultWaitLimitMs();
		bool broken = false;
		std::string brokenWhy;
		static double DefaultWaitLimitMs();
		void CheckUsable() const; // throws when the batch is broken
		[[noreturn]] void Stall(const char* what);
		// a deadline of the current wait limit; Expired() is cheap enough for every turn of a poll loop
		struct Deadline
		{
			std::chrono::steady_clock::time_point end;
			bool limited;
			explicit Deadline(double ms) : end(std::chrono::steady_clock::now() + std::chrono::microseconds((long long)(ms > 0 ? ms * 1000.0 : 0))), limited(ms > 0) {}
			bool Expired() const { return limited && std::chrono::steady_clock::now() > end; }
		};
		hipEvent_t forkEvent = nullptr;

		// cached hipGraph of the multi-group fork/launch/join sequence (see ProcessDevice)
		struct GraphKey
		{
			const float* dIn;
			float* dOut;
			size_t n;
			long inStride, outStride;
			unsigned long version;
		};
		struct GraphEntry
		{
			GraphKey key;
			hipGraphExec_t exec;
		};
		std::vector<GraphEntry> graphCache; // a handful of call signatures (hosts cycle through a few fixed buffers)
		unsigned long topologyVersion = 0;
		std::vector<std::unique_ptr<ModelGroup>> groups;
		std::vector<StreamRef> streams;
		bool peerWeights = false;
		std::vector<ModelGroup*> awaitingWeights;
		std::vector<std::pair<ModelGroup*, std::vector<int>>> pendingPrewarm;

		float* hostStage = nullptr; // pinned
		float* devStage = nullptr;
		size_t stageFloats = 0;

		static constexpr int kMaxChains = 4;
		struct PipeSlot
		{
			float* hostIn = nullptr;  // pinned
			float* hostOut = nullptr; // pinned
			float* dev = nullptr;
			size_t floats = 0, n = 0;
			size_t rows = 0; // rows of the [streams][n] block this slot was submitted with
			hipEvent_t uploaded = nullptr, computed = nullptr, downloaded = nullptr;
			hipStream_t own = nullptr; // upload, kernel and download of this slot's buffer, in order (batches that run as one launch)
			bool onOwnStream = false;
			hipEvent_t halfDone[kMaxChains] = {}; // free-running half-batch chains (Submit): this buffer's kernel of each chain
			bool onHalfStreams = false;
			bool busy = false;
		};
		void DrainPipeline();
		// Two free-running chains: a batch whose active streams all run on the f16-split WaveNet kernels as ONE launch per buffer (one
		// model, or a mixed / packed batch that shares a fused launch; >= 512 kernel-level streams) can run a buffer as two launches of half
		// of every group's streams, each half on its own HIP stream, in submission order -- the halves never wait for each other (streams
		// are independent), so the tail of one launch overlaps the other half's work: 40.1 -> 37.4 us per 1024 x 128 buffer of kernel
		// time (tools/split_launch_probe2.py).  A call ordered on ONE stream cannot do this (it would have to join the halves every
		// call: 65.9 us); Submit / Collect can (Collect waits for both halves of its ticket), and so can NA_BatchProcessDevice on a batch
		// whose stream nobody else has seen.
		hipStream_t halfStream[kMaxChains] = {};
		int numChains = 2; // (tuning knob NA_HOST_CHAINS: 2 .. kMaxChains parts instead of halves)
		bool markOpen = false; // between MarkTime(0) and MarkTime(1): a chain stream created now gets its start mark right away
		bool halfChainsUsed = false;
		bool lastStepHalves = false; // the last device-pointer / submitted buffer ran as two half-batch launches
		bool streamObserved = false; // GetStream() was called (or the stream is the caller's): launches are ordered on `stream`
		// The resident launch (gpu_batch_chains.cpp): one launch that stays on the chip and walks consecutive buffers, fed through a command
		// ring in pinned host memory -- for the batches the chains serve whose buffer is one launch of 128-frame A1 Standard blocks
		struct ResidentState;
		std::unique_ptr<ResidentState> residentState;
		WnLaunchTable wnTable[4]; // device tables of launch lists with more groups than a launch's kernarg segment holds (gpu_batch.cpp launchWnList; [3]: the recurrent list)
		bool lastStepResident = false;
		bool residentWanted = Tuning::Get().residentOn;
		bool TryResident(const float* dIn, float* dOut, size_t n, long inStride, long outStride);
		bool ResidentConfigure();
		void ResidentEnsureRunning();
		void ResidentExitAfterPosted();
		void ResidentFinishMarked();
		void DrainResident(); // every posted command has run, the launch is gone (called by everything that touches state or the batch stream)
		// nothing of this batch is in flight anywhere: resident launch, half-batch chains, pipeline slots, the batch stream
		void Quiesce();
		struct HalfLists; // the two launch lists of the buffer (gpu_batch.cpp)
		std::unique_ptr<HalfLists> halfLists;
		bool PrepareHalves(size_t n);
		void ProcessDeviceOrdered(const float* dIn, float* dOut, size_t n, long inStride, long outStride);
		void BeginHalves();
		void LaunchChain(int h, const float* dIn, float* dOut, size_t n, long inStride, long outStride, bool hostRows);
		void LaunchHalves(const float* dIn, float* dOut, size_t n, long inStride, long outStride, hipEvent_t* done, bool hostRows);
		void JoinHalves(); // the half-batch chains are done (host-side wait); the next launches go to the batch stream again
		hipEvent_t marks[1 + kMaxChains][2] = {};
		void ProcessDeviceOn(hipStream_t launch, const float* dIn, float* dOut, size_t n, long inStride, long outStride);
		// ordering between the batch stream and the slot streams: the stream state makes every kernel launch depend on the previous one
		bool pipelineUsed = false;
		hipEvent_t lastKernelEvent = nullptr, mainDone = nullptr;
		hipStream_t lastKernelStream = nullptr;
		unsigned long submitTopology = 0;
		PipeSlot pipe[kPipelineSlots];
		hipStream_t copyIn = nullptr, copyOut = nullptr;
		int nextSlot = 0;
		void EnsurePipeSlot(PipeSlot& s, size_t floats);
	};
}
