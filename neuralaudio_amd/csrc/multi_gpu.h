// multi_gpu.h -- the C++ multi-GPU host of the batched engine: one na::GpuBatch + one host thread per device (one HIP stream each),
// the global stream list sharded across them by cost.  north_star: "C++ host code ... sharding stream batches across the 8 GPUs of
// one node"; SURVEY.md 7 step 6: one host thread + HIP stream per device.
//
// The reference has no counterpart: it runs one NeuralModel per audio stream on the caller's thread (NeuralAudio/NeuralModel.h:127).
// Streams are independent, so the multi-GPU path is a pure partition of the (architecture-sorted) global stream list into contiguous
// ranges of near-equal cost; state never moves between devices and there is NO data-path collective.  The fan-in is the host's
// [streams][n] output array itself: every shard's download lands in its own rows.
#pragma once

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gpu_batch.h"
#include "rccl_dyn.h"

namespace na
{
	// Contiguous partition of items [0, n) into `parts` ranges of near-equal total cost: bounds[p] .. bounds[p + 1] is range p
	// (bounds has parts + 1 entries, bounds[0] = 0, bounds[parts] = n).  A range is closed as soon as adding the next item would
	// carry the running total past p-th share of the whole; every later range keeps at least one item while items remain.
	std::vector<int> ShardByCost(const double* cost, int n, int parts);

	class MultiGpuBatch
	{
	public:
		// `devices`: HIP device index per shard (an index may repeat: several shards on one GPU -- tests, or over-subscription)
		explicit MultiGpuBatch(const std::vector<int>& devices);
		~MultiGpuBatch();

		MultiGpuBatch(const MultiGpuBatch&) = delete;
		MultiGpuBatch& operator=(const MultiGpuBatch&) = delete;

		// Appends `count` streams to the GLOBAL list (ids are consecutive); nothing is created on a device before Commit()
		int AddStreams(const std::shared_ptr<const LoadedModel>& model, float quality, int count, bool prewarm, bool onDemand);
		// Shards the global list by cost, creates the per-device batches (each on its own worker thread) and adds every shard's streams
		void Commit();
		bool Committed() const { return committed; }

		// Fan-out / fan-in between the devices.  HostRows (default): every shard uploads its weights from the host and downloads its
		// rows into the caller's [streams][n] array -- no device talks to another.  Rccl (librccl.so bound at run time, rccl_dyn.h; one
		// rank per shard, so the devices must be distinct): at Commit() a model's weight images are replicated from the first shard that
		// holds it to the others over xGMI (ncclSend / ncclRecv inside one group), and Process() gathers the shards' output rows into a
		// [streams][n] DEVICE buffer on every GPU (one ncclBroadcast per shard inside a group = an all-gather of unequal parts) from
		// which shard 0 serves the host array in ONE download; GatheredOutput(shard) is the buffer for device-side consumers.  Set
		// before Commit().  The data path between the kernels needs no collective either way.
		enum class FanIn { HostRows, Rccl };
		void SetFanIn(FanIn mode);
		FanIn GetFanIn() const { return fanIn; }
		const float* GatheredOutput(int shard) const; // Rccl mode: device pointer to [streams][n] of the last Process() on that shard's GPU

		int NumStreams() const { return total; }
		int NumShards() const { return (int)shards.size(); }
		void ShardRange(int shard, int& begin, int& end, int& device) const;

		// host arrays [streams][n]: every worker runs its shard's rows through GpuBatch::ProcessHost; returns when all are done
		void Process(const float* in, float* out, size_t n);
		// pipelined: GpuBatch::Submit / Collect on every shard (up to GpuBatch::kPipelineSlots buffers in flight)
		int Submit(const float* in, size_t n);
		void Collect(int ticket, float* out);
		void SetQuality(int stream, float quality);

	private:
		struct Entry
		{
			std::shared_ptr<const LoadedModel> model;
			float quality;
			int count;
			bool prewarm, onDemand;
		};
		struct Shard
		{
			int device = 0, begin = 0, end = 0;
			std::unique_ptr<GpuBatch> batch;
			std::thread worker;
			std::mutex m;
			std::condition_variable cv;
			std::function<void()> job; // the pending command (one at a time per shard)
			bool busy = false, quit = false;
			std::string error;
			// Rccl mode
			int rank = 0;
			rccl::Comm comm = nullptr;
			float* gathered = nullptr; // device [total][n] (owned by the worker: allocated / freed on its thread)
			size_t gatheredFloats = 0;
		};
		void Run(Shard& s);
		void CheckShard(const Shard& s) const;
		void CheckUsable() const;
		void StopWorkers(); // joins the workers (each destroys its batch on its own thread) and drops the shards
		void Post(const std::function<void(Shard&)>& f); // the same command on every shard's thread, waits for all, rethrows the first error

		std::vector<int> devices;
		std::vector<Entry> entries;
		std::vector<std::unique_ptr<Shard>> shards;
		int total = 0;
		bool committed = false;
		std::string broken; // set by a submission that only part of the shards took
		FanIn fanIn = FanIn::HostRows;
		const rccl::Api* nccl = nullptr;
		void InitRccl();          // communicators (one per shard) + weight replication
		void ReplicateWeights();
		std::vector<int> HoldersOf(const LoadedModel* model) const;
		int FirstHolderOf(const LoadedModel* model) const;
		void ProcessGathered(const float* in, float* out, size_t n);
	};
}
