// multi_gpu.cpp -- see multi_gpu.h.  Host side only: sharding, worker threads, fan-in through the caller's arrays.
#include "multi_gpu.h"

#include <algorithm>
#include <stdexcept>

namespace na
{
	std::vector<int> ShardByCost(const double* cost, int n, int parts)
	{
		if (parts < 1) throw std::runtime_error("neuralaudio_amd: ShardByCost with parts < 1");
		if (n < 0) throw std::runtime_error("neuralaudio_amd: ShardByCost with a negative item count");
		double total = 0.0;
		for (int i = 0; i < n; i++) total += cost[i];
		std::vector<int> bounds((size_t)parts + 1, n);
		bounds[0] = 0;
		int begin = 0;
		double acc = 0.0; // cost of items [0, begin)
		for (int p = 0; p + 1 < parts; p++)
		{
			const double target = total * (double)(p + 1) / (double)parts;
			int end = begin;
			while (end < n && acc + cost[end] <= target + 1e-9)
			{
				acc += cost[end];
				end++;
			}
			// at least one item per range while items remain: take one even if it alone overshoots the share ...
			const int cap = std::max(begin, n - (parts - p - 1));
			if (end == begin && end < cap) end++;
			// ... and leave one for each remaining range
			end = std::min(end, cap);
			acc = 0.0; // (re-summed in order: the running total never depends on a subtraction)
			for (int i = 0; i < end; i++) acc += cost[i];
			bounds[(size_t)p + 1] = end;
			begin = end;
		}
		return bounds;
	}

	MultiGpuBatch::MultiGpuBatch(const std::vector<int>& devs) : devices(devs)
	{
		if (devices.empty()) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch needs at least one device");
		const int visible = VisibleDeviceCount();
		if (visible <= 0) throw std::runtime_error("neuralaudio_amd: no HIP device is visible; this library has no CPU fallback");
		for (int d : devices)
			if (d < 0 || d >= visible) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch: invalid HIP device index");
	}

	MultiGpuBatch::~MultiGpuBatch() { StopWorkers(); }

	int MultiGpuBatch::AddStreams(const std::shared_ptr<const LoadedModel>& model, float quality, int count, bool prewarm, bool onDemand)
	{
		if (committed) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch::AddStreams after Commit (the shards are fixed)");
		if (!model || count < 1) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch::AddStreams: bad argument");
		const int first = total;
		entries.push_back({ model, quality, count, prewarm, onDemand });
		total += count;
		return first;
	}

	// the worker of one shard: owns the shard's GpuBatch (created, used and destroyed on this thread) and runs one command at a time
	void MultiGpuBatch::Run(Shard& s)
	{
		std::unique_lock<std::mutex> lock(s.m);
		for (;;)
		{
			s.cv.wait(lock, [&] { return s.quit || s.busy; });
			if (s.busy)
			{
				std::function<void()> job = std::move(s.job);
				lock.unlock();
				std::string err;
				try
				{
					job();
				}
				catch (const std::exception& e)
				{
					err = e.what();
				}
				catch (...)
				{
					err = "unknown exception";
				}
				lock.lock();
				s.error = err;
				s.busy = false;
				s.cv.notify_all();
				continue;
			}
			if (s.quit) break;
		}
		lock.unlock();
		if (s.gathered)
		{
			(void)hipSetDevice(s.device);
			(void)hipFree(s.gathered);
			s.gathered = nullptr;
		}
		if (s.comm && nccl) (void)nccl->CommDestroy(s.comm);
		s.comm = nullptr;
		s.batch.reset(); // on the thread that used it
	}

	void MultiGpuBatch::Post(const std::function<void(Shard&)>& f)
	{
		for (auto& sp : shards)
		{
			Shard& s = *sp;
			{
				std::lock_guard<std::mutex> lock(s.m);
				s.job = [&f, &s] { f(s); };
				s.busy = true;
			}
			s.cv.notify_all();
		}
		std::string first;
		for (auto& sp : shards)
		{
			Shard& s = *sp;
			std::unique_lock<std::mutex> lock(s.m);
			s.cv.wait(lock, [&] { return !s.busy; });
			if (first.empty() && !s.error.empty()) first = s.error;
			s.error.clear();
		}
		if (!first.empty()) throw std::runtime_error(first);
	}

	void MultiGpuBatch::Commit()
	{
		if (committed) return;
		if (total < 1) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch::Commit on an empty stream list");
		std::vector<double> cost;
		cost.reserve((size_t)total);
		for (const Entry& e : entries)
		{
			const double c = EstimateStreamCost(*e.model, e.quality);
			cost.insert(cost.end(), (size_t)e.count, c);
		}
		const std::vector<int> bounds = ShardByCost(cost.data(), total, (int)devices.size());
		for (size_t p = 0; p < devices.size(); p++)
		{
			std::unique_ptr<Shard> s(new Shard());
			s->device = devices[p];
			s->begin = bounds[p];
			s->end = bounds[p + 1];
			shards.push_back(std::move(s));
		}
		for (auto& sp : shards)
		{
			Shard* s = sp.get();
			s->worker = std::thread([this, s] { Run(*s); });
		}
		// every worker builds its own batch: the streams of the global list that fall into its range, entry by entry.  A shard that fails
		// (bad device, out of memory) fails the whole Commit: the workers are joined, the shards dropped, and the object is back in its
		// pre-Commit state -- `committed` is set only when every non-empty range has a batch with exactly its rows.
		try
		{
			// RCCL fan-out: only the FIRST shard that runs a model uploads its weight images from the host; the others size theirs and
			// receive the content over the communicator (ReplicateWeights), their streams' prewarm waits for it
			const bool fanOut = fanIn == FanIn::Rccl;
			for (size_t i = 0; i < shards.size(); i++) shards[i]->rank = (int)i;
			Post([this, fanOut](Shard& s) {
				if (s.end <= s.begin) return;
				s.batch.reset(new GpuBatch(s.device));
				int first = 0;
				for (const Entry& e : entries)
				{
					const int a = std::max(first, s.begin), b = std::min(first + e.count, s.end);
					if (b > a)
					{
						s.batch->SetPeerWeights(fanOut && FirstHolderOf(e.model.get()) != s.rank);
						s.batch->AddStreams(e.model, e.quality, b - a, e.prewarm, e.onDemand);
					}
					first += e.count;
				}
				s.batch->SetPeerWeights(false);
			});
			for (auto& sp : shards) CheckShard(*sp);
			if (fanOut) InitRccl();
		}
		catch (...)
		{
			StopWorkers();
			throw;
		}
		committed = true;
	}

	void MultiGpuBatch::SetFanIn(FanIn mode)
	{
		if (committed) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch::SetFanIn after Commit");
		fanIn = mode;
	}

	static void CheckNccl(const rccl::Api* api, rccl::Result r, const char* what)
	{
		if (r != 0) throw std::runtime_error(std::string("neuralaudio_amd: ") + what + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error"));
	}

	// One RCCL rank per shard (ncclCommInitAll: all communicators of this process at once), then the weight fan-out
	void MultiGpuBatch::InitRccl()
	{
		std::string error;
		nccl = rccl::Active(error);
		if (!nccl) throw std::runtime_error(error);
		if (!rccl::OverrideActive()) // (the loopback table of the one-GPU tests takes any device list)
			for (size_t i = 0; i < devices.size(); i++)
				for (size_t k = i + 1; k < devices.size(); k++)
					if (devices[i] == devices[k]) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch: RCCL fan-in needs one distinct device per shard");
		for (auto& sp : shards)
			if (sp->end <= sp->begin) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch: RCCL fan-in needs at least one stream per shard");
		std::vector<rccl::Comm> comms(devices.size(), nullptr);
		CheckNccl(nccl, nccl->CommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
		for (size_t i = 0; i < shards.size(); i++)
		{
			shards[i]->rank = (int)i;
			shards[i]->comm = comms[i];
		}
		ReplicateWeights();
	}

	// shards whose stream range meets an entry of `model`, ascending (the first one is the model's weight source)
	std::vector<int> MultiGpuBatch::HoldersOf(const LoadedModel* model) const
	{
		std::vector<int> h;
		int first = 0;
		for (const Entry& e : entries)
		{
			if (e.model.get() == model)
				for (size_t s = 0; s < shards.size(); s++)
					if (std::max(first, shards[s]->begin) < std::min(first + e.count, shards[s]->end)) h.push_back((int)s);
			first += e.count;
		}
		std::sort(h.begin(), h.end());
		h.erase(std::unique(h.begin(), h.end()), h.end());
		return h;
	}

	int MultiGpuBatch::FirstHolderOf(const LoadedModel* model) const
	{
		const std::vector<int> h = HoldersOf(model);
		return h.empty() ? -1 : h.front();
	}

	// Weight fan-out: for every model that more than one shard runs, the first of them (the only one that uploaded the model's weight
	// tables from the host, Commit) sends its device copies to the others (ncclSend / ncclRecv, all transfers of a rank inside one
	// group); the receivers then derive what their constructors left open and run their deferred prewarms (GpuBatch::WeightsArrived).
	// Everything that can fail is done BEFORE a rank opens its group, the ranks' image lists are compared on the host first (a rank
	// that posted a different count or size would leave its peer waiting), and a group that was opened is always closed: a failure
	// inside it is reported, never left as peers blocked in their transfers.
	void MultiGpuBatch::ReplicateWeights()
	{
		std::vector<const LoadedModel*> models;
		for (const Entry& e : entries)
			if (std::find(models.begin(), models.end(), e.model.get()) == models.end()) models.push_back(e.model.get());
		struct Plan
		{
			std::vector<int> holders;
			std::vector<std::vector<std::pair<void*, size_t>>> images; // per holder (same order)
		};
		std::vector<Plan> plans(models.size());
		for (size_t m = 0; m < models.size(); m++)
		{
			plans[m].holders = HoldersOf(models[m]);
			plans[m].images.resize(plans[m].holders.size());
		}
		// 1. every holder lists its images (on its own thread: the batch is the worker's)
		Post([&](Shard& s) {
			for (size_t m = 0; m < models.size(); m++)
				for (size_t k = 0; k < plans[m].holders.size(); k++)
					if (plans[m].holders[k] == s.rank) s.batch->WeightImages(*models[m], plans[m].images[k]);
		});
		// 2. cross-rank agreement: same number of images, same sizes
		for (size_t m = 0; m < models.size(); m++)
			for (size_t k = 1; k < plans[m].holders.size(); k++)
			{
				const auto &a = plans[m].images[0], &b = plans[m].images[k];
				bool same = a.size() == b.size();
				for (size_t i = 0; same && i < a.size(); i++) same = a[i].second == b[i].second;
				if (!same) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch: the shards disagree on a model's weight images (fan-out refused)");
			}
		// 3. the transfers
		Post([&](Shard& s) {
			CheckHip(hipSetDevice(s.device), "hipSetDevice");
			hipStream_t st = s.batch->GetStream();
			CheckNccl(nccl, nccl->GroupStart(), "ncclGroupStart");
			std::string failed;
			try
			{
				for (size_t m = 0; m < models.size(); m++)
				{
					const Plan& p = plans[m];
					if (p.holders.size() < 2) continue;
					const size_t me = (size_t)(std::find(p.holders.begin(), p.holders.end(), s.rank) - p.holders.begin());
					if (me == p.holders.size()) continue;
					for (const auto& img : p.images[me])
					{
						if (me == 0)
						{
							for (size_t k = 1; k < p.holders.size(); k++)
								CheckNccl(nccl, nccl->Send(img.first, img.second, rccl::kUint8, p.holders[k], s.comm, st), "ncclSend");
						}
						else CheckNccl(nccl, nccl->Recv(img.first, img.second, rccl::kUint8, p.holders[0], s.comm, st), "ncclRecv");
					}
				}
			}
			catch (const std::exception& e)
			{
				failed = e.what();
			}
			const rccl::Result closed = nccl->GroupEnd(); // always: the peers' transfers of this group must not be left waiting
			if (!failed.empty()) throw std::runtime_error(failed);
			CheckNccl(nccl, closed, "ncclGroupEnd");
			s.batch->WaitStreamBounded(st, "hipStreamSynchronize");
			s.batch->WeightsArrived();
		});
	}

	const float* MultiGpuBatch::GatheredOutput(int shard) const { return shards.at((size_t)shard)->gathered; }

	// Rccl fan-in: every shard runs its rows into ITS part of a [streams][n] device buffer, the parts travel to every GPU (an
	// all-gather of unequal parts: one broadcast per shard, all inside one group), and shard 0 downloads the whole array once
	void MultiGpuBatch::ProcessGathered(const float* in, float* out, size_t n)
	{
		const size_t totalFloats = (size_t)total * n;
		try
		{
		Post([&, in, out, n](Shard& s) {
			CheckHip(hipSetDevice(s.device), "hipSetDevice");
			if (s.gatheredFloats < totalFloats)
			{
				if (s.gathered) (void)hipFree(s.gathered);
				s.gathered = nullptr;
				s.gatheredFloats = 0;
				CheckHip(hipMalloc(reinterpret_cast<void**>(&s.gathered), totalFloats * sizeof(float)), "hipMalloc");
				s.gatheredFloats = totalFloats;
			}
			hipStream_t st = s.batch->GetStream();
			s.batch->ProcessHostToDevice(in + (size_t)s.begin * n, s.gathered + (size_t)s.begin * n, n, (long)n);
			// (everything fallible that is this rank's own business is done: from here a failure is inside the group, which is closed
			// whatever happens -- the peers' broadcasts must not be left waiting for this rank)
			CheckNccl(nccl, nccl->GroupStart(), "ncclGroupStart");
			std::string failed;
			try
			{
				for (const auto& rp : shards)
				{
					float* part = s.gathered + (size_t)rp->begin * n;
					CheckNccl(nccl, nccl->Broadcast(part, part, (size_t)(rp->end - rp->begin) * n, rccl::kFloat32, rp->rank, s.comm, st), "ncclBroadcast");
				}
			}
			catch (const std::exception& e)
			{
				failed = e.what();
			}
			const rccl::Result closed = nccl->GroupEnd();
			if (!failed.empty()) throw std::runtime_error(failed);
			CheckNccl(nccl, closed, "ncclGroupEnd");
			if (s.rank == 0) CheckHip(hipMemcpyAsync(out, s.gathered, totalFloats * sizeof(float), hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H");
			s.batch->WaitStreamBounded(st, "hipStreamSynchronize");
		});
		}
		catch (const std::exception& e)
		{
			// some ranks ran the buffer and some did not (or a transfer failed): the shards' stream states no longer agree with the rows
			// the caller holds -- like a failed Submit, the object refuses further work
			broken = e.what();
			throw;
		}
	}

	// a non-empty range without a batch of exactly its rows would silently leave the caller's rows untouched
	void MultiGpuBatch::CheckShard(const Shard& s) const
	{
		if (s.end <= s.begin) return;
		if (!s.batch || s.batch->NumStreams() != s.end - s.begin)
			throw std::runtime_error("neuralaudio_amd: MultiGpuBatch: a shard has no batch for its stream range");
	}

	void MultiGpuBatch::StopWorkers()
	{
		for (auto& sp : shards)
		{
			Shard& s = *sp;
			{
				std::lock_guard<std::mutex> lock(s.m);
				s.quit = true;
			}
			s.cv.notify_all();
			if (s.worker.joinable()) s.worker.join();
		}
		shards.clear();
	}

	void MultiGpuBatch::CheckUsable() const
	{
		if (!broken.empty()) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch is unusable after a failed submission (" + broken + ")");
		for (const auto& sp : shards) CheckShard(*sp);
	}

	void MultiGpuBatch::ShardRange(int shard, int& begin, int& end, int& device) const
	{
		const Shard& s = *shards.at((size_t)shard);
		begin = s.begin;
		end = s.end;
		device = s.device;
	}

	void MultiGpuBatch::Process(const float* in, float* out, size_t n)
	{
		if (!committed) Commit();
		CheckUsable();
		if (fanIn == FanIn::Rccl)
		{
			ProcessGathered(in, out, n);
			return;
		}
		Post([=](Shard& s) {
			if (s.batch) s.batch->ProcessHost(in + (size_t)s.begin * n, out + (size_t)s.begin * n, n);
		});
	}

	int MultiGpuBatch::Submit(const float* in, size_t n)
	{
		if (!committed) Commit();
		if (fanIn == FanIn::Rccl) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch: the pipelined interface downloads per shard (HostRows fan-in only)");
		// every shard advances its slot ring in lock-step, so one ticket names the same slot everywhere
		CheckUsable();
		int ticket = -1;
		std::mutex tm;
		try
		{
			Post([&, in, n](Shard& s) {
				if (!s.batch) return;
				const int t = s.batch->Submit(in + (size_t)s.begin * n, n);
				std::lock_guard<std::mutex> lock(tm);
				if (ticket >= 0 && ticket != t) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch: shards out of step");
				ticket = t;
			});
		}
		catch (const std::exception& e)
		{
			// some shards took the buffer and some did not: their slot rings (and stream states) no longer agree.  Nothing can put the
			// failed shard's streams back in step, so the object refuses further work instead of returning rows it never computed.
			broken = e.what();
			throw;
		}
		return ticket;
	}

	void MultiGpuBatch::Collect(int ticket, float* out)
	{
		CheckUsable();
		Post([=](Shard& s) {
			if (s.batch) s.batch->Collect(ticket, out ? out + (size_t)s.begin * s.batch->SlotFrames(ticket) : nullptr);
		});
	}

	void MultiGpuBatch::SetQuality(int stream, float quality)
	{
		if (!committed) Commit();
		if (stream < 0 || stream >= total) throw std::runtime_error("neuralaudio_amd: MultiGpuBatch::SetQuality: stream id outside the batch");
		Post([=](Shard& s) {
			if (s.batch && stream >= s.begin && stream < s.end) s.batch->SetQuality(stream - s.begin, quality);
		});
	}
}
