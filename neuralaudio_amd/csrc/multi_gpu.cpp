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
			Post([this](Shard& s) {
				if (s.end <= s.begin) return;
				s.batch.reset(new GpuBatch(s.device));
				int first = 0;
				for (const Entry& e : entries)
				{
					const int a = std::max(first, s.begin), b = std::min(first + e.count, s.end);
					if (b > a) s.batch->AddStreams(e.model, e.quality, b - a, e.prewarm, e.onDemand);
					first += e.count;
				}
			});
			for (auto& sp : shards) CheckShard(*sp);
		}
		catch (...)
		{
			StopWorkers();
			throw;
		}
		committed = true;
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
		Post([=](Shard& s) {
			if (s.batch) s.batch->ProcessHost(in + (size_t)s.begin * n, out + (size_t)s.begin * n, n);
		});
	}

	int MultiGpuBatch::Submit(const float* in, size_t n)
	{
		if (!committed) Commit();
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
