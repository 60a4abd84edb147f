// gpu_batch_chains.cpp -- how a buffer of a large batch gets onto the chip when the caller does not order the work on a stream of its own
// (GpuBatch::ProcessDevice on the batch's own, never handed-out stream; the pipelined host-buffer entry points):
//   * free-running chains: the buffer as two launches of half of every group's streams each, on two HIP streams that never wait for each
//     other;
//   * the resident launch: ONE launch stays on the chip and walks consecutive buffers itself, fed through a command ring in pinned host
//     memory (wavenet_launch.h ResidentCtrl, wavenet_spec_impl.h WaveNetSpecResidentKernel);
// and the timing marks bench.py brackets its steps with.  Part of class GpuBatch (gpu_batch.h).
#include "gpu_batch_internal.h"
#include <immintrin.h>
#include <thread>

namespace na
{
	// ---- the resident launch ---------------------------------------------------------------------------------------------------------
	// Who: a batch on its own, never handed-out stream (like the chains: nobody outside can observe the order of work on streams they
	// never saw) whose buffer is ONE launch of 128-frame blocks of A1 Standard streams (wavenet_spec_kernels.hip ResidentDispatch), at
	// least 512 of them.  What: NA_BatchProcessDevice posts a command (pointers, strides) into a ring in pinned host memory and returns;
	// the launch -- started by the first command, as many workgroups as are resident -- picks it up, workgroup by workgroup, each as
	// soon as it is through with the previous buffer: no launch boundary between buffers, no dispatch ramp, no drain.  Every entry
	// point that touches stream state, index lists or the batch stream first retires the launch (DrainResident: exitAfter = last posted
	// command, wait for the stream); a launch whose workgroups found nothing to do for NA_RESIDENT_IDLE_US (200 us) leaves by itself and
	// is started again by the next command -- the chip is never held by an idle batch, and hipDeviceSynchronize() always returns.
	// Contract of the device pointers (include/neuralaudio_amd.h): input rows complete before the call, output rows valid after
	// NA_BatchSynchronize (or the closing mark); rows are read and written at system scope (no kernel boundary orders the caches).
	namespace
	{
		constexpr unsigned long long kNever = ~0ull;

		unsigned long long HostLoad(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
		// a word for the launch: the block may be device memory behind the BAR (write-combined on the host side) -- the fence pushes the
		// store, and everything written before it, out of the core's write-combining buffers
		void HostStore(unsigned long long* p, unsigned long long v)
		{
			__atomic_store_n(p, v, __ATOMIC_RELEASE);
			_mm_sfence();
		}
	}

	// ---- bounded waits (gpu_batch.h) --------------------------------------------------------------------------------------------------
	double GpuBatch::DefaultWaitLimitMs()
	{
		static const double ms = [] {
			const char* e = getenv("NA_WAIT_LIMIT_MS");
			return (e && *e) ? atof(e) : 2000.0;
		}();
		return ms;
	}

	void GpuBatch::CheckUsable() const
	{
		if (broken) throw std::runtime_error("neuralaudio_amd: the batch is broken (" + brokenWhy + "); destroy it");
	}

	void GpuBatch::Stall(const char* what)
	{
		char text[256];
		snprintf(text, sizeof text, "the device did not answer within %.0f ms: %s", waitLimitMs, what);
		if (!broken) brokenWhy = text;
		broken = true;
		throw std::runtime_error(std::string("neuralaudio_amd: ") + text);
	}

	namespace
	{
		// one turn of a poll loop: busy for the first millisecond (a buffer takes 40 - 300 us), then 50 us naps
		struct Poller
		{
			std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
			void Turn()
			{
				if (std::chrono::steady_clock::now() - start < std::chrono::milliseconds(1)) _mm_pause();
				else std::this_thread::sleep_for(std::chrono::microseconds(50));
			}
		};
	}

	void GpuBatch::WaitStreamBounded(hipStream_t s, const char* what)
	{
		if (waitLimitMs <= 0)
		{
			CheckHip(hipStreamSynchronize(s), what);
			return;
		}
		const Deadline deadline(waitLimitMs);
		Poller poll;
		for (;;)
		{
			const hipError_t q = hipStreamQuery(s);
			if (q == hipSuccess) return;
			if (q != hipErrorNotReady) CheckHip(q, what);
			if (deadline.Expired()) Stall(what);
			poll.Turn();
		}
	}

	void GpuBatch::WaitEventBounded(hipEvent_t e, const char* what)
	{
		if (waitLimitMs <= 0)
		{
			CheckHip(hipEventSynchronize(e), what);
			return;
		}
		const Deadline deadline(waitLimitMs);
		Poller poll;
		for (;;)
		{
			const hipError_t q = hipEventQuery(e);
			if (q == hipSuccess) return;
			if (q != hipErrorNotReady) CheckHip(q, what);
			if (deadline.Expired()) Stall(what);
			poll.Turn();
		}
	}

	void GpuBatch::DebugStallDevice(double ms)
	{
		CheckUsable();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainResident();
		CheckHip(LaunchStallKernel(std::min(std::max(ms, 0.0), 10000.0), stream), "stall kernel");
		if (halfChainsUsed)
			for (hipStream_t hs : halfStream)
				if (hs) CheckHip(LaunchStallKernel(std::min(std::max(ms, 0.0), 10000.0), hs), "stall kernel");
	}

	// (re)starts the launch of the current generation unless one is still on the stream
	void GpuBatch::ResidentEnsureRunning()
	{
		ResidentState& r = *residentState;
		if (r.launched)
		{
			const hipError_t q = hipEventQuery(r.gen);
			if (q == hipErrorNotReady) return;
			CheckHip(q, "hipEventQuery (resident launch)");
		}
		ResidentArgs ra = {};
		ra.ctrl = r.dCtrl;
		ra.status = r.dStatus;
		ra.doneCount = r.dDone;
		ra.wgDone = r.dWgDone;
		ra.base = r.base;
		ra.idleTicks = (unsigned)std::max(1, Tuning::Get().residentIdleUs) * 100u; // s_memrealtime: 100 MHz
		ra.startDelay = (unsigned)std::max(0, Tuning::Get().residentDelayUs) * 100u;
		int grid = 0;
		CheckHip(LaunchWaveNetSpecResident(r.list.data(), (int)r.list.size(), WN_MAX_FRAMES, ra, stream, &grid), "WaveNet resident launch");
		if (grid != r.grid) throw std::runtime_error("neuralaudio_amd: internal: resident grid changed inside a generation");
		if (!r.gen) CheckHip(hipEventCreateWithFlags(&r.gen, hipEventDisableTiming), "hipEventCreate");
		CheckHip(hipEventRecord(r.gen, stream), "hipEventRecord");
		r.launched = true;
	}

	// The launch list of the batch as it is now, for the resident launch; false: this topology does not run resident.
	bool GpuBatch::ResidentConfigure()
	{
		ResidentState& r = *residentState;
		r.configured = false;
		r.unsupported = true;
		r.topology = topologyVersion;
		r.list.clear();
		int active = 0, total = 0;
		for (const auto& g : groups)
		{
			if (g->NumActive() == 0) continue;
			const int c = g->LaunchClass();
			if (c != 1 && c != -1) return false; // (plain f16-split launches only)
			active++;
		}
		if (active == 0 || active > WN_FRAME_MAX_GROUPS) return false;
		// whatever else this batch has in flight comes first (state resets and prewarms on the batch stream, slot streams, the chains) ...
		DrainPipeline();
		WaitStreamBounded(stream, "hipStreamSynchronize");
		for (const auto& g : groups)
		{
			if (g->NumActive() == 0) continue;
			WnFrameGroup a = {};
			int list = 0;
			if (!g->FusedLaunchArgs(a, list)) return false; // (uploads a changed index list, asynchronously on the batch stream)
			if (a.pack > 1 || a.model->spec_arch != WN_SPEC_STD) return false;
			total += a.numStreams;
			r.list.push_back(a);
		}
		// ... and so do the index lists
		WaitStreamBounded(stream, "hipStreamSynchronize");
		if (total < 512) return false; // (a launch that does not fill the chip: nothing to gain)
		const int grid = WaveNetSpecResidentGrid(r.list.data(), (int)r.list.size(), WN_MAX_FRAMES);
		if (grid < 1) return false;
		if (!r.ctrl)
		{
			// the command block: device memory the host can write, where the BAR covers it (wavenet_launch.h ResidentCtrl)
			int largeBar = 0;
			if (!Tuning::Get().residentHostRing && hipDeviceGetAttribute(&largeBar, hipDeviceAttributeIsLargeBar, device) == hipSuccess && largeBar &&
				hipExtMallocWithFlags(reinterpret_cast<void**>(&r.ctrl), sizeof(ResidentCtrl), hipDeviceMallocFinegrained) == hipSuccess)
			{
				r.ctrlInDeviceMemory = true;
				r.dCtrl = r.ctrl;
				memset(r.ctrl, 0, sizeof(ResidentCtrl)); // (through the BAR)
			}
			else
			{
				(void)hipGetLastError();
				r.ctrl = nullptr;
				CheckHip(hipHostMalloc(reinterpret_cast<void**>(&r.ctrl), sizeof(ResidentCtrl), hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc (command ring)");
				memset(r.ctrl, 0, sizeof(ResidentCtrl));
				CheckHip(hipHostGetDevicePointer(reinterpret_cast<void**>(&r.dCtrl), r.ctrl, 0), "hipHostGetDevicePointer");
			}
			HostStore(&r.ctrl->exitAfter, kNever);
			CheckHip(hipHostMalloc(reinterpret_cast<void**>(&r.status), sizeof(ResidentStatus), hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc (resident status)");
			memset(r.status, 0, sizeof(ResidentStatus));
			CheckHip(hipHostGetDevicePointer(reinterpret_cast<void**>(&r.dStatus), r.status, 0), "hipHostGetDevicePointer");
			CheckHip(hipMalloc(reinterpret_cast<void**>(&r.dDone), RESIDENT_RING * sizeof(unsigned)), "hipMalloc");
			CheckHip(hipMemsetAsync(r.dDone, 0, RESIDENT_RING * sizeof(unsigned), stream), "hipMemsetAsync");
		}
		if (grid > r.wgCapacity)
		{
			if (r.dWgDone) (void)hipFree(r.dWgDone);
			r.dWgDone = nullptr;
			r.wgCapacity = 0;
			CheckHip(hipMalloc(reinterpret_cast<void**>(&r.dWgDone), (size_t)grid * sizeof(unsigned)), "hipMalloc");
			r.wgCapacity = grid;
		}
		CheckHip(hipMemsetAsync(r.dWgDone, 0, (size_t)grid * sizeof(unsigned), stream), "hipMemsetAsync"); // (the launch is ordered behind it)
		r.grid = grid;
		r.unsupported = false;
		r.configured = true;
		return true;
	}

	// The buffer through the resident launch; false: not this batch / this buffer (the caller runs it the other ways).
	bool GpuBatch::TryResident(const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		if (!residentWanted || n % (size_t)WN_MAX_FRAMES != 0)
		{
			DrainResident(); // (a buffer of another length runs as ordinary launches: behind everything the resident launch still holds)
			return false;
		}
		if (!residentState) residentState.reset(new ResidentState());
		ResidentState& r = *residentState;
		bool dirty = false;
		for (const auto& g : groups) dirty = dirty || (g->NumActive() > 0 && g->ListsDirty());
		if (r.topology != topologyVersion || dirty || (!r.configured && !r.unsupported))
		{
			DrainResident(); // (the running launch holds the old tables)
			if (!ResidentConfigure()) return false;
		}
		if (!r.configured) return false;
		if (pipelineUsed)
			for (PipeSlot& p : pipe) // (buffers submitted through the pipelined host interface run on per-slot streams: behind them as well)
				if (p.own) WaitStreamBounded(p.own, "hipStreamSynchronize");
		if (halfChainsUsed)
		{
			// an earlier buffer of another length ran as half-batch launches on the chain streams: the command comes behind them
			for (hipStream_t hs : halfStream)
				if (hs) WaitStreamBounded(hs, "hipStreamSynchronize");
			halfChainsUsed = false;
		}
		if (r.exitRequested) DrainResident(); // a closing mark asked the launch to leave: this command starts the next generation
		for (size_t offset = 0; offset < n; offset += (size_t)WN_MAX_FRAMES)
		{
			const unsigned long long seq = r.posted + 1;
			// back-pressure: a slot (and its done counter) is free once the command RESIDENT_RING before it has completed
			if (seq - HostLoad(&r.status->completed) >= (unsigned long long)RESIDENT_RING - 1)
			{
				const Deadline deadline(waitLimitMs);
				while (seq - HostLoad(&r.status->completed) >= (unsigned long long)RESIDENT_RING - 1)
				{
					ResidentEnsureRunning();
					if (deadline.Expired()) Stall("resident launch: no free command slot");
				}
			}
			// (write-only on this side: the block may be device memory behind the BAR)
			ResidentCmd& c = r.ctrl->cmd[seq % RESIDENT_RING];
			c.in = dIn + offset;
			c.out = dOut + offset;
			c.inStride = inStride;
			c.outStride = outStride;
			c.check = ResidentCmdCheck((unsigned long long)(size_t)(dIn + offset), (unsigned long long)(size_t)(dOut + offset), (unsigned long long)inStride, (unsigned long long)outStride, seq);
			HostStore(&c.seq, seq);
			r.posted = seq;
			ResidentEnsureRunning();
		}
		lastStepHalves = false;
		lastStepResident = true;
		return true;
	}

	void GpuBatch::SetResidentLaunch(bool on)
	{
		CheckUsable();
		if (!on) DrainResident();
		residentWanted = on;
	}

	// Every posted command has run and the launch has left the chip; the next command starts a new generation.
	void GpuBatch::DrainResident()
	{
		if (!residentState) return;
		ResidentState& r = *residentState;
		if (!r.launched && r.posted == r.base) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		HostStore(&r.ctrl->exitAfter, r.posted);
		const Deadline deadline(waitLimitMs);
		for (;;)
		{
			if (r.launched) WaitStreamBounded(stream, "resident launch: leaving the chip");
			r.launched = false;
			if (HostLoad(&r.status->completed) >= r.posted) break;
			if (deadline.Expired()) Stall("resident launch: posted commands not completed");
			ResidentEnsureRunning(); // (it idled out, or left at an earlier exit mark, before it saw the last commands: once more)
		}
		HostStore(&r.ctrl->exitAfter, kNever);
		r.exitRequested = false;
		r.base = r.posted;
		if (r.dWgDone && r.grid > 0) CheckHip(hipMemsetAsync(r.dWgDone, 0, (size_t)r.grid * sizeof(unsigned), stream), "hipMemsetAsync");
	}

	void GpuBatch::WaitOutputs()
	{
		CheckUsable();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		if (residentState && (residentState->launched || residentState->posted != residentState->base))
		{
			// (the launch stays up: only its done count is awaited -- and it goes in again should it have idled out early)
			ResidentState& r = *residentState;
			const Deadline deadline(waitLimitMs);
			while (HostLoad(&r.status->completed) < r.posted)
			{
				ResidentEnsureRunning();
				if (deadline.Expired()) Stall("resident launch: waiting for the outputs");
			}
			return;
		}
		if (halfChainsUsed)
			for (hipStream_t hs : halfStream)
				if (hs) WaitStreamBounded(hs, "hipStreamSynchronize");
		WaitStreamBounded(stream, "hipStreamSynchronize");
	}

	// closing timing mark: the launch leaves behind the last posted command, without waiting for it
	void GpuBatch::ResidentExitAfterPosted()
	{
		if (!residentState) return;
		ResidentState& r = *residentState;
		if (r.posted == r.base && !r.launched) return;
		HostStore(&r.ctrl->exitAfter, r.posted);
		r.exitRequested = true;
		r.markPosted = r.posted;
		if (HostLoad(&r.status->completed) < r.posted) ResidentEnsureRunning();
	}

	// the ONE helper behind every entry point that touches stream state, index lists or device allocations
	void GpuBatch::Quiesce()
	{
		CheckUsable();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline(); // (resident launch, pipeline slots, half-batch chains)
		WaitStreamBounded(stream, "hipStreamSynchronize");
	}

	void GpuBatch::JoinHalves()
	{
		DrainResident();
		if (!halfChainsUsed) return;
		for (hipStream_t hs : halfStream)
			if (hs) WaitStreamBounded(hs, "hipStreamSynchronize");
		halfChainsUsed = false;
	}

	// The buffer as two launch lists of half of every group's streams each (see halfStream); false: it runs as ordered launches.
	bool GpuBatch::PrepareHalves(size_t n)
	{
		if (Tuning::Get().hostHalvesOff) return false;
		bool dirty = false, packed = false, plain = false;
		int active = 0, kernelStreams = 0;
		for (const auto& g : groups)
		{
			const int members = g->NumActive();
			if (members == 0) continue;
			const int c = g->LaunchClass(); // 1 / 2 / -1: the f16-split kernels' plain launch / packed launch / either (gpu_batch.cpp LaunchClass)
			if (c != 1 && c != 2 && c != -1) return false;
			packed = packed || c == 2;
			plain = plain || c == 1;
			dirty = dirty || g->ListsDirty();
			kernelStreams += (members + g->PackFactor() - 1) / g->PackFactor();
			active++;
		}
		if (active == 0 || active > WN_FRAME_MAX_GROUPS || (packed && plain)) return false; // (two launches per buffer: not split)
		if (kernelStreams < 512) return false; // (a small batch: nothing below is worth its host time; the exact count is checked at the end)
		// changed index lists are re-uploaded below (asynchronously, on the batch stream): nothing in flight may still read the old ones
		if (dirty && (halfChainsUsed || pipelineUsed)) DrainPipeline();
		if (!halfLists) halfLists.reset(new HalfLists());
		HalfLists& hl = *halfLists;
		for (auto& part : hl.part) part.clear();
		hl.listsUploaded = dirty;
		int total = 0;
		bool compact = false;
		for (const auto& g : groups)
		{
			if (g->NumActive() == 0) continue;
			WnFrameGroup a = {};
			int list = 0;
			if (!g->FusedLaunchArgs(a, list)) return false;
			WaveNetGroup* wg = static_cast<WaveNetGroup*>(g.get());
			if (list < 0 && packed) a.slots = wg->listSlots; // a plain group in the packed launch passes its index lists
			total += a.numStreams;
			compact = compact || a.model->compact_rings != 0;
			// contiguous parts of whole workgroups (two streams each)
			int first = 0;
			for (int c = 0; c < numChains; c++)
			{
				const int end = c + 1 == numChains ? a.numStreams : std::min(a.numStreams, (int)(((long)a.numStreams * (c + 1) / numChains + 1) & ~1L));
				if (end <= first) continue;
				WnFrameGroup part = a;
				part.numStreams = end - first;
				part.slot0 += first;
				part.row0 += first;
				if (part.slots) part.slots += first;
				if (part.slots || a.pack > 1) part.rows += (size_t)first * (size_t)a.pack;
				hl.part[c].push_back(part);
				first = end;
			}
		}
		// (below 512 kernel-level streams a launch does not fill the chip anyway: nothing to overlap)
		hl.compact = compact;
		(void)n; // (any buffer length: a chain runs the chunks of a long buffer one after the other, LaunchChain)
		return total >= 512;
	}

	// the chains are about to take launches: whatever else is in flight for this batch comes first, and their streams exist
	void GpuBatch::BeginHalves()
	{
		if (!halfChainsUsed || submitTopology != topologyVersion || halfLists->listsUploaded)
		{
			// whatever the batch stream (state resets, prewarms of new streams, index lists) or a slot stream still has in flight comes first
			DrainPipeline();
			WaitStreamBounded(stream, "hipStreamSynchronize");
			submitTopology = topologyVersion;
		}
		for (int h = 0; h < numChains; h++)
		{
			if (halfStream[h]) continue;
			CheckHip(hipStreamCreateWithFlags(&halfStream[h], hipStreamNonBlocking), "hipStreamCreate");
			if (markOpen)
			{
				if (!marks[1 + h][0]) CheckHip(hipEventCreate(&marks[1 + h][0]), "hipEventCreate");
				CheckHip(hipEventRecord(marks[1 + h][0], halfStream[h]), "hipEventRecord");
			}
		}
		halfChainsUsed = true;
		lastStepHalves = true;
		lastStepResident = false;
	}

	// list h of PrepareHalves on its own stream, behind that chain's previous launch
	void GpuBatch::LaunchChain(int h, const float* dIn, float* dOut, size_t n, long inStride, long outStride, bool hostRows)
	{
		// Workgroup shape: for rows in HBM, sized for what is resident with both chains on the chip (full-size workgroups: 36.7 vs 37.1 us
		// per 1024 x 128 Standard step); for rows in pinned host memory the half-size workgroups win (43.5-44.4 vs 45.1-46.2 us per buffer
		// host to host: twice the workgroups keep more PCIe reads in flight)
		const std::vector<WnFrameGroup>& part = halfLists->part[h];
		// (trace builds, tools/trace_split_timeline.py: the traced workgroup index exists in every chain's launch -- only chain NA_TRACE_CHAIN stamps)
		long long* const trace = GetWaveNetTraceBuffer();
		const int traceChain = Tuning::Get().traceChain;
		if (trace != nullptr && h != traceChain) SetWaveNetTraceBuffer(nullptr);
		if (!part.empty())
		{
			// (a buffer longer than a launch takes: the chunks one after the other on this chain -- the chains still never wait for each other)
			size_t offset = 0, left = n;
			while (left > 0)
			{
				const int chunk = NextWaveNetChunk(left, halfLists->compact);
				CheckHip(LaunchWaveNetSplitFused(part.data(), (int)part.size(), dIn + offset, dOut + offset, inStride, outStride, chunk, halfStream[h],
					(hostRows ? 1 : numChains) | ((!Tuning::Get().wnNtOff && StateBytes() > ((size_t)Tuning::Get().wnNtFromMB << 20)) ? WN_SHARING_BEYOND_CACHE : 0)), "WaveNet kernel (half batch)");
				offset += (size_t)chunk;
				left -= (size_t)chunk;
			}
		}
		if (trace != nullptr) SetWaveNetTraceBuffer(trace);
	}

	// the lists of PrepareHalves, each on its own stream; `done`: events to record
	void GpuBatch::LaunchHalves(const float* dIn, float* dOut, size_t n, long inStride, long outStride, hipEvent_t* done, bool hostRows)
	{
		BeginHalves();
		for (int h = 0; h < numChains; h++)
		{
			LaunchChain(h, dIn, dOut, n, inStride, outStride, hostRows);
			if (done) CheckHip(hipEventRecord(done[h], halfStream[h]), "hipEventRecord");
		}
	}

	void GpuBatch::MarkTime(int which)
	{
		CheckUsable();
		if (which < 0 || which > 1) throw std::runtime_error("neuralaudio_amd: MarkTime(0 | 1)");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		// (a chain stream that does not exist yet is created by the first launch that needs it -- 13 ms, not inside a timed window if
		// nothing will run on it -- and gets its start mark then: LaunchHalves)
		markOpen = which == 0;
		// the resident launch: an opening mark starts a new generation (its launch comes after the mark on the batch stream), a closing
		// mark tells the launch to leave behind the last posted command -- the two events on the batch stream bracket exactly those steps
		if (which == 0) DrainResident();
		else ResidentExitAfterPosted();
		for (int i = 0; i <= kMaxChains; i++)
		{
			hipStream_t s = i == 0 ? stream : halfStream[i - 1];
			if (!s || (which == 1 && !marks[i][0])) continue;
			if (!marks[i][which]) CheckHip(hipEventCreate(&marks[i][which]), "hipEventCreate");
			CheckHip(hipEventRecord(marks[i][which], s), "hipEventRecord");
		}
	}

	// polls the closing marks (a benchmark's closing wait should not pay the wake-up latency of a blocking synchronisation)
	void GpuBatch::WaitMarks()
	{
		CheckUsable();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		// (polled from the first turn on: WaitEventBounded)
		for (int i = 0; i <= kMaxChains; i++)
			if (marks[i][0] && marks[i][1]) WaitEventBounded(marks[i][1], "closing timing mark");
		ResidentFinishMarked();
	}

	// The closing mark was reached but a workgroup of the resident launch left before it saw the last commands (it had idled out at that
	// very moment): the launch goes in again and the closing mark moves behind it, so that the span covers all the marked work.
	void GpuBatch::ResidentFinishMarked()
	{
		if (!residentState || !residentState->exitRequested) return;
		ResidentState& r = *residentState;
		const Deadline deadline(waitLimitMs);
		while (HostLoad(&r.status->completed) < r.markPosted)
		{
			if (deadline.Expired()) Stall("resident launch: marked commands not completed");
			ResidentEnsureRunning();
			if (marks[0][1]) CheckHip(hipEventRecord(marks[0][1], stream), "hipEventRecord");
			WaitEventBounded(r.gen, "resident launch: leaving the chip");
		}
	}

	float GpuBatch::ElapsedMs()
	{
		CheckUsable();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		WaitMarks();
		float longest = 0.0f;
		for (int i = 0; i <= kMaxChains; i++)
		{
			if (!marks[i][0] || !marks[i][1]) continue;
			// (polled: a benchmark's closing wait should not pay the wake-up latency of a blocking synchronisation -- ~25 us of a 20-step run)
			WaitEventBounded(marks[i][1], "closing timing mark");
			float ms = 0.0f;
			CheckHip(hipEventElapsedTime(&ms, marks[i][0], marks[i][1]), "hipEventElapsedTime");
			longest = std::max(longest, ms);
		}
		return longest;
	}

}
