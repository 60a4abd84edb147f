// gpu_batch_host.cpp -- the host-buffer entry points of GpuBatch (gpu_batch.h): the blocking call (ProcessHost), registered host blocks, and
// the pipelined Submit / Collect interface with its pinned slots.  Reference counterpart: the caller's float* buffers of
// NeuralModel::Process (NeuralAudio/NeuralModel.h:127) -- here [streams][n] arrays in host memory.
#include "gpu_batch_internal.h"

#include <mutex>

namespace na
{
	// every buffer still in flight on a slot stream (pipelined interface) is done after this
	void GpuBatch::DrainPipeline()
	{
		DrainResident();
		for (PipeSlot& p : pipe)
			if (p.own) WaitStreamBounded(p.own, "hipStreamSynchronize");
		for (hipStream_t hs : halfStream)
			if (hs) WaitStreamBounded(hs, "hipStreamSynchronize");
		halfChainsUsed = false; // (the next half-batch launches wait for the batch stream first: LaunchHalves)
	}

	void GpuBatch::EnsureStaging(size_t floats)
	{
		if (floats <= stageFloats) return;
		if (hostStage) (void)hipHostFree(hostStage);
		if (devStage) (void)hipFree(devStage);
		hostStage = nullptr;
		devStage = nullptr;
		stageFloats = 0;
		CheckHip(hipHostMalloc(reinterpret_cast<void**>(&hostStage), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
		CheckHip(hipMalloc(reinterpret_cast<void**>(&devStage), floats * sizeof(float)), "hipMalloc");
		stageFloats = floats;
	}

	// Direct mode (the default; NA_HOST_DIRECT=0 selects the copy engines): the kernels read the block straight from the pinned host buffer
	// and write their output straight into it (every kernel touches `in` once in its prologue and `out` once in its head), so a buffer
	// is ONE launch instead of copy + launch + copy.  The 2 x 512 KB of a 1024 x 128 block still cross PCIe, inside the kernel, but the
	// two asynchronous copies (each ~10 us of latency before its first byte moves) and the waits between them are gone.  Measured on
	// MI355X with this round's kernels (tools/HostPipeBench, 1024 streams x 128 frames; round 2 had it the other way round for the
	// pipelined path, 66.8 vs 61.9 us, and kept the copies):
	//   A1 Standard  Submit..Collect in place 91.8 -> 59.4 us (p50), pipelined 52.9 -> 51.4 us per buffer;  64 streams: 44.9 -> 28.9 us
	//   Nano / Feather / LSTM 1x16 / 2x8      75.8 / 74.6 / 69.4 / 71.2 -> 54.0 / 52.4 / 49.9 / 50.5 us;  A2 96.9 -> 68.4;  4096 Standard 262 -> 180
	bool HostDirect()
	{
		return Tuning::Get().hostDirect;
	}

	// ---- registered host blocks ----
	namespace
	{
		struct HostBlock
		{
			char* host;
			char* dev;
			size_t bytes;
		};
		std::mutex gHostBlocksMutex;
		std::vector<HostBlock> gHostBlocks;
	}
	bool RegisterHostBuffer(void* p, size_t bytes, std::string& error)
	{
		if (!p || bytes == 0) { error = "neuralaudio_amd: RegisterHostBuffer with an empty block"; return false; }
		hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
		if (e != hipSuccess) { error = std::string("neuralaudio_amd: hipHostRegister: ") + hipGetErrorString(e); return false; }
		void* d = nullptr;
		e = hipHostGetDevicePointer(&d, p, 0);
		if (e != hipSuccess || !d)
		{
			(void)hipHostUnregister(p);
			error = std::string("neuralaudio_amd: hipHostGetDevicePointer: ") + hipGetErrorString(e);
			return false;
		}
		std::lock_guard<std::mutex> lock(gHostBlocksMutex);
		gHostBlocks.push_back({ static_cast<char*>(p), static_cast<char*>(d), bytes });
		return true;
	}
	bool UnregisterHostBuffer(void* p)
	{
		std::lock_guard<std::mutex> lock(gHostBlocksMutex);
		for (size_t i = 0; i < gHostBlocks.size(); i++)
		{
			if (gHostBlocks[i].host != p) continue;
			gHostBlocks.erase(gHostBlocks.begin() + (long)i);
			return hipHostUnregister(p) == hipSuccess;
		}
		return false;
	}
	void* RegisteredDevicePointer(const void* p, size_t bytes)
	{
		const char* c = static_cast<const char*>(p);
		std::lock_guard<std::mutex> lock(gHostBlocksMutex);
		for (const HostBlock& b : gHostBlocks)
			if (c >= b.host && c + bytes <= b.host + b.bytes) return b.dev + (c - b.host);
		return nullptr;
	}

	// (Splitting the buffer into chunks so that host copies overlap the DMA was measured and dropped: every extra asynchronous copy /
	// event costs more than it hides -- 1024 x 128 frames: 114 us per call as one piece, 134 / 186 / 282 us in 2 / 4 / 8 chunks.)
	void GpuBatch::ProcessHost(const float* in, float* out, size_t n)
	{
		CheckUsable();
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		const size_t total = streams.size() * n;
		if (HostDirect())
		{
			// blocks the caller registered: the kernels run on them as they are (no staging copies: 81 -> ~62 us for 1024 x 128)
			float* dIn = static_cast<float*>(RegisteredDevicePointer(in, total * sizeof(float)));
			float* dOut = static_cast<float*>(RegisteredDevicePointer(out, total * sizeof(float)));
			if (dIn && dOut)
			{
				ProcessDeviceOrdered(dIn, dOut, n, (long)n, (long)n);
				WaitStreamBounded(stream, "hipStreamSynchronize");
				ZeroRetiredRows(out, n, streams.size());
				return;
			}
		}
		EnsureStaging(total);
		float* dStage = nullptr;
		const bool direct = HostDirect() && hipHostGetDevicePointer(reinterpret_cast<void**>(&dStage), hostStage, 0) == hipSuccess && dStage != nullptr;
		if (direct && PrepareHalves(n) && halfLists->RowRangesOnly())
		{
			// The blocking call in two halves: the rows of the first half are staged and launched, the second half is staged while the
			// first runs, and the first half's result is copied out while the second still runs -- the two 512 KB host copies of a
			// 1024 x 128 buffer (2 x 10 us) hide behind the kernels: p50 75-77 -> 60 us.
			BeginHalves();
			for (int h = 0; h < numChains; h++)
			{
				for (const WnFrameGroup& g : halfLists->part[h])
					memcpy(hostStage + (size_t)g.row0 * n, in + (size_t)g.row0 * n, (size_t)g.numStreams * n * sizeof(float));
				LaunchChain(h, dStage, dStage, n, (long)n, (long)n, true);
			}
			for (int h = 0; h < numChains; h++)
			{
				WaitStreamBounded(halfStream[h], "hipStreamSynchronize");
				for (const WnFrameGroup& g : halfLists->part[h])
					memcpy(out + (size_t)g.row0 * n, hostStage + (size_t)g.row0 * n, (size_t)g.numStreams * n * sizeof(float));
			}
			halfChainsUsed = false; // (both chains are idle again)
			ZeroRetiredRows(out, n, streams.size());
			return;
		}
		memcpy(hostStage, in, total * sizeof(float));
		// (a pinned block the device cannot address -- not seen on MI355X -- goes through the copy engines instead of failing)
		if (direct)
			ProcessDeviceOrdered(dStage, dStage, n, (long)n, (long)n);
		else
		{
			CheckHip(hipMemcpyAsync(devStage, hostStage, total * sizeof(float), hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
			ProcessDeviceOrdered(devStage, devStage, n, (long)n, (long)n);
			CheckHip(hipMemcpyAsync(hostStage, devStage, total * sizeof(float), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H");
		}
		JoinHalves();
		WaitStreamBounded(stream, "hipStreamSynchronize");
		memcpy(out, hostStage, total * sizeof(float));
		ZeroRetiredRows(out, n, streams.size());
	}

	void GpuBatch::ProcessHostToDevice(const float* in, float* dOut, size_t n, long outStride)
	{
		CheckUsable();
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		const size_t total = streams.size() * n;
		// the previous call's kernels may still be reading the pinned block
		WaitStreamBounded(stream, "hipStreamSynchronize");
		EnsureStaging(total);
		memcpy(hostStage, in, total * sizeof(float));
		float* dStage = nullptr;
		if (HostDirect() && hipHostGetDevicePointer(reinterpret_cast<void**>(&dStage), hostStage, 0) == hipSuccess && dStage != nullptr)
			ProcessDeviceOrdered(dStage, dOut, n, (long)n, outStride);
		else
		{
			CheckHip(hipMemcpyAsync(devStage, hostStage, total * sizeof(float), hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
			ProcessDeviceOrdered(devStage, dOut, n, (long)n, outStride);
		}
	}

	void GpuBatch::EnsurePipeSlot(PipeSlot& p, size_t floats)
	{
		if (!p.uploaded)
		{
			CheckHip(hipEventCreateWithFlags(&p.uploaded, hipEventDisableTiming), "hipEventCreate");
			CheckHip(hipEventCreateWithFlags(&p.computed, hipEventDisableTiming), "hipEventCreate");
			CheckHip(hipEventCreateWithFlags(&p.downloaded, hipEventDisableTiming), "hipEventCreate");
			// the set-up side of the pipelined interface: the half-batch chains' streams too (creating a HIP stream takes ~13 ms)
			for (int h = 0; h < numChains; h++)
				if (!halfStream[h]) CheckHip(hipStreamCreateWithFlags(&halfStream[h], hipStreamNonBlocking), "hipStreamCreate");
		}
		if (floats <= p.floats) return;
		if (p.hostIn) (void)hipHostFree(p.hostIn);
		if (p.hostOut) (void)hipHostFree(p.hostOut);
		if (p.dev) (void)hipFree(p.dev);
		p.hostIn = p.hostOut = p.dev = nullptr;
		p.floats = 0;
		CheckHip(hipHostMalloc(reinterpret_cast<void**>(&p.hostIn), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
		CheckHip(hipHostMalloc(reinterpret_cast<void**>(&p.hostOut), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
		CheckHip(hipMalloc(reinterpret_cast<void**>(&p.dev), floats * sizeof(float)), "hipMalloc");
		p.floats = floats;
	}

	int GpuBatch::Submit(const float* in, size_t n)
	{
		CheckUsable();
		if (n == 0 || streams.empty()) throw std::runtime_error("neuralaudio_amd: Submit on an empty batch / buffer");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		const int ticket = nextSlot;
		PipeSlot& p = pipe[ticket];
		if (p.busy) throw std::runtime_error("neuralaudio_amd: Submit with every pipeline slot in flight (Collect the oldest ticket first)");
		const size_t total = streams.size() * n;
		EnsurePipeSlot(p, total);
		p.n = n;
		p.rows = streams.size(); // Collect sizes its copy by THIS (AddStreams / RemoveStreams may run while the ticket is in flight)
		if (in) memcpy(p.hostIn, in, total * sizeof(float)); // nullptr: the caller filled NextInput() in place
		const bool direct = HostDirect(); // (see ProcessHost)
		float *dIn = nullptr, *dOut = nullptr;
		if (direct && hipHostGetDevicePointer(reinterpret_cast<void**>(&dIn), p.hostIn, 0) == hipSuccess && dIn != nullptr &&
			hipHostGetDevicePointer(reinterpret_cast<void**>(&dOut), p.hostOut, 0) == hipSuccess && dOut != nullptr)
		{
			// (a lone buffer gains nothing from being split -- 55-58 vs 60 us Submit .. Collect -- so only with another ticket in flight)
			bool othersInFlight = false;
			for (const PipeSlot& o : pipe) othersInFlight = othersInFlight || (&o != &p && o.busy);
			// (nor does a submission whose caller has the library copy its rows: the host thread is the bottleneck there, 2 x 512 KB of
			// memcpy per buffer, and a second launch only adds to it -- 48-50 vs 52-59 us per buffer)
			if (in == nullptr && (othersInFlight || halfChainsUsed) && PrepareHalves(n))
			{
				// two free-running half-batch chains (see halfStream): each half in submission order on its own stream
				for (int h = 0; h < numChains; h++)
					if (!p.halfDone[h]) CheckHip(hipEventCreateWithFlags(&p.halfDone[h], hipEventDisableTiming), "hipEventCreate");
				LaunchHalves(dIn, dOut, n, (long)n, (long)n, p.halfDone, true);
				halfChainsUsed = true;
				pipelineUsed = true;
				lastKernelEvent = nullptr; // (ProcessDevice after this drains the half streams itself)
				lastKernelStream = nullptr;
				p.onOwnStream = false;
				p.onHalfStreams = true;
				p.busy = true;
				nextSlot = (nextSlot + 1) % kPipelineSlots;
				return ticket;
			}
			JoinHalves(); // back on the batch stream: the half chains first
			ProcessDeviceOn(stream, dIn, dOut, n, (long)n, (long)n);
			CheckHip(hipEventRecord(p.downloaded, stream), "hipEventRecord");
			p.onOwnStream = false;
			p.onHalfStreams = false;
			p.busy = true;
			nextSlot = (nextSlot + 1) % kPipelineSlots;
			return ticket;
		}
		// One launch per buffer (the usual case): the whole buffer -- upload, kernel, download -- rides on the slot's OWN stream, in order,
		// with no event between them; the only cross-stream edge is the stream state: this buffer's kernel waits for the previous
		// buffer's.  The upload of buffer k + 1 (its stream's first operation) overlaps the kernel of buffer k, the download of buffer k
		// (behind its kernel) overlaps the kernel of buffer k + 1.  Per buffer: 2 copies, 1 launch, 1 event wait, 1 event record --
		// the round-2 path cost 2 more waits and 2 more records on the compute stream, 15 us per buffer (tools/microbench/host_pipe_probe.cpp).
		if (LaunchUnitsAfterSwitch(nullptr, nullptr) <= 1)
		{
			JoinHalves(); // (device-pointer steps may have run as half-batch chains: this buffer's kernel comes after both)
			if (!p.own) CheckHip(hipStreamCreateWithFlags(&p.own, hipStreamNonBlocking), "hipStreamCreate");
			bool listsChanged = false, dirty = false;
			for (auto& g : groups) dirty = dirty || (g->NumActive() > 0 && g->ListsDirty());
			if (dirty)
			{
				// the index lists are re-uploaded on the batch stream: not before the kernels still reading the old ones are done
				if (lastKernelEvent && lastKernelStream != stream) CheckHip(hipStreamWaitEvent(stream, lastKernelEvent, 0), "hipStreamWaitEvent");
				for (auto& g : groups)
					if (g->NumActive() > 0) listsChanged = g->SyncActiveLists() || listsChanged;
			}
			if (listsChanged || submitTopology != topologyVersion || !pipelineUsed)
			{
				// everything the batch stream still has in flight for this batch (state resets of new streams, index lists) comes first
				if (!mainDone) CheckHip(hipEventCreateWithFlags(&mainDone, hipEventDisableTiming), "hipEventCreate");
				CheckHip(hipEventRecord(mainDone, stream), "hipEventRecord");
				CheckHip(hipStreamWaitEvent(p.own, mainDone, 0), "hipStreamWaitEvent");
				submitTopology = topologyVersion;
			}
			pipelineUsed = true;
			CheckHip(hipMemcpyAsync(p.dev, p.hostIn, total * sizeof(float), hipMemcpyHostToDevice, p.own), "hipMemcpyAsync H2D");
			if (lastKernelEvent && lastKernelStream != p.own) CheckHip(hipStreamWaitEvent(p.own, lastKernelEvent, 0), "hipStreamWaitEvent");
			ProcessDeviceOn(p.own, p.dev, p.dev, n, (long)n, (long)n);
			CheckHip(hipEventRecord(p.computed, p.own), "hipEventRecord");
			lastKernelEvent = p.computed;
			lastKernelStream = p.own;
			CheckHip(hipMemcpyAsync(p.hostOut, p.dev, total * sizeof(float), hipMemcpyDeviceToHost, p.own), "hipMemcpyAsync D2H");
			p.onOwnStream = true;
			p.onHalfStreams = false;
			p.busy = true;
			nextSlot = (nextSlot + 1) % kPipelineSlots;
			return ticket;
		}
		// several launch units per buffer (a captured hipGraph on the batch stream): copies on the copy streams, events in between
		if (!copyIn)
		{
			CheckHip(hipStreamCreateWithFlags(&copyIn, hipStreamNonBlocking), "hipStreamCreate");
			CheckHip(hipStreamCreateWithFlags(&copyOut, hipStreamNonBlocking), "hipStreamCreate");
		}
		CheckHip(hipMemcpyAsync(p.dev, p.hostIn, total * sizeof(float), hipMemcpyHostToDevice, copyIn), "hipMemcpyAsync H2D");
		CheckHip(hipEventRecord(p.uploaded, copyIn), "hipEventRecord");
		CheckHip(hipStreamWaitEvent(stream, p.uploaded, 0), "hipStreamWaitEvent");
		ProcessDeviceOrdered(p.dev, p.dev, n, (long)n, (long)n);
		CheckHip(hipEventRecord(p.computed, stream), "hipEventRecord");
		CheckHip(hipStreamWaitEvent(copyOut, p.computed, 0), "hipStreamWaitEvent");
		CheckHip(hipMemcpyAsync(p.hostOut, p.dev, total * sizeof(float), hipMemcpyDeviceToHost, copyOut), "hipMemcpyAsync D2H");
		CheckHip(hipEventRecord(p.downloaded, copyOut), "hipEventRecord");
		p.onOwnStream = false;
		p.onHalfStreams = false;
		p.busy = true;
		nextSlot = (nextSlot + 1) % kPipelineSlots;
		return ticket;
	}

	void GpuBatch::Collect(int ticket, float* out)
	{
		CheckUsable();
		if (ticket < 0 || ticket >= kPipelineSlots || !pipe[ticket].busy) throw std::runtime_error("neuralaudio_amd: Collect with an invalid ticket");
		PipeSlot& p = pipe[ticket];
		if (p.onHalfStreams)
		{
			for (int h = 0; h < numChains; h++) WaitEventBounded(p.halfDone[h], "hipEventSynchronize");
		}
		else if (p.onOwnStream) WaitStreamBounded(p.own, "hipStreamSynchronize"); // the download is the stream's last operation
		else WaitEventBounded(p.downloaded, "hipEventSynchronize");
		// the slot holds the rows the batch had at Submit: ids retired since then are zeroed only inside that block
		if (!retired.empty()) ZeroRetiredRows(p.hostOut, p.n, p.rows);
		if (out) memcpy(out, p.hostOut, p.rows * p.n * sizeof(float)); // nullptr: the caller reads OutputView() in place
		p.busy = false;
	}

	float* GpuBatch::NextInput(size_t n)
	{
		CheckUsable();
		if (n == 0 || streams.empty()) throw std::runtime_error("neuralaudio_amd: NextInput on an empty batch / buffer");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		PipeSlot& p = pipe[nextSlot];
		if (p.busy) throw std::runtime_error("neuralaudio_amd: NextInput with every pipeline slot in flight (Collect the oldest ticket first)");
		EnsurePipeSlot(p, streams.size() * n);
		return p.hostIn;
	}

	const float* GpuBatch::OutputView(int ticket) const
	{
		if (ticket < 0 || ticket >= kPipelineSlots) throw std::runtime_error("neuralaudio_amd: OutputView with an invalid ticket");
		return pipe[ticket].hostOut;
	}

}
