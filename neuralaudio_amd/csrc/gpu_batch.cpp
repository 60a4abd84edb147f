// gpu_batch.cpp -- see gpu_batch.h.  Host side only: device memory, tables, launches.
#include "gpu_batch.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <hip/hip_runtime.h>

#include "gpu_batch_internal.h"

namespace na
{
	void CheckHip(hipError_t e, const char* what)
	{
		if (e != hipSuccess) throw HipError(e, what);
	}

	int VisibleDeviceCount()
	{
		int count = 0;
		const hipError_t e = hipGetDeviceCount(&count);
		if (e != hipSuccess)
		{
			(void)hipGetLastError();
			return 0;
		}
		return count;
	}

}

namespace na
{
	// Host side only (no device): which kernel family a batch would pick for `streams` streams of this submodel -- the constructor logic
	// of WaveNetGroup (pack / pad / range proof / FamilyFor) -- and the facts of the range proof behind the choice.
	ModelKernelInfo PredictModelKernel(const LoadedModel& model, float quality, int streams)
	{
		ModelKernelInfo info;
		if (model.subModels.empty()) return info;
		const int idx = model.isComposite ? model.ModelIndexFromQuality(quality) : 0;
		const ModelDesc& d = *model.subModels[(size_t)idx].desc;
		if (d.kind != MODEL_WAVENET)
		{
			info.kernel = "recurrent";
			return info;
		}
		const WaveNetPlan real = BuildWaveNetPlan(d.wavenet, true);
		info.inputLimit = real.condLimit;
		info.rangeProven = real.splitRangeProven;
		info.weightsOk = real.splitWeightsOk;
		const int packHint = (!model.isComposite && model.subModels.size() == 1) ? streams : 0;
		const int pack = WaveNetGroup::PackFor(d.wavenet, packHint);
		const bool isVirtual = pack > 1 || WaveNetGroup::PadFor(d.wavenet);
		const WnFamily fam = isVirtual ? WN_FAMILY_SPLIT : FamilyFor(real);
		info.pack = pack;
		info.kernel = fam == WN_FAMILY_SPLIT ? "f16-split" : (fam == WN_FAMILY_GENERIC ? "generic" : "frame");
		if (fam != WN_FAMILY_SPLIT) info.inputLimit = INFINITY;
		return info;
	}

	// Cost of one stream for the multi-GPU sharder: time = per-launch skeleton + bytes (WaveNet) / multiply-accumulates (recurrent),
	// two-point fits per kernel family to the round-3 measurements (us per 1024 streams x 128 frames): specialised / split chains
	// 22.3 + 0.0143 B (Standard 41.6, Lite 33.7), f32 frame kernel 31.6 + 0.0164 B (A2-Lite 46.6, A2-Full 71.4), LDS-free recurrent
	// kernel 8.4 + 0.0107 MAC (LSTM 1x16 20.2, 2x16 42.1), runtime-shaped kernels by their measured per-block times.
	double EstimateStreamCost(const LoadedModel& model, float quality)
	{
		if (model.subModels.empty()) return 1.0;
		const int idx = model.isComposite ? model.ModelIndexFromQuality(quality) : 0;
		const ModelDesc& d = *model.subModels[(size_t)idx].desc;
		if (d.kind == MODEL_WAVENET)
		{
			const WaveNetPlan plan = BuildWaveNetPlan(d.wavenet, true);
			const double B = plan.AlgorithmicBytesPerSample(WN_MAX_FRAMES);
			if (plan.genericOnly) return 1000.0 * plan.maxChannels / 32.0; // 1.0 ms per block at 32 channels (<= 256 streams)
			const bool composite = model.isComposite;
			const bool split = plan.splitFastT == 2 || (!composite && (WaveNetPackFactor(d.wavenet) > 1 || WaveNetWantsPadding(d.wavenet)));
			return split ? 22.3 + 0.0143 * B : 31.6 + 0.0164 * B;
		}
		const LSTMDesc& l = d.lstm;
		const double gates = (l.cell == CELL_GRU) ? 3.0 : 4.0;
		double macs = 0.0;
		for (int k = 0; k < l.numLayers; k++) macs += gates * l.hiddenSize * ((k == 0 ? 1 : l.hiddenSize) + l.hiddenSize);
		macs += l.hiddenSize;
		const bool dpp = l.tail.empty() && l.numLayers <= 2 && (l.hiddenSize <= 16 || (l.numLayers == 1 && l.hiddenSize <= 32));
		return dpp ? 8.4 + 0.0107 * macs : 60.0 + 0.06 * macs;
	}

	// ------------------------------------------------------------------------------------------ GpuBatch

	GpuBatch::GpuBatch(int dev, hipStream_t borrowedStream) : device(dev)
	{
		const int count = VisibleDeviceCount();
		if (count <= 0) throw std::runtime_error("neuralaudio_amd: no HIP device is visible; this library has no CPU fallback");
		if (dev < 0 || dev >= count) throw std::runtime_error("neuralaudio_amd: invalid HIP device index");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		if (borrowedStream)
		{
			stream = borrowedStream;
			ownsStream = false;
			streamObserved = true; // the caller orders its own work on it
		}
		else
		{
			CheckHip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
		}
		numChains = std::min(std::max(Tuning::Get().hostChains, 2), kMaxChains);
	}

	GpuBatch::~GpuBatch()
	{
		(void)hipSetDevice(device);
		// everything this batch has in flight, under the wait limit (gpu_batch.h "bounded waits").  A device that does not come back keeps
		// the allocations: a kernel that is still running may write them, and hipFree would wait for it without a limit.
		bool idle = true;
		try
		{
			if (broken)
			{
				// (the resident launch of a broken batch: asked to leave, not waited for command by command)
				if (residentState && residentState->ctrl) __atomic_store_n(&residentState->ctrl->exitAfter, 0ull, __ATOMIC_RELEASE);
			}
			else DrainResident();
			for (PipeSlot& p : pipe)
				if (p.own) WaitStreamBounded(p.own, "closing the batch");
			for (hipStream_t hs : halfStream)
				if (hs) WaitStreamBounded(hs, "closing the batch");
			if (stream) WaitStreamBounded(stream, "closing the batch");
		}
		catch (...)
		{
			idle = false;
		}
		if (!idle)
		{
			for (auto& g : groups) (void)g.release();
			(void)residentState.release();
			for (WnLaunchTable& t : wnTable) t.entries.clear();
			return;
		}
		residentState.reset();
		groups.clear();
		if (hostStage) (void)hipHostFree(hostStage);
		if (devStage) (void)hipFree(devStage);
		for (PipeSlot& p : pipe)
		{
			if (p.hostIn) (void)hipHostFree(p.hostIn);
			if (p.hostOut) (void)hipHostFree(p.hostOut);
			if (p.dev) (void)hipFree(p.dev);
			if (p.uploaded) (void)hipEventDestroy(p.uploaded);
			if (p.computed) (void)hipEventDestroy(p.computed);
			if (p.downloaded) (void)hipEventDestroy(p.downloaded);
			for (hipEvent_t e : p.halfDone)
				if (e) (void)hipEventDestroy(e);
			if (p.own) (void)hipStreamDestroy(p.own);
		}
		if (mainDone) (void)hipEventDestroy(mainDone);
		for (hipStream_t hs : halfStream)
			if (hs) (void)hipStreamDestroy(hs);
		for (auto& m : marks)
			for (hipEvent_t e : m)
				if (e) (void)hipEventDestroy(e);

		if (copyIn) (void)hipStreamDestroy(copyIn);
		if (copyOut) (void)hipStreamDestroy(copyOut);
		for (auto& e : graphCache) (void)hipGraphExecDestroy(e.exec);
		if (forkEvent) (void)hipEventDestroy(forkEvent);
		if (stream && ownsStream) (void)hipStreamDestroy(stream);
	}

	ModelGroup* GpuBatch::GroupFor(const std::shared_ptr<const ModelDesc>& desc, int packHint)
	{
		for (auto& g : groups)
			if (g->desc.get() == desc.get()) return g.get();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		std::unique_ptr<ModelGroup> g;
		if (desc->kind == MODEL_WAVENET) g.reset(new WaveNetGroup(desc, stream, packHint, peerWeights));
		else if (desc->kind == MODEL_LSTM) g.reset(new LstmGroup(desc, stream, peerWeights));
		else throw std::runtime_error("neuralaudio_amd: unsupported model kind");
		if (peerWeights) awaitingWeights.push_back(g.get());
		groups.push_back(std::move(g));
		return groups.back().get();
	}

	int GpuBatch::AddStream(const std::shared_ptr<const LoadedModel>& model, float quality, bool prewarm, bool onDemand)
	{
		return AddStreams(model, quality, 1, prewarm, onDemand);
	}

	// ids for `count` new streams: retired ones first (the lowest for a single stream, a run of consecutive ones for several), else new rows
	int GpuBatch::AllocateIds(int count)
	{
		for (size_t i = 0; i + (size_t)count <= retired.size(); i++)
		{
			if (retired[i + (size_t)count - 1] == retired[i] + count - 1)
			{
				const int first = retired[i];
				retired.erase(retired.begin() + (long)i, retired.begin() + (long)i + count);
				return first;
			}
		}
		const int first = (int)streams.size();
		streams.resize(streams.size() + (size_t)count);
		for (int i = 0; i < count; i++) streams[(size_t)(first + i)].live = false;
		return first;
	}

	int GpuBatch::AddStreams(const std::shared_ptr<const LoadedModel>& model, float quality, int count, bool prewarm, bool onDemand)
	{
		CheckUsable();
		if (!model || model->subModels.empty()) throw std::runtime_error("neuralaudio_amd: AddStream with an empty model");
		if (count < 1) throw std::runtime_error("neuralaudio_amd: AddStreams with count < 1");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline(); // (state arrays may be re-allocated below)
		topologyVersion++;
		const int active = model->isComposite ? model->ModelIndexFromQuality(quality) : 0;
		const size_t numSub = model->subModels.size();
		std::vector<ModelGroup*> subGroups(numSub);
		std::vector<std::vector<int>> newMembers(numSub);
		for (size_t k = 0; k < numSub; k++) subGroups[k] = GroupFor(model->subModels[k].desc, (!model->isComposite && numSub == 1) ? count : 0);
		const int first = AllocateIds(count);
		// Everything below may throw (hipMalloc inside AddMember / Reset / Prewarm).  A failed call must leave the batch as it was: the
		// members it created are removed again, the ids go back to `retired` (rows appended by AllocateIds leave the arrays).
		int built = 0; // rows [first, first + built) are complete StreamRefs
		std::vector<std::pair<ModelGroup*, int>> partial; // members of the row under construction
		const size_t pendingBefore = pendingPrewarm.size(); // (deferred prewarms this call files: dropped again if it fails)
		try
		{
			for (int i = 0; i < count; i++)
			{
				StreamRef ref;
				ref.model = model;
				ref.quality = quality;
				ref.active = active;
				ref.onDemand = onDemand;
				ref.live = true;
				ref.prewarmed.assign(numSub, 0);
				const int row = first + i;
				partial.clear();
				for (size_t k = 0; k < numSub; k++)
				{
					const int member = subGroups[k]->AddMember();
					partial.push_back({ subGroups[k], member });
					newMembers[k].push_back(member);
				}
				ref.members = partial;
				partial.clear();
				ref.members[(size_t)active].first->SetActive(ref.members[(size_t)active].second, row);
				streams[(size_t)row] = ref;
				built = i + 1;
			}
			// fresh state for every new member, then prewarm: every submodel (LoadAll, CompositeModel.h:111-118) or only the active one
			// (OnDemand, :104-109 -- the others are prewarmed when a quality change first selects them, :52-60)
			for (size_t k = 0; k < numSub; k++)
			{
				std::sort(newMembers[k].begin(), newMembers[k].end());
				subGroups[k]->Reset(newMembers[k]);
				const bool now = prewarm && (!onDemand || (int)k == active);
				const bool waits = now && std::find(awaitingWeights.begin(), awaitingWeights.end(), subGroups[k]) != awaitingWeights.end();
				if (waits) pendingPrewarm.push_back({ subGroups[k], newMembers[k] }); // (its weights are not on the device yet: WeightsArrived)
				else if (now) subGroups[k]->Prewarm(newMembers[k]);
				for (int i = 0; i < count; i++) streams[(size_t)(first + i)].prewarmed[k] = now ? 1 : 0;
			}
		}
		catch (...)
		{
			pendingPrewarm.resize(pendingBefore); // (WeightsArrived must not prewarm member slots that are gone or recycled)
			for (auto& gm : partial) gm.first->RemoveMember(gm.second);
			for (int i = 0; i < count; i++)
			{
				StreamRef& ref = streams[(size_t)(first + i)];
				if (i < built)
					for (auto& gm : ref.members) gm.first->RemoveMember(gm.second);
				ref = StreamRef();
				ref.live = false;
				retired.insert(std::lower_bound(retired.begin(), retired.end(), first + i), first + i);
			}
			DropTrailingRetiredRows();
			throw;
		}
		// the half-batch chains' streams on this, the set-up side (creating a HIP stream takes ~13 ms: not inside the first buffer)
		if (ownsStream && !streamObserved && streams.size() >= 512)
			for (int h = 0; h < numChains; h++)
				if (!halfStream[h]) CheckHip(hipStreamCreateWithFlags(&halfStream[h], hipStreamNonBlocking), "hipStreamCreate");
		return first;
	}

	// trailing retired rows leave the [streams][n] arrays altogether (their ids are the largest entries of the sorted `retired` list)
	void GpuBatch::DropTrailingRetiredRows()
	{
		while (!streams.empty() && !streams.back().live)
		{
			if (!retired.empty() && retired.back() == (int)streams.size() - 1) retired.pop_back();
			streams.pop_back();
		}
	}

	hipStream_t GpuBatch::GetStream()
	{
		if (!streamObserved)
		{
			(void)hipSetDevice(device);
			JoinHalves();
			streamObserved = true;
		}
		return stream;
	}

	void GpuBatch::RemoveStreams(int first, int count)
	{
		CheckUsable();
		if (count < 1 || first < 0 || (size_t)first + (size_t)count > streams.size()) throw std::runtime_error("neuralaudio_amd: RemoveStreams: id range outside the batch");
		for (int i = 0; i < count; i++)
			if (!streams[(size_t)(first + i)].live) throw std::runtime_error("neuralaudio_amd: RemoveStreams: stream was already removed");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		// the slots may be handed out again right away: nothing of theirs may still be in flight
		Quiesce();
		topologyVersion++;
		for (int i = 0; i < count; i++)
		{
			StreamRef& ref = streams[(size_t)(first + i)];
			for (auto& gm : ref.members) gm.first->RemoveMember(gm.second);
			ref = StreamRef();
			ref.live = false;
			retired.insert(std::lower_bound(retired.begin(), retired.end(), first + i), first + i);
		}
		DropTrailingRetiredRows();
	}

	unsigned GpuBatch::StreamPrewarmedMask(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		unsigned mask = 0;
		for (size_t k = 0; k < ref.prewarmed.size() && k < 32; k++) mask |= ref.prewarmed[k] ? (1u << k) : 0u;
		return mask;
	}

	void GpuBatch::ZeroRetiredRows(float* hostRows, size_t n, size_t rows) const
	{
		for (int id : retired)
			if ((size_t)id < rows) memset(hostRows + (size_t)id * n, 0, n * sizeof(float));
	}

	void GpuBatch::SetQuality(int s, float quality)
	{
		CheckUsable();
		StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) throw std::runtime_error("neuralaudio_amd: stream was removed");
		ref.quality = quality;
		if (!ref.model->isComposite) return;
		const int idx = ref.model->ModelIndexFromQuality(quality);
		if (idx == ref.active) return;
		ref.members[(size_t)ref.active].first->SetActive(ref.members[(size_t)ref.active].second, -1);
		ref.active = idx;
		ref.members[(size_t)idx].first->SetActive(ref.members[(size_t)idx].second, s);
		topologyVersion++;
		if (ref.onDemand && !ref.prewarmed[(size_t)idx])
		{
			// CompositeModel::SetCurrentModelIndex (CompositeModel.h:52-60): first use of this submodel -- NOT real-time safe, which
			// IsQualityChangeRealtimeSafe() reports beforehand
			Quiesce(); // (the prewarm runs on the batch stream: behind a resident launch it would wait for that to idle out)
			ref.members[(size_t)idx].first->Prewarm({ ref.members[(size_t)idx].second });
			ref.prewarmed[(size_t)idx] = 1;
		}
	}

	// How many launches a buffer takes once `leaving` has lost / `entering` has gained an active stream (ProcessDevice's grouping: the
	// WaveNet launch lists, the fused recurrent launch, everything else on its own)
	int GpuBatch::LaunchUnitsAfterSwitch(const ModelGroup* leaving, const ModelGroup* entering) const
	{
		bool lists[3] = { false, false, false }, joiner = false, rec = false;
		int singles = 0;
		for (const auto& g : groups)
		{
			int active = g->NumActive();
			if (g.get() == leaving) active -= 1;
			if (g.get() == entering) active += 1;
			if (active <= 0) continue;
			const int c = g->LaunchClass();
			if (c >= 0 && c <= 2) lists[c] = true;
			else if (c == -1) joiner = true;
			else if (c == 3) rec = true;
			else singles++;
		}
		if (joiner && !lists[2]) lists[1] = true; // plain fast-flavour split groups ride in the packed launch when there is one
		return (int)lists[0] + (int)lists[1] + (int)lists[2] + (int)rec + singles;
	}

	// CompositeModel::IsModelChangeRealtimeSafe (CompositeModel.h:44-50, HadInitialPrewarm): false when the target submodel never had its
	// prewarm -- the switch would prewarm it (OnDemand) or run it cold (a stream added without prewarm) -- and false when the batch
	// would take several launches per buffer afterwards: those run as a captured hipGraph, which a switch re-captures.  Otherwise a
	// switch only re-uploads two pinned index lists asynchronously.
	bool GpuBatch::IsQualityChangeRealtimeSafe(int s, float quality) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return false;
		if (!ref.model->isComposite) return true;
		const int idx = ref.model->ModelIndexFromQuality(quality);
		if (idx == ref.active) return true;
		if (!ref.prewarmed[(size_t)idx]) return false;
		return LaunchUnitsAfterSwitch(ref.members[(size_t)ref.active].first, ref.members[(size_t)idx].first) <= 1;
	}

	float GpuBatch::GetQuality(int s) const { return streams.at((size_t)s).quality; }
	int GpuBatch::GetActiveSubModel(int s) const { return streams.at((size_t)s).active; }

	void GpuBatch::Prewarm(int s)
	{
		CheckUsable();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline();
		StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return;
		// LoadAll: every submodel is prewarmed (CompositeModel.h:111-118); OnDemand: the current one (:104-109)
		for (size_t k = 0; k < ref.members.size(); k++)
		{
			if (ref.onDemand && (int)k != ref.active) continue;
			ref.members[k].first->Prewarm({ ref.members[k].second });
			ref.prewarmed[k] = 1;
		}
	}

	void GpuBatch::ProcessDevice(const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		CheckUsable();
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		// A batch on its own stream that nobody has seen: a buffer of one contiguous WaveNet group runs as two free-running half-batch
		// launches (the order of work on the internal streams is not observable from outside; Synchronize() and the host-buffer entry
		// points wait for all of them).  1024 x A1 Standard x 128 frames: 40.1 -> 37.4 us per step.
		if (ownsStream && !streamObserved)
		{
			if (TryResident(dIn, dOut, n, inStride, outStride)) return;
			if (PrepareHalves(n))
			{
				LaunchHalves(dIn, dOut, n, inStride, outStride, nullptr, false);
				return;
			}
		}
		ProcessDeviceOrdered(dIn, dOut, n, inStride, outStride);
	}

	// on the batch stream, behind everything launched so far (the host-buffer entry points: their copies are on that stream; a lone
	// blocking buffer gains nothing from two half launches -- 57 vs 60 us)
	void GpuBatch::ProcessDeviceOrdered(const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		lastStepHalves = false;
		lastStepResident = false;
		JoinHalves(); // earlier buffers ran on the resident launch / as two half-batch chains: this call's kernels come after them
		if (pipelineUsed)
		{
			// buffers submitted through the pipelined interface run on per-slot streams: this call's kernels come after theirs ...
			if (lastKernelStream != stream && lastKernelEvent) CheckHip(hipStreamWaitEvent(stream, lastKernelEvent, 0), "hipStreamWaitEvent");
		}
		ProcessDeviceOn(stream, dIn, dOut, n, inStride, outStride);
		if (pipelineUsed)
		{
			// ... and the next submitted buffer after this call's
			if (!mainDone) CheckHip(hipEventCreateWithFlags(&mainDone, hipEventDisableTiming), "hipEventCreate");
			CheckHip(hipEventRecord(mainDone, stream), "hipEventRecord");
			lastKernelEvent = mainDone;
			lastKernelStream = stream;
		}
	}

	// `launch` != the batch stream is only used for a batch that runs as ONE launch per buffer (Submit checks)
	void GpuBatch::ProcessDeviceOn(hipStream_t launch, const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		int activeGroups = 0;
		for (auto& g : groups) activeGroups += (g->NumActive() > 0);
		if (activeGroups <= 1)
		{
			for (auto& g : groups) g->Process(dIn, dOut, inStride, outStride, n, launch);
			return;
		}
		// Mixed batch.  Groups that can share a launch are fused: all WaveNet groups on the frame kernel into one launch, all LSTM / GRU
		// groups with an LDS-free kernel instance into another (the workgroups of all architectures share the chip, no fork/join per
		// group).  What remains are independent "units" (disjoint rows, disjoint state); one unit runs directly on the batch stream.
		constexpr int NUM_WN_LISTS = 3; // frame kernel | f16-split kernel | f16-split kernel, packed streams: one launch each
		std::vector<WnFrameGroup> fusedWn[NUM_WN_LISTS];
		std::vector<RecurrentGroup> fusedRec;
		std::vector<ModelGroup*> singles;
		ModelGroup* wnOwner[NUM_WN_LISTS] = {}; // lends its side stream / event to the fused unit
		std::vector<std::pair<WnFrameGroup, ModelGroup*>> joiners;
		ModelGroup* recOwner = nullptr;
		for (auto& g : groups)
		{
			if (g->NumActive() == 0) continue;
			WnFrameGroup a = {};
			RecurrentGroup r;
			int list = 0;
			if (g->FusedLaunchArgs(a, list))
			{
				if (list < 0)
				{
					// a plain fast-flavour split group: joins the packed launch if there is one (decided below), else the plain split launch
					a.slots = static_cast<WaveNetGroup*>(g.get())->listSlots;
					joiners.push_back({ a, g.get() });
					continue;
				}
				fusedWn[list].push_back(a);
				if (!wnOwner[list]) wnOwner[list] = g.get();
			}
			else if (g->FusedRecurrentArgs(r))
			{
				fusedRec.push_back(r);
				if (!recOwner) recOwner = g.get();
			}
			else
			{
				g->SyncActiveLists();
				singles.push_back(g.get());
			}
		}
		for (auto& j : joiners)
		{
			const int list = fusedWn[2].empty() ? 1 : 2;
			WnFrameGroup a = j.first;
			if (list == 1 && static_cast<WaveNetGroup*>(j.second)->IsContiguous()) a.slots = nullptr; // the plain kernel's shortcut
			fusedWn[list].push_back(a);
			if (!wnOwner[list]) wnOwner[list] = j.second;
		}
		// prepareOnly: upload the group tables the list's launches will look up and launch nothing (the pass in front of a stream capture)
		auto launchWnList = [&](int which, hipStream_t s, bool prepareOnly) {
			const std::vector<WnFrameGroup>& list = fusedWn[which];
			bool compact = false;
			for (const WnFrameGroup& g : list) compact = compact || g.model->compact_rings != 0;
			size_t offset = 0, left = n;
			while (left > 0)
			{
				const int chunk = NextWaveNetChunk(left, compact);
				// more groups than one launch's kernarg table holds (a batch of many different models): ONE launch with the table in device
				// memory where the chains have one (128-frame blocks of the A1 families), else launches of eight groups each
				if (which != 0 && list.size() > (size_t)WN_FRAME_MAX_GROUPS)
				{
					wnTable[which].prepareOnly = prepareOnly;
					const hipError_t te = LaunchWaveNetSpecTable(list.data(), (int)list.size(), dIn + offset, dOut + offset, inStride, outStride, chunk, s, wnTable[which]);
					wnTable[which].prepareOnly = false;
					if (te == hipSuccess)
					{
						offset += (size_t)chunk;
						left -= (size_t)chunk;
						continue;
					}
					if (te != hipErrorNotSupported) CheckHip(te, "WaveNet kernel (table launch)");
					(void)hipGetLastError();
				}
				if (prepareOnly)
				{
					offset += (size_t)chunk;
					left -= (size_t)chunk;
					continue;
				}
				for (size_t first = 0; first < list.size(); first += WN_FRAME_MAX_GROUPS)
				{
					const int count = (int)std::min<size_t>(list.size() - first, (size_t)WN_FRAME_MAX_GROUPS);
					CheckHip(which == 0 ? LaunchWaveNetFrameFused(list.data() + first, count, dIn + offset, dOut + offset, inStride, outStride, chunk, s)
										: LaunchWaveNetSplitFused(list.data() + first, count, dIn + offset, dOut + offset, inStride, outStride, chunk, s,
											1 | ((!Tuning::Get().wnNtOff && StateBytes() > ((size_t)Tuning::Get().wnNtFromMB << 20)) ? WN_SHARING_BEYOND_CACHE : 0)),
						"WaveNet kernel (fused)");
				}
				offset += (size_t)chunk;
				left -= (size_t)chunk;
			}
		};
		auto launchRec = [&](hipStream_t s, bool prepareOnly) {
			size_t offset = 0, left = n;
			while (left > 0)
			{
				const int chunk = (int)std::min<size_t>(left, (size_t)LSTM_MAX_FRAMES);
				if (fusedRec.size() > (size_t)RECURRENT_MAX_GROUPS)
				{
					// (many different recurrent models: one launch, the group table in device memory)
					wnTable[3].prepareOnly = prepareOnly;
					const hipError_t te = LaunchRecurrentDppTable(fusedRec.data(), (int)fusedRec.size(), dIn + offset, dOut + offset, inStride, outStride, chunk, s, wnTable[3]);
					wnTable[3].prepareOnly = false;
					CheckHip(te, "RecurrentDppKernel (table launch)");
					offset += (size_t)chunk;
					left -= (size_t)chunk;
					continue;
				}
				if (prepareOnly) break;
				for (size_t first = 0; first < fusedRec.size(); first += RECURRENT_MAX_GROUPS)
					CheckHip(LaunchRecurrentDpp(fusedRec.data() + first, (int)std::min<size_t>(fusedRec.size() - first, (size_t)RECURRENT_MAX_GROUPS),
						dIn + offset, dOut + offset, inStride, outStride, chunk, s), "RecurrentDppKernel (fused)");
				offset += (size_t)chunk;
				left -= (size_t)chunk;
			}
		};
		// group tables of an earlier topology go (their graphs first)
		if (!graphCache.empty() && graphCache.front().key.version != topologyVersion)
		{
			for (auto& e : graphCache) (void)hipGraphExecDestroy(e.exec);
			graphCache.clear();
		}
		for (WnLaunchTable& t : wnTable) t.NewGeneration(topologyVersion);
		size_t units = (fusedRec.empty() ? 0 : 1) + singles.size();
		for (int l = 0; l < NUM_WN_LISTS; l++) units += fusedWn[l].empty() ? 0 : 1;
		if (units == 1)
		{
			for (int l = 0; l < NUM_WN_LISTS; l++)
				if (!fusedWn[l].empty())
				{
					launchWnList(l, launch, false);
					return;
				}
			if (!fusedRec.empty()) launchRec(launch, false);
			else singles[0]->Process(dIn, dOut, inStride, outStride, n, launch);
			return;
		}
		{
			// tuning knob: the units one after the other on the batch stream instead of concurrently on side streams
			const bool serial = Tuning::Get().batchSerial;
			if (serial)
			{
				for (int l = 0; l < NUM_WN_LISTS; l++)
					if (!fusedWn[l].empty()) launchWnList(l, stream, false);
				if (!fusedRec.empty()) launchRec(stream, false);
				for (ModelGroup* g : singles) g->Process(dIn, dOut, inStride, outStride, n, stream);
				return;
			}
		}
		// Several units: fork onto side streams so their kernels share the GPU, then join back into the batch stream.  The fork/join
		// costs ~5 HIP calls per unit, which would make a buffer host-bound, so the sequence is captured once into a hipGraph and
		// replayed while the call signature (pointers, n, strides) and the active-stream lists stay the same -- the steady state of a
		// real-time host.
		// The runtime this process actually runs on may be OLDER than the ROCm 7.2 this library is built against: a host that loads
		// PyTorch first gets PyTorch's bundled libamdhip64 (HIP 7.0.51831 with torch 2.10+rocm7.0) for the whole process, and that
		// runtime's graph replay crashes after other graphs of the process were destroyed (hip::Graph::UpdateStreams under
		// hipGraphLaunch; seen in tests/test_gpu_multi.py behind any other test file, never on 7.2: profiles/r06_gputest_timing.txt).  On a
		// runtime older than the one it was built for the sequence is therefore issued directly, every buffer: ~5 HIP calls per launch
		// unit of host time, the same streams, events and results.
		static const bool graphsTrusted = [] {
			int v = 0;
			return hipRuntimeGetVersion(&v) == hipSuccess && v >= 70200000 && !Tuning::Get().batchNoGraph;
		}();
		if (!graphsTrusted)
		{
			if (!forkEvent) CheckHip(hipEventCreateWithFlags(&forkEvent, hipEventDisableTiming), "hipEventCreate");
			CheckHip(hipEventRecord(forkEvent, stream), "hipEventRecord");
			auto direct = [&](ModelGroup* owner, const std::function<void(hipStream_t)>& work) {
				hipStream_t side = owner->SideStream();
				CheckHip(hipStreamWaitEvent(side, forkEvent, 0), "hipStreamWaitEvent");
				work(side);
				CheckHip(hipEventRecord(owner->DoneEvent(), side), "hipEventRecord");
				CheckHip(hipStreamWaitEvent(stream, owner->DoneEvent(), 0), "hipStreamWaitEvent");
			};
			for (int l = 0; l < NUM_WN_LISTS; l++)
				if (!fusedWn[l].empty()) direct(wnOwner[l], [&, l](hipStream_t s) { launchWnList(l, s, false); });
			if (!fusedRec.empty()) direct(recOwner, [&](hipStream_t s) { launchRec(s, false); });
			for (ModelGroup* g : singles) direct(g, [&](hipStream_t s) { g->Process(dIn, dOut, inStride, outStride, n, s); });
			return;
		}
		hipGraphExec_t graphExec = nullptr;
		for (auto& e : graphCache)
			if (e.key.dIn == dIn && e.key.dOut == dOut && e.key.n == n && e.key.inStride == inStride && e.key.outStride == outStride) graphExec = e.exec;
		if (!graphExec)
		{
			if (graphCache.size() >= 16)
			{
				(void)hipGraphExecDestroy(graphCache.front().exec);
				graphCache.erase(graphCache.begin());
			}
			// the group tables of the table launches are uploaded here, in front of the capture (WnLaunchTable)
			for (int l = 0; l < NUM_WN_LISTS; l++)
				if (!fusedWn[l].empty()) launchWnList(l, stream, true);
			if (!fusedRec.empty()) launchRec(stream, true);
			hipGraph_t graph = nullptr;
			CheckHip(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture");
			try
			{
				if (!forkEvent) CheckHip(hipEventCreateWithFlags(&forkEvent, hipEventDisableTiming), "hipEventCreate");
				CheckHip(hipEventRecord(forkEvent, stream), "hipEventRecord");
				auto branch = [&](ModelGroup* owner, const std::function<void(hipStream_t)>& work) {
					hipStream_t side = owner->SideStream();
					CheckHip(hipStreamWaitEvent(side, forkEvent, 0), "hipStreamWaitEvent");
					work(side);
					CheckHip(hipEventRecord(owner->DoneEvent(), side), "hipEventRecord");
					CheckHip(hipStreamWaitEvent(stream, owner->DoneEvent(), 0), "hipStreamWaitEvent");
				};
				for (int l = 0; l < NUM_WN_LISTS; l++)
					if (!fusedWn[l].empty()) branch(wnOwner[l], [&, l](hipStream_t s) { launchWnList(l, s, false); });
				if (!fusedRec.empty()) branch(recOwner, [&](hipStream_t s) { launchRec(s, false); });
				for (ModelGroup* g : singles) branch(g, [&](hipStream_t s) { g->Process(dIn, dOut, inStride, outStride, n, s); });
			}
			catch (...)
			{
				(void)hipStreamEndCapture(stream, &graph);
				if (graph) (void)hipGraphDestroy(graph);
				throw;
			}
			CheckHip(hipStreamEndCapture(stream, &graph), "hipStreamEndCapture");
			const hipError_t e = hipGraphInstantiate(&graphExec, graph, nullptr, nullptr, 0);
			(void)hipGraphDestroy(graph);
			CheckHip(e, "hipGraphInstantiate");
			graphCache.push_back({ { dIn, dOut, n, inStride, outStride, topologyVersion }, graphExec });
		}
		CheckHip(hipGraphLaunch(graphExec, stream), "hipGraphLaunch");
	}

	void GpuBatch::WeightImages(const LoadedModel& model, std::vector<std::pair<void*, size_t>>& out) const
	{
		for (const auto& sub : model.subModels)
			for (const auto& g : groups)
				if (g->desc.get() == sub.desc.get()) g->WeightImages(out);
	}

	void GpuBatch::WeightsArrived()
	{
		CheckUsable();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		for (ModelGroup* g : awaitingWeights) g->WeightsArrived();
		awaitingWeights.clear();
		for (auto& pp : pendingPrewarm) pp.first->Prewarm(pp.second);
		pendingPrewarm.clear();
		peerWeights = false;
	}

	void GpuBatch::Synchronize()
	{
		Quiesce();
	}

	double GpuBatch::AlgorithmicBytesPerSample(int blockFrames) const
	{
		double sum = 0.0;
		int n = 0;
		for (const auto& g : groups)
		{
			sum += g->AlgorithmicBytesPerSample(blockFrames) * g->NumActive();
			n += g->NumActive();
		}
		return n ? sum / n : 0.0;
	}

	double GpuBatch::MacsPerSample() const
	{
		double sum = 0.0;
		int n = 0;
		for (const auto& g : groups)
		{
			sum += g->MacsPerSample() * g->NumActive();
			n += g->NumActive();
		}
		return n ? sum / n : 0.0;
	}

	const char* GpuBatch::StreamKernelName(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return "";
		return ref.members[(size_t)ref.active].first->KernelName();
	}

	float GpuBatch::StreamInputLimit(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return 0.0f;
		return ref.members[(size_t)ref.active].first->InputLimit();
	}

	int GpuBatch::StreamRangeEvents(int s)
	{
		CheckUsable();
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return 0;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline();
		int total = 0;
		for (const auto& gm : ref.members) total += gm.first->RangeEvents(gm.second);
		return total;
	}

	int GpuBatch::StreamPackFactor(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return 0;
		return ref.members[(size_t)ref.active].first->PackFactor();
	}

	size_t GpuBatch::StateBytes() const
	{
		size_t total = 0;
		for (const auto& g : groups) total += g->StateBytesPerStream() * (size_t)g->NumInUse();
		return total;
	}
}
