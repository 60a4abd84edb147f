// gpu_batch_internal.h -- declarations shared by the translation units of class GpuBatch (gpu_batch.cpp: lifecycle and launch
// dispatch; gpu_batch_chains.cpp: half-batch chains, resident launch, timing marks; gpu_batch_host.cpp: host-buffer entry points)
#pragma once

#include "gpu_groups.h"

namespace na
{
	struct GpuBatch::HalfLists
	{
		std::vector<WnFrameGroup> part[GpuBatch::kMaxChains];
		bool listsUploaded = false; // an index list went to the device on the batch stream while the lists were built
		bool compact = false;       // some group has compact rings: chunk lengths of WnCompactSafeFrames() only
		// every part is a contiguous range of rows (no index lists, no packed streams): the host can stage and collect a half by itself
		bool RowRangesOnly() const
		{
			for (const auto& list : part)
				for (const WnFrameGroup& g : list)
					if (g.slots != nullptr || g.pack > 1) return false;
			return true;
		}
	};

	// the resident launch of a batch (gpu_batch_chains.cpp): command ring, counters, the launch list it was started with
	struct GpuBatch::ResidentState
	{
		// host -> device words: fine-grained device memory written by the host through the BAR (one address for both sides); without a
		// large BAR a pinned, coherent host block (`ctrl` its host address, `dCtrl` its device address: the workgroups then poll over PCIe)
		ResidentCtrl* ctrl = nullptr;
		ResidentCtrl* dCtrl = nullptr;
		bool ctrlInDeviceMemory = false;
		ResidentStatus* status = nullptr;  // device -> host word: pinned, coherent host block ...
		ResidentStatus* dStatus = nullptr; // ... and its device address
		unsigned* dDone = nullptr;     // [RESIDENT_RING]
		unsigned* dWgDone = nullptr;   // [wgCapacity]
		int wgCapacity = 0;
		std::vector<WnFrameGroup> list;
		bool configured = false;       // `list` is the launch list of topology `topology`, its index lists are on the device
		bool unsupported = false;      // ... or that topology cannot run resident
		unsigned long topology = ~0ul;
		unsigned long long posted = 0; // sequence number of the last command posted
		unsigned long long base = 0;   // ... of the last command before this generation of launches (all of them done, wgDone zero)
		unsigned long long markPosted = 0; // `posted` at the closing timing mark
		bool launched = false;         // a launch of this generation is (or was) on the stream: `gen` is recorded behind it
		bool exitRequested = false;    // exitAfter was set without waiting (closing timing mark): the next command starts a new generation
		hipEvent_t gen = nullptr;
		int grid = 0;
		~ResidentState()
		{
			if (gen) (void)hipEventDestroy(gen);
			if (dDone) (void)hipFree(dDone);
			if (dWgDone) (void)hipFree(dWgDone);
			if (ctrl) (void)(ctrlInDeviceMemory ? hipFree(ctrl) : hipHostFree(ctrl));
			if (status) (void)hipHostFree(status);
		}
	};

	bool HostDirect(); // (gpu_batch_host.cpp) the kernels read / write pinned host blocks themselves instead of the copy engines
}
