// wavenet_launch.h -- host-callable launchers of the WaveNet kernels (wavenet_split_kernels.hip, wavenet_frame_kernels.hip, wavenet_prewarm_kernels.hip)
#pragma once

#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "wavenet_dev.h"

namespace na
{
	// One block of n <= 128 frames for `numStreams` streams of one model:
	//   slots[i]: state slot of active stream i; rows[i]: its row in `in`/`out` (row stride in floats).
	// Lane = frame kernel on v_mfma_f32_4x4x1_16b_f32 (wavenet_frame_kernels.hip).
	// One launch over several model groups (a heterogeneous batch): workgroups are assigned to the groups in order.
	constexpr int WN_FRAME_MAX_GROUPS = 8;
	struct WnFrameGroup
	{
		const WnModelDev* model;
		float* state;
		const int* slots; // nullptr: contiguous (slot0 + i, row0 + i)
		const int* rows;
		int numStreams, slot0, row0;
		// f16-split kernel only: > 1 = packed group (WaveNetPlan::pack real streams per virtual stream): numStreams / slots count VIRTUAL
		// streams and rows holds `pack` entries per virtual stream (-1: no real stream in that position); slots must not be nullptr
		int pack;
	};
	hipError_t LaunchWaveNetFrameFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream);

	// Same contract, the f16-split MFMA kernel (wavenet_split_kernels.hip) -- the shipped path.  Its stream state uses split quads in
	// frame-major rings (see WnSplitStage), so a model group stays on one kernel family for its whole life.
	// sharing: how many launches of this size run on the chip at the same time (the free-running half-batch chains: 2) -- the
	// workgroup shape is chosen for what is resident, not for what one launch brings
	hipError_t LaunchWaveNetSplitFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, int sharing = 1);

	// Same contract on the compile-time specialised layer chains of the official architectures (wavenet_spec_kernels.hip): blocks of
	// exactly 128 / 64 / 32 frames, every group of the launch from one architecture family (WnModelDev::spec_arch); returns
	// hipErrorNotSupported otherwise -- LaunchWaveNetSplitFused tries it first and falls back to its stage interpreter.
	// `sharing` of the fused launches below: the number of launches that share the chip (free-running chains), OR-ed with this bit when the
	// batch's stream state does not fit the 256 MB Infinity Cache (the chains then mark the long dilations' ring traffic non-temporal)
	constexpr int WN_SHARING_BEYOND_CACHE = 1 << 16;
	// ... from this much stream state on (measured, us per 1024 A1 Standard streams with / without the non-temporal bits: 1280 streams =
	// 311 MB 40.4 / 39.2 -- part of the state still lives in the cache --, 2048 = 498 MB 37.6 / 39.2, 8192 = 2 GB 37.4 - 38.4 / 37.9 - 39.2)
	constexpr size_t WN_BEYOND_CACHE_BYTES = (size_t)400 << 20;
	hipError_t LaunchWaveNetSpecFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, int sharing = 1);
	// test hook (GpuBatch::DebugStallDevice): one wave that keeps `stream` busy for `ms` milliseconds (s_memrealtime, 100 MHz)
	hipError_t LaunchStallKernel(double ms, hipStream_t stream);
	// ---- table launches of the specialised chains (wavenet_spec_impl.h WaveNetSpecTableKernel): any number of model groups in one launch --
	// The device copies of a launch list's group tables, owned by the batch (one WnLaunchTable per launch list); see LaunchWaveNetSpecTable.
	// An entry is IMMUTABLE once uploaded -- a table with other contents (another block length, other members) gets its own device
	// buffer -- so a captured graph that replays a table launch keeps reading what it was captured with, and nothing is allocated,
	// freed or copied inside a stream capture: the batch runs the launch list once with `prepareOnly` set before it begins the capture
	// (gpu_batch.cpp ProcessDeviceOn), which uploads every table the captured launches will look up.  The upload is a blocking copy
	// into a buffer no launch has seen yet.  Entries are dropped when the batch's topology changes (NewGeneration, after the graphs
	// that point at them are gone; hipFree waits for the device).
	struct WnLaunchTable
	{
		struct Entry
		{
			void* dev = nullptr;
			std::vector<char> host; // what `dev` holds
		};
		std::vector<Entry> entries;
		unsigned long generation = 0;
		bool prepareOnly = false; // the launch functions return after Ensure()
		WnLaunchTable() = default;
		WnLaunchTable(const WnLaunchTable&) = delete;
		WnLaunchTable& operator=(const WnLaunchTable&) = delete;
		~WnLaunchTable() { Clear(); }
		void Clear()
		{
			for (Entry& e : entries)
				if (e.dev) (void)hipFree(e.dev);
			entries.clear();
		}
		void NewGeneration(unsigned long g)
		{
			if (g != generation) Clear();
			generation = g;
		}
		// the device copy of `bytes` bytes at `fresh`, uploaded now if no entry holds them
		hipError_t Ensure(const void* fresh, size_t bytes, hipStream_t stream, const void** dev)
		{
			for (const Entry& e : entries)
				if (e.host.size() == bytes && memcmp(e.host.data(), fresh, bytes) == 0)
				{
					*dev = e.dev;
					return hipSuccess;
				}
			hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
			(void)hipStreamIsCapturing(stream, &capturing);
			if (capturing != hipStreamCaptureStatusNone) return hipErrorStreamCaptureUnsupported; // (the prepare pass has not seen this table: a bug)
			if (entries.size() >= 16) Clear(); // contents that change without a topology change: keep the table bounded
			Entry e;
			hipError_t err = hipMalloc(&e.dev, bytes);
			if (err != hipSuccess) return err;
			err = hipMemcpy(e.dev, fresh, bytes, hipMemcpyHostToDevice);
			if (err != hipSuccess)
			{
				(void)hipFree(e.dev);
				return err;
			}
			e.host.assign(static_cast<const char*>(fresh), static_cast<const char*>(fresh) + bytes);
			*dev = e.dev;
			entries.push_back(std::move(e));
			return hipSuccess;
		}
	};
	// A launch list of MORE than WN_FRAME_MAX_GROUPS groups of one architecture family (all Standard; lite-family groups, packed or not;
	// the A2 submodels) as ONE launch of 128-frame blocks; hipErrorNotSupported otherwise (the caller then cuts the list into launches of eight).
	hipError_t LaunchWaveNetSpecTable(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, WnLaunchTable& table);

	// ---- resident ("persistent") launches of the specialised chains (wavenet_spec_impl.h WaveNetSpecResidentKernel) --------------
	// One launch stays on the chip and walks consecutive buffers by itself: the host posts a command per buffer into a ring, every
	// workgroup polls the command it needs next, runs its streams' block and counts itself done.  Workgroups never wait for each other
	// -- only for the host -- so the launch is live whatever part of its grid is resident, and a workgroup that finds no command for
	// `idleTicks` (or has reached `exitAfter`) leaves; the host relaunches when there is work again (gpu_batch_chains.cpp).
	// Where the words live (tools/microbench/resident_cmd_probe.hip, profiles/r05_microbench_resident_cmd_probe.txt): 512 workgroups
	// polling a ring in pinned HOST memory cost 34 - 700 us per command when they all wait and 6 us per command behind 30 us of work
	// (two dependent PCIe round trips each); a ring in fine-grained DEVICE memory that the host writes through the BAR costs 1.7 us.
	// So host -> device words (commands, exitAfter) are in device memory, the one device -> host word (`completed`) in host memory:
	// nobody ever polls across PCIe.
	constexpr int RESIDENT_RING = 64; // commands in flight at most (host-side back-pressure)
	struct ResidentCmd
	{
		const float* in;
		float* out;
		long inStride, outStride;
		unsigned long long seq;   // the command's sequence number ...
		unsigned long long check; // ... and ResidentCmdCheck of the five words above: a torn read of the line does not pass
		unsigned long long pad[2];
	};
#ifdef __HIPCC__
	__host__ __device__
#endif
	inline unsigned long long ResidentCmdCheck(unsigned long long in, unsigned long long out, unsigned long long inStride, unsigned long long outStride, unsigned long long seq)
	{
		return seq * 0x9E3779B97F4A7C15ull + in * 0xC2B2AE3D27D4EB4Full + out * 0x165667B19E3779F9ull + inStride * 0xD6E8FEB86659FD93ull + outStride * 0xFF51AFD7ED558CCDull + 1ull;
	}
	struct ResidentCtrl // host -> device: fine-grained device memory the host writes through the BAR (pinned host memory without a large BAR)
	{
		unsigned long long exitAfter; // workgroups leave once they have run this sequence number
		unsigned long long pad0[7];
		ResidentCmd cmd[RESIDENT_RING];
	};
	struct ResidentStatus // device -> host: pinned host memory
	{
		unsigned long long completed; // every workgroup has run the commands up to this one
		unsigned long long pad1[7];
	};
	static_assert(sizeof(ResidentCmd) == 64 && sizeof(ResidentCtrl) == 64 + 64 * RESIDENT_RING && sizeof(ResidentStatus) == 64, "command ring layout");
	struct ResidentArgs
	{
		const ResidentCtrl* ctrl;     // device address of the command block
		ResidentStatus* status;       // device address of the pinned status block
		unsigned* doneCount;          // [RESIDENT_RING] device memory: workgroups that have run command seq (slot seq % RESIDENT_RING)
		unsigned* wgDone;             // [grid] device memory: commands this workgroup has run since `base` (a relaunch resumes there)
		unsigned long long base;      // sequence number of the last command before this generation of launches
		int numBlocks;                // workgroups' worth of streams per command: workgroup b runs blocks b, b + grid, ...
		unsigned idleTicks;           // s_memrealtime ticks (100 MHz) without a command after which a workgroup leaves
		unsigned startDelay;          // ticks the second half of the grid (the second workgroup of every CU) waits before its first command
	};
	// Starts (or restarts) the resident launch of a launch list that LaunchWaveNetSpecFused would run as ONE launch of 128-frame blocks;
	// hipErrorNotSupported otherwise.  grid = min(workgroups of the list, what is resident at the kernel's occupancy); *gridOut says how
	// many workgroups were launched (wgDone must hold that many counters).
	hipError_t LaunchWaveNetSpecResident(const WnFrameGroup* groups, int numGroups, int n, const ResidentArgs& ra, hipStream_t stream, int* gridOut);
	// the grid LaunchWaveNetSpecResident would use (0: the list cannot run resident)
	int WaveNetSpecResidentGrid(const WnFrameGroup* groups, int numGroups, int n);

	bool WaveNetSpecEnabled();
	void SetWaveNetSpecEnabled(bool on); // process-wide; not for use while launches are being issued from other threads
	// which specialised chain (WnSpecArch) runs a split-kernel plan, WN_SPEC_NONE if none; host data
	int WaveNetSpecArchId(const WnSplitStage* stages, int nstages, int stateF4, int wsplitQuads);

	// The runtime-shaped block kernel (wavenet_generic_kernels.hip): up to 64 channels per layer array, dense heads; walks the
	// natural-layout tensor table (WaveNetPlan::prewarm) over the flat reference-order weights; frame-kernel stream-state format.
	hipError_t LaunchWaveNetGeneric(const WnPrewarmLayer* layers, int numLayers, const float* weights, const int* ringOffF4, const int* ringFrames,
		const int* ringG, int nrings, int stateF4, int maxChannels, float headScale, float* state, const int* slots, const int* rows, int numStreams,
		int slot0, int row0, const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream);

	// slots == nullptr: the active streams are contiguous -- stream i uses state slot slot0 + i and matrix row row0 + i (saves the
	// kernel a dependent global load before it can touch the stream's state)
	hipError_t LaunchWaveNetFrame(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in,
		float* out, long inStride, long outStride, int n, hipStream_t stream, int slot0 = 0, int row0 = 0);

	// tuning aid: device buffer of long long[stages*4*waves] that workgroup 0 stamps with the shader clock (nullptr: off)
	void SetWaveNetTraceBuffer(long long* deviceBuffer);
	long long* GetWaveNetTraceBuffer();

	// Zero-input steady-state columns per ring (once per model), cols = [nrings][WN_COL_STRIDE] floats.
	hipError_t LaunchWaveNetPrewarmColumns(const WnPrewarmLayer* layers, int numLayers, const float* weights, float* cols,
		hipStream_t stream);

	// Broadcast the columns into the rings of the listed stream slots and zero their cursors.  splitFormat: the f16-split kernel's state
	// (split quads, frame-major rings) instead of the frame kernel's (f32 quads, tile layout).
	hipError_t LaunchWaveNetFillRings(float* state, int stateF4, const int* slots, int numStreams, int numRings, const int* ringOffF4,
		const int* ringFrames, const int* ringG, const float* cols, hipStream_t stream, bool splitFormat, const int* sub = nullptr, int pack = 1,
		bool zero = false); // packed groups: (slots[i], sub[i]) = one real stream, only its channel groups are written (see the kernel)
}
