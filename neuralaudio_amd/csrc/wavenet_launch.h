// wavenet_launch.h -- host-callable launchers of the WaveNet kernels (wavenet_split_kernels.hip, wavenet_frame_kernels.hip, wavenet_prewarm_kernels.hip)
#pragma once

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
	hipError_t LaunchWaveNetSpecFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, int sharing = 1);
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
