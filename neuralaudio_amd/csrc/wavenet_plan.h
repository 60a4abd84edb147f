// wavenet_plan.h -- lowers a WaveNetDesc into the stage program + packed MFMA operand tables
// consumed by the gfx950 kernels (see wavenet_dev.h for the layouts).
#pragma once

#include <cstdint>
#include <vector>

#include "model_desc.h"
#include "wavenet_dev.h"

namespace na
{
	struct WnRingInfo
	{
		int G;          // channel groups
		int frames;     // ring length in frames (multiple of 16)
		int offF4;      // float4 offset within the stream state
		int channels;   // real channel count
	};

	constexpr double kSplitMinInputLimit = 8.0; // +18 dBFS: below this input limit the f16-split kernels do not take a model

	struct WaveNetPlan
	{
		std::vector<WnStage> stages;
		std::vector<float> wpack;   // size multiple of 4
		std::vector<float> wpk;     // weights for the packed-FMA (lane = frame) kernel
		std::vector<WnRingInfo> rings;
		std::vector<WnPrewarmLayer> prewarm;
		std::vector<WnSplitStage> sstages; // stage program of the f16-split kernel
		std::vector<uint16_t> wsplit;      // its A-operand image: f16 bit patterns, 512 per MFMA operand
		int maxSplitOps = 0;
		int maxG = 1;
		int splitFastT = 0;                // see WnModelDev::split_fast_T
		int maxChannels = 0;               // widest layer array
		int pack = 1;                      // streams per virtual stream (BuildPackedWaveNetPlan)
		bool packDense = false;            // ... and the 2-channel arrays of the streams sit side by side, two streams per channel group (PackWaveNetDesc)
		bool isVirtual() const { return !packedWeights.empty(); } // packed and / or padded: arrays, rings, stages describe the virtual model
		std::vector<float> packedWeights;  // pack > 1: flat weights of the virtual model (reference order)
		bool genericOnly = false;          // > 16 channels: only rings + the natural-layout table are built (runtime-shaped block kernel)
		bool genericOk = false;            // the runtime-shaped block kernel can run it (dense heads only)
		int stateF4 = 0;            // per-stream state in float4 units
		int maxA4Floats = 0;        // largest per-stage A-operand block of the frame kernel (floats)
		float headScale = 0.0f;
		float condLimit = 32752.0f; // f16-split kernels: input samples are clamped to +-condLimit (range contract)
		// static proof of that contract (wavenet_plan.cpp): with inputs inside +-condLimit >= kSplitMinInputLimit no value of the chain leaves
		// the f16 range, and the weights fit the f16-split operand format.  A plan that fails either runs on the f32 frame kernel.
		bool compactRings = false;  // some ring is a compact one: launches take buffer lengths of WnCompactSafeFrames() only
		bool splitRangeProven = true;
		bool splitWeightsOk = true;
		int receptiveField = 0;

		// roofline bookkeeping: compulsory HBM bytes and MACs per sample at block N
		double AlgorithmicBytesPerSample(int blockFrames) const;
		double MacsPerSample() const;

		std::vector<WnArrayCfg> arrays;
	};

	// throws std::runtime_error("Wrong number of weights...") like WaveNet.h:704-709
	// splitStateFormat: the stream state is laid out for the f16-split kernels (rings of layers with a dilation >= 128 are exactly
	// (K - 1) d frames long, wavenet_plan.cpp AddRing); false: for the frame / runtime-shaped kernels (every ring roundup16 + 128)
	WaveNetPlan BuildWaveNetPlan(const WaveNetDesc& desc, bool splitStateFormat = false);
	// Stream packing (wavenet_plan.cpp): how many streams of this model fit one virtual stream of the f16-split kernel (1: none),
	// the virtual model, and its plan (WaveNetPlan::pack = P; arrays / rings / stages describe the VIRTUAL model).
	int WaveNetPackFactor(const WaveNetDesc& desc);
	bool WaveNetWantsPadding(const WaveNetDesc& desc); // P == 1 "packing": widen the arrays to full lane modes (A1 Lite: 12 / 6 -> 16 / 8)
	// dense: a 2-channel array of a four-stream pack is NOT padded to a channel group per stream -- two streams share one (Nano: 16 / 8
	// virtual channels instead of 16 / 16; the kernels then feed two conditions through one aux operand, FillSplitAux)
	bool WaveNetPackCanBeDense(const WaveNetDesc& desc, int P);
	WaveNetDesc PackWaveNetDesc(const WaveNetDesc& desc, int P, bool dense = false);
	WaveNetPlan BuildPackedWaveNetPlan(const WaveNetDesc& desc, int P, bool dense = false);
}
