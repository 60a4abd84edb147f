// lstm_dev.h -- device-side data model for the LSTM recurrent path (reference: NeuralAudio/LSTM.h,
// NeuralAudio/LSTMDynamic.h -- same arithmetic).
//
// Mapping: the recurrence is strictly sample-serial inside a stream, so parallelism comes from
// streams only: one lane = one stream, 64 streams per wave64.  Weights are wave-uniform (scalar
// loads / SGPR broadcast), h and c live in LDS as [element][lane] (conflict-free columns) and are
// pulled into VGPRs for each gate mat-vec.
//
// HBM state per model group: float state[(layer*2H + k) * capacity + slot], k < H: hidden, k >= H: cell
// (structure-of-arrays over streams so a wave's 64 lanes load/store 256 contiguous bytes).
#pragma once

#include <algorithm>
#include <cstdlib>

#include "tuning.h"

namespace na
{
	constexpr int LSTM_MAX_LAYERS = 8;
	constexpr int LSTM_MAX_FRAMES = 128;
	constexpr int LSTM_MAX_TAIL = 8;        // dense layers of a generic keras stack (after lowering: activation / batchnorm / prelu layers become dense ones)
	constexpr int LSTM_MAX_TAIL_WIDTH = 256; // units per dense layer (two [width][64] scratch arrays in LDS: the shape predicates below check the fit)
	constexpr int LSTM_MAX_TAIL_HISTORY = 1024; // conv1d layers of a keras stack: (taps - 1) x dilation samples of input history at most

	enum { LSTM_CELL_LSTM = 0, LSTM_CELL_GRU = 1 };
	enum { LSTM_MATH_FAST = 0, LSTM_MATH_STD = 1 };

	// Shapes with a kernel (host-side predicates, no HIP types: the loader rejects everything else at load time).
	//   LSTM: any hidden size / layer count whose lane = stream working set fits the 160 KB LDS (LstmGenericKernel); the usual sizes
	//   have shaped kernels.  GRU: likewise (GruGenericKernel; GruWaveKernel / DPP instances for 1-2 layers of 8, 12, 16, 20).
	// tailWidth: widest dense layer of a generic keras stack (0: the classic 1-unit head); such a model may have no recurrent layer
	//   Round 3: the runtime-shaped wave kernel streams weights that do not fit the LDS from L2 (RecurrentWaveRtKernel, weights
	//   transposed for coalesced reads), so every shape up to RECURRENT_WAVE_MAX_HIDDEN units has a real-time kernel whatever the
	//   weight size (LSTMDynamic.h:95-108,166-179 runs any size on the CPU).
	//   Round 4: from 65 gate rows on (LSTM / GRU of more than 16 / 21 units on this kernel) a stream is a WORKGROUP of 2 .. 16 waves that share the
	//   gate rows (RecurrentWaveWaves; a barrier where the one-wave version has a wave fence), so the limit is 1024 units; beyond 128
	//   units the 1-unit head is evaluated inside the sample loop (no [samples][H] buffer) and a dense tail needs that buffer to fit.
	constexpr int RECURRENT_WAVE_MAX_HIDDEN = 1024;
	constexpr int RECURRENT_HEAD_IN_LOOP_FROM = 129; // hidden sizes from here on: classic head inside the sample loop
	// waves per stream of the runtime-shaped kernel: one gate row per lane up to 16 waves (the kernel is bound by the latency of its
	// weight loads from L2, and more waves keep more of them in flight: LSTM 1x256 x 64 streams 2.32 / 1.63 / 1.34 ms per 128-sample
	// block at four / two / one row per lane, LSTM 1x128 1.22 / 0.90 / 0.68; round 3, one wave: 2.0)
	inline int RecurrentWaveWaves(int gateRows)
	{
		const int rpl = Tuning::Get().recRpl; // tuning knob: gate rows per lane
		int waves = 1;
		while (waves < 16 && gateRows > 64 * rpl * waves) waves *= 2;
		return waves;
	}
	// tailHistory > 0: the tail has conv1d layers -- it is evaluated layer by layer over the whole block (recurrent_tail.h ConvTail) and its two
	// scratch arrays are [tailWidth][tailHistory + 128] (tailWidth then covers the widest INPUT of a tail layer as well)
#ifdef __HIPCC__
	__host__ __device__
#endif
	inline long RecurrentTailScratchFloats(int tailWidth, int tailHistory) { return tailHistory > 0 ? 2L * tailWidth * (tailHistory + LSTM_MAX_FRAMES) : 2L * tailWidth * 64; }
	inline bool RecurrentWaveShape(int hidden, int numLayers, int tailWidth, int tailHistory = 0)
	{
		if (!(hidden >= 1 && hidden <= RECURRENT_WAVE_MAX_HIDDEN && numLayers >= (tailWidth > 0 ? 0 : 1) && numLayers <= LSTM_MAX_LAYERS &&
			tailWidth <= std::max(LSTM_MAX_TAIL_WIDTH, tailHistory > 0 ? hidden : 0) && tailHistory <= LSTM_MAX_TAIL_HISTORY)) return false;
		// LDS without the weights (they stream from L2 when they do not fit): xin | h, c | gates | [samples][H] of the last layer (small models
		// and dense tails) | tail scratch
		const bool hseq = hidden < RECURRENT_HEAD_IN_LOOP_FROM || tailWidth > 0;
		const long floats = LSTM_MAX_FRAMES + 2L * numLayers * hidden + 6L * hidden + (hseq ? 2L * (numLayers > 0 ? hidden : 1) * 64 : 0) +
			RecurrentTailScratchFloats(tailWidth, tailHistory) + 64;
		return floats * 4 <= 160L * 1024;
	}
	// the lane = stream kernels' bound (LstmGenericKernel / GruGenericKernel: state of 64 streams in LDS)
	inline bool LstmLaneKernelShape(int hidden, int numLayers, int tailWidth = 0)
	{
		if (hidden < 1 || numLayers < (tailWidth > 0 ? 0 : 1) || numLayers > LSTM_MAX_LAYERS || tailWidth > LSTM_MAX_TAIL_WIDTH) return false;
		const long bytes = (64L * (LSTM_MAX_FRAMES + 1) + (long)numLayers * 2 * hidden * 64 + (long)hidden * 64 + 2L * tailWidth * 64) * 4;
		return bytes <= 160L * 1024;
	}
	inline bool GruLaneKernelShape(int hidden, int numLayers, int tailWidth = 0)
	{
		if (hidden < 1 || numLayers < 1 || numLayers > LSTM_MAX_LAYERS || tailWidth > LSTM_MAX_TAIL_WIDTH) return false;
		const long bytes = (64L * (LSTM_MAX_FRAMES + 1) + (long)numLayers * hidden * 64 + 6L * hidden * 64 + 2L * tailWidth * 64) * 4; // GruGenericKernel's LDS
		return bytes <= 160L * 1024;
	}
	// (a tail with conv1d layers -- tailHistory > 0 -- only runs on the runtime-shaped wave kernel)
	inline bool LstmShapeSupported(int hidden, int numLayers, int tailWidth = 0, int tailHistory = 0)
	{
		return (tailHistory == 0 && LstmLaneKernelShape(hidden, numLayers, tailWidth)) || RecurrentWaveShape(hidden, numLayers, tailWidth, tailHistory);
	}
	inline bool GruShapeSupported(int hidden, int numLayers, int tailWidth = 0, int tailHistory = 0)
	{
		return (tailHistory == 0 && GruLaneKernelShape(hidden, numLayers, tailWidth)) || (numLayers >= 1 && RecurrentWaveShape(hidden, numLayers, tailWidth, tailHistory));
	}

	struct LstmModelDev
	{
		// packed per layer l: W row-major [4H][I_l + H] (gate row blocks i,f,g,o -- LSTM.h:33-36), then bias[4H];
		// after the last layer: head weights [H], head bias [1]
		// GRU (cell == LSTM_CELL_GRU): per layer W row-major [3H][I_l + H] (gate row blocks z,r,c), then b_in[3H], b_rec[3H]
		const float* w;
		int cell;
		int numLayers;
		int hidden;
		int layerOff[LSTM_MAX_LAYERS]; // float offset of layer l's W
		int headOff;
		int math;      // LSTM only: 0 = FastMath (Activation.h:83-96), 1 = StdMath (Activation.h:20-45) -- the reference's LSTM_MATH build option
		// generic keras stack: tailLayers > 0 replaces the head by a chain of dense layers, layer t = W row-major [out][in] at
		// tailOff[t], then bias[out]; activation codes = DenseActivation (model_desc.h).  Only the runtime-shaped kernels evaluate it.
		int tailLayers;
		int tailOff[LSTM_MAX_TAIL], tailIn[LSTM_MAX_TAIL], tailOut[LSTM_MAX_TAIL], tailAct[LSTM_MAX_TAIL];
		int tailWidth; // widest layer (with conv1d layers: widest input or output of a tail layer)
		// conv1d layers of the tail (recurrent_tail.h ConvTail): taps / dilation per layer (1 / 1: a dense layer; weights [out][taps][in]), the
		// state row where the layer's input history starts (history x in rows, oldest sample first, behind the recurrent state's rows) and
		// the longest history (0: the tail has no conv1d layer)
		int tailK[LSTM_MAX_TAIL], tailDil[LSTM_MAX_TAIL], tailHistRow[LSTM_MAX_TAIL];
		int tailHistMax;
		// the same gate matrices transposed for the wave kernel's L2-streamed mode (weights larger than the LDS): per layer
		// [Qi + Qh][rowsPad][4] floats -- quad q of row r = weights of inputs 4q .. 4q + 3 (input part padded to Qi = ceil(I / 4) quads, hidden
		// part to Qh = ceil(H / 4)), rows padded to a multiple of 64: the 64 lanes of a wave read 64 consecutive rows of one quad = 1 KB
		const float* wT;
		int layerOffT[LSTM_MAX_LAYERS]; // float offsets into wT
		int rowsPad; // (a multiple of 64 x waves)
		int waves;   // waves per stream of the runtime-shaped kernel (RecurrentWaveWaves)
	};
}
