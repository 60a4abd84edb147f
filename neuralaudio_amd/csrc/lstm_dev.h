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

namespace na
{
	constexpr int LSTM_MAX_LAYERS = 8;
	constexpr int LSTM_MAX_FRAMES = 128;

	enum { LSTM_CELL_LSTM = 0, LSTM_CELL_GRU = 1 };

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
	};
}
