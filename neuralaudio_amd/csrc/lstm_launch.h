// lstm_launch.h -- host-callable launchers for the kernels in lstm_kernels.hip
#pragma once

#include <hip/hip_runtime_api.h>

#include "lstm_dev.h"

namespace na
{

	// One block of n <= 128 samples for `numStreams` streams of one model (lane = stream).
	hipError_t LaunchLstmBlock(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream);

	// LDS-free kernels for hidden size 8 / 16, 1-2 layers, LSTM or GRU (recurrent_dpp_kernels.hip): one launch over several model groups
	constexpr int RECURRENT_MAX_GROUPS = 8;
	struct RecurrentGroup
	{
		LstmModelDev model;
		float* state;
		int capacity;
		const int* slots; // nullptr: the active streams are contiguous -- stream i uses state slot slot0 + i and matrix row row0 + i
		const int* rows;  // (saves the kernel a dependent global load before it can touch the stream's state)
		int numStreams;
		int slot0, row0;
	};
	bool RecurrentDppSupported(const LstmModelDev& m);
	// four streams per wave (RecurrentQuadKernel) for launches of at least this many streams of LSTMs up to 2x16; tests / tuning
	bool RecurrentQuadSupported(const LstmModelDev& m);
	int RecurrentQuadMinStreams();
	int SetRecurrentQuadMinStreams(int streams); // returns the previous value
	long RecurrentQuadLaunches();
	// The runtime-shaped one-wave-per-stream kernel (lstm_kernels.hip): LSTM or GRU cells, hidden <= 64, any layer count, classic head or
	// dense chain, as long as the weights of all layers fit the LDS.  false: not launched (the lane = stream kernels take the model).
	bool LaunchRecurrentWaveRt(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream, hipError_t& err);
	hipError_t LaunchRecurrentDpp(const RecurrentGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream);
	// ... any number of groups in one launch, the group table in device memory (`table`: the batch's cache of it; wavenet_launch.h)
	struct WnLaunchTable;
	hipError_t LaunchRecurrentDppTable(const RecurrentGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, WnLaunchTable& table);

	// keras GRU (gru_kernels.hip): same state layout (only the h half of every layer is used), m.cell == LSTM_CELL_GRU
	hipError_t LaunchGruBlock(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream);

	// state[k*capacity + slot] = init[k] for the listed slots
	hipError_t LaunchLstmInitState(float* state, int capacity, const int* slots, int numStreams, const float* init, int numElems,
		hipStream_t stream);
}
