// model_desc.h -- host-side description of a loaded model, independent of any device.
//
// Produced by the loader (model_loader.cpp, mirrors NeuralAudio/NeuralModel.cpp:338-581) and
// consumed by the plan builders that lower it to device tables.
#pragma once

#include <string>
#include <utility>
#include <vector>

namespace na
{
	enum ActivationType { ACT_TANH = 0, ACT_LEAKYRELU = 1 };
	// the reference's WAVENET_MATH / LSTM_MATH build options (Activation.h:12-118) as a per-model field
	enum MathMode { MATH_FAST = 0, MATH_STD = 1 };

	// Template parameters of WaveNetLayerArrayT (NeuralAudio/WaveNet.h:503)
	struct WnArrayCfg
	{
		int inputSize = 1;
		int conditionSize = 1;
		int headSize = 1;
		int headKernelSize = 1;
		int headDilation = 1;
		int channels = 0;
		bool hasHeadBias = false;
		int activation = ACT_TANH;
		std::vector<int> kernelSizes;
		std::vector<int> dilations;
	};

	struct WaveNetDesc
	{
		std::vector<WnArrayCfg> arrays;
		std::vector<float> weights; // flat, reference order (WaveNet.h:700-719)
		bool isStatic = false;      // matched one of the official architectures (InternalModel.h:12-20)
		int mathMode = MATH_FAST;   // tanh policy of ACT_TANH layers

		size_t ExpectedNumWeights() const;
		int ReceptiveFieldSize() const; // WaveNet.h:534-542,674-684
	};

	struct LSTMLayerDesc
	{
		int inputSize = 1;
		std::vector<float> w;    // row-major [4H][I+H]  (LSTM.h:27); GRU: [3H][I+H], gate row blocks z,r,c
		std::vector<float> bias; // [4H]; GRU: [6H] = input bias | recurrent bias
		std::vector<float> h0;   // [H] initial hidden (NAM files carry it, LSTM.h:51-55; keras: zeros)
		std::vector<float> c0;   // [H]
	};

	enum RecurrentCell { CELL_LSTM = 0, CELL_GRU = 1 };

	// one keras "dense" or "conv1d" layer of a generic stack (RTNeural's Dense / Conv1D + activation layer, RTNeuralModel.h:300)
	enum DenseActivation { DENSE_LINEAR = 0, DENSE_TANH = 1, DENSE_RELU = 2, DENSE_SIGMOID = 3, DENSE_ELU = 4, DENSE_SOFTMAX = 5 };
	struct DenseLayerDesc
	{
		int in = 0, out = 0;
		int activation = DENSE_LINEAR;
		// conv1d (causal, stride 1): ksize taps `dilation` samples apart, tap k reads the input of (ksize - 1 - k) * dilation samples ago
		// (Keras Conv1D(padding = "causal")); a dense layer is ksize == 1
		int ksize = 1, dilation = 1;
		std::vector<float> w; // row-major [out][ksize * in], a row tap-major: index k * in + i
		std::vector<float> b; // [out]
		int RowLen() const { return in * ksize; }
		int History() const { return (ksize - 1) * dilation; }
	};

	struct LSTMDesc
	{
		int cell = CELL_LSTM; // CELL_GRU: keras GRU (RTNeural's arithmetic in the reference, NeuralModel.cpp:565-572)
		int numLayers = 0;
		int hiddenSize = 0;
		std::vector<LSTMLayerDesc> layers;
		std::vector<float> headWeights; // [H]
		float headBias = 0.0f;
		std::vector<float> headBiasVec; // scratch of the keras readers
		// Generic keras stacks (the reference's RTNeuralModelDyn, NeuralModel.cpp:565-572): 0..8 recurrent layers of one kind followed by
		// this chain of dense layers; the model output is unit 0 of the last one.  Empty: the classic [H] -> 1 linear head above.
		// numLayers == 0 with a tail: a pure dense stack on the input sample.
		std::vector<DenseLayerDesc> tail;
		bool isStatic = false;
		int mathMode = MATH_FAST; // LSTM only (the GRU follows RTNeural's accurate maths)
	};

	// load-time checks (throw std::runtime_error): weight count (WaveNet.h:704-709) and the shapes the gfx950 kernels accept
	void ValidateWaveNetDesc(const WaveNetDesc& desc);   // wavenet_plan.cpp
	void ValidateRecurrentDesc(const LSTMDesc& desc);    // model_loader.cpp

	enum ModelKind { MODEL_NONE = 0, MODEL_WAVENET = 1, MODEL_LSTM = 2 };

	struct ModelDesc
	{
		ModelKind kind = MODEL_NONE;
		WaveNetDesc wavenet;
		LSTMDesc lstm;
	};
}
