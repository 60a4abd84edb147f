// recurrent_tail.h -- the dense-layer chain of a generic keras stack (RTNeural's Dense + activation layers behind the reference's
// RTNeuralModelDyn, RTNeuralModel.h:300,417-421), evaluated per sample inside the runtime-shaped lane = stream kernels
// (LstmGenericKernel, GruGenericKernel).  Arithmetic: accurate tanh, sigmoid = (tanh(x/2)+1)/2 -- the reference's FastMathsProvider
// (RTNeuralModel.h:10-31) -- on the exp2 / rcp units (absolute error ~1e-7).  Parity unpinned (RTNeural is an absent submodule).
#pragma once

#include <hip/hip_runtime.h>

#include "dpp_recurrent.h"
#include "lstm_dev.h"

namespace na
{
	__device__ __forceinline__ float DenseActivate(float v, int act)
	{
		switch (act) // wave-uniform
		{
		case 1: return GruTanh(v);
		case 2: return v > 0.0f ? v : 0.0f;
		case 3: return GruSigmoid(v);
		case 4: return v > 0.0f ? v : (__builtin_amdgcn_exp2f(v * 1.4426950408889634f) - 1.0f); // elu, alpha = 1
		default: return v;
		}
	}

	// vec: the last recurrent layer's h as [k * 64 + lane] (n0 = its size), or nullptr with n0 == 0: the input is the scalar x0.
	// bufA / bufB: two [tailWidth][64] scratch arrays in LDS.  Returns unit 0 of the last layer for this lane's stream.
	__device__ __forceinline__ float DenseTail(const LstmModelDev& m, const float* vec, int n0, float x0, float* bufA, float* bufB, int lane)
	{
		const float* cur = vec;
		int curN = n0;
		for (int t = 0; t < m.tailLayers; t++)
		{
			const int in = m.tailIn[t], out = m.tailOut[t], act = m.tailAct[t];
			const float* w = m.w + m.tailOff[t];
			const float* b = w + (size_t)out * in;
			float* dst = (t & 1) ? bufB : bufA;
			for (int o = 0; o < out; o++)
			{
				float acc = b[o];
				if (curN == 0) acc += w[o] * x0;
				else
					for (int k = 0; k < in; k++) acc += w[(size_t)o * in + k] * cur[k * 64 + lane];
				dst[o * 64 + lane] = DenseActivate(acc, act);
			}
			cur = dst;
			curN = out;
		}
		return cur[lane];
	}
}
