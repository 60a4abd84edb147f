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
		default: return v; // (5 = softmax: across the units of a layer, applied by the caller -- DenseTail / ConvTail below)
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
				dst[o * 64 + lane] = act == 5 ? acc : DenseActivate(acc, act);
			}
			if (act == 5) // softmax across the units of the layer, this lane's stream (as ConvTail does per sample)
			{
				float mx = dst[lane];
				for (int o = 1; o < out; o++) mx = fmaxf(mx, dst[o * 64 + lane]);
				float sum = 0.0f;
				for (int o = 0; o < out; o++)
				{
					const float e = __builtin_amdgcn_exp2f((dst[o * 64 + lane] - mx) * 1.4426950408889634f);
					dst[o * 64 + lane] = e;
					sum += e;
				}
				const float r = 1.0f / sum;
				for (int o = 0; o < out; o++) dst[o * 64 + lane] *= r;
			}
			cur = dst;
			curN = out;
		}
		return cur[lane];
	}

	// ---- tails with conv1d layers (keras Conv1D, padding = "causal", stride 1: RTNeural's Conv1D behind the reference's RTNeuralModelDyn,
	// RTNeuralModel.h:300; third-party arithmetic, parity unpinned like the rest of the generic stacks) -------------------------------
	// A tail has no recurrence between samples -- a conv1d layer only looks back at its own INPUT -- so such a tail is evaluated layer by
	// layer over the whole block, lane = sample: the sequence a layer reads is [unit][hist + n] in LDS (two arrays of `stride` =
	// tailHistMax + 128 floats per unit, used in turn), the `hist` = (taps - 1) x dilation samples in front of the block come from the
	// stream's state (rows tailHistRow[t] .., oldest sample first) and the block's last `hist` input samples go back there.  softmax
	// (RTNeural's SoftmaxActivation: exp(x - max) / sum over the units of a layer) is a second pass of the lane over its own sample.
	// One wave runs it (the caller's lanes 0 .. 63); hseq: the last recurrent layer's h as [pass][k][64], Hin its size (0: the input is xin).
	__device__ __forceinline__ void TailWaveSync()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
	}

	__device__ __forceinline__ void ConvTail(const LstmModelDev& m, const float* hseq, int Hin, const float* xin, float* bufA, float* bufB, float* __restrict__ state,
		int capacity, int slot, int n, int lane, float* __restrict__ outRow)
	{
		const int HM = m.tailHistMax, S = HM + LSTM_MAX_FRAMES;
		// the tail's input sequence, and the history of its first layer
		if (Hin > 0)
		{
			for (int i = 0; i < Hin; i++)
				for (int f = lane; f < n; f += 64) bufA[i * S + HM + f] = hseq[(size_t)((f >> 6) * Hin + i) * 64 + (f & 63)];
		}
		else
			for (int f = lane; f < n; f += 64) bufA[HM + f] = xin[f];
		{
			const int hist = (m.tailK[0] - 1) * m.tailDil[0], in = m.tailIn[0];
			for (int idx = lane; idx < hist * in; idx += 64)
				bufA[(idx % in) * S + HM - hist + idx / in] = state[(size_t)(m.tailHistRow[0] + idx) * capacity + slot];
		}
		TailWaveSync();
		float* cur = bufA;
		float* dst = bufB;
		for (int t = 0; t < m.tailLayers; t++)
		{
			const int in = m.tailIn[t], out = m.tailOut[t], act = m.tailAct[t], K = m.tailK[t], d = m.tailDil[t], hist = (K - 1) * d;
			const float* w = m.w + m.tailOff[t];
			const float* b = w + (size_t)out * K * in;
			for (int f = lane; f < n; f += 64)
			{
				for (int o = 0; o < out; o++)
				{
					float acc = b[o];
					for (int k = 0; k < K; k++)
					{
						const float* src = cur + (HM + f - (K - 1 - k) * d);
						const float* wr = w + ((size_t)o * K + k) * in;
						for (int i = 0; i < in; i++) acc += wr[i] * src[i * S];
					}
					dst[o * S + HM + f] = act == 5 ? acc : DenseActivate(acc, act);
				}
				if (act == 5)
				{
					float mx = dst[HM + f];
					for (int o = 1; o < out; o++) mx = fmaxf(mx, dst[o * S + HM + f]);
					float sum = 0.0f;
					for (int o = 0; o < out; o++)
					{
						const float e = __builtin_amdgcn_exp2f((dst[o * S + HM + f] - mx) * 1.4426950408889634f);
						dst[o * S + HM + f] = e;
						sum += e;
					}
					const float r = 1.0f / sum;
					for (int o = 0; o < out; o++) dst[o * S + HM + f] *= r;
				}
			}
			// this layer's input history for the next block: the last `hist` samples of [history | block]
			for (int idx = lane; idx < hist * in; idx += 64)
				state[(size_t)(m.tailHistRow[t] + idx) * capacity + slot] = cur[(idx % in) * S + HM + n - hist + idx / in];
			// the next layer's history in front of the sequence this layer has just written
			if (t + 1 < m.tailLayers)
			{
				const int histN = (m.tailK[t + 1] - 1) * m.tailDil[t + 1], inN = m.tailIn[t + 1];
				for (int idx = lane; idx < histN * inN; idx += 64)
					dst[(idx % inN) * S + HM - histN + idx / inN] = state[(size_t)(m.tailHistRow[t + 1] + idx) * capacity + slot];
			}
			TailWaveSync();
			float* sw = cur; cur = dst; dst = sw;
		}
		for (int f = lane; f < n; f += 64) outRow[f] = cur[HM + f];
	}
}
