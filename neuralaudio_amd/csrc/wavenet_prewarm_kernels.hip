// wavenet_prewarm_kernels.hip -- zero-input steady state of a WaveNet model (WaveNetModelT::Prewarm, NeuralAudio/WaveNet.h:746-766,
// 607-630, 74-82) and its broadcast into the history rings of freshly added / re-prewarmed streams, for both stream-state formats
// (f16-split kernel: split quads, frame-major rings; frame kernel: f32 quads, tile layout).
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

namespace na
{
	typedef float f32x4 __attribute__((ext_vector_type(4)));

	// Activation.h:83-91
	__device__ __forceinline__ float FastTanh(float x)
	{
		const float ax = fabsf(x);
		const float x2 = x * x;
		return (x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2)) /
			(2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax));
	}

	// Activation.h:110-118
	__device__ __forceinline__ float LeakyReLU(float x) { return x > 0.0f ? x : 0.01f * x; }

	// ------------------------------------------------------------------------------------------
	// Prewarm (WaveNet.h:746-766): zero-input steady state.  Every layer input is a constant column
	// that depends only on the weights, so it is computed once per MODEL by one small workgroup (thread = channel),
	// in the reference's natural weight layout, then broadcast into every stream's rings.
	// ------------------------------------------------------------------------------------------
	__global__ void __launch_bounds__(WN_COL_STRIDE) WaveNetPrewarmColumnsKernel(const WnPrewarmLayer* __restrict__ layers, int numLayers,
		const float* __restrict__ w, float* __restrict__ cols /* [ring][WN_COL_STRIDE] */)
	{
		constexpr int CW = WN_COL_STRIDE; // 128 = one thread per channel
		__shared__ float x[CW], z[CW], head[CW], lin[CW];
		const int i = threadIdx.x;
		x[i] = 0.0f; z[i] = 0.0f; head[i] = 0.0f; lin[i] = 0.0f; // condition = 0 (:748), headArray zero (:750)
		__syncthreads();

		for (int li = 0; li < numLayers; li++)
		{
			const WnPrewarmLayer L = layers[li];
			if (L.kind == 0)
			{
				if (L.rechannel >= 0)
				{
					// rechannel.Process (:609): x = W_re * layer_inputs
					float v = 0.0f;
					if (i < L.cin)
						for (int c = 0; c < L.rech_in; c++) v += w[L.rechannel + i * L.rech_in + c] * lin[c];
					__syncthreads();
					x[i] = (i < L.cin) ? v : 0.0f;
					__syncthreads();
				}
				cols[L.ring_id * CW + i] = x[i]; // CopyBuffer (:74-82): the whole receptive field holds this column

				float acc = 0.0f;
				if (i < L.cout)
				{
					for (int k = 0; k < L.ksize; k++)
						for (int c = 0; c < L.cin; c++) acc += w[L.wconv + (i * L.cin + c) * L.ksize + k] * x[c];
					acc += w[L.bconv + i];
					// mix-in * condition(0) adds nothing
					acc = (L.act == 1) ? LeakyReLU(acc) : (L.act == 2 ? (1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc * 2.885390081777927f) + 1.0f)) : FastTanh(acc));
				}
				__syncthreads();
				z[i] = (i < L.cout) ? acc : 0.0f;
				head[i] += z[i];
				__syncthreads();
				float y = 0.0f;
				if (i < L.cout)
				{
					for (int c = 0; c < L.cin; c++) y += w[L.w1 + i * L.cin + c] * z[c];
					y += w[L.b1 + i];
					y += x[i];
				}
				__syncthreads();
				if (L.last_of_array) lin[i] = (i < L.cout) ? y : 0.0f; // arrayOutputs feeds the next array's rechannel
				else x[i] = (i < L.cout) ? y : 0.0f;
				__syncthreads();
			}
			else
			{
				// head rechannel (:625-629): steady-state head column, then conv over a constant history
				if (L.ring_id >= 0) cols[L.ring_id * CW + i] = (i < L.cin) ? head[i] : 0.0f;
				float acc = 0.0f;
				if (i < L.cout)
				{
					for (int k = 0; k < L.ksize; k++)
						for (int c = 0; c < L.cin; c++) acc += w[L.wconv + (i * L.cin + c) * L.ksize + k] * head[c];
					if (L.bconv >= 0) acc += w[L.bconv + i];
				}
				__syncthreads();
				head[i] = (i < L.cout) ? acc : 0.0f; // becomes the next array's head accumulator (:785-789)
				__syncthreads();
			}
		}
	}

	// float quad -> split quad [h0 h1 | h2 h3 | l0 l1 | l2 l3] (h = f16(v), l = f16(v - h)): the storage format of the f16-split kernel
	__device__ __forceinline__ f32x4 SplitQuadBits(f32x4 v)
	{
		typedef _Float16 h2 __attribute__((ext_vector_type(2)));
		typedef float f2 __attribute__((ext_vector_type(2)));
		const h2 h01 = __builtin_convertvector(f2{ v.x, v.y }, h2), h23 = __builtin_convertvector(f2{ v.z, v.w }, h2);
		const h2 l01 = __builtin_convertvector(f2{ v.x - (float)h01.x, v.y - (float)h01.y }, h2);
		const h2 l23 = __builtin_convertvector(f2{ v.z - (float)h23.x, v.w - (float)h23.y }, h2);
		return f32x4{ __builtin_bit_cast(float, h01), __builtin_bit_cast(float, h23), __builtin_bit_cast(float, l01), __builtin_bit_cast(float, l23) };
	}

	// grid = (streams to fill, rings), block = 256: fill ring r of stream slot with its steady-state column.
	// split == 0: f32 quads in the tile layout (frame kernel); split == 1: split quads, frame-major rings (f16-split kernel).
	// Packed groups (pack > 1, split format): entry i names one REAL stream = (virtual stream slots[i], position sub[i]); only the
	// channel groups of that position are written and the cursors -- shared by the streams of a virtual stream -- are left alone (a
	// steady-state column fills every ring position, so where the cursor stands does not matter).  zero != 0: zeros instead of the column.
	__global__ void __launch_bounds__(256) WaveNetFillRingsKernel(f32x4* __restrict__ state, int stateF4, const int* __restrict__ slots,
		const int* __restrict__ ringOffF4, const int* __restrict__ ringFrames, const int* __restrict__ ringG, const float* __restrict__ cols, int split,
		const int* __restrict__ sub, int pack, int zero)
	{
		const int slot = slots[blockIdx.x];
		const int r = blockIdx.y;
		f32x4* st = state + (size_t)slot * (size_t)stateF4;
		const int G = ringG[r];
		const int nF4 = (ringFrames[r] / 16) * G * 16;
		f32x4* ring = st + ringOffF4[r];
		const int gs = pack > 1 ? G / pack : G;            // channel groups per real stream
		if (pack > 1 && gs == 0)
		{
			// a dense pack (wavenet_plan.cpp PackWaveNetDesc): G x 4 / pack = 2 channels per stream, two streams per channel group.  Of a
			// split quad [h0 h1 | h2 h3 | l0 l1 | l2 l3] the first stream of a pair owns dwords 0 and 2, the second one dwords 1 and 3:
			// plain dword stores (streams of one virtual stream may be filled by different workgroups of this launch)
			const int cgOwn = sub[blockIdx.x] >> 1, half = sub[blockIdx.x] & 1;
			float* ringW = reinterpret_cast<float*>(ring);
			for (int idx = threadIdx.x; idx < nF4; idx += blockDim.x)
			{
				if (idx % G != cgOwn) continue;
				const float* c = cols + r * WN_COL_STRIDE + cgOwn * 4;
				const f32x4 v = SplitQuadBits(zero ? f32x4{ 0.0f, 0.0f, 0.0f, 0.0f } : f32x4{ c[0], c[1], c[2], c[3] });
				ringW[(size_t)idx * 4 + half] = half ? v.y : v.x;
				ringW[(size_t)idx * 4 + 2 + half] = half ? v.w : v.z;
			}
			return; // (packed: the cursors are left alone)
		}
		const int cgFirst = pack > 1 ? sub[blockIdx.x] * gs : 0;
		for (int idx = threadIdx.x; idx < nF4; idx += blockDim.x)
		{
			const int cg = split ? (idx % G) : ((idx >> 4) % G);
			if (cg < cgFirst || cg >= cgFirst + gs) continue;
			const float* c = cols + r * WN_COL_STRIDE + cg * 4;
			const f32x4 v = zero ? f32x4{ 0.0f, 0.0f, 0.0f, 0.0f } : f32x4{ c[0], c[1], c[2], c[3] };
			ring[idx] = split ? SplitQuadBits(v) : v;
		}
		if (pack <= 1 && r == 0 && threadIdx.x < WN_MAX_RINGS) reinterpret_cast<int*>(st)[threadIdx.x] = 0; // cursors
	}

	// ------------------------------------------------------------------------------------------ launchers

	static long long* g_traceBuffer = nullptr;
	void SetWaveNetTraceBuffer(long long* p) { g_traceBuffer = p; }
	long long* GetWaveNetTraceBuffer() { return g_traceBuffer; }

	hipError_t LaunchWaveNetPrewarmColumns(const WnPrewarmLayer* layers, int numLayers, const float* weights, float* cols,
		hipStream_t stream)
	{
		hipLaunchKernelGGL(WaveNetPrewarmColumnsKernel, dim3(1), dim3(WN_COL_STRIDE), 0, stream, layers, numLayers, weights, cols);
		return hipGetLastError();
	}

	hipError_t LaunchWaveNetFillRings(float* state, int stateF4, const int* slots, int numStreams, int numRings, const int* ringOffF4,
		const int* ringFrames, const int* ringG, const float* cols, hipStream_t stream, bool splitFormat, const int* sub, int pack, bool zero)
	{
		if (numStreams <= 0) return hipSuccess;
		hipLaunchKernelGGL(WaveNetFillRingsKernel, dim3((unsigned)numStreams, (unsigned)numRings), dim3(256), 0, stream,
			reinterpret_cast<f32x4*>(state), stateF4, slots, ringOffF4, ringFrames, ringG, cols, splitFormat ? 1 : 0, sub, pack, zero ? 1 : 0);
		return hipGetLastError();
	}

	// ---- test hook: a stream that does not answer (GpuBatch::DebugStallDevice) ----
	__global__ void __launch_bounds__(64) StallKernel(unsigned long long ticks)
	{
		const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
		while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
	}

	hipError_t LaunchStallKernel(double ms, hipStream_t stream)
	{
		hipLaunchKernelGGL(StallKernel, dim3(1), dim3(64), 0, stream, (unsigned long long)(ms * 1.0e5));
		return hipGetLastError();
	}
}
