// gru_kernels.hip -- gfx950 kernel for keras GRU models (BASELINE config 4).
//
// In the reference a keras "gru" model is evaluated by RTNeural, not by NeuralAudio's own code
// (NeuralAudio/NeuralModel.cpp:565-572 -> NeuralAudio/RTNeuralModel.h:300, 417-429; the RTNeural submodule is an empty
// directory in the reference tree).  The arithmetic here is the published GRU(reset_after=True) of Keras / RTNeural's
// GRULayer with the reference's FastMathsProvider (RTNeuralModel.h:10-31: accurate tanh, sigmoid(x) = (tanh(x/2)+1)/2):
//     z = sigma(W_z x + U_z h + b_z0 + b_z1);  r = sigma(W_r x + U_r h + b_r0 + b_r1)
//     c = tanh(W_c x + b_c0 + r o (U_c h + b_c1));  h = (1 - z) o c + z o h
// PARITY UNPINNED against the reference; the tests check it against a CPU restatement and torch.nn.GRU.
//
// Mapping: one wave per stream, lane r owns gate row r of every layer (3H <= 64 rows: z | r | c), weights in VGPRs for
// the whole block, [x; h] broadcast from LDS, the z / r gates meet the c rows through LDS, the dense head is computed for
// the whole block after the sample loop (same structure as LstmWaveKernel).
#include "device_once.h"
#include "tuning.h"
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "dpp_recurrent.h"
#include "lstm_dev.h"
#include "lstm_launch.h"
#include "recurrent_tail.h"

namespace na
{
	namespace
	{

		__device__ __forceinline__ void GruWaveSync()
		{
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
		}

		// this lane's row of one layer: packed per layer as W row-major [3H][I + H], then b_in[3H], then b_rec[3H]
		template <int H, int I>
		struct GruRow
		{
			float wi[I], wh[H], bi, bh;

			__device__ __forceinline__ void Load(const float* __restrict__ w, int r)
			{
				const bool valid = r < 3 * H;
#pragma unroll
				for (int k = 0; k < I; k++) wi[k] = valid ? w[(size_t)r * (I + H) + k] : 0.0f;
#pragma unroll
				for (int k = 0; k < H; k++) wh[k] = valid ? w[(size_t)r * (I + H) + I + k] : 0.0f;
				bi = valid ? w[(size_t)3 * H * (I + H) + r] : 0.0f;
				bh = valid ? w[(size_t)3 * H * (I + H) + 3 * H + r] : 0.0f;
			}
		};

		// one layer, one sample: xin[I] and h[H] are LDS vectors, zr[2H] an LDS scratch; h is updated in place
		template <int H, int I>
		__device__ __forceinline__ void GruLayerStep(const GruRow<H, I>& row, const float* xin, float* h, float* zr, int lane)
		{
			float ai = row.bi, ah = row.bh;
#pragma unroll
			for (int k = 0; k < I; k++) ai += row.wi[k] * xin[k];
#pragma unroll
			for (int k = 0; k < H; k++) ah += row.wh[k] * h[k];
			if (lane < 2 * H) zr[lane] = GruSigmoid(ai + ah);
			GruWaveSync(); // every lane has read the old h; z and r are visible
			if (lane >= 2 * H && lane < 3 * H)
			{
				const int u = lane - 2 * H;
				const float c = GruTanh(ai + zr[H + u] * ah);
				const float z = zr[u];
				h[u] = (1.0f - z) * c + z * h[u];
			}
			GruWaveSync();
		}

		// grid = active streams, block = 64 (one wave per stream).  L in {1, 2}; 3H <= 64.
		template <int H, int L>
		__global__ void __launch_bounds__(64) GruWaveKernel(LstmModelDev m, float* __restrict__ state, int capacity, const int* __restrict__ slots,
			const int* __restrict__ rows, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n)
		{
			static_assert(3 * H <= 64, "one gate row per lane");
			constexpr int HP = H + 1;
			__shared__ float xin[LSTM_MAX_FRAMES];
			__shared__ float hvec[L][H];
			__shared__ float zr[2 * H];
			__shared__ float hout[LSTM_MAX_FRAMES * HP];

			const int lane = threadIdx.x;
			const int slot = slots[blockIdx.x];
			const int row = rows[blockIdx.x];
			const float* inRow = in + (size_t)row * inStride;
			float* outRow = out + (size_t)row * outStride;

			GruRow<H, 1> row0;
			row0.Load(m.w + m.layerOff[0], lane);
			GruRow<H, H> row1;
			if (L > 1) row1.Load(m.w + m.layerOff[L > 1 ? 1 : 0], lane);

			for (int f = lane; f < n; f += 64) xin[f] = inRow[f];
#pragma unroll
			for (int l = 0; l < L; l++)
				if (lane < H) hvec[l][lane] = state[(size_t)(l * 2 * H + lane) * capacity + slot];
			GruWaveSync();

			for (int f = 0; f < n; f++)
			{
				GruLayerStep<H, 1>(row0, xin + f, hvec[0], zr, lane);
				if (L > 1) GruLayerStep<H, H>(row1, hvec[0], hvec[L > 1 ? 1 : 0], zr, lane);
				if (lane < H) hout[f * HP + lane] = hvec[L - 1][lane];
			}
			GruWaveSync();

			// dense head for the whole block, lane = sample
			const float* headW = m.w + m.headOff;
			for (int f = lane; f < n; f += 64)
			{
				float acc = 0.0f;
#pragma unroll
				for (int k = 0; k < H; k++) acc += headW[k] * hout[f * HP + k];
				outRow[f] = acc + headW[H];
			}
#pragma unroll
			for (int l = 0; l < L; l++)
				if (lane < H) state[(size_t)(l * 2 * H + lane) * capacity + slot] = hvec[l][lane];
		}


		// Any hidden size / layer count (what RTNeural's run-time model accepts): lane = stream, run-time loops, weights through
		// wave-uniform loads.  LDS: io[64][n + 1] | h[numLayers][H][64] | a[2][3H][64] (input and recurrent pre-activations).
		__global__ void __launch_bounds__(64) GruGenericKernel(LstmModelDev m, float* __restrict__ state, int capacity, const int* __restrict__ slots,
			const int* __restrict__ rows, int numStreams, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n)
		{
			extern __shared__ __attribute__((aligned(16))) float lds[];
			const int H = m.hidden;
			float* io = lds;
			const int ioStride = n + 1;
			float* hAll = lds + 64 * ioStride;
			float* ai = hAll + (size_t)m.numLayers * H * 64;
			float* ah = ai + (size_t)3 * H * 64;
			float* tailA = ah + (size_t)3 * H * 64; // generic keras stack only: two [tailWidth][64] arrays
			float* tailB = tailA + (size_t)m.tailWidth * 64;
			const int lane = threadIdx.x;
			const int idx = blockIdx.x * 64 + lane;
			const bool active = idx < numStreams;
			const int slot = active ? slots[idx] : 0;
			for (int r = 0; r < 64; r++)
			{
				const int ridx = blockIdx.x * 64 + r;
				if (ridx < numStreams)
				{
					const float* src = in + (size_t)rows[ridx] * inStride;
					for (int f = lane; f < n; f += 64) io[r * ioStride + f] = src[f];
				}
			}
			for (int l = 0; l < m.numLayers; l++)
				for (int k = 0; k < H; k++) hAll[(l * H + k) * 64 + lane] = active ? state[(size_t)(l * 2 * H + k) * capacity + slot] : 0.0f;
			__syncthreads();
			const float* headW = m.w + m.headOff;
			for (int f = 0; f < n; f++)
			{
				const float x0 = io[lane * ioStride + f];
				for (int l = 0; l < m.numLayers; l++)
				{
					const int I = (l == 0) ? 1 : H, W = I + H;
					const float* w = m.w + m.layerOff[l];
					const float* bIn = w + (size_t)3 * H * W;
					const float* bRec = bIn + 3 * H;
					float* h = hAll + (size_t)l * H * 64;
					const float* below = hAll + (size_t)(l > 0 ? l - 1 : 0) * H * 64;
					for (int r = 0; r < 3 * H; r++)
					{
						const float* row = w + (size_t)r * W;
						float a = bIn[r], b = bRec[r];
						if (l == 0) a += row[0] * x0;
						else
							for (int k = 0; k < H; k++) a += row[k] * below[k * 64 + lane];
						for (int k = 0; k < H; k++) b += row[I + k] * h[k * 64 + lane];
						ai[r * 64 + lane] = a;
						ah[r * 64 + lane] = b;
					}
					for (int u = 0; u < H; u++)
					{
						const float z = GruSigmoid(ai[u * 64 + lane] + ah[u * 64 + lane]);
						const float rr = GruSigmoid(ai[(H + u) * 64 + lane] + ah[(H + u) * 64 + lane]);
						const float c = GruTanh(ai[(2 * H + u) * 64 + lane] + rr * ah[(2 * H + u) * 64 + lane]);
						h[u * 64 + lane] = (1.0f - z) * c + z * h[u * 64 + lane];
					}
				}
				const float* hl = hAll + (size_t)(m.numLayers - 1) * H * 64;
				if (m.tailLayers > 0)
				{
					io[lane * ioStride + f] = DenseTail(m, hl, H, x0, tailA, tailB, lane);
					continue;
				}
				float acc = 0.0f;
				for (int k = 0; k < H; k++) acc += headW[k] * hl[k * 64 + lane];
				io[lane * ioStride + f] = acc + headW[H];
			}
			__syncthreads();
			for (int l = 0; l < m.numLayers; l++)
				for (int k = 0; k < H; k++)
					if (active) state[(size_t)(l * 2 * H + k) * capacity + slot] = hAll[(l * H + k) * 64 + lane];
			for (int r = 0; r < 64; r++)
			{
				const int ridx = blockIdx.x * 64 + r;
				if (ridx < numStreams)
				{
					float* dst = out + (size_t)rows[ridx] * outStride;
					for (int f = lane; f < n; f += 64) dst[f] = io[r * ioStride + f];
				}
			}
		}

		hipError_t LaunchGruGeneric(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams, const float* in,
			float* out, long inStride, long outStride, int n, hipStream_t stream)
		{
			const size_t ldsBytes = ((size_t)64 * (n + 1) + (size_t)m.numLayers * m.hidden * 64 + (size_t)6 * m.hidden * 64 + (size_t)2 * (m.tailLayers > 0 ? m.tailWidth : 0) * 64) * sizeof(float);
			if (ldsBytes > 160 * 1024) return hipErrorInvalidValue;
			static PerDeviceOnce attr; // (hipFuncSetAttribute applies to the current device's copy of the kernel)
			(void)attr.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&GruGenericKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
			hipLaunchKernelGGL(GruGenericKernel, dim3((unsigned)((numStreams + 63) / 64)), dim3(64), ldsBytes, stream, m, state, capacity, slots, rows,
				numStreams, in, out, inStride, outStride, n);
			return hipGetLastError();
		}

		template <int H, int L>
		hipError_t LaunchHL(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams, const float* in,
			float* out, long inStride, long outStride, int n, hipStream_t stream)
		{
			hipLaunchKernelGGL((GruWaveKernel<H, L>), dim3((unsigned)numStreams), dim3(64), 0, stream, m, state, capacity, slots, rows, in, out, inStride,
				outStride, n);
			return hipGetLastError();
		}
	}

	hipError_t LaunchGruBlock(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams, const float* in,
		float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		if (n > LSTM_MAX_FRAMES || !GruShapeSupported(m.hidden, m.numLayers, m.tailLayers > 0 ? m.tailWidth : 0, m.tailLayers > 0 ? m.tailHistMax : 0)) return hipErrorInvalidValue;
		if (m.tailLayers > 0)
		{
			// generic keras stack: the runtime-shaped wave kernel if the weights fit the LDS, else the lane = stream kernel
			hipError_t err = hipSuccess;
			if (LaunchRecurrentWaveRt(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream, err)) return err;
			if (m.tailHistMax > 0) return hipErrorNotSupported; // (conv1d tails: the runtime-shaped wave kernel only)
			return LaunchGruGeneric(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		}
		const bool noDpp = Tuning::Get().gruNoDpp; // tuning knob: the LDS-broadcast kernel for every shape
		if (!noDpp && RecurrentDppSupported(m))
		{
			const RecurrentGroup g = { m, state, capacity, slots, rows, numStreams };
			return LaunchRecurrentDpp(&g, 1, in, out, inStride, outStride, n, stream);
		}
#define NA_GRU_CASE(HH) \
	if (m.hidden == HH && m.numLayers <= 2) return m.numLayers == 1 ? LaunchHL<HH, 1>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream) \
												: LaunchHL<HH, 2>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		NA_GRU_CASE(8)
		NA_GRU_CASE(12)
		NA_GRU_CASE(16)
		NA_GRU_CASE(20)
#undef NA_GRU_CASE
		{
			hipError_t err = hipSuccess;
			if (LaunchRecurrentWaveRt(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream, err)) return err;
		}
		return LaunchGruGeneric(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
	}
}
