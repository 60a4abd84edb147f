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
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "dpp_recurrent.h"
#include "lstm_dev.h"
#include "lstm_launch.h"

namespace na
{
	namespace
	{
		__device__ __forceinline__ float GruSigmoid(float x) { return (tanhf(x * 0.5f) + 1.0f) * 0.5f; }

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
				const float c = tanhf(ai + zr[H + u] * ah);
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

		// ------------------------------------------------------------------------------------------------------------
		// H = 8 / 16: nothing on the recurrence touches LDS (same idea as LstmDppKernel).  lane = H*gate + unit with gate rows z, r, c
		// and the fourth row duplicating c (for H = 8 the upper 32 lanes mirror the lower 32); every lane keeps h[unit].  The mat-vec
		// reads h[(unit - n) mod H] with DPP row_ror:n against weights rotated at load time; z and r reach every lane through two
		// lane swaps, c through one.  Each lane sums its row starting at column `unit` (different rounding order than the plain
		// kernel; ~1e-7 RMS).
		// ------------------------------------------------------------------------------------------------------------
		template <int H>
		__device__ __forceinline__ float GruDppCell(float ai, float ah, float h)
		{
			int zr = __builtin_bit_cast(int, GruSigmoid(ai + ah)); // meaningful on the z and r rows
			int zr2 = zr;
			float z, r;
			if constexpr (H == 16)
			{
				LaneSwap32(zr, zr2); // zr: rows z r z r
				int rr = zr;
				LaneSwap16(zr, rr);  // zr: z everywhere, rr: r everywhere
				z = __builtin_bit_cast(float, zr);
				r = __builtin_bit_cast(float, rr);
			}
			else
			{
				LaneSwap16(zr, zr2); // zr: every row = [z | r]  (rows 0 and 2 hold it, the upper half of the wave mirrors the lower)
				z = __builtin_bit_cast(float, RowLowHalf(zr));
				r = __builtin_bit_cast(float, RowHighHalf(zr));
			}
			int c = __builtin_bit_cast(int, tanhf(ai + r * ah)); // meaningful on the c rows (2 and 3 for H = 16; 1 and 3 for H = 8)
			int c2 = c;
			if constexpr (H == 16) LaneSwap32(c, c2); // c2: rows c c c c
			else LaneSwap16(c, c2);                   // c2: rows 1 1 3 3 = c everywhere
			const float cv = __builtin_bit_cast(float, c2);
			return (1.0f - z) * cv + z * h;
		}

		template <int H, int L>
		__global__ void __launch_bounds__(64) GruDppKernel(LstmModelDev m, float* __restrict__ state, int capacity, const int* __restrict__ slots,
			const int* __restrict__ rows, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n)
		{
			static_assert(H == 8 || H == 16, "a 16-lane DPP row must hold the units a whole number of times");
			constexpr int HP = H + 1;
			__shared__ float xin[LSTM_MAX_FRAMES];
			__shared__ float hout[LSTM_MAX_FRAMES * HP];

			const int lane = threadIdx.x;
			const int unit = lane % H;
			const int gate = min((lane / H) & 3, 2); // rows z, r, c, c
			const int r = gate * H + unit;
			const int slot = slots[blockIdx.x];
			const int row = rows[blockIdx.x];
			const float* inRow = in + (size_t)row * inStride;
			float* outRow = out + (size_t)row * outStride;

			// layer 0: W row-major [3H][1 + H], b_in[3H], b_rec[3H]; h weights rotated so that row_ror:k pairs wh0[k] with h[(unit - k) mod H]
			const float* w0 = m.w + m.layerOff[0];
			const float wx0 = w0[(size_t)r * (1 + H)];
			float wh0[H];
#pragma unroll
			for (int k = 0; k < H; k++) wh0[k] = w0[(size_t)r * (1 + H) + 1 + ((unit - k + H) % H)];
			const float bi0 = w0[(size_t)3 * H * (1 + H) + r], bh0 = w0[(size_t)3 * H * (1 + H) + 3 * H + r];
			float wi1[H], wh1[H];
			float bi1 = 0.0f, bh1 = 0.0f;
			if (L > 1)
			{
				const float* w1 = m.w + m.layerOff[L > 1 ? 1 : 0];
#pragma unroll
				for (int k = 0; k < H; k++)
				{
					wi1[k] = w1[(size_t)r * (2 * H) + ((unit - k + H) % H)];
					wh1[k] = w1[(size_t)r * (2 * H) + H + ((unit - k + H) % H)];
				}
				bi1 = w1[(size_t)3 * H * (2 * H) + r];
				bh1 = w1[(size_t)3 * H * (2 * H) + 3 * H + r];
			}

			for (int f = lane; f < n; f += 64) xin[f] = inRow[f];
			float h[L];
#pragma unroll
			for (int l = 0; l < L; l++) h[l] = state[(size_t)(l * 2 * H + unit) * capacity + slot];
			GruWaveSync();

			float x = xin[0];
			for (int f = 0; f < n; f++)
			{
				const float xNext = xin[(f + 1 < n) ? f + 1 : f]; // off the recurrence: fetched a step ahead
				float ah = bh0;
				DppDot<H>(ah, wh0, h[0]);
				h[0] = GruDppCell<H>(wx0 * x + bi0, ah, h[0]);
				if (L > 1)
				{
					float ai1 = bi1, ah1 = bh1;
					DppDot<H>(ai1, wi1, h[0]);
					DppDot<H>(ah1, wh1, h[L > 1 ? 1 : 0]);
					h[L > 1 ? 1 : 0] = GruDppCell<H>(ai1, ah1, h[L > 1 ? 1 : 0]);
				}
				if (lane < H) hout[f * HP + lane] = h[L - 1];
				x = xNext;
			}
			GruWaveSync();

			const float* headW = m.w + m.headOff;
			for (int f = lane; f < n; f += 64)
			{
				float acc = 0.0f;
#pragma unroll
				for (int k = 0; k < H; k++) acc += headW[k] * hout[f * HP + k];
				outRow[f] = acc + headW[H];
			}
			if (lane < H)
			{
#pragma unroll
				for (int l = 0; l < L; l++) state[(size_t)(l * 2 * H + lane) * capacity + slot] = h[l];
			}
		}

		template <int H, int L>
		hipError_t LaunchDppHL(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams, const float* in,
			float* out, long inStride, long outStride, int n, hipStream_t stream)
		{
			hipLaunchKernelGGL((GruDppKernel<H, L>), dim3((unsigned)numStreams), dim3(64), 0, stream, m, state, capacity, slots, rows, in, out, inStride,
				outStride, n);
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

	bool GruShapeSupported(int hidden, int numLayers)
	{
		return (hidden == 8 || hidden == 12 || hidden == 16 || hidden == 20) && (numLayers == 1 || numLayers == 2);
	}

	hipError_t LaunchGruBlock(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams, const float* in,
		float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		if (n > LSTM_MAX_FRAMES || !GruShapeSupported(m.hidden, m.numLayers)) return hipErrorInvalidValue;
		static const bool noDpp = getenv("NA_GRU_NO_DPP") != nullptr; // tuning knob: the LDS-broadcast kernel for every shape
		if (!noDpp && (m.hidden == 8 || m.hidden == 16))
		{
			if (m.hidden == 8)
				return m.numLayers == 1 ? LaunchDppHL<8, 1>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream)
										: LaunchDppHL<8, 2>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
			return m.numLayers == 1 ? LaunchDppHL<16, 1>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream)
									: LaunchDppHL<16, 2>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		}
#define NA_GRU_CASE(HH) \
	if (m.hidden == HH) return m.numLayers == 1 ? LaunchHL<HH, 1>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream) \
												: LaunchHL<HH, 2>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		NA_GRU_CASE(8)
		NA_GRU_CASE(12)
		NA_GRU_CASE(16)
		NA_GRU_CASE(20)
#undef NA_GRU_CASE
		return hipErrorInvalidValue;
	}
}
