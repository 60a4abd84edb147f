// lstm_kernels.hip -- gfx950 kernels for the LSTM recurrent gate loop.
//
// Replaces, for many independent streams at once:
//   LSTMModelT::Process   NeuralAudio/LSTM.h:164-191   (sample loop, layer chain, dense head)
//   LSTMLayerT::Process   NeuralAudio/LSTM.h:87-100    (g = W[4H x (I+H)] [x;h] + b; i,f,g,o gates)
//   LSTMLayer::Process    NeuralAudio/LSTMDynamic.h:95-108 (same arithmetic, runtime shaped)
//   FastMath Tanh/Sigmoid NeuralAudio/Activation.h:83-96
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
	// Activation.h:83-91
	__device__ __forceinline__ float LstmFastTanh(float x)
	{
		const float ax = fabsf(x);
		const float x2 = x * x;
		const float num = x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2);
		const float den = 2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax);
		return num / den;
	}

	// Activation.h:93-96
	__device__ __forceinline__ float LstmFastSigmoid(float x) { return 0.5f * (LstmFastTanh(x * 0.5f) + 1.0f); }

	// the reference's LSTM_MATH build option as a wave-uniform run-time switch (these kernels are the fallback shapes; the H = 8 / 16
	// kernels in recurrent_dpp_kernels.hip carry the policy as a template parameter)
	__device__ __forceinline__ float LstmTanh(float x, int math) { return math == LSTM_MATH_STD ? StdTanh(x) : LstmFastTanh(x); }
	__device__ __forceinline__ float LstmSigmoid(float x, int math) { return math == LSTM_MATH_STD ? StdSigmoid(x) : LstmFastSigmoid(x); }

	// One layer step for one stream (lane).  state = [x (I values); h (H values)] in registers.
	//   cell/hidden columns live in LDS: hc[k * 64 + lane]
	template <int H, int I>
	__device__ __forceinline__ void LstmLayerStep(const float* __restrict__ w, const float (&xin)[I], float* hc, int lane, int math)
	{
		constexpr int W = I + H;
		float s[W];
#pragma unroll
		for (int k = 0; k < I; k++) s[k] = xin[k];
#pragma unroll
		for (int k = 0; k < H; k++) s[I + k] = hc[k * 64 + lane];

		const float* bias = w + 4 * H * W;
		// s[] holds the pre-update hidden state, so h/c can be updated in place unit by unit
#pragma unroll 4
		for (int i = 0; i < H; i++)
		{
			float gi = 0.0f, gf = 0.0f, gg = 0.0f, go = 0.0f;
			const float* ri = w + (size_t)(0 * H + i) * W;
			const float* rf = w + (size_t)(1 * H + i) * W;
			const float* rg = w + (size_t)(2 * H + i) * W;
			const float* ro = w + (size_t)(3 * H + i) * W;
#pragma unroll
			for (int k = 0; k < W; k++)
			{
				gi += ri[k] * s[k];
				gf += rf[k] * s[k];
				gg += rg[k] * s[k];
				go += ro[k] * s[k];
			}
			gi += bias[0 * H + i];
			gf += bias[1 * H + i];
			gg += bias[2 * H + i];
			go += bias[3 * H + i];
			// LSTM.h:94-99
			const float c = (LstmSigmoid(gf, math) * hc[(H + i) * 64 + lane]) + (LstmSigmoid(gi, math) * LstmTanh(gg, math));
			hc[(H + i) * 64 + lane] = c;
			hc[i * 64 + lane] = LstmSigmoid(go, math) * LstmTanh(c, math);
		}
	}

	// grid = ceil(active/64), block = 64.  lane = stream.
	template <int H>
	__global__ void __launch_bounds__(64) LstmBlockKernel(LstmModelDev m, float* __restrict__ state, int capacity,
		const int* __restrict__ slots, const int* __restrict__ rows, int numStreams, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n)
	{
		extern __shared__ __attribute__((aligned(16))) float lds[];
		// lds: io[64][n + 1] then hc[numLayers][2H][64]
		float* io = lds;
		const int ioStride = n + 1;
		float* hcAll = lds + 64 * ioStride;

		const int lane = threadIdx.x;
		const int idx = blockIdx.x * 64 + lane;
		const bool active = idx < numStreams;
		const int slot = active ? slots[idx] : 0;

		// stage the input tile [64 streams][n] through LDS so global reads are row-contiguous
		for (int r = 0; r < 64; r++)
		{
			const int ridx = blockIdx.x * 64 + r;
			if (ridx < numStreams)
			{
				const float* src = in + (size_t)rows[ridx] * inStride;
				for (int f = lane; f < n; f += 64) io[r * ioStride + f] = src[f];
			}
		}
		// load h, c
		for (int l = 0; l < m.numLayers; l++)
			for (int k = 0; k < 2 * H; k++)
				hcAll[(l * 2 * H + k) * 64 + lane] = active ? state[(size_t)(l * 2 * H + k) * capacity + slot] : 0.0f;
		__syncthreads();

		const float* headW = m.w + m.headOff;
		for (int f = 0; f < n; f++)
		{
			float x1[1] = { io[lane * ioStride + f] };
			LstmLayerStep<H, 1>(m.w + m.layerOff[0], x1, hcAll, lane, m.math); // LSTM.h:168
			for (int l = 1; l < m.numLayers; l++)
			{
				float xh[H];
#pragma unroll
				for (int k = 0; k < H; k++) xh[k] = hcAll[((l - 1) * 2 * H + k) * 64 + lane];
				LstmLayerStep<H, H>(m.w + m.layerOff[l], xh, hcAll + (size_t)l * 2 * H * 64, lane, m.math); // LSTM.h:170-180
			}
			const float* hl = hcAll + (size_t)(m.numLayers - 1) * 2 * H * 64;
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < H; k++) acc += headW[k] * hl[k * 64 + lane];
			io[lane * ioStride + f] = acc + headW[H]; // LSTM.h:182-189
		}
		__syncthreads();

		for (int l = 0; l < m.numLayers; l++)
			for (int k = 0; k < 2 * H; k++)
				if (active) state[(size_t)(l * 2 * H + k) * capacity + slot] = hcAll[(l * 2 * H + k) * 64 + lane];
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

	// ------------------------------------------------------------------------------------------------------------
	// Fast path: ONE WAVE PER STREAM, the 4H gate rows of a layer spread over the 64 lanes (lane r owns rows r, r+64, ..),
	// each lane keeps its rows of W (and bias) in VGPRs for the whole block, the state vector [x; h] is broadcast from
	// LDS, gates are exchanged through LDS, lanes i < H own unit i (cell state in a register).  The dense head is
	// taken off the serial path: h_last[t] is parked in LDS and all outputs are computed after the sample loop.
	// Same arithmetic and summation order as LstmLayerStep (and the reference, LSTM.h:87-100).
	// ------------------------------------------------------------------------------------------------------------
	template <int H, int I>
	struct LstmRows
	{
		static constexpr int RPL = (4 * H + 63) / 64; // rows per lane
		float w[RPL][I + H];
		float b[RPL];
	};

	template <int H, int I>
	__device__ __forceinline__ void LoadRows(LstmRows<H, I>& rows, const float* __restrict__ w, int lane)
	{
		constexpr int W = I + H;
#pragma unroll
		for (int q = 0; q < LstmRows<H, I>::RPL; q++)
		{
			const int r = lane + 64 * q;
			const bool valid = r < 4 * H;
#pragma unroll
			for (int k = 0; k < W; k++) rows.w[q][k] = valid ? w[(size_t)r * W + k] : 0.0f;
			rows.b[q] = valid ? w[(size_t)4 * H * W + r] : 0.0f;
		}
	}

	// gates of one layer for this sample: s = [x (I values from `xin`), h (H values from `hvec`)] broadcast from LDS
	template <int H, int I>
	__device__ __forceinline__ void GateRows(const LstmRows<H, I>& rows, const float* xin, const float* hvec, float* gates, int lane, int math)
	{
		constexpr int W = I + H;
		float sv[W];
#pragma unroll
		for (int k = 0; k < I; k++) sv[k] = xin[k];
#pragma unroll
		for (int k = 0; k < H; k++) sv[I + k] = hvec[k];
#pragma unroll
		for (int q = 0; q < LstmRows<H, I>::RPL; q++)
		{
			const int r = lane + 64 * q;
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < W; k++) acc += rows.w[q][k] * sv[k];
			acc += rows.b[q];
			// rows [2H, 3H) are the cell candidate (tanh), the others sigmoid = 0.5*(tanh(0.5 x) + 1)  (LSTM.h:33-36,94-99)
			const bool isG = (r >= 2 * H) && (r < 3 * H);
			float gv;
			if (math == LSTM_MATH_STD) gv = isG ? StdTanh(acc) : StdSigmoid(acc); // Activation.h:37-45
			else
			{
				const float t = LstmFastTanh(isG ? acc : acc * 0.5f);
				gv = isG ? t : 0.5f * (t + 1.0f);
			}
			if (r < 4 * H) gates[r] = gv;
		}
	}

	__device__ __forceinline__ void LstmWaveSync()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
	}

	// grid = active streams, block = 64 (one wave per stream).  L in {1, 2}.
	template <int H, int L>
	__global__ void __launch_bounds__(64) LstmWaveKernel(LstmModelDev m, float* __restrict__ state, int capacity, const int* __restrict__ slots,
		const int* __restrict__ rows, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n)
	{
		constexpr int HP = H + 1; // padded row of the parked hidden states
		__shared__ float xin[LSTM_MAX_FRAMES];
		__shared__ float hvec[L][H];
		__shared__ float gates[4 * H];
		__shared__ float hout[LSTM_MAX_FRAMES * HP];

		const int lane = threadIdx.x;
		const int slot = slots[blockIdx.x];
		const int row = rows[blockIdx.x];
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;

		LstmRows<H, 1> rows0;
		LoadRows<H, 1>(rows0, m.w + m.layerOff[0], lane);
		LstmRows<H, H> rows1;
		if (L > 1) LoadRows<H, H>(rows1, m.w + m.layerOff[L > 1 ? 1 : 0], lane);

		for (int f = lane; f < n; f += 64) xin[f] = inRow[f];
		float c[L];
#pragma unroll
		for (int l = 0; l < L; l++)
		{
			c[l] = 0.0f;
			if (lane < H)
			{
				hvec[l][lane] = state[(size_t)(l * 2 * H + lane) * capacity + slot];
				c[l] = state[(size_t)(l * 2 * H + H + lane) * capacity + slot];
			}
		}
		LstmWaveSync();

		for (int f = 0; f < n; f++)
		{
			GateRows<H, 1>(rows0, xin + f, hvec[0], gates, lane, m.math); // LSTM.h:168
			LstmWaveSync();
			if (lane < H)
			{
				// LSTM.h:94-99
				c[0] = (gates[H + lane] * c[0]) + (gates[lane] * gates[2 * H + lane]);
				const float h = gates[3 * H + lane] * LstmTanh(c[0], m.math);
				hvec[0][lane] = h;
				if (L == 1) hout[f * HP + lane] = h;
			}
			LstmWaveSync();
			if (L > 1)
			{
				GateRows<H, H>(rows1, hvec[0], hvec[L > 1 ? 1 : 0], gates, lane, m.math); // LSTM.h:170-180
				LstmWaveSync();
				if (lane < H)
				{
					c[L - 1] = (gates[H + lane] * c[L - 1]) + (gates[lane] * gates[2 * H + lane]);
					const float h = gates[3 * H + lane] * LstmTanh(c[L - 1], m.math);
					hvec[L > 1 ? 1 : 0][lane] = h;
					hout[f * HP + lane] = h;
				}
				LstmWaveSync();
			}
		}

		// dense head for the whole block, lane = sample (LSTM.h:182-189)
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
		{
			if (lane < H)
			{
				state[(size_t)(l * 2 * H + lane) * capacity + slot] = hvec[l][lane];
				state[(size_t)(l * 2 * H + H + lane) * capacity + slot] = c[l];
			}
		}
	}

	template <int H, int L>
	static hipError_t LaunchWaveHL(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		hipLaunchKernelGGL((LstmWaveKernel<H, L>), dim3((unsigned)numStreams), dim3(64), 0, stream, m, state, capacity, slots, rows, in, out,
			inStride, outStride, n);
		return hipGetLastError();
	}

	// returns false when (H, layers) has no wave-per-stream instance (the lane-per-stream kernel is used instead)
	static bool LaunchLstmWave(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream, hipError_t& err)
	{
		const bool noDpp = Tuning::Get().lstmNoDpp; // tuning knob: fall back to the LDS-broadcast wave kernel
		if (!noDpp && RecurrentDppSupported(m))
		{
			const RecurrentGroup g = { m, state, capacity, slots, rows, numStreams };
			err = LaunchRecurrentDpp(&g, 1, in, out, inStride, outStride, n, stream);
			return true;
		}
#define NA_LSTM_WAVE(HH) \
	if (m.hidden == HH && m.numLayers == 1) { err = LaunchWaveHL<HH, 1>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream); return true; } \
	if (m.hidden == HH && m.numLayers == 2) { err = LaunchWaveHL<HH, 2>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream); return true; }
		NA_LSTM_WAVE(8)
		NA_LSTM_WAVE(12)
		NA_LSTM_WAVE(16)
		NA_LSTM_WAVE(20)
		NA_LSTM_WAVE(24)
		NA_LSTM_WAVE(32)
#undef NA_LSTM_WAVE
		return false;
	}

	// ------------------------------------------------------------------------------------------------------------
	// The same mapping with run-time shapes: ONE WAVE PER STREAM for any hidden size up to 64 and any layer count whose weights fit
	// the LDS -- 1x18, 3x16, 3x24, 2x40 ... (what LSTMDynamic.h:95-108 accepts; the lane = stream kernels below take these shapes
	// too but need 6-250 ms per 128-sample block, this one 0.1-1 ms), LSTM or keras GRU cells (RecurrentWaveRtKernel serves
	// gru_kernels.hip as well), with the classic 1-unit head or the dense chain of a generic keras stack (also without any recurrent
	// layer).  The gate rows of a layer (4H / 3H) are spread over the 64 lanes (lane r owns rows r, r + 64, ...); the weights of ALL
	// layers are copied to LDS once per block (row stride padded to an odd number of floats: lanes read different rows at the same
	// column without bank conflicts), the state vector [x; h] is broadcast from LDS, gates are exchanged through LDS, lanes i < H own
	// unit i.  Same arithmetic and summation order as LstmLayerStep / GruLayerStep.  The head or the dense chain runs after the
	// recurrence for the whole block with lane = sample (no dependence between samples there): the h of the last layer is kept as
	// [pass][k][64] (sample = 64 pass + lane), the layout DenseTail reads.
	// LDS: xin[128] | hvec[L][H] | cvec[L][H] | gates[6H] | hseq[2][Hs][64] | tail scratch 2 x [tailWidth][64] | w[all layers: padded rows | biases]
	// ------------------------------------------------------------------------------------------------------------
	__device__ __forceinline__ int OddStride(int w) { return w | 1; }

	static size_t RecurrentWaveRtLdsFloats(const LstmModelDev& m, bool weightsInLds = true)
	{
		const int H = m.hidden, L = m.numLayers;
		const int rowsPerLayer = (m.cell == LSTM_CELL_GRU ? 3 : 4) * H, biases = (m.cell == LSTM_CELL_GRU ? 6 : 4) * H;
		const bool hseq = H < RECURRENT_HEAD_IN_LOOP_FROM || m.tailLayers > 0; // (else the head is evaluated inside the sample loop)
		size_t f = (size_t)LSTM_MAX_FRAMES + (size_t)2 * L * H + (size_t)6 * H + (hseq ? (size_t)2 * (L > 0 ? H : 1) * 64 : 0) +
			(size_t)RecurrentTailScratchFloats(m.tailLayers > 0 ? m.tailWidth : 0, m.tailLayers > 0 ? m.tailHistMax : 0);
		if (weightsInLds)
			for (int l = 0; l < L; l++) f += (size_t)rowsPerLayer * (size_t)(((l == 0 ? 1 : H) + H) | 1) + (size_t)biases;
		return f;
	}

	// The dot products of the lane's gate rows r = lane + 64 i (i < RPL) from the transposed L2-resident weights (LstmModelDev::wT), all
	// rows of the lane side by side: per quad of inputs RPL independent 1 KB weight loads per wave and one set of broadcast state reads
	// (the rows of a lane were evaluated one after the other at first: LSTM 2x64 2.97 ms per block; side by side the loads of all rows
	// are in flight together).  Input part and hidden part apart (the GRU keeps them apart; the LSTM adds them).  Same term order per
	// row as the LDS path: k = 0 .. I-1, then 0 .. H-1 (padding terms are zeros).
	typedef float rt_f4 __attribute__((ext_vector_type(4)));
	template <int RPL>
	__device__ __forceinline__ void RowsDotL2(const float* __restrict__ wTl, int rowsPad, int lane, int nt, int I, int H, const float* sIn, const float* sH, float (&ai)[RPL],
		float (&ah)[RPL])
	{
		// (lane = thread of the stream's workgroup, nt = its size: row r = lane + nt i)
		const int Qi = (I + 3) / 4, Qh = (H + 3) / 4;
		const rt_f4* wq = reinterpret_cast<const rt_f4*>(wTl) + lane;
		for (int q = 0; q < Qi; q++)
		{
			const float s0 = sIn[4 * q], s1 = (4 * q + 1 < I) ? sIn[4 * q + 1] : 0.0f, s2 = (4 * q + 2 < I) ? sIn[4 * q + 2] : 0.0f, s3 = (4 * q + 3 < I) ? sIn[4 * q + 3] : 0.0f;
#pragma unroll
			for (int i = 0; i < RPL; i++)
			{
				const rt_f4 w4 = wq[(size_t)q * rowsPad + nt * i];
				ai[i] += w4.x * s0;
				ai[i] += w4.y * s1;
				ai[i] += w4.z * s2;
				ai[i] += w4.w * s3;
			}
		}
		wq += (size_t)Qi * rowsPad;
#ifndef NA_REC_UNROLL
#define NA_REC_UNROLL 4 // (quads of hidden inputs whose weight loads are in flight together; 8 spills at 128 VGPRs: 4 x slower)
#endif
#pragma unroll NA_REC_UNROLL
		for (int q = 0; q < Qh; q++)
		{
			const float s0 = sH[4 * q], s1 = (4 * q + 1 < H) ? sH[4 * q + 1] : 0.0f, s2 = (4 * q + 2 < H) ? sH[4 * q + 2] : 0.0f, s3 = (4 * q + 3 < H) ? sH[4 * q + 3] : 0.0f;
#pragma unroll
			for (int i = 0; i < RPL; i++)
			{
				const rt_f4 w4 = wq[(size_t)q * rowsPad + nt * i];
				ah[i] += w4.x * s0;
				ah[i] += w4.y * s1;
				ah[i] += w4.z * s2;
				ah[i] += w4.w * s3;
			}
		}
	}

	// gate pre-activations of one layer from L2-streamed weights -> gates[] (LSTM: activated; GRU: ai | ah), RPL = rows per lane
	template <int RPL>
	__device__ __forceinline__ void GateRowsL2(const LstmModelDev& m, int l, int lane, int nt, const float* sIn, const float* sH, float* gates)
	{
		const int H = m.hidden, I = (l == 0) ? 1 : H, W = I + H;
		const bool gru = m.cell == LSTM_CELL_GRU;
		const int rows = (gru ? 3 : 4) * H;
		const float* bias = m.w + m.layerOff[l] + (size_t)rows * W;
		float ai[RPL], ah[RPL];
#pragma unroll
		for (int i = 0; i < RPL; i++)
		{
			const int r = lane + nt * i;
			ai[i] = (gru && r < rows) ? bias[r] : 0.0f;
			ah[i] = (gru && r < rows) ? bias[3 * H + r] : 0.0f;
		}
		RowsDotL2<RPL>(m.wT + m.layerOffT[l], m.rowsPad, lane, nt, I, H, sIn, sH, ai, ah);
#pragma unroll
		for (int i = 0; i < RPL; i++)
		{
			const int r = lane + nt * i;
			if (r >= rows) continue;
			if (gru)
			{
				gates[r] = ai[i];
				gates[3 * H + r] = ah[i];
			}
			else
			{
				const float acc = (ai[i] + ah[i]) + bias[r];
				const bool isG = (r >= 2 * H) && (r < 3 * H); // rows [2H, 3H) are the cell candidate (tanh), the others sigmoid (LSTM.h:33-36,94-99)
				gates[r] = isG ? LstmTanh(acc, m.math) : LstmSigmoid(acc, m.math);
			}
		}
	}

	__device__ __forceinline__ void GateRowsL2Dispatch(const LstmModelDev& m, int l, int lane, int nt, const float* sIn, const float* sH, float* gates)
	{
		const int rpl = (((m.cell == LSTM_CELL_GRU) ? 3 : 4) * m.hidden + nt - 1) / nt;
		switch (rpl)
		{
		case 1: GateRowsL2<1>(m, l, lane, nt, sIn, sH, gates); break;
		case 2: GateRowsL2<2>(m, l, lane, nt, sIn, sH, gates); break;
		case 3: GateRowsL2<3>(m, l, lane, nt, sIn, sH, gates); break;
		case 4: GateRowsL2<4>(m, l, lane, nt, sIn, sH, gates); break;
		case 5: GateRowsL2<5>(m, l, lane, nt, sIn, sH, gates); break;
		case 6: GateRowsL2<6>(m, l, lane, nt, sIn, sH, gates); break;
		case 7: GateRowsL2<7>(m, l, lane, nt, sIn, sH, gates); break;
		default: GateRowsL2<8>(m, l, lane, nt, sIn, sH, gates); break;
		}
	}

	// l2w: the gate matrices are streamed from L2 (m.wT) instead of living in LDS -- for weights larger than the LDS (LSTM 2x64: 197 KB)
	// MAXT = 64: one wave per stream (wave fences); MAXT = 1024: a workgroup of m.waves waves per stream shares the gate rows (barriers)
	template <int MAXT>
	__device__ __forceinline__ void RecurrentRtSync()
	{
		if (MAXT == 64) LstmWaveSync(); // (one wave: LDS fence + wave barrier)
		else __syncthreads();
	}
	template <int MAXT>
	__global__ void __launch_bounds__(MAXT) RecurrentWaveRtKernel(LstmModelDev m, float* __restrict__ state, int capacity, const int* __restrict__ slots,
		const int* __restrict__ rows, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n, int l2w)
	{
		extern __shared__ __attribute__((aligned(16))) float lds[];
		const int H = m.hidden, L = m.numLayers;
		const bool gru = m.cell == LSTM_CELL_GRU;
		const int G = gru ? 3 : 4;          // gate row blocks per layer
		const int NB = gru ? 6 * H : 4 * H; // bias floats per layer
		const int Hs = L > 0 ? H : 1;
		float* xin = lds;
		float* hvec = xin + LSTM_MAX_FRAMES;   // [L][H]
		float* cvec = hvec + L * H;            // [L][H] (LSTM)
		float* gates = cvec + L * H;           // LSTM: [4H] activated gates; GRU: ai[3H] | ah[3H]
		const bool headInLoop = H >= RECURRENT_HEAD_IN_LOOP_FROM && m.tailLayers == 0; // (no [samples][H] buffer then)
		float* hseq = gates + 6 * H;           // [2][Hs][64]
		float* tailA = hseq + (headInLoop ? 0 : (size_t)2 * Hs * 64);
		const size_t tailOne = (size_t)RecurrentTailScratchFloats(m.tailLayers > 0 ? m.tailWidth : 0, m.tailLayers > 0 ? m.tailHistMax : 0) / 2;
		float* tailB = tailA + tailOne;
		float* wl = tailB + tailOne; // per layer: [G H][stride] then the biases

		const int lane = threadIdx.x;            // thread of the stream's workgroup
		const int nt = MAXT == 64 ? 64 : (int)blockDim.x;
		const int slot = slots[blockIdx.x];
		const int row = rows[blockIdx.x];
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;

		// weights -> LDS (global layout per layer: W row-major [G H][I + H], then the biases: LSTM bias[4H]; GRU b_in[3H], b_rec[3H])
		if (!l2w)
		{
			float* dst = wl;
			for (int l = 0; l < L; l++)
			{
				const int W = (l == 0 ? 1 : H) + H, stride = OddStride(W);
				const float* src = m.w + m.layerOff[l];
				for (int i = lane; i < G * H * W; i += nt) dst[(i / W) * stride + (i % W)] = src[i];
				for (int i = lane; i < NB; i += nt) dst[(size_t)G * H * stride + i] = src[(size_t)G * H * W + i];
				dst += (size_t)G * H * stride + NB;
			}
		}
		for (int f = lane; f < n; f += nt) xin[f] = inRow[f];
		for (int i = lane; i < L * H; i += nt)
		{
			const int l = i / H, k = i % H;
			hvec[i] = state[(size_t)(l * 2 * H + k) * capacity + slot];
			if (!gru) cvec[i] = state[(size_t)(l * 2 * H + H + k) * capacity + slot];
		}
		RecurrentRtSync<MAXT>();

		for (int f = 0; f < n; f++)
		{
			const float* wlay = wl;
			for (int l = 0; l < L; l++)
			{
				const int I = (l == 0) ? 1 : H, W = I + H, stride = OddStride(W);
				const float* sIn = (l == 0) ? (xin + f) : (hvec + (l - 1) * H); // LSTM.h:168 / :170-180
				float* sH = hvec + l * H;
				const float* bias = wlay + (size_t)G * H * stride; // (LDS mode)
				if (l2w) GateRowsL2Dispatch(m, l, lane, nt, sIn, sH, gates);
				if (!gru)
				{
					if (!l2w)
					for (int r = lane; r < 4 * H; r += nt)
					{
						const float* wr = wlay + (size_t)r * stride;
						float acc = 0.0f;
#pragma unroll 8
						for (int k = 0; k < I; k++) acc += wr[k] * sIn[k];
#pragma unroll 8
						for (int k = 0; k < H; k++) acc += wr[I + k] * sH[k]; // (unrolled: eight LDS reads in flight instead of one round trip per term)
						acc += bias[r];
						// rows [2H, 3H) are the cell candidate (tanh), the others sigmoid (LSTM.h:33-36,94-99)
						const bool isG = (r >= 2 * H) && (r < 3 * H);
						gates[r] = isG ? LstmTanh(acc, m.math) : LstmSigmoid(acc, m.math);
					}
					RecurrentRtSync<MAXT>();
					for (int u = lane; u < H; u += nt)
					{
						// LSTM.h:94-99
						const float c = (gates[H + u] * cvec[l * H + u]) + (gates[u] * gates[2 * H + u]);
						cvec[l * H + u] = c;
						sH[u] = gates[3 * H + u] * LstmTanh(c, m.math);
					}
				}
				else
				{
					// keras GRU, reset_after (gru_kernels.hip GruLayerStep): input and recurrent pre-activations kept apart
					if (!l2w)
					for (int r = lane; r < 3 * H; r += nt)
					{
						const float* wr = wlay + (size_t)r * stride;
						float ai = bias[r], ah = bias[3 * H + r];
#pragma unroll 8
						for (int k = 0; k < I; k++) ai += wr[k] * sIn[k];
#pragma unroll 8
						for (int k = 0; k < H; k++) ah += wr[I + k] * sH[k];
						gates[r] = ai;
						gates[3 * H + r] = ah;
					}
					RecurrentRtSync<MAXT>();
					for (int u = lane; u < H; u += nt)
					{
						const float z = GruSigmoid(gates[u] + gates[3 * H + u]);
						const float rr = GruSigmoid(gates[H + u] + gates[4 * H + u]);
						const float c = GruTanh(gates[2 * H + u] + rr * gates[5 * H + u]);
						sH[u] = (1.0f - z) * c + z * sH[u];
					}
				}
				RecurrentRtSync<MAXT>();
				wlay += (size_t)G * H * stride + NB;
			}
			if (headInLoop)
			{
				// LSTM.h:182-189 for this sample, by the first wave: the other waves are already in the next sample's gate rows (they read
				// h, which changes only behind the next barrier -- and the first wave reaches that barrier after this)
				if (lane < 64)
				{
					const float* headW = m.w + m.headOff;
					float acc = 0.0f;
					for (int k = lane; k < H; k += 64) acc += headW[k] * hvec[(L - 1) * H + k];
					for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
					if (lane == 0) xin[f] = acc + headW[H]; // (xin[f] was consumed by layer 0 of this sample)
				}
			}
			else if (L > 0)
				for (int u = lane; u < H; u += nt) hseq[(size_t)((f >> 6) * H + u) * 64 + (f & 63)] = hvec[(L - 1) * H + u];
		}
		RecurrentRtSync<MAXT>();

		// head / dense chain for the whole block, lane = sample (LSTM.h:182-189; RTNeuralModel.h:417-421)
		const float* headW = m.w + m.headOff;
		if (headInLoop)
			for (int f = lane; f < n; f += nt) outRow[f] = xin[f];
		else if (m.tailLayers > 0 && m.tailHistMax > 0)
		{
			// a tail with conv1d layers: layer by layer over the whole block (recurrent_tail.h)
			if (lane < 64) ConvTail(m, L > 0 ? hseq : nullptr, L > 0 ? H : 0, xin, tailA, tailB, state, capacity, slot, n, lane, outRow);
		}
		else if (lane < 64)
		for (int pass = 0; pass * 64 < n; pass++)
		{
			const int f = pass * 64 + lane;
			const float* hs = hseq + (size_t)pass * Hs * 64;
			float y;
			if (m.tailLayers > 0) y = DenseTail(m, L > 0 ? hs : nullptr, L > 0 ? H : 0, xin[f < n ? f : 0], tailA, tailB, lane);
			else
			{
				float acc = 0.0f;
				for (int k = 0; k < H; k++) acc += headW[k] * hs[k * 64 + lane];
				y = acc + headW[H];
			}
			if (f < n) outRow[f] = y;
		}
		for (int i = lane; i < L * H; i += nt)
		{
			const int l = i / H, k = i % H;
			state[(size_t)(l * 2 * H + k) * capacity + slot] = hvec[i];
			if (!gru) state[(size_t)(l * 2 * H + H + k) * capacity + slot] = cvec[i];
		}
	}

	// false: the shape does not fit (hidden > 64 or the weights exceed the LDS): the lane = stream kernels take it
	bool LaunchRecurrentWaveRt(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream, hipError_t& err)
	{
		// tuning knob / tests: the lane = stream kernels for every shape -- except tails with conv1d layers, which only this kernel evaluates
		const bool off = Tuning::Get().lstmNoWaveRt && !(m.tailLayers > 0 && m.tailHistMax > 0);
		if (off || m.hidden > RECURRENT_WAVE_MAX_HIDDEN || m.numLayers < 0 || (m.numLayers == 0 && m.tailLayers == 0)) return false;
		size_t ldsBytes = RecurrentWaveRtLdsFloats(m) * sizeof(float);
		// weights larger than the LDS (LSTM 2x64: 197 KB): streamed from L2, transposed for coalesced reads (NA_REC_L2W=1 forces the mode)
		const bool forceL2 = Tuning::Get().recL2w;
		const int l2w = (ldsBytes > 160 * 1024 || (forceL2 && m.numLayers > 0)) ? 1 : 0;
		if (l2w)
		{
			if (m.wT == nullptr) return false;
			ldsBytes = RecurrentWaveRtLdsFloats(m, false) * sizeof(float);
			if (ldsBytes > 160 * 1024) return false;
		}
		static PerDeviceOnce attr, attrBlock; // (hipFuncSetAttribute applies to the current device's copy of the kernel)
		const int waves = m.waves > 1 ? m.waves : 1;
		if (waves == 1)
		{
			(void)attr.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&RecurrentWaveRtKernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
			hipLaunchKernelGGL(RecurrentWaveRtKernel<64>, dim3((unsigned)numStreams), dim3(64), ldsBytes, stream, m, state, capacity, slots, rows, in, out, inStride,
				outStride, n, l2w);
		}
		else
		{
			(void)attrBlock.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&RecurrentWaveRtKernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
			hipLaunchKernelGGL(RecurrentWaveRtKernel<1024>, dim3((unsigned)numStreams), dim3(64u * (unsigned)waves), ldsBytes, stream, m, state, capacity, slots, rows, in,
				out, inStride, outStride, n, l2w);
		}
		err = hipGetLastError();
		return true;
	}

	// initial hidden / cell state of the listed slots (NAM: stored in the weights, LSTM.h:51-55; keras: zeros)
	__global__ void LstmInitStateKernel(float* __restrict__ state, int capacity, const int* __restrict__ slots, int numStreams,
		const float* __restrict__ init /* [numLayers*2H] */, int numElems)
	{
		const int idx = blockIdx.x * blockDim.x + threadIdx.x;
		if (idx >= numStreams) return;
		const int slot = slots[idx];
		for (int k = 0; k < numElems; k++) state[(size_t)k * capacity + slot] = init[k];
	}

	template <int H>
	static hipError_t LaunchH(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		const size_t ldsBytes = ((size_t)64 * (n + 1) + (size_t)m.numLayers * 2 * H * 64) * sizeof(float);
		if (ldsBytes > 160 * 1024) return hipErrorInvalidValue;
		static PerDeviceOnce attr; // per instantiation and device
		(void)attr.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&LstmBlockKernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
		hipLaunchKernelGGL(LstmBlockKernel<H>, dim3((unsigned)((numStreams + 63) / 64)), dim3(64), ldsBytes, stream, m, state, capacity,
			slots, rows, numStreams, in, out, inStride, outStride, n);
		return hipGetLastError();
	}


	// ------------------------------------------------------------------------------------------------------------
	// Any hidden size / layer count (the reference's runtime-shaped path, LSTMDynamic.h:95-108,175-215): lane = stream, runtime loops,
	// weights through wave-uniform loads.  LDS: io[64][n + 1] | hc[numLayers][2H][64] | hnew[H][64].  Slow next to the shaped kernels
	// (no unrolling, every weight is a scalar load per use) but it accepts what the reference accepts.
	// ------------------------------------------------------------------------------------------------------------
	__global__ void __launch_bounds__(64) LstmGenericKernel(LstmModelDev m, float* __restrict__ state, int capacity, const int* __restrict__ slots,
		const int* __restrict__ rows, int numStreams, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n)
	{
		extern __shared__ __attribute__((aligned(16))) float lds[];
		const int H = m.hidden;
		float* io = lds;
		const int ioStride = n + 1;
		float* hcAll = lds + 64 * ioStride;
		float* hnew = hcAll + (size_t)m.numLayers * 2 * H * 64;
		float* tailA = hnew + (size_t)H * 64; // generic keras stack only: two [tailWidth][64] arrays
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
		for (int k = 0; k < m.numLayers * 2 * H; k++) hcAll[k * 64 + lane] = active ? state[(size_t)k * capacity + slot] : 0.0f;
		__syncthreads();

		const float* headW = m.w + m.headOff;
		for (int f = 0; f < n; f++)
		{
			const float x0 = io[lane * ioStride + f];
			for (int l = 0; l < m.numLayers; l++)
			{
				const int I = (l == 0) ? 1 : H;
				const int W = I + H;
				const float* w = m.w + m.layerOff[l];
				const float* bias = w + (size_t)4 * H * W;
				float* hc = hcAll + (size_t)l * 2 * H * 64;
				const float* below = hcAll + (size_t)(l > 0 ? l - 1 : 0) * 2 * H * 64; // h of the layer below (already updated for this sample)
				for (int i = 0; i < H; i++)
				{
					float g[4];
					for (int q = 0; q < 4; q++)
					{
						const float* r = w + (size_t)(q * H + i) * W;
						float acc = 0.0f;
						if (l == 0) acc += r[0] * x0; // LSTM.h:168
						else
							for (int k = 0; k < H; k++) acc += r[k] * below[k * 64 + lane]; // LSTM.h:170-180
						for (int k = 0; k < H; k++) acc += r[I + k] * hc[k * 64 + lane];
						g[q] = acc + bias[q * H + i];
					}
					// LSTM.h:94-99 (gate row blocks i, f, g, o)
					const float c = (LstmSigmoid(g[1], m.math) * hc[(H + i) * 64 + lane]) + (LstmSigmoid(g[0], m.math) * LstmTanh(g[2], m.math));
					hc[(H + i) * 64 + lane] = c;
					hnew[i * 64 + lane] = LstmSigmoid(g[3], m.math) * LstmTanh(c, m.math);
				}
				for (int i = 0; i < H; i++) hc[i * 64 + lane] = hnew[i * 64 + lane];
			}
			const float* hl = hcAll + (size_t)(m.numLayers > 0 ? m.numLayers - 1 : 0) * 2 * H * 64;
			if (m.tailLayers > 0)
			{
				io[lane * ioStride + f] = DenseTail(m, m.numLayers > 0 ? hl : nullptr, m.numLayers > 0 ? H : 0, x0, tailA, tailB, lane);
				continue;
			}
			float acc = 0.0f;
			for (int k = 0; k < H; k++) acc += headW[k] * hl[k * 64 + lane];
			io[lane * ioStride + f] = acc + headW[H]; // LSTM.h:182-189
		}
		__syncthreads();

		for (int k = 0; k < m.numLayers * 2 * H; k++)
			if (active) state[(size_t)k * capacity + slot] = hcAll[k * 64 + lane];
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

	static size_t LstmGenericLdsBytes(int hidden, int numLayers, int n, int tailWidth)
	{
		return ((size_t)64 * (n + 1) + (size_t)numLayers * 2 * hidden * 64 + (size_t)hidden * 64 + (size_t)2 * tailWidth * 64) * sizeof(float);
	}

	static hipError_t LaunchGeneric(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams, const float* in,
		float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		const size_t ldsBytes = LstmGenericLdsBytes(m.hidden, m.numLayers, n, m.tailLayers > 0 ? m.tailWidth : 0);
		if (ldsBytes > 160 * 1024) return hipErrorInvalidValue;
		static PerDeviceOnce attr;
		(void)attr.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&LstmGenericKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
		hipLaunchKernelGGL(LstmGenericKernel, dim3((unsigned)((numStreams + 63) / 64)), dim3(64), ldsBytes, stream, m, state, capacity, slots, rows,
			numStreams, in, out, inStride, outStride, n);
		return hipGetLastError();
	}

	hipError_t LaunchLstmBlock(const LstmModelDev& m, float* state, int capacity, const int* slots, const int* rows, int numStreams,
		const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		if (n > LSTM_MAX_FRAMES) return hipErrorInvalidValue;
		{
			const bool forceLaneKernel = Tuning::Get().lstmLaneKernel && !(m.tailLayers > 0 && m.tailHistMax > 0); // tuning knob (conv1d tails: the wave kernel only)
			hipError_t err = hipSuccess;
			if (!forceLaneKernel && m.tailLayers == 0 && LaunchLstmWave(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream, err)) return err;
			if (!forceLaneKernel && LaunchRecurrentWaveRt(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream, err)) return err;
		}
		if (m.tailLayers > 0 && m.tailHistMax > 0) return hipErrorNotSupported; // (conv1d tails: the runtime-shaped wave kernel only)
		if (m.tailLayers > 0) return LaunchGeneric(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream); // generic keras stack
#define NA_LSTM_CASE(HH) case HH: return LaunchH<HH>(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream)
		switch (m.hidden)
		{
			NA_LSTM_CASE(4);
			NA_LSTM_CASE(8);
			NA_LSTM_CASE(12);
			NA_LSTM_CASE(16);
			NA_LSTM_CASE(20);
			NA_LSTM_CASE(24);
			NA_LSTM_CASE(32);
			NA_LSTM_CASE(40);
		default: return LaunchGeneric(m, state, capacity, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		}
#undef NA_LSTM_CASE
	}

	hipError_t LaunchLstmInitState(float* state, int capacity, const int* slots, int numStreams, const float* init, int numElems,
		hipStream_t stream)
	{
		if (numStreams <= 0) return hipSuccess;
		hipLaunchKernelGGL(LstmInitStateKernel, dim3((unsigned)((numStreams + 255) / 256)), dim3(256), 0, stream, state, capacity, slots,
			numStreams, init, numElems);
		return hipGetLastError();
	}
}
