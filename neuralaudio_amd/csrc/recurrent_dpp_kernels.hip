// recurrent_dpp_kernels.hip -- the LDS-free recurrent kernels (LSTM and keras GRU, hidden size up to 16 padded into an 8- or 16-unit
// lane layout, 1-2 layers; one-layer LSTMs up to 32 units on a 32-unit layout: every shape the reference builds statically,
// NeuralModel.cpp:32-38) as ONE
// gfx950 kernel that serves up to RECURRENT_MAX_GROUPS model groups per launch: a batch holding several recurrent models
// (BASELINE config 4: LSTM 2x16 + GRU) runs without stream fork/join and its waves share the chip.
//
// Reference arithmetic: LSTMModelT/LSTMLayerT::Process (NeuralAudio/LSTM.h:164-191, 87-100), FastMath (Activation.h:83-96);
// keras GRU = RTNeural's GRULayer (NeuralAudio/RTNeuralModel.h:300,417-421; third-party, parity unpinned -- see gru_kernels.hip).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

#include <hip/hip_runtime.h>

#include "dpp_recurrent.h"
#include "device_once.h"
#include "tuning.h"
#include "lstm_dev.h"
#include "lstm_launch.h"
#include "wavenet_launch.h"

namespace na
{
	__device__ __forceinline__ void RecurrentWaveSync()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
	}

	// p[idx] where `on`, else 0 -- as an unconditional load from a clamped index plus a select: a `cond ? p[i] : 0` per weight turns the
	// prologue into a chain of exec-masked loads (measured: +1.5 us per launch)
	__device__ __forceinline__ float LoadIf(const float* __restrict__ p, size_t idx, bool on)
	{
		const float v = p[on ? idx : (size_t)0];
		return on ? v : 0.0f;
	}

	// ------------------------------------------------------------------------------------------------------------
	// Fastest path (H = 8 or 16): one wave per stream, NO LDS on the recurrence.
	//   lane = H*gate + unit (gate order i,f,g,o; for H = 8 the upper 32 lanes mirror the lower 32), every lane keeps h[unit] and c[unit]
	//   (replicated across the gate rows).  The mat-vec reads h[(unit - n) mod H] from a neighbour lane with DPP row_ror:n (a 16-lane row
	//   holds the H units once or twice), against weights that were rotated the same way when they were loaded -- 1 instruction per
	//   term, no broadcast through LDS or SGPRs.  The four gates of a unit meet through gfx950 lane swaps (ReplicateRows, LstmCellState8).
	//   Each lane sums its row starting at column `unit` and walking down instead of 0..H-1: same products, different rounding order than
	//   LSTM.h:87-100 (observed difference vs the oracle ~1e-7 RMS, tolerance 5e-6).  tanh divides with v_rcp_f32 like the WaveNet path.
	// ------------------------------------------------------------------------------------------------------------
	__device__ __forceinline__ float LstmRcpTanh(float x)
	{
		const float ax = fabsf(x);
		const float x2 = x * x;
		const float num = x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2);
		const float den = 2.44506634652299f + (2.44506634652299f + x2) * (ax + 0.814642734961073f * x2); // |x + e x|x|| == |x| + e x^2
		return num * __builtin_amdgcn_rcpf(den);
	}

	// Gate activation with the gate's identity folded into per-lane constants (no select, no branch: a lone wave pays ~5 cycles per
	// instruction issued, whatever it is).  FastMath (LSTM.h:33-36,94-99; Activation.h:83-96): the g row takes Tanh(x), the i / f / o
	// rows Sigmoid(x) = 0.5 (Tanh(0.5 x) + 1).  The inner 0.5 rides in the weights of those rows (scaling by a power of two is
	// exact: every partial sum is exactly half the reference's), the outer 0.5 in the numerator constants (same argument), so
	// value = A tanh(y) + B is one polynomial and one fma.  StdMath (Activation.h:37-45): rcp(1 + exp2(k y)) with k = -log2 e
	// (sigmoid) or 2 log2 e and the result mapped by 1 - 2 t (tanh).
	template <bool STD>
	struct GateK;
	template <>
	struct GateK<false>
	{
		float aA, bA, cA, B;
	};
	template <>
	struct GateK<true>
	{
		float k, A, B;
	};

	template <bool STD>
	__device__ __forceinline__ GateK<STD> MakeGateK(bool isG)
	{
		GateK<STD> K;
		if constexpr (STD)
		{
			K.k = isG ? 2.885390081777927f : -1.4426950408889634f;
			K.A = isG ? -2.0f : 1.0f;
			K.B = isG ? 1.0f : 0.0f;
		}
		else
		{
			const float A = isG ? 1.0f : 0.5f;
			K.aA = 2.45550750702956f * A;
			K.bA = 0.893229853513558f * A;
			K.cA = 0.821226666969744f * A;
			K.B = isG ? 0.0f : 0.5f;
		}
		return K;
	}

	// the factor the weights and the bias of this lane's gate row are loaded with
	template <bool STD>
	__device__ __forceinline__ float GateRowScale(bool isG) { return (STD || isG) ? 1.0f : 0.5f; }

	template <bool STD>
	__device__ __forceinline__ float GateAct(float y, const GateK<STD>& K)
	{
		if constexpr (STD) return __builtin_fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y * K.k) + 1.0f), K.A, K.B);
		else
		{
			const float ay = fabsf(y);
			const float y2 = y * y;
			const float p = __builtin_fmaf(y2, __builtin_fmaf(ay, K.cA, K.bA), __builtin_fmaf(ay, K.aA, K.aA));
			const float den = 2.44506634652299f + (2.44506634652299f + y2) * (ay + 0.814642734961073f * y2);
			return __builtin_fmaf(y * p, __builtin_amdgcn_rcpf(den), K.B);
		}
	}

	// gate pre-activation (row-scaled, see GateRowScale) -> (c, h) update for this lane's unit; returns the new h.
	// The four gates of a unit meet through gfx950 lane swaps (v_permlane32_swap: a.hi <-> b.lo; v_permlane16_swap: odd rows of a <->
	// even rows of b; probed on the box), no LDS.  H = 16: lane = 16 gate + unit, every row gets every gate row (ReplicateRows).
	// H = 8: a row holds two gate blocks ([i | f] and [g | o]); the lanes of the first block compute the unit (i and g are their own,
	// f and o sit in the other half of the row: DPP row_ror:8 inside the consuming instruction) and their h is copied over the
	// second block's -- whose c stays undefined and is never read.
	template <int H, bool STD>
	__device__ __forceinline__ float DppCellUpdate(float acc, const GateK<STD>& K, float& c)
	{
		const float gv = GateAct<STD>(acc, K);
		if constexpr (H == 16)
		{
			float gi, gf, gg, go;
			ReplicateRows(gv, gi, gf, gg, go);
			c = __builtin_fmaf(gf, c, gi * gg);
			return go * (STD ? StdTanh(c) : LstmRcpTanh(c));
		}
		else
		{
			float go;
			LstmCellState8(gv, c, go);
			return LstmCellOut8(go, STD ? StdTanh(c) : LstmRcpTanh(c));
		}
	}

	// xin: this wave's input samples in LDS, read back four at a time (xs must be 16-byte aligned where the groups start, with at
	// least 4 readable floats past the block); hout: the h of every sample, written by EVERY lane (lanes of the same unit hold
	// the same value and write the same word: no exec mask, no branch on the recurrence)
	constexpr int REC_XIN_FLOATS = LSTM_MAX_FRAMES + 16;
	constexpr int REC_HOUT_FLOATS = (LSTM_MAX_FRAMES + 2) * 18;
	constexpr int REC_HOUT32_FLOATS = (LSTM_MAX_FRAMES + 2) * 33; // the 32-unit layout (LstmDpp32Body)

	// one stream (slot, row), one block of n samples
	template <int H, int L, bool STD>
	__device__ __forceinline__ void LstmDppBodyM(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, float* xin, float* hout)
	{
		static_assert(H == 8 || H == 16, "a 16-lane DPP row must hold the units a whole number of times");
		constexpr int HP = H + 1;
		const int lane = (int)threadIdx.x & 63; // (the two-wave kernel runs these bodies on the waves of a 256-thread workgroup)
		const int unit = lane % H;
		const int gate = (lane / H) & 3;
		// H is the lane layout (8 or 16 units per gate block); the model's hidden size hr may be smaller (12 on the 16 layout, 4 .. 7 on
		// the 8 layout): the surplus units carry zero weights, zero bias and zero state, so their gates stay sigmoid(0) / tanh(0), their
		// c and h stay exactly 0 and they add nothing to anyone's row sum.
		const int hr = m.hidden;
		const bool real = unit < hr;
		const int r = gate * hr + unit; // this lane's gate row in the model's tensors
		auto col = [&](int k) { return (unit - k + H) % H; }; // row_ror:k hands lane p the value of lane p - k
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;
		const GateK<STD> K = MakeGateK<STD>(gate == 2);
		const float gs = GateRowScale<STD>(gate == 2);

		// layer 0: W row-major [4H][1 + H], then bias[4H] (LSTM.h:42-56); h weights rotated by `unit`
		const float* w0 = m.w + m.layerOff[0];
		const float wx0 = gs * LoadIf(w0, (size_t)r * (1 + hr), real);
		float wh0[H];
#pragma unroll
		for (int k = 0; k < H; k++) wh0[k] = gs * LoadIf(w0, (size_t)r * (1 + hr) + 1 + col(k), real && col(k) < hr);
		const float b0 = gs * LoadIf(w0, (size_t)4 * hr * (1 + hr) + r, real);
		// layer 1: W [4H][H + H]: input = layer-0 h, then own h
		float wi1[H], wh1[H];
		float b1 = 0.0f;
		if (L > 1)
		{
			const float* w1 = m.w + m.layerOff[L > 1 ? 1 : 0];
#pragma unroll
			for (int k = 0; k < H; k++)
			{
				const bool on = real && col(k) < hr;
				wi1[k] = gs * LoadIf(w1, (size_t)r * (2 * hr) + col(k), on);
				wh1[k] = gs * LoadIf(w1, (size_t)r * (2 * hr) + hr + col(k), on);
			}
			b1 = gs * LoadIf(w1, (size_t)4 * hr * (2 * hr) + r, real);
		}

		for (int f = lane; f < n + 4; f += 64) xin[f] = f < n ? inRow[f] : 0.0f;
		float h[L], c[L];
#pragma unroll
		for (int l = 0; l < L; l++)
		{
			h[l] = LoadIf(state, (size_t)(l * 2 * hr + unit) * capacity + slot, real);
			c[l] = LoadIf(state, (size_t)(l * 2 * hr + hr + unit) * capacity + slot, real);
		}
		RecurrentWaveSync();

		// The h of sample f - 1 is stored right after the first dot of sample f: an instruction between the asm block and the first
		// use of its result saves the wait state the compiler otherwise puts there.  (Entry 0 of hout = the h the block started with.)
		float* hw = hout + unit;
		auto step = [&](float x, float* dst) {
			float acc;
			DppDotFrom<H>(acc, wx0, x, b0, wh0, h[0]); // LSTM.h:168 -- column 0 is the input sample
			*dst = h[L - 1];
			h[0] = DppCellUpdate<H, STD>(acc, K, c[0]);
			if constexpr (L > 1)
			{
				float acc1;
				DppDotFrom2<H>(acc1, b1, wi1, h[0], wh1[0], h[1]); // LSTM.h:170-180
				DppDotTail<H>(acc1, wh1, h[1]);
				h[1] = DppCellUpdate<H, STD>(acc1, K, c[1]);
			}
		};
		int f = 0;
		for (; f + 4 <= n; f += 4)
		{
			const float4 xv = *reinterpret_cast<const float4*>(xin + f);
			step(xv.x, hw + (f + 0) * HP);
			step(xv.y, hw + (f + 1) * HP);
			step(xv.z, hw + (f + 2) * HP);
			step(xv.w, hw + (f + 3) * HP);
		}
		for (; f < n; f++) step(xin[f], hw + f * HP);
		hw[n * HP] = h[L - 1];
		RecurrentWaveSync();

		// dense head for the whole block, lane = sample (LSTM.h:182-189)
		const float* headW = m.w + m.headOff;
		for (int f = lane; f < n; f += 64)
		{
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < H; k++) acc += LoadIf(headW, (size_t)k, k < hr) * hout[(f + 1) * HP + k];
			outRow[f] = acc + headW[hr];
		}
		if (lane < hr)
		{
#pragma unroll
			for (int l = 0; l < L; l++)
			{
				state[(size_t)(l * 2 * hr + lane) * capacity + slot] = h[l];
				state[(size_t)(l * 2 * hr + hr + lane) * capacity + slot] = c[l];
			}
		}
	}


	// H = 8, two layers: the two layers run side by side, one step apart.  With H = 8 the four gate rows fill 32 lanes, so instead of
	// mirroring them into the upper half (above), the lower 32 lanes hold layer 0 and the upper 32 layer 1: in tick t layer 0 processes
	// sample t while layer 1 processes sample t - 1, whose input h0(t - 1) it takes from the lower half with ONE lane swap.  A tick is
	// one cell update (the same instructions serve both halves) instead of two in sequence: the dependent chain per sample -- which is
	// all that bounds a 1024-stream batch, one wave per SIMD -- is cut from (3 dots + 2 cell updates) to (2 dots + 1), and the two
	// dots of the upper half run into one sum (own state first, then the input; LSTM.h:170-180 the other way round: ~1e-7 RMS apart).  n + 1 ticks per
	// block: the first and the last one (layer 1 has no sample -1, layer 0 no sample n) are peeled off with their masks, so the
	// saved state is the reference's at every block boundary.

	// both dots of a tick in one block: hin = [h.lo, h.lo] (the layer input of both halves) through one v_permlane32_swap whose wait
	// states are filled with the start of the sum; acc = wx x + b + wb . h(rotated) + wa . hin(rotated), one running sum like LSTM.h:170-180
	__device__ __forceinline__ float SkewTickDots(float h, float wx, float x, float b, const float (&wa)[8], const float (&wb)[8])
	{
		float acc, hin, tmp;
#define NA_SK_TERM(S, N, OP) "v_fmac_f32_dpp %0, %" #S ", %" #OP " row_ror:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
		asm volatile(
			"v_mov_b32 %1, %3\n"
			"v_mov_b32 %2, %3\n"
			"v_fma_f32 %0, %4, %5, %6\n"
			"v_fmac_f32 %0, %15, %3\n"
			"v_permlane32_swap_b32 %1, %2\n"
			NA_SK_TERM(3, 1, 16) NA_SK_TERM(3, 2, 17) NA_SK_TERM(3, 3, 18) NA_SK_TERM(3, 4, 19) NA_SK_TERM(3, 5, 20) NA_SK_TERM(3, 6, 21) NA_SK_TERM(3, 7, 22)
			"v_fmac_f32 %0, %7, %1\n"
			NA_SK_TERM(1, 1, 8) NA_SK_TERM(1, 2, 9) NA_SK_TERM(1, 3, 10) NA_SK_TERM(1, 4, 11) NA_SK_TERM(1, 5, 12) NA_SK_TERM(1, 6, 13) NA_SK_TERM(1, 7, 14)
			: "=&v"(acc), "=&v"(hin), "=&v"(tmp)
			: "v"(h), "v"(wx), "v"(x), "v"(b), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wa[4]), "v"(wa[5]), "v"(wa[6]), "v"(wa[7]), "v"(wb[0]), "v"(wb[1]),
			"v"(wb[2]), "v"(wb[3]), "v"(wb[4]), "v"(wb[5]), "v"(wb[6]), "v"(wb[7]));
#undef NA_SK_TERM
		return acc;
	}

	template <bool STD>
	__device__ __forceinline__ void LstmDppSkewBody(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, float* xin, float* hout)
	{
		constexpr int H = 8, HP = H + 1;
		constexpr int HREGION = (LSTM_MAX_FRAMES + 2) * HP; // per layer: the h before every tick and after the last
		static_assert(2 * HREGION <= REC_HOUT_FLOATS, "");
		const int lane = (int)threadIdx.x & 63; // (the two-wave kernel runs these bodies on the waves of a 256-thread workgroup)
		const int unit = lane % H;
		const int gate = (lane / H) & 3;
		const int layer = lane >> 5; // 0: lanes 0..31, 1: lanes 32..63
		const int hr = m.hidden; // <= 8: surplus units carry zero weights and zero state (see LstmDppBodyM)
		const bool real = unit < hr;
		const int r = gate * hr + unit;
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;
		const GateK<STD> K = MakeGateK<STD>(gate == 2);
		const float gs = GateRowScale<STD>(gate == 2);

		// layer 0: W row-major [4H][1 + H], bias[4H]; layer 1: W [4H][H + H] (input = layer-0 h, then own h), bias[4H] (LSTM.h:42-56).
		// wa multiplies the layer input h (own h for layer 0), wb the own h of layer 1; rotated by `unit` for the DPP walk.
		const float* w0 = m.w + m.layerOff[0];
		const float* w1 = m.w + m.layerOff[1];
		const float wx = gs * LoadIf(w0, (size_t)r * (1 + hr), layer == 0 && real);
		float wa[H], wb[H];
#pragma unroll
		for (int k = 0; k < H; k++)
		{
			const int col = (unit - k + H) % H; // row_ror:k hands lane p the value of lane p - k
			const bool on = real && col < hr;
			wa[k] = gs * LoadIf(layer == 0 ? w0 : w1, layer == 0 ? (size_t)r * (1 + hr) + 1 + col : (size_t)r * (2 * hr) + col, on);
			wb[k] = gs * LoadIf(w1, (size_t)r * (2 * hr) + hr + col, layer == 1 && on);
		}
		const float b = gs * LoadIf(layer == 0 ? w0 : w1, layer == 0 ? (size_t)4 * hr * (1 + hr) + r : (size_t)4 * hr * (2 * hr) + r, real);

		float* xs = xin + 3; // ticks 1, 5, 9, ... start the groups of four: &xs[1] is 16-byte aligned
		for (int f = lane; f < n + 8; f += 64) xs[f] = f < n ? inRow[f] : 0.0f;
		float h = LoadIf(state, (size_t)(layer * 2 * hr + unit) * capacity + slot, real);
		float c = LoadIf(state, (size_t)(layer * 2 * hr + hr + unit) * capacity + slot, real);
		RecurrentWaveSync();

		// after tick t the lower half holds h0(t), the upper half h1(t - 1); both are stored (region `layer`, entry t + 1): no exec mask
		float* hw = hout + layer * HREGION + unit;
		auto tick = [&](auto masked, float x, int t) {
			const float acc = SkewTickDots(h, wx, x, b, wa, wb);
			hw[t * HP] = h; // the state BEFORE tick t = after tick t - 1 (placed here: see LstmDppBodyM)
			float cn = c;
			const float hn = DppCellUpdate<H, STD>(acc, K, cn);
			if constexpr (decltype(masked)::value)
			{
				const bool active = layer == 0 ? (t < n) : (t > 0); // layer 0 has no sample n, layer 1 no sample -1
				h = active ? hn : h;
				c = active ? cn : c;
			}
			else
			{
				h = hn;
				c = cn;
			}
		};
		tick(std::true_type{}, xs[0], 0);
		int t = 1;
		for (; t + 4 <= n; t += 4)
		{
			const float4 xv = *reinterpret_cast<const float4*>(xs + t);
			tick(std::false_type{}, xv.x, t + 0);
			tick(std::false_type{}, xv.y, t + 1);
			tick(std::false_type{}, xv.z, t + 2);
			tick(std::false_type{}, xv.w, t + 3);
		}
		for (; t < n; t++) tick(std::false_type{}, xs[t], t);
		tick(std::true_type{}, 0.0f, n);
		hw[(n + 1) * HP] = h;
		RecurrentWaveSync();

		// dense head for the whole block, lane = sample (LSTM.h:182-189): h1 of sample f = the state after tick f + 1 = entry f + 2
		const float* headW = m.w + m.headOff;
		const float* h1 = hout + HREGION;
		for (int f = lane; f < n; f += 64)
		{
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < H; k++) acc += LoadIf(headW, (size_t)k, k < hr) * h1[(f + 2) * HP + k];
			outRow[f] = acc + headW[hr];
		}
		if (gate == 0 && real) // lanes 0..7: layer 0, lanes 32..39: layer 1
		{
			state[(size_t)(layer * 2 * hr + unit) * capacity + slot] = h;
			state[(size_t)(layer * 2 * hr + hr + unit) * capacity + slot] = c;
		}
	}

	// H = 16 layout, two layers (LSTM 2x16 / 2x12: BASELINE config 4): TWO WAVES PER STREAM, one per layer, a few samples apart.
	// A lone wave issues an instruction every ~5 cycles while its SIMD could take one every ~2 (tools/microbench: lone_wave_issue vs
	// valu_rate_saturated), and a batch of <= 1024 streams is one wave per SIMD: half the VALU issue slots of the chip are idle and the
	// step time is the dependent chain of ONE wave -- 3 dots + 2 cell updates per sample (LstmDppBodyM<16, 2>: 112 instructions, 37 us
	// per 128-sample block).  The two layers of a stream depend on each other in ONE direction only (layer 1 of sample t needs h0(t),
	// LSTM.h:170-180), so they run as a pipeline: wave 0 owns layer 0 and runs ahead freely, writing h0 of every sample into LDS and a
	// progress word every four samples; wave 1 owns layer 1, follows a group behind and takes its input from there (four ds_reads per
	// group, issued together).  Nobody waits for a round trip: the chain per sample is max(1 dot + 1 update, 2 dots + 1 update) = 63
	// instructions instead of 112, on a chip that now carries two waves per SIMD.  Same lane layout, same weights, same order of
	// floating-point operations as LstmDppBodyM<16, 2>: bit-identical results (tests/test_gpu_batch.py, NA_REC_NOPIPE=1 selects the
	// one-wave body).  LDS: h0[n + 2][17] | h1[n + 2][17] | progress.
#ifndef NA_PIPE_PUBLISH_LATE
#define NA_PIPE_PUBLISH_LATE 0 // (A/B: the layer-0 wave publishes at the end of a group instead of after its first step)
#endif
	constexpr int REC_PIPE_HP = 17;
	constexpr int REC_PIPE_REGION = (LSTM_MAX_FRAMES + 2) * REC_PIPE_HP;
	constexpr int REC_PIPE_FLOATS = 2 * REC_PIPE_REGION + 16;

	__device__ __forceinline__ bool RecurrentPipeGroup(const LstmModelDev& m)
	{
		return m.cell == LSTM_CELL_LSTM && m.numLayers == 2 && m.hidden > 8 && m.hidden <= 16 && m.tailLayers == 0;
	}

	template <bool STD>
	__device__ __forceinline__ void LstmDppPipeBody(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, int layer, float* xin, float* lds)
	{
		// layer (wave-uniform): which of the stream's two waves this is; xin: the layer-0 wave's input samples; lds: the pair's region
		constexpr int H = 16, HP = REC_PIPE_HP;
		const int lane = (int)threadIdx.x & 63;
		const int unit = lane % H;
		const int gate = lane / H;
		const int hr = m.hidden; // 9 .. 16: surplus units carry zero weights and zero state (see LstmDppBodyM)
		const bool real = unit < hr;
		const int r = gate * hr + unit;
		auto col = [&](int k) { return (unit - k + H) % H; };
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;
		const GateK<STD> K = MakeGateK<STD>(gate == 2);
		const float gs = GateRowScale<STD>(gate == 2);
		float* hb0 = lds;
		float* hb1 = lds + REC_PIPE_REGION;
		int* prog = reinterpret_cast<int*>(lds + 2 * REC_PIPE_REGION);

		// layer 0: W row-major [4H][1 + H], bias[4H]; layer 1: W [4H][H + H] (input = layer-0 h, then own h), bias[4H] (LSTM.h:42-56).
		// wa: the weights of the layer input's h part (layer 0: its own h), wb: layer 1's own h; rotated by `unit` for the DPP walk
		const float* w0 = m.w + m.layerOff[0];
		const float* w1 = m.w + m.layerOff[1];
		const float wx = gs * LoadIf(w0, (size_t)r * (1 + hr), layer == 0 && real);
		float wa[H], wb[H];
#pragma unroll
		for (int k = 0; k < H; k++)
		{
			const bool on = real && col(k) < hr;
			wa[k] = gs * LoadIf(layer == 0 ? w0 : w1, layer == 0 ? (size_t)r * (1 + hr) + 1 + col(k) : (size_t)r * (2 * hr) + col(k), on);
			wb[k] = gs * LoadIf(w1, (size_t)r * (2 * hr) + hr + col(k), layer == 1 && on);
		}
		const float b = gs * LoadIf(layer == 0 ? w0 : w1, layer == 0 ? (size_t)4 * hr * (1 + hr) + r : (size_t)4 * hr * (2 * hr) + r, real);
		if (layer == 0)
			for (int f = lane; f < n + 4; f += 64) xin[f] = f < n ? inRow[f] : 0.0f;
		float h = LoadIf(state, (size_t)(layer * 2 * hr + unit) * capacity + slot, real);
		float c = LoadIf(state, (size_t)(layer * 2 * hr + hr + unit) * capacity + slot, real);
		if (layer == 0) __hip_atomic_store(prog, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__syncthreads(); // (all live waves of the workgroup: pairs whose stream does not exist have left and do not count)

		if (layer == 0)
		{
			// entry t of hb0 = h0 BEFORE sample t (stored right after the first dot of sample t: see LstmDppBodyM), entry n = after the last
			float* hw = hb0 + unit;
			auto step = [&](float x, float* dst) {
				float acc;
				DppDotFrom<H>(acc, wx, x, b, wa, h); // LSTM.h:168 -- column 0 is the input sample
				*dst = h;
				h = DppCellUpdate<H, STD>(acc, K, c);
			};
			int f = 0;
			for (; f + 4 <= n; f += 4)
			{
				const float4 xv = *reinterpret_cast<const float4*>(xin + f);
				step(xv.x, hw + (f + 0) * HP);
				// entries 0 .. f are complete (LDS operations of a wave retire in order; the release orders the compiler): the group
				// f - 4 .. f - 1 of the layer-1 wave, which ends with entry f, may start
#if !NA_PIPE_PUBLISH_LATE
				__hip_atomic_store(prog, f + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
				step(xv.y, hw + (f + 1) * HP);
				step(xv.z, hw + (f + 2) * HP);
				step(xv.w, hw + (f + 3) * HP);
#if NA_PIPE_PUBLISH_LATE
				__hip_atomic_store(prog, f + 4, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
			}
			for (; f < n; f++) step(xin[f], hw + f * HP);
			hw[n * HP] = h;
			__hip_atomic_store(prog, n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (lane < hr)
			{
				state[(size_t)lane * capacity + slot] = h;
				state[(size_t)(hr + lane) * capacity + slot] = c;
			}
			return;
		}

		// layer 1: sample t takes h0 after sample t = entry t + 1 of hb0
		const float* hr0 = hb0 + unit;
		float* hw = hb1 + unit;
		auto step = [&](float hin, float* dst) {
			float acc;
			DppDotFrom2<H>(acc, b, wa, hin, wb[0], h); // LSTM.h:170-180
			DppDotTail<H>(acc, wb, h);
			*dst = h;
			h = DppCellUpdate<H, STD>(acc, K, c);
		};
		auto waitFor = [&](int entries) {
			while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < entries) __builtin_amdgcn_s_sleep(1);
		};
		// (The look at the progress word and the four reads of a group are two LDS round trips in front of its steps.  Requesting both in
		// front of the PREVIOUS group's steps -- a seen-progress counter, four more registers -- was built and measured: LSTM 2x16 x 256
		// 26.1 us either way, config 4 33.6 against 32.7 us.  The wave is bound by the instructions it issues, the other wave of the SIMD
		// fills the round trips, and the bookkeeping costs more than it hides: profiles/r06_cfg4_pipeline.txt.)
		int f = 0;
		for (; f + 4 <= n; f += 4)
		{
			waitFor(f + 5);
			const float i0 = hr0[(f + 1) * HP], i1 = hr0[(f + 2) * HP], i2 = hr0[(f + 3) * HP], i3 = hr0[(f + 4) * HP];
			step(i0, hw + (f + 0) * HP);
			step(i1, hw + (f + 1) * HP);
			step(i2, hw + (f + 2) * HP);
			step(i3, hw + (f + 3) * HP);
		}
		if (f < n) waitFor(n + 1);
		for (; f < n; f++) step(hr0[(f + 1) * HP], hw + f * HP);
		hw[n * HP] = h;
		RecurrentWaveSync();

		// dense head for the whole block, lane = sample (LSTM.h:182-189)
		const float* headW = m.w + m.headOff;
		for (int s = lane; s < n; s += 64)
		{
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < H; k++) acc += LoadIf(headW, (size_t)k, k < hr) * hb1[(s + 1) * HP + k];
			outRow[s] = acc + headW[hr];
		}
		if (lane < hr)
		{
			state[(size_t)(2 * hr + lane) * capacity + slot] = h;
			state[(size_t)(2 * hr + hr + lane) * capacity + slot] = c;
		}
	}

	// One layer, hidden 17 .. 32 (the reference's static 1x24, NeuralModel.cpp:35): the 32-unit layout.  A 16-lane row still walks 16
	// units with row_ror, so the state lives as TWO vectors replicated in every row (a = h[0..15], b = h[16..31]) and a gate row sum is two
	// DPP walks; every lane owns unit 16 r + j (r = row & 1) and TWO gates of it: rows 0-1 (pair A) the i and g gates, rows 2-3 (pair B)
	// f and o.  So slot 0 is a sigmoid everywhere, i g is local to pair A, f c local to pair B, their sum meets through one
	// v_permlane32_swap, h = o tanh(c') comes out in pair B and one more swap32 + one swap16 hand both halves to every row.  ~115
	// instructions per sample (two gates x 33 for the sums): 72.7 -> 38 us per 1024 x 128 step against the LDS-broadcast wave kernel.
	template <bool STD, bool B8>
	__device__ __forceinline__ void LstmDpp32Body(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, float* xin, float* hout)
	{
		constexpr int HP = 33;
		const int lane = (int)threadIdx.x & 63; // (the two-wave kernel runs these bodies on the waves of a 256-thread workgroup)
		const int j = lane & 15, rho = lane >> 4, r = rho & 1, pair = rho >> 1; // pair 0 = A (i, g), 1 = B (f, o)
		const int hr = m.hidden, unit = 16 * r + j;
		const bool real = unit < hr;
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;
		const GateK<STD> K0 = MakeGateK<STD>(false), K1 = MakeGateK<STD>(pair == 0);
		const float gs0 = GateRowScale<STD>(false), gs1 = GateRowScale<STD>(pair == 0);

		// W row-major [4 hr][1 + hr], then bias[4 hr] (LSTM.h:42-56); gate blocks i, f, g, o; columns rotated for the DPP walk
		const float* w0 = m.w + m.layerOff[0];
		const int R0 = (0 + pair) * hr + unit, R1 = (2 + pair) * hr + unit;
		auto colA = [&](int k) { return (j - k + 16) & 15; };
		const float wx0 = gs0 * LoadIf(w0, (size_t)R0 * (1 + hr), real), wx1 = gs1 * LoadIf(w0, (size_t)R1 * (1 + hr), real);
		const float b0 = gs0 * LoadIf(w0, (size_t)4 * hr * (1 + hr) + R0, real), b1 = gs1 * LoadIf(w0, (size_t)4 * hr * (1 + hr) + R1, real);
		// B8 (hidden <= 24): the upper vector has at most 8 units and is kept TWICE in every row ([h16..h23 | h16..h23]), so its walk
		// takes 8 terms instead of 16 (the same trick as the 8-unit layout)
		constexpr int NB = B8 ? 8 : 16;
		float wa0[16], wa1[16], wb0[NB], wb1[NB];
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			const int ca = colA(k);
			wa0[k] = gs0 * LoadIf(w0, (size_t)R0 * (1 + hr) + 1 + ca, real);
			wa1[k] = gs1 * LoadIf(w0, (size_t)R1 * (1 + hr) + 1 + ca, real);
		}
#pragma unroll
		for (int k = 0; k < NB; k++)
		{
			const int cb = 16 + ((j - k + 16) & (NB - 1));
			wb0[k] = gs0 * LoadIf(w0, (size_t)R0 * (1 + hr) + 1 + cb, real && cb < hr);
			wb1[k] = gs1 * LoadIf(w0, (size_t)R1 * (1 + hr) + 1 + cb, real && cb < hr);
		}

		for (int f = lane; f < n + 4; f += 64) xin[f] = f < n ? inRow[f] : 0.0f;
		float a = state[(size_t)j * capacity + slot];                                      // h[j], every row
		float b = LoadIf(state, (size_t)(16 + (j & (NB - 1))) * capacity + slot, 16 + (j & (NB - 1)) < hr); // h[16 + j] (B8: h[16 + j % 8]), every row
		float c = LoadIf(state, (size_t)(hr + unit) * capacity + slot, real);
		RecurrentWaveSync();

		float* hw = hout + unit;
		auto step = [&](float x, float* dst) {
			float acc0, acc1;
			DppDotFrom<16>(acc0, wx0, x, b0, wa0, a); // LSTM.h:168 -- column 0 is the input sample
			DppDotFrom<16>(acc1, wx1, x, b1, wa1, a);
			*dst = r ? b : a; // the h before this sample (entry f of hout), placed here: see LstmDppBodyM
			acc0 = __builtin_fmaf(wb0[0], b, acc0);
			acc1 = __builtin_fmaf(wb1[0], b, acc1);
			DppDotTail<NB>(acc0, wb0, b);
			DppDotTail<NB>(acc1, wb1, b);
			const float g0 = GateAct<STD>(acc0, K0); // i (pair A) / f (pair B)
			const float g1 = GateAct<STD>(acc1, K1); // g / o
			// i g in pair A, f c in pair B; both halves of the wave get both through one swap
			int t = __builtin_bit_cast(int, g0 * (pair == 0 ? g1 : c)), u;
			asm volatile("v_mov_b32 %1, %0\ns_nop 1\nv_permlane32_swap_b32 %0, %1\n" : "+v"(t), "=&v"(u)); // t: i g everywhere, u: f c everywhere
			c = __builtin_bit_cast(float, t) + __builtin_bit_cast(float, u);
			// h = o tanh(c') is right in pair B (rows 2, 3 = h[0..15], h[16..31]); y <- [h0 h1 h0 h1], then a <- h0, b <- h1 in every row
			int hv = __builtin_bit_cast(int, g1 * (STD ? StdTanh(c) : LstmRcpTanh(c))), y, z;
			asm volatile(
				"v_mov_b32 %1, %0\n"
				"s_nop 1\n"
				"v_permlane32_swap_b32 %0, %1\n"
				"v_mov_b32 %2, %1\n"
				"s_nop 1\n"
				"v_permlane16_swap_b32 %1, %2\n"
				: "+v"(hv), "=&v"(y), "=&v"(z));
			a = __builtin_bit_cast(float, y);
			b = __builtin_bit_cast(float, B8 ? RowLowHalf(z) : z); // B8: lanes 8..15 of a row <- lanes 0..7
		};
		int f = 0;
		for (; f + 4 <= n; f += 4)
		{
			const float4 xv = *reinterpret_cast<const float4*>(xin + f);
			step(xv.x, hw + (f + 0) * HP);
			step(xv.y, hw + (f + 1) * HP);
			step(xv.z, hw + (f + 2) * HP);
			step(xv.w, hw + (f + 3) * HP);
		}
		for (; f < n; f++) step(xin[f], hw + f * HP);
		hw[n * HP] = r ? b : a;
		RecurrentWaveSync();

		// dense head for the whole block, lane = sample (LSTM.h:182-189)
		const float* headW = m.w + m.headOff;
		for (int f = lane; f < n; f += 64)
		{
			float acc = 0.0f;
			for (int k = 0; k < hr; k++) acc += headW[k] * hout[(f + 1) * HP + k];
			outRow[f] = acc + headW[hr];
		}
		if (pair == 0 && real)
		{
			state[(size_t)unit * capacity + slot] = r ? b : a;
			state[(size_t)(hr + unit) * capacity + slot] = c;
		}
	}

	template <int H, int L>
	__device__ __forceinline__ void LstmDppBody(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, float* xin, float* hout)
	{
		LstmDppBodyM<H, L, false>(m, state, capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
	}

	template <int H, int L>
	__device__ __forceinline__ void LstmDppBodyStd(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, float* xin, float* hout)
	{
		LstmDppBodyM<H, L, true>(m, state, capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
	}

	// ------------------------------------------------------------------------------------------------------------
	// H = 8 / 16: nothing on the recurrence touches LDS (same idea as the LSTM bodies above).  lane = H*gate + unit with gate rows z, r, c
	// and the fourth row duplicating c (for H = 8 the upper 32 lanes mirror the lower 32); every lane keeps h[unit].  The mat-vec
	// reads h[(unit - n) mod H] with DPP row_ror:n against weights rotated at load time; z and r reach every lane through two
	// lane swaps, c through one.  Each lane sums its row starting at column `unit` (different rounding order than the plain
	// kernel; ~1e-7 RMS).
	// ------------------------------------------------------------------------------------------------------------
	// One GRU cell update (keras reset_after form, RTNeural GRULayer): acc = the recurrent row sum, which on the z and r rows already
	// holds the input part as well (folded into the start of the sum: the per-lane wxA / bA below), on the c rows only b_rec + U h;
	// aic = W x + b_in of this lane's row (used on the c rows).  sigmoid = rcp(1 + exp2(-x log2 e)), tanh = 1 - 2 rcp(exp2(2 x log2 e) + 1).
	template <int H>
	__device__ __forceinline__ float GruDppCell(float acc, float aic, float h)
	{
		const float zrv = StdSigmoid(acc); // meaningful on the z and r rows
		float z, r;
		if constexpr (H == 16)
		{
			int zr = __builtin_bit_cast(int, zrv), zr2, rr;
			asm volatile(
				"s_nop 0\n" // zrv comes straight from v_rcp_f32: one wait state before a VALU may read a transcendental's result
				"v_mov_b32 %1, %0\n"
				"s_nop 1\n"
				"v_permlane32_swap_b32 %0, %1\n" // zr: rows z r z r
				"v_mov_b32 %2, %0\n"
				"s_nop 1\n"
				"v_permlane16_swap_b32 %0, %2\n" // zr: z everywhere, rr: r everywhere
				: "+v"(zr), "=&v"(zr2), "=&v"(rr));
			z = __builtin_bit_cast(float, zr);
			r = __builtin_bit_cast(float, rr);
		}
		else
		{
			int zr = __builtin_bit_cast(int, zrv), zr2;
			asm volatile("s_nop 0\nv_mov_b32 %1, %0\ns_nop 1\nv_permlane16_swap_b32 %0, %1\n" : "+v"(zr), "=&v"(zr2)); // zr: every row = [z | r] (the DPP reads below: the compiler pads them)
			z = __builtin_bit_cast(float, RowLowHalf(zr));
			r = __builtin_bit_cast(float, RowHighHalf(zr));
		}
		int c = __builtin_bit_cast(int, StdTanh(__builtin_fmaf(r, acc, aic))); // meaningful on the c rows (2 and 3 for H = 16; 1 and 3 for H = 8)
		int c2;
		if constexpr (H == 16) asm volatile("v_mov_b32 %1, %0\ns_nop 1\nv_permlane32_swap_b32 %0, %1\n" : "+v"(c), "=&v"(c2)); // c2: rows c c c c
		else asm volatile("v_mov_b32 %1, %0\ns_nop 1\nv_permlane16_swap_b32 %0, %1\n" : "+v"(c), "=&v"(c2));                 // c2: rows 1 1 3 3 = c everywhere
		const float cv = __builtin_bit_cast(float, c2);
		return __builtin_fmaf(z, h - cv, cv); // (1 - z) c + z h
	}

	template <int H, int L>
	__device__ __forceinline__ void GruDppBody(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, float* xin, float* hout)
	{
		static_assert(H == 8 || H == 16, "a 16-lane DPP row must hold the units a whole number of times");
		constexpr int HP = H + 1;
		const int lane = (int)threadIdx.x & 63; // (the two-wave kernel runs these bodies on the waves of a 256-thread workgroup)
		const int unit = lane % H;
		const int gate = min((lane / H) & 3, 2); // rows z, r, c, c
		const bool isC = gate == 2;
		const int hr = m.hidden; // <= H: surplus units carry zero weights and zero state (z = 0.5, c = 0: h stays 0)
		const bool real = unit < hr;
		const int r = gate * hr + unit;
		auto col = [&](int k) { return (unit - k + H) % H; };
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;

		// layer 0: W row-major [3H][1 + H], b_in[3H], b_rec[3H]; h weights rotated so that row_ror:k pairs wh0[k] with h[(unit - k) mod H].
		// z / r rows: the input part joins the recurrent sum (start wx x + (b_in + b_rec)); c rows keep it apart (the reset gate scales
		// the recurrent part only).
		const float* w0 = m.w + m.layerOff[0];
		const float wx0 = LoadIf(w0, (size_t)r * (1 + hr), real);
		float wh0[H];
#pragma unroll
		for (int k = 0; k < H; k++) wh0[k] = LoadIf(w0, (size_t)r * (1 + hr) + 1 + col(k), real && col(k) < hr);
		const float bi0 = LoadIf(w0, (size_t)3 * hr * (1 + hr) + r, real), bh0 = LoadIf(w0, (size_t)3 * hr * (1 + hr) + 3 * hr + r, real);
		const float wxA0 = isC ? 0.0f : wx0, bA0 = isC ? bh0 : bi0 + bh0;
		float wi1[H], wh1[H];
		float bi1 = 0.0f, bh1 = 0.0f;
		if (L > 1)
		{
			const float* w1 = m.w + m.layerOff[L > 1 ? 1 : 0];
#pragma unroll
			for (int k = 0; k < H; k++)
			{
				const bool on = real && col(k) < hr;
				wi1[k] = LoadIf(w1, (size_t)r * (2 * hr) + col(k), on);
				wh1[k] = LoadIf(w1, (size_t)r * (2 * hr) + hr + col(k), on);
			}
			bi1 = LoadIf(w1, (size_t)3 * hr * (2 * hr) + r, real);
			bh1 = LoadIf(w1, (size_t)3 * hr * (2 * hr) + 3 * hr + r, real);
		}

		for (int f = lane; f < n + 4; f += 64) xin[f] = f < n ? inRow[f] : 0.0f;
		float h[L];
#pragma unroll
		for (int l = 0; l < L; l++) h[l] = LoadIf(state, (size_t)(l * 2 * hr + unit) * capacity + slot, real);
		RecurrentWaveSync();

		// (stores of h: every lane, after the first dot of the NEXT sample -- see LstmDppBodyM)
		float* hw = hout + unit;
		auto step = [&](float x, float* dst) {
			float acc;
			DppDotFrom<H>(acc, wxA0, x, bA0, wh0, h[0]);
			*dst = h[L - 1];
			h[0] = GruDppCell<H>(acc, __builtin_fmaf(wx0, x, bi0), h[0]);
			if constexpr (L > 1)
			{
				// layer 1: the input part is a dot of its own; on the z / r rows it is added to the recurrent sum, on the c rows it is `aic`
				float ai1, ah1;
				DppDotFrom<H>(ai1, 0.0f, 0.0f, bi1, wi1, h[0]);
				DppDotFrom<H>(ah1, 0.0f, 0.0f, bh1, wh1, h[1]);
				h[1] = GruDppCell<H>(isC ? ah1 : ai1 + ah1, ai1, h[1]);
			}
		};
		int f = 0;
		for (; f + 4 <= n; f += 4)
		{
			const float4 xv = *reinterpret_cast<const float4*>(xin + f);
			step(xv.x, hw + (f + 0) * HP);
			step(xv.y, hw + (f + 1) * HP);
			step(xv.z, hw + (f + 2) * HP);
			step(xv.w, hw + (f + 3) * HP);
		}
		for (; f < n; f++) step(xin[f], hw + f * HP);
		hw[n * HP] = h[L - 1];
		RecurrentWaveSync();

		const float* headW = m.w + m.headOff;
		for (int f = lane; f < n; f += 64)
		{
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < H; k++) acc += LoadIf(headW, (size_t)k, k < hr) * hout[(f + 1) * HP + k];
			outRow[f] = acc + headW[hr];
		}
		if (lane < hr)
		{
#pragma unroll
			for (int l = 0; l < L; l++) state[(size_t)(l * 2 * hr + lane) * capacity + slot] = h[l];
		}
	}


	// One keras GRU layer of 17 .. 32 units on the 32-unit layout (see LstmDpp32Body): rows 0-1 (pair A) own the z gate of unit 16 r + j,
	// rows 2-3 (pair B) the r gate AND the candidate c -- c = tanh(W_c x + b_in + r (U_c h + b_rec)) needs r and the recurrent sum of c in
	// one lane, which is exactly what pair B holds.  Slot 0 (z | r) is a sigmoid everywhere; ONE v_permlane32_swap then hands z and c to
	// every row, every lane updates its own unit (h' = c + z (h - c)), and one v_permlane16_swap rebuilds the two replicated state
	// vectors.  ~87 instructions per sample; 1x24 158 -> 32 us per 1024 x 128 step against the runtime-shaped wave kernel.
	template <bool B8>
	__device__ __forceinline__ void GruDpp32Body(const LstmModelDev& m, float* __restrict__ state, int capacity, int slot, int row, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, float* xin, float* hout)
	{
		constexpr int HP = 33;
		constexpr int NB = B8 ? 8 : 16;
		const int lane = (int)threadIdx.x & 63; // (the two-wave kernel runs these bodies on the waves of a 256-thread workgroup)
		const int j = lane & 15, rho = lane >> 4, r = rho & 1, pair = rho >> 1; // pair 0 = A (z), 1 = B (r, c)
		const int hr = m.hidden, unit = 16 * r + j;
		const bool real = unit < hr, isB = pair == 1;
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;

		// W row-major [3 hr][1 + hr] (rows z | r | c), b_in[3 hr], b_rec[3 hr]; recurrent columns rotated for the DPP walk
		const float* w0 = m.w + m.layerOff[0];
		const size_t W = (size_t)(1 + hr), bIn = (size_t)3 * hr * W, bRec = bIn + (size_t)3 * hr;
		const int R0 = pair * hr + unit, R1 = 2 * hr + unit;
		const float wx0 = LoadIf(w0, (size_t)R0 * W, real);                                        // z / r: the input part joins the sum
		const float bA0 = LoadIf(w0, bIn + R0, real) + LoadIf(w0, bRec + R0, real);
		const float wx1 = LoadIf(w0, (size_t)R1 * W, real && isB), bi1 = LoadIf(w0, bIn + R1, real && isB); // c: W x + b_in kept apart
		const float bh1 = LoadIf(w0, bRec + R1, real && isB);
		float wa0[16], wa1[16], wb0[NB], wb1[NB];
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			const int ca = (j - k + 16) & 15;
			wa0[k] = LoadIf(w0, (size_t)R0 * W + 1 + ca, real);
			wa1[k] = LoadIf(w0, (size_t)R1 * W + 1 + ca, real && isB);
		}
#pragma unroll
		for (int k = 0; k < NB; k++)
		{
			const int cb = 16 + ((j - k + 16) & (NB - 1));
			wb0[k] = LoadIf(w0, (size_t)R0 * W + 1 + cb, real && cb < hr);
			wb1[k] = LoadIf(w0, (size_t)R1 * W + 1 + cb, real && isB && cb < hr);
		}

		for (int f = lane; f < n + 4; f += 64) xin[f] = f < n ? inRow[f] : 0.0f;
		float a = state[(size_t)j * capacity + slot];                                                               // h[j], every row
		float b = LoadIf(state, (size_t)(16 + (j & (NB - 1))) * capacity + slot, 16 + (j & (NB - 1)) < hr);          // h[16 + j] (B8: h[16 + j % 8])
		float hown = r ? b : a;
		RecurrentWaveSync();

		float* hw = hout + unit;
		auto step = [&](float x, float* dst) {
			float s0, s1;
			DppDotFrom<16>(s0, wx0, x, bA0, wa0, a);
			DppDotFrom<16>(s1, 0.0f, 0.0f, bh1, wa1, a);
			*dst = hown; // the h before this sample (entry f of hout), placed here: see LstmDppBodyM
			s0 = __builtin_fmaf(wb0[0], b, s0);
			s1 = __builtin_fmaf(wb1[0], b, s1);
			DppDotTail<NB>(s0, wb0, b);
			DppDotTail<NB>(s1, wb1, b);
			const float zr = StdSigmoid(s0);                                             // z (pair A) / r (pair B)
			const float cv = StdTanh(__builtin_fmaf(zr, s1, __builtin_fmaf(wx1, x, bi1))); // c, right in pair B
			int v = __builtin_bit_cast(int, isB ? cv : zr), vc;
			asm volatile("v_mov_b32 %1, %0\ns_nop 1\nv_permlane32_swap_b32 %0, %1\n" : "+v"(v), "=&v"(vc)); // v: z everywhere, vc: c everywhere
			const float z = __builtin_bit_cast(float, v), c = __builtin_bit_cast(float, vc);
			int hn = __builtin_bit_cast(int, __builtin_fmaf(z, hown - c, c)), hc; // (1 - z) c + z h, every lane for its own unit
			asm volatile("v_mov_b32 %1, %0\ns_nop 1\nv_permlane16_swap_b32 %0, %1\n" : "+v"(hn), "=&v"(hc)); // hn: h[0..15] everywhere, hc: h[16..31]
			a = __builtin_bit_cast(float, hn);
			b = __builtin_bit_cast(float, B8 ? RowLowHalf(hc) : hc);
			hown = r ? b : a;
		};
		int f = 0;
		for (; f + 4 <= n; f += 4)
		{
			const float4 xv = *reinterpret_cast<const float4*>(xin + f);
			step(xv.x, hw + (f + 0) * HP);
			step(xv.y, hw + (f + 1) * HP);
			step(xv.z, hw + (f + 2) * HP);
			step(xv.w, hw + (f + 3) * HP);
		}
		for (; f < n; f++) step(xin[f], hw + f * HP);
		hw[n * HP] = hown;
		RecurrentWaveSync();

		const float* headW = m.w + m.headOff;
		for (int f = lane; f < n; f += 64)
		{
			float acc = 0.0f;
			for (int k = 0; k < hr; k++) acc += headW[k] * hout[(f + 1) * HP + k];
			outRow[f] = acc + headW[hr];
		}
		if (pair == 0 && real) state[(size_t)unit * capacity + slot] = hown;
	}

	struct RecurrentGroupArgs
	{
		LstmModelDev m;
		float* state;
		const int* slots;
		const int* rows;
		int capacity, numStreams;
		int slot0, row0; // slots == nullptr: contiguous
		int firstBlock; // workgroups (= streams) [firstBlock, next group's firstBlock) belong to this group
	};

	struct RecurrentLaunchArgs
	{
		RecurrentGroupArgs g[RECURRENT_MAX_GROUPS];
		int numGroups;
		int noSkew; // tuning / tests (NA_REC_NOSKEW): two-layer H = 8 LSTMs on the sequential body
		int houtWave; // four-wave workgroups (RecurrentDppKernel<true>): floats of dynamic LDS per wave
	};

	// ------------------------------------------------------------------------------------------------------------
	// Large batches (thousands of streams: more waves than the chip has SIMDs, so the launch is bound by instruction issue and not by
	// the latency of a lone wave): FOUR streams per wave.  lane = 16 * stream + unit and every lane owns ALL FOUR gate rows of its unit,
	// as two packed pairs (i, f) and (g, o): a row sum is v_pk_fma_f32 over the 16 state values of the lane's stream -- 2 x 17 packed
	// instructions per sample for four streams, where the one-stream layout spends 17 (and its DPP terms cost ~6 cycles each, measured,
	// against ~4.3 for a plain or packed VALU instruction).  The state reaches the lanes through LDS: every lane writes the h of its unit
	// (the entry the dense head reads afterwards anyway) and reads the 16 values of its stream back as 16-byte broadcasts (of {h, h} pairs: see QUAD_HP).  The gates
	// of a unit meet in one lane (no lane swaps), the activations run on packed pairs with the gate's identity in per-component constants,
	// tanh(c) on full lanes.  ~80 instructions per sample for four streams against 52 per stream.  The dependent chain of a wave is
	// longer, so the one-stream layout stays the choice for batches that leave SIMDs idle (LaunchRecurrentDpp picks by the stream count).
	// Row sums run over the columns in the reference's order (input, then h[0 .. hr - 1]; LSTM.h:87-100), not rotated by the unit as in
	// the one-stream layout: the two layouts agree to rounding (~1e-7), not bit for bit.  Hidden sizes below 16 are zero-padded.
	// ------------------------------------------------------------------------------------------------------------
	// NA_QUAD_NOPK (tuning builds): the pairs as two scalars -- no v_pk_*_f32 in the kernel (10 - 17 % slower).  How the op_sel hazard of
	// QUAD_HP below was cornered (profiles/r06_quad_race.txt): this build was the first variant that did not show it.
#ifndef NA_QUAD_NOPK
#define NA_QUAD_NOPK 0
#endif
#if NA_QUAD_NOPK
	struct quad_f2
	{
		float x, y;
	};
	__device__ __forceinline__ quad_f2 operator*(quad_f2 a, quad_f2 b) { return quad_f2{ a.x * b.x, a.y * b.y }; }
	__device__ __forceinline__ quad_f2 operator+(quad_f2 a, quad_f2 b) { return quad_f2{ a.x + b.x, a.y + b.y }; }
#else
	typedef float quad_f2 __attribute__((ext_vector_type(2)));
#endif
	constexpr int QUAD_CHUNK = 16;                      // samples between two head passes (bounds the LDS of a wave)
	constexpr int QUAD_XROW = LSTM_MAX_FRAMES + 4;      // input samples of one stream in LDS (+4: the float4 reads may run past the block)
	// An h entry holds every unit TWICE, {h[k], h[k]} pairs: the row sums are packed FMAs against such a pair, and the pair has to come out of
	// LDS as it is.  Rounds 3 - 5 kept one copy and let the FMA spread it (op_sel / op_sel_hi on the operand the ds_read_b128 had just
	// delivered): v_pk_fma_f32 with a non-default op_sel on an LDS-DELIVERED register computes garbage in lanes 48 .. 63 every now and then
	// while waves of another kernel issue f16 MFMAs on the same SIMD (tools/microbench/pk_lds_opsel.hip: 0.7 - 1.5 M mismatches per run beside
	// MFMA waves, none beside anything else, none with pairs out of LDS, none with copies, none with VALU-born operands;
	// profiles/r06_quad_race.txt) -- the fourth stream of a wave went wrong beside the f16-split WaveNet kernels.
	constexpr int QUAD_HP = 40;                         // floats per h entry: 16 pairs + padding (16-byte aligned rows, spread over the banks)
	constexpr int QUAD_HROW = (QUAD_CHUNK + 1) * QUAD_HP; // output-layer h of one stream: the state before the chunk, then after each of its samples
	constexpr int QUAD_LDS_FLOATS = 4 * QUAD_XROW + 4 * QUAD_HROW;
	// an input sample out of LDS, through a VALU copy (it is broadcast with op_sel below: see QUAD_HP)
	__device__ __forceinline__ float QuadCopy(float v)
	{
		float r;
		asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
		return r;
	}

#if NA_QUAD_NOPK
	__device__ __forceinline__ quad_f2 QuadFma(quad_f2 a, quad_f2 b, quad_f2 c) { return quad_f2{ __builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y) }; }
#else
	__device__ __forceinline__ quad_f2 QuadFma(quad_f2 a, quad_f2 b, quad_f2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
	__device__ __forceinline__ quad_f2 QuadSplat(float v) { return quad_f2{ v, v }; }

	// GateAct on a pair of gate rows: component 0 / 1 take the constants of their own gate (sigmoid or tanh)
	template <bool STD>
	struct QuadGateK
	{
		quad_f2 a, b, c, B; // FastMath: aA, bA, cA, B of GateK<false>; StdMath: a = k, b = A, c unused
	};
	template <bool STD>
	__device__ __forceinline__ QuadGateK<STD> MakeQuadGateK(bool g0, bool g1)
	{
		const GateK<STD> k0 = MakeGateK<STD>(g0), k1 = MakeGateK<STD>(g1);
		QuadGateK<STD> K;
		if constexpr (STD)
		{
			K.a = quad_f2{ k0.k, k1.k };
			K.b = quad_f2{ k0.A, k1.A };
			K.c = quad_f2{ 0.0f, 0.0f };
			K.B = quad_f2{ k0.B, k1.B };
		}
		else
		{
			K.a = quad_f2{ k0.aA, k1.aA };
			K.b = quad_f2{ k0.bA, k1.bA };
			K.c = quad_f2{ k0.cA, k1.cA };
			K.B = quad_f2{ k0.B, k1.B };
		}
		return K;
	}
	template <bool STD>
	__device__ __forceinline__ quad_f2 QuadGateAct(quad_f2 y, const QuadGateK<STD>& K)
	{
		if constexpr (STD)
		{
			const quad_f2 e = y * K.a;
			const quad_f2 r = quad_f2{ __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(e.x) + 1.0f), __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(e.y) + 1.0f) };
			return QuadFma(r, K.b, K.B);
		}
		else
		{
			// (same expression tree as GateAct<false>, two rows at a time)
			const quad_f2 ay = quad_f2{ fabsf(y.x), fabsf(y.y) };
			const quad_f2 y2 = y * y;
			const quad_f2 p = QuadFma(y2, QuadFma(ay, K.c, K.b), QuadFma(ay, K.a, K.a));
			const quad_f2 den = QuadFma(QuadSplat(2.44506634652299f) + y2, QuadFma(QuadSplat(0.814642734961073f), y2, ay), QuadSplat(2.44506634652299f));
			const quad_f2 r = quad_f2{ __builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y) };
			return QuadFma(y * p, r, K.B);
		}
	}

	template <bool STD>
	__device__ __forceinline__ void LstmQuadBody(const RecurrentGroupArgs& ga, int idx0, const float* __restrict__ in, float* __restrict__ out, long inStride,
		long outStride, int n, float* lds)
	{
		constexpr int H = 16, HP = QUAD_HP;
		const LstmModelDev& m = ga.m;
		float* __restrict__ state = ga.state;
		const int capacity = ga.capacity;
		const int lane = threadIdx.x, unit = lane & 15, sub = lane >> 4;
		const int hr = m.hidden;
		const bool real = unit < hr;
		const bool live = idx0 + sub < ga.numStreams; // a short last wave: the surplus rows repeat the last stream and store nothing
		const int idx = live ? idx0 + sub : ga.numStreams - 1;
		const int slot = ga.slots ? ga.slots[idx] : ga.slot0 + idx;
		const int row = ga.slots ? ga.rows[idx] : ga.row0 + idx;
		const QuadGateK<STD> KIF = MakeQuadGateK<STD>(false, false), KGO = MakeQuadGateK<STD>(true, false);
		const float sS = GateRowScale<STD>(false), sT = GateRowScale<STD>(true); // (the inner 0.5 of a sigmoid rides in its row: see GateAct)

		// layer 0: W row-major [4 hr][1 + hr], then bias[4 hr] (LSTM.h:42-56); gate blocks i, f, g, o.  Pair IF = rows (i, f), GO = (g, o)
		const float* w0 = m.w + m.layerOff[0];
		const int W0 = 1 + hr;
		auto pairOf = [&](const float* w, int rowStride, int col, bool on, int qa, float sa, int qb, float sb) {
			return quad_f2{ sa * LoadIf(w, (size_t)(qa * hr + unit) * rowStride + col, on), sb * LoadIf(w, (size_t)(qb * hr + unit) * rowStride + col, on) };
		};
		const quad_f2 wxIF = pairOf(w0, W0, 0, real, 0, sS, 1, sS), wxGO = pairOf(w0, W0, 0, real, 2, sT, 3, sS);
		const quad_f2 b0IF = quad_f2{ sS * LoadIf(w0, (size_t)4 * hr * W0 + 0 * hr + unit, real), sS * LoadIf(w0, (size_t)4 * hr * W0 + 1 * hr + unit, real) };
		const quad_f2 b0GO = quad_f2{ sT * LoadIf(w0, (size_t)4 * hr * W0 + 2 * hr + unit, real), sS * LoadIf(w0, (size_t)4 * hr * W0 + 3 * hr + unit, real) };
		quad_f2 wh0IF[H], wh0GO[H];
#pragma unroll
		for (int k = 0; k < H; k++)
		{
			wh0IF[k] = pairOf(w0, W0, 1 + k, real && k < hr, 0, sS, 1, sS);
			wh0GO[k] = pairOf(w0, W0, 1 + k, real && k < hr, 2, sT, 3, sS);
		}
		// LDS: the input block of the four streams [stream][QUAD_XROW]; their h [stream][entry][HP]
		float* xin = lds;
		float* hout = lds + 4 * QUAD_XROW;
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const float* inRow = in + (size_t)__builtin_amdgcn_readlane(row, 16 * s) * inStride;
			for (int f = lane; f < n + 4; f += 64) xin[s * QUAD_XROW + f] = f < n ? inRow[f] : 0.0f;
		}
		float h = LoadIf(state, (size_t)unit * capacity + slot, real);
		float c = LoadIf(state, (size_t)(hr + unit) * capacity + slot, real);
		quad_f2* hw = reinterpret_cast<quad_f2*>(hout + sub * QUAD_HROW + 2 * unit); // this lane's pair of an entry of h
		const float* hrd = hout + sub * QUAD_HROW;                                  // the 16 pairs of an entry
		constexpr int HPP = HP / 2;                                                 // entry stride in pairs
		hw[0] = quad_f2{ h, h };
		RecurrentWaveSync();

		auto read16 = [&](const float* p, quad_f2 (&v)[H]) {
#pragma unroll
			for (int q = 0; q < 8; q++)
			{
				const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
				v[2 * q + 0] = quad_f2{ t.x, t.y };
				v[2 * q + 1] = quad_f2{ t.z, t.w };
			}
		};
		auto cell = [&](quad_f2 aIF, quad_f2 aGO, float& cc) {
			const quad_f2 gIF = QuadGateAct<STD>(aIF, KIF), gGO = QuadGateAct<STD>(aGO, KGO);
			cc = __builtin_fmaf(gIF.y, cc, gIF.x * gGO.x);
			return gGO.y * (STD ? StdTanh(cc) : LstmRcpTanh(cc));
		};
		quad_f2 hv0[H]; // h, all units of this lane's stream, as {h[k], h[k]}
		read16(hrd, hv0);
		// one sample: entry e of hout = h after sample e - 1 of the chunk
		auto step = [&](float x, int e) {
			quad_f2 aIF = QuadFma(wxIF, QuadSplat(x), b0IF), aGO = QuadFma(wxGO, QuadSplat(x), b0GO); // LSTM.h:168 -- column 0 is the input sample
#pragma unroll
			for (int k = 0; k < H; k++)
			{
				aIF = QuadFma(wh0IF[k], hv0[k], aIF);
				aGO = QuadFma(wh0GO[k], hv0[k], aGO);
			}
			h = cell(aIF, aGO, c);
			hw[(e + 1) * HPP] = quad_f2{ h, h };
			RecurrentWaveSync();
			read16(hrd + (e + 1) * HP, hv0);
		};

		const float* xs = xin + sub * QUAD_XROW;
		const float* headW = m.w + m.headOff;
		for (int f0 = 0; f0 < n; f0 += QUAD_CHUNK)
		{
			const int cn = min(QUAD_CHUNK, n - f0);
			int f = 0;
			for (; f + 4 <= cn; f += 4)
			{
				const float4 xv = *reinterpret_cast<const float4*>(xs + f0 + f);
				step(QuadCopy(xv.x), f + 0);
				step(QuadCopy(xv.y), f + 1);
				step(QuadCopy(xv.z), f + 2);
				step(QuadCopy(xv.w), f + 3);
			}
			for (; f < cn; f++) step(QuadCopy(xs[f0 + f]), f);
			// dense head of the chunk (LSTM.h:182-189): output o = QUAD_CHUNK * stream + sample, one per lane
			{
				const int s = lane / QUAD_CHUNK, ff = lane % QUAD_CHUNK;
				const float* hs = hout + s * QUAD_HROW + (ff + 1) * HP;
				float acc = 0.0f;
#pragma unroll
				for (int k = 0; k < H; k++) acc += LoadIf(headW, (size_t)k, k < hr) * hs[2 * k];
				const int orow = __builtin_amdgcn_ds_bpermute(4 * (16 * s), row);
				if (ff < cn && idx0 + s < ga.numStreams) out[(size_t)orow * outStride + f0 + ff] = acc + headW[hr];
			}
			RecurrentWaveSync();
			hw[0] = quad_f2{ h, h }; // entry 0 of the next chunk
			RecurrentWaveSync();
		}
		if (real && live)
		{
			state[(size_t)unit * capacity + slot] = h;
			state[(size_t)(hr + unit) * capacity + slot] = c;
		}
	}

	// One keras GRU layer of up to 16 units (reset_after form, RTNeural GRULayer; see GruDppCell) on the same four-streams-per-wave layout:
	// the z and r rows of a unit as one packed pair, the recurrent sum of the candidate row as (even terms, odd terms) against the
	// (h[k], h[k + 1]) pairs the LDS read-back delivers anyway.  ~50 instructions per sample for four streams (one-stream layout: ~45 per
	// stream).  sigmoid / tanh on the exp2 / rcp units like every GRU kernel here (StdSigmoid, StdTanh).
	__device__ __forceinline__ void GruQuadBody(const RecurrentGroupArgs& ga, int idx0, const float* __restrict__ in, float* __restrict__ out, long inStride,
		long outStride, int n, float* lds)
	{
		constexpr int H = 16, HP = QUAD_HP;
		const LstmModelDev& m = ga.m;
		float* __restrict__ state = ga.state;
		const int capacity = ga.capacity;
		const int lane = threadIdx.x, unit = lane & 15, sub = lane >> 4;
		const int hr = m.hidden;
		const bool real = unit < hr;
		const bool live = idx0 + sub < ga.numStreams;
		const int idx = live ? idx0 + sub : ga.numStreams - 1;
		const int slot = ga.slots ? ga.slots[idx] : ga.slot0 + idx;
		const int row = ga.slots ? ga.rows[idx] : ga.row0 + idx;

		// W row-major [3 hr][1 + hr] (rows z | r | c), b_in[3 hr], b_rec[3 hr]
		const float* w0 = m.w + m.layerOff[0];
		const int W0 = 1 + hr;
		const size_t bIn = (size_t)3 * hr * W0, bRec = bIn + (size_t)3 * hr;
		const int rz = unit, rr = hr + unit, rc = 2 * hr + unit;
		const quad_f2 wxZR = quad_f2{ LoadIf(w0, (size_t)rz * W0, real), LoadIf(w0, (size_t)rr * W0, real) };
		const quad_f2 bZR = quad_f2{ LoadIf(w0, bIn + rz, real) + LoadIf(w0, bRec + rz, real), LoadIf(w0, bIn + rr, real) + LoadIf(w0, bRec + rr, real) };
		const float wxC = LoadIf(w0, (size_t)rc * W0, real), biC = LoadIf(w0, bIn + rc, real), bhC = LoadIf(w0, bRec + rc, real);
		quad_f2 whZR[H], whC[H / 2];
#pragma unroll
		for (int k = 0; k < H; k++) whZR[k] = quad_f2{ LoadIf(w0, (size_t)rz * W0 + 1 + k, real && k < hr), LoadIf(w0, (size_t)rr * W0 + 1 + k, real && k < hr) };
#pragma unroll
		for (int k = 0; k < H / 2; k++)
			whC[k] = quad_f2{ LoadIf(w0, (size_t)rc * W0 + 1 + 2 * k, real && 2 * k < hr), LoadIf(w0, (size_t)rc * W0 + 2 + 2 * k, real && 2 * k + 1 < hr) };

		float* xin = lds;
		float* hout = lds + 4 * QUAD_XROW;
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const float* inRow = in + (size_t)__builtin_amdgcn_readlane(row, 16 * s) * inStride;
			for (int f = lane; f < n + 4; f += 64) xin[s * QUAD_XROW + f] = f < n ? inRow[f] : 0.0f;
		}
		float h = LoadIf(state, (size_t)unit * capacity + slot, real);
		quad_f2* hw = reinterpret_cast<quad_f2*>(hout + sub * QUAD_HROW + 2 * unit); // this lane's {h, h} pair of an entry (see QUAD_HP)
		const float* hrd = hout + sub * QUAD_HROW;
		constexpr int HPP = HP / 2;
		hw[0] = quad_f2{ h, h };
		RecurrentWaveSync();

		quad_f2 hv[H]; // {h[k], h[k]} of this lane's stream
		auto read16 = [&](const float* p) {
#pragma unroll
			for (int q = 0; q < 8; q++)
			{
				const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
				hv[2 * q] = quad_f2{ t.x, t.y };
				hv[2 * q + 1] = quad_f2{ t.z, t.w };
			}
		};
		read16(hrd);
		auto step = [&](float x, int e) {
			quad_f2 aZR = QuadFma(wxZR, QuadSplat(x), bZR);
			quad_f2 aC = quad_f2{ bhC, 0.0f };
#pragma unroll
			for (int k = 0; k < H / 2; k++)
			{
				aZR = QuadFma(whZR[2 * k], hv[2 * k], aZR);
				aZR = QuadFma(whZR[2 * k + 1], hv[2 * k + 1], aZR);
				// the candidate row as (even terms, odd terms) like before, but with scalar FMAs: its operand pair (h[2k], h[2k + 1]) does not exist
				// in the duplicated layout, and scalar instructions may read delivered registers any way they like
				aC = quad_f2{ __builtin_fmaf(whC[k].x, hv[2 * k].x, aC.x), __builtin_fmaf(whC[k].y, hv[2 * k + 1].x, aC.y) };
			}
			const quad_f2 e2 = aZR * QuadSplat(-1.4426950408889634f);
			const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e2.x)), r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e2.y));
			const float c = StdTanh(__builtin_fmaf(r, aC.x + aC.y, __builtin_fmaf(wxC, x, biC)));
			h = __builtin_fmaf(z, h - c, c); // (1 - z) c + z h
			hw[(e + 1) * HPP] = quad_f2{ h, h };
			RecurrentWaveSync();
			read16(hrd + (e + 1) * HP);
		};

		const float* xs = xin + sub * QUAD_XROW;
		const float* headW = m.w + m.headOff;
		for (int f0 = 0; f0 < n; f0 += QUAD_CHUNK)
		{
			const int cn = min(QUAD_CHUNK, n - f0);
			int f = 0;
			for (; f + 4 <= cn; f += 4)
			{
				const float4 xv = *reinterpret_cast<const float4*>(xs + f0 + f);
				step(QuadCopy(xv.x), f + 0);
				step(QuadCopy(xv.y), f + 1);
				step(QuadCopy(xv.z), f + 2);
				step(QuadCopy(xv.w), f + 3);
			}
			for (; f < cn; f++) step(QuadCopy(xs[f0 + f]), f);
			{
				const int s = lane / QUAD_CHUNK, ff = lane % QUAD_CHUNK;
				const float* hs = hout + s * QUAD_HROW + (ff + 1) * HP;
				float acc = 0.0f;
#pragma unroll
				for (int k = 0; k < H; k++) acc += LoadIf(headW, (size_t)k, k < hr) * hs[2 * k];
				const int orow = __builtin_amdgcn_ds_bpermute(4 * (16 * s), row);
				if (ff < cn && idx0 + s < ga.numStreams) out[(size_t)orow * outStride + f0 + ff] = acc + headW[hr];
			}
			RecurrentWaveSync();
			hw[0] = quad_f2{ h, h };
			RecurrentWaveSync();
			read16(hrd); // (the same values: keeps the entry the next chunk starts from and the registers in one place)
		}
		if (real && live) state[(size_t)unit * capacity + slot] = h;
	}

	// grid = sum over the groups of ceil(streams / 4), block = 64 (four streams per wave); groups: one-layer LSTM or keras GRU, hidden <= 16.
	// (The gate weights live in registers: ~150 VGPRs, three waves per SIMD.)
	__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) RecurrentQuadKernel(const RecurrentLaunchArgs args,
		const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n)
	{
		__shared__ __attribute__((aligned(16))) float lds[QUAD_LDS_FLOATS];
		int gi = 0;
		for (int i = 1; i < args.numGroups; i++)
			if ((int)blockIdx.x >= args.g[i].firstBlock) gi = i;
		const RecurrentGroupArgs& ga = args.g[gi];
		const int idx0 = 4 * ((int)blockIdx.x - ga.firstBlock);
		if (ga.m.cell == LSTM_CELL_GRU) GruQuadBody(ga, idx0, in, out, inStride, outStride, n, lds);
		else if (ga.m.math == LSTM_MATH_STD) LstmQuadBody<true>(ga, idx0, in, out, inStride, outStride, n, lds);
		else LstmQuadBody<false>(ga, idx0, in, out, inStride, outStride, n, lds);
	}

	bool RecurrentQuadSupported(const LstmModelDev& m)
	{
		// One layer.  (Two layers were built and measured: the second layer's 128 weight registers per lane leave one wave per SIMD and the
		// layout loses to the one-stream kernel -- 2x16 x 8192: 199 vs 205 us, 2x8: 174 vs 95 us.)
		return m.tailLayers == 0 && (m.cell == LSTM_CELL_LSTM || m.cell == LSTM_CELL_GRU) && m.hidden >= 1 && m.hidden <= 16 && m.numLayers == 1;
	}

	// streams in one launch from which the four-streams-per-wave layout is used (0: never); NA_REC_QUAD_MIN, tests: SetRecurrentQuadMinStreams
	static std::atomic<int> gQuadMin{ -1 };
	static std::atomic<long> gQuadLaunches{ 0 };
	int RecurrentQuadMinStreams()
	{
		int v = gQuadMin.load(std::memory_order_relaxed);
		if (v < 0)
		{
			// default: from the first stream that would be a THIRD wave on some SIMD of the one-stream layout (2049 on MI355X) -- measured
			// (tools/runs/r06ah_quadsizes.py, us per 128-sample step, one stream per wave / four):
			//   streams     1024  1536  2048  2560  3072  3584  4096  5120  6144
			//   LSTM 1x16   19.6  29.4  30.1  41.1  40.8  51.2  51.1  61.6  72.0  /  32.1  33.8  34.2  35.6  36.0  37.8  39.5  54.9  57.8
			//   GRU 1x16    19.6  27.3  27.3  37.0  37.1  45.9  45.9  55.0  64.1  /  26.9  28.3  28.3  29.9  29.8  31.2  32.5  42.0  43.5
			v = Tuning::Get().recQuadMin;
			if (v < 0) v = 2 * 4 * CurrentDeviceCUs() + 1;
			gQuadMin.store(v, std::memory_order_relaxed);
		}
		return v;
	}
	int SetRecurrentQuadMinStreams(int streams)
	{
		const int before = RecurrentQuadMinStreams();
		gQuadMin.store(streams < 0 ? 0 : streams, std::memory_order_relaxed);
		return before;
	}
	long RecurrentQuadLaunches() { return gQuadLaunches.load(std::memory_order_relaxed); }

	// grid = all streams of all groups, block = 64 (one wave per stream)
	__device__ __forceinline__ void RecurrentDppRun(const RecurrentGroupArgs& ga, int idx, int noSkew, const float* __restrict__ in, float* __restrict__ out, long inStride,
		long outStride, int n, float* xin, float* hout)
	{
		const int slot = ga.slots ? ga.slots[idx] : ga.slot0 + idx;
		const int row = ga.slots ? ga.rows[idx] : ga.row0 + idx;
		if (ga.m.hidden > 16 && ga.m.cell == LSTM_CELL_GRU) // one-layer GRUs of 17 .. 32 units
		{
			if (ga.m.hidden <= 24) GruDpp32Body<true>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
			else GruDpp32Body<false>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
			return;
		}
		if (ga.m.hidden > 16) // one-layer LSTMs of 17 .. 32 units (RecurrentDppSupported)
		{
			const bool std32 = ga.m.math == LSTM_MATH_STD;
			if (ga.m.hidden <= 24)
			{
				if (std32) LstmDpp32Body<true, true>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
				else LstmDpp32Body<false, true>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
			}
			else
			{
				if (std32) LstmDpp32Body<true, false>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
				else LstmDpp32Body<false, false>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
			}
			return;
		}
		const int layoutH = ga.m.hidden <= 8 ? 8 : 16; // the lane layout the hidden size is padded into
		const int key = ga.m.cell * 100 + layoutH * 4 + ga.m.numLayers + ((ga.m.cell == LSTM_CELL_LSTM && ga.m.math == LSTM_MATH_STD) ? 1000 : 0);
#define NA_REC_CASE(CELL, HH, LL, BODY) \
	case (CELL) * 100 + HH * 4 + LL: BODY<HH, LL>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout); break;
		switch (key)
		{
			NA_REC_CASE(LSTM_CELL_LSTM, 8, 1, LstmDppBody)
			case LSTM_CELL_LSTM * 100 + 8 * 4 + 2:
				if (noSkew) LstmDppBody<8, 2>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
				else LstmDppSkewBody<false>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
				break;
			NA_REC_CASE(LSTM_CELL_LSTM, 16, 1, LstmDppBody)
			NA_REC_CASE(LSTM_CELL_LSTM, 16, 2, LstmDppBody)
			NA_REC_CASE(10 + LSTM_CELL_LSTM, 8, 1, LstmDppBodyStd)
			case (10 + LSTM_CELL_LSTM) * 100 + 8 * 4 + 2:
				if (noSkew) LstmDppBodyStd<8, 2>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
				else LstmDppSkewBody<true>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, xin, hout);
				break;
			NA_REC_CASE(10 + LSTM_CELL_LSTM, 16, 1, LstmDppBodyStd)
			NA_REC_CASE(10 + LSTM_CELL_LSTM, 16, 2, LstmDppBodyStd)
			NA_REC_CASE(LSTM_CELL_GRU, 8, 1, GruDppBody)
			NA_REC_CASE(LSTM_CELL_GRU, 8, 2, GruDppBody)
			NA_REC_CASE(LSTM_CELL_GRU, 16, 1, GruDppBody)
			NA_REC_CASE(LSTM_CELL_GRU, 16, 2, GruDppBody)
		default: break;
		}
#undef NA_REC_CASE
	}

	// PIPE: workgroups of FOUR waves (one per SIMD of the CU).  A group that pipelines its two layers (RecurrentPipeGroup) puts two streams
	// into a workgroup, waves [A0 B0 B1 A1] (A: layer 0, B: layer 1); every other group two streams as well (waves 0 and 3), on its one-wave body.
	// firstBlock counts these workgroups (host: PipeBlocksOf).  Measured (profiles/r06_cfg4_pipeline.txt, us per 128-sample step, two waves
	// per stream / one): LSTM 2x16 x 256 streams 27.7 / 35.1, x 512: 28.0 / 36.1, BASELINE config 4 (512 LSTM 2x16 + 512 GRU) 33.0 / 37.1;
	// x 1024: 41.1 / 38.0 -- with two waves on every SIMD a layer-0 wave no longer stays ahead of its layer-1 wave (which then spins), so
	// such launches keep one wave per stream (UsePipe has the rule and the measurements over the batch size).  Swapping the A / B order between workgroups that share a
	// CU (five patterns tried) changes nothing.
	constexpr int REC_PIPE_WG_WAVES = 4;
	template <bool PIPE>
	__device__ __forceinline__ void RecurrentDppDispatch(const RecurrentGroupArgs& ga, int noSkew, const float* __restrict__ in, float* __restrict__ out, long inStride,
		long outStride, int n, float* xinAll, float* houtAll, int houtWave)
	{
		if constexpr (PIPE)
		{
			const int wave = (int)threadIdx.x >> 6, blk = (int)blockIdx.x - ga.firstBlock;
			if (RecurrentPipeGroup(ga.m))
			{
				const int pair = wave >> 1, idx = blk * 2 + pair;
				if (idx >= ga.numStreams) return;
				const int layer = ((wave + 1) >> 1) & 1; // waves 0 1 2 3 -> layers 0 1 1 0
				const int slot = ga.slots ? ga.slots[idx] : ga.slot0 + idx;
				const int row = ga.slots ? ga.rows[idx] : ga.row0 + idx;
				float* xin = xinAll + pair * REC_XIN_FLOATS;
				float* lds = houtAll + (size_t)pair * 2 * houtWave;
				if (ga.m.math == LSTM_MATH_STD) LstmDppPipeBody<true>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, layer, xin, lds);
				else LstmDppPipeBody<false>(ga.m, ga.state, ga.capacity, slot, row, in, out, inStride, outStride, n, layer, xin, lds);
				return;
			}
			// a one-wave group of such a launch: TWO streams per workgroup as well, on waves 0 and 3 (1 and 2 leave at once).  Four per
			// workgroup put eight waves on the CUs that get a second workgroup and four on the others (config 4: 384 workgroups on 256 CUs);
			// with two, the GRU half of config 4 is one workgroup of two waves on EVERY CU: 32.9 -> 32.4 us (which two waves: no difference)
			if (wave == 1 || wave == 2) return;
			const int idx = blk * 2 + (wave == 3 ? 1 : 0);
			if (idx >= ga.numStreams) return;
			RecurrentDppRun(ga, idx, noSkew, in, out, inStride, outStride, n, xinAll + wave * REC_XIN_FLOATS, houtAll + (size_t)wave * houtWave);
		}
		else RecurrentDppRun(ga, (int)blockIdx.x - ga.firstBlock, noSkew, in, out, inStride, outStride, n, xinAll, houtAll);
	}

	template <bool PIPE>
	__global__ void __launch_bounds__(PIPE ? 256 : 64) RecurrentDppKernel(const RecurrentLaunchArgs args, const float* __restrict__ in, float* __restrict__ out, long inStride,
		long outStride, int n)
	{
		__shared__ __attribute__((aligned(16))) float xin[(PIPE ? REC_PIPE_WG_WAVES : 1) * REC_XIN_FLOATS];
		extern __shared__ __attribute__((aligned(16))) float hout[]; // REC_HOUT_FLOATS; REC_HOUT32_FLOATS when a 32-unit-layout group is in the launch; REC_PIPE_FLOATS with a pipelined one
		int gi = 0;
		for (int i = 1; i < args.numGroups; i++)
			if ((int)blockIdx.x >= args.g[i].firstBlock) gi = i;
		RecurrentDppDispatch<PIPE>(args.g[gi], args.noSkew, in, out, inStride, outStride, n, xin, hout, args.houtWave);
	}

	// The same with the group table in device memory: any number of model groups in one launch (a batch in which every stream plays its
	// own capture; the kernarg segment holds RECURRENT_MAX_GROUPS).  A wave finds its group by binary search over firstBlock and copies
	// the entry out of the constant address space (scalar loads).  See wavenet_spec_impl.h WaveNetSpecTableKernel.
	template <bool PIPE>
	__global__ void __launch_bounds__(PIPE ? 256 : 64) RecurrentDppTableKernel(const RecurrentGroupArgs* __restrict__ table, int numGroups, int noSkew, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, int houtWave)
	{
		__shared__ __attribute__((aligned(16))) float xin[(PIPE ? REC_PIPE_WG_WAVES : 1) * REC_XIN_FLOATS];
		extern __shared__ __attribute__((aligned(16))) float hout[];
		typedef const __attribute__((address_space(4))) RecurrentGroupArgs* TablePtr;
		TablePtr tab = (TablePtr)(size_t)table;
		int lo = 0, hi = numGroups - 1;
		while (lo < hi)
		{
			const int mid = (lo + hi + 1) >> 1;
			if (tab[mid].firstBlock <= (int)blockIdx.x) lo = mid;
			else hi = mid - 1;
		}
		RecurrentGroupArgs ga;
		{
			static_assert(sizeof(RecurrentGroupArgs) % 4 == 0, "dword copy");
			typedef const __attribute__((address_space(4))) unsigned* WordPtr;
			WordPtr src = (WordPtr)(size_t)(table + lo);
			unsigned* dst = reinterpret_cast<unsigned*>(&ga);
#pragma unroll
			for (int i = 0; i < (int)(sizeof(RecurrentGroupArgs) / 4); i++) dst[i] = src[i];
		}
		RecurrentDppDispatch<PIPE>(ga, noSkew, in, out, inStride, outStride, n, xin, hout, houtWave);
	}

	// two waves per stream for the launch?  Some group must pipeline, and the second waves must find SIMDs with room (UsePipe)
	static bool HostPipeGroup(const LstmModelDev& m)
	{
		return m.cell == LSTM_CELL_LSTM && m.numLayers == 2 && m.hidden > 8 && m.hidden <= 16 && m.tailLayers == 0;
	}
	static int PipeBlocksOf(const RecurrentGroup& g) { return (g.numStreams + 1) / 2; } // (two streams per four-wave workgroup, pipelined or not)
	static_assert(2 * REC_HOUT_FLOATS >= REC_PIPE_FLOATS, "a pair of waves' LDS regions hold the pipelined body's two h arrays");
	// Measured (tools/runs/r06ag_pipesizes.py, LSTM 2x16 alone, us per 128-sample step, 1024 SIMDs):
	//   streams        384   512   640   768   896  1024  1152  1280  1536  1792  2048  2304  2560  3072
	//   one wave      35.9  36.1  37.0  37.1  37.9  38.0  59.9  60.0  60.7  61.1  61.5  86.0  86.3  86.8    a step per wave a SIMD holds
	//   two waves     27.4  27.6  40.1  40.2  40.3  40.3  52.1  52.2  52.4  63.0  63.1  75.5  75.7  88.9    half a step per wave
	// -- the pipeline wins whenever its 2 S waves fill an ODD number of waves per SIMD (the one-wave layout then pays a whole step for a
	// round that is at most half full), and loses by ~5 % otherwise.  Launches that also hold other recurrent groups (BASELINE config 4:
	// 512 LSTM 2x16 + 512 GRU = 1536 waves, 33.0 against 37.1 us) pipeline up to one and a half waves per SIMD, as measured there.
	// NA_REC_PIPE_MAX = a plain wave-count threshold for every launch (tuning), NA_REC_NOPIPE = never.
	static bool UsePipe(const RecurrentGroup* groups, int numGroups, int)
	{
		const Tuning& t = Tuning::Get();
		if (t.recNoPipe) return false;
		long piped = 0, others = 0;
		for (int i = 0; i < numGroups; i++) (HostPipeGroup(groups[i].model) ? piped : others) += groups[i].numStreams;
		if (piped == 0) return false;
		const long waves = 2 * piped + others, simds = 4L * CurrentDeviceCUs();
		if (t.recPipeMax > 0) return waves <= (long)t.recPipeMax;
		if (others != 0) return 2 * waves <= 3 * simds;
		return (((waves + simds - 1) / simds) & 1) != 0;
	}

	bool RecurrentDppSupported(const LstmModelDev& m)
	{
		// hidden sizes below a layout (8 or 16 units per gate block) are padded into it: 12 (the reference's static 1x12 / 2x12) runs as 16
		// ... and one-layer LSTMs (the reference's static 1x24) / keras GRUs of 17 .. 32 units on the 32-unit layout
		const bool no32 = Tuning::Get().recNoDpp32;
		if (m.tailLayers != 0) return false; // generic keras stacks run on the runtime-shaped kernels
		if ((m.cell == LSTM_CELL_LSTM || m.cell == LSTM_CELL_GRU) && m.numLayers == 1 && m.hidden > 16 && m.hidden <= 32) return !no32;
		return m.hidden >= 1 && m.hidden <= 16 && (m.numLayers == 1 || m.numLayers == 2) && (m.cell == LSTM_CELL_LSTM || m.cell == LSTM_CELL_GRU);
	}

	hipError_t LaunchRecurrentDppTable(const RecurrentGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, WnLaunchTable& t)
	{
		if (n <= 0 || numGroups <= 0) return hipSuccess;
		if (n > LSTM_MAX_FRAMES) return hipErrorInvalidValue;
		std::vector<RecurrentGroupArgs> fresh((size_t)numGroups);
		int blocks = 0;
		bool any32 = false;
		for (int i = 0; i < numGroups; i++)
		{
			if (!RecurrentDppSupported(groups[i].model) || groups[i].numStreams <= 0) return hipErrorInvalidValue;
			fresh[(size_t)i] = {};
			RecurrentGroupArgs& a = fresh[(size_t)i];
			a.m = groups[i].model;
			a.state = groups[i].state;
			a.slots = groups[i].slots;
			a.rows = groups[i].rows;
			a.slot0 = groups[i].slot0;
			a.row0 = groups[i].row0;
			a.capacity = groups[i].capacity;
			a.numStreams = groups[i].numStreams;
			a.firstBlock = blocks;
			blocks += groups[i].numStreams;
			any32 |= groups[i].model.hidden > 16;
		}
		if (!any32 && UsePipe(groups, numGroups, blocks))
		{
			// four-wave workgroups of two streams each (RecurrentDppDispatch)
			blocks = 0;
			for (int i = 0; i < numGroups; i++)
			{
				fresh[(size_t)i].firstBlock = blocks;
				blocks += PipeBlocksOf(groups[i]);
			}
			const void* devp = nullptr;
			const hipError_t ep = t.Ensure(fresh.data(), fresh.size() * sizeof(RecurrentGroupArgs), stream, &devp);
			if (ep != hipSuccess) return ep;
			if (t.prepareOnly) return hipSuccess;
			hipLaunchKernelGGL(RecurrentDppTableKernel<true>, dim3((unsigned)blocks), dim3(64 * REC_PIPE_WG_WAVES), sizeof(float) * (size_t)REC_PIPE_WG_WAVES * REC_HOUT_FLOATS, stream,
				reinterpret_cast<const RecurrentGroupArgs*>(devp), numGroups, Tuning::Get().recNoSkew ? 1 : 0, in, out, inStride, outStride, n, REC_HOUT_FLOATS);
			return hipGetLastError();
		}
		const void* dev = nullptr;
		const hipError_t ee = t.Ensure(fresh.data(), fresh.size() * sizeof(RecurrentGroupArgs), stream, &dev);
		if (ee != hipSuccess) return ee;
		if (t.prepareOnly) return hipSuccess;
		const size_t lds = sizeof(float) * (size_t)(any32 ? REC_HOUT32_FLOATS : REC_HOUT_FLOATS);
		hipLaunchKernelGGL(RecurrentDppTableKernel<false>, dim3((unsigned)blocks), dim3(64), lds, stream, reinterpret_cast<const RecurrentGroupArgs*>(dev), numGroups,
			Tuning::Get().recNoSkew ? 1 : 0, in, out, inStride, outStride, n, 0);
		return hipGetLastError();
	}

	hipError_t LaunchRecurrentDpp(const RecurrentGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream)
	{
		if (n <= 0 || numGroups <= 0) return hipSuccess;
		if (n > LSTM_MAX_FRAMES || numGroups > RECURRENT_MAX_GROUPS) return hipErrorInvalidValue;
		RecurrentLaunchArgs args = {};
		args.numGroups = numGroups;
		const bool noSkew = Tuning::Get().recNoSkew;
		args.noSkew = noSkew ? 1 : 0;
		int blocks = 0;
		for (int i = 0; i < numGroups; i++)
		{
			if (!RecurrentDppSupported(groups[i].model) || groups[i].numStreams <= 0) return hipErrorInvalidValue;
			RecurrentGroupArgs& a = args.g[i];
			a.m = groups[i].model;
			a.state = groups[i].state;
			a.slots = groups[i].slots;
			a.rows = groups[i].rows;
			a.slot0 = groups[i].slot0;
			a.row0 = groups[i].row0;
			a.capacity = groups[i].capacity;
			a.numStreams = groups[i].numStreams;
			a.firstBlock = blocks;
			blocks += groups[i].numStreams;
		}
		// a batch with more waves than the chip can hold at three per SIMD: four streams per wave (all groups must have the layout)
		const int quadMin = RecurrentQuadMinStreams();
		bool quad = RecurrentQuadMinStreams() > 0 && blocks >= quadMin;
		for (int i = 0; i < numGroups; i++) quad = quad && RecurrentQuadSupported(groups[i].model);
		if (quad)
		{
			blocks = 0;
			for (int i = 0; i < numGroups; i++)
			{
				args.g[i].firstBlock = blocks;
				blocks += (groups[i].numStreams + 3) / 4;
			}
			gQuadLaunches.fetch_add(1, std::memory_order_relaxed);
			hipLaunchKernelGGL(RecurrentQuadKernel, dim3((unsigned)blocks), dim3(64), 0, stream, args, in, out, inStride, outStride, n);
			return hipGetLastError();
		}
		bool any32 = false;
		for (int i = 0; i < numGroups; i++) any32 |= groups[i].model.hidden > 16;
		if (!any32 && UsePipe(groups, numGroups, blocks))
		{
			blocks = 0;
			for (int i = 0; i < numGroups; i++)
			{
				args.g[i].firstBlock = blocks;
				blocks += PipeBlocksOf(groups[i]);
			}
			args.houtWave = REC_HOUT_FLOATS;
			hipLaunchKernelGGL(RecurrentDppKernel<true>, dim3((unsigned)blocks), dim3(64 * REC_PIPE_WG_WAVES), sizeof(float) * (size_t)REC_PIPE_WG_WAVES * REC_HOUT_FLOATS, stream, args, in, out,
				inStride, outStride, n);
			return hipGetLastError();
		}
		const size_t lds = sizeof(float) * (size_t)(any32 ? REC_HOUT32_FLOATS : REC_HOUT_FLOATS);
		hipLaunchKernelGGL(RecurrentDppKernel<false>, dim3((unsigned)blocks), dim3(64), lds, stream, args, in, out, inStride, outStride, n);
		return hipGetLastError();
	}
}
