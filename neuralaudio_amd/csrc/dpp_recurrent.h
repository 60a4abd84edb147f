// dpp_recurrent.h -- cross-lane building blocks of the LDS-free recurrent kernel (RecurrentDppKernel: LSTM and keras GRU bodies) on gfx950.
//
// Layout they assume: lane = H * gate + unit with H = 8 or 16, so a 16-lane DPP row holds the H units once (H = 16) or twice
// (H = 8), and every lane keeps h[unit] (replicated across the gate rows).
#pragma once

#include <hip/hip_runtime.h>

namespace na
{
	// Row sums: acc += sum_n w[n] * h[lane - n within its 16-lane row]: v_fmac_f32 with a DPP row_ror:n source, one instruction per term.
	// (Written as asm: the compiler keeps a separate v_mov_b32_dpp per term otherwise.  A DPP read of a VGPR needs two wait states
	// after the VALU write that produced it, which the hazard recognizer cannot see inside an asm block: see the variants below.)
#define NA_DPP_TERM(N, OP) "v_fmac_f32_dpp %0, %1, %" #OP " row_ror:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
	// The sum started inside the block: acc = a0 * a1 + a2, then acc += w[0] * h, then the DPP terms.  The two leading
	// instructions are the two wait states a DPP read of `h` needs after the VALU write that produced it, so `h` may come straight
	// from the previous instruction and no s_nop is spent -- a lone wave pays ~5 cycles for EVERY instruction it issues, s_nop and
	// scalar ones included (tools/microbench/lone_wave_issue.hip), so the recurrence is written for instruction count.
	template <int H>
	__device__ __forceinline__ void DppDotFrom(float& acc, float a0, float a1, float a2, const float (&w)[H], float h)
	{
		static_assert(H == 8 || H == 16, "");
		float r;
		if constexpr (H == 8)
			asm volatile("v_fma_f32 %0, %2, %3, %4\nv_fmac_f32 %0, %5, %1\n" NA_DPP_TERM(1, 6) NA_DPP_TERM(2, 7) NA_DPP_TERM(3, 8) NA_DPP_TERM(4, 9) NA_DPP_TERM(5, 10) NA_DPP_TERM(6, 11) NA_DPP_TERM(7, 12)
				: "=&v"(r) : "v"(h), "v"(a0), "v"(a1), "v"(a2), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
		else
			asm volatile("v_fma_f32 %0, %2, %3, %4\nv_fmac_f32 %0, %5, %1\n" NA_DPP_TERM(1, 6) NA_DPP_TERM(2, 7) NA_DPP_TERM(3, 8) NA_DPP_TERM(4, 9) NA_DPP_TERM(5, 10) NA_DPP_TERM(6, 11) NA_DPP_TERM(7, 12)
				NA_DPP_TERM(8, 13) NA_DPP_TERM(9, 14) NA_DPP_TERM(10, 15) NA_DPP_TERM(11, 16) NA_DPP_TERM(12, 17) NA_DPP_TERM(13, 18) NA_DPP_TERM(14, 19) NA_DPP_TERM(15, 20)
				: "=&v"(r) : "v"(h), "v"(a0), "v"(a1), "v"(a2), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]),
				"v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
		acc = r;
	}

	// Two-input row (layer >= 1: input vector hin, own state h): acc = w[0] * hin + b; acc += u0 * h; DPP terms of hin.  The DPP
	// terms of h follow with DppDotTail (an asm block takes at most 30 operands).
	template <int H>
	__device__ __forceinline__ void DppDotFrom2(float& acc, float b, const float (&w)[H], float hin, float u0, float h)
	{
		static_assert(H == 8 || H == 16, "");
		float r;
		if constexpr (H == 8)
			asm volatile("v_fma_f32 %0, %5, %1, %2\nv_fmac_f32 %0, %3, %4\n" NA_DPP_TERM(1, 6) NA_DPP_TERM(2, 7) NA_DPP_TERM(3, 8) NA_DPP_TERM(4, 9) NA_DPP_TERM(5, 10) NA_DPP_TERM(6, 11) NA_DPP_TERM(7, 12)
				: "=&v"(r) : "v"(hin), "v"(b), "v"(u0), "v"(h), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
		else
			asm volatile("v_fma_f32 %0, %5, %1, %2\nv_fmac_f32 %0, %3, %4\n" NA_DPP_TERM(1, 6) NA_DPP_TERM(2, 7) NA_DPP_TERM(3, 8) NA_DPP_TERM(4, 9) NA_DPP_TERM(5, 10) NA_DPP_TERM(6, 11) NA_DPP_TERM(7, 12)
				NA_DPP_TERM(8, 13) NA_DPP_TERM(9, 14) NA_DPP_TERM(10, 15) NA_DPP_TERM(11, 16) NA_DPP_TERM(12, 17) NA_DPP_TERM(13, 18) NA_DPP_TERM(14, 19) NA_DPP_TERM(15, 20)
				: "=&v"(r) : "v"(hin), "v"(b), "v"(u0), "v"(h), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]),
				"v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
		acc = r;
	}

	// acc += sum_{n >= 1} w[n] * h[lane - n]: the DPP terms only, no leading wait states -- for an `h` that was written at least two
	// instructions earlier (the caller's responsibility: e.g. the state of a layer while another row is being summed)
	template <int H>
	__device__ __forceinline__ void DppDotTail(float& acc, const float (&w)[H], float h)
	{
		static_assert(H == 8 || H == 16, "");
		if constexpr (H == 8)
			asm volatile(NA_DPP_TERM(1, 2) NA_DPP_TERM(2, 3) NA_DPP_TERM(3, 4) NA_DPP_TERM(4, 5) NA_DPP_TERM(5, 6) NA_DPP_TERM(6, 7) NA_DPP_TERM(7, 8) : "+v"(acc) : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
		else
			asm volatile(NA_DPP_TERM(1, 2) NA_DPP_TERM(2, 3) NA_DPP_TERM(3, 4) NA_DPP_TERM(4, 5) NA_DPP_TERM(5, 6) NA_DPP_TERM(6, 7) NA_DPP_TERM(7, 8) NA_DPP_TERM(8, 9) NA_DPP_TERM(9, 10) NA_DPP_TERM(10, 11) NA_DPP_TERM(11, 12) NA_DPP_TERM(12, 13) NA_DPP_TERM(13, 14) NA_DPP_TERM(14, 15) NA_DPP_TERM(15, 16) : "+v"(acc) : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
	}
#undef NA_DPP_TERM

	// gfx950 lane swaps, written as asm with both registers in-out: the builtins' second result is mis-folded by this compiler
	// (ROCm 7.2) when both operands derive from one value (tools/microbench/permlane_probe2.hip stores a[0] twice).  s_nops (or
	// useful instructions) cover VALU write -> permlane read (2 wait states); the hazard recognizer does not look inside asm.
	//   v_permlane32_swap a, b: a.lanes[32..63] <-> b.lanes[0..31]        a = [a.lo, b.lo], b = [a.hi, b.hi]
	//   v_permlane16_swap a, b: odd 16-lane rows of a <-> even rows of b    a = rows [a0, b0, a2, b2], b = rows [a1, b1, a3, b3]

	// Every 16-lane row of v replicated into all four rows, in ONE block with the fewest instructions the hazard allows: two wait
	// states between a VALU write (a swap counts) of a register and a lane swap that reads it -- nothing is needed between a swap and a
	// plain VALU read of its result (the LLVM gfx950 rule "VALU write vdst -> v_permlane read" is the only one; copies double as wait
	// states).  r0..r3 = row 0..3 of v everywhere.
	__device__ __forceinline__ void ReplicateRows(float v, float& r0, float& r1, float& r2, float& r3)
	{
		int x = __builtin_bit_cast(int, v), y, x2, y2;
		asm volatile(
			"v_mov_b32 %1, %0\n"
			"s_nop 1\n"
			"v_permlane32_swap_b32 %0, %1\n" // x: rows 0 1 0 1, y: rows 2 3 2 3
			"v_mov_b32 %2, %0\n"
			"v_mov_b32 %3, %1\n"
			"s_nop 0\n"
			"v_permlane16_swap_b32 %0, %2\n" // x: row 0 everywhere, x2: row 1
			"v_permlane16_swap_b32 %1, %3\n" // y: row 2, y2: row 3
			: "+v"(x), "=&v"(y), "=&v"(x2), "=&v"(y2));
		r0 = __builtin_bit_cast(float, x);
		r1 = __builtin_bit_cast(float, x2);
		r2 = __builtin_bit_cast(float, y);
		r3 = __builtin_bit_cast(float, y2);
	}

	// H = 8 LSTM cell state update in one block.  A row holds two gate blocks: after the swap every row of x is [i | f] and every row
	// of y is [g | o].  The lanes of the first block compute the unit: c' = f c + i g with i and g their own and f from the other
	// half of the row (DPP row_ror:8).  Returns c' (defined in lanes 0..7 of every row only) and y (o for the caller).
	__device__ __forceinline__ void LstmCellState8(float gv, float& c, float& go)
	{
		int x = __builtin_bit_cast(int, gv), y;
		float t, cn;
		asm volatile(
			"v_mov_b32 %1, %0\n"
			"s_nop 1\n"
			"v_permlane16_swap_b32 %0, %1\n"
			"s_nop 1\n"
			"v_mov_b32_dpp %2, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
			"v_mul_f32 %3, %0, %1\n"
			"v_fmac_f32 %3, %2, %4\n"
			: "+v"(x), "=&v"(y), "=&v"(t), "=&v"(cn) : "v"(c));
		c = cn;
		go = __builtin_bit_cast(float, y);
	}

	// h = o tanh(c') in the first block of every row (o from the other half of the row), then copied over the second block
	__device__ __forceinline__ float LstmCellOut8(float go, float tanhc)
	{
		float h;
		asm volatile(
			"v_mul_f32_dpp %0, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
			"s_nop 1\n"
			"v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc\n"
			: "=&v"(h) : "v"(go), "v"(tanhc));
		return h;
	}


	// H = 8: a row is [lo half | hi half]; copy one half over the other (DPP row_ror:8 with a bank mask)
	__device__ __forceinline__ int RowLowHalf(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xC, false); }  // lanes 8..15 <- lanes 0..7
	__device__ __forceinline__ int RowHighHalf(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0x3, false); } // lanes 0..7 <- lanes 8..15

	// tanh(x) = 1 - 2 / (e^(2x) + 1) on the hardware exp2 / rcp units: absolute error ~1e-7 (the reference's Eigen tanh is itself a
	// few-ulp rational approximation), saturates correctly (e^(2x) -> inf: 1, -> 0: -1), 6 instructions instead of libm's ~30
	__device__ __forceinline__ float GruTanh(float x)
	{
		const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f); // 2 * log2(e)
		return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
	}

	__device__ __forceinline__ float GruSigmoid(float x) { return (GruTanh(x * 0.5f) + 1.0f) * 0.5f; }

	// StdMath (Activation.h:37-45): std::tanh and 1 / (1 + exp(-x)), on the same exp2 / rcp units (absolute error ~1e-7)
	__device__ __forceinline__ float StdTanh(float x) { return GruTanh(x); }
	__device__ __forceinline__ float StdSigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
}
