// dpp_recurrent.h -- cross-lane building blocks of the LDS-free recurrent kernels (LstmDppKernel, GruDppKernel) on gfx950.
//
// Layout they assume: lane = H * gate + unit with H = 8 or 16, so a 16-lane DPP row holds the H units once (H = 16) or twice
// (H = 8), and every lane keeps h[unit] (replicated across the gate rows).
#pragma once

#include <hip/hip_runtime.h>

namespace na
{
	// acc += sum_n w[n] * h[lane - n within its 16-lane row]: v_fmac_f32 with a DPP row_ror:n source, one instruction per term.
	// (Written as asm: the compiler keeps a separate v_mov_b32_dpp per term otherwise.  The leading s_nop covers the VALU-write ->
	// DPP-read hazard on h, which the hazard recognizer cannot see inside an asm block.)
#define NA_DPP_TERM(N, OP) "v_fmac_f32_dpp %0, %1, %" #OP " row_ror:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
	template <int H>
	__device__ __forceinline__ void DppDot(float& acc, const float (&w)[H], float h)
	{
		static_assert(H == 8 || H == 16, "");
		acc = __builtin_fmaf(w[0], h, acc);
		if constexpr (H == 8)
			asm volatile("s_nop 1\n" NA_DPP_TERM(1, 2) NA_DPP_TERM(2, 3) NA_DPP_TERM(3, 4) NA_DPP_TERM(4, 5) NA_DPP_TERM(5, 6) NA_DPP_TERM(6, 7) NA_DPP_TERM(7, 8) : "+v"(acc) : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
		else
			asm volatile("s_nop 1\n" NA_DPP_TERM(1, 2) NA_DPP_TERM(2, 3) NA_DPP_TERM(3, 4) NA_DPP_TERM(4, 5) NA_DPP_TERM(5, 6) NA_DPP_TERM(6, 7) NA_DPP_TERM(7, 8) NA_DPP_TERM(8, 9) NA_DPP_TERM(9, 10) NA_DPP_TERM(10, 11) NA_DPP_TERM(11, 12) NA_DPP_TERM(12, 13) NA_DPP_TERM(13, 14) NA_DPP_TERM(14, 15) NA_DPP_TERM(15, 16) : "+v"(acc) : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
	}
#undef NA_DPP_TERM

	// gfx950 lane swaps, written as asm with both registers in-out: the builtins' second result is mis-folded by this compiler
	// (ROCm 7.2) when both operands derive from one value (tools/microbench/permlane_probe2.hip stores a[0] twice).  The s_nops
	// cover VALU write -> permlane read (2 wait states); the hazard recognizer does not look inside asm.
	//   LaneSwap32(a, b): a.lanes[32..63] <-> b.lanes[0..31]              a = [a.lo, b.lo], b = [a.hi, b.hi]
	//   LaneSwap16(a, b): odd 16-lane rows of a <-> even rows of b          a = rows [a0, b0, a2, b2], b = rows [a1, b1, a3, b3]
	__device__ __forceinline__ void LaneSwap32(int& a, int& b) { asm volatile("s_nop 1\nv_permlane32_swap_b32 %0, %1\ns_nop 1\n" : "+v"(a), "+v"(b)); }
	__device__ __forceinline__ void LaneSwap16(int& a, int& b) { asm volatile("s_nop 1\nv_permlane16_swap_b32 %0, %1\ns_nop 1\n" : "+v"(a), "+v"(b)); }

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
