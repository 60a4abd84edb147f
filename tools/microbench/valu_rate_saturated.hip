// valu_rate_saturated.hip -- issue rate of the VALU instruction classes of the recurrent kernels with the SIMDs FULL (W waves per SIMD, every
// wave four independent chains): time per instruction and SIMD relative to v_fmac_f32.  (lone_wave_issue.hip is the one-wave counterpart.)
// The four-streams-per-wave recurrent layout was first written with DPP row sums; the fit of the two kernels' instruction mixes to
// their run times said a v_fmac_f32_dpp costs ~6 cycles where a plain or packed VALU instruction costs ~4.3 -- this probe measures it.
//   hipcc --offload-arch=gfx950 -O3 -o bin/valu_rate_saturated valu_rate_saturated.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)

#define KERNEL(NAME, ASM)                                                                                              \
	__global__ void __launch_bounds__(256) NAME(float* sink, int iters, float seed)                                       \
	{                                                                                                                      \
		float a = seed + threadIdx.x, b = seed * 0.5f, c = seed * 0.25f, d = seed * 0.125f, w = 0.999f, h = 0.5f + threadIdx.x * 0.001f; \
		double p = seed, q = seed * 0.5, pw = 0.999;                                                                       \
		for (int i = 0; i < iters; i++) asm volatile(REP16(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q) : "v"(w), "v"(h), "v"(pw)); \
		sink[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (float)p + (float)q;                                         \
	}

// 4 instructions per group x 16 groups per iteration
KERNEL(k_fmac, "v_fmac_f32 %0, %6, %7\nv_fmac_f32 %1, %6, %7\nv_fmac_f32 %2, %6, %7\nv_fmac_f32 %3, %6, %7\n")
KERNEL(k_dpp, "v_fmac_f32_dpp %0, %7, %6 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_fmac_f32_dpp %1, %7, %6 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
			  "v_fmac_f32_dpp %2, %7, %6 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_fmac_f32_dpp %3, %7, %6 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(k_pk, "v_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\nv_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\n")
KERNEL(k_mix, "v_fmac_f32 %0, %6, %7\nv_fmac_f32_dpp %1, %7, %6 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_fmac_f32 %2, %6, %7\nv_fmac_f32_dpp %3, %7, %6 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")

typedef void (*Kern)(float*, int, float);

int main()
{
	int cus = 256;
	hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	float* sink;
	hipMalloc(&sink, (size_t)cus * 8 * 256 * sizeof(float));
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 20000; // x 64 instructions per wave
	struct { const char* name; Kern k; } cases[] = { { "v_fmac_f32", k_fmac }, { "v_fmac_f32_dpp row_ror", k_dpp }, { "v_pk_fma_f32", k_pk }, { "v_rcp_f32", k_rcp },
		{ "v_exp_f32", k_exp }, { "v_fmac_f32 / v_fmac_f32_dpp alternating", k_mix } };
	for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2)
	{
		double base = 0.0;
		std::printf("%d wave(s) per SIMD (%d workgroups of 4 waves):\n", wavesPerSimd, cus * wavesPerSimd);
		for (auto& c : cases)
		{
			hipLaunchKernelGGL(c.k, dim3(cus * wavesPerSimd), dim3(256), 0, 0, sink, 200, 1.0f);
			hipDeviceSynchronize();
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(c.k, dim3(cus * wavesPerSimd), dim3(256), 0, 0, sink, iters, 1.0f);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0.0f;
			hipEventElapsedTime(&ms, e0, e1);
			const double nsPerInstr = (double)ms * 1e6 / ((double)iters * 64.0 * wavesPerSimd); // per instruction and SIMD
			if (base == 0.0) base = nsPerInstr;
			std::printf("  %-42s %.3f ns per instruction and SIMD = %.2f x v_fmac_f32 (%.2f cycles at 4 per v_fmac_f32)\n", c.name, nsPerInstr, nsPerInstr / base, 4.0 * nsPerInstr / base);
		}
	}
	return 0;
}
