// lone_wave_issue.hip -- what one instruction costs a wave that has its SIMD to itself (the recurrent kernels at 1024 streams):
// shader-clock cycles per instruction for dependent / independent chains of the instruction classes on the LSTM / GRU recurrence.
//   hipcc --offload-arch=gfx950 -O3 -o bin/lone_wave_issue lone_wave_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

// each case: 64 instructions (or 64 groups) between two s_memtime reads, 16 repetitions, minimum reported
#define CASE(ID, GROUPSIZE, ASM, ...)                                                                                         \
	{                                                                                                                          \
		long long best = 1ll << 60;                                                                                            \
		for (int r = 0; r < 16; r++)                                                                                           \
		{                                                                                                                      \
			long long t0, t1;                                                                                                  \
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");          \
			asm volatile(REP64(ASM) : __VA_ARGS__);                                                                            \
			asm volatile("s_nop 7\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");                                \
			if (t1 - t0 < best) best = t1 - t0;                                                                                \
		}                                                                                                                      \
		if (threadIdx.x == 0) res[ID] = (float)best / (64.0f * GROUPSIZE);                                                     \
	}

__global__ void k(float* res, float* sink, float seed)
{
	float a = seed + threadIdx.x, b = seed * 0.5f, c = seed * 0.25f, d = seed * 0.125f, w = 0.999f, h = 0.5f + threadIdx.x * 0.001f;
	int ia = threadIdx.x, ib = threadIdx.x * 3, sb = 0;
	__shared__ float lds[256];
	lds[threadIdx.x] = a;
	__syncthreads();
	float* lp = &lds[threadIdx.x];
	unsigned laddr = (unsigned)(size_t)lp;

	CASE(0, 1, "v_fmac_f32 %0, %1, %2\n", "+v"(a) : "v"(w), "v"(h))                                       // dependent fmac
	CASE(1, 4, "v_fmac_f32 %0, %4, %5\nv_fmac_f32 %1, %4, %5\nv_fmac_f32 %2, %4, %5\nv_fmac_f32 %3, %4, %5\n", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(h)) // 4 independent chains
	CASE(2, 1, "v_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "+v"(a) : "v"(h), "v"(w)) // dependent dpp fmac (h constant)
	CASE(3, 2, "v_fmac_f32_dpp %0, %2, %3 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_fmac_f32_dpp %1, %2, %3 row_ror:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "+v"(a), "+v"(b) : "v"(h), "v"(w))
	CASE(4, 1, "v_rcp_f32 %0, %0\n", "+v"(a))                                                              // dependent rcp
	CASE(5, 2, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\n", "+v"(a), "+v"(b))
	CASE(6, 1, "v_exp_f32 %0, %0\n", "+v"(b))
	CASE(7, 2, "v_rcp_f32 %0, %0\nv_fma_f32 %0, %0, %1, 1.0\n", "+v"(a) : "v"(w))                          // rcp + dependent VALU
	CASE(8, 1, "s_nop 1\nv_permlane32_swap_b32 %0, %1\ns_nop 1\n", "+v"(ia), "+v"(ib))
	CASE(9, 1, "s_nop 1\nv_permlane16_swap_b32 %0, %1\ns_nop 1\n", "+v"(ia), "+v"(ib))
	CASE(10, 1, "v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n", "+v"(ia))
	CASE(11, 1, "v_mov_b32 %0, %1\n", "=v"(ia) : "v"(ib))
	CASE(12, 1, "ds_write_b32 %0, %1\n", : "v"(laddr), "v"(a) : "memory")                                  // LDS write issue
	CASE(13, 1, "ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\n", "=v"(a) : "v"(laddr) : "memory")             // LDS read round trip
	CASE(14, 1, "s_nop 0\n", )
	CASE(15, 2, "v_mul_f32 %0, %0, %1\nv_add_f32 %0, %0, %1\n", "+v"(a) : "v"(w))
	CASE(16, 1, "v_cndmask_b32 %0, %0, %1, vcc\n", "+v"(a) : "v"(w))
	CASE(17, 2, "v_fmac_f32 %0, %1, %2\ns_nop 0\n", "+v"(a) : "v"(w), "v"(h))                             // does a 1-cycle nop fit in the shadow?
	CASE(18, 3, "v_fmac_f32 %0, %2, %3\ns_add_u32 %1, %1, 1\ns_cmp_lt_i32 %1, 0\n", "+v"(a), "+s"(sb) : "v"(w), "v"(h) : "scc") // scalar ops between VALU
	CASE(19, 1, "v_readlane_b32 %0, %1, 3\n", "=s"(sb) : "v"(ia))
	CASE(20, 1, "v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n", "+v"(ia) : "v"(ib))
	CASE(21, 1, "v_fmaak_f32 %0, %0, %1, 0x401c7bf8\n", "+v"(a) : "v"(w))
	CASE(22, 1, "v_fma_f32 %0, |%0|, %1, %1\n", "+v"(a) : "v"(w))
	CASE(23, 1, "v_pk_fma_f32 %0, %0, %1, %1\n", "+v"(*(double*)&lds[0]) : "v"(*(double*)&lds[2]))
	// a taken branch per group (forward jumps)
	{
		long long best = 1ll << 60;
		for (int r = 0; r < 16; r++)
		{
			long long t0, t1;
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
			asm volatile(REP64("v_fmac_f32 %0, %1, %2\ns_branch 1f\nv_fmac_f32 %0, %1, %2\n1:\n") : "+v"(a) : "v"(w), "v"(h));
			asm volatile("s_nop 7\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
			if (t1 - t0 < best) best = t1 - t0;
		}
		if (threadIdx.x == 0) res[24] = (float)best / 64.0f;
	}
	sink[threadIdx.x] = a + b + c + d + (float)ia + (float)ib + (float)sb + lds[(threadIdx.x + 1) & 63];
}

int main()
{
	float *res, *sink;
	hipMalloc(&res, 64 * 4);
	hipMalloc(&sink, 256 * 4);
	hipMemset(res, 0, 64 * 4);
	for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, res, sink, 1.0f);
	float h[64];
	hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost);
	const char* names[] = { "v_fmac dependent", "v_fmac 4 chains (per instr)", "v_fmac_dpp dependent", "v_fmac_dpp 2 chains (per instr)", "v_rcp dependent",
		"v_rcp 2 chains (per instr)", "v_exp dependent", "v_rcp+v_fma dependent (per instr)", "permlane32_swap + 2 s_nop 1 (group)", "permlane16_swap + 2 s_nop 1 (group)",
		"v_mov_dpp dependent", "v_mov independent", "ds_write issue", "ds_read round trip", "s_nop 0", "v_mul+v_add dependent (per instr)", "v_cndmask dependent",
		"v_fmac + s_nop 0 (per instr)", "v_fmac + 2 scalar (per instr)", "v_readlane", "v_mov_dpp row_bcast", "v_fmaak dependent", "v_fma |x| dependent", "v_pk_fma_f32 dependent",
		"v_fmac + taken s_branch (group)" };
	for (int i = 0; i < 25; i++) printf("%-44s %6.2f cycles\n", names[i], h[i]);
	return 0;
}
