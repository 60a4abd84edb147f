// Which VALU / LDS instructions issue in the shadow of v_mfma_f32_16x16x32_f16 on gfx950, (a) from the same wave, (b) from another
// wave of the same SIMD?  (mfma_f16_probe.hip found v_pk_fma_f32 does not: 4 MFMA + 4 pk_fma = 133 cycles vs 68 for the MFMAs alone.)
// One kernel per candidate instruction X:  alone (16 X), 4 MFMA + 8 X interleaved 1:2, 4 MFMA + 16 X interleaved 1:4, and a split
// workgroup (first half of the waves: MFMA only, second half: X only; waves w and w + nw/2 share a SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_valu_mix mfma_f16_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)

template <int WHICH, int MODE>
__global__ void __launch_bounds__(1024) K(float* out, long long* cyc, int iters)
{
	__shared__ __attribute__((aligned(16))) char smem[64 * 1024];
	const int wave = threadIdx.x >> 6;
	const int nw = blockDim.x >> 6;
	f32x4 c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
	const float s = threadIdx.x * 0.001f;
	float q0 = s, q1 = s + 1, q2 = s + 2, q3 = s + 3;
	float2 p0 = { s, s }, p1 = { 1.0001f, 0.9999f }, p2 = { s + 1, s }, p3 = { 0.5f, 0.25f };
	f16x8 a, b;
	for (int e = 0; e < 8; e++) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(0.001f * e); }
	const float x = 1.0001f, y = 0.001f;
	const unsigned lds = (unsigned)(size_t)(smem) + (threadIdx.x & 63) * 16 + (wave & 3) * 4096;
	reinterpret_cast<float*>(smem)[threadIdx.x] = s;
	unsigned hwid;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
	const int simd = (hwid >> 4) & 3;
	const int role = (MODE == 3) ? (wave >= nw / 2 ? 1 : 0) : 0;
	__syncthreads();
	const long long t0 = __builtin_readcyclecounter();
#define XS(op) asm volatile(op : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a), "+v"(b) : "v"(x), "v"(y), "v"(lds) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115")
#define M0 "v_mfma_f32_16x16x32_f16 %8, %12, %13, %8\n"
#define M1 "v_mfma_f32_16x16x32_f16 %9, %12, %13, %9\n"
#define M2 "v_mfma_f32_16x16x32_f16 %10, %12, %13, %10\n"
#define M3 "v_mfma_f32_16x16x32_f16 %11, %12, %13, %11\n"
#define RUN(XA, XB, XC, XD)                                                                                        \
	for (int i = 0; i < iters; i++)                                                                                  \
	{                                                                                                                \
		if (MODE == 0 || (MODE == 3 && role == 1)) XS(REP4(XA XB XC XD) "s_waitcnt lgkmcnt(0)\n");                     \
		if (MODE == 1) XS(M0 XA XB M1 XC XD M2 XA XB M3 XC XD "s_waitcnt lgkmcnt(0)\n");                               \
		if (MODE == 2) XS(M0 XA XB XC XD M1 XA XB XC XD M2 XA XB XC XD M3 XA XB XC XD "s_waitcnt lgkmcnt(0)\n");       \
		if (MODE == 3 && role == 0) XS(M0 M1 M2 M3);                                                                   \
		if (MODE == 4) XS(M0 M1 M2 M3);                                                                                \
	}
	if (WHICH == 0) { RUN("v_fma_f32 %0, %0, %14, %15\n", "v_fma_f32 %1, %1, %14, %15\n", "v_fma_f32 %2, %2, %14, %15\n", "v_fma_f32 %3, %3, %14, %15\n") }
	if (WHICH == 1) { RUN("v_mul_f32 %0, %0, %14\n", "v_mul_f32 %1, %1, %14\n", "v_mul_f32 %2, %2, %14\n", "v_mul_f32 %3, %3, %14\n") }
	if (WHICH == 2) { RUN("v_pk_fma_f32 %4, %4, %5, %5\n", "v_pk_fma_f32 %5, %5, %6, %6\n", "v_pk_fma_f32 %6, %6, %7, %7\n", "v_pk_fma_f32 %7, %7, %4, %4\n") }
	if (WHICH == 3) { RUN("v_pk_mul_f32 %4, %4, %5\n", "v_pk_mul_f32 %5, %5, %6\n", "v_pk_mul_f32 %6, %6, %7\n", "v_pk_mul_f32 %7, %7, %4\n") }
	if (WHICH == 4) { RUN("v_pk_add_f32 %4, %4, %5\n", "v_pk_add_f32 %5, %5, %6\n", "v_pk_add_f32 %6, %6, %7\n", "v_pk_add_f32 %7, %7, %4\n") }
	if (WHICH == 5) { RUN("v_cvt_pk_f16_f32 %0, %14, %15\n", "v_cvt_pk_f16_f32 %1, %14, %15\n", "v_cvt_pk_f16_f32 %2, %14, %15\n", "v_cvt_pk_f16_f32 %3, %14, %15\n") }
	if (WHICH == 6) { RUN("v_rcp_f32 %0, %0\n", "v_rcp_f32 %1, %1\n", "v_rcp_f32 %2, %2\n", "v_rcp_f32 %3, %3\n") }
	if (WHICH == 7) { RUN("v_fma_mix_f32 %0, %0, %14, %15 op_sel_hi:[1,0,0]\n", "v_fma_mix_f32 %1, %1, %14, %15 op_sel_hi:[1,0,0]\n", "v_fma_mix_f32 %2, %2, %14, %15 op_sel_hi:[1,0,0]\n", "v_fma_mix_f32 %3, %3, %14, %15 op_sel_hi:[1,0,0]\n") }
	if (WHICH == 8) { RUN("v_perm_b32 %0, %0, %14, %15\n", "v_perm_b32 %1, %1, %14, %15\n", "v_perm_b32 %2, %2, %14, %15\n", "v_perm_b32 %3, %3, %14, %15\n") }
	if (WHICH == 9) { RUN("v_pk_fma_f16 %0, %0, %14, %15\n", "v_pk_fma_f16 %1, %1, %14, %15\n", "v_pk_fma_f16 %2, %2, %14, %15\n", "v_pk_fma_f16 %3, %3, %14, %15\n") }
	if (WHICH == 10) { RUN("v_mov_b32 %0, %14\n", "v_mov_b32 %1, %14\n", "v_mov_b32 %2, %14\n", "v_mov_b32 %3, %14\n") }
	if (WHICH == 11) { RUN("ds_read_b128 v[100:103], %16\n", "ds_read_b128 v[104:107], %16 offset:1024\n", "ds_read_b128 v[108:111], %16 offset:2048\n", "ds_read_b128 v[112:115], %16 offset:3072\n") }
	if (WHICH == 12) { RUN("ds_write_b128 %16, v[100:103]\n", "ds_write_b128 %16, v[104:107] offset:1024\n", "ds_write_b128 %16, v[108:111] offset:2048\n", "ds_write_b128 %16, v[112:115] offset:3072\n") }
	if (WHICH == 13) { RUN("v_max_f32 %0, %0, %14\n", "v_add_f32 %1, %1, %14\n", "v_max_f32 %2, %2, %14\n", "v_add_f32 %3, %3, %14\n") }
	if (WHICH == 14) { RUN("v_exp_f32 %0, %0\n", "v_exp_f32 %1, %1\n", "v_exp_f32 %2, %2\n", "v_exp_f32 %3, %3\n") }
	const long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * 1024 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + p0.x + p1.y + p2.x + p3.x + q0 + q1 + q2 + q3 + reinterpret_cast<float*>(smem)[(threadIdx.x * 7) & 1023];
	if ((threadIdx.x & 63) == 0) { cyc[wave * 2] = t1 - t0; cyc[wave * 2 + 1] = simd | (role << 8); }
}

template <int WHICH, int MODE>
static double Once(int threads, float* d, long long* dc, long long* h)
{
	const int iters = 2000;
	hipLaunchKernelGGL((K<WHICH, MODE>), dim3(1), dim3(threads), 0, 0, d, dc, 10);
	hipLaunchKernelGGL((K<WHICH, MODE>), dim3(1), dim3(threads), 0, 0, d, dc, iters);
	(void)hipDeviceSynchronize();
	(void)hipMemcpy(h, dc, 32 * sizeof(long long), hipMemcpyDeviceToHost);
	return (double)h[0] / iters;
}

template <int WHICH>
static void Row(const char* name, float* d, long long* dc)
{
	long long h[32];
	const double alone = Once<WHICH, 0>(64, d, dc, h);      // 16 X
	const double mf = Once<WHICH, 4>(64, d, dc, h);         // 4 MFMA
	const double mix2 = Once<WHICH, 1>(64, d, dc, h);       // 4 MFMA + 8 X
	const double mix4 = Once<WHICH, 2>(64, d, dc, h);       // 4 MFMA + 16 X
	Once<WHICH, 3>(512, d, dc, h);                          // 2 waves / SIMD: MFMA wave (4 per iter) beside X wave (16 per iter)
	const double sM = (double)h[0] / 2000, sX = (double)h[8] / 2000;
	const bool sameSimd = (h[1] & 3) == (h[9] & 3);
	Once<WHICH, 2>(512, d, dc, h);                          // 2 waves / SIMD, both 4 MFMA + 16 X
	const double both0 = (double)h[0] / 2000, both1 = (double)h[8] / 2000;
	printf("%-18s 16X alone %6.1f | 4M %5.1f | 4M+8X %6.1f | 4M+16X %6.1f | split%s: M-wave %6.1f X-wave %6.1f | 2 waves of 4M+16X: %6.1f %6.1f\n", name, alone, mf, mix2, mix4,
		sameSimd ? "" : "(!simd)", sM, sX, both0, both1);
}

int main()
{
	float* d; long long* dc;
	(void)hipMalloc(&d, 1024 * sizeof(float));
	(void)hipMalloc(&dc, 32 * sizeof(long long));
	printf("cycles per iteration; one iteration = the listed instruction counts (X = candidate, M = v_mfma_f32_16x16x32_f16)\n");
	Row<0>("v_fma_f32", d, dc);
	Row<1>("v_mul_f32", d, dc);
	Row<13>("v_max/v_add_f32", d, dc);
	Row<2>("v_pk_fma_f32", d, dc);
	Row<3>("v_pk_mul_f32", d, dc);
	Row<4>("v_pk_add_f32", d, dc);
	Row<5>("v_cvt_pk_f16_f32", d, dc);
	Row<6>("v_rcp_f32", d, dc);
	Row<14>("v_exp_f32", d, dc);
	Row<7>("v_fma_mix_f32", d, dc);
	Row<8>("v_perm_b32", d, dc);
	Row<9>("v_pk_fma_f16", d, dc);
	Row<10>("v_mov_b32", d, dc);
	Row<11>("ds_read_b128", d, dc);
	Row<12>("ds_write_b128", d, dc);
	return 0;
}
