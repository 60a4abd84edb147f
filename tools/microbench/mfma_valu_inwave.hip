// Can independent VALU / LDS instructions issue under the shadow of an fp32 MFMA (a) from the same wave, (b) from another wave
// of the same SIMD?  Cycle counts come from s_memtime inside the kernel, so they do not depend on the clock the chip settles at.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_inwave mfma_valu_inwave.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

#define MF16 "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n"
#define MF16B "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n"
#define MF4 "v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n"
#define MF4B "v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n"
#define VA "v_fma_f32 %2, %2, %9, %8\n"
#define VB "v_fma_f32 %3, %3, %9, %8\n"
#define VC "v_fma_f32 %4, %4, %9, %8\n"
#define VD "v_fma_f32 %5, %5, %9, %8\n"
#define PK "v_pk_fma_f32 %6, %6, %7, %7\n"

#define DS "ds_read_b128 v[100:103], %10\n"
#define DSW "s_waitcnt lgkmcnt(0)\n"
#define SA "s_add_u32 s20, s20, 1\n"
#define BODY(name, text)                                                                                                              \
	__device__ __forceinline__ void name(f32x4& c0, f32x4& c1, float& v0, float& v1, float& v2, float& v3, float2& p0, float2& p1, float a, float b, unsigned ldsAddr) \
	{                                                                                                                                  \
		asm volatile(REP8(text) : "+v"(c0), "+v"(c1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(p0), "+v"(p1) : "v"(a), "v"(b), "v"(ldsAddr) : "v100", "v101", "v102", "v103", "s20");                \
	}

// 16x16x4 (8 passes = 32 cycles) + n VALU
BODY(m16_v0, MF16 MF16B)
BODY(m16_v4, MF16 VA VB MF16B VC VD)
BODY(m16_v8, MF16 VA VB VC VD MF16B VA VB VC VD)
BODY(m16_v12, MF16 VA VB VC VD VA VB MF16B VC VD VA VB VC VD)
BODY(m16_v16, MF16 VA VB VC VD VA VB VC VD MF16B VA VB VC VD VA VB VC VD)
BODY(m16_v24, MF16 VA VB VC VD VA VB VC VD VA VB VC VD MF16B VA VB VC VD VA VB VC VD VA VB VC VD)
BODY(v8_only, VA VB VC VD VA VB VC VD)
BODY(pk8_only, PK PK PK PK PK PK PK PK)
// 4x4x1 (2 passes = 8 cycles) + n VALU
BODY(m4_v0, MF4 MF4B MF4 MF4B MF4 MF4B MF4 MF4B)
BODY(m4_v8, MF4 VA MF4B VB MF4 VC MF4B VD MF4 VA MF4B VB MF4 VC MF4B VD)
BODY(m4_v16, MF4 VA VB MF4B VC VD MF4 VA VB MF4B VC VD MF4 VA VB MF4B VC VD MF4 VA VB MF4B VC VD)
BODY(m4_ds8, MF4 DS MF4B DS MF4 DS MF4B DS MF4 DS MF4B DS MF4 DS MF4B DS DSW)
BODY(m4_ds2, MF4 DS MF4B MF4 MF4B MF4 DS MF4B MF4 MF4B DSW)
BODY(m4_sa8, MF4 SA MF4B SA MF4 SA MF4B SA MF4 SA MF4B SA MF4 SA MF4B SA)
BODY(ds8_only, DS DS DS DS DS DS DS DS DSW)
BODY(m4_pk8, MF4 PK MF4B PK MF4 PK MF4B PK MF4 PK MF4B PK MF4 PK MF4B PK)

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters, int split)
{
	const int wave = threadIdx.x >> 6;
	float a = threadIdx.x * 0.001f, b = 1.0001f;
	f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
	float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
	float2 p0 = {a, a}, p1 = {b, b};
	__shared__ float lds[4096];
	lds[threadIdx.x] = a;
	const unsigned ldsAddr = (threadIdx.x & 3) * 16;
	unsigned hwid;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
	const int simd = (hwid >> 4) & 3;
	// split mode: waves on the same SIMD do different things (first wave seen on a SIMD = role 0, others role 1)
	const int role = split ? ((wave >> 2) & 1) : 0;
	__syncthreads();
	const long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++)
	{
		if (MODE == 0) m16_v0(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 1) m16_v4(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 2) m16_v8(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 3) m16_v12(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 4) m16_v16(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 5) m16_v24(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 6) v8_only(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 7) pk8_only(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 8) m4_v0(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 9) m4_v8(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 10) m4_v16(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 11) m4_pk8(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 14) m4_ds8(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 15) m4_ds2(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 16) m4_sa8(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 17) ds8_only(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		if (MODE == 12) // two waves per SIMD: role 0 = 2x MFMA16 per iter, role 1 = 8 VALU per iter
		{
			if (role == 0) m16_v0(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
			else v8_only(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		}
		if (MODE == 13) // role 0 = 8x MFMA4, role 1 = 8 VALU
		{
			if (role == 0) m4_v0(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
			else v8_only(c0, c1, v0, v1, v2, v3, p0, p1, a, b, ldsAddr);
		}
	}
	const long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y + v0 + v1 + v2 + v3 + p0.x + p0.y;
	if ((threadIdx.x & 63) == 0) { cyc[wave * 2] = t1 - t0; cyc[wave * 2 + 1] = simd | (role << 8); }
}

template <int MODE>
void run(const char* name, int threads, int split, float* d, long long* dc)
{
	const int iters = 4000;
	hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, dc, 10, split);
	hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, dc, iters, split);
	hipDeviceSynchronize();
	long long h[16];
	hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
	printf("%-44s waves=%d :", name, threads / 64);
	for (int w = 0; w < threads / 64; w++) printf(" [simd%lld r%lld] %.1f", h[2 * w + 1] & 3, h[2 * w + 1] >> 8, (double)h[2 * w] / iters / 8);
	printf("  cycles/iter\n");
}

int main()
{
	float* d; long long* dc;
	hipMalloc(&d, 512 * sizeof(float));
	hipMalloc(&dc, 16 * sizeof(long long));
	run<0>("2x mfma16x16x4 (expect 64)", 64, 0, d, dc);
	run<1>("2x mfma16 + 4 valu", 64, 0, d, dc);
	run<2>("2x mfma16 + 8 valu", 64, 0, d, dc);
	run<3>("2x mfma16 + 12 valu", 64, 0, d, dc);
	run<4>("2x mfma16 + 16 valu", 64, 0, d, dc);
	run<5>("2x mfma16 + 24 valu", 64, 0, d, dc);
	run<6>("8 valu fma only", 64, 0, d, dc);
	run<7>("8 pk_fma only", 64, 0, d, dc);
	run<8>("8x mfma4x4x1 (expect 64)", 64, 0, d, dc);
	run<9>("8x mfma4 + 8 valu", 64, 0, d, dc);
	run<10>("8x mfma4 + 16 valu", 64, 0, d, dc);
	run<11>("8x mfma4 + 8 pk_fma", 64, 0, d, dc);
	run<14>("8x mfma4 + 8 ds_read_b128 (broadcast)", 64, 0, d, dc);
	run<15>("8x mfma4 + 2 ds_read_b128", 64, 0, d, dc);
	run<16>("8x mfma4 + 8 salu", 64, 0, d, dc);
	run<17>("8 ds_read_b128 only", 64, 0, d, dc);
	run<14>("8x mfma4 + 8 ds_read, 8 waves", 512, 0, d, dc);
	run<0>("2x mfma16, 8 waves (2/SIMD)", 512, 0, d, dc);
	run<6>("8 valu, 8 waves (2/SIMD)", 512, 0, d, dc);
	run<12>("split: mfma16 wave + valu wave per SIMD", 512, 1, d, dc);
	run<8>("8x mfma4, 8 waves", 512, 0, d, dc);
	run<13>("split: mfma4 wave + valu wave per SIMD", 512, 1, d, dc);
	return 0;
}
