// v_mfma_f32_16x16x32_f16 on gfx950: (A) operand / result layout, (B) accuracy of the 3-product f16 split of an f32 mat-mul
// (W*X ~= Wh*Xh + Wh*Xl + Wl*Xh with h = f16(v), l = f16(v - h)), including operands whose low parts are f16 subnormals,
// (C) issue cost alone, next to VALU work of the same wave, and next to a VALU-only wave on the same SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_probe mfma_f16_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- (A) + (B)
// A: [16][32] row-major f32, B: [32][16] row-major f32 (k major), D: [16][16].  mode 0: operands rounded to f16 (layout check);
// mode 1: three-product split
__global__ void MatKernel(const float* A, const float* B, float* D, int mode)
{
	const int l = threadIdx.x, i = l & 15, q = l >> 4;
	f16x8 ah, al, bh, bl;
	for (int e = 0; e < 8; e++)
	{
		const float a = A[i * 32 + 8 * q + e];
		const float b = B[(8 * q + e) * 16 + i];
		ah[e] = (_Float16)a;
		al[e] = (_Float16)(a - (float)ah[e]);
		bh[e] = (_Float16)b;
		bl[e] = (_Float16)(b - (float)bh[e]);
	}
	f32x4 acc = { 0, 0, 0, 0 };
	acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
	if (mode == 1)
	{
		acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
	}
	for (int r = 0; r < 4; r++) D[(4 * q + r) * 16 + i] = acc[r]; // row = 4*(lane>>4)+r, col = lane&15
}

static void Accuracy(const char* name, float scaleA, float scaleB)
{
	std::vector<float> A(16 * 32), B(32 * 16), D(256);
	srand(7);
	for (auto& v : A) v = scaleA * ((rand() % 20001) / 10000.0f - 1.0f);
	for (auto& v : B) v = scaleB * ((rand() % 20001) / 10000.0f - 1.0f);
	float *dA, *dB, *dD;
	hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 256 * 4);
	hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
	hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
	for (int mode = 0; mode < 2; mode++)
	{
		hipLaunchKernelGGL(MatKernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode);
		hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
		double maxRel = 0, maxRef = 0, maxErr32 = 0;
		for (int i = 0; i < 16; i++)
			for (int j = 0; j < 16; j++)
			{
				double ref = 0, mag = 0; float f = 0;
				for (int k = 0; k < 32; k++) { ref += (double)A[i * 32 + k] * B[k * 16 + j]; mag += fabs((double)A[i * 32 + k] * B[k * 16 + j]); f = fmaf(A[i * 32 + k], B[k * 16 + j], f); }
				maxRel = fmax(maxRel, fabs(D[i * 16 + j] - ref) / mag);
				maxErr32 = fmax(maxErr32, fabs(f - ref) / mag);
				maxRef = fmax(maxRef, fabs(ref));
			}
		printf("%-34s mode %d (%s): max |err| / sum|a*b| = %.3g   (f32 fma chain: %.3g, max |ref| %.3g)\n", name, mode, mode ? "3-product split" : "f16 operands", maxRel, maxErr32, maxRef);
	}
	hipFree(dA); hipFree(dB); hipFree(dD);
}

// ---------------------------------------------------------------- (C)
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define MA "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n"
#define MB "v_mfma_f32_16x16x32_f16 %1, %6, %7, %1\n"
#define MC "v_mfma_f32_16x16x32_f16 %2, %6, %7, %2\n"
#define MD "v_mfma_f32_16x16x32_f16 %3, %6, %7, %3\n"
#define PK "v_pk_fma_f32 %4, %4, %5, %5\n"
#define PJ "v_pk_fma_f32 %8, %8, %5, %5\n"
#define CV "v_cvt_pk_f16_f32 %9, %10, %11\n"
#define BODY(name, text)                                                                                                          \
	__device__ __forceinline__ void name(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, float2& p0, float2& p1, float2& p2, f16x8 a, f16x8 b, unsigned& cv, float x, float y) \
	{                                                                                                                              \
		asm volatile(REP8(text) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(p0), "+v"(p1) : "v"(a), "v"(b), "v"(p2), "v"(cv), "v"(x), "v"(y)); \
	}
// p2 is read-modify-write in PJ: declare it as in/out via a second macro family to keep the constraint list simple
#undef BODY
#define BODY(name, text)                                                                                                          \
	__device__ __forceinline__ void name(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, float2& p0, float2& p1, float2& p2, f16x8 a, f16x8 b, unsigned& cv, float x, float y) \
	{                                                                                                                              \
		asm volatile(REP8(text) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(p0), "+v"(p1), "+v"(a), "+v"(b), "+v"(p2), "+v"(cv) : "v"(x), "v"(y)); \
	}
BODY(m4, MA MB MC MD)                                   // 4 MFMAs on 4 accumulators
BODY(m1dep, MA MA MA MA)                                // dependent chain on one accumulator
BODY(m4_v4, MA PK MB PJ MC PK MD PJ)                    // 1 VALU per MFMA
BODY(m4_v8, MA PK PJ MB PK PJ MC PK PJ MD PK PJ)        // 2 per MFMA
BODY(m4_v12, MA PK PJ PK MB PJ PK PJ MC PK PJ PK MD PJ PK PJ) // 3 per MFMA
BODY(m4_v16, MA PK PJ PK PJ MB PK PJ PK PJ MC PK PJ PK PJ MD PK PJ PK PJ)
BODY(m4_v24, MA PK PJ PK PJ PK PJ MB PK PJ PK PJ PK PJ MC PK PJ PK PJ PK PJ MD PK PJ PK PJ PK PJ)
BODY(v8, PK PJ PK PJ PK PJ PK PJ)
BODY(v16, PK PJ PK PJ PK PJ PK PJ PK PJ PK PJ PK PJ PK PJ)
BODY(cv8, CV CV CV CV CV CV CV CV)

template <int MODE>
__global__ void __launch_bounds__(1024) TimeKernel(float* out, long long* cyc, int iters, int split)
{
	const int wave = threadIdx.x >> 6;
	f32x4 c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
	const float s = threadIdx.x * 0.001f;
	float2 p0 = { s, s }, p1 = { 1.0001f, 0.9999f }, p2 = { s + 1, s };
	f16x8 a, b;
	for (int e = 0; e < 8; e++) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(0.001f * e); }
	unsigned cv = 0;
	unsigned hwid;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
	const int simd = (hwid >> 4) & 3;
	const int nw = blockDim.x >> 6;
	const int role = split ? (wave >= nw / 2 ? 1 : 0) : 0; // waves w and w + nw/2 share a SIMD (checked through HW_ID in the printout)
	__syncthreads();
	const long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++)
	{
		if (MODE == 0) m4(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 1) m1dep(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 2) m4_v4(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 3) m4_v8(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 4) m4_v12(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 5) m4_v16(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 6) m4_v24(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 7) v8(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 8) cv8(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		if (MODE == 9) // role 0: 4 MFMAs per rep, role 1: 16 VALU per rep
		{
			if (role == 0) m4(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
			else v16(c0, c1, c2, c3, p0, p1, p2, a, b, cv, s, s);
		}
	}
	const long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * 1024 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + p0.x + p0.y + p2.x + (float)cv;
	if ((threadIdx.x & 63) == 0) { cyc[wave * 2] = t1 - t0; cyc[wave * 2 + 1] = simd | (role << 8); }
}

template <int MODE>
static void Run(const char* name, int threads, int split, float* d, long long* dc)
{
	const int iters = 2000;
	hipLaunchKernelGGL(TimeKernel<MODE>, dim3(1), dim3(threads), 0, 0, d, dc, 10, split);
	hipLaunchKernelGGL(TimeKernel<MODE>, dim3(1), dim3(threads), 0, 0, d, dc, iters, split);
	hipDeviceSynchronize();
	long long h[32];
	hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
	printf("%-46s waves=%2d :", name, threads / 64);
	const int nw = threads / 64;
	for (int w = 0; w < nw; w += (nw > 8 ? 4 : (nw > 4 ? (split ? 4 : 8) : 1))) printf(" [simd%lld r%lld] %.1f", h[2 * w + 1] & 3, h[2 * w + 1] >> 8, (double)h[2 * w] / iters / 8);
	printf("  cycles per rep\n");
}

int main()
{
	Accuracy("operands ~ U(-1,1)", 1.0f, 1.0f);
	Accuracy("A ~ 1, B ~ 1e-3 (lo parts subnormal)", 1.0f, 1e-3f);
	Accuracy("A ~ 1e-2, B ~ 1e-4", 1e-2f, 1e-4f);
	Accuracy("A ~ 30, B ~ 100", 30.0f, 100.0f);
	float* d; long long* dc;
	hipMalloc(&d, 1024 * sizeof(float));
	hipMalloc(&dc, 32 * sizeof(long long));
	Run<0>("4 mfma16x16x32f16, 4 accumulators", 64, 0, d, dc);
	Run<1>("4 mfma dependent (one accumulator)", 64, 0, d, dc);
	Run<2>("4 mfma + 4 pk_fma", 64, 0, d, dc);
	Run<3>("4 mfma + 8 pk_fma", 64, 0, d, dc);
	Run<4>("4 mfma + 12 pk_fma", 64, 0, d, dc);
	Run<5>("4 mfma + 16 pk_fma", 64, 0, d, dc);
	Run<6>("4 mfma + 24 pk_fma", 64, 0, d, dc);
	Run<7>("8 pk_fma only", 64, 0, d, dc);
	Run<8>("8 cvt_pk_f16_f32 only", 64, 0, d, dc);
	Run<0>("4 mfma, 8 waves (2/SIMD)", 512, 0, d, dc);
	Run<0>("4 mfma, 16 waves (4/SIMD)", 1024, 0, d, dc);
	Run<7>("8 pk_fma, 8 waves (2/SIMD)", 512, 0, d, dc);
	Run<7>("8 pk_fma, 16 waves (4/SIMD)", 1024, 0, d, dc);
	Run<5>("4 mfma + 16 pk_fma, 8 waves (2/SIMD)", 512, 0, d, dc);
	Run<5>("4 mfma + 16 pk_fma, 16 waves (4/SIMD)", 1024, 0, d, dc);
	Run<6>("4 mfma + 24 pk_fma, 16 waves (4/SIMD)", 1024, 0, d, dc);
	Run<9>("split: mfma wave(s) + valu wave(s), 2/SIMD", 512, 1, d, dc);
	Run<9>("split: mfma wave(s) + valu wave(s), 4/SIMD", 1024, 1, d, dc);
	return 0;
}
