// mfma_lo16_probe.hip -- is v_mfma_f32_16x16x16_f16 on the first 64 bits of the operands of v_mfma_f32_16x16x32_f16 the same product when the
// A operand's upper four halfs (per lane) are zero?  (The "lo" product of the f16 split: A = [Wl | 0 0 0 0].)  Prints the largest difference.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_lo16_probe mfma_lo16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void Probe(const f16x8* a, const f16x8* b, f32x4* d32, f32x4* d16, f32x4* chain)
{
	const int l = threadIdx.x;
	const f16x8 A = a[l], B = b[l];
	const f32x4 z = { 0.f, 0.f, 0.f, 0.f };
	d32[l] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, z, 0, 0, 0);
	const f16x4 A4 = { A[0], A[1], A[2], A[3] }, B4 = { B[0], B[1], B[2], B[3] };
	d16[l] = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, B4, z, 0, 0, 0);
	// dependent chains across the two instructions (the accumulator of one is the next one's srcC)
	f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, z, 0, 0, 0);          // K=32 -> K=16
	c = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, B4, c, 0, 0, 0);
	chain[64 + l] = c;
	f32x4 e = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, B4, z, 0, 0, 0);          // K=16 -> K=32
	e = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, e, 0, 0, 0);
	chain[128 + l] = e;
	f32x4 g = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, B4, z, 0, 0, 0);          // K=16 -> (wait) -> K=32
	asm volatile("s_nop 15\n\ts_nop 15" : "+v"(g));
	g = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, g, 0, 0, 0);
	chain[192 + l] = g;
	f32x4 h = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, B4, z, 0, 0, 0);          // K=16 -> K=16
	h = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, B4, h, 0, 0, 0);
	chain[256 + l] = h;
	c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
	c = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, B4, c, 0, 0, 0);
	chain[l] = c;
}

int main()
{
	f16x8 ha[64], hb[64];
	srand(5);
	for (int l = 0; l < 64; l++)
		for (int t = 0; t < 8; t++)
		{
			ha[l][t] = t < 4 ? (_Float16)((rand() % 2001 - 1000) / 1000.0f * 0.001f) : (_Float16)0.0f;
			hb[l][t] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
		}
	f16x8 *da, *db; f32x4 *d32, *d16, *dch;
	hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&d32, 64 * 16); hipMalloc(&d16, 64 * 16); hipMalloc(&dch, 5 * 64 * 16);
	hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(Probe, dim3(1), dim3(64), 0, 0, da, db, d32, d16, dch);
	f32x4 r32[64], r16[64];
	hipMemcpy(r32, d32, sizeof(r32), hipMemcpyDeviceToHost); hipMemcpy(r16, d16, sizeof(r16), hipMemcpyDeviceToHost);
	double worst = 0.0, mag = 0.0;
	int bad = 0;
	for (int l = 0; l < 64; l++)
		for (int t = 0; t < 4; t++)
		{
			const double e = fabs((double)r32[l][t] - (double)r16[l][t]);
			if (!(e == e)) bad++;
			worst = e > worst ? e : worst;
			mag = fabs((double)r32[l][t]) > mag ? fabs((double)r32[l][t]) : mag;
		}
	f32x4 rch[5 * 64];
	hipMemcpy(rch, dch, sizeof(rch), hipMemcpyDeviceToHost);
	const char* names[5] = { "K=32 -> 16 -> 32 -> 16 (4 x)", "K=32 -> K=16 (2 x)", "K=16 -> K=32 (2 x)", "K=16 -> s_nop -> K=32 (2 x)", "K=16 -> K=16 (2 x)" };
	const double mult[5] = { 4, 2, 2, 2, 2 };
	for (int v = 0; v < 5; v++)
	{
		double worstc = 0.0;
		for (int l = 0; l < 64; l++)
			for (int t = 0; t < 4; t++) { const double e = fabs((double)rch[v * 64 + l][t] - mult[v] * (double)r32[l][t]); worstc = e > worstc ? e : worstc; }
		printf("dependent chain %-30s: largest difference from the expected multiple %.3e\n", names[v], worstc);
	}
	printf("largest |K=32 - K=16| = %.3e (largest |value| %.3e), NaNs %d; lane 0: %g %g | %g %g\n", worst, mag, bad, r32[0][0], r32[0][1], r16[0][0], r16[0][1]);
	return 0;
}
