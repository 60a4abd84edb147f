// Probe: A-matrix broadcast (CBSZ / ABID) of v_mfma_f32_4x4x1_16b_f32 on gfx950.
// A = lane id, B = 1.0: without broadcast lane l gets D[r] = 4*(l/4) + r; with cbsz=4, abid=k every lane must get D[r] = 4*k + r.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/mfma_cbsz_probe mfma_cbsz_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void k(float* out)
{
	const float a = (float)threadIdx.x, b = 1.0f + 0.001f * threadIdx.x;
	f32x4 c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
	for (int r = 0; r < 4; r++) out[threadIdx.x * 4 + r] = c[r];
}

template <int ABID>
int check(float* d)
{
	hipLaunchKernelGGL(k<ABID>, dim3(1), dim3(64), 0, 0, d);
	float h[256];
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	int bad = 0;
	for (int l = 0; l < 64; l++)
		for (int r = 0; r < 4; r++)
		{
			const float want = (float)(4 * ABID + r) * (1.0f + 0.001f * l);
			if (fabsf(h[l * 4 + r] - want) > 1e-4f * (1 + fabsf(want))) bad++;
		}
	printf("abid=%2d: lane0 D = %.3f %.3f %.3f %.3f | lane 37 D = %.3f %.3f %.3f %.3f | mismatches %d\n", ABID, h[0], h[1], h[2], h[3], h[148], h[149], h[150], h[151], bad);
	return bad;
}

int main()
{
	float* d;
	hipMalloc(&d, 256 * sizeof(float));
	int bad = check<0>(d) + check<1>(d) + check<5>(d) + check<15>(d);
	printf(bad ? "FAIL\n" : "OK: cbsz=4 broadcasts block ABID's A operand to all 16 blocks\n");
	return bad != 0;
}
