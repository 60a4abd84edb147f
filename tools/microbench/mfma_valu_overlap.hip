// Does v_mfma_f32_16x16x4_f32 overlap with f32 VALU work issued from OTHER waves of the same SIMD?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run on the GPU box.
// Each workgroup = 8 waves (2 per SIMD).  mode 0: all waves MFMA; 1: all waves VALU; 2: even waves MFMA, odd waves VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(float* out, int iters, int mode)
{
	const int wave = threadIdx.x >> 6;
	const bool doMfma = (mode == 0) || (mode == 2 && (wave & 4) == 0);
	float a = threadIdx.x * 0.001f, b = 1.0001f;
	f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
	if (doMfma)
	{
		for (int i = 0; i < iters; i++)
		{
			c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
			c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
			c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
		}
	}
	else
	{
		for (int i = 0; i < iters; i++)
		{
			// 32 independent-ish v_fma per iteration (4 MFMAs = 128 cycles = 32 VALU issue slots of 4 cycles)
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a); v3 = __builtin_fmaf(v3, b, a);
				v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a); v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
			}
		}
	}
	out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

int main()
{
	float* d;
	hipMalloc(&d, 256 * 512 * sizeof(float));
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 20000;
	for (int mode = 0; mode < 3; mode++)
	{
		hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 100, mode);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const char* names[] = {"all waves MFMA (2 waves/SIMD)", "all waves VALU (2 waves/SIMD)", "1 MFMA wave + 1 VALU wave per SIMD"};
		printf("mode %d %-40s %.3f ms  (%.1f cycles/iter/SIMD-pair at 2.4 GHz)\n", mode, names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
	}
	return 0;
}
