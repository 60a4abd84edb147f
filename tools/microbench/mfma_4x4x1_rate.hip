// Issue rate of v_mfma_f32_4x4x1_16b_f32 vs v_mfma_f32_16x16x4_f32 vs v_pk_fma_f32 (one wave per SIMD, N independent accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
	float a = threadIdx.x * 0.001f, b = 1.0001f;
	f32x4 c[NACC];
	for (int i = 0; i < NACC; i++) c[i] = f32x4{0, 0, 0, 0};
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int u = 0; u < 8; u++)
#pragma unroll
			for (int i = 0; i < NACC; i++)
			{
				if (MODE == 0) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
				else if (MODE == 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
				else { f32x2 t = __builtin_elementwise_fma(f32x2{a, b}, f32x2{c[i].x, c[i].y}, f32x2{c[i].z, c[i].w}); c[i].x = t.x; c[i].y = t.y; }
			}
	}
	float s = 0; for (int i = 0; i < NACC; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, float* d)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 4000;
	hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(256), 0, 0, d, 10);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(256), 0, 0, d, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	printf("%-34s NACC=%d : %.2f cycles/instr/SIMD (2.4 GHz)\n", name, NACC, ms * 1e-3 * 2.4e9 / (iters * 8.0 * NACC));
}

int main()
{
	float* d; hipMalloc(&d, 256 * 256 * 4);
	run<0, 1>("v_mfma_f32_4x4x1_16b_f32", d); run<0, 2>("v_mfma_f32_4x4x1_16b_f32", d); run<0, 4>("v_mfma_f32_4x4x1_16b_f32", d); run<0, 8>("v_mfma_f32_4x4x1_16b_f32", d);
	run<1, 1>("v_mfma_f32_16x16x4_f32", d); run<1, 4>("v_mfma_f32_16x16x4_f32", d);
	run<2, 4>("v_pk_fma_f32", d); run<2, 8>("v_pk_fma_f32", d);
	return 0;
}
