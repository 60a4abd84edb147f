// What does a buffer_store_dwordx4 / buffer_load_dwordx4 cost the issuing wave next to MFMA work?  s_memtime cycle counts, 2 waves per SIMD.
// Per iteration: 256 v_mfma_f32_4x4x1 (~2200 cycles, one 16-channel WaveNet layer) + 4 VMEM instructions of the given kind.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/vmem_issue_cost vmem_issue_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* buf, long long* cyc, int iters, unsigned bytes)
{
	const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
	const int wave = threadIdx.x >> 6;
	float a = threadIdx.x * 0.001f, b = 1.0001f;
	f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	f32x4 ld = {0, 0, 0, 0};
	// coalesced like the WaveNet ring stores: each instruction writes the wave's 64 lanes x 16 B = 1 KB contiguous; only every 5th wave
	// touches memory, so the chip-wide rate (~1 TB/s) is that of the real kernel and the count shows the ISSUE cost, not a bandwidth wall
	const int waveGlobal = blockIdx.x * 8 + wave;
	const bool doMem = (waveGlobal % 5) == 0;
	const int voffIn = (waveGlobal / 5) * (16 * 4096) + (threadIdx.x & 63) * 16;
	const int voff = (MODE == 2 || !doMem) ? (int)0x80000000 : voffIn;
	__syncthreads();
	const long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int u = 0; u < 64; u++)
		{
			c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
			c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
			c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
		}
		const int o = voff + ((MODE == 2 || !doMem) ? 0 : (i & 15) * 4096);
		if (MODE == 1 || MODE == 2)
		{
			__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, c0), r, o, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, c1), r, o + 1024, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, c2), r, o + 2048, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, c3), r, o + 3072, 0, 0);
		}
		if (MODE == 3)
		{
			ld += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0));
			ld += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o + 1024, 0, 0));
			ld += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o + 2048, 0, 0));
			ld += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o + 3072, 0, 0));
		}
		if (MODE == 4) // one dword store per lane instead of four dwordx4
		{
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c0.x), r, o, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c1.x), r, o + 1024, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c2.x), r, o + 2048, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c3.x), r, o + 3072, 0, 0);
		}
	}
	const long long t1 = __builtin_readcyclecounter();
	if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0; // block 0: waves 0 and 5 touch memory
	if (c0.x + c1.y + c2.z + c3.w + ld.x == 12345.678f) buf[0] = 1.0f;
}

template <int MODE>
void run(const char* name, float* d, long long* dc, unsigned bytes)
{
	const int iters = 500;
	hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, dc, 10, bytes);
	hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, dc, iters, bytes);
	hipDeviceSynchronize();
	long long h[8];
	hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
	printf("%-52s memory waves %.1f / %.1f, others %.1f cycles/iter (2 waves per SIMD, all 256 CUs busy)\n", name, (double)h[0] / iters, (double)h[5] / iters,
		((double)h[1] + h[2] + h[3] + h[4] + h[6] + h[7]) / 6 / iters);
}

int main()
{
	const unsigned bytes = 32u << 20; // 410 memory waves x 64 KB
	float* d; long long* dc;
	hipMalloc(&d, bytes);
	hipMalloc(&dc, 8 * sizeof(long long));
	run<0>("256 mfma4x4x1 only", d, dc, bytes);
	run<1>("256 mfma + 4 buffer_store_dwordx4 (in bounds)", d, dc, bytes);
	run<2>("256 mfma + 4 buffer_store_dwordx4 (all lanes OOB)", d, dc, bytes);
	run<3>("256 mfma + 4 buffer_load_dwordx4", d, dc, bytes);
	run<4>("256 mfma + 4 buffer_store_dword", d, dc, bytes);
	return 0;
}
