// Probe: buffer_load_dwordx4 ... lds on gfx950 (global -> LDS without VGPRs).  Where does lane l's 16 bytes land, and does the
// builtin need anything besides a wave-uniform LDS pointer?
// build: hipcc --offload-arch=gfx950 -O3 -o bin/lds_dma_probe lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* src, float* out, unsigned bytes)
{
	__shared__ f32x4 buf[128];
	const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)bytes, 0x00020000);
	for (int i = threadIdx.x; i < 128; i += 64) buf[i] = f32x4{ -1, -1, -1, -1 };
	__syncthreads();
	// lane l fetches float4 number (63 - l) (a permutation, to see that the LDS slot follows the LANE, not the address); lanes >= 60 out of bounds
	const int voff = (threadIdx.x < 60) ? (63 - (int)threadIdx.x) * 16 : (int)0x80000000;
	__builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)&buf[32], 16, voff, 0, 0, 0);
	__builtin_amdgcn_s_waitcnt(0); // vmcnt(0) lgkmcnt(0) expcnt(0) on gfx9 encoding
	__syncthreads();
	for (int i = threadIdx.x; i < 128; i += 64) { out[i * 4] = buf[i].x; out[i * 4 + 1] = buf[i].y; out[i * 4 + 2] = buf[i].z; out[i * 4 + 3] = buf[i].w; }
}

int main()
{
	float h[512], *d, *o;
	for (int i = 0; i < 512; i++) h[i] = (float)i;
	hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
	hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, (unsigned)(64 * 16));
	hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
	printf("buf[31] = %.0f (untouched -1)\n", h[31 * 4]);
	for (int l : {0, 1, 2, 30, 59, 60, 63}) printf("buf[32 + %2d] = %.0f %.0f %.0f %.0f   (lane %d asked for float4 %d)\n", l, h[(32 + l) * 4], h[(32 + l) * 4 + 1], h[(32 + l) * 4 + 2], h[(32 + l) * 4 + 3], l, 63 - l);
	printf("buf[96] = %.0f (untouched -1)\n", h[96 * 4]);
	return 0;
}
