#include <hip/hip_runtime.h>
__global__ void k(int* out)
{
	int a = threadIdx.x, b = threadIdx.x + 100;
	auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
	auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
	out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main()
{
	int* d; hipMalloc(&d, 256 * 4);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	for (int v = 0; v < 4; v++) { printf("%s:", v == 0 ? "p32.a" : v == 1 ? "p32.b" : v == 2 ? "p16.a" : "p16.b"); for (int i = 0; i < 64; i += 8) printf(" %d", h[v * 64 + i]); printf("\n"); }
	return 0;
}
