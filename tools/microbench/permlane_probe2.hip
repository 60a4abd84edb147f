#include <hip/hip_runtime.h>
__global__ void k(float* out, float scale)
{
	float gv = threadIdx.x * scale + 1.0f; // VALU-written right before the swaps
	const int x = __builtin_bit_cast(int, gv);
	const auto lohi = __builtin_amdgcn_permlane32_swap(x, x, false, false);
	const auto a = __builtin_amdgcn_permlane16_swap(lohi[0], lohi[0], false, false);
	const auto b = __builtin_amdgcn_permlane16_swap(lohi[1], lohi[1], false, false);
	out[threadIdx.x] = __builtin_bit_cast(float, a[0]);
	out[64 + threadIdx.x] = __builtin_bit_cast(float, a[1]);
	out[128 + threadIdx.x] = __builtin_bit_cast(float, b[0]);
	out[192 + threadIdx.x] = __builtin_bit_cast(float, b[1]);
}
int main()
{
	float* d; hipMalloc(&d, 256 * 4);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.0f);
	float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	int bad = 0;
	for (int g = 0; g < 4; g++) { printf("g%d:", g); for (int i = 0; i < 64; i += 5) printf(" %.0f", h[g * 64 + i]); printf("\n"); for (int i = 0; i < 64; i++) if (h[g * 64 + i] != (float)(16 * g + (i & 15)) + 1.0f) bad++; }
	printf("bad=%d\n", bad);
	return 0;
}
