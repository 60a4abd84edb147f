// launch_floor.hip -- per-launch cost of back-to-back kernels in one stream (1024 workgroups x 64 threads, like the recurrent kernel):
// an empty kernel, one that does a dependent global load chain (two round trips), and one with a kernarg struct of ~1.5 KB.
//   hipcc --offload-arch=gfx950 -O3 -o bin/launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Big { int v[384]; };

__global__ void Empty(float* p) { if (p == nullptr && threadIdx.x == 999) p[0] = 1.0f; }
__global__ void Chain(const int* idx, const float* src, float* dst)
{
	const int i = idx[blockIdx.x];                 // round trip 1
	const float v = src[i * 64 + threadIdx.x];      // round trip 2
	dst[blockIdx.x * 64 + threadIdx.x] = v + 1.0f;
}
__global__ void BigArgs(Big b, float* p) { if (b.v[blockIdx.x & 255] == 12345 && threadIdx.x == 999) p[0] = 1.0f; }

template <typename F>
static double TimeIt(F&& launch, int iters)
{
	for (int i = 0; i < 200; i++) launch();
	hipDeviceSynchronize();
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < iters; i++) launch();
	hipDeviceSynchronize();
	return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters * 1e6;
}

int main()
{
	hipStream_t s;
	hipStreamCreate(&s);
	int* idx; float *src, *dst;
	hipMalloc(&idx, 1024 * 4); hipMalloc(&src, 1024 * 64 * 4); hipMalloc(&dst, 1024 * 64 * 4);
	hipMemset(idx, 0, 1024 * 4); hipMemset(src, 0, 1024 * 64 * 4);
	Big b = {};
	printf("empty kernel           %.2f us per launch\n", TimeIt([&] { hipLaunchKernelGGL(Empty, dim3(1024), dim3(64), 0, s, dst); }, 5000));
	printf("two dependent loads    %.2f us per launch\n", TimeIt([&] { hipLaunchKernelGGL(Chain, dim3(1024), dim3(64), 0, s, idx, src, dst); }, 5000));
	printf("1.5 KB kernarg         %.2f us per launch\n", TimeIt([&] { hipLaunchKernelGGL(BigArgs, dim3(1024), dim3(64), 0, s, b, dst); }, 5000));
	return 0;
}
