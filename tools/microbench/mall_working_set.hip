// mall_working_set.hip -- does the 256 MB Infinity Cache (memory-side, all HBM traffic passes through it) hold a working set that one
// launch after the other walks cyclically?  The WaveNet kernels read and write their ring state once per launch: 1024 Standard streams
// x 315 KB = 322 MB of ADDRESSES today (rings of roundup16(2 d) + 128 frames), 201 MB with rings of exactly 2 d frames.
// Per size: in-place read-modify-write of the whole set (16 B per lane, coalesced), back-to-back launches; then read-only and write-only.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mall_working_set mall_working_set.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE> // 0 read-modify-write, 1 read only, 2 write only
__global__ void __launch_bounds__(256) Walk(u32x4* __restrict__ p, size_t quads, unsigned* sink)
{
	// four independent 16-byte accesses per lane and trip (4 KB in flight per wave): the walk is bound by bandwidth, not by the latency
	// of one load per wave
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	u32x4 acc = { 0, 0, 0, 0 };
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < quads; i += 4 * stride)
	{
		if (MODE == 0)
		{
			u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
			a += 1u; b += 1u; c += 1u; d += 1u;
			p[i] = a; p[i + stride] = b; p[i + 2 * stride] = c; p[i + 3 * stride] = d;
		}
		else if (MODE == 1) acc += p[i] + p[i + stride] + p[i + 2 * stride] + p[i + 3 * stride];
		else
		{
			const u32x4 v = { (unsigned)i, 1u, 2u, 3u };
			p[i] = v; p[i + stride] = v; p[i + 2 * stride] = v; p[i + 3 * stride] = v;
		}
	}
	if (MODE == 1 && (acc.x + acc.y + acc.z + acc.w) == 0x12345u) sink[0] = 1;
}

template <typename F>
static double TimeIt(F&& launch, int iters)
{
	for (int i = 0; i < 20; i++) launch();
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
	const size_t maxBytes = (size_t)2048 << 20;
	u32x4* buf; unsigned* sink;
	hipMalloc(&buf, maxBytes); hipMalloc(&sink, 4);
	hipMemset(buf, 0, maxBytes);
	const int sizesMB[] = { 32, 64, 128, 160, 192, 208, 224, 240, 256, 288, 320, 384, 512, 1024, 2048 };
	printf("%8s %12s %12s %12s   (TB/s of bytes moved: rmw counts read + write)\n", "set MB", "rmw", "read", "write");
	for (int mb : sizesMB)
	{
		const size_t quads = ((size_t)mb << 20) / 16;
		const int grid = 256 * 8;
		const int iters = mb <= 256 ? 200 : 60;
		const double rmw = TimeIt([&] { hipLaunchKernelGGL(Walk<0>, dim3(grid), dim3(256), 0, s, buf, quads, sink); }, iters);
		const double rd = TimeIt([&] { hipLaunchKernelGGL(Walk<1>, dim3(grid), dim3(256), 0, s, buf, quads, sink); }, iters);
		const double wr = TimeIt([&] { hipLaunchKernelGGL(Walk<2>, dim3(grid), dim3(256), 0, s, buf, quads, sink); }, iters);
		const double bytes = (double)mb * 1048576.0;
		printf("%8d %9.2f us %9.2f us %9.2f us   %6.2f %6.2f %6.2f\n", mb, rmw, rd, wr, 2 * bytes / rmw * 1e-6, bytes / rd * 1e-6, bytes / wr * 1e-6);
	}
	return 0;
}
