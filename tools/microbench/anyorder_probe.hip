// anyorder_probe.hip -- does hipExtAnyOrderLaunch let consecutive kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is
// "not supported on AMD GFX9xx boards".)  A kernel of one workgroup that spins ~20 us, launched 16 times back to back: ~320 us if the
// launches serialise, ~20 us + launch costs if they overlap.  Also the back-to-back gap of ordinary launches of a 20 us kernel.
//   hipcc --offload-arch=gfx950 -O3 -o bin/anyorder_probe anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void Spin(long long ticks, int* sink)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
	if (ticks < 0) sink[0] = 1;
}

int main()
{
	hipStream_t s;
	hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	int* sink; hipMalloc(&sink, 4);
	const long long ticks = 2000; // 20 us at 100 MHz
	for (int mode = 0; mode < 2; mode++)
	{
		for (int rep = 0; rep < 3; rep++)
		{
			hipStreamSynchronize(s);
			const auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < 16; i++)
			{
				if (mode == 0) hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, ticks, sink);
				else hipExtLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
			}
			hipStreamSynchronize(s);
			const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
			printf("%s: 16 x 20 us kernels on one stream: %.1f us (%.2f us per kernel)\n", mode == 0 ? "ordinary launches" : "hipExtAnyOrderLaunch", us, us / 16);
		}
	}
	return 0;
}
