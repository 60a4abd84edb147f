// ring_pattern_walk.hip -- the headline kernel's HBM traffic without its arithmetic: is 173 MB per 1024-stream step in 1 KB wave granules,
// scattered over 1024 x 243 KB of stream state, worth 36 us by itself?  Per stream and launch: 101 KB read + 65 KB written (the measured
// FETCH / WRITE sizes of profiles/r04_p2_pmc_summary.txt), every access one wave-wide 16 B / lane instruction = 1 KB contiguous, at
// granule positions spread over the stream's state (a different rotation every launch, like the ring cursors).  Grid 512 x 512 threads =
// the kernel's launch shape (two streams per workgroup, four waves per stream); DEPTH loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -o bin/ring_pattern_walk ring_pattern_walk.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int STATE_KB = 243, READ_KB = 101, WRITE_KB = 65;

template <int DEPTH>
__global__ void __launch_bounds__(512) Traffic(u32x4* __restrict__ state, int streams, int rot, unsigned* sink)
{
	const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, sub = threadIdx.x >> 8;
	const int s = blockIdx.x * 2 + sub;
	if (s >= streams) return;
	u32x4* base = state + (size_t)s * (STATE_KB * 64); // 64 quads per KB
	u32x4 acc = { 0, 0, 0, 0 };
	// granule g of the launch sits at KB (g * 97 + rot) mod STATE_KB: 97 is coprime to 243
	for (int g0 = wave; g0 < READ_KB; g0 += 4 * DEPTH)
	{
		u32x4 v[DEPTH];
#pragma unroll
		for (int k = 0; k < DEPTH; k++)
		{
			const int g = g0 + 4 * k;
			v[k] = g < READ_KB ? base[(size_t)((g * 97 + rot) % STATE_KB) * 64 + lane] : u32x4{ 0, 0, 0, 0 };
		}
#pragma unroll
		for (int k = 0; k < DEPTH; k++) acc += v[k];
	}
	for (int g = wave; g < WRITE_KB; g += 4) base[(size_t)((g * 89 + rot + 7) % STATE_KB) * 64 + lane] = acc + (unsigned)g;
	if (acc.x == 0x12345678u) sink[0] = 1;
}

template <int DEPTH>
static void Run(u32x4* state, unsigned* sink, int streams, hipStream_t st)
{
	int rot = 0;
	auto launch = [&] { hipLaunchKernelGGL(Traffic<DEPTH>, dim3((streams + 1) / 2), dim3(512), 0, st, state, streams, rot, sink); rot = (rot + 13) % STATE_KB; };
	for (int i = 0; i < 50; i++) launch();
	hipStreamSynchronize(st);
	const auto t0 = std::chrono::steady_clock::now();
	const int iters = 500;
	for (int i = 0; i < iters; i++) launch();
	hipStreamSynchronize(st);
	const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters * 1e6;
	const double mb = (double)streams * (READ_KB + WRITE_KB) * 1024.0 / 1e6;
	printf("%4d streams, %d loads in flight per wave: %7.2f us per launch, %6.1f MB -> %5.2f TB/s\n", streams, DEPTH, us, mb, mb / us);
}

int main()
{
	hipStream_t st;
	hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	const int maxStreams = 1024;
	u32x4* state; unsigned* sink;
	hipMalloc(&state, (size_t)maxStreams * STATE_KB * 1024); hipMalloc(&sink, 4);
	hipMemset(state, 0, (size_t)maxStreams * STATE_KB * 1024);
	for (int streams : { 1024, 512 })
	{
		Run<1>(state, sink, streams, st);
		Run<2>(state, sink, streams, st);
		Run<4>(state, sink, streams, st);
		Run<8>(state, sink, streams, st);
	}
	return 0;
}
