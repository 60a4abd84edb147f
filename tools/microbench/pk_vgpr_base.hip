// Follow-up of pk_beside_mfma.hip (profiles/r06_quad_race.txt): does a wave's packed-f32 arithmetic depend on WHERE its VGPRs were allocated?
// The fault showed only beside the one-stream-per-workgroup flavour of the split kernel -- 130 VGPRs = an allocation of 136, not a multiple
// of 16 -- and not beside the 128-VGPR flavour.  Here "filler" waves with an allocation of exactly F registers (F = 128, 136, 144, 152)
// sit on every SIMD first (they spin on the clock), then checker waves (packed FMA chains against scalar ones) are allocated behind them.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_vgpr_base pk_vgpr_base.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TOP>
__global__ void __launch_bounds__(256) Filler(float* out, unsigned long long ticks)
{
	float v = threadIdx.x;
	// touch the highest register of the wanted allocation so that the kernel's VGPR count is TOP + 1
	if (TOP == 127) asm volatile("v_mov_b32 v127, %0" ::"v"(v) : "v127");
	if (TOP == 135) asm volatile("v_mov_b32 v135, %0" ::"v"(v) : "v135");
	if (TOP == 143) asm volatile("v_mov_b32 v143, %0" ::"v"(v) : "v143");
	if (TOP == 151) asm volatile("v_mov_b32 v151, %0" ::"v"(v) : "v151");
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
	{
		v = __builtin_fmaf(v, 0.999f, 0.001f);
		__builtin_amdgcn_s_sleep(2);
	}
	out[blockIdx.x * 256 + threadIdx.x] = v;
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) Checker(unsigned* hist, int iters, float seed)
{
	const int lane = threadIdx.x;
	f2 p[24];
	float s0[24], s1[24];
	for (int k = 0; k < 24; k++)
	{
		p[k] = f2{ seed + 0.01f * k, seed - 0.02f * k };
		s0[k] = p[k].x;
		s1[k] = p[k].y;
	}
	unsigned bad = 0;
	for (int i = 0; i < iters; i++)
	{
		const float a = 0.999f - 1e-4f * (float)(i & 15), b = 1e-3f * (float)((i & 7) - 3);
		const f2 A = f2{ a, -a }, B = f2{ b, 0.5f * b };
#pragma unroll
		for (int k = 0; k < 24; k++)
		{
			p[k] = __builtin_elementwise_fma(p[k], A, B);
			s0[k] = __builtin_fmaf(s0[k], a, b);
			s1[k] = __builtin_fmaf(s1[k], -a, 0.5f * b);
		}
		if ((i & 31) == 31)
		{
#pragma unroll
			for (int k = 0; k < 24; k++)
			{
				// (compared as floats through opaque copies: this compiler lowers `bit_cast<unsigned>(p[k].y) != ...` to a compare of the pair's
				// LOW register -- v_cmp_ne_u32 s[2:3], v48, v98 where v49 was meant; seen in the first build of this probe)
				float px = p[k].x, py = p[k].y;
				asm volatile("" : "+v"(px), "+v"(py));
				if (px != s0[k] || py != s1[k])
				{
					bad++;
					p[k] = f2{ s0[k], s1[k] };
				}
			}
		}
	}
	if (bad) atomicAdd(&hist[lane], bad);
}

template <int TOP>
static void Run(const char* name, int cus, unsigned* hist, float* sink, hipStream_t sa, hipStream_t sb, int rounds)
{
	CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
	for (int r = 0; r < rounds; r++)
	{
		if (TOP > 0) hipLaunchKernelGGL(Filler<TOP>, dim3(cus), dim3(256), 0, sb, sink, 2000000ull); // 20 ms, one wave per SIMD
		for (int q = 0; q < 4; q++) hipLaunchKernelGGL(Checker, dim3(cus * 8), dim3(64), 0, sa, hist, 60000, 0.3f + 0.01f * q);
		CHECK(hipStreamSynchronize(sa));
		CHECK(hipStreamSynchronize(sb));
	}
	std::vector<unsigned> h(64);
	CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
	unsigned long q[4] = { 0, 0, 0, 0 };
	for (int l = 0; l < 64; l++) q[l / 16] += h[l];
	printf("%-34s packed vs scalar mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n", name, q[0], q[1], q[2], q[3]);
}

int main(int argc, char** argv)
{
	const int rounds = argc > 1 ? atoi(argv[1]) : 10;
	int cus = 0;
	CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
	unsigned* hist;
	float* sink;
	CHECK(hipMalloc(&hist, 64 * sizeof(unsigned)));
	CHECK(hipMalloc(&sink, (size_t)cus * 256 * sizeof(float)));
	hipStream_t sa, sb;
	CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
	CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
	Run<0>("checker alone", cus, hist, sink, sa, sb, rounds);
	Run<127>("behind fillers of 128 VGPRs", cus, hist, sink, sa, sb, rounds);
	Run<135>("behind fillers of 136 VGPRs", cus, hist, sink, sa, sb, rounds);
	Run<143>("behind fillers of 144 VGPRs", cus, hist, sink, sa, sb, rounds);
	Run<151>("behind fillers of 152 VGPRs", cus, hist, sink, sa, sb, rounds);
	return 0;
}
