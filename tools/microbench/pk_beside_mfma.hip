// Are packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) of one wave safe while waves of ANOTHER kernel run MFMAs
// on the same SIMD?  Found in round 6: the four-streams-per-wave LSTM kernel (RecurrentQuadKernel: ~390 packed instructions per sample
// loop) was occasionally wrong in lanes 48..63 -- only while the f16-split WaveNet chain (MFMA-heavy) ran beside it, never alone, never
// with its pairs evaluated as two scalar FMAs (tools/runs/r06n_quadrace.py).  This probe takes the kernels out of the picture:
//   checker: every lane runs the same chain of packed FMAs and, beside it, the same chain as scalar FMAs; the two must agree bit for bit
//            (fma is correctly rounded either way).  Mismatches are counted per lane.
//   burner : waves that issue v_mfma_f32_16x16x32_f16 back to back (mode 1) or plain VALU FMAs (mode 2), one workgroup of 4 waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_beside_mfma pk_beside_mfma.hip ; run on the GPU box:  ./pk_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) Checker(unsigned* hist, unsigned* firstBad, int iters, float seed)
{
	const int lane = threadIdx.x;
	// 16 independent accumulator pairs (like the kernel's gate sums), coefficients that keep the values bounded
	f2 p[16];
	float s0[16], s1[16];
	for (int k = 0; k < 16; k++)
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
		for (int k = 0; k < 16; k++)
		{
			p[k] = __builtin_elementwise_fma(p[k], A, B); // v_pk_fma_f32
			p[k] = p[k] * f2{ 0.75f, 0.875f } + f2{ 0.1f, -0.1f }; // v_pk_mul_f32 / v_pk_add_f32 (or another pk_fma)
			s0[k] = __builtin_fmaf(s0[k], a, b);
			s1[k] = __builtin_fmaf(s1[k], -a, 0.5f * b);
			s0[k] = __builtin_fmaf(s0[k], 0.75f, 0.1f);
			s1[k] = __builtin_fmaf(s1[k], 0.875f, -0.1f);
		}
		if ((i & 63) == 63)
		{
#pragma unroll
			for (int k = 0; k < 16; k++)
			{
				// (contraction of mul + add into fma is the compiler's choice on BOTH sides: compare with a tolerance of a few ulp, a
				// corrupted lane is off by far more)
				const float dx = fabsf(p[k].x - s0[k]), dy = fabsf(p[k].y - s1[k]);
				if (dx > 1e-5f * (1.0f + fabsf(s0[k])) || dy > 1e-5f * (1.0f + fabsf(s1[k])) || !(dx == dx) || !(dy == dy))
				{
					bad++;
					p[k] = f2{ s0[k], s1[k] }; // resynchronise
				}
			}
		}
	}
	if (bad)
	{
		atomicAdd(&hist[lane], bad);
		atomicMin(firstBad, (unsigned)blockIdx.x);
	}
}

// The pattern of the kernel's gate activation: FOUR transcendentals back to back (quarter rate: lanes 48..63 of the last one come out
// last), a few independent packed instructions, then packed instructions that read the transcendentals' results as register PAIRS.
// Beside it the same values from one transcendental at a time with scalar consumers.  nops: wait states forced between the
// transcendentals and the packed consumers (0: what the compiler emits).
template <int NOPS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) TransChecker(unsigned* hist, int iters, float seed)
{
	const int lane = threadIdx.x;
	f2 y0 = f2{ seed, seed + 0.25f }, y1 = f2{ seed - 0.5f, seed + 0.125f };
	unsigned bad = 0;
	for (int i = 0; i < iters; i++)
	{
		__builtin_amdgcn_sched_barrier(0);
		const f2 d0 = y0 * y0 + f2{ 2.445f, 2.445f }, d1 = y1 * y1 + f2{ 2.445f, 2.445f };
		f2 r0 = f2{ __builtin_amdgcn_rcpf(d0.x), __builtin_amdgcn_rcpf(d0.y) };
		f2 r1 = f2{ __builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y) };
		const f2 p0 = __builtin_elementwise_fma(y0, f2{ 0.82f, 0.82f }, f2{ 0.89f, 0.89f }), p1 = __builtin_elementwise_fma(y1, f2{ 0.82f, 0.82f }, f2{ 0.89f, 0.89f });
		const f2 q0 = y0 * p0, q1 = y1 * p1;
		if (NOPS == 1) asm volatile("s_nop 7\ns_nop 7" : "+v"(r0), "+v"(r1));
		f2 g0 = __builtin_elementwise_fma(q0, r0, f2{ 0.5f, 0.5f }), g1 = __builtin_elementwise_fma(q1, r1, f2{ 0.5f, 0.0f });
		__builtin_amdgcn_sched_barrier(0); // (the reference below must not be scheduled in between: it would be the wait states)
		asm volatile("" : "+v"(g0), "+v"(g1));
		// reference: each reciprocal on its own, consumed by scalar instructions behind a scheduling fence
		float e[4] = { d0.x, d0.y, d1.x, d1.y }, qq[4] = { q0.x, q0.y, q1.x, q1.y }, cc[4] = { 0.5f, 0.5f, 0.5f, 0.0f }, ref[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			float r = __builtin_amdgcn_rcpf(e[k]);
			asm volatile("s_nop 7\ns_nop 7" : "+v"(r));
			ref[k] = __builtin_fmaf(qq[k], r, cc[k]);
		}
		const float got[4] = { g0.x, g0.y, g1.x, g1.y };
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (__builtin_bit_cast(unsigned, got[k]) != __builtin_bit_cast(unsigned, ref[k])) bad++;
		// next values: keep them moving and bounded
		y0 = f2{ ref[0] - 0.3f + 1e-3f * (float)(i & 31), ref[1] * 0.7f - 0.2f };
		y1 = f2{ ref[2] * 1.3f - 0.6f, ref[3] + 0.4f - 2e-3f * (float)(i & 15) };
	}
	if (bad) atomicAdd(&hist[lane], bad);
}

__global__ void __launch_bounds__(256) Burner(float* out, int iters, int mode)
{
	f32x4 c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
	f16x8 a, b;
	for (int k = 0; k < 8; k++)
	{
		a[k] = (_Float16)(0.001f * (float)(threadIdx.x + k));
		b[k] = (_Float16)(0.002f * (float)(k + 1));
	}
	float v0 = threadIdx.x * 0.001f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
	for (int i = 0; i < iters; i++)
	{
		if (mode == 1)
		{
			c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
			c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
			c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
		}
		else if (mode == 3)
		{
			v0 = __builtin_amdgcn_rcpf(v0 + 1.5f);
			v1 = __builtin_amdgcn_rcpf(v1 + 1.5f);
			v2 = __builtin_amdgcn_exp2f(v2 * 0.01f);
			v3 = __builtin_amdgcn_rcpf(v3 + 1.5f);
		}
		else
		{
			v0 = __builtin_fmaf(v0, 0.999f, 0.001f);
			v1 = __builtin_fmaf(v1, 0.999f, 0.001f);
			v2 = __builtin_fmaf(v2, 0.999f, 0.001f);
			v3 = __builtin_fmaf(v3, 0.999f, 0.001f);
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + v0 + v1 + v2 + v3;
}

int main(int argc, char** argv)
{
	const int rounds = argc > 1 ? atoi(argv[1]) : 20;
	int cus = 0;
	CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
	unsigned *hist, *firstBad;
	float* sink;
	CHECK(hipMalloc(&hist, 64 * sizeof(unsigned)));
	CHECK(hipMalloc(&firstBad, sizeof(unsigned)));
	CHECK(hipMalloc(&sink, (size_t)cus * 8 * 256 * sizeof(float)));
	hipStream_t sa, sb;
	CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
	CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
	const char* names[3] = { "checker alone", "checker beside MFMA waves", "checker beside VALU waves" };
	for (int mode = 0; mode < 3; mode++)
	{
		CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
		CHECK(hipMemset(firstBad, 0xff, sizeof(unsigned)));
		for (int r = 0; r < rounds; r++)
		{
			if (mode > 0) hipLaunchKernelGGL(Burner, dim3(cus * 2), dim3(256), 0, sb, sink, 400000, mode);
			for (int q = 0; q < 8; q++) hipLaunchKernelGGL(Checker, dim3(cus * 6), dim3(64), 0, sa, hist, firstBad, 20000, 0.3f + 0.01f * q);
			CHECK(hipStreamSynchronize(sa));
			CHECK(hipStreamSynchronize(sb));
		}
		std::vector<unsigned> h(64);
		CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
		unsigned long q[4] = { 0, 0, 0, 0 };
		for (int l = 0; l < 64; l++) q[l / 16] += h[l];
		printf("%-28s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n", names[mode], q[0], q[1], q[2], q[3]);
	}
	// the transcendental -> packed-pair pattern, beside nothing / MFMA waves / VALU waves / transcendental-heavy waves; then with wait states
	const char* bn[4] = { "nothing", "MFMA waves", "VALU waves", "transcendental waves" };
	for (int nops = 0; nops < 2; nops++)
		for (int mode = 0; mode < 4; mode++)
		{
			CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
			for (int r = 0; r < rounds; r++)
			{
				if (mode > 0) hipLaunchKernelGGL(Burner, dim3(cus * 4), dim3(256), 0, sb, sink, mode == 1 ? 400000 : 1500000, mode);
				for (int q = 0; q < 8; q++)
				{
					if (nops == 0) hipLaunchKernelGGL(TransChecker<0>, dim3(cus * 6), dim3(64), 0, sa, hist, 200000, 0.3f + 0.01f * q);
					else hipLaunchKernelGGL(TransChecker<1>, dim3(cus * 6), dim3(64), 0, sa, hist, 200000, 0.3f + 0.01f * q);
				}
				CHECK(hipStreamSynchronize(sa));
				CHECK(hipStreamSynchronize(sb));
			}
			std::vector<unsigned> h(64);
			CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
			unsigned long q[4] = { 0, 0, 0, 0 };
			for (int l = 0; l < 64; l++) q[l / 16] += h[l];
			printf("4 x v_rcp -> packed pairs%s, beside %-22s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n",
				nops ? " + 16 wait states" : "                 ", bn[mode], q[0], q[1], q[2], q[3]);
		}
	return 0;
}
