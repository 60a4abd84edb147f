// Third probe of the four-streams-per-wave kernel's fault (profiles/r06_quad_race.txt).  Bisection inside the kernel says the vulnerable part is
// its ROW SUMS: v_pk_fma_f32 acc, w, h, acc with op_sel / op_sel_hi broadcasting ONE half of a register pair that a ds_read_b128 has just
// delivered (16 lanes of a stream read the same 16 bytes), while the packed activations are fine.  This probe does exactly that --
// LDS write of a per-"stream" vector by 16 lanes, wave-level sync, four ds_read_b128 broadcasts, 32 packed FMAs with op_sel broadcasts --
// beside the same sums with scalar FMAs, and counts disagreements per lane.  Run it alone and beside tools/runs/r06t_aggressor.py (another
// process stepping the split WaveNet kernel) or beside its own MFMA / LDS-DMA-free burner.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_lds_opsel pk_lds_opsel.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void WaveSync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// VARIANT 0: the values come from the LDS broadcast and are spread over the pair by op_sel (the kernel's form).  1: same values, but every
// broadcast pair {x, x} is built in registers first (two v_mov: the packed FMA runs with default op_sel).  2: no LDS at all -- the 16
// values come from VALU arithmetic -- with the op_sel broadcast.  3: the LDS holds every value TWICE ({x, x} pairs, eight ds_read_b128): the
// packed FMA reads the delivered pair as it is, default op_sel, no copy.  4: as 0, but the values arrive by sixteen ds_read_b32 (what the
// compiler's own pairing of scalar code produces: lstm_kernels.hip's runtime-shaped kernel had 200 such sites).  5: the values arrive by
// global_load_dwordx4 (from gsrc, L2-resident).  6: as 0 with ds_read_b64.
__device__ const float* g_src;
template <int VARIANT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) Checker(unsigned* hist, int iters, float seed)
{
	__shared__ __attribute__((aligned(16))) float lds[4 * 20 * 17 + 4 * 132 + 4 * 40 * 17];
	const int lane = threadIdx.x, unit = lane & 15, sub = lane >> 4;
	float* hw = lds + 4 * 132 + sub * 340 + unit;
	const float* hrd = lds + 4 * 132 + sub * 340;
	float* hw2 = lds + 4 * 20 * 17 + 4 * 132 + sub * 680 + 2 * unit; // variant 3: [stream][entry][16 pairs]
	const float* hrd2 = lds + 4 * 20 * 17 + 4 * 132 + sub * 680;
	f2 wA[16], wB[16];
	for (int k = 0; k < 16; k++)
	{
		wA[k] = f2{ 0.02f * (float)((unit + k) % 16) - 0.15f, 0.03f * (float)((unit * 3 + k) % 16) - 0.2f };
		wB[k] = f2{ 0.025f * (float)((unit + 5 * k) % 16) - 0.18f, 0.015f * (float)((unit * 7 + k) % 16) - 0.1f };
	}
	float h = seed + 0.01f * lane;
	unsigned bad = 0;
	for (int i = 0; i < iters; i++)
	{
		const int e = i & 15;
		hw[(e + 1) * 20] = h;
		if (VARIANT == 3) *reinterpret_cast<f2*>(hw2 + (e + 1) * 40) = f2{ h, h };
		WaveSync();
		float hv[16];
#pragma unroll
		for (int q = 0; q < 4; q++)
		{
			const float4 t = *reinterpret_cast<const float4*>(hrd + (e + 1) * 20 + 4 * q);
			hv[4 * q + 0] = t.x; hv[4 * q + 1] = t.y; hv[4 * q + 2] = t.z; hv[4 * q + 3] = t.w;
		}
		f2 aP = f2{ 0.1f, -0.1f }, bP = f2{ 0.05f, 0.02f };
		float a0 = 0.1f, a1 = -0.1f, b0 = 0.05f, b1 = 0.02f;
		f2 hp[16];
		if (VARIANT == 3)
		{
#pragma unroll
			for (int q = 0; q < 8; q++)
			{
				const float4 t = *reinterpret_cast<const float4*>(hrd2 + (e + 1) * 40 + 4 * q);
				hp[2 * q] = f2{ t.x, t.y };
				hp[2 * q + 1] = f2{ t.z, t.w };
			}
		}
		if (VARIANT == 2)
		{
#pragma unroll
			for (int k = 0; k < 16; k++) hv[k] = __builtin_fmaf(h, 0.37f + 0.01f * k, 0.05f * (float)(k - 8)); // (register-born values)
		}
		if (VARIANT == 4 || VARIANT == 6)
		{
			const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(hrd + (e + 1) * 20);
			if (VARIANT == 4)
			{
#pragma unroll
				for (int k = 0; k < 16; k++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(hv[k]) : "v"(a), "n"(4 * k));
			}
			else
			{
#pragma unroll
				for (int k = 0; k < 8; k++)
				{
					f2 t;
					asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(a), "n"(8 * k));
					hv[2 * k] = t.x; hv[2 * k + 1] = t.y;
				}
			}
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]), "+v"(hv[6]), "+v"(hv[7]), "+v"(hv[8]),
				"+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]), "+v"(hv[12]), "+v"(hv[13]), "+v"(hv[14]), "+v"(hv[15]));
		}
		if (VARIANT == 5)
		{
			typedef float gf4 __attribute__((ext_vector_type(4)));
			const __attribute__((address_space(1))) gf4* gp = (const __attribute__((address_space(1))) gf4*)(g_src + ((i & 63) * 64 + sub * 16));
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				const gf4 t = gp[q];
				hv[4 * q + 0] = t.x; hv[4 * q + 1] = t.y; hv[4 * q + 2] = t.z; hv[4 * q + 3] = t.w;
			}
		}
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			if (VARIANT == 3)
			{
				aP = __builtin_elementwise_fma(wA[k], hp[k], aP);
				bP = __builtin_elementwise_fma(wB[k], hp[k], bP);
			}
			else if (VARIANT == 1)
			{
				f2 xx = f2{ hv[k], hv[k] };
				asm volatile("" : "+v"(xx)); // the pair exists as two registers: no op_sel on the FMA
				aP = __builtin_elementwise_fma(wA[k], xx, aP);
				bP = __builtin_elementwise_fma(wB[k], xx, bP);
			}
			else
			{
				aP = __builtin_elementwise_fma(wA[k], f2{ hv[k], hv[k] }, aP); // v_pk_fma_f32 with an op_sel broadcast of hv[k]
				bP = __builtin_elementwise_fma(wB[k], f2{ hv[k], hv[k] }, bP);
			}
		}
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			float x = hv[k];
			asm volatile("" : "+v"(x)); // (an opaque copy: the scalar sums must not be re-paired)
			a0 = __builtin_fmaf(wA[k].x, x, a0);
			a1 = __builtin_fmaf(wA[k].y, x, a1);
			b0 = __builtin_fmaf(wB[k].x, x, b0);
			b1 = __builtin_fmaf(wB[k].y, x, b1);
		}
		float pax = aP.x, pay = aP.y, pbx = bP.x, pby = bP.y;
		asm volatile("" : "+v"(pax), "+v"(pay), "+v"(pbx), "+v"(pby));
		if (pax != a0 || pay != a1 || pbx != b0 || pby != b1) bad++;
		// next state from the scalar sums: bounded, different per lane
		h = 0.5f * a0 - 0.25f * b1 + 0.1f * a1 * b0 + 1e-3f * (float)(i & 7);
	}
	if (bad) atomicAdd(&hist[lane], bad);
}

// ---- the same question for v_fma_mix_f32, the other instruction of the library that selects register halves with op_sel (the split kernels
// read f16 halves of 128-bit loads with it: tools/analysis found 369 sites behind ds_read_b128, 384 behind global / flat dwordx4 loads) ----
// SRC 0: sixteen packed-f16 words out of LDS by four ds_read_b128; 1: out of global memory by four global_load_dwordx4.  Checked against the
// same sum from v_mov copies of the words, halves widened with v_cvt_f32_f16 (v_fma_mix converts exactly and fuses: bit-identical).
__device__ const unsigned* g_src16;
template <int SRC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) MixChecker(unsigned* hist, int iters, float seed)
{
	__shared__ __attribute__((aligned(16))) unsigned lds[4 * 340];
	const int lane = threadIdx.x, unit = lane & 15, sub = lane >> 4;
	unsigned* hw = lds + sub * 340 + unit;
	const unsigned* hrd = lds + sub * 340;
	float w[16];
	for (int k = 0; k < 16; k++) w[k] = 0.02f * (float)((unit + 3 * k) % 16) - 0.15f;
	float h = seed + 0.01f * lane;
	unsigned bad = 0;
	for (int i = 0; i < iters; i++)
	{
		const int e = i & 15;
		typedef _Float16 h2 __attribute__((ext_vector_type(2)));
		const h2 ph = { (_Float16)h, (_Float16)(0.5f * h + 0.125f) };
		hw[(e + 1) * 20] = __builtin_bit_cast(unsigned, ph);
		WaveSync();
		unsigned q[16];
		if (SRC == 0)
		{
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				const uint4 t = *reinterpret_cast<const uint4*>(hrd + (e + 1) * 20 + 4 * j);
				q[4 * j + 0] = t.x; q[4 * j + 1] = t.y; q[4 * j + 2] = t.z; q[4 * j + 3] = t.w;
			}
		}
		else
		{
			typedef unsigned gu4 __attribute__((ext_vector_type(4)));
			const __attribute__((address_space(1))) gu4* gp = (const __attribute__((address_space(1))) gu4*)(g_src16 + ((i & 63) * 64 + sub * 16));
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				const gu4 t = gp[j];
				q[4 * j + 0] = t.x; q[4 * j + 1] = t.y; q[4 * j + 2] = t.z; q[4 * j + 3] = t.w;
			}
		}
		float mHi = 0.1f, mLo = -0.1f;
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(mHi) : "v"(q[k]), "v"(w[k]));
			asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(mLo) : "v"(q[k]), "v"(w[k]));
		}
		__builtin_amdgcn_sched_barrier(0);
		float rHi = 0.1f, rLo = -0.1f;
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			unsigned c;
			asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(q[k]));
			const h2 hh = __builtin_bit_cast(h2, c);
			rLo = __builtin_fmaf((float)hh.x, w[k], rLo);
			rHi = __builtin_fmaf((float)hh.y, w[k], rHi);
		}
		if (__builtin_bit_cast(unsigned, mHi) != __builtin_bit_cast(unsigned, rHi) || __builtin_bit_cast(unsigned, mLo) != __builtin_bit_cast(unsigned, rLo)) bad++;
		h = 0.5f * rHi - 0.25f * rLo + 1e-3f * (float)(i & 7);
		h = h > 4.0f ? 4.0f : (h < -4.0f ? -4.0f : h);
	}
	if (bad) atomicAdd(&hist[lane], bad);
}

// ---- built-in aggressors (another stream of this process): what in the split kernel does it? ----
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// mode 1: MFMA f16 back to back; 2: LDS-DMA (buffer_load ... lds, 16 bytes per lane) back to back; 3: ds_read_b128 / ds_write_b128 traffic;
// 4: plain VALU; 5: MFMA + LDS-DMA + LDS reads together (the split kernel's mix)
__global__ void __launch_bounds__(256) Burner(const float* __restrict__ src, float* out, int iters, int mode)
{
	__shared__ __attribute__((aligned(16))) float buf[2 * 256 * 4];
	f32x4 c0 = { 0, 0, 0, 0 }, c1 = c0;
	f16x8 a, b;
	for (int k = 0; k < 8; k++)
	{
		a[k] = (_Float16)(0.001f * (float)(threadIdx.x + k));
		b[k] = (_Float16)(0.002f * (float)(k + 1));
	}
	__amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
	float v0 = threadIdx.x * 0.001f;
	f32x4 t = { 0, 0, 0, 0 };
	for (int i = 0; i < iters; i++)
	{
		if (mode == 1 || mode == 5)
		{
			c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
			c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
		}
		if (mode == 2 || mode == 5)
		{
			// (a wave's 64 lanes land in 1 KB of LDS at the M0 base: the wave's quarter of buf, this half)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf + (i & 1) * 1024 + (threadIdx.x >> 6) * 256), 16, (int)((threadIdx.x * 16 + (i & 63) * 4096) & 0xfffff), 0, 0, 0);
		}
		if (mode == 3 || mode == 5)
		{
			t += *reinterpret_cast<const f32x4*>(buf + ((threadIdx.x * 4 + i * 16) & 2044));
			if (mode == 3) *reinterpret_cast<f32x4*>(buf + threadIdx.x * 4) = t;
		}
		if (mode == 4) v0 = __builtin_fmaf(v0, 0.999f, 0.001f);
		if (mode == 6) // v_fma_mix_f32 with op_sel picking f16 halves (the split kernel's SplitQuad)
		{
			float r;
			asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, v0)), "v"(1.0f), "v"(0.001f));
			v0 = r * 0.5f + 0.1f;
		}
		if (mode == 7) // packed f32 FMA with the OTHER op_sel pattern
		{
			f2 pp = f2{ v0, t.x };
			asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,1] op_sel_hi:[0,1,0]\n" : "+v"(pp) : "v"(f2{ 0.5f, 0.25f }), "v"(f2{ 0.1f, 0.2f }));
			v0 = pp.x; t.x = pp.y;
		}
		if (mode == 8) // v_cvt_pk_f16_f32 + v_pk_fma_f16 with op_sel
		{
			unsigned hh;
			asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\nv_pk_fma_f16 %0, %0, %0, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n" : "=&v"(hh) : "v"(v0), "v"(t.x));
			v0 = v0 * 0.999f + (float)(hh & 1) * 1e-6f;
		}
	}
	__builtin_amdgcn_s_waitcnt(0);
	out[blockIdx.x * 256 + threadIdx.x] = c0.x + c1.y + v0 + t.x + buf[threadIdx.x];
}

int main(int argc, char** argv)
{
	if (argc > 2)
	{
		// pk_lds_opsel <rounds> burners: every built-in aggressor in turn
		const int rounds = atoi(argv[1]);
		int cus = 0;
		CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
		unsigned* hist;
		float *src, *sink;
		CHECK(hipMalloc(&hist, 64 * sizeof(unsigned)));
		CHECK(hipMalloc(&src, 1 << 20));
		CHECK(hipMemset(src, 0, 1 << 20));
		CHECK(hipMalloc(&sink, (size_t)cus * 4 * 256 * sizeof(float)));
		hipStream_t sa, sb;
		CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
		CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
		const char* names[9] = { "nothing", "MFMA waves", "LDS-DMA waves", "LDS read / write waves", "VALU waves", "MFMA + LDS-DMA + LDS reads", "v_fma_mix_f32 op_sel waves",
			"v_pk_fma_f32 other-op_sel waves", "v_cvt_pk_f16 / v_pk_fma_f16 waves" };
		const int its[9] = { 0, 1500000, 300000, 1500000, 4000000, 300000, 3000000, 3000000, 3000000 };
		for (int mode = 0; mode < 9; mode++)
		{
			CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
			for (int r = 0; r < rounds; r++)
			{
				if (mode > 0) hipLaunchKernelGGL(Burner, dim3(cus * 4), dim3(256), 0, sb, src, sink, its[mode], mode);
				for (int q = 0; q < 4; q++) hipLaunchKernelGGL(Checker<0>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
				CHECK(hipStreamSynchronize(sa));
				if (mode > 0 && r == 0) printf("   (burner %s when the checkers were done)\n", hipStreamQuery(sb) == hipErrorNotReady ? "still running" : "ALREADY FINISHED");
				CHECK(hipStreamSynchronize(sb));
			}
			std::vector<unsigned> h(64);
			CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
			unsigned long q[4] = { 0, 0, 0, 0 };
			for (int l = 0; l < 64; l++) q[l / 16] += h[l];
			printf("beside %-28s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n", names[mode], q[0], q[1], q[2], q[3]);
		}
		// what in the victim matters: the three variants beside the MFMA waves
		const char* vn[7] = { "LDS values, op_sel broadcast", "LDS values, pairs built in registers", "VALU values, op_sel broadcast", "LDS PAIRS, default op_sel",
			"ds_read_b32 values, op_sel", "global_load_dwordx4 values, op_sel", "ds_read_b64 values, op_sel" };
		{
			std::vector<float> pat(64 * 64);
			for (size_t k = 0; k < pat.size(); k++) pat[k] = 0.05f * (float)((k * 37) % 23) - 0.5f;
			float* gs;
			CHECK(hipMalloc(&gs, pat.size() * sizeof(float)));
			CHECK(hipMemcpy(gs, pat.data(), pat.size() * sizeof(float), hipMemcpyHostToDevice));
			CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_src), &gs, sizeof(gs)));
		}
		for (int v = 0; v < 7; v++)
		{
			CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
			for (int r = 0; r < rounds; r++)
			{
				hipLaunchKernelGGL(Burner, dim3(cus * 4), dim3(256), 0, sb, src, sink, its[1], 1);
				for (int q = 0; q < 4; q++)
				{
					if (v == 0) hipLaunchKernelGGL(Checker<0>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
					if (v == 1) hipLaunchKernelGGL(Checker<1>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
					if (v == 2) hipLaunchKernelGGL(Checker<2>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
					if (v == 3) hipLaunchKernelGGL(Checker<3>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
					if (v == 4) hipLaunchKernelGGL(Checker<4>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
					if (v == 5) hipLaunchKernelGGL(Checker<5>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
					if (v == 6) hipLaunchKernelGGL(Checker<6>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
				}
				CHECK(hipStreamSynchronize(sa));
				CHECK(hipStreamSynchronize(sb));
			}
			std::vector<unsigned> h(64);
			CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
			unsigned long q[4] = { 0, 0, 0, 0 };
			for (int l = 0; l < 64; l++) q[l / 16] += h[l];
			printf("beside MFMA waves, victim = %-38s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n", vn[v], q[0], q[1], q[2], q[3]);
		}
		// v_fma_mix_f32 with op_sel as the victim
		{
			std::vector<unsigned> pat(64 * 64);
			for (size_t k = 0; k < pat.size(); k++)
			{
				const _Float16 lo = (_Float16)(0.05f * (float)((k * 37) % 23) - 0.5f), hi = (_Float16)(0.03f * (float)((k * 11) % 29) - 0.4f);
				unsigned short ul, uh;
				memcpy(&ul, &lo, 2); memcpy(&uh, &hi, 2);
				pat[k] = (unsigned)ul | ((unsigned)uh << 16);
			}
			unsigned* gs;
			CHECK(hipMalloc(&gs, pat.size() * sizeof(unsigned)));
			CHECK(hipMemcpy(gs, pat.data(), pat.size() * sizeof(unsigned), hipMemcpyHostToDevice));
			CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_src16), &gs, sizeof(gs)));
			const char* mn[2] = { "ds_read_b128 words", "global_load_dwordx4 words" };
			for (int beside = 0; beside < 2; beside++)
				for (int v = 0; v < 2; v++)
				{
					CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
					for (int r = 0; r < rounds; r++)
					{
						if (beside) hipLaunchKernelGGL(Burner, dim3(cus * 4), dim3(256), 0, sb, src, sink, its[1], 1);
						for (int q = 0; q < 4; q++)
						{
							if (v == 0) hipLaunchKernelGGL(MixChecker<0>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
							else hipLaunchKernelGGL(MixChecker<1>, dim3(cus * 4), dim3(64), 0, sa, hist, 40000, 0.2f + 0.003f * (4 * r + q));
						}
						CHECK(hipStreamSynchronize(sa));
						CHECK(hipStreamSynchronize(sb));
					}
					std::vector<unsigned> hh(64);
					CHECK(hipMemcpy(hh.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
					unsigned long qq[4] = { 0, 0, 0, 0 };
					for (int l = 0; l < 64; l++) qq[l / 16] += hh[l];
					printf("beside %-10s victim = v_fma_mix_f32 op_sel on %-26s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n", beside ? "MFMA waves," : "nothing,", mn[v], qq[0], qq[1], qq[2], qq[3]);
				}
		}
		return 0;
	}
	const int rounds = argc > 1 ? atoi(argv[1]) : 40;
	int cus = 0;
	CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
	unsigned* hist;
	CHECK(hipMalloc(&hist, 64 * sizeof(unsigned)));
	CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
	for (int r = 0; r < rounds; r++)
	{
		hipLaunchKernelGGL(Checker<0>, dim3(cus * 4), dim3(64), 0, 0, hist, 40000, 0.2f + 0.003f * r);
		CHECK(hipDeviceSynchronize());
	}
	std::vector<unsigned> h(64);
	CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
	unsigned long q[4] = { 0, 0, 0, 0 };
	for (int l = 0; l < 64; l++) q[l / 16] += h[l];
	printf("LDS broadcast -> packed FMA with op_sel vs scalar: mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n", q[0], q[1], q[2], q[3]);
	return 0;
}
