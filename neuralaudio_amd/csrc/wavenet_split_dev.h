// wavenet_split_dev.h -- device-side helpers shared by the two f16-split MFMA kernels: the stage interpreter (wavenet_split_kernels.hip,
// any model the plan builder accepts) and the compile-time specialised chains of the official architectures (wavenet_spec_kernels.hip).
// Arithmetic of one value: h = f16(v), l = f16(v - h); W*x = Wh*xh + Wh*xl + Wl*xh on v_mfma_f32_16x16x32_f16 with f32 accumulation.
#pragma once

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

namespace na
{
	namespace sp
	{
		typedef float f32x4 __attribute__((ext_vector_type(4)));
		typedef float f32x2 __attribute__((ext_vector_type(2)));
		typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
		typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
		typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
		typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
		typedef const int __attribute__((address_space(4)))* CInt;

#ifndef NA_ABL
#define NA_ABL 0 // ablation bit mask for tuning builds only (make SUFFIX=_ablN KEXTRA="-DNA_ABL=N -DNA_SP_QUICK", tools/ab_bench.sh); 0 in
                 // the product.  1: no activation math, 2: no MFMA, 4: no history loads / ring stores, 8: no barrier, 16: no weight staging,
                 // 32: no split arithmetic, 64: no LDS publish, 128: no A-operand LDS reads (one operand reused), 256: no tap reads,
                 // 512: ring loads and stores issued but all out of range (no traffic), 1024: only the stores so, 2048: only the loads
                 // (round-2 measurements: profiles/r02_ablation.txt)
#endif
#ifndef NA_PK_TANH
#define NA_PK_TANH 1 // tuning builds: 0 = the unpacked tanh in the activation phase
#endif
#define WOP(m) (((NA_ABL & 128) ? 0 : (m)) * 64)
		constexpr int OOB = (int)0x80000000;
		constexpr int FRAMES = WN_MAX_FRAMES; // 128 frames per launch = 8 tiles
		// LDS block image: one plane per channel group, GUARD quads in front of frame 0; the quad just before frame 0 is kept zero, so a tap
		// whose frame lies before the block start reads zeros by clamping its frame index to -1 (no exec masking, no select)
		constexpr int GUARD = 16;
		constexpr int PLANE = GUARD + FRAMES; // quads per plane (a multiple of 16: lanes of different planes never share a bank group)
		__device__ __forceinline__ int ImgIdx(int cg, int f) { return cg * PLANE + GUARD + f; }

		__device__ __forceinline__ __amdgpu_buffer_rsrc_t MakeRsrc(const void* base, unsigned bytes)
		{
			return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
		}

		__device__ __forceinline__ u32x4 BufLoad(__amdgpu_buffer_rsrc_t r, int voff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0); }
		__device__ __forceinline__ void BufStore(__amdgpu_buffer_rsrc_t r, u32x4 v, int voff) { __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0); }
		// ring traffic (streamed: every byte is read once, a launch or more after it was written; nt / sc0 / sc1 cache policies measured:
		// nt 2-4 % slower, the others within noise -- default policy)
		__device__ __forceinline__ u32x4 RingLoad(__amdgpu_buffer_rsrc_t r, int voff, int soff = 0) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }
		__device__ __forceinline__ void RingStore(__amdgpu_buffer_rsrc_t r, u32x4 v, int voff, int soff = 0) { __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0); }
		// tuning builds (NA_SPK_NT): the same with a cache-policy immediate (1 sc0, 2 nt, 16 sc1) for the rings of the long dilations
		template <int AUX>
		__device__ __forceinline__ u32x4 RingLoadAux(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX); }
		template <int AUX>
		__device__ __forceinline__ void RingStoreAux(__amdgpu_buffer_rsrc_t r, u32x4 v, int voff, int soff) { __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, AUX); }

		// ---- arithmetic -------------------------------------------------------------------------------------------------------

		__device__ __forceinline__ f32x4 Mfma(u32x4 a, u32x4 b, f32x4 c)
		{
			if (NA_ABL & 2) return f32x4{ c.x + __builtin_bit_cast(float, a.x ^ b.x), c.y, c.z, c.w };
			return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
		}

		// the "lo" product of the three-product split, Wl . xh: its A operand is [Wl | 0 0 0 0] per lane and k-block and only the h half
		// of the split quad B takes part -- the first 64 bits of either are exactly the operands of the K = 16 instruction (lane (i, q): k =
		// 4 q .. 4 q + 3), which does the same sixteen products per row in half the passes of the matrix pipe (tuning builds: -DNA_LO16=1)
#ifndef NA_LO16
#define NA_LO16 0
#endif
		// (SITE: 1 conv taps, 2 the 1x1, 4 array links, 8 heads -- NA_LO16 is a mask of the sites that use the K = 16 instruction)
		template <int SITE>
		__device__ __forceinline__ f32x4 MfmaLo(u32x4 a, u32x4 b, f32x4 c)
		{
#if NA_LO16
			if (!(NA_LO16 & SITE)) return Mfma(a, b, c);
			if (NA_ABL & 2) return f32x4{ c.x + __builtin_bit_cast(float, a.x ^ b.x), c.y, c.z, c.w };
			typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
			const u32x2 a2 = { a.x, a.y }, b2 = { b.x, b.y };
			f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a2), __builtin_bit_cast(f16x4_t, b2), c, 0, 0, 0);
			// the sources stay live past the instruction: with -amdgpu-mfma-vgpr-form this LLVM lets the result of the K = 16 form land on the
			// registers of a source that dies here (seen: v_mfma_f32_16x16x16_f16 v[18:21], v[54:55], v[18:19], v[20:23] -- the later passes
			// then read what the first ones wrote; the K = 32 form is protected)
			asm("" : "+v"(d) : "v"(a), "v"(b));
			return d;
#else
			return Mfma(a, b, c);
#endif
		}

		__device__ __forceinline__ unsigned PackHalf2(float a, float b)
		{
			const f16x2 h = __builtin_convertvector(f32x2{ a, b }, f16x2); // v_cvt_pk_f16_f32 (round to nearest even)
			return __builtin_bit_cast(unsigned, h);
		}

		// 4 channels (f32) -> split quad [h0 h1 | h2 h3 | l0 l1 | l2 l3]
		__device__ __forceinline__ u32x4 SplitQuad(f32x4 v)
		{
			if (NA_ABL & 32) return __builtin_bit_cast(u32x4, v);
			const f16x2 h01 = __builtin_convertvector(f32x2{ v.x, v.y }, f16x2);
			const f16x2 h23 = __builtin_convertvector(f32x2{ v.z, v.w }, f16x2);
			u32x4 q;
			q.x = __builtin_bit_cast(unsigned, h01);
			q.y = __builtin_bit_cast(unsigned, h23);
			// v - f32(h), exact in f32, one instruction each: v_fma_mix_f32 reads the f16 half directly (the compiler emits
			// v_cvt_f32_f16 + v_sub_f32 for the plain expression)
			float r0, r1, r2, r3;
			asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(q.x), "v"(v.x));
			asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(q.x), "v"(v.y));
			asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(q.y), "v"(v.z));
			asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(q.y), "v"(v.w));
			q.z = PackHalf2(r0, r1);
			q.w = PackHalf2(r2, r3);
			return q;
		}

		// Saturating variant for chains the static range proof does not cover (LeakyReLU models: wavenet_plan.cpp), used
		// for every value that reaches the stream STATE -- the residual stream (block images, rings) and the head accumulator (head
		// ring): a value beyond the f16 range is clamped to +-65504 instead of becoming inf (inf - inf = NaN in the lo half would poison
		// the rings for good; v_med3_f32 also turns a NaN into -65504).  A lane that clamped something drops a 1 into the stream's flag word
		// in LDS (`flagAddr`: a branch that audio never takes, no register carried through a chain that has none to spare); the wave that
		// closes the block counts a "range event" in the stream's state header when the word is set (CountRangeEvent).  The activations
		// themselves are split plainly: an overflow there surfaces in the residual stream one mat-mul later.  Inside the range it is
		// SplitQuad bit for bit.  (The wave's sticky IEEE overflow flag, TRAPSTS.EXCP, would have been the free way to remember: with the
		// default exception mask v_cvt_f16_f32(1e6) leaves it clear on gfx950 -- tools/microbench/trapsts_probe.hip.)
		__device__ __forceinline__ u32x4 SplitQuadSat(f32x4 v, unsigned flagAddr)
		{
			const float m = 65504.0f;
			// (plain expressions on purpose: `v` usually comes straight out of an MFMA, and the wait states a VALU read of an MFMA result needs
			// are inserted by the compiler for its own instructions, not for inline assembly -- a hand-written v_max3_f32 here read
			// half-written accumulators and raised false events)
			const float top = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z)), __builtin_fabsf(v.w));
			if (!(top <= m)) // (a NaN counts)
				*reinterpret_cast<__attribute__((address_space(3))) unsigned*>((__attribute__((address_space(3))) char*)(size_t)flagAddr) = 1u;
			v.x = __builtin_amdgcn_fmed3f(v.x, -m, m);
			v.y = __builtin_amdgcn_fmed3f(v.y, -m, m);
			v.z = __builtin_amdgcn_fmed3f(v.z, -m, m);
			v.w = __builtin_amdgcn_fmed3f(v.w, -m, m);
			return SplitQuad(v);
		}

		// the same for the stage interpreter, whose flag the compiler cannot prove wave-uniform (it travels through the stage loop): plain
		// expressions, the flag where the compiler wants it
		__device__ __forceinline__ u32x4 SplitQuadSatLoose(f32x4 v, int& hit)
		{
			const float m = 65504.0f;
			const float top = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z)), __builtin_fabsf(v.w));
			hit |= !(top <= m) ? 1 : 0;
			v.x = __builtin_amdgcn_fmed3f(v.x, -m, m);
			v.y = __builtin_amdgcn_fmed3f(v.y, -m, m);
			v.z = __builtin_amdgcn_fmed3f(v.z, -m, m);
			v.w = __builtin_amdgcn_fmed3f(v.w, -m, m);
			return SplitQuad(v);
		}
		// one event per wave and block in which a value left the f16 range (`live`: not a shadow wave of a partial workgroup)
		__device__ __forceinline__ void CountRangeEvent(int* header, int hit, int lane, bool live)
		{
			if (__builtin_amdgcn_ballot_w64(hit != 0) != 0 && live && lane == 0) __hip_atomic_fetch_add(&header[WN_RANGE_EVENT_SLOT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}

		// Activation.h:83-91, plain (unpacked) VALU on purpose: packed f32 instructions do not issue beside MFMAs on gfx950
		// (tools/microbench/mfma_f16_valu_mix.hip).  |x + e*x*|x|| == |x| + e*x^2 since 1 + e|x| > 0; division = num * v_rcp_f32(den).
		__device__ __forceinline__ float FastTanh(float x)
		{
			if (NA_ABL & 1) return x * 0.5f;
			const float ax = __builtin_fabsf(x);
			const float x2 = x * x;
			const float p = __builtin_fmaf(__builtin_fmaf(0.821226666969744f, ax, 0.893229853513558f), x2, __builtin_fmaf(2.45550750702956f, ax, 2.45550750702956f));
			const float den = __builtin_fmaf(2.44506634652299f + x2, __builtin_fmaf(0.814642734961073f, x2, ax), 2.44506634652299f);
			return (x * p) * __builtin_amdgcn_rcpf(den);
		}

		// The same on two channels with packed f32 math (13 instructions per pair instead of 20).  Packed f32 instructions stall next to
		// MFMAs of the same wave, but the activation is a phase of its own between the conv and the 1x1 chains, where halving the
		// instruction count wins (tools/microbench/mfma_f16_valu_mix.hip: 3 vs 6 cycles per element in bulk).
		__device__ __forceinline__ f32x2 FastTanh2(f32x2 x)
		{
			if (NA_ABL & 1) return x * 0.5f;
			f32x2 ax;
			ax.x = __builtin_fabsf(x.x);
			ax.y = __builtin_fabsf(x.y);
			const f32x2 x2 = x * x;
			const f32x2 num = x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2);
			const f32x2 den = 2.44506634652299f + (2.44506634652299f + x2) * (ax + 0.814642734961073f * x2);
			f32x2 r;
			r.x = __builtin_amdgcn_rcpf(den.x);
			r.y = __builtin_amdgcn_rcpf(den.y);
			return num * r;
		}

		// StdMath policy (Activation.h:37-40): tanh(x) = 1 - 2 / (e^(2x) + 1) on the exp2 / rcp units (absolute error ~1e-7)
		__device__ __forceinline__ float StdTanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.0f); }

		// Activation.h:110-118
		__device__ __forceinline__ float LeakyReLU(float v) { return v > 0.0f ? v : 0.01f * v; }

		// Range contract of the f16-split path: the hi part of every value is an f16, so the condition (= the input sample,
		// WaveNet.h:770) is clamped to +-limit -- a per-model bound (WaveNetPlan::condLimit) under which the residual stream stays below
		// 65504 -- where the reference's f32 chain would still be finite (its tanh saturates either way); a NaN sample reads as silence.
		__device__ __forceinline__ float ClampCond(float c, float limit)
		{
			c = (c != c) ? 0.0f : c;
			return __builtin_fminf(__builtin_fmaxf(c, -limit), limit);
		}

		__device__ __forceinline__ f32x4 Activate(f32x4 a, int flags)
		{
			f32x4 z;
			if (flags & WN_FLAG_LEAKY) // wave-uniform
			{
				z.x = LeakyReLU(a.x); z.y = LeakyReLU(a.y); z.z = LeakyReLU(a.z); z.w = LeakyReLU(a.w);
			}
			else if (flags & WN_FLAG_STD_TANH)
			{
				z.x = StdTanh(a.x); z.y = StdTanh(a.y); z.z = StdTanh(a.z); z.w = StdTanh(a.w);
			}
			else
			{
				z.x = FastTanh(a.x); z.y = FastTanh(a.y); z.z = FastTanh(a.z); z.w = FastTanh(a.w);
			}
			return z;
		}

		template <int NWAVES>
		__device__ __forceinline__ void BlockBarrier()
		{
			if (NA_ABL & 8) return;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
			__builtin_amdgcn_s_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
		}


		// per model group of one (possibly fused) launch; passed by value in the kernarg segment
		struct GroupArgs
		{
			const WnSplitStage* stages;
			const void* wsplit;
			const int* ringFrames;
			u32x4* state;
			const int* slots; // nullptr: contiguous, stream i uses slot0 + i / row0 + i
			const int* rows;
			int nstages, nrings, stateF4, wsplitQuads;
			float headScale;
			int numStreams, slot0, row0;
			int maxG;
			int firstBlock; // workgroups [firstBlock, next group's firstBlock) belong to this group
			int pack;       // packed launches: real streams per virtual stream; rows[] then holds `pack` rows per virtual stream (-1: unused)
			float condLimit; // input samples are clamped to +-condLimit (WnModelDev::cond_limit), NaN reads as silence
			int arch;       // specialised chains (wavenet_spec_kernels.hip): which member of the launch's architecture family
			int gps0, gps1; // ... packed launches: channel groups per real stream of the first / the last layer array (1, 2, 4; 0: two streams per group)
			int saturate;   // stage interpreter: 1 = split with SplitQuadSat and count range events (models without a range proof)
		};

		struct LaunchArgs
		{
			GroupArgs g[WN_FRAME_MAX_GROUPS];
			int numGroups;
		};

	}
}
