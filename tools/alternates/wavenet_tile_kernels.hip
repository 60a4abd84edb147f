// wavenet_kernels.hip -- gfx950 (CDNA4) kernels for the NAM WaveNet hot path.
//
// Replaces the CPU inner loops of the reference's Internal WaveNet engine for MANY independent
// audio streams at once:
//   WaveNetModelT::Process          NeuralAudio/WaveNet.h:768-799
//   WaveNetLayerArrayT::Process     NeuralAudio/WaveNet.h:632-661
//   WaveNetLayerT::Process          NeuralAudio/WaveNet.h:462-494
//   Conv1DT::Process                NeuralAudio/WaveNet.h:139-290
//   DenseLayerT::Process/ProcessAcc NeuralAudio/WaveNet.h:336-383
//   FastMath::Tanh / LeakyReLU      NeuralAudio/Activation.h:83-91,110-118
//   WaveNetModelT::Prewarm          NeuralAudio/WaveNet.h:746-766 (+ :607-630, :74-82)
//
// Mapping (see wavenet_dev.h): one wave64 = one stream x one block of <= 128 frames.  The wave
// walks the stage program; every mat-mul is v_mfma_f32_16x16x4_f32 (exact f32, M = out channels,
// N = 16 frames, K = 4 input channels).  Activations stay in the MFMA C/D fragment layout, which
// is also the layout of the LDS block buffer and of the per-layer HBM history rings, so a tile
// is stored/loaded as one float4 per lane (1 KB coalesced per 16 frames x 16 channels).
// History taps that fall inside the current block come from LDS, older ones from the HBM ring.
#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

namespace na
{
	typedef float f32x4 __attribute__((ext_vector_type(4)));

	typedef float f32x2 __attribute__((ext_vector_type(2)));

#if defined(NA_ABL) && (NA_ABL & 1)
#define NA_MFMA(a, b, c, x, y, z) ((c) + (a) * (b))
#else
#define NA_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z)
#endif
#ifndef NA_ABL
#define NA_ABL 0 // ablation bit mask for tuning builds only (tools/ablate.sh); 0 in the product
#endif

	// Activation.h:83-91 -- same association as the reference; IEEE division.  (prewarm kernel, once per model)
	__device__ __forceinline__ float FastTanh(float x)
	{
		const float ax = fabsf(x);
		const float x2 = x * x;
		const float num = x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2);
		const float den = 2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax);
		return num / den;
	}

	// Activation.h:110-118
	__device__ __forceinline__ float LeakyReLU(float x) { return x > 0.0f ? x : 0.01f * x; }

	__device__ __forceinline__ f32x2 Abs2(f32x2 v)
	{
		f32x2 r;
		r.x = __builtin_fabsf(v.x);
		r.y = __builtin_fabsf(v.y);
		return r;
	}

	// Hot-path form of Activation.h:83-91 on two lanes' worth of data: same association, packed f32 math
	// (v_pk_fma/v_pk_mul), division as num * v_rcp_f32(den) (1 ulp; the north-star tolerance is 1e-4 RMS).
	__device__ __forceinline__ f32x2 FastTanh2(f32x2 x)
	{
		const f32x2 ax = Abs2(x);
		const f32x2 x2 = x * x;
		const f32x2 num = x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2);
		const f32x2 den = 2.44506634652299f + (2.44506634652299f + x2) * Abs2(x + 0.814642734961073f * x * ax);
		f32x2 r;
		r.x = __builtin_amdgcn_rcpf(den.x);
		r.y = __builtin_amdgcn_rcpf(den.y);
		return num * r;
	}

	__device__ __forceinline__ f32x4 Activate(f32x4 v, bool leaky)
	{
		f32x4 r;
		if (NA_ABL & 8) return v;
		if (leaky)
		{
			r.x = LeakyReLU(v.x); r.y = LeakyReLU(v.y); r.z = LeakyReLU(v.z); r.w = LeakyReLU(v.w);
		}
		else
		{
			const f32x2 lo = FastTanh2(f32x2{ v.x, v.y });
			const f32x2 hi = FastTanh2(f32x2{ v.z, v.w });
			r = f32x4{ lo.x, lo.y, hi.x, hi.y };
		}
		return r;
	}

	__device__ __forceinline__ f32x4 Mfma4(f32x4 a, f32x4 b, f32x4 c)
	{
		// four k-steps: element kk of the A/B float4s is k-slot (lane group, kk)
		c = NA_MFMA(a.x, b.x, c, 0, 0, 0);
		c = NA_MFMA(a.y, b.y, c, 0, 0, 0);
		c = NA_MFMA(a.z, b.z, c, 0, 0, 0);
		c = NA_MFMA(a.w, b.w, c, 0, 0, 0);
		return c;
	}

	typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
	constexpr int WN_OOB = (int)0x80000000; // buffer offset that is always out of range: loads return 0, stores are dropped

	// Raw buffer access: SGPR resource (base + size) + 32-bit per-lane byte offset + SGPR offset.  No 64-bit
	// per-lane address math, and out-of-range lanes are predicated off by the hardware bounds check.
	__device__ __forceinline__ __amdgpu_buffer_rsrc_t MakeRsrc(const void* base, unsigned bytes)
	{
		return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
	}

	__device__ __forceinline__ f32x4 BufLoad(__amdgpu_buffer_rsrc_t r, int voff, int soff)
	{
		return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
	}

	__device__ __forceinline__ void BufStore(__amdgpu_buffer_rsrc_t r, f32x4 v, int voff)
	{
		__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
	}

	// Workgroup barrier that orders LDS traffic only: global stores in flight (this layer's ring stores, which
	// nobody reads back inside the launch) are NOT drained, unlike __syncthreads().
	template <int WPS>
	__device__ __forceinline__ void BlockBarrier()
	{
		if (NA_ABL & 16) return;
		if (WPS > 1)
		{
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
			__builtin_amdgcn_s_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
		}
		else
		{
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
			__builtin_amdgcn_wave_barrier();
		}
	}

	// Store this wave's TPW tiles (block tiles tb .. tb+TPW-1) of a layer output: to the LDS block buffer
	// (in-block taps of the next layer) and to the next layer's HBM ring (history for LATER blocks: only the
	// last R-128 = roundup16((K-1)*dilation) frames of a block can ever be read back).
	template <int TPW>
	__device__ __forceinline__ void Publish(const f32x4 (&x)[TPW], f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int G, int pos0, int R,
		int n, int tb, int g, int j)
	{
		if (g < G)
		{
			const int step = G * 16;
			const int idx0 = (tb * G + g) * 16 + j;
			const int f0 = tb * 16 + j;
			const int firstKept = n - (R - WN_MAX_FRAMES);
			int p = pos0 + f0;
			if (p >= R) p -= R;
#pragma unroll
			for (int t = 0; t < TPW; t++)
			{
				xb[idx0 + t * step] = x[t];
				const int f = f0 + t * 16;
				const int voff = (ringOff + ((p >> 4) * G + g) * 16 + (p & 15)) * 16;
				if (!(NA_ABL & 4)) BufStore(srsrc, x[t], (f < n && f >= firstKept) ? voff : WN_OOB);
				p += 16;
				if (p >= R) p -= R;
			}
		}
	}

	constexpr int WN_STAGE_INTS = (int)(sizeof(WnStage) / sizeof(int));
	constexpr int WN_APF = 3; // conv rounds whose weight fragments are prefetched one stage ahead (A1: K = 3 -> all of them)

	// The stage table is wave-uniform and read-only: read it through the constant address space so the compiler
	// emits scalar loads (s_load_dwordx*) and every field lives in an SGPR -- control flow stays scalar, no VALU.
	typedef const int __attribute__((address_space(4)))* ConstIntPtr;

	__device__ __forceinline__ WnStage LoadStage(const WnStage* __restrict__ stages, int s)
	{
		WnStage sd;
		ConstIntPtr src = (ConstIntPtr)(const int*)(stages + s);
		int* dst = reinterpret_cast<int*>(&sd);
#pragma unroll
		for (int i = 0; i < WN_STAGE_INTS; i++) dst[i] = src[i];
		return sd;
	}

	template <int TPW>
	__device__ __forceinline__ void MfmaRound(f32x4 (&acc)[TPW], f32x4 a, const f32x4 (&b)[TPW])
	{
		// element kk of the A/B float4s is k-slot (lane group, kk); tiles interleaved so no MFMA waits on its predecessor
#pragma unroll
		for (int t = 0; t < TPW; t++) acc[t] = NA_MFMA(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
		for (int t = 0; t < TPW; t++) acc[t] = NA_MFMA(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
		for (int t = 0; t < TPW; t++) acc[t] = NA_MFMA(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
		for (int t = 0; t < TPW; t++) acc[t] = NA_MFMA(a.w, b[t].w, acc[t], 0, 0, 0);
	}

	// Generic conv (any channel-group count; tap / channel group of every lane group from the quad table in LDS).
	// B fragment of (round, tile): frame off = 16*(tb+t) + j - shift; off >= 0 -> LDS copy of the current block,
	// off < 0 -> ring history in HBM; classified per (round, tile) with scalar compares on the round's min/max shift.
	template <int TPW>
	__device__ __forceinline__ void ConvRounds(f32x4 (&acc)[TPW], const f32x4 (&apf)[WN_APF], const WnStage& sd, const WnQuad* sQ,
		__amdgpu_buffer_rsrc_t wrsrc, const f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int pos0, int tb, int lane, int g, int j)
	{
		const int G = sd.G;
		const int R = sd.ring_frames;
		const int step = G * 16;
		for (int r = 0; r < sd.nrounds; r++)
		{
			f32x4 a;
			if (r == 0) a = apf[0];
			else if (r == 1) a = apf[1];
			else if (r == 2) a = apf[2];
			else a = BufLoad(wrsrc, lane * 16, (sd.wconv_off + r * 64) * 16);
			const WnQuad qd = sQ[sd.qdesc_off + r * 4 + g];
			const int smin = __builtin_amdgcn_readfirstlane(qd.smin);
			const int smax = __builtin_amdgcn_readfirstlane(qd.smax);
			const int off0 = tb * 16 + j - qd.shift;
			const int idx0 = ((off0 >> 4) * G + qd.cg) * 16 + (off0 & 15);
			int p = pos0 + off0;
			if (p < 0) p += R;
			f32x4 b[TPW];
#pragma unroll
			for (int t = 0; t < TPW; t++)
			{
				const int base = (tb + t) * 16;
				if (base - smax >= 0)
				{
					b[t] = xb[idx0 + t * step];
				}
				else
				{
					const int voff = (sd.ring_off + ((p >> 4) * G + qd.cg) * 16 + (p & 15)) * 16;
					if ((NA_ABL & 2)) b[t] = xb[0];
					else if (base + 15 - smin < 0) b[t] = BufLoad(srsrc, voff, 0);
					else
					{
						const int off = off0 + t * 16;
						const int idx = idx0 + t * step;
						const f32x4 l = xb[idx < 0 ? 0 : idx];
						const f32x4 h = BufLoad(srsrc, off < 0 ? voff : WN_OOB, 0);
						b[t] = (off < 0) ? h : l;
					}
				}
				p += 16;
				if (p >= R) p -= R;
			}
			MfmaRound<TPW>(acc, a, b);
		}
	}

	// Conv for channel-group counts that divide 4 (C = 4, 8, 16 -> G = 1, 2, 4): a round then holds 4/G whole taps, so
	// tap / channel group / shift of a lane group follow from (round, lane group) arithmetically and the in-block /
	// history classification of a tile needs only SCALAR compares -- no quad-table lookups.  Same packing as PackConv
	// (quad q = 4r + g -> tap q / G, channel group q % G), same results as ConvRounds.
	template <int G, int TPW>
	__device__ __forceinline__ void ConvFast(f32x4 (&acc)[TPW], const f32x4 (&apf)[WN_APF], const WnStage& sd, __amdgpu_buffer_rsrc_t wrsrc,
		const f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int pos0, int tb, int lane, int g, int j)
	{
		constexpr int TPR = 4 / G; // taps per round
		const int K = sd.ksize;
		const int d = sd.dilation;
		const int R = sd.ring_frames;
		const int tg = g / G;
		const int cg16 = (g % G) * 16;
		const int ringBase = sd.ring_off + cg16;
		for (int r = 0; r < sd.nrounds; r++)
		{
			f32x4 a;
			if (r == 0) a = apf[0];
			else if (r == 1) a = apf[1];
			else if (r == 2) a = apf[2];
			else a = BufLoad(wrsrc, lane * 16, (sd.wconv_off + r * 64) * 16);
			const int tapLo = r * TPR;                               // scalar; <= K-1 because r < nrounds
			const int smax = d * (K - 1 - tapLo);                    // largest shift in this round
			const int smin = d * max(K - 1 - (tapLo + TPR - 1), 0);  // smallest (idle quads read shift 0)
			int shift = smax;
			if (TPR > 1) shift = d * max(K - 1 - (tapLo + tg), 0);
			const int off0 = tb * 16 + j - shift;
			const int idx0 = (off0 >> 4) * (G * 16) + (off0 & 15) + cg16;
			f32x4 b[TPW];
#pragma unroll
			for (int t = 0; t < TPW; t++)
			{
				const int base = (tb + t) * 16; // scalar
				if (base - smax >= 0)
				{
					b[t] = xb[idx0 + t * (G * 16)]; // whole tile inside the current block
				}
				else
				{
					int p = pos0 + off0 + t * 16;
					if (p < 0) p += R;
					const int voff = (ringBase + (p >> 4) * (G * 16) + (p & 15)) * 16;
					if ((NA_ABL & 2)) b[t] = xb[0];
					else if (base + 15 - smin < 0)
					{
						b[t] = BufLoad(srsrc, voff, 0); // whole tile is history
					}
					else
					{
						const int off = off0 + t * 16;
						const int idx = idx0 + t * (G * 16);
						const f32x4 l = xb[idx < 0 ? 0 : idx];
						const f32x4 h = BufLoad(srsrc, off < 0 ? voff : WN_OOB, 0);
						b[t] = (off < 0) ? h : l;
					}
				}
			}
			MfmaRound<TPW>(acc, a, b);
		}
	}

	// Tuning aid: when a trace buffer is installed (NA_DebugSetTraceBuffer), workgroup 0 stamps the shader clock at four
	// points of every stage: trace[((stage * 4 + point) * waves + wave)].  Null in normal use (one scalar branch).
#define NA_TRACE(point) \
	if (trace != nullptr && (int)blockIdx.x == traceBlock && lane == 0) trace[((s * 4 + (point)) * WPS) + wave] = (long long)__builtin_readcyclecounter()

	// grid = active streams of one model; block = WPS waves: the WPS waves of a workgroup split the stream's block of
	// TPW*WPS tiles (16 frames each) along time and meet at one LDS-only barrier per layer.  4 workgroups per CU
	// (1024 streams = 4 per CU) need 4 waves per SIMD, i.e. <= 128 registers per lane.
	// Per stage the wave first issues the loads whose latency it can hide -- the NEXT stage's weight fragments --
	// BEFORE its own ring stores: gfx950 has one vmcnt for loads and stores, so a load issued after a store would
	// otherwise wait for that store's acknowledgement.
	// dynamic LDS: xbuf[2][NTB*64] float4 | quad table
	template <int TPW, int WPS>
	__global__ void __launch_bounds__(64 * WPS) __attribute__((amdgpu_waves_per_eu(WPS >= 8 ? 8 : (WPS >= 4 ? 4 : (WPS == 2 ? 2 : 1)), WPS >= 8 ? 8 : (WPS >= 4 ? 4 : (WPS == 2 ? 2 : 1))))) WaveNetBlockKernel(
		const WnStage* __restrict__ stages, const float* __restrict__ wpack, const WnQuad* __restrict__ qdesc,
		const int* __restrict__ ringFrames, int nstages, int nqdesc, int wpackF4, int nrings, int stateF4, float headScale,
		f32x4* __restrict__ state, const int* __restrict__ slots, const int* __restrict__ rows, const float* __restrict__ in,
		float* __restrict__ out, long inStride, long outStride, int n, long long* __restrict__ trace, int traceBlock)
	{
		constexpr int NTB = TPW * WPS;
		extern __shared__ __attribute__((aligned(16))) char smem[];
		f32x4* xbuf = reinterpret_cast<f32x4*>(smem);                           // [2][NTB*64]
		WnQuad* sQ = reinterpret_cast<WnQuad*>(smem + 2 * NTB * 64 * 16);       // [nqdesc]

		const int lane = threadIdx.x & 63;
		const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform by construction: keep it in an SGPR
		const int tb = wave * TPW; // first block tile owned by this wave
		const int g = lane >> 4;
		const int j = lane & 15;

		{
			const f32x4* qsrc = reinterpret_cast<const f32x4*>(qdesc);
			f32x4* qdst = reinterpret_cast<f32x4*>(sQ);
			for (int i = threadIdx.x; i < nqdesc; i += 64 * WPS) qdst[i] = qsrc[i];
		}

		const int slot = slots[blockIdx.x];
		const int row = rows[blockIdx.x];
		f32x4* st = state + (size_t)slot * (size_t)stateF4;
		int* header = reinterpret_cast<int*>(st);
		const int myPos = header[lane]; // lane r holds the write cursor of ring r
		const __amdgpu_buffer_rsrc_t wrsrc = MakeRsrc(wpack, (unsigned)wpackF4 * 16u);
		const __amdgpu_buffer_rsrc_t srsrc = MakeRsrc(st, (unsigned)stateF4 * 16u);
		const float* inRow = in + (size_t)row * inStride;
		float* outRow = out + (size_t)row * outStride;

		float cond[TPW];
		f32x4 xcur[TPW];
		f32x4 head[TPW];
#pragma unroll
		for (int t = 0; t < TPW; t++)
		{
			const int f = (tb + t) * 16 + j;
			cond[t] = (f < n) ? inRow[f] : 0.0f; // WaveNet.h:770 (input -> condition)
			xcur[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
			head[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; // WaveNet.h:772 headArray.SetZero()
		}

		BlockBarrier<WPS>();

		int cur = 0;
		WnStage sd = LoadStage(stages, 0);
		f32x4 apf[WN_APF];
#pragma unroll
		for (int r = 0; r < WN_APF; r++) apf[r] = BufLoad(wrsrc, lane * 16, (sd.wconv_off + r * 64) * 16);

		for (int s = 0; s < nstages; s++)
		{
			NA_TRACE(0);
			WnStage sdn = sd;
			if (s + 1 < nstages) sdn = LoadStage(stages, s + 1);

			const int outPos0 = (sd.out_ring_id >= 0) ? __builtin_amdgcn_readlane(myPos, sd.out_ring_id) : 0;
			const int inPos0 = (sd.ring_id >= 0) ? __builtin_amdgcn_readlane(myPos, sd.ring_id) : 0;
			f32x4* xbCur = xbuf + cur * (NTB * 64);
			f32x4* xbNext = xbuf + (cur ^ 1) * (NTB * 64);
			const int vecOff = (sd.vec_off + g) * 16; // [0..3] conv/dense bias, [4..7] mix-in w, [8..11] 1x1 bias, [12..15] aux

			if (sd.type == WN_ST_LAYER)
			{
				// this stage's small operands: needed only after the conv, their L2 latency hides behind it
				const f32x4 w1 = BufLoad(wrsrc, lane * 16, sd.w1_off * 16);
				const f32x4 bias4 = BufLoad(wrsrc, vecOff, 0);
				const f32x4 wm4 = BufLoad(wrsrc, vecOff, 4 * 16);
				const f32x4 b14 = BufLoad(wrsrc, vecOff, 8 * 16);

				f32x4 acc[TPW];
#pragma unroll
				for (int t = 0; t < TPW; t++) acc[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				// WaveNet.h:468
				if (sd.G == 4) ConvFast<4, TPW>(acc, apf, sd, wrsrc, xbCur, srsrc, inPos0, tb, lane, g, j);
				else if (sd.G == 2) ConvFast<2, TPW>(acc, apf, sd, wrsrc, xbCur, srsrc, inPos0, tb, lane, g, j);
				else if (sd.G == 1) ConvFast<1, TPW>(acc, apf, sd, wrsrc, xbCur, srsrc, inPos0, tb, lane, g, j);
				else ConvRounds<TPW>(acc, apf, sd, sQ, wrsrc, xbCur, srsrc, inPos0, tb, lane, g, j);
				NA_TRACE(1);

				// next stage's weight fragments: issued BEFORE this stage's ring stores (see kernel comment)
#pragma unroll
				for (int r = 0; r < WN_APF; r++) apf[r] = BufLoad(wrsrc, lane * 16, (sdn.wconv_off + r * 64) * 16);

				const bool leaky = (sd.flags & WN_FLAG_LEAKY) != 0;
				f32x4 z[TPW];
#pragma unroll
				for (int t = 0; t < TPW; t++)
				{
					// + conv bias (:288-289) + W_mix * cond (:471), then activation (:473-480)
					z[t] = Activate(acc[t] + (bias4 + wm4 * cond[t]), leaky);
					head[t] += z[t]; // :482
				}
				if (sd.flags & WN_FLAG_NEED_OUTPUT)
				{
					// 1x1 + bias + residual (:486-491); z in D layout is already a B fragment
#pragma unroll
					for (int t = 0; t < TPW; t++) xcur[t] += b14;
					MfmaRound<TPW>(xcur, w1, z);
				}
				if (sd.flags & WN_FLAG_PUBLISH)
				{
					Publish<TPW>(xcur, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
					cur ^= 1;
				}
			}
			else
			{
#pragma unroll
				for (int r = 0; r < WN_APF; r++) apf[r] = BufLoad(wrsrc, lane * 16, (sdn.wconv_off + r * 64) * 16);
				if (sd.type == WN_ST_RECHANNEL_COND)
				{
					const f32x4 wre4 = BufLoad(wrsrc, vecOff, 12 * 16);
#pragma unroll
					for (int t = 0; t < TPW; t++) xcur[t] = wre4 * cond[t]; // :637 with InputSize == 1
					Publish<TPW>(xcur, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
					cur ^= 1;
				}
				else if (sd.type == WN_ST_ARRAY_LINK)
				{
					// previous array's headRechannel (K=1, :658-660) and this array's rechannel (:637)
					const f32x4 w1 = BufLoad(wrsrc, lane * 16, sd.w1_off * 16);
					const f32x4 w2 = BufLoad(wrsrc, lane * 16, sd.w2_off * 16);
					f32x4 hb = { 0.0f, 0.0f, 0.0f, 0.0f };
					if (sd.flags & WN_FLAG_BIAS) hb = BufLoad(wrsrc, vecOff, 0);
					f32x4 hnew[TPW], xnew[TPW];
#pragma unroll
					for (int t = 0; t < TPW; t++)
					{
						hnew[t] = hb;
						xnew[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
					}
					MfmaRound<TPW>(hnew, w1, head);
					MfmaRound<TPW>(xnew, w2, xcur);
#pragma unroll
					for (int t = 0; t < TPW; t++)
					{
						head[t] = hnew[t];
						xcur[t] = xnew[t];
					}
					Publish<TPW>(xcur, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
					cur ^= 1;
				}
				else if (sd.type == WN_ST_HEAD_DENSE_OUT)
				{
					const f32x4 w1 = BufLoad(wrsrc, lane * 16, sd.w1_off * 16);
					f32x4 hb = { 0.0f, 0.0f, 0.0f, 0.0f };
					if (sd.flags & WN_FLAG_BIAS) hb = BufLoad(wrsrc, vecOff, 0);
					f32x4 o[TPW];
#pragma unroll
					for (int t = 0; t < TPW; t++) o[t] = hb;
					MfmaRound<TPW>(o, w1, head);
#pragma unroll
					for (int t = 0; t < TPW; t++)
					{
						const int f = (tb + t) * 16 + j;
						if (g == 0 && f < n) outRow[f] = headScale * o[t].x; // :793-798
					}
				}
				else // WN_ST_HEAD_CONV_OUT
				{
					// A2 head: Conv1D(C -> 1, K = 16) over the accumulated head signal (:658-660)
					f32x4 wh[WN_APF];
#pragma unroll
					for (int r = 0; r < WN_APF; r++) wh[r] = BufLoad(wrsrc, lane * 16, (sd.wconv_off + r * 64) * 16);
					f32x4 hb = { 0.0f, 0.0f, 0.0f, 0.0f };
					if (sd.flags & WN_FLAG_BIAS) hb = BufLoad(wrsrc, vecOff, 0);
					Publish<TPW>(head, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
					cur ^= 1;
					BlockBarrier<WPS>();
					f32x4 acc[TPW];
#pragma unroll
					for (int t = 0; t < TPW; t++) acc[t] = hb;
					ConvRounds<TPW>(acc, wh, sd, sQ, wrsrc, xbNext, srsrc, inPos0, tb, lane, g, j);
#pragma unroll
					for (int t = 0; t < TPW; t++)
					{
						const int f = (tb + t) * 16 + j;
						if (g == 0 && f < n) outRow[f] = headScale * acc[t].x;
					}
				}
			}
			NA_TRACE(2);
			BlockBarrier<WPS>();
			NA_TRACE(3);
			sd = sdn;
		}

		// advance every ring cursor by n (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
		if (wave == 0 && lane < nrings)
		{
			const int R = ringFrames[lane];
			int p = myPos + n;
			if (p >= R) p -= R;
			header[lane] = p;
		}
	}

	// ------------------------------------------------------------------------------------------
	// Prewarm (WaveNet.h:746-766): zero-input steady state.  Every layer input is a constant column
	// that depends only on the weights, so it is computed once per MODEL by one wave (lane = channel),
	// in the reference's natural weight layout, then broadcast into every stream's rings.
	// ------------------------------------------------------------------------------------------
	__global__ void __launch_bounds__(64) WaveNetPrewarmColumnsKernel(const WnPrewarmLayer* __restrict__ layers, int numLayers,
		const float* __restrict__ w, float* __restrict__ cols /* [ring][16] */)
	{
		__shared__ float x[16], z[16], head[16], lin[16];
		const int i = threadIdx.x;
		if (i < 16)
		{
			x[i] = 0.0f; z[i] = 0.0f; head[i] = 0.0f; lin[i] = 0.0f; // condition = 0 (:748), headArray zero (:750)
		}
		__syncthreads();

		for (int li = 0; li < numLayers; li++)
		{
			const WnPrewarmLayer L = layers[li];
			if (L.kind == 0)
			{
				if (L.rechannel >= 0)
				{
					// rechannel.Process (:609): x = W_re * layer_inputs
					float v = 0.0f;
					if (i < L.cin)
						for (int c = 0; c < L.rech_in; c++) v += w[L.rechannel + i * L.rech_in + c] * lin[c];
					__syncthreads();
					if (i < 16) x[i] = (i < L.cin) ? v : 0.0f;
					__syncthreads();
				}
				if (i < 16) cols[L.ring_id * 16 + i] = x[i]; // CopyBuffer (:74-82): the whole receptive field holds this column

				float acc = 0.0f;
				if (i < L.cout)
				{
					for (int k = 0; k < L.ksize; k++)
						for (int c = 0; c < L.cin; c++) acc += w[L.wconv + (i * L.cin + c) * L.ksize + k] * x[c];
					acc += w[L.bconv + i];
					// mix-in * condition(0) adds nothing
					acc = (L.act == 1) ? LeakyReLU(acc) : (L.act == 2 ? (1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc * 2.885390081777927f) + 1.0f)) : FastTanh(acc));
				}
				__syncthreads();
				if (i < 16)
				{
					z[i] = (i < L.cout) ? acc : 0.0f;
					head[i] += z[i];
				}
				__syncthreads();
				float y = 0.0f;
				if (i < L.cout)
				{
					for (int c = 0; c < L.cin; c++) y += w[L.w1 + i * L.cin + c] * z[c];
					y += w[L.b1 + i];
					y += x[i];
				}
				__syncthreads();
				if (i < 16)
				{
					if (L.last_of_array) lin[i] = (i < L.cout) ? y : 0.0f; // arrayOutputs feeds the next array's rechannel
					else x[i] = (i < L.cout) ? y : 0.0f;
				}
				__syncthreads();
			}
			else
			{
				// head rechannel (:625-629): steady-state head column, then conv over a constant history
				if (L.ring_id >= 0 && i < 16) cols[L.ring_id * 16 + i] = (i < L.cin) ? head[i] : 0.0f;
				float acc = 0.0f;
				if (i < L.cout)
				{
					for (int k = 0; k < L.ksize; k++)
						for (int c = 0; c < L.cin; c++) acc += w[L.wconv + (i * L.cin + c) * L.ksize + k] * head[c];
					if (L.bconv >= 0) acc += w[L.bconv + i];
				}
				__syncthreads();
				if (i < 16) head[i] = (i < L.cout) ? acc : 0.0f; // becomes the next array's head accumulator (:785-789)
				__syncthreads();
			}
		}
	}

	// float quad -> split quad [h0 h1 | h2 h3 | l0 l1 | l2 l3] (h = f16(v), l = f16(v - h)): the storage format of the f16-split kernel
	__device__ __forceinline__ f32x4 SplitQuadBits(f32x4 v)
	{
		typedef _Float16 h2 __attribute__((ext_vector_type(2)));
		typedef float f2 __attribute__((ext_vector_type(2)));
		const h2 h01 = __builtin_convertvector(f2{ v.x, v.y }, h2), h23 = __builtin_convertvector(f2{ v.z, v.w }, h2);
		const h2 l01 = __builtin_convertvector(f2{ v.x - (float)h01.x, v.y - (float)h01.y }, h2);
		const h2 l23 = __builtin_convertvector(f2{ v.z - (float)h23.x, v.w - (float)h23.y }, h2);
		return f32x4{ __builtin_bit_cast(float, h01), __builtin_bit_cast(float, h23), __builtin_bit_cast(float, l01), __builtin_bit_cast(float, l23) };
	}

	// grid = (streams to fill, rings), block = 256: fill ring r of stream slot with its steady-state column.
	// split == 0: f32 quads in the tile layout (frame kernel); split == 1: split quads, frame-major rings (f16-split kernel).
	__global__ void __launch_bounds__(256) WaveNetFillRingsKernel(f32x4* __restrict__ state, int stateF4, const int* __restrict__ slots,
		const int* __restrict__ ringOffF4, const int* __restrict__ ringFrames, const int* __restrict__ ringG, const float* __restrict__ cols, int split)
	{
		const int slot = slots[blockIdx.x];
		const int r = blockIdx.y;
		f32x4* st = state + (size_t)slot * (size_t)stateF4;
		const int G = ringG[r];
		const int nF4 = (ringFrames[r] / 16) * G * 16;
		f32x4* ring = st + ringOffF4[r];
		for (int idx = threadIdx.x; idx < nF4; idx += blockDim.x)
		{
			const int cg = split ? (idx % G) : ((idx >> 4) % G);
			const float* c = cols + r * 16 + cg * 4;
			const f32x4 v = f32x4{ c[0], c[1], c[2], c[3] };
			ring[idx] = split ? SplitQuadBits(v) : v;
		}
		if (r == 0 && threadIdx.x < WN_MAX_RINGS) reinterpret_cast<int*>(st)[threadIdx.x] = 0; // cursors
	}

	// ------------------------------------------------------------------------------------------ launchers

	static long long* g_traceBuffer = nullptr;
	void SetWaveNetTraceBuffer(long long* p) { g_traceBuffer = p; }
	long long* GetWaveNetTraceBuffer() { return g_traceBuffer; }

	template <int TPW, int WPS>
	static hipError_t LaunchBlock(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in,
		float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		const size_t lds = (size_t)2 * TPW * WPS * 64 * 16 + (size_t)m.nqdesc * sizeof(WnQuad);
		if (lds > 64 * 1024) return hipErrorInvalidValue;
		hipLaunchKernelGGL((WaveNetBlockKernel<TPW, WPS>), dim3((unsigned)numStreams), dim3(64 * WPS), lds, stream, m.stages, m.wpack, m.qdesc,
			m.ring_frames, m.nstages, m.nqdesc, m.wpack_f4, m.nrings, m.state_f4, m.head_scale, reinterpret_cast<f32x4*>(state), slots, rows,
			in, out, inStride, outStride, n, g_traceBuffer,
			[]() { const char* e = getenv("NA_TRACE_BLOCK"); return e ? atoi(e) : 0; }());
		return hipGetLastError();
	}

	hipError_t LaunchWaveNetBlock(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in,
		float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		if (n > WN_MAX_FRAMES) return hipErrorInvalidValue;
		// smallest tile grid that covers n frames: (tiles per wave) x (waves per stream)
		if (n > 64)
		{
			static const int cfg = []() { const char* e = getenv("NA_WN_CFG"); return e ? atoi(e) : 24; }(); // tuning knob: TPW WPS
			if (cfg == 18) return LaunchBlock<1, 8>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
			if (cfg == 42) return LaunchBlock<4, 2>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
			if (cfg == 81) return LaunchBlock<8, 1>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
			return LaunchBlock<2, 4>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		}
		if (n > 32) return LaunchBlock<1, 4>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		if (n > 16) return LaunchBlock<1, 2>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		return LaunchBlock<1, 1>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
	}

	hipError_t LaunchWaveNetPrewarmColumns(const WnPrewarmLayer* layers, int numLayers, const float* weights, float* cols,
		hipStream_t stream)
	{
		hipLaunchKernelGGL(WaveNetPrewarmColumnsKernel, dim3(1), dim3(64), 0, stream, layers, numLayers, weights, cols);
		return hipGetLastError();
	}

	hipError_t LaunchWaveNetFillRings(float* state, int stateF4, const int* slots, int numStreams, int numRings, const int* ringOffF4,
		const int* ringFrames, const int* ringG, const float* cols, hipStream_t stream, bool splitFormat)
	{
		if (numStreams <= 0) return hipSuccess;
		hipLaunchKernelGGL(WaveNetFillRingsKernel, dim3((unsigned)numStreams, (unsigned)numRings), dim3(256), 0, stream,
			reinterpret_cast<f32x4*>(state), stateF4, slots, ringOffF4, ringFrames, ringG, cols, splitFormat ? 1 : 0);
		return hipGetLastError();
	}
}
