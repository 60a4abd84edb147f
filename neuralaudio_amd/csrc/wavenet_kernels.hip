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
		c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
		c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
		c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
		c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
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
	template <int TPW, int WPS>
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
				BufStore(srsrc, x[t], (f < n && f >= firstKept) ? voff : WN_OOB);
				p += 16;
				if (p >= R) p -= R;
			}
		}
		BlockBarrier<WPS>();
	}

	// acc += conv(x) for this wave's tiles (Conv1DT::Process, WaveNet.h:139-290).
	// B fragment of (round, tile): frame off = 16*(tb+t) + j - shift of channel group cg, per lane group.
	// The source is classified per (round, tile) with SCALAR compares on the min/max shift of the four lane groups:
	//   whole tile inside the current block -> one LDS read;  whole tile in the past -> one ring read from HBM;
	//   straddling the block start -> both + select.
	template <int TPW>
	__device__ __forceinline__ void ConvRounds(f32x4 (&acc)[TPW], const WnStage& sd, const WnQuad* sQ, __amdgpu_buffer_rsrc_t wrsrc,
		const f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int pos0, int tb, int lane, int g, int j)
	{
		const int G = sd.G;
		const int R = sd.ring_frames;
		const int step = G * 16;
		for (int r = 0; r < sd.nrounds; r++)
		{
			const f32x4 a = BufLoad(wrsrc, lane * 16, (sd.wconv_off + r * 64) * 16);
			const WnQuad qd = sQ[sd.qdesc_off + r * 4 + g];
			const int s0 = __builtin_amdgcn_readlane(qd.shift, 0), s1 = __builtin_amdgcn_readlane(qd.shift, 16);
			const int s2 = __builtin_amdgcn_readlane(qd.shift, 32), s3 = __builtin_amdgcn_readlane(qd.shift, 48);
			const int smin = min(min(s0, s1), min(s2, s3));
			const int smax = max(max(s0, s1), max(s2, s3));
			const int off0 = tb * 16 + j - qd.shift;
			const int idx0 = ((off0 >> 4) * G + qd.cg) * 16 + (off0 & 15);
			int p0 = pos0 + off0;
			if (p0 < 0) p0 += R;
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
					int p = p0 + t * 16;
					if (p >= R) p -= R;
					const int voff = (sd.ring_off + ((p >> 4) * G + qd.cg) * 16 + (p & 15)) * 16;
					if (base + 15 - smin < 0)
					{
						b[t] = BufLoad(srsrc, voff, 0);
					}
					else
					{
						const int off = off0 + t * 16;
						const int idx = idx0 + t * step;
						const f32x4 l = xb[idx < 0 ? 0 : idx];
						const f32x4 h = BufLoad(srsrc, off < 0 ? voff : WN_OOB, 0);
						b[t] = (off < 0) ? h : l;
					}
				}
			}
#pragma unroll
			for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
			for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
			for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
			for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
		}
	}

	constexpr int WN_STAGE_INTS = (int)(sizeof(WnStage) / sizeof(int));

	__device__ __forceinline__ WnStage LoadStageLds(const int* sStages, int s)
	{
		// the stage table is wave-uniform: pin every field into an SGPR so control flow stays scalar
		WnStage sd;
		int* dst = reinterpret_cast<int*>(&sd);
#pragma unroll
		for (int i = 0; i < WN_STAGE_INTS; i++) dst[i] = __builtin_amdgcn_readfirstlane(sStages[s * WN_STAGE_INTS + i]);
		return sd;
	}

	// grid = active streams of one model; block = WPS waves: the WPS waves of a workgroup split the stream's block
	// of TPW*WPS tiles (16 frames each) along time and meet at one LDS barrier per layer.
	// dynamic LDS: xbuf[2][NTB*64] float4 | stage table | quad table
	template <int TPW, int WPS>
	__global__ void __launch_bounds__(64 * WPS) WaveNetBlockKernel(const WnStage* __restrict__ stages, const float* __restrict__ wpack,
		const WnQuad* __restrict__ qdesc, const int* __restrict__ ringFrames, int nstages, int nqdesc, int wpackF4, int nrings, int stateF4,
		float headScale, f32x4* __restrict__ state, const int* __restrict__ slots, const int* __restrict__ rows,
		const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride, int n)
	{
		constexpr int NTB = TPW * WPS;
		extern __shared__ __attribute__((aligned(16))) char smem[];
		f32x4* xbuf = reinterpret_cast<f32x4*>(smem);                       // [2][NTB*64]
		int* sStages = reinterpret_cast<int*>(smem + 2 * NTB * 64 * 16);    // [nstages][WN_STAGE_INTS]
		WnQuad* sQ = reinterpret_cast<WnQuad*>(sStages + ((nstages * WN_STAGE_INTS + 3) & ~3)); // [nqdesc]

		const int lane = threadIdx.x & 63;
		const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform by construction: keep it in an SGPR
		const int tb = wave * TPW; // first block tile owned by this wave
		const int g = lane >> 4;
		const int j = lane & 15;

		// model tables -> LDS (a few KB, L2-resident): takes their latency off every layer's critical path
		{
			const int* src = reinterpret_cast<const int*>(stages);
			for (int i = threadIdx.x; i < nstages * WN_STAGE_INTS; i += 64 * WPS) sStages[i] = src[i];
			const f32x4* qsrc = reinterpret_cast<const f32x4*>(qdesc);
			f32x4* qdst = reinterpret_cast<f32x4*>(sQ);
			for (int i = threadIdx.x; i < nqdesc; i += 64 * WPS) qdst[i] = qsrc[i];
		}

		const int slot = slots[blockIdx.x];
		const int row = rows[blockIdx.x];
		f32x4* st = state + (size_t)slot * (size_t)stateF4;
		int* header = reinterpret_cast<int*>(st);
		const int myPos = header[lane]; // lane r holds the write cursor of ring r
		const f32x4* wp = reinterpret_cast<const f32x4*>(wpack);
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

		if (WPS > 1) __syncthreads();
		else __builtin_amdgcn_wave_barrier();

		int cur = 0;

		for (int s = 0; s < nstages; s++)
		{
			const WnStage sd = LoadStageLds(sStages, s);
			const int outPos0 = (sd.out_ring_id >= 0) ? __builtin_amdgcn_readlane(myPos, sd.out_ring_id) : 0;
			const int inPos0 = (sd.ring_id >= 0) ? __builtin_amdgcn_readlane(myPos, sd.ring_id) : 0;
			f32x4* xbCur = xbuf + cur * (NTB * 64);
			f32x4* xbNext = xbuf + (cur ^ 1) * (NTB * 64);

			if (sd.type == WN_ST_LAYER)
			{
				const f32x4 w1 = wp[sd.w1_off + lane];
				const f32x4 bias4 = wp[sd.vec_off + g];
				const f32x4 wm4 = wp[sd.vec_off + 4 + g];
				const f32x4 b14 = wp[sd.vec_off + 8 + g];

				f32x4 acc[TPW];
#pragma unroll
				for (int t = 0; t < TPW; t++) acc[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				ConvRounds<TPW>(acc, sd, sQ, wrsrc, xbCur, srsrc, inPos0, tb, lane, g, j); // WaveNet.h:468

				const bool leaky = (sd.flags & WN_FLAG_LEAKY) != 0;
				const bool needOutput = (sd.flags & WN_FLAG_NEED_OUTPUT) != 0;
#pragma unroll
				for (int t = 0; t < TPW; t++)
				{
					// + conv bias (:288-289) + W_mix * cond (:471), then activation (:473-480)
					const f32x4 z = Activate(acc[t] + (bias4 + wm4 * cond[t]), leaky);
					head[t] += z; // :482
					// 1x1 + bias + residual (:486-491); z in D layout is already a B fragment
					if (needOutput) xcur[t] = Mfma4(w1, z, xcur[t] + b14);
				}

				if (sd.flags & WN_FLAG_PUBLISH)
				{
					Publish<TPW, WPS>(xcur, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
					cur ^= 1;
				}
			}
			else if (sd.type == WN_ST_RECHANNEL_COND)
			{
				const f32x4 wre4 = wp[sd.vec_off + 12 + g];
#pragma unroll
				for (int t = 0; t < TPW; t++) xcur[t] = wre4 * cond[t]; // :637 with InputSize == 1
				Publish<TPW, WPS>(xcur, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
				cur ^= 1;
			}
			else if (sd.type == WN_ST_ARRAY_LINK)
			{
				// previous array's headRechannel (K=1, :658-660) and this array's rechannel (:637)
				const f32x4 w1 = wp[sd.w1_off + lane];
				const f32x4 w2 = wp[sd.w2_off + lane];
				f32x4 hb = { 0.0f, 0.0f, 0.0f, 0.0f };
				if (sd.flags & WN_FLAG_BIAS) hb = wp[sd.vec_off + g];
#pragma unroll
				for (int t = 0; t < TPW; t++)
				{
					head[t] = Mfma4(w1, head[t], hb);
					xcur[t] = Mfma4(w2, xcur[t], f32x4{ 0.0f, 0.0f, 0.0f, 0.0f });
				}
				Publish<TPW, WPS>(xcur, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
				cur ^= 1;
			}
			else if (sd.type == WN_ST_HEAD_DENSE_OUT)
			{
				const f32x4 w1 = wp[sd.w1_off + lane];
				f32x4 hb = { 0.0f, 0.0f, 0.0f, 0.0f };
				if (sd.flags & WN_FLAG_BIAS) hb = wp[sd.vec_off + g];
#pragma unroll
				for (int t = 0; t < TPW; t++)
				{
					const f32x4 o = Mfma4(w1, head[t], hb);
					const int f = (tb + t) * 16 + j;
					if (g == 0 && f < n) outRow[f] = headScale * o.x; // :793-798
				}
			}
			else // WN_ST_HEAD_CONV_OUT
			{
				// A2 head: Conv1D(C -> 1, K = 16) over the accumulated head signal (:658-660)
				Publish<TPW, WPS>(head, xbNext, srsrc, sd.out_ring_off, sd.out_G, outPos0, sd.out_ring_frames, n, tb, g, j);
				cur ^= 1;
				f32x4 hb = { 0.0f, 0.0f, 0.0f, 0.0f };
				if (sd.flags & WN_FLAG_BIAS) hb = wp[sd.vec_off + g];
				f32x4 acc[TPW];
#pragma unroll
				for (int t = 0; t < TPW; t++) acc[t] = hb;
				ConvRounds<TPW>(acc, sd, sQ, wrsrc, xbNext, srsrc, inPos0, tb, lane, g, j);
#pragma unroll
				for (int t = 0; t < TPW; t++)
				{
					const int f = (tb + t) * 16 + j;
					if (g == 0 && f < n) outRow[f] = headScale * acc[t].x;
				}
			}
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
					acc = (L.act == 1) ? LeakyReLU(acc) : FastTanh(acc);
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

	// grid = (streams to fill, rings), block = 256: fill ring r of stream slot with its steady-state column
	__global__ void __launch_bounds__(256) WaveNetFillRingsKernel(f32x4* __restrict__ state, int stateF4, const int* __restrict__ slots,
		const int* __restrict__ ringOffF4, const int* __restrict__ ringFrames, const int* __restrict__ ringG, const float* __restrict__ cols)
	{
		const int slot = slots[blockIdx.x];
		const int r = blockIdx.y;
		f32x4* st = state + (size_t)slot * (size_t)stateF4;
		const int G = ringG[r];
		const int nF4 = (ringFrames[r] / 16) * G * 16;
		f32x4* ring = st + ringOffF4[r];
		for (int idx = threadIdx.x; idx < nF4; idx += blockDim.x)
		{
			const int cg = (idx >> 4) % G;
			const float* c = cols + r * 16 + cg * 4;
			ring[idx] = f32x4{ c[0], c[1], c[2], c[3] };
		}
		if (r == 0 && threadIdx.x < WN_MAX_RINGS) reinterpret_cast<int*>(st)[threadIdx.x] = 0; // cursors
	}

	// ------------------------------------------------------------------------------------------ launchers

	template <int TPW, int WPS>
	static void LaunchBlock(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in, float* out,
		long inStride, long outStride, int n, hipStream_t stream)
	{
		const size_t lds = (size_t)2 * TPW * WPS * 64 * 16 + (size_t)((m.nstages * WN_STAGE_INTS + 3) & ~3) * sizeof(int) +
			(size_t)m.nqdesc * sizeof(WnQuad);
		hipLaunchKernelGGL((WaveNetBlockKernel<TPW, WPS>), dim3((unsigned)numStreams), dim3(64 * WPS), lds, stream, m.stages, m.wpack, m.qdesc,
			m.ring_frames, m.nstages, m.nqdesc, m.wpack_f4, m.nrings, m.state_f4, m.head_scale, reinterpret_cast<f32x4*>(state), slots, rows, in, out,
			inStride, outStride, n);
	}

	hipError_t LaunchWaveNetBlock(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in,
		float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		if (n > WN_MAX_FRAMES) return hipErrorInvalidValue;
		// smallest tile grid that covers n frames: (tiles per wave) x (waves per stream)
		if (n > 64) LaunchBlock<2, 4>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		else if (n > 32) LaunchBlock<1, 4>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		else if (n > 16) LaunchBlock<1, 2>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		else LaunchBlock<1, 1>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		return hipGetLastError();
	}

	hipError_t LaunchWaveNetPrewarmColumns(const WnPrewarmLayer* layers, int numLayers, const float* weights, float* cols,
		hipStream_t stream)
	{
		hipLaunchKernelGGL(WaveNetPrewarmColumnsKernel, dim3(1), dim3(64), 0, stream, layers, numLayers, weights, cols);
		return hipGetLastError();
	}

	hipError_t LaunchWaveNetFillRings(float* state, int stateF4, const int* slots, int numStreams, int numRings, const int* ringOffF4,
		const int* ringFrames, const int* ringG, const float* cols, hipStream_t stream)
	{
		if (numStreams <= 0) return hipSuccess;
		hipLaunchKernelGGL(WaveNetFillRingsKernel, dim3((unsigned)numStreams, (unsigned)numRings), dim3(256), 0, stream,
			reinterpret_cast<f32x4*>(state), stateF4, slots, ringOffF4, ringFrames, ringG, cols);
		return hipGetLastError();
	}
}
