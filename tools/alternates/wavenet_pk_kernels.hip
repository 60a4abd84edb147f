// wavenet_pk_kernels.hip -- the packed-FMA ("lane = frame") WaveNet block kernel for gfx950.
//
// Same path, same HBM/LDS data layout and same stage program as wavenet_kernels.hip (reference functions:
// WaveNetModelT/LayerArrayT/LayerT::Process, Conv1DT::Process, DenseLayerT::Process -- NeuralAudio/WaveNet.h:768-799,
// 632-661,462-494,139-290,336-383; FastMath -- NeuralAudio/Activation.h:83-118), different mapping of the arithmetic:
//
//   * lane = one audio frame, a wave = 64 consecutive frames, a workgroup = the 1-2 waves of one stream's block;
//   * every lane holds ALL channels of its frame in registers; mat-muls are chains of v_pk_fma_f32 producing two
//     output channels at a time, the weight pair coming straight from SGPRs (s_load_dwordx16 through the constant
//     address space), the input channel broadcast from one VGPR (op_sel).
//
// Why not MFMA here: measured on MI355X (tools/microbench/mfma_valu_overlap.hip) v_mfma_f32_16x16x4_f32 runs at the
// f32 VALU rate AND does not overlap with VALU work of other waves on the same SIMD -- it is the same datapath.  The MFMA
// mapping pays for padding (8-channel layers fill half of the 16-row tile; activations run on padded lanes), this
// mapping issues exactly the useful MACs and the activation sees dense data.
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

namespace na
{
	namespace pk
	{
		typedef float f32x2 __attribute__((ext_vector_type(2)));
		typedef float f32x4 __attribute__((ext_vector_type(4)));
		typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
		typedef const float __attribute__((address_space(4)))* CFloat; // wave-uniform read-only data -> scalar loads
		typedef const int __attribute__((address_space(4)))* CInt;

		constexpr int OOB = (int)0x80000000;
		constexpr int STAGE_INTS = (int)(sizeof(WnStage) / sizeof(int));
		constexpr int MAXC = 16;

		__device__ __forceinline__ __amdgpu_buffer_rsrc_t MakeRsrc(const void* base, unsigned bytes)
		{
			return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
		}

		__device__ __forceinline__ f32x4 BufLoad(__amdgpu_buffer_rsrc_t r, int voff)
		{
			return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
		}

		__device__ __forceinline__ void BufStore(__amdgpu_buffer_rsrc_t r, f32x4 v, int voff)
		{
			__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
		}

		__device__ __forceinline__ WnStage LoadStage(const WnStage* __restrict__ stages, int s)
		{
			WnStage sd;
			CInt src = (CInt)(const int*)(stages + s);
			int* dst = reinterpret_cast<int*>(&sd);
#pragma unroll
			for (int i = 0; i < STAGE_INTS; i++) dst[i] = src[i];
			return sd;
		}

		__device__ __forceinline__ f32x2 Abs2(f32x2 v)
		{
			f32x2 r;
			r.x = __builtin_fabsf(v.x);
			r.y = __builtin_fabsf(v.y);
			return r;
		}

		// Activation.h:83-91 on two channels: same association, packed math, division = num * v_rcp_f32(den)
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

		// Activation.h:110-118
		__device__ __forceinline__ f32x2 LeakyReLU2(f32x2 v)
		{
			f32x2 r;
			r.x = v.x > 0.0f ? v.x : 0.01f * v.x;
			r.y = v.y > 0.0f ? v.y : 0.01f * v.y;
			return r;
		}

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

		// float4 index of (frame, channel group) in a tiled image with G groups: ((frame>>4)*G + cg)*16 + (frame&15)
		__device__ __forceinline__ int TileIdx(int frame, int G, int cg) { return ((frame >> 4) * G + cg) * 16 + (frame & 15); }

		// Channels [4*cg, 4*cg+4) of the frame `off` frames from the block start (off < 0: history) for this lane.
		// lo/hi: range of `off` over the wave (scalar) -> whole wave in block / whole wave in history / mixed.
		template <int G>
		__device__ __forceinline__ void FetchFrame(float (&x)[4 * G], const f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int off, int lo, int hi,
			int pos0, int R)
		{
			int p = pos0 + off;
			if (p < 0) p += R;
			if (p >= R) p -= R;
#pragma unroll
			for (int cg = 0; cg < G; cg++)
			{
				f32x4 v;
				if (lo >= 0)
				{
					v = xb[TileIdx(off, G, cg)];
				}
				else
				{
					const int voff = (ringOff + TileIdx(p, G, cg)) * 16;
					if (hi < 0)
					{
						v = BufLoad(srsrc, voff);
					}
					else
					{
						const f32x4 l = xb[TileIdx(off < 0 ? 0 : off, G, cg)];
						const f32x4 h = BufLoad(srsrc, off < 0 ? voff : OOB);
						v = (off < 0) ? h : l;
					}
				}
				x[4 * cg] = v.x; x[4 * cg + 1] = v.y; x[4 * cg + 2] = v.z; x[4 * cg + 3] = v.w;
			}
		}

		// acc[o] += sum_c w[c][o] * x[c]   (w: [CIN][COUT] floats, scalar loads; two output channels per v_pk_fma_f32)
		template <int CIN, int COUT>
		__device__ __forceinline__ void DensePk(f32x2 (&acc)[COUT / 2], CFloat w, const float (&x)[CIN])
		{
#pragma unroll
			for (int c = 0; c < CIN; c++)
			{
#pragma unroll
				for (int o = 0; o < COUT / 2; o++)
				{
#ifdef NA_PK_NOWEIGHTS
					const f32x2 wp = f32x2{ 0.001f * (c + 1), 0.002f * (o + 1) }; // ablation: no scalar weight loads (results wrong)
#else
					const f32x2 wp = f32x2{ w[c * COUT + 2 * o], w[c * COUT + 2 * o + 1] };
#endif
					acc[o] = __builtin_elementwise_fma(wp, f32x2{ x[c], x[c] }, acc[o]);
				}
			}
		}

		// this lane's frame of a layer output -> LDS block image (in-block taps of the next layer) and the next layer's
		// HBM ring (history for LATER blocks: only the last R-128 frames of a block can ever be read back)
		template <int G>
		__device__ __forceinline__ void PublishFrame(const float (&x)[MAXC], f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int pos0, int R,
			int n, int f)
		{
			const int firstKept = n - (R - WN_MAX_FRAMES);
			int p = pos0 + f;
			if (p >= R) p -= R;
			const bool keep = (f < n) && (f >= firstKept);
#pragma unroll
			for (int cg = 0; cg < G; cg++)
			{
				const f32x4 v = f32x4{ x[4 * cg], x[4 * cg + 1], x[4 * cg + 2], x[4 * cg + 3] };
				xb[TileIdx(f, G, cg)] = v;
				BufStore(srsrc, v, keep ? (ringOff + TileIdx(p, G, cg)) * 16 : OOB);
			}
		}

		// WaveNetLayerT::Process (WaveNet.h:462-494) for one frame per lane
		template <int G, int WPS>
		__device__ __forceinline__ void LayerPk(const WnStage& sd, CFloat wpk, CFloat vec, const f32x4* xbCur, f32x4* xbNext,
			__amdgpu_buffer_rsrc_t srsrc, int inPos0, int outPos0, int n, int f, int wave, float cond, float (&xc)[MAXC], float (&hd)[MAXC])
		{
			constexpr int C = 4 * G;
			const int K = sd.ksize;
			const int d = sd.dilation;

			// acc = conv bias (:288-289) + W_mix * cond (:471)
			f32x2 acc[C / 2];
#pragma unroll
			for (int o = 0; o < C / 2; o++)
			{
				const f32x2 b = f32x2{ vec[2 * o], vec[2 * o + 1] };
				const f32x2 wm = f32x2{ vec[16 + 2 * o], vec[16 + 2 * o + 1] };
				acc[o] = __builtin_elementwise_fma(wm, f32x2{ cond, cond }, b);
			}

			// dilated conv (:139-290): tap k reads the frame d*(K-1-k) back; the last tap is the layer input itself (registers)
			CFloat wconv = wpk + sd.pk_conv_off;
			for (int k = 0; k < K - 1; k++)
			{
				const int shift = d * (K - 1 - k);
				const int lo = wave * 64 - shift;
				float x[C];
				FetchFrame<G>(x, xbCur, srsrc, sd.ring_off, f - shift, lo, lo + 63, inPos0, sd.ring_frames);
				DensePk<C, C>(acc, wconv + k * (C * C), x);
			}
			{
				float x[C];
#pragma unroll
				for (int c = 0; c < C; c++) x[c] = xc[c];
				DensePk<C, C>(acc, wconv + (K - 1) * (C * C), x);
			}

			// activation (:473-480), head accumulate (:482)
			float z[C];
			const bool leaky = (sd.flags & WN_FLAG_LEAKY) != 0;
#pragma unroll
			for (int o = 0; o < C / 2; o++)
			{
				const f32x2 a = leaky ? LeakyReLU2(acc[o]) : FastTanh2(acc[o]);
				z[2 * o] = a.x;
				z[2 * o + 1] = a.y;
				hd[2 * o] += a.x;
				hd[2 * o + 1] += a.y;
			}

			if (sd.flags & WN_FLAG_NEED_OUTPUT)
			{
				// 1x1 + bias + residual (:486-491)
				f32x2 y[C / 2];
#pragma unroll
				for (int o = 0; o < C / 2; o++) y[o] = f32x2{ vec[32 + 2 * o] + xc[2 * o], vec[32 + 2 * o + 1] + xc[2 * o + 1] };
				DensePk<C, C>(y, wpk + sd.pk_w1_off, z);
#pragma unroll
				for (int o = 0; o < C / 2; o++)
				{
					xc[2 * o] = y[o].x;
					xc[2 * o + 1] = y[o].y;
				}
			}
			if (sd.flags & WN_FLAG_PUBLISH) PublishFrame<G>(xc, xbNext, srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, n, f);
		}

		template <int G>
		__device__ __forceinline__ void PublishDispatch(int G_rt, const float (&x)[MAXC], f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int pos0,
			int R, int n, int f);

		__device__ __forceinline__ void PublishAny(int G, const float (&x)[MAXC], f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int pos0, int R,
			int n, int f)
		{
			if (G == 4) PublishFrame<4>(x, xb, srsrc, ringOff, pos0, R, n, f);
			else if (G == 3) PublishFrame<3>(x, xb, srsrc, ringOff, pos0, R, n, f);
			else if (G == 2) PublishFrame<2>(x, xb, srsrc, ringOff, pos0, R, n, f);
			else PublishFrame<1>(x, xb, srsrc, ringOff, pos0, R, n, f);
		}

		// A2 head: out = scale * (bias + sum_k sum_c w[k][c] * head[t - (K-1-k)*dil][c])   (WaveNet.h:658-660, Conv1D C -> 1, K = 16)
		template <int G>
		__device__ __forceinline__ float HeadConvPk(const WnStage& sd, CFloat wpk, const f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int pos0, int f,
			int wave, float bias)
		{
			constexpr int C = 4 * G;
			float acc = bias;
			CFloat w = wpk + sd.pk_conv_off;
			for (int k = 0; k < sd.ksize; k++)
			{
				const int shift = sd.dilation * (sd.ksize - 1 - k);
				const int lo = wave * 64 - shift;
				float x[C];
				FetchFrame<G>(x, xb, srsrc, sd.ring_off, f - shift, lo, lo + 63, pos0, sd.ring_frames);
#pragma unroll
				for (int c = 0; c < C; c++) acc = __builtin_fmaf(w[k * C + c], x[c], acc);
			}
			return acc;
		}

		// grid = active streams of one model; block = WPS waves of 64 frames (WPS = 2: 128-frame blocks).
		// dynamic LDS: xbuf[2][WPS*4 tiles * 64] float4
		template <int WPS>
		__global__ void __launch_bounds__(64 * WPS) WaveNetPkKernel(const WnStage* __restrict__ stages, const float* __restrict__ wpack,
			const float* __restrict__ wpkGlobal, const int* __restrict__ ringFrames, int nstages, int nrings, int stateF4, float headScale,
			f32x4* __restrict__ state, const int* __restrict__ slots, const int* __restrict__ rows, const float* __restrict__ in,
			float* __restrict__ out, long inStride, long outStride, int n)
		{
			constexpr int NTB = WPS * 4; // tiles in the block
			extern __shared__ __attribute__((aligned(16))) char smem[];
			f32x4* xbuf = reinterpret_cast<f32x4*>(smem); // [2][NTB*64]

			const int lane = threadIdx.x & 63;
			const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			const int f = wave * 64 + lane; // this lane's frame in the block

			const int slot = slots[blockIdx.x];
			const int row = rows[blockIdx.x];
			f32x4* st = state + (size_t)slot * (size_t)stateF4;
			int* header = reinterpret_cast<int*>(st);
			const int myPos = header[lane]; // lane r holds the write cursor of ring r
			const __amdgpu_buffer_rsrc_t srsrc = MakeRsrc(st, (unsigned)stateF4 * 16u);
			CFloat wpk = (CFloat)wpkGlobal;
			CFloat wvec = (CFloat)wpack;

			const float cond = (f < n) ? in[(size_t)row * inStride + f] : 0.0f; // WaveNet.h:770 (input -> condition)
			float xc[MAXC], hd[MAXC];
#pragma unroll
			for (int c = 0; c < MAXC; c++)
			{
				xc[c] = 0.0f;
				hd[c] = 0.0f; // WaveNet.h:772 headArray.SetZero()
			}

			int cur = 0;
			for (int s = 0; s < nstages; s++)
			{
				const WnStage sd = LoadStage(stages, s);
				const int outPos0 = (sd.out_ring_id >= 0) ? __builtin_amdgcn_readlane(myPos, sd.out_ring_id) : 0;
				const int inPos0 = (sd.ring_id >= 0) ? __builtin_amdgcn_readlane(myPos, sd.ring_id) : 0;
				f32x4* xbCur = xbuf + cur * (NTB * 64);
				f32x4* xbNext = xbuf + (cur ^ 1) * (NTB * 64);
				CFloat vec = wvec + sd.vec_off * 4; // [0..15] conv/dense bias, [16..31] mix-in w, [32..47] 1x1 bias, [48..63] aux

				if (sd.type == WN_ST_LAYER)
				{
					if (sd.G == 4) LayerPk<4, WPS>(sd, wpk, vec, xbCur, xbNext, srsrc, inPos0, outPos0, n, f, wave, cond, xc, hd);
					else if (sd.G == 3) LayerPk<3, WPS>(sd, wpk, vec, xbCur, xbNext, srsrc, inPos0, outPos0, n, f, wave, cond, xc, hd);
					else if (sd.G == 2) LayerPk<2, WPS>(sd, wpk, vec, xbCur, xbNext, srsrc, inPos0, outPos0, n, f, wave, cond, xc, hd);
					else LayerPk<1, WPS>(sd, wpk, vec, xbCur, xbNext, srsrc, inPos0, outPos0, n, f, wave, cond, xc, hd);
					if (sd.flags & WN_FLAG_PUBLISH) cur ^= 1;
				}
				else if (sd.type == WN_ST_RECHANNEL_COND)
				{
#pragma unroll
					for (int c = 0; c < MAXC; c++) xc[c] = vec[48 + c] * cond; // :637 with InputSize == 1
					PublishAny(sd.out_G, xc, xbNext, srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, n, f);
					cur ^= 1;
				}
				else if (sd.type == WN_ST_ARRAY_LINK)
				{
					// previous array's headRechannel (K=1, :658-660) and this array's rechannel (:637); weights padded to 16x16
					f32x2 hn[MAXC / 2], xn[MAXC / 2];
#pragma unroll
					for (int o = 0; o < MAXC / 2; o++)
					{
						hn[o] = (sd.flags & WN_FLAG_BIAS) ? f32x2{ vec[2 * o], vec[2 * o + 1] } : f32x2{ 0.0f, 0.0f };
						xn[o] = f32x2{ 0.0f, 0.0f };
					}
					DensePk<MAXC, MAXC>(hn, wpk + sd.pk_w1_off, hd);
					DensePk<MAXC, MAXC>(xn, wpk + sd.pk_w2_off, xc);
#pragma unroll
					for (int o = 0; o < MAXC / 2; o++)
					{
						hd[2 * o] = hn[o].x; hd[2 * o + 1] = hn[o].y;
						xc[2 * o] = xn[o].x; xc[2 * o + 1] = xn[o].y;
					}
					PublishAny(sd.out_G, xc, xbNext, srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, n, f);
					cur ^= 1;
				}
				else if (sd.type == WN_ST_HEAD_DENSE_OUT)
				{
					float o = (sd.flags & WN_FLAG_BIAS) ? vec[0] : 0.0f;
					CFloat wh = wpk + sd.pk_w1_off;
#pragma unroll
					for (int c = 0; c < MAXC; c++) o = __builtin_fmaf(wh[c], hd[c], o);
					if (f < n) out[(size_t)row * outStride + f] = headScale * o; // :793-798
				}
				else // WN_ST_HEAD_CONV_OUT
				{
					PublishAny(sd.out_G, hd, xbNext, srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, n, f);
					cur ^= 1;
					BlockBarrier<WPS>();
					const float bias = (sd.flags & WN_FLAG_BIAS) ? vec[0] : 0.0f;
					float o;
					if (sd.G == 4) o = HeadConvPk<4>(sd, wpk, xbNext, srsrc, inPos0, f, wave, bias);
					else if (sd.G == 3) o = HeadConvPk<3>(sd, wpk, xbNext, srsrc, inPos0, f, wave, bias);
					else if (sd.G == 2) o = HeadConvPk<2>(sd, wpk, xbNext, srsrc, inPos0, f, wave, bias);
					else o = HeadConvPk<1>(sd, wpk, xbNext, srsrc, inPos0, f, wave, bias);
					if (f < n) out[(size_t)row * outStride + f] = headScale * o;
				}
				BlockBarrier<WPS>();
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

		template <int WPS>
		static hipError_t Launch(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in, float* out,
			long inStride, long outStride, int n, hipStream_t stream)
		{
			const size_t lds = (size_t)2 * WPS * 4 * 64 * 16;
			hipLaunchKernelGGL((WaveNetPkKernel<WPS>), dim3((unsigned)numStreams), dim3(64 * WPS), lds, stream, m.stages, m.wpack, m.wpk, m.ring_frames,
				m.nstages, m.nrings, m.state_f4, m.head_scale, reinterpret_cast<f32x4*>(state), slots, rows, in, out, inStride, outStride, n);
			return hipGetLastError();
		}
	}

	hipError_t LaunchWaveNetPk(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in, float* out,
		long inStride, long outStride, int n, hipStream_t stream)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		if (n > WN_MAX_FRAMES) return hipErrorInvalidValue;
		if (n > 64) return pk::Launch<2>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
		return pk::Launch<1>(m, state, slots, rows, numStreams, in, out, inStride, outStride, n, stream);
	}
}
