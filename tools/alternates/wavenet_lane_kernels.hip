// wavenet_lane_kernels.hip -- the "one wave = one stream" WaveNet block kernel for gfx950, for NARROW models (every layer array <= 8
// channels, dense head: NAM A1 Feather / Nano and the like).
//
// Same path and same stream-state format as wavenet_frame_kernels.hip (reference functions: WaveNetModelT / LayerArrayT / LayerT::Process,
// Conv1DT::Process, DenseLayerT::Process -- NeuralAudio/WaveNet.h:768-799, 632-661, 462-494, 139-290, 336-383; FastMath --
// NeuralAudio/Activation.h:83-118).  Why a third mapping: a 4-channel layer is 64 multiply-adds per frame -- there is nothing to feed
// a matrix pipe with, and what the other two kernels pay per layer (a workgroup barrier between the waves that share a stream's block,
// weight staging through LDS, its wait) is the whole cost: 23 stages x 1.3 us = 30 us per 1024 x 128 Nano block, 5 % of which is
// arithmetic.  Here nothing is shared between waves, so nothing is synchronised:
//
//   * one wave owns one stream's whole 128-frame block: lane l holds frames l and l + 64 ("halves"), ALL channels of both in registers;
//   * mat-muls are v_pk_fma_f32 chains, two output channels per instruction, weights straight from SGPRs (scalar loads of the
//     [tap][in][out] tables -- wave-uniform, cached, no LDS staging), each weight pair used for both halves;
//   * the dilated taps of in-block frames come from a wave-private LDS image of the layer input, updated in place (all reads of a
//     layer precede its writes in program order; LDS executes one wave's operations in order): no barrier, no double buffer;
//   * taps before the block start come from the per-layer HBM rings, requested one layer ahead into registers, predicated per lane;
//   * workgroup = 4 independent waves (streams); a surplus wave exits at once.
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

namespace na
{
	namespace ln
	{
		typedef float f32x2 __attribute__((ext_vector_type(2)));
		typedef float f32x4 __attribute__((ext_vector_type(4)));
		typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
		typedef const float __attribute__((address_space(4)))* CFloat; // wave-uniform read-only data -> scalar loads
		typedef const int __attribute__((address_space(4)))* CInt;

#ifndef NA_LN_ABL
#define NA_LN_ABL 0 // tuning builds: 1 = ring loads / stores all out of range, 2 = constant weights (no scalar loads)
#endif
		constexpr int OOB = (int)0x80000000;
		constexpr int FRAMES = WN_MAX_FRAMES; // 128 = 2 halves of 64 lanes
		constexpr int MAXG = 2;               // channel groups (of 4) held per frame
		constexpr int MAXC = 4 * MAXG;
		constexpr int NW = 4;                 // waves (= streams) per workgroup
		constexpr int HPF = 2;                // shifted taps whose ring history is requested a layer ahead (K = 3: all of them)

		__device__ __forceinline__ __amdgpu_buffer_rsrc_t MakeRsrc(const void* base, unsigned bytes)
		{
			return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
		}
		__device__ __forceinline__ f32x4 BufLoad(__amdgpu_buffer_rsrc_t r, int voff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0)); }
		__device__ __forceinline__ void BufStore(__amdgpu_buffer_rsrc_t r, f32x4 v, int voff) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0); }

		// the ints of a stage record this kernel reads (WnStage's first 18), as plain scalars
		struct Stage
		{
			int type, flags, G, ksize, dilation, ring_id, ring_off, ring_frames, out_ring_id, out_ring_off, out_ring_frames, out_G, a4_off, a4_floats, vec_off,
				pk_conv_off, pk_w1_off, pk_w2_off;
		};
		static_assert(offsetof(WnStage, pk_conv_off) == 15 * sizeof(int) && offsetof(WnStage, pk_w1_off) == 16 * sizeof(int) && offsetof(WnStage, pk_w2_off) == 17 * sizeof(int),
			"Stage mirrors the head of WnStage");

		__device__ __forceinline__ Stage LoadStage(const WnStage* __restrict__ stages, int s)
		{
			Stage sd;
			CInt src = (CInt)(const int*)(stages + s);
			int* dst = reinterpret_cast<int*>(&sd);
#pragma unroll
			for (int i = 0; i < 18; i++) dst[i] = src[i];
			return sd;
		}

		// Activation.h:83-91 on two channels (packed f32 math; |x + e x |x|| == |x| + e x^2 since 1 + e |x| > 0; division = num * v_rcp_f32(den))
		__device__ __forceinline__ f32x2 FastTanh2(f32x2 x)
		{
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
		// StdMath policy (Activation.h:37-40): tanh(x) = 1 - 2 / (e^(2x) + 1) on the exp2 / rcp units
		__device__ __forceinline__ float StdTanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.0f); }
		// Activation.h:110-118
		__device__ __forceinline__ float LeakyReLU(float v) { return v > 0.0f ? v : 0.01f * v; }

		__device__ __forceinline__ f32x2 Activate2(f32x2 a, int flags)
		{
			if (flags & WN_FLAG_LEAKY) return f32x2{ LeakyReLU(a.x), LeakyReLU(a.y) }; // wave-uniform branches
			if (flags & WN_FLAG_STD_TANH) return f32x2{ StdTanh(a.x), StdTanh(a.y) };
			return FastTanh2(a);
		}

		// float4 index of (frame, channel group) in the tiled ring image with G groups: ((frame >> 4) G + cg) 16 + (frame & 15)
		__device__ __forceinline__ int TileIdx(int frame, int G, int cg) { return ((frame >> 4) * G + cg) * 16 + (frame & 15); }

		struct Ctx
		{
			const WnStage* __restrict__ stages;
			int nstages;
			CFloat wvec; // wpack: per-stage vectors [0..15] conv / dense bias, [16..31] mix-in, [32..47] 1x1 bias, [48..63] aux
			CFloat wpk;  // [tap][in][out] / [in][out] tables
			f32x4* xb;   // this wave's LDS image of the current layer input: [cg][FRAMES] quads
			__amdgpu_buffer_rsrc_t srsrc; // this stream's state
			int myPos;   // lane r: write cursor of ring r
			int n;       // frames in the block
			int lane;
			float cond[2]; // WaveNet.h:770: the input sample is the condition of every layer
		};

		// acc[h][o] += sum_c w[c][o] * x[h][c]   (w: [CIN][COUT] floats from SGPRs; two output channels per v_pk_fma_f32; each weight
		// pair serves both halves)
		template <int CIN, int COUT>
		__device__ __forceinline__ void DensePk(f32x2 (&acc)[2][COUT / 2], CFloat w, const float (&x)[2][CIN])
		{
#pragma unroll
			for (int c = 0; c < CIN; c++)
			{
#pragma unroll
				for (int o = 0; o < COUT / 2; o++)
				{
					const f32x2 wp = (NA_LN_ABL & 2) ? f32x2{ 0.001f * (c + 1), 0.002f * (o + 1) } : f32x2{ w[c * COUT + 2 * o], w[c * COUT + 2 * o + 1] };
#pragma unroll
					for (int h = 0; h < 2; h++) acc[h][o] = __builtin_elementwise_fma(wp, f32x2{ x[h][c], x[h][c] }, acc[h][o]);
				}
			}
		}

		// ring history of one shifted tap for both halves: the lanes whose frame f - shift lies before the block start (predicated
		// through the offset: an out-of-range buffer load returns zeros and costs nothing)
		template <int G>
		__device__ __forceinline__ void LoadHistory(const Ctx& cx, f32x4 (&hist)[2][G], int ringOff, int R, int pos0, int shift, bool valid)
		{
#pragma unroll
			for (int h = 0; h < 2; h++)
			{
				const int off = cx.lane + 64 * h - shift;
				int p = pos0 + off; // off >= -(R - FRAMES): one wrap
				if (p < 0) p += R;
#pragma unroll
				for (int cg = 0; cg < G; cg++) hist[h][cg] = BufLoad(cx.srsrc, (!(NA_LN_ABL & 1) && valid && off < 0) ? (ringOff + TileIdx(p, G, cg)) * 16 : OOB);
			}
		}

		// one shifted tap of both halves into x[h][c]: LDS image for in-block frames, the prefetched history otherwise (decided per
		// half on the scalar unit where the whole half lies on one side)
		template <int G>
		__device__ __forceinline__ void TapOperand(const Ctx& cx, float (&x)[2][4 * G], const f32x4 (&hist)[2][G], int shift)
		{
#pragma unroll
			for (int h = 0; h < 2; h++)
			{
				const int off = cx.lane + 64 * h - shift;
				const int lo = 64 * h - shift; // wave-uniform: first frame of the half; last = lo + 63
#pragma unroll
				for (int cg = 0; cg < G; cg++)
				{
					f32x4 v;
					if (lo >= 0) v = cx.xb[cg * FRAMES + off];
					else if (lo + 63 < 0) v = hist[h][cg];
					else
					{
						const f32x4 l = cx.xb[cg * FRAMES + (off < 0 ? 0 : off)];
						const f32x4 r = hist[h][cg];
						v = f32x4{ off < 0 ? r.x : l.x, off < 0 ? r.y : l.y, off < 0 ? r.z : l.z, off < 0 ? r.w : l.w };
					}
					x[h][4 * cg] = v.x; x[h][4 * cg + 1] = v.y; x[h][4 * cg + 2] = v.z; x[h][4 * cg + 3] = v.w;
				}
			}
		}

		// ... without a prefetch (taps beyond the first HPF of a large kernel)
		template <int G>
		__device__ __forceinline__ void TapOperandInline(const Ctx& cx, float (&x)[2][4 * G], int ringOff, int R, int pos0, int shift)
		{
			f32x4 hist[2][G];
			LoadHistory<G>(cx, hist, ringOff, R, pos0, shift, true);
			TapOperand<G>(cx, x, hist, shift);
		}

		// layer output of both halves -> the LDS image (in-block taps of the next layer) and -> the next layer's ring (history for LATER
		// blocks: only the last R - FRAMES frames of a block can ever be read back)
		template <int G>
		__device__ __forceinline__ void Publish(const Ctx& cx, const float (&xc)[2][MAXC], int outRingOff, int outR, int outPos0)
		{
			const int firstKept = cx.n - (outR - FRAMES);
#pragma unroll
			for (int h = 0; h < 2; h++)
			{
				const int f = cx.lane + 64 * h;
				int p = outPos0 + f;
				if (p >= outR) p -= outR;
				const bool keep = (f < cx.n) && (f >= firstKept);
#pragma unroll
				for (int cg = 0; cg < G; cg++)
				{
					const f32x4 v = f32x4{ xc[h][4 * cg], xc[h][4 * cg + 1], xc[h][4 * cg + 2], xc[h][4 * cg + 3] };
					cx.xb[cg * FRAMES + f] = v;
					BufStore(cx.srsrc, v, (!(NA_LN_ABL & 1) && keep) ? (outRingOff + TileIdx(p, G, cg)) * 16 : OOB);
				}
			}
		}

		// A run of consecutive WaveNet layer stages (WaveNetLayerT::Process, WaveNet.h:462-494) with the same channel-group count
		template <int G>
		__device__ __forceinline__ void RunLayers(const Ctx& cx, int& s, Stage& sd, float (&xc)[2][MAXC], float (&hd)[2][MAXC])
		{
			constexpr int C = 4 * G;
			f32x4 hist[HPF][2][G];
			{
				const int pos0 = __builtin_amdgcn_readlane(cx.myPos, sd.ring_id);
#pragma unroll
				for (int t = 0; t < HPF; t++) LoadHistory<G>(cx, hist[t], sd.ring_off, sd.ring_frames, pos0, sd.dilation * (sd.ksize - 1 - t), t < sd.ksize - 1);
			}
			do
			{
				Stage sdn = sd;
				sdn.type = -1;
				if (s + 1 < cx.nstages) sdn = LoadStage(cx.stages, s + 1);
				const int K = sd.ksize, d = sd.dilation;
				CFloat vec = cx.wvec + sd.vec_off * 4;
				CFloat wconv = cx.wpk + sd.pk_conv_off;
				const int inPos0 = __builtin_amdgcn_readlane(cx.myPos, sd.ring_id);

				// acc = conv bias (:288-289) + W_mix * cond (:471)
				f32x2 acc[2][C / 2];
#pragma unroll
				for (int o = 0; o < C / 2; o++)
				{
					const f32x2 b = f32x2{ vec[2 * o], vec[2 * o + 1] };
					const f32x2 wm = f32x2{ vec[16 + 2 * o], vec[16 + 2 * o + 1] };
#pragma unroll
					for (int h = 0; h < 2; h++) acc[h][o] = __builtin_elementwise_fma(wm, f32x2{ cx.cond[h], cx.cond[h] }, b);
				}
				// dilated conv (:139-290): tap k reads the frame d (K-1-k) back; the last tap is the layer input itself (registers)
#pragma unroll
				for (int k = 0; k < HPF; k++)
				{
					if (k < K - 1)
					{
						float x[2][C];
						TapOperand<G>(cx, x, hist[k], d * (K - 1 - k));
						DensePk<C, C>(acc, wconv + k * (C * C), x);
					}
				}
				// history of the NEXT layer's first HPF taps (the registers are free again)
				{
					const bool haveNext = (s + 1 < cx.nstages) && sdn.type == WN_ST_LAYER && sdn.G == G;
					const int nextPos0 = __builtin_amdgcn_readlane(cx.myPos, haveNext ? sdn.ring_id : 0);
#pragma unroll
					for (int t = 0; t < HPF; t++)
						LoadHistory<G>(cx, hist[t], sdn.ring_off, sdn.ring_frames, nextPos0, sdn.dilation * (sdn.ksize - 1 - t), haveNext && t < sdn.ksize - 1);
				}
				for (int k = HPF; k < K - 1; k++)
				{
					float x[2][C];
					TapOperandInline<G>(cx, x, sd.ring_off, sd.ring_frames, inPos0, d * (K - 1 - k));
					DensePk<C, C>(acc, wconv + k * (C * C), x);
				}
				{
					float x[2][C];
#pragma unroll
					for (int h = 0; h < 2; h++)
#pragma unroll
						for (int c = 0; c < C; c++) x[h][c] = xc[h][c];
					DensePk<C, C>(acc, wconv + (K - 1) * (C * C), x);
				}

				// activation (:473-480), head accumulate (:482)
				float z[2][C];
#pragma unroll
				for (int h = 0; h < 2; h++)
#pragma unroll
					for (int o = 0; o < C / 2; o++)
					{
						const f32x2 a = Activate2(acc[h][o], sd.flags);
						z[h][2 * o] = a.x;
						z[h][2 * o + 1] = a.y;
						hd[h][2 * o] += a.x;
						hd[h][2 * o + 1] += a.y;
					}
				if (sd.flags & WN_FLAG_NEED_OUTPUT)
				{
					// 1x1 + bias + residual (:486-491)
					f32x2 y[2][C / 2];
#pragma unroll
					for (int h = 0; h < 2; h++)
#pragma unroll
						for (int o = 0; o < C / 2; o++) y[h][o] = f32x2{ vec[32 + 2 * o] + xc[h][2 * o], vec[32 + 2 * o + 1] + xc[h][2 * o + 1] };
					DensePk<C, C>(y, cx.wpk + sd.pk_w1_off, z);
#pragma unroll
					for (int h = 0; h < 2; h++)
#pragma unroll
						for (int o = 0; o < C / 2; o++)
						{
							xc[h][2 * o] = y[h][o].x;
							xc[h][2 * o + 1] = y[h][o].y;
						}
				}
				if (sd.flags & WN_FLAG_PUBLISH) Publish<G>(cx, xc, sd.out_ring_off, sd.out_ring_frames, __builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id));
				sd = sdn;
				s++;
			} while (s < cx.nstages && sd.type == WN_ST_LAYER && sd.G == G);
		}

		// previous array's head rechannel (K = 1, WaveNet.h:658-660) and this array's rechannel (:637); GP / GN: channel groups of the
		// previous / this array (the tables are padded to them)
		template <int GP, int GN>
		__device__ __forceinline__ void LinkStage(const Ctx& cx, const Stage& sd, float (&xc)[2][MAXC], float (&hd)[2][MAXC])
		{
			constexpr int CP = 4 * GP, CN = 4 * GN;
			CFloat vec = cx.wvec + sd.vec_off * 4;
			f32x2 hn[2][CN / 2], xn[2][CN / 2];
#pragma unroll
			for (int h = 0; h < 2; h++)
#pragma unroll
				for (int o = 0; o < CN / 2; o++)
				{
					hn[h][o] = (sd.flags & WN_FLAG_BIAS) ? f32x2{ vec[2 * o], vec[2 * o + 1] } : f32x2{ 0.0f, 0.0f };
					xn[h][o] = f32x2{ 0.0f, 0.0f };
				}
			float hin[2][CP], xin[2][CP];
#pragma unroll
			for (int h = 0; h < 2; h++)
#pragma unroll
				for (int c = 0; c < CP; c++)
				{
					hin[h][c] = hd[h][c];
					xin[h][c] = xc[h][c];
				}
			DensePk<CP, CN>(hn, cx.wpk + sd.pk_w1_off, hin);
			DensePk<CP, CN>(xn, cx.wpk + sd.pk_w2_off, xin);
#pragma unroll
			for (int h = 0; h < 2; h++)
			{
#pragma unroll
				for (int c = 0; c < MAXC; c++)
				{
					hd[h][c] = 0.0f;
					xc[h][c] = 0.0f;
				}
#pragma unroll
				for (int o = 0; o < CN / 2; o++)
				{
					hd[h][2 * o] = hn[h][o].x; hd[h][2 * o + 1] = hn[h][o].y;
					xc[h][2 * o] = xn[h][o].x; xc[h][2 * o + 1] = xn[h][o].y;
				}
			}
			const int outPos0 = __builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id);
			Publish<GN>(cx, xc, sd.out_ring_off, sd.out_ring_frames, outPos0);
		}

		struct GroupArgs
		{
			const WnStage* stages;
			const float* wpack;
			const float* wpk;
			const int* ringFrames;
			f32x4* state;
			const int* slots;
			const int* rows;
			int nstages, nrings, stateF4;
			float headScale;
			int numStreams, slot0, row0, firstBlock;
		};
		struct LaunchArgs
		{
			GroupArgs g[WN_FRAME_MAX_GROUPS];
			int numGroups;
		};

		// grid = sum over groups of ceil(streams / NW); dynamic LDS: NW x [MAXG][FRAMES] quads
		__global__ void __launch_bounds__(64 * NW) WaveNetLaneKernel(const LaunchArgs args, const float* __restrict__ in, float* __restrict__ out, long inStride,
			long outStride, int n)
		{
			int gi = 0;
			for (int i = 1; i < args.numGroups; i++)
				if ((int)blockIdx.x >= args.g[i].firstBlock) gi = i;
			const GroupArgs& ga = args.g[gi];
			const int lane = threadIdx.x & 63;
			const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			const int sidx = ((int)blockIdx.x - ga.firstBlock) * NW + wave;
			if (sidx >= ga.numStreams) return; // no barriers in this kernel: a surplus wave just leaves
			extern __shared__ __attribute__((aligned(16))) char smem[];

			const int slot = ga.slots ? ga.slots[sidx] : ga.slot0 + sidx;
			const int row = ga.slots ? ga.rows[sidx] : ga.row0 + sidx;
			f32x4* st = ga.state + (size_t)slot * (size_t)ga.stateF4;
			int* header = reinterpret_cast<int*>(st);

			Ctx cx;
			cx.stages = ga.stages;
			cx.nstages = ga.nstages;
			cx.wvec = (CFloat)ga.wpack;
			cx.wpk = (CFloat)ga.wpk;
			cx.xb = reinterpret_cast<f32x4*>(smem) + wave * (MAXG * FRAMES);
			cx.srsrc = MakeRsrc(st, (unsigned)ga.stateF4 * 16u);
			cx.myPos = header[lane]; // lane r holds the write cursor of ring r
			cx.n = n;
			cx.lane = lane;
#pragma unroll
			for (int h = 0; h < 2; h++) cx.cond[h] = (lane + 64 * h < n) ? in[(size_t)row * inStride + lane + 64 * h] : 0.0f;

			float xc[2][MAXC], hd[2][MAXC];
#pragma unroll
			for (int h = 0; h < 2; h++)
#pragma unroll
				for (int c = 0; c < MAXC; c++)
				{
					xc[h][c] = 0.0f;
					hd[h][c] = 0.0f; // WaveNet.h:772 headArray.SetZero()
				}

			Stage sd = LoadStage(cx.stages, 0);
			int s = 0;
			int G = 1; // channel groups of the array being processed
			while (s < cx.nstages)
			{
				if (sd.type == WN_ST_LAYER)
				{
					G = sd.G;
					if (G == 2) RunLayers<2>(cx, s, sd, xc, hd);
					else RunLayers<1>(cx, s, sd, xc, hd);
					continue; // sd / s already advanced
				}
				if (sd.type == WN_ST_RECHANNEL_COND)
				{
					CFloat vec = cx.wvec + sd.vec_off * 4;
#pragma unroll
					for (int h = 0; h < 2; h++)
#pragma unroll
						for (int c = 0; c < MAXC; c++) xc[h][c] = vec[48 + c] * cx.cond[h]; // :637 with InputSize == 1 (padding channels: zero weights)
					G = sd.out_G;
					const int outPos0 = __builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id);
					if (G == 2) Publish<2>(cx, xc, sd.out_ring_off, sd.out_ring_frames, outPos0);
					else Publish<1>(cx, xc, sd.out_ring_off, sd.out_ring_frames, outPos0);
				}
				else if (sd.type == WN_ST_ARRAY_LINK)
				{
					const int GN = sd.out_G;
					if (G == 2 && GN == 2) LinkStage<2, 2>(cx, sd, xc, hd);
					else if (G == 2) LinkStage<2, 1>(cx, sd, xc, hd);
					else if (GN == 2) LinkStage<1, 2>(cx, sd, xc, hd);
					else LinkStage<1, 1>(cx, sd, xc, hd);
					G = GN;
				}
				else // WN_ST_HEAD_DENSE_OUT (the host never routes a conv head here)
				{
					CFloat vec = cx.wvec + sd.vec_off * 4;
					CFloat wh = cx.wpk + sd.pk_w1_off;
#pragma unroll
					for (int h = 0; h < 2; h++)
					{
						float o = (sd.flags & WN_FLAG_BIAS) ? vec[0] : 0.0f;
#pragma unroll
						for (int c = 0; c < MAXC; c++) o = __builtin_fmaf(wh[c], hd[h][c], o);
						const int f = lane + 64 * h;
						if (f < n) out[(size_t)row * outStride + f] = ga.headScale * o; // :793-798
					}
				}
				s++;
				if (s < cx.nstages) sd = LoadStage(cx.stages, s);
			}

			// advance every ring cursor by n (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
			if (lane < ga.nrings)
			{
				const int R = ga.ringFrames[lane];
				int p = cx.myPos + n;
				if (p >= R) p -= R;
				header[lane] = p;
			}
		}
	}

	hipError_t LaunchWaveNetLaneFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		if (n <= 0 || numGroups <= 0) return hipSuccess;
		if (n > WN_MAX_FRAMES || numGroups > WN_FRAME_MAX_GROUPS) return hipErrorInvalidValue;
		ln::LaunchArgs args;
		args.numGroups = numGroups;
		int blocks = 0;
		for (int i = 0; i < numGroups; i++)
		{
			const WnFrameGroup& g = groups[i];
			if (g.numStreams <= 0) return hipErrorInvalidValue;
			const WnModelDev& m = *g.model;
			ln::GroupArgs& a = args.g[i];
			a.stages = m.stages;
			a.wpack = m.wpack;
			a.wpk = m.wpk;
			a.ringFrames = m.ring_frames;
			a.state = reinterpret_cast<ln::f32x4*>(g.state);
			a.slots = g.slots;
			a.rows = g.rows;
			a.nstages = m.nstages;
			a.nrings = m.nrings;
			a.stateF4 = m.state_f4;
			a.headScale = m.head_scale;
			a.numStreams = g.numStreams;
			a.slot0 = g.slot0;
			a.row0 = g.row0;
			a.firstBlock = blocks;
			blocks += (g.numStreams + ln::NW - 1) / ln::NW;
		}
		const size_t lds = (size_t)ln::NW * ln::MAXG * ln::FRAMES * 16;
		hipLaunchKernelGGL(ln::WaveNetLaneKernel, dim3((unsigned)blocks), dim3(64 * ln::NW), lds, stream, args, in, out, inStride, outStride, n);
		return hipGetLastError();
	}
}
