// wavenet_frame_kernels.hip -- the "lane = frame" WaveNet block kernel for gfx950 (v_mfma_f32_4x4x1_16b_f32).
//
// The f32 kernel of round 1; since round 2 it serves the models the f16-split kernel (wavenet_split_kernels.hip) is not faster on --
// narrow (<= 4-channel arrays) models in batches of up to 1024 streams and large-kernel (A2) architectures, see FamilyFor() / PackFor() /
// PadFor() in gpu_batch.cpp (reference functions:
// WaveNetModelT/LayerArrayT/LayerT::Process, Conv1DT::Process, DenseLayerT::Process -- NeuralAudio/WaveNet.h:768-799,
// 632-661,462-494,139-290,336-383; FastMath -- NeuralAudio/Activation.h:83-118), different mapping of the arithmetic:
//
//   * lane = one audio frame, a wave = 64 consecutive frames, a workgroup = the 1-2 waves of one stream's block;
//   * every lane holds ALL channels of its frame in registers; mat-muls are chains of v_mfma_f32_4x4x1_16b_f32:
//     16 independent 4x4 outer products per instruction, block b = lanes 4b..4b+3 = frames 4b..4b+3.  B operand = one
//     input channel of the lane's frame (a plain VGPR).  A operand: register w[og] holds W[4*og + (lane&3)][c = lane>>2],
//     i.e. the weights of 16 input channels spread over the 16 blocks, and MFMA number c is issued with CBSZ=4 / ABID=c so
//     that block c's four A values are broadcast to every block (probe: tools/microbench/mfma_cbsz_probe.hip).  Result = 4
//     output channels of the lane's own frame.  Operands and results are in the SAME lane = frame layout, so a layer runs
//     conv -> activation -> 1x1 -> residual entirely in registers, with no padding: an 8-channel layer issues exactly half
//     the MFMAs of a 16-channel one and the activation only touches real channels.
//   * the A operands of a stage (4.3 KB for a 16-channel layer) are staged into LDS one stage ahead; a tap costs ONE LDS read
//     per lane (16 B for 16 channels) for its C*C/4 MFMAs.
//
// Why not the f32 16x16x4 tile mapping (tools/alternates/wavenet_tile_kernels.hip, round 1): measured on MI355X (tools/microbench/mfma_valu_overlap.hip)
// f32 MFMA runs at the f32 VALU rate and does NOT overlap with VALU work on the same SIMD, so every padded MFMA row and
// every activation evaluated on a padding lane is pure loss; the tile mapping pads 8-channel layers to 16 rows.
#include "device_once.h"
#include "tuning.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

namespace na
{
	namespace fr
	{
		typedef float f32x2 __attribute__((ext_vector_type(2)));
		typedef float f32x4 __attribute__((ext_vector_type(4)));
		typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
		typedef const float __attribute__((address_space(4)))* CFloat; // wave-uniform read-only data -> scalar loads
		typedef const int __attribute__((address_space(4)))* CInt;

#ifndef NA_ABL
#define NA_ABL 0 // ablation bit mask for tuning builds only (tools/ablate.sh); 0 in the product.  1: no activation math, 2: no MFMA,
                 // 4: no history loads / ring stores, 8: no barrier, 16: no weight staging, 32: staging loads but no LDS writes, 64: LDS writes but no loads, 128: history loads from cache-resident slots, 256: no ring stores
#endif
		constexpr int OOB = (int)0x80000000;
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

		// the 16 hot ints of a stage descriptor (WnStage's first 64 bytes: one scalar load), as a plain struct of scalars so that it
		// lives in SGPRs (the full 128-byte record with its reserved[] tail would be copied through scratch memory)
		struct FrStage
		{
			int type, flags, G, ksize, dilation, ring_id, ring_off, ring_frames, out_ring_id, out_ring_off, out_ring_frames, out_G, a4_off, a4_floats, vec_off,
				pk_conv_off;
		};
		constexpr int STAGE_HOT_INTS = (int)(sizeof(FrStage) / sizeof(int));
		static_assert(STAGE_HOT_INTS == 16 && offsetof(WnStage, type) == 0 && offsetof(WnStage, a4_floats) == 13 * sizeof(int) &&
						  offsetof(WnStage, pk_conv_off) == 15 * sizeof(int) && offsetof(WnStage, pk_w1_off) == 16 * sizeof(int),
			"FrStage mirrors the first 16 ints of WnStage");

		__device__ __forceinline__ FrStage LoadStage(const WnStage* __restrict__ stages, int s)
		{
			FrStage sd;
			CInt src = (CInt)(const int*)(stages + s);
			int* dst = reinterpret_cast<int*>(&sd);
#pragma unroll
			for (int i = 0; i < STAGE_HOT_INTS; i++) dst[i] = src[i];
			return sd;
		}

		__device__ __forceinline__ f32x2 Abs2(f32x2 v)
		{
			f32x2 r;
			r.x = __builtin_fabsf(v.x);
			r.y = __builtin_fabsf(v.y);
			return r;
		}

		// Activation.h:83-91 on two channels, packed math, division = num * v_rcp_f32(den)
		__device__ __forceinline__ f32x2 FastTanh2(f32x2 x)
		{
			if (NA_ABL & 1) return x * 0.5f;
			const f32x2 ax = Abs2(x);
			const f32x2 x2 = x * x;
			const f32x2 num = x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2);
			// |x + e*x*|x|| == |x| + e*x^2 (1 + e|x| > 0): three packed ops for the denominator instead of six
			const f32x2 den = 2.44506634652299f + (2.44506634652299f + x2) * (ax + 0.814642734961073f * x2);
			f32x2 r;
			r.x = __builtin_amdgcn_rcpf(den.x);
			r.y = __builtin_amdgcn_rcpf(den.y);
			return num * r;
		}

		// StdMath policy (Activation.h:37-40): tanh(x) = 1 - 2 / (e^(2x) + 1) on the exp2 / rcp units (absolute error ~1e-7)
		__device__ __forceinline__ f32x2 StdTanh2(f32x2 x)
		{
			f32x2 r;
			r.x = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x.x * 2.885390081777927f) + 1.0f);
			r.y = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x.y * 2.885390081777927f) + 1.0f);
			return r;
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
			if (NA_ABL & 8) return;
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

		// float4 index of (frame, channel group) in the LDS block image: [64-frame wave part][cg][64 frames] -- a lane reads or writes its
		// frame's G float4s 1 KB apart (one immediate offset per channel group), 64 consecutive frames are 1 KB contiguous
		__device__ __forceinline__ int LdsIdx(int frame, int G, int cg) { return ((frame >> 6) * G + cg) * 64 + (frame & 63); }

		// Channels [4*cg, 4*cg+4) of the frame `off` frames from the block start (off < 0: history) for this lane.
		// lo/hi: range of `off` over the wave (scalar) -> whole wave in block / whole wave in history / mixed.
		// float4 per thread staged per stage by WeightStager (>= 6 KB per workgroup >= any official stage; larger blocks use its tail loop)
		constexpr int StagerWcopy(int nwaves) { return (384 + 64 * nwaves - 1) / (64 * nwaves); }

		// HPF = number of shifted taps (most shifted first) whose ring history is requested one layer ahead: 2 covers every tap of the K = 3
		// architectures; runs of narrow layers (G <= 2) of models with larger kernels (A2: K = 6 / 15) use HPF_WIDE, the rest of their taps
		// load in line.  HPF_LDS sizes the per-wave LDS history buffers of the PF == 2 variant.
		constexpr int HPF_NARROW = 2, HPF_WIDE = 5, HPF_LDS = 2;

		// history part of one tap for this lane's frame: channels of frame (pos0 + off) from the ring (lanes inside the block: nothing)
		// = frame (f - shift) of the block for the lanes with f < shift, when `valid` (wave-uniform); all other lanes load nothing.
		// The wave-uniform part of the ring arithmetic stays on the scalar unit: 9 VALU instructions per tap.
		template <int G>
		__device__ __forceinline__ void LoadHistory(f32x4 (&h)[G], __amdgpu_buffer_rsrc_t srsrc, int ringOff, int f, int shift, bool valid, int pos0, int R)
		{
			if (NA_ABL & 4)
			{
#pragma unroll
				for (int cg = 0; cg < G; cg++) h[cg] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				return;
			}
			int base = pos0 - shift; // shift <= R - 128, so one wrap is enough
			if (base < 0) base += R;
			unsigned p = (unsigned)(base + f);
			p = __builtin_elementwise_min(p, p - (unsigned)R); // p >= R ? p - R : p
			const int addr = (int)((p >> 4) * (unsigned)(G * 256) + (unsigned)(ringOff * 16)) + (int)((p & 15u) << 4);
			const int hoff = (valid && f < shift) ? addr : OOB;
#pragma unroll
			for (int cg = 0; cg < G; cg++) h[cg] = BufLoad(srsrc, hoff + cg * 256);
		}

		// FetchFrame with the history part already in registers.  Branch-free on purpose: with a fixed number of VMEM operations per
		// layer on every path the compiler can place COUNTED vmcnt waits; a conditional load anywhere in the loop makes it wait for
		// the youngest loads too, which would expose the very HBM latency the prefetch is meant to hide.
		template <int G>
		__device__ __forceinline__ void FetchFramePre(float (&x)[4 * G], const f32x4* xb, int off, const f32x4 (&hpre)[G])
		{
			// one divergent region per tap (not one per channel group): the in-block lanes overwrite the prefetched history with
			// the LDS image, all G reads back to back
			f32x4 v[G];
#pragma unroll
			for (int cg = 0; cg < G; cg++) v[cg] = hpre[cg];
			if (off >= 0)
			{
				const int base = LdsIdx(off, G, 0);
#pragma unroll
				for (int cg = 0; cg < G; cg++) v[cg] = xb[base + cg * 64];
			}
#pragma unroll
			for (int cg = 0; cg < G; cg++)
			{
				x[4 * cg] = v[cg].x; x[4 * cg + 1] = v[cg].y; x[4 * cg + 2] = v[cg].z; x[4 * cg + 3] = v[cg].w;
			}
		}

		template <int G>
		__device__ __forceinline__ void FetchFrame(float (&x)[4 * G], const f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int off, int lo, int hi,
			int pos0, int R)
		{
			f32x4 v[G];
			if (lo >= 0)
			{
				// whole wave inside the current block: LDS only
				const int base = LdsIdx(off, G, 0);
#pragma unroll
				for (int cg = 0; cg < G; cg++) v[cg] = xb[base + cg * 64];
			}
			else
			{
				int p = pos0 + off;
				if (p < 0) p += R;
				if (p >= R) p -= R;
				const int vbase = (ringOff + TileIdx(p, G, 0)) * 16;
				if (hi < 0)
				{
					// whole wave in the past: ring only
#pragma unroll
					for (int cg = 0; cg < G; cg++) v[cg] = BufLoad(srsrc, vbase + cg * 256);
				}
				else
				{
					// the wave straddles the block start: both, all loads issued before any is consumed
					const int base = LdsIdx(off < 0 ? 0 : off, G, 0);
					const int hoff = (off < 0) ? vbase : OOB;
					f32x4 l[G], h[G];
#pragma unroll
					for (int cg = 0; cg < G; cg++)
					{
						l[cg] = xb[base + cg * 64];
						h[cg] = BufLoad(srsrc, hoff + cg * 256);
					}
#pragma unroll
					for (int cg = 0; cg < G; cg++) v[cg] = (off < 0) ? h[cg] : l[cg];
				}
			}
#pragma unroll
			for (int cg = 0; cg < G; cg++)
			{
				x[4 * cg] = v[cg].x; x[4 * cg + 1] = v[cg].y; x[4 * cg + 2] = v[cg].z; x[4 * cg + 3] = v[cg].w;
			}
		}

		// the ABID field is an immediate: unroll over the input channel at compile time
		template <int CIN, int COUT, int CI>
		__device__ __forceinline__ void DenseMfmaRegs(f32x4 (&acc)[COUT / 4], const float (&w)[COUT / 4], const float (&x)[CIN])
		{
			if constexpr (CI < CIN)
			{
#pragma unroll
				for (int og = 0; og < COUT / 4; og++) acc[og] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[og], x[CI], acc[og], 4, CI, 0);
				DenseMfmaRegs<CIN, COUT, CI + 1>(acc, w, x);
			}
		}

		// acc[og] += W[4og..4og+3][0..CIN) * x   for this lane's frame.  a: this lane's COUT/4 floats of the tap's LDS image (see
		// PackConvA4): register w[og] holds W[4og + (lane&3)][c = lane>>2]; MFMA number c broadcasts block c's A operand to all 16
		// blocks (CBSZ=4, ABID=c), so one LDS read per tap feeds all CIN*COUT/4 MFMAs of the tap.
		template <int CIN, int COUT>
		__device__ __forceinline__ void DenseMfma(f32x4 (&acc)[COUT / 4], const float* a, const float (&x)[CIN])
		{
			constexpr int NOG = COUT / 4;
			static_assert(CIN <= 16, "one A register covers at most 16 input channels");
			if (NA_ABL & 2)
			{
#pragma unroll
				for (int og = 0; og < NOG; og++) acc[og].x += x[og];
				return;
			}
			float w[NOG];
			if constexpr (NOG == 4)
			{
				const f32x4 v = *reinterpret_cast<const f32x4*>(a);
				w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
			}
			else if constexpr (NOG == 2)
			{
				const f32x2 v = *reinterpret_cast<const f32x2*>(a);
				w[0] = v.x; w[1] = v.y;
			}
			else
			{
#pragma unroll
				for (int og = 0; og < NOG; og++) w[og] = a[og];
			}
			DenseMfmaRegs<CIN, COUT, 0>(acc, w, x);
		}

		// this lane's frame of a layer output -> LDS block image (in-block taps of the next layer)
		template <int G>
		__device__ __forceinline__ void PublishLds(const float (&x)[MAXC], f32x4* xb, int f)
		{
#pragma unroll
			for (int cg = 0; cg < G; cg++) xb[LdsIdx(f, G, cg)] = f32x4{ x[4 * cg], x[4 * cg + 1], x[4 * cg + 2], x[4 * cg + 3] };
		}

		// ... and -> the next layer's HBM ring (history for LATER blocks: only the last R-128 frames of a block can ever be read back).
		// Always G store instructions (predicated through the offset) so that the VMEM count per layer is fixed.
		template <int G>
		__device__ __forceinline__ void StoreRing(const float (&x)[MAXC], __amdgpu_buffer_rsrc_t srsrc, int ringOff, int pos0, int R, int n, int f)
		{
			if (NA_ABL & (4 | 256)) return;
			const int firstKept = n - (R - WN_MAX_FRAMES);
			unsigned p = (unsigned)(pos0 + f);
			p = __builtin_elementwise_min(p, p - (unsigned)R); // p >= R ? p - R : p
			const bool keep = (f < n) && (f >= firstKept);
			const int addr = (int)((p >> 4) * (unsigned)(G * 256) + (unsigned)(ringOff * 16)) + (int)((p & 15u) << 4);
			const int soff = keep ? addr : OOB;
#pragma unroll
			for (int cg = 0; cg < G; cg++) BufStore(srsrc, f32x4{ x[4 * cg], x[4 * cg + 1], x[4 * cg + 2], x[4 * cg + 3] }, soff + cg * 256);
		}

		template <int G>
		__device__ __forceinline__ void PublishFrame(const float (&x)[MAXC], f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int pos0, int R,
			int n, int f)
		{
			PublishLds<G>(x, xb, f);
			StoreRing<G>(x, srsrc, ringOff, pos0, R, n, f);
		}

		// What a layer needs to know about the NEXT layer to request its ring history (PF == 2)
		struct NextHistory
		{
			bool valid;
			int ringOff, dilation, ksize, pos0, R;
		};

		// PF == 2: ring history straight into LDS (buffer_load_dwordx4 ... lds, no VGPRs): lane l's frame (f - shift) of channel group cg
		// lands in hb[cg * 64 + l] -- the same [cg][64] shape as a wave part of the block image, so a tap reads either place through one
		// selected base address.  Always G instructions (predicated through the offset): the VMEM count per layer stays fixed.
		template <int G>
		__device__ __forceinline__ void DmaHistory(f32x4* hb, __amdgpu_buffer_rsrc_t srsrc, int ringOff, int f, int shift, bool valid, int pos0, int R)
		{
			if (NA_ABL & 4) return;
			int base = pos0 - shift; // shift <= R - 128, so one wrap is enough
			if (base < 0) base += R;
			unsigned p = (unsigned)(base + f);
			p = __builtin_elementwise_min(p, p - (unsigned)R); // p >= R ? p - R : p
			const int addr = (int)((p >> 4) * (unsigned)(G * 256) + (unsigned)(ringOff * 16)) + (int)((p & 15u) << 4);
			const int hoff = (valid && f < shift) ? addr : OOB;
#pragma unroll
			for (int cg = 0; cg < G; cg++)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (__attribute__((address_space(3))) void*)(hb + cg * 64), 16, hoff + cg * 256, 0, 0, 0);
		}

		// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt in bits 3:0 and 15:14; expcnt 6:4 and lgkmcnt 11:8 left at "don't wait")
		template <int N>
		__device__ __forceinline__ void WaitVmcnt()
		{
			static_assert(N >= 0 && N < 64, "");
			asm volatile("" ::: "memory");
			__builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
			asm volatile("" ::: "memory");
		}

		// WaveNetLayerT::Process (WaveNet.h:462-494) for one frame per lane
		template <int G, int WPS, int PF, int NW, int HPF>
		__device__ __forceinline__ void LayerFr(const FrStage& sd, const f32x4* wl, CFloat vec, const f32x4* xbCur, f32x4* xbNext,
			__amdgpu_buffer_rsrc_t srsrc, int inPos0, int outPos0, int n, int nSt, int f, int wave, int lane, float cond, float (&xc)[MAXC], float (&hd)[MAXC],
			const f32x4 (&hcur)[HPF][PF == 2 ? 1 : G], bool haveCur, f32x4* hb, const NextHistory& nh, __amdgpu_buffer_rsrc_t lrsrc, bool counted,
			long long* sub = nullptr)
		{
			// trace builds: 7 in-layer shader-clock stamps per (stage, wave), see tools/trace_stage_timeline.py
#ifdef NA_FR_TRACE
#define FR_SUB(i) if (sub != nullptr && lane == 0) sub[i] = (long long)__builtin_readcyclecounter()
#else
#define FR_SUB(i) (void)0
#endif
			FR_SUB(0);
			constexpr int C = 4 * G;
			const int K = sd.ksize;
			const int d = sd.dilation;
			const float* a4 = reinterpret_cast<const float*>(wl) + lane * G; // this lane's A operands, tap stride 64*G floats
			const f32x4* vl = wl + (K + 1) * (16 * G);                       // conv bias | mixin | 1x1 bias (tail of the staged block, see BuildWaveNetPlan)

			// acc = conv bias (:288-289) + W_mix * cond (:471)
			f32x4 acc[G];
#pragma unroll
			for (int og = 0; og < G; og++)
			{
				acc[og] = vl[og] + vl[G + og] * cond;
			}

			// dilated conv (:139-290): tap k reads the frame d*(K-1-k) back; the last tap is the layer input itself (registers).
			// With PF the history of the first HPF taps was loaded during the previous layer (hcur); those taps are peeled so that each
			// names its registers statically.
			FR_SUB(1); // bias / mix-in read
			int kFirst = 0;
			if constexpr (PF == 2)
			{
				// History of tap k was requested into hb[k] while the previous layer ran.  VMEM instructions this wave issued after that
				// request, on every path: the other tap's G, the previous layer's G ring stores, WCOPY weight loads -- all others may stay
				// in flight.  (`counted` is false for the first layer of a run, whose request has a different tail.)
				static_assert(HPF == HPF_LDS, "the LDS history buffers hold HPF_LDS taps");
				constexpr int LATER = 2 * G + StagerWcopy(NW);
#pragma unroll
				for (int k = 0; k < HPF; k++)
				{
					if (k < K - 1)
					{
						const int off = f - d * (K - 1 - k);
						if (counted) WaitVmcnt<LATER>();
						else WaitVmcnt<0>();
						const f32x4* src = (off < 0) ? hb + (k * G) * 64 + lane : xbCur + LdsIdx(off < 0 ? 0 : off, G, 0);
						float x[C];
#pragma unroll
						for (int cg = 0; cg < G; cg++)
						{
							const f32x4 v = src[cg * 64];
							x[4 * cg] = v.x; x[4 * cg + 1] = v.y; x[4 * cg + 2] = v.z; x[4 * cg + 3] = v.w;
						}
						DenseMfma<C, C>(acc, a4 + k * (64 * G), x);
					}
					// hb[k] is free again: request the next layer's tap k (issued even when there is nothing to fetch, see DmaHistory)
					DmaHistory<G>(hb + (k * G) * 64, lrsrc, nh.ringOff, f, nh.dilation * (nh.ksize - 1 - k), nh.valid && k < nh.ksize - 1, nh.pos0, nh.R);
				}
				kFirst = HPF;
			}
			else if constexpr (PF == 1)
			{
#pragma unroll
				for (int k = 0; k < HPF; k++)
				{
					if (k < K - 1)
					{
						const int shift = d * (K - 1 - k);
						float x[C];
						FetchFramePre<G>(x, xbCur, f - shift, hcur[k]);
						DenseMfma<C, C>(acc, a4 + k * (64 * G), x);
					}
				}
				kFirst = HPF;
			}
			for (int k = kFirst; k < K - 1; k++)
			{
				const int shift = d * (K - 1 - k);
				const int lo = wave * 64 - shift;
				float x[C];
				FetchFrame<G>(x, xbCur, srsrc, sd.ring_off, f - shift, lo, lo + 63, inPos0, sd.ring_frames);
				DenseMfma<C, C>(acc, a4 + k * (64 * G), x);
			}
			FR_SUB(2); // shifted taps done
			{
				float x[C];
#pragma unroll
				for (int c = 0; c < C; c++) x[c] = xc[c];
				DenseMfma<C, C>(acc, a4 + (K - 1) * (64 * G), x);
			}
			FR_SUB(3); // last tap done

			// activation (:473-480), head accumulate (:482)
			float z[C];
			if (sd.flags & WN_FLAG_LEAKY) // one uniform branch per layer, not one per channel pair
			{
#pragma unroll
				for (int og = 0; og < G; og++)
				{
					const f32x2 lo2 = LeakyReLU2(f32x2{ acc[og].x, acc[og].y });
					const f32x2 hi2 = LeakyReLU2(f32x2{ acc[og].z, acc[og].w });
					z[4 * og] = lo2.x; z[4 * og + 1] = lo2.y; z[4 * og + 2] = hi2.x; z[4 * og + 3] = hi2.y;
				}
			}
			else if (sd.flags & WN_FLAG_STD_TANH)
			{
#pragma unroll
				for (int og = 0; og < G; og++)
				{
					const f32x2 lo2 = StdTanh2(f32x2{ acc[og].x, acc[og].y });
					const f32x2 hi2 = StdTanh2(f32x2{ acc[og].z, acc[og].w });
					z[4 * og] = lo2.x; z[4 * og + 1] = lo2.y; z[4 * og + 2] = hi2.x; z[4 * og + 3] = hi2.y;
				}
			}
			else
			{
#pragma unroll
				for (int og = 0; og < G; og++)
				{
					const f32x2 lo2 = FastTanh2(f32x2{ acc[og].x, acc[og].y });
					const f32x2 hi2 = FastTanh2(f32x2{ acc[og].z, acc[og].w });
					z[4 * og] = lo2.x; z[4 * og + 1] = lo2.y; z[4 * og + 2] = hi2.x; z[4 * og + 3] = hi2.y;
				}
			}
#pragma unroll
			for (int c = 0; c < C; c++) hd[c] += z[c];

			FR_SUB(4); // activation + head accumulate done
			if (sd.flags & WN_FLAG_NEED_OUTPUT)
			{
				// 1x1 + bias + residual (:486-491)
				f32x4 y[G];
#pragma unroll
				for (int og = 0; og < G; og++)
					y[og] = vl[2 * G + og] + f32x4{ xc[4 * og], xc[4 * og + 1], xc[4 * og + 2], xc[4 * og + 3] };
				DenseMfma<C, C>(y, a4 + K * (64 * G), z);
#pragma unroll
				for (int og = 0; og < G; og++)
				{
					xc[4 * og] = y[og].x; xc[4 * og + 1] = y[og].y; xc[4 * og + 2] = y[og].z; xc[4 * og + 3] = y[og].w;
				}
			}
			FR_SUB(5); // 1x1 done
			// (deferring the ring store to the start of the next layer was tried: no gain, the cost is the store instructions themselves)
			if (PF) PublishFrame<G>(xc, xbNext, srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, (sd.flags & WN_FLAG_PUBLISH) ? nSt : 0, f);
			else if (sd.flags & WN_FLAG_PUBLISH) PublishFrame<G>(xc, xbNext, srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, nSt, f);
			FR_SUB(6); // published
#undef FR_SUB
		}

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
		__device__ __forceinline__ float HeadConvPk(const FrStage& sd, CFloat wpk, const f32x4* xb, __amdgpu_buffer_rsrc_t srsrc, int pos0, int f,
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

		// Stages the NEXT stage's A-operand block into the other LDS weight buffer with LDS-DMA loads (buffer_load_dwordx4 ... lds: lane l's
		// 16 bytes land at ldsBase + 16 l, no VGPRs, no ds_write), issued at the start of a stage -- before the stage's ring stores, gfx950
		// has one vmcnt for loads and stores -- and awaited just before the closing barrier.
		template <int NWAVES>
		struct WeightStager
		{
			static constexpr int NTHREADS = 64 * NWAVES;
			static constexpr int WCOPY = StagerWcopy(NWAVES);

			__device__ __forceinline__ void Begin(f32x4* wlNext, __amdgpu_buffer_rsrc_t wrsrc, const FrStage& sdn, int waveAll)
			{
				if (NA_ABL & 16) return;
				const int nextF4 = sdn.a4_floats / 4;
				const int lane = (int)threadIdx.x & 63;
#pragma unroll
				for (int c = 0; c < WCOPY; c++)
				{
					const int i0 = c * NTHREADS + waveAll * 64; // first float4 of this wave's 1 KB slice (wave-uniform)
					const int i = i0 + lane;
					__builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(wlNext + i0), 16, (i < nextF4) ? (sdn.a4_off / 4 + i) * 16 : OOB, 0, 0, 0);
				}
			}

			// LATER = number of VMEM instructions this wave issued after Begin() on every path (they may stay in flight), or 0
			template <int LATER>
			__device__ __forceinline__ void End(f32x4* wlNext, __amdgpu_buffer_rsrc_t wrsrc, const FrStage& sdn)
			{
				if (NA_ABL & 16) return;
				const int nextF4 = sdn.a4_floats / 4;
				for (int i = (int)threadIdx.x + WCOPY * NTHREADS; i < nextF4; i += NTHREADS) wlNext[i] = BufLoad(wrsrc, (sdn.a4_off / 4 + i) * 16); // oversized (A2 K=15)
				// the DMA data must be in LDS before the closing barrier lets other waves read it; the workgroup release fence only covers
				// lgkmcnt, so wait on vmcnt here (gfx9 s_waitcnt: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "don't wait")
				__builtin_amdgcn_s_waitcnt((LATER & 15) | ((LATER >> 4) << 14) | (7 << 4) | (15 << 8));
			}
		};

		// tuning aid (see NA_DebugSetTraceBuffer): trace[((stage * 4 + point) * waves) + wave] = shader clock, workgroup `traceBlock` only.
		// Compiled in only with -DNA_FR_TRACE (make SUFFIX=_trace EXTRA=-DNA_FR_TRACE): a conditional store inside the layer loop makes
		// the compiler's vmcnt waits conservative.
#ifdef NA_FR_TRACE
#define FR_TRACE(point) \
	if (trace != nullptr && (int)blockIdx.x == traceBlock && lane == 0) trace[((s * 4 + (point)) * (WPS * SPB)) + waveAll] = (long long)__builtin_readcyclecounter()
#else
#define FR_TRACE(point) (void)0
#endif

		// Head conv whose whole reach (K-1)*dil fits in 16 frames (every official A2 head: K = 16, dil = 1): the 16 frames before the
		// block start were parked in `pad` ([cg][16] float4, see OtherStage), so every tap reads LDS only -- no ring loads, no waits per tap.
		template <int G>
		__device__ __forceinline__ float HeadConvLds(const FrStage& sd, CFloat wpk, const f32x4* xb, const f32x4* pad, int f, float bias)
		{
			constexpr int C = 4 * G;
			float acc = bias;
			CFloat w = wpk + sd.pk_conv_off;
			for (int k = 0; k < sd.ksize; k++)
			{
				const int off = f - sd.dilation * (sd.ksize - 1 - k);
#pragma unroll
				for (int cg = 0; cg < G; cg++)
				{
					const f32x4* src = (off < 0) ? pad + cg * 16 + (off + 16) : xb + LdsIdx(off < 0 ? 0 : off, G, cg);
					const f32x4 v = *src;
					acc = __builtin_fmaf(w[k * C + 4 * cg + 0], v.x, acc);
					acc = __builtin_fmaf(w[k * C + 4 * cg + 1], v.y, acc);
					acc = __builtin_fmaf(w[k * C + 4 * cg + 2], v.z, acc);
					acc = __builtin_fmaf(w[k * C + 4 * cg + 3], v.w, acc);
				}
			}
			return acc;
		}

		// everything a stage needs that does not change from stage to stage
		struct FrCtx
		{
			const WnStage* __restrict__ stages;
			int nstages;
			f32x4* wbuf;   // [2][maxA4F4] staged A images
			int maxA4F4;
			__amdgpu_buffer_rsrc_t wrsrc; // wpk as a buffer (weight staging)
			CFloat wvec;   // wpack (bias vectors of the non-layer stages), scalar loads
			CFloat wpk;    // wpk, scalar loads (head weights)
			f32x4* xbuf;   // this stream's [2][NTB*64] block images
			f32x4* hbuf;   // PF == 2: this WAVE's [HPF_LDS][4][64] history buffers
			__amdgpu_buffer_rsrc_t srsrc; // this stream's state
			__amdgpu_buffer_rsrc_t lrsrc; // = srsrc (tuning builds: NA_ABL & 128 redirects the history loads)
			int myPos;     // lane r: write cursor of ring r
			int n, nSt;    // frames in the block; frames this wave may store (0 for a shadow wave)
			int f, wave, waveAll, lane;
			float cond;
			float* __restrict__ out;
			size_t outBase;
			float headScale;
			long long* __restrict__ trace;
			int traceBlock;
		};

		// One non-layer stage (rechannel / array link / head), including the staging of the next stage's weights and the closing barrier.
		template <int WPS, int SPB, bool HEADS>
		__device__ __forceinline__ void OtherStage(const FrCtx& cx, int& s, FrStage& sd, const FrStage& sdn, int& cur, float (&xc)[MAXC], float (&hd)[MAXC])
		{
			constexpr int NTB = WPS * 4;
			const int lane = cx.lane, waveAll = cx.waveAll, f = cx.f;
			long long* __restrict__ trace = cx.trace;
			const int traceBlock = cx.traceBlock;
			(void)waveAll; (void)trace; (void)traceBlock;
			FR_TRACE(0);
			const f32x4* wl = cx.wbuf + (s & 1) * cx.maxA4F4;
			f32x4* wlNext = cx.wbuf + ((s + 1) & 1) * cx.maxA4F4;
			WeightStager<WPS * SPB> stager;
			stager.Begin(wlNext, cx.wrsrc, sdn, cx.waveAll);
			const int outPos0 = (sd.out_ring_id >= 0) ? __builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id) : 0;
			const int inPos0 = (sd.ring_id >= 0) ? __builtin_amdgcn_readlane(cx.myPos, sd.ring_id) : 0;
			f32x4* xbNext = cx.xbuf + (cur ^ 1) * (NTB * 64);
			CFloat vec = cx.wvec + sd.vec_off * 4; // [0..15] conv/dense bias, [48..63] aux

			if (sd.type == WN_ST_RECHANNEL_COND)
			{
#pragma unroll
				for (int c = 0; c < MAXC; c++) xc[c] = vec[48 + c] * cx.cond; // :637 with InputSize == 1
				PublishAny(sd.out_G, xc, xbNext, cx.srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, cx.nSt, f);
				cur ^= 1;
			}
			else if (sd.type == WN_ST_ARRAY_LINK)
			{
				// previous array's headRechannel (K=1, :658-660) and this array's rechannel (:637); weights padded to 16x16
				f32x4 hn[4], xn[4];
#pragma unroll
				for (int og = 0; og < 4; og++)
				{
					hn[og] = (sd.flags & WN_FLAG_BIAS) ? f32x4{ vec[4 * og], vec[4 * og + 1], vec[4 * og + 2], vec[4 * og + 3] } : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
					xn[og] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				}
				DenseMfma<MAXC, MAXC>(hn, reinterpret_cast<const float*>(wl) + lane * 4, hd);
				DenseMfma<MAXC, MAXC>(xn, reinterpret_cast<const float*>(wl) + 256 + lane * 4, xc);
#pragma unroll
				for (int og = 0; og < 4; og++)
				{
					hd[4 * og] = hn[og].x; hd[4 * og + 1] = hn[og].y; hd[4 * og + 2] = hn[og].z; hd[4 * og + 3] = hn[og].w;
					xc[4 * og] = xn[og].x; xc[4 * og + 1] = xn[og].y; xc[4 * og + 2] = xn[og].z; xc[4 * og + 3] = xn[og].w;
				}
				PublishAny(sd.out_G, xc, xbNext, cx.srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, cx.nSt, f);
				cur ^= 1;
			}
			else if (HEADS && sd.type == WN_ST_HEAD_DENSE_OUT)
			{
				float o = (sd.flags & WN_FLAG_BIAS) ? vec[0] : 0.0f;
				CFloat wh = cx.wpk + ((CInt)(const int*)(cx.stages + s))[STAGE_HOT_INTS]; // pk_w1_off = first cold field (static_assert below)
#pragma unroll
				for (int c = 0; c < MAXC; c++) o = __builtin_fmaf(wh[c], hd[c], o);
				if (f < cx.nSt) cx.out[cx.outBase + f] = cx.headScale * o; // :793-798
			}
			else if (HEADS) // WN_ST_HEAD_CONV_OUT
			{
				PublishAny(sd.out_G, hd, xbNext, cx.srsrc, sd.out_ring_off, outPos0, sd.out_ring_frames, cx.nSt, f);
				cur ^= 1;
				// short reach: park the 16 frames before the block start next to the image (the other block buffer is dead by now)
				const bool shortReach = sd.dilation * (sd.ksize - 1) <= 16;
				f32x4* pad = cx.xbuf + (cur ^ 1) * (NTB * 64); // = the buffer the last layer read from
				if (shortReach && cx.wave == 0 && lane < 16)
				{
					const int R = sd.ring_frames;
					int p = inPos0 - 16 + lane;
					if (p < 0) p += R;
					for (int cg = 0; cg < sd.G; cg++) pad[cg * 16 + lane] = BufLoad(cx.srsrc, (sd.ring_off + TileIdx(p, sd.G, cg)) * 16);
				}
				BlockBarrier<WPS * SPB>();
				const float bias = (sd.flags & WN_FLAG_BIAS) ? vec[0] : 0.0f;
				float o;
				if (shortReach)
				{
					if (sd.G == 4) o = HeadConvLds<4>(sd, cx.wpk, xbNext, pad, f, bias);
					else if (sd.G == 3) o = HeadConvLds<3>(sd, cx.wpk, xbNext, pad, f, bias);
					else if (sd.G == 2) o = HeadConvLds<2>(sd, cx.wpk, xbNext, pad, f, bias);
					else o = HeadConvLds<1>(sd, cx.wpk, xbNext, pad, f, bias);
				}
				else if (sd.G == 4) o = HeadConvPk<4>(sd, cx.wpk, xbNext, cx.srsrc, inPos0, f, cx.wave, bias);
				else if (sd.G == 3) o = HeadConvPk<3>(sd, cx.wpk, xbNext, cx.srsrc, inPos0, f, cx.wave, bias);
				else if (sd.G == 2) o = HeadConvPk<2>(sd, cx.wpk, xbNext, cx.srsrc, inPos0, f, cx.wave, bias);
				else o = HeadConvPk<1>(sd, cx.wpk, xbNext, cx.srsrc, inPos0, f, cx.wave, bias);
				if (f < cx.nSt) cx.out[cx.outBase + f] = cx.headScale * o;
			}
			FR_TRACE(1);
			stager.template End<0>(wlNext, cx.wrsrc, sdn);
			FR_TRACE(2);
			BlockBarrier<WPS * SPB>();
			FR_TRACE(3);
			sd = sdn;
			s++;
		}

		// A run of consecutive WaveNet layer stages with the same channel-group count G.  With `pre`, sd is the rechannel / array-link
		// stage in front of the run and sdFirst its first layer: the ring history of that layer is requested BEFORE the pre-stage
		// computes, so its HBM latency hides behind it (the per-frame state stays in registers typed by G either way).
		template <int G, int WPS, int PF, int SPB, int HPF>
		__device__ __forceinline__ void RunLayers(const FrCtx& cx, int& s, FrStage& sd, const FrStage& sdFirst, bool pre, int& cur, float (&xc)[MAXC], float (&hd)[MAXC])
		{
			constexpr int NTB = WPS * 4;
			const int lane = cx.lane, waveAll = cx.waveAll, f = cx.f;
			long long* __restrict__ trace = cx.trace;
			const int traceBlock = cx.traceBlock;
			(void)waveAll; (void)trace; (void)traceBlock; (void)lane;
			// history of the first HPF taps of the current layer; requested here for the first layer of the run, afterwards one layer ahead
			// (PF == 1: into registers; PF == 2: into this wave's LDS history buffers)
			f32x4 hcur[HPF][PF == 2 ? 1 : G];
			f32x4* hb = cx.hbuf;
#pragma unroll
			for (int t = 0; t < HPF; t++)
			{
				const int shift0 = sdFirst.dilation * (sdFirst.ksize - 1 - t);
				const int pos0 = __builtin_amdgcn_readlane(cx.myPos, sdFirst.ring_id);
				if constexpr (PF == 2) DmaHistory<G>(hb + (t * G) * 64, cx.lrsrc, sdFirst.ring_off, f, shift0, t < sdFirst.ksize - 1, pos0, sdFirst.ring_frames);
				else LoadHistory<G>(hcur[t], cx.lrsrc, sdFirst.ring_off, f, shift0, PF && t < sdFirst.ksize - 1, pos0, sdFirst.ring_frames);
			}
			if (pre) OtherStage<WPS, SPB, false>(cx, s, sd, sdFirst, cur, xc, hd);
			const bool haveCur = true;
			bool counted = false; // the first layer's history request is followed by an irregular number of VMEM instructions
			do
			{
				FR_TRACE(0);
				FrStage sdn = sd;
				sdn.a4_floats = 0;
				sdn.type = -1;
				if (s + 1 < cx.nstages) sdn = LoadStage(cx.stages, s + 1);
				const f32x4* wl = cx.wbuf + (s & 1) * cx.maxA4F4;
				f32x4* wlNext = cx.wbuf + ((s + 1) & 1) * cx.maxA4F4;
				WeightStager<WPS * SPB> stager;
				stager.Begin(wlNext, cx.wrsrc, sdn, cx.waveAll);
				const int outPos0 = (sd.out_ring_id >= 0) ? __builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id) : 0;
				const int inPos0 = __builtin_amdgcn_readlane(cx.myPos, sd.ring_id);
				// history of the NEXT layer's first HPF taps: issued at the start of this layer (before its ring stores), consumed a layer
				// later.  Always HPF*G loads, predicated through the offset, so the VMEM count per layer is the same on every path.
				f32x4 hnext[HPF][PF == 2 ? 1 : G];
				const bool haveNext = PF && (s + 1 < cx.nstages) && sdn.type == WN_ST_LAYER && sdn.G == G;
				const int nextPos0 = __builtin_amdgcn_readlane(cx.myPos, haveNext ? sdn.ring_id : 0);
				if constexpr (PF == 1)
				{
#pragma unroll
					for (int t = 0; t < HPF; t++)
					{
						const int shiftN = sdn.dilation * (sdn.ksize - 1 - t);
						LoadHistory<G>(hnext[t], cx.lrsrc, sdn.ring_off, f, shiftN, haveNext && t < sdn.ksize - 1, nextPos0, sdn.ring_frames);
					}
				}
				const NextHistory nh = { haveNext, sdn.ring_off, sdn.dilation, sdn.ksize, nextPos0, sdn.ring_frames };

				LayerFr<G, WPS, PF, WPS * SPB, HPF>(sd, wl, cx.wvec + sd.vec_off * 4, cx.xbuf + cur * (NTB * 64), cx.xbuf + (cur ^ 1) * (NTB * 64), cx.srsrc, inPos0,
					outPos0, cx.n, cx.nSt, f, cx.wave, lane, cx.cond, xc, hd, hcur, haveCur, hb, nh, cx.lrsrc, counted,
#ifdef NA_FR_TRACE
					(trace != nullptr && (int)blockIdx.x == traceBlock) ? trace + ((cx.nstages + 1) * 4 + s * 8) * (WPS * SPB) + waveAll * 8 : nullptr
#else
					nullptr
#endif
				);
				counted = true;
				if constexpr (PF == 1)
				{
#pragma unroll
					for (int t = 0; t < HPF; t++)
#pragma unroll
						for (int cg = 0; cg < G; cg++) hcur[t][cg] = hnext[t][cg];
				}
				if (sd.flags & WN_FLAG_PUBLISH) cur ^= 1;
				FR_TRACE(1);
				stager.template End<PF ? (HPF + 1) * G : 0>(wlNext, cx.wrsrc, sdn); // PF: HPF*G history loads + G ring stores follow Begin() on every path
				FR_TRACE(2);
				BlockBarrier<WPS * SPB>();
				FR_TRACE(3);
				sd = sdn;
				s++;
			} while (s < cx.nstages && sd.type == WN_ST_LAYER && sd.G == G);
		}

		// grid = active streams of one model / SPB; workgroup = SPB streams x WPS waves of 64 frames (WPS = 2: 128-frame blocks).  The SPB
		// streams of a workgroup share one staged copy of the weights.
		// Per stage: issue the loads of the NEXT stage's A-operand block first (before this stage's ring stores: gfx950 has one
		// vmcnt for loads and stores), compute, park the block in the other LDS weight buffer, meet at an LDS-only barrier.
		// dynamic LDS: xbuf[SPB][2][WPS*4 tiles * 64] float4 | wbuf[2][maxA4Floats/4] float4
		// per model group of one (possibly fused) launch; passed by value in the kernarg segment
		struct FrGroupArgs
		{
			const WnStage* stages;
			const float* wpack;
			const float* wpk;
			const int* ringFrames;
			f32x4* state;
			const int* slots; // nullptr: contiguous, stream i uses slot0 + i / row0 + i
			const int* rows;
			int nstages, nrings, stateF4, wpkFloats;
			float headScale;
			int numStreams, slot0, row0;
			int maxKsize;   // largest conv kernel of the model
			int firstBlock; // workgroups [firstBlock, next group's firstBlock) belong to this group
		};

		struct FrLaunchArgs
		{
			FrGroupArgs g[WN_FRAME_MAX_GROUPS];
			int numGroups;
		};

		template <int WPS, int PF, int SPB>
		__global__ void __launch_bounds__(64 * WPS * SPB) WaveNetFrameKernel(const FrLaunchArgs args, int maxA4F4, const float* __restrict__ in,
			float* __restrict__ out, long inStride, long outStride, int n, long long* __restrict__ trace, int traceBlock)
		{
			// which model group this workgroup serves (heterogeneous batches run as ONE launch: no stream fork/join, and the
			// workgroups of all architectures share the chip)
			int gi = 0;
			for (int i = 1; i < args.numGroups; i++)
				if ((int)blockIdx.x >= args.g[i].firstBlock) gi = i;
			const FrGroupArgs& ga = args.g[gi];
			const WnStage* __restrict__ stages = ga.stages;
			const float* __restrict__ wpack = ga.wpack;
			const float* __restrict__ wpkGlobal = ga.wpk;
			const int* __restrict__ ringFrames = ga.ringFrames;
			f32x4* __restrict__ state = ga.state;
			const int* __restrict__ slots = ga.slots;
			const int* __restrict__ rows = ga.rows;
			const int nstages = ga.nstages, nrings = ga.nrings, stateF4 = ga.stateF4, wpkFloats = ga.wpkFloats;
			const float headScale = ga.headScale;
			const int numStreams = ga.numStreams, slot0 = ga.slot0, row0 = ga.row0;
			const int groupBlock = (int)blockIdx.x - ga.firstBlock;
			constexpr int NTB = WPS * 4; // tiles in one stream's block
			extern __shared__ __attribute__((aligned(16))) char smem[];
			constexpr int NTHREADS = 64 * WPS * SPB;

			const int lane = threadIdx.x & 63;
			const int waveAll = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef NA_FR_TRACE
			const long long tEntry = (long long)__builtin_readcyclecounter();
#endif
			const int sub = waveAll / WPS;  // stream within the workgroup
			const int wave = waveAll % WPS; // 64-frame part of the stream's block
			const int f = wave * 64 + lane; // this lane's frame in the block
			f32x4* xbuf = reinterpret_cast<f32x4*>(smem) + sub * (2 * NTB * 64);  // [2][NTB*64] per stream
			f32x4* wbuf = reinterpret_cast<f32x4*>(smem) + SPB * (2 * NTB * 64);  // [2][maxA4F4], shared

			// a partial last workgroup: the surplus waves shadow the last stream (they must keep staging weights and meeting barriers)
			// but write nothing
			int sidx = groupBlock * SPB + sub;
			const bool live = sidx < numStreams;
			if (!live) sidx = numStreams - 1;
			const int nSt = live ? n : 0;
			const int slot = slots ? slots[sidx] : slot0 + sidx;
			const int row = slots ? rows[sidx] : row0 + sidx;
			f32x4* st = state + (size_t)slot * (size_t)stateF4;
			int* header = reinterpret_cast<int*>(st);
			const int myPos = header[lane]; // lane r holds the write cursor of ring r
			const float cond = (f < n) ? in[(size_t)row * inStride + f] : 0.0f; // WaveNet.h:770 (input -> condition)
			float xc[MAXC], hd[MAXC];
#pragma unroll
			for (int c = 0; c < MAXC; c++)
			{
				xc[c] = 0.0f;
				hd[c] = 0.0f; // WaveNet.h:772 headArray.SetZero()
			}

			FrCtx cx;
			cx.stages = stages;
			cx.nstages = nstages;
			cx.wbuf = wbuf;
			cx.maxA4F4 = maxA4F4;
			cx.wrsrc = MakeRsrc(wpkGlobal, (unsigned)wpkFloats * 4u);
			cx.wvec = (CFloat)wpack;
			cx.wpk = (CFloat)wpkGlobal;
			cx.xbuf = xbuf;
			cx.hbuf = reinterpret_cast<f32x4*>(smem) + SPB * (2 * NTB * 64) + 2 * maxA4F4 + waveAll * (HPF_LDS * 4 * 64); // after wbuf, PF == 2 only
			cx.srsrc = MakeRsrc(st, (unsigned)stateF4 * 16u);
			cx.lrsrc = (NA_ABL & 128) ? MakeRsrc(state + (size_t)(groupBlock & 7) * (size_t)stateF4, (unsigned)stateF4 * 16u) : cx.srsrc; // 128: history loads hit 8 hot slots
			cx.myPos = myPos;
			cx.n = n;
			cx.nSt = nSt;
			cx.f = f;
			cx.wave = wave;
			cx.waveAll = waveAll;
			cx.lane = lane;
			cx.cond = cond;
			cx.out = out;
			cx.outBase = (size_t)row * outStride;
			cx.headScale = headScale;
			cx.trace = trace;
			cx.traceBlock = traceBlock;

			FrStage sd = LoadStage(stages, 0);
			for (int i = threadIdx.x; i < sd.a4_floats / 4; i += NTHREADS) wbuf[i] = BufLoad(cx.wrsrc, (sd.a4_off / 4 + i) * 16);
			BlockBarrier<WPS * SPB>();

			int cur = 0;
			int s = 0;
			while (s < nstages)
			{
				// Hot path: runs of WaveNet layers with the same channel-group count execute in their own tight loop, so the per-frame
				// state (xc, hd) stays in fixed registers across layers (no phi copies at the stage-type branches).  The rechannel /
				// array-link stage in front of a run executes inside it (see RunLayers).
				FrStage sdn = sd;
				sdn.a4_floats = 0;
				sdn.type = -1;
				if (s + 1 < nstages) sdn = LoadStage(stages, s + 1);
				const bool pre = (sd.type == WN_ST_RECHANNEL_COND || sd.type == WN_ST_ARRAY_LINK) && sdn.type == WN_ST_LAYER;
				if (pre || sd.type == WN_ST_LAYER)
				{
					const FrStage& first = pre ? sdn : sd;
					// narrow layers of a model with kernels larger than 3 (A2): request 5 taps ahead instead of 2
					const bool wide = PF == 1 && ga.maxKsize > 3;
					if (first.G == 4) RunLayers<4, WPS, PF, SPB, HPF_NARROW>(cx, s, sd, first, pre, cur, xc, hd);
					else if (first.G == 3) RunLayers<3, WPS, PF, SPB, HPF_NARROW>(cx, s, sd, first, pre, cur, xc, hd);
					else if (first.G == 2)
					{
						if (wide) RunLayers<2, WPS, PF, SPB, PF == 1 ? HPF_WIDE : HPF_NARROW>(cx, s, sd, first, pre, cur, xc, hd);
						else RunLayers<2, WPS, PF, SPB, HPF_NARROW>(cx, s, sd, first, pre, cur, xc, hd);
					}
					else
					{
						if (wide) RunLayers<1, WPS, PF, SPB, PF == 1 ? HPF_WIDE : HPF_NARROW>(cx, s, sd, first, pre, cur, xc, hd);
						else RunLayers<1, WPS, PF, SPB, HPF_NARROW>(cx, s, sd, first, pre, cur, xc, hd);
					}
					continue;
				}
				OtherStage<WPS, SPB, true>(cx, s, sd, sdn, cur, xc, hd);
			}

#ifdef NA_FR_TRACE
			{
				const int s = nstages; // slot after the last stage: [0] = kernel entry, [1] = after the stage loop
				if (trace != nullptr && (int)blockIdx.x == traceBlock && lane == 0)
				{
					trace[((s * 4 + 0) * (WPS * SPB)) + waveAll] = tEntry;
					trace[((s * 4 + 1) * (WPS * SPB)) + waveAll] = (long long)__builtin_readcyclecounter();
				}
			}
#endif
			// advance every ring cursor by n (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
			if (wave == 0 && live && lane < nrings)
			{
				const int R = ringFrames[lane];
				int p = myPos + n;
				if (p >= R) p -= R;
				header[lane] = p;
			}
		}

		template <int WPS, int PF, int SPB>
		static hipError_t Launch(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream)
		{
			// stride of the two LDS weight buffers: the LDS-DMA staging always writes WCOPY * NTHREADS float4 slots (zeros past the block)
			constexpr int NT = 64 * WPS * SPB;
			FrLaunchArgs args = {};
			args.numGroups = numGroups;
			int maxA4F4 = WeightStager<WPS * SPB>::WCOPY * NT;
			int blocks = 0;
			for (int i = 0; i < numGroups; i++)
			{
				const WnFrameGroup& g = groups[i];
				const WnModelDev& m = *g.model;
				FrGroupArgs& a = args.g[i];
				a.stages = m.stages; a.wpack = m.wpack; a.wpk = m.wpk; a.ringFrames = m.ring_frames;
				a.state = reinterpret_cast<f32x4*>(g.state); a.slots = g.slots; a.rows = g.rows;
				a.nstages = m.nstages; a.nrings = m.nrings; a.stateF4 = m.state_f4; a.wpkFloats = m.wpk_floats;
				a.headScale = m.head_scale;
				a.numStreams = g.numStreams; a.slot0 = g.slot0; a.row0 = g.row0;
				a.maxKsize = m.max_ksize;
				a.firstBlock = blocks;
				blocks += (g.numStreams + SPB - 1) / SPB;
				maxA4F4 = std::max(maxA4F4, (m.max_a4_floats + 3) / 4);
			}
			const size_t lds = (size_t)SPB * 2 * WPS * 4 * 64 * 16 + (size_t)2 * maxA4F4 * 16 + (PF == 2 ? (size_t)WPS * SPB * HPF_LDS * 4 * 64 * 16 : 0);
			if (lds > 160 * 1024) return hipErrorInvalidValue;
			auto kernel = WaveNetFrameKernel<WPS, PF, SPB>;
			if (lds > 64 * 1024)
			{
				// per instantiation and device: the whole LDS of a CU once (granting it does not change what a launch uses)
				static PerDeviceOnce attr;
				const hipError_t e = attr.Run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
				if (e != hipSuccess) return e;
			}
			hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * WPS * SPB), lds, stream, args, maxA4F4, in, out, inStride, outStride, n,
				GetWaveNetTraceBuffer(), Tuning::Get().traceBlock);
			return hipGetLastError();
		}
	}

	hipError_t LaunchWaveNetFrameFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream)
	{
		if (n <= 0 || numGroups <= 0) return hipSuccess;
		if (n > WN_MAX_FRAMES || numGroups > WN_FRAME_MAX_GROUPS) return hipErrorInvalidValue;
		int total = 0;
		size_t ldsWeights = 0;
		for (int i = 0; i < numGroups; i++)
		{
			if (groups[i].numStreams <= 0) return hipErrorInvalidValue;
			total += groups[i].numStreams;
			ldsWeights = std::max(ldsWeights, (size_t)2 * ((groups[i].model->max_a4_floats + 3) / 4) * 16);
		}
		const int prefetch = Tuning::Get().frPrefetch; // tuning knob: 0 none, 1 history prefetch into registers, 2 into LDS (LDS-DMA)
		const int spbEnv = Tuning::Get().frSpb;            // tuning knob: streams per workgroup (1, 2, 4)
		// streams per workgroup: two streams share one staged copy of the weights once there are enough streams to cover all 256 CUs
		// (measured 1024 x Standard: SPB 1 / 2 / 4 = 61.5 / 61.3 / 65.0 us -- at 4 the 8-wave barrier skew eats the saving)
		const int spb = spbEnv > 0 ? spbEnv : (total >= 512 ? 2 : 1);
		if (n > 64)
		{
			if (!prefetch) return fr::Launch<2, 0, 1>(groups, numGroups, in, out, inStride, outStride, n, stream);
			if (spb >= 4 && ldsWeights + 4 * 16384 <= 160 * 1024) return fr::Launch<2, 1, 4>(groups, numGroups, in, out, inStride, outStride, n, stream);
			if (prefetch == 2) return spb >= 2 ? fr::Launch<2, 2, 2>(groups, numGroups, in, out, inStride, outStride, n, stream)
									   : fr::Launch<2, 2, 1>(groups, numGroups, in, out, inStride, outStride, n, stream);
			if (spb >= 2) return fr::Launch<2, 1, 2>(groups, numGroups, in, out, inStride, outStride, n, stream);
			return fr::Launch<2, 1, 1>(groups, numGroups, in, out, inStride, outStride, n, stream);
		}
		return fr::Launch<1, 0, 1>(groups, numGroups, in, out, inStride, outStride, n, stream);
	}

	hipError_t LaunchWaveNetFrame(const WnModelDev& m, float* state, const int* slots, const int* rows, int numStreams, const float* in, float* out,
		long inStride, long outStride, int n, hipStream_t stream, int slot0, int row0)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		WnFrameGroup g = { &m, state, slots, rows, numStreams, slot0, row0 };
		return LaunchWaveNetFrameFused(&g, 1, in, out, inStride, outStride, n, stream);
	}
}
