// wavenet_split_kernels.hip -- the shipped WaveNet block kernel for gfx950: every mat-mul of the path on v_mfma_f32_16x16x32_f16,
// f32 accuracy from a three-product f16 split.
//
// Reference arithmetic being replaced (one stream, per-sample loops on the CPU): WaveNetModelT / LayerArrayT / LayerT::Process,
// Conv1DT::Process, DenseLayerT::Process (NeuralAudio/WaveNet.h:768-799, 632-661, 462-494, 139-290, 336-383), FastMath / StdMath
// activations (Activation.h:20-118).
//
// Why f16 MFMA for an f32 path: f32-input MFMA runs at the f32 VECTOR rate on gfx950 (32 MAC / cycle / SIMD) and, measured in
// round 1, shares its issue slots with the VALU -- the frame kernel (wavenet_frame_kernels.hip) spends 23 of its 60 us issuing them.
// v_mfma_f32_16x16x32_f16 runs 16x faster (8192 MAC in 17 cycles), overlaps with plain VALU work of the same and of other waves
// (tools/microbench/mfma_f16_valu_mix.hip) and accumulates in f32.  Every value v is carried as h = f16(v), l = f16(v - h) (22
// mantissa bits) and W*x is evaluated as Wh*xh + Wh*xl + Wl*xh; the dropped Wl*xl term is 2^-22 relative.  Measured against a
// float64 evaluation of the real models the split is as accurate as f32 arithmetic (1.1e-7 vs 1.3e-7 RMS on Standard).
//
// Mapping (16x16x32: D[16 rows][16 cols] += A[16][32] * B[32][16]; lane = (j = lane & 15, q = lane >> 4)):
//   * columns = 16 consecutive frames (a tile), rows = output channels, k = input channels x split parts;
//   * B operand of lane (j, q) = 8 halfs = ONE "split quad": channels 4cg..4cg+3 of the lane's frame as [h0 h1 h2 h3 l0 l1 l2 l3].
//     That is also the storage format of the activations (LDS block image and HBM rings, 16 bytes per 4 channels -- the same bytes as
//     f32), so a conv tap is ONE 16-byte load straight into the MFMA operand, with no conversion;
//   * D of lane (j, q) = rows 4q..4q+3 of column j: 4 channels of the lane's frame in f32 -> activation, split (8 VALU per quad) and
//     the result is again the B operand of the next mat-mul: nothing is ever shuffled between lanes;
//   * A operands (weights, host-packed, wavenet_plan.cpp) are staged global -> LDS one stage ahead by LDS-DMA and read with one
//     ds_read_b128 per MFMA, shared by all tiles of the wave;
//   * narrow layers: with C <= 8 (<= 4) channels two (four) tiles share one MFMA through block-diagonal A operands ("lane mode"
//     Gp = channel groups per tile = 4, 2, 1), so the VALU never sees padding lanes: an 8-channel layer costs half of a 16-channel one;
//   * bias and the input mix-in ride in the MFMA too: an "aux" operand (cond, 1, 0, 0) per frame against weights (w_mix, bias).
// A wave owns T consecutive tiles of one stream's block; a workgroup = SPB streams x (8 / T) waves sharing one staged copy of the
// weights; one LDS-only barrier per layer (the dependency is causal).
#include "device_once.h"
#include "tuning.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

#include "wavenet_split_dev.h"

namespace na
{
	namespace sp
	{
		// tuning aid (make SUFFIX=_trace EXTRA=-DNA_SP_TRACE, tools/trace_split_timeline.py): shader-clock stamps of one workgroup,
		// trace[(stage * 8 + point) * waves + wave]; the scheduling barriers pin the stamp between the phases (and cost a little overlap)
#ifdef NA_SP_TRACE
#define SP_STAMP(point) do { __builtin_amdgcn_sched_barrier(0); if (cx.trace != nullptr && cx.lane == 0) cx.trace[((s * 8 + (point)) * cx.nwaves) + cx.waveAll] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SP_STAMP(point) (void)0
#endif
		struct Stage
		{
			int type, flags, G, Gp, ksize, dilation, ring_off, ring_frames, ring_id, out_ring_off, out_ring_frames, out_ring_id, out_G, a_off, a_ops, reserved;
		};
		static_assert(sizeof(Stage) == sizeof(WnSplitStage), "Stage mirrors WnSplitStage");

		__device__ __forceinline__ Stage LoadStage(const WnSplitStage* __restrict__ stages, int s)
		{
			Stage sd;
			CInt src = (CInt)(const int*)(stages + s);
			int* dst = reinterpret_cast<int*>(&sd);
#pragma unroll
			for (int i = 0; i < 16; i++) dst[i] = src[i];
			return sd;
		}

		// ---- lane geometry ----------------------------------------------------------------------------------------------------
		// Mode GP (channel groups per tile: 4, 2, 1): P = 4 / GP tiles share one MFMA "set"; a wave of T tiles has S sets.
		// lane (j, q): tile slot p = q / GP, channel group cg = q % GP; frame of set s: f = F0 + 16 * (P * s + p) + j.
		template <int GP, int T>
		struct Geo
		{
			static constexpr int P = 4 / GP;
			static constexpr int S = (T + P - 1) / P;
			static constexpr bool PARTIAL = (T % P) != 0; // some tile slots of the (only) set lie beyond the wave's tiles
		};

		// everything a stage needs that does not change from stage to stage
		struct Ctx
		{
			const WnSplitStage* __restrict__ stages;
			int nstages;
			const u32x4* wbuf;            // [2][wstride] staged A operands (quads)
			int wstride;
			__amdgpu_buffer_rsrc_t wrsrc; // split weight image (staging source)
			u32x4* img;                   // this stream's [2][maxG][PLANE] block images (quads)
			int imgStride;                // quads per image = maxG * PLANE
			const u32x4* idop;            // identity A operand [64 lanes] (head accumulation rides on the matrix pipe)
			const u32x4* auxq;            // this stream's aux operands in LDS: [cond_h, 1, cond_l, 1, cond_h, 0, 0, 0] per frame [FRAMES]
			// packed launches (PK: several real streams as the channel groups of this virtual stream, wavenet_plan.cpp PackWaveNetDesc):
			const u32x2* auxp;            // [pack][FRAMES] x 8 bytes: the first two dwords of stream q's aux operands (the rest follows from them)
			int pack;                     // real streams in this virtual stream
			long outRow[4];               // their output rows (float offsets into `out`), -1: slot not in use
			__amdgpu_buffer_rsrc_t srsrc; // this stream's state
			int myPos;                    // lane r: write cursor of ring r
			int n, nSt;                   // frames in the block; frames this wave may store (0 for a shadow wave)
			int F0;                       // first frame of this wave
			int lane, j, q, waveAll;
			int saturate;                 // wave-uniform: SplitQuadSat instead of SplitQuad (GroupArgs::saturate)
			float* __restrict__ out;
			size_t outBase;
			float headScale;
			long long* trace; // nullptr unless this is the traced workgroup of a trace build
			int nwaves;
		};

		// ring address (bytes) of channel group `cg` of ring position p, frame-major image: (p * G + cg) * 16
		// (v_mad_u32_u24: full rate; a plain 32-bit multiply is a quarter-rate instruction and rings are far smaller than 2^24 quads)
		__device__ __forceinline__ int RingByte(int ringOff, int G, int p, int cg) { return (int)(__umul24((unsigned)p, (unsigned)(G * 16)) + (unsigned)((ringOff + cg) * 16)); }

		// History part of one tap for one set: the split quad of frame (f - shift) for the lanes with f < shift (frames before the block
		// start), from the ring whose write cursor is pos0.  Always one load per call, predicated through the offset, so that the number
		// of VMEM operations per layer is the same on every path (the compiler can then place counted vmcnt waits).
		__device__ __forceinline__ u32x4 LoadHistory(const Ctx& cx, int ringOff, int R, int G, int pos0, int f, int cg, int shift, bool valid)
		{
			if (NA_ABL & 4) return u32x4{ 0, 0, 0, 0 };
			int base = pos0 - shift; // shift <= R - FRAMES: one wrap is enough
			if (base < 0) base += R;
			unsigned p = (unsigned)(base + f);
			p = __builtin_elementwise_min(p, p - (unsigned)R); // p >= R ? p - R : p
			const int addr = RingByte(ringOff, G, (int)p, cg);
			return RingLoad(cx.srsrc, (!(NA_ABL & (512 | 2048)) && valid && f < shift) ? addr : OOB);
		}

		// In-block part of one conv tap operand: the split quad of frame `off` of the block image, zeros for the lanes whose frame lies
		// before the block start.  Their part was prefetched from the ring (LoadHistory: zeros for the in-block lanes, an out-of-range
		// buffer load returns 0) and goes through its own MFMA -- the conv is linear in the operand, so nothing has to be merged.
		__device__ __forceinline__ u32x4 TapInBlock(const u32x4* img, int off, int cg)
		{
			if (NA_ABL & 256) return u32x4{ (unsigned)off, 0, 0, 0 };
			return img[ImgIdx(cg, off < -1 ? -1 : off)];
		}

		// ... without a prefetch (taps beyond the prefetched ones, head conv): LDS, ring, or both, decided per wave
		__device__ __forceinline__ u32x4 TapOperandInline(const Ctx& cx, const u32x4* img, int ringOff, int R, int G, int pos0, int f, int cg, int shift, int lo, int hi)
		{
			const int off = f - shift;
			if (lo >= 0) return img[ImgIdx(cg, off)];
			int p = pos0 + off;
			if (p < 0) p += R;
			if (p >= R) p -= R;
			if (hi < 0) return BufLoad(cx.srsrc, RingByte(ringOff, G, p, cg));
			const u32x4 h = BufLoad(cx.srsrc, off < 0 ? RingByte(ringOff, G, p, cg) : OOB);
			const u32x4 l = img[ImgIdx(cg, off < 0 ? 0 : off)];
			u32x4 v;
			v.x = off < 0 ? h.x : l.x; v.y = off < 0 ? h.y : l.y; v.z = off < 0 ? h.z : l.z; v.w = off < 0 ? h.w : l.w;
			return v;
		}

		// write cursor of a ring after a block of n <= 128 frames (rings may be shorter than a block: compact rings, wavenet_dev.h)
		__device__ __forceinline__ int RingAdvance(int pos0, int n, int R)
		{
			int p = pos0 + n;
			if (p >= R) p -= R;
			if (p >= R) p -= R;
			if (p >= R) p -= R; // n <= 128, R >= 48
			return p;
		}

		// layer output -> LDS block image (in-block taps of the next layer) and -> the next layer's HBM ring (history for LATER blocks:
		// only the last WnRingKeep(R) frames of a block can ever be read back).  One store instruction on every path.  A kept frame's ring
		// position is counted back from the cursor AFTER the block (outPosEnd = RingAdvance(cursor, n, R)): n - f <= R frames behind it.
		__device__ __forceinline__ void Publish(const Ctx& cx, u32x4* imgNext, u32x4 v, int f, int cg, bool liveLane, int outRingOff, int outR, int outG, int outPosEnd, int nSt)
		{
			if (!(NA_ABL & 64) && liveLane) imgNext[ImgIdx(cg, f)] = v;
			if (NA_ABL & 4) return;
			const int firstKept = nSt - WnRingKeep(outR);
			int p = outPosEnd - (cx.n - f);
			p = p < 0 ? p + outR : p;
			const bool keep = liveLane && (f < nSt) && (f >= firstKept);
			RingStore(cx.srsrc, v, (!(NA_ABL & (512 | 1024)) && keep) ? RingByte(outRingOff, outG, p, cg) : OOB);
		}

		// Stages the NEXT stage's A-operand block into the other LDS weight buffer with LDS-DMA loads (buffer_load_dwordx4 ... lds: lane l's
		// 16 bytes land at ldsBase + 16 l, no VGPRs, no ds_write), issued at the start of a stage and awaited just before its closing
		// barrier.  WCOPY loads per thread cover 16 KB per workgroup of 512; larger blocks (K = 15 layers) use the tail loop.
		template <int NTHREADS, bool GEN>
		struct WeightStager
		{
			static constexpr int WCOPY = (1024 + NTHREADS - 1) / NTHREADS; // quads per thread in the fixed part (16 KB)

			__device__ __forceinline__ void Begin(const Ctx& cx, int nextBuf, const Stage& sdn) const
			{
				if (NA_ABL & 16) return;
				const int nextQ = sdn.a_ops * 64;
#pragma unroll
				for (int c = 0; c < WCOPY; c++)
				{
					const int i0 = c * NTHREADS + cx.waveAll * 64; // first quad of this wave's 1 KB slice (wave-uniform)
					const int i = i0 + cx.lane;
					__builtin_amdgcn_raw_ptr_buffer_load_lds(cx.wrsrc, (__attribute__((address_space(3))) void*)(const_cast<u32x4*>(cx.wbuf) + nextBuf * cx.wstride + i0), 16,
						(i < nextQ) ? (sdn.a_off + i) * 16 : OOB, 0, 0, 0);
				}
			}

			// LATER = number of VMEM instructions this wave issued after Begin() on every path (they may stay in flight), or 0
			template <int LATER>
			__device__ __forceinline__ void End(const Ctx& cx, int nextBuf, const Stage& sdn) const
			{
				const int nextQ = sdn.a_ops * 64;
				u32x4* dst = const_cast<u32x4*>(cx.wbuf) + nextBuf * cx.wstride;
				if constexpr (GEN) // blocks beyond the fixed 16 KB part (K = 15 layers); the fast instantiation is only used when there are none
					for (int i = (int)threadIdx.x + WCOPY * NTHREADS; i < nextQ; i += NTHREADS) dst[i] = BufLoad(cx.wrsrc, (sdn.a_off + i) * 16);
				// the DMA data must be in LDS before the closing barrier lets other waves read it (the workgroup fence only covers lgkmcnt)
				// gfx9 s_waitcnt: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "don't wait"
				__builtin_amdgcn_s_waitcnt((LATER & 15) | ((LATER >> 4) << 14) | (7 << 4) | (15 << 8));
			}
		};

		// per-wave register state carried from stage to stage (sized for the widest mode; narrow modes use the first S entries)
		template <int T>
		struct State
		{
			f32x4 xc[T]; // layer input (residual stream), f32; its split quad lives in the LDS block image (the unshifted tap reads it back)
			f32x4 hd[T]; // head accumulator
			int hit;     // saturating runs (Ctx::saturate): this lane saturated a value
		};

		// f32 quad -> split quad of a value that reaches the stream state; saturating for models without a static range proof -- same
		// places, same values as the specialised chains (wavenet_spec_impl.h Split)
		template <int T>
		__device__ __forceinline__ u32x4 SplitOf(const Ctx& cx, State<T>& st, f32x4 v)
		{
			if (cx.saturate) return SplitQuadSatLoose(v, st.hit);
			return SplitQuad(v);
		}

		// The lane index goes through an opaque asm at the start of every stage function: without it the compiler hoists the lane
		// geometry (frames, channel groups, LDS / ring address parts) of EVERY inlined stage variant to the top of the kernel and
		// keeps all of it alive across the stage loop -- dozens of VGPRs for code paths that never run.
		__device__ __forceinline__ int OpaqueLane(const Ctx& cx)
		{
			int lane = cx.lane;
			asm volatile("" : "+v"(lane));
			return lane;
		}

		template <int GP, int T>
		__device__ __forceinline__ void FrameOf(const Ctx& cx, int lane, int s, int& f, int& cg, bool& tileLive)
		{
			constexpr int P = Geo<GP, T>::P;
			const int q = lane >> 4;
			const int p = q / GP;
			cg = q % GP;
			f = cx.F0 + 16 * (P * s + p) + (lane & 15);
			tileLive = !Geo<GP, T>::PARTIAL || (P * s + p < T);
		}

		// this lane's aux operand for frame f
		// packed: channel group cg belongs to stream cg >> gsShift (gsShift = log2 of the channel groups per stream of the current array)
		template <bool PK>
		__device__ __forceinline__ u32x4 AuxOf(const Ctx& cx, int f, int cg, int gsShift)
		{
			if constexpr (PK)
			{
				if (gsShift < 0)
				{
					// dense pack: streams 2 cg and 2 cg + 1 share the group -- [cA_h, 1 | cA_l, 1 | cA_h, cB_h | cB_l, cB_h] (wavenet_plan.cpp FillSplitAux)
					const u32x2 v = cx.auxp[(2 * cg) * FRAMES + (f & (FRAMES - 1))], w = cx.auxp[(2 * cg + 1) * FRAMES + (f & (FRAMES - 1))];
					return u32x4{ v.x, v.y, (v.x & 0xffffu) | (w.x << 16), (w.y & 0xffffu) | (w.x << 16) };
				}
				const u32x2 v = cx.auxp[(cg >> gsShift) * FRAMES + (f & (FRAMES - 1))];
				return u32x4{ v.x, v.y, v.x & 0xffffu, 0u };
			}
			else return cx.auxq[f & (FRAMES - 1)];
		}

		// A run of consecutive WaveNet layer stages (WaveNetLayerT::Process, WaveNet.h:462-494) of one lane mode.
		// VMEM operations per layer, in this order on every path: WCOPY weight DMA loads, HPF*S history loads (for the NEXT layer,
		// issued once this layer's taps have consumed the previous ones, into the same registers), S ring stores.
		template <int GP, int T, int NTHREADS, int HPF, bool GEN, bool PK>
		__device__ __forceinline__ void RunLayers(const Ctx& cx, int& s, Stage& sd, int& cur, State<T>& st)
		{
			constexpr int S = Geo<GP, T>::S;
			constexpr int P = Geo<GP, T>::P;
			const WeightStager<NTHREADS, GEN> stager;
			const int lane = OpaqueLane(cx);
			int f[S], cg[S];
			bool live[S];
#pragma unroll
			for (int i = 0; i < S; i++) FrameOf<GP, T>(cx, lane, i, f[i], cg[i], live[i]);

			// ring history of the first HPF taps of the current layer: requested here for the first layer of the run, afterwards during
			// the previous layer
			u32x4 hist[HPF][S];
			{
				const int pos0 = __builtin_amdgcn_readlane(cx.myPos, sd.ring_id);
#pragma unroll
				for (int t = 0; t < HPF; t++)
#pragma unroll
					for (int i = 0; i < S; i++)
						hist[t][i] = LoadHistory(cx, sd.ring_off, sd.ring_frames, sd.G, pos0, f[i], cg[i], sd.dilation * (sd.ksize - 1 - t), t < sd.ksize - 1 && cg[i] < sd.G);
			}
			do
			{
				Stage sdn = sd;
				sdn.a_ops = 0;
				sdn.type = -1;
				if (s + 1 < cx.nstages) sdn = LoadStage(cx.stages, s + 1);
				SP_STAMP(0);
				stager.Begin(cx, (s + 1) & 1, sdn);
				SP_STAMP(6);
				const u32x4* wl = cx.wbuf + (s & 1) * cx.wstride + lane; // this lane's quad of operand m: wl[m * 64]
				const u32x4* imgCur = cx.img + cur * cx.imgStride;
				u32x4* imgNext = cx.img + (cur ^ 1) * cx.imgStride;
				// the fast instantiation (GEN == false) is only launched for models whose layers all have K == 3 and fill their lane mode
				const int K = GEN ? sd.ksize : 3, d = sd.dilation, G = GEN ? sd.G : GP;
				const bool mask = GEN && (Geo<GP, T>::PARTIAL || G < GP); // wave-uniform: some lanes have no channel group / no tile of their own
				const int inPos0 = __builtin_amdgcn_readlane(cx.myPos, sd.ring_id);
				const int outPos0 = (sd.out_ring_id >= 0) ? RingAdvance(__builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id), cx.n, sd.out_ring_frames) : 0; // cursor AFTER the block
				const int gsShift = PK ? (sd.reserved == 0 ? -1 : (sd.reserved >> 1)) : 0; // channel groups per packed stream: 1, 2, 4 -> 0, 1, 2; 0 (dense pack) -> -1

				// dilated conv (WaveNet.h:139-290): tap k reads the frame d*(K-1-k) back; accumulation starts from zero, bias and mix-in
				// arrive through the aux operand
				f32x4 acc[S];
#pragma unroll
				for (int i = 0; i < S; i++) acc[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
				for (int k = 0; k < HPF; k++)
				{
					if (!GEN || k < K - 1)
					{
						const u32x4 ah = wl[WOP(2 * k)], al = wl[WOP(2 * k + 1)];
						// Where do the frames of this tap lie for this wave?  All before the block start: the ring prefetch is the whole
						// operand.  All inside the block: the LDS image is.  Otherwise both parts go through the MFMA (the conv is linear
						// in the operand: the ring part is zero for in-block lanes and vice versa, nothing has to be merged).
						const int shift = d * (K - 1 - k);
						const int lo = cx.F0 - shift, hi = cx.F0 + 16 * P * S - 1 - shift; // wave-uniform
						if (hi < 0)
						{
#pragma unroll
							for (int i = 0; i < S; i++)
							{
								acc[i] = Mfma(ah, hist[k][i], acc[i]);
								acc[i] = MfmaLo<1>(al, hist[k][i], acc[i]);
							}
						}
						else if (lo >= 0)
						{
#pragma unroll
							for (int i = 0; i < S; i++)
							{
								u32x4 b = imgCur[ImgIdx(cg[i], f[i] - shift)];
								if (mask) b = (live[i] && cg[i] < G) ? b : u32x4{ 0, 0, 0, 0 }; // no garbage (NaN) into the MFMA
								acc[i] = Mfma(ah, b, acc[i]);
								acc[i] = MfmaLo<1>(al, b, acc[i]);
							}
						}
						else
						{
#pragma unroll
							for (int i = 0; i < S; i++)
							{
								u32x4 b = TapInBlock(imgCur, f[i] - shift, cg[i]);
								if (mask) b = (live[i] && cg[i] < G) ? b : u32x4{ 0, 0, 0, 0 };
								acc[i] = Mfma(ah, hist[k][i], acc[i]);
								acc[i] = MfmaLo<1>(al, hist[k][i], acc[i]);
								acc[i] = Mfma(ah, b, acc[i]);
								acc[i] = MfmaLo<1>(al, b, acc[i]);
							}
						}
					}
				}
				SP_STAMP(7);
				// history of the NEXT layer's first HPF taps (the registers are free again)
				{
					const bool haveNext = (s + 1 < cx.nstages) && sdn.type == WN_ST_LAYER && sdn.Gp == GP;
					const int nextPos0 = __builtin_amdgcn_readlane(cx.myPos, haveNext ? sdn.ring_id : 0);
#pragma unroll
					for (int t = 0; t < HPF; t++)
#pragma unroll
						for (int i = 0; i < S; i++)
							hist[t][i] = LoadHistory(cx, sdn.ring_off, sdn.ring_frames, sdn.G, nextPos0, f[i], cg[i], sdn.dilation * (sdn.ksize - 1 - t),
								haveNext && t < sdn.ksize - 1 && cg[i] < sdn.G);
				}
				if constexpr (GEN)
				for (int k = HPF; k < K - 1; k++)
				{
					const u32x4 ah = wl[WOP(2 * k)], al = wl[WOP(2 * k + 1)];
					const int shift = d * (K - 1 - k);
#pragma unroll
					for (int i = 0; i < S; i++)
					{
						const int lo = cx.F0 + 16 * P * i - shift; // first frame of the set (wave-uniform)
						u32x4 b = TapOperandInline(cx, imgCur, sd.ring_off, sd.ring_frames, G, inPos0, f[i], cg[i] < G ? cg[i] : 0, shift, lo, lo + 16 * P - 1);
						if (mask) b = (live[i] && cg[i] < G) ? b : u32x4{ 0, 0, 0, 0 };
						acc[i] = Mfma(ah, b, acc[i]);
						acc[i] = MfmaLo<1>(al, b, acc[i]);
					}
				}
				{
					// unshifted tap (the layer input itself, read back from the block image) and the aux operand:
					// (mix-in, conv bias) * (cond, 1)   (:288-289, :471)
					const u32x4 ah = wl[WOP(2 * K - 2)], al = wl[WOP(2 * K - 1)], xa = wl[WOP(2 * K)];
#pragma unroll
					for (int i = 0; i < S; i++)
					{
						u32x4 b = imgCur[ImgIdx(cg[i], f[i])];
						if (mask) b = (live[i] && cg[i] < G) ? b : u32x4{ 0, 0, 0, 0 };
						const u32x4 ax = AuxOf<PK>(cx, f[i], cg[i], gsShift);
						acc[i] = Mfma(ah, b, acc[i]);
						acc[i] = MfmaLo<1>(al, b, acc[i]);
						acc[i] = Mfma(xa, ax, acc[i]);
					}
				}

				SP_STAMP(1);
				// activation (:473-480), one wave-uniform branch per layer
				f32x4 z[S];
				if (sd.flags & WN_FLAG_LEAKY)
				{
#pragma unroll
					for (int i = 0; i < S; i++) z[i] = f32x4{ LeakyReLU(acc[i].x), LeakyReLU(acc[i].y), LeakyReLU(acc[i].z), LeakyReLU(acc[i].w) };
				}
				else if (sd.flags & WN_FLAG_STD_TANH)
				{
#pragma unroll
					for (int i = 0; i < S; i++) z[i] = f32x4{ StdTanh(acc[i].x), StdTanh(acc[i].y), StdTanh(acc[i].z), StdTanh(acc[i].w) };
				}
				else
				{
#pragma unroll
					for (int i = 0; i < S; i++)
					{
						if (NA_PK_TANH)
						{
							const f32x2 lo = FastTanh2(f32x2{ acc[i].x, acc[i].y }), hi = FastTanh2(f32x2{ acc[i].z, acc[i].w });
							z[i] = f32x4{ lo.x, lo.y, hi.x, hi.y };
						}
						else z[i] = f32x4{ FastTanh(acc[i].x), FastTanh(acc[i].y), FastTanh(acc[i].z), FastTanh(acc[i].w) };
					}
				}
				SP_STAMP(2);
				// head accumulate (:482) on the matrix pipe: head += I * (zh + zl); 1x1 + bias + residual (:486-491).  ONE path for every
				// layer: the last layer of an array publishes nothing (its stores are predicated off), the very last layer's 1x1 output is
				// dead but computed all the same -- a second code path costs more (registers carried around the loop, copies to merge the
				// state) than the two 1x1s it would save.
				const bool pub = (sd.flags & WN_FLAG_PUBLISH) != 0;
				const u32x4 idop = cx.idop[lane];
				{
					const u32x4 w1h = wl[WOP(2 * K + 1)], w1l = wl[WOP(2 * K + 2)], b1a = wl[WOP(2 * K + 3)];
#pragma unroll
					for (int i = 0; i < S; i++)
					{
						const u32x4 zs = SplitQuad(z[i]);
						const u32x4 ax = AuxOf<PK>(cx, f[i], cg[i], gsShift);
						st.hd[i] = Mfma(idop, zs, st.hd[i]);
						f32x4 y = st.xc[i];
						y = Mfma(w1h, zs, y);
						y = MfmaLo<2>(w1l, zs, y);
						y = Mfma(b1a, ax, y);
						st.xc[i] = y;
						// always one store per set (predicated through the offset): fixed VMEM count per layer
						Publish(cx, imgNext, SplitOf(cx, st, y), f[i], cg[i], pub && (!GEN || (live[i] && cg[i] < sd.out_G)), sd.out_ring_off, sd.out_ring_frames, sd.out_G, outPos0,
							pub ? cx.nSt : 0);
					}
				}
				if (sd.flags & WN_FLAG_PUBLISH) cur ^= 1;
				SP_STAMP(3);
				stager.template End<(HPF + 1) * S>(cx, (s + 1) & 1, sdn); // HPF*S history loads + S ring stores follow Begin() on every path
				SP_STAMP(4);
				BlockBarrier<NTHREADS / 64>();
				SP_STAMP(5);
				sd = sdn;
				s++;
			} while (s < cx.nstages && sd.type == WN_ST_LAYER && sd.Gp == GP);
		}

		// array 0 rechannel: x = w_re * cond (WaveNet.h:637 with InputSize == 1) -- the aux operand against (w_re, 0)
		template <int GP, int T, int NTHREADS, bool GEN, bool PK>
		__device__ __forceinline__ void RechannelStage(const Ctx& cx, int& s, Stage& sd, int& cur, State<T>& st)
		{
			constexpr int S = Geo<GP, T>::S;
			const WeightStager<NTHREADS, GEN> stager;
			const int lane = OpaqueLane(cx);
			Stage sdn = LoadStage(cx.stages, s + 1);
			SP_STAMP(0);
			stager.Begin(cx, (s + 1) & 1, sdn);
			const u32x4* wl = cx.wbuf + (s & 1) * cx.wstride + lane;
			u32x4* imgNext = cx.img + (cur ^ 1) * cx.imgStride;
			const int outPos0 = RingAdvance(__builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id), cx.n, sd.out_ring_frames); // cursor AFTER the block
			const u32x4 ra = wl[0];
#pragma unroll
			for (int i = 0; i < S; i++)
			{
				int f, cg; bool live;
				FrameOf<GP, T>(cx, lane, i, f, cg, live);
				const u32x4 ax = AuxOf<PK>(cx, f, cg, PK ? (sd.reserved >> 1) : 0);
				f32x4 x = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				x = Mfma(ra, ax, x);
				st.xc[i] = x;
				st.hd[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; // WaveNet.h:772 headArray.SetZero()
				Publish(cx, imgNext, SplitOf(cx, st, x), f, cg, live && cg < sd.out_G, sd.out_ring_off, sd.out_ring_frames, sd.out_G, outPos0, cx.nSt);
			}
			cur ^= 1;
			SP_STAMP(1); SP_STAMP(2); SP_STAMP(3);
			stager.template End<0>(cx, (s + 1) & 1, sdn);
			SP_STAMP(4);
			BlockBarrier<NTHREADS / 64>();
			SP_STAMP(5);
			sd = sdn;
			s++;
		}

		// array link: previous array's head rechannel (K = 1, WaveNet.h:658-660) and this array's rechannel (:637), tile by tile; the
		// operand of tile t writes the rows of tile slot t % Pn of the new mode from the k-blocks of slot t % Po of the old one
		template <int GPO, int GPN, int T, int NTHREADS, bool GEN, bool PK>
		__device__ __forceinline__ void LinkStage(const Ctx& cx, int& s, Stage& sd, int& cur, State<T>& st)
		{
			constexpr int Po = 4 / GPO, Pn = 4 / GPN, NC = Po > Pn ? Po : Pn;
			constexpr int So = Geo<GPO, T>::S, Sn = Geo<GPN, T>::S;
			const WeightStager<NTHREADS, GEN> stager;
			const int lane = OpaqueLane(cx);
			Stage sdn = LoadStage(cx.stages, s + 1);
			SP_STAMP(0);
			stager.Begin(cx, (s + 1) & 1, sdn);
			const u32x4* wl = cx.wbuf + (s & 1) * cx.wstride + lane;
			u32x4* imgNext = cx.img + (cur ^ 1) * cx.imgStride;
			const int outPos0 = RingAdvance(__builtin_amdgcn_readlane(cx.myPos, sd.out_ring_id), cx.n, sd.out_ring_frames); // cursor AFTER the block

			u32x4 hs[So], xs[So];
#pragma unroll
			for (int i = 0; i < So; i++)
			{
				int f, cg; bool live;
				FrameOf<GPO, T>(cx, lane, i, f, cg, live);
				hs[i] = SplitOf(cx, st, st.hd[i]);
				xs[i] = SplitOf(cx, st, st.xc[i]);
				if (Geo<GPO, T>::PARTIAL || GPO == 4)
				{
					const bool ok = live && cg < sd.G;
					hs[i] = ok ? hs[i] : u32x4{ 0, 0, 0, 0 };
					xs[i] = ok ? xs[i] : u32x4{ 0, 0, 0, 0 };
				}
			}
			f32x4 hn[Sn], xn[Sn];
			int fn[Sn], cgn[Sn];
			bool liven[Sn];
#pragma unroll
			for (int i = 0; i < Sn; i++)
			{
				FrameOf<GPN, T>(cx, lane, i, fn[i], cgn[i], liven[i]);
				hn[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				xn[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				if (sd.flags & WN_FLAG_BIAS)
				{
					const u32x4 ax = AuxOf<PK>(cx, fn[i], 0, 0); // bias only: any stream's aux operand carries the ones it multiplies
					hn[i] = Mfma(wl[(4 * NC) * 64], ax, hn[i]);
				}
			}
#pragma unroll
			for (int t = 0; t < T; t++)
			{
				const int u = t % NC, so = t / Po, sn = t / Pn;
				hn[sn] = Mfma(wl[(4 * u) * 64], hs[so], hn[sn]);
				hn[sn] = MfmaLo<4>(wl[(4 * u + 1) * 64], hs[so], hn[sn]);
				xn[sn] = Mfma(wl[(4 * u + 2) * 64], xs[so], xn[sn]);
				xn[sn] = MfmaLo<4>(wl[(4 * u + 3) * 64], xs[so], xn[sn]);
			}
#pragma unroll
			for (int i = 0; i < Sn; i++)
			{
				st.hd[i] = hn[i];
				st.xc[i] = xn[i];
				Publish(cx, imgNext, SplitOf(cx, st, xn[i]), fn[i], cgn[i], liven[i] && cgn[i] < sd.out_G, sd.out_ring_off, sd.out_ring_frames, sd.out_G, outPos0, cx.nSt);
			}
			cur ^= 1;
			SP_STAMP(1); SP_STAMP(2); SP_STAMP(3);
			stager.template End<0>(cx, (s + 1) & 1, sdn);
			SP_STAMP(4);
			BlockBarrier<NTHREADS / 64>();
			SP_STAMP(5);
			sd = sdn;
			s++;
		}

		// last array's head: out = scale * (conv_K(head) + b)[0]  (WaveNet.h:658-660, :793-798); K = 1 (A1) straight from registers,
		// K > 1 (A2: 16) through the LDS image / head ring like a layer conv.  One output row per tile slot.
		template <int GP, int T, int NTHREADS, bool PK>
		__device__ __forceinline__ void HeadStage(const Ctx& cx, int& s, Stage& sd, int& cur, State<T>& st)
		{
			constexpr int S = Geo<GP, T>::S;
			constexpr int P = Geo<GP, T>::P;
			const int lane = OpaqueLane(cx);
			const u32x4* wl = cx.wbuf + (s & 1) * cx.wstride + lane;
			const int K = sd.ksize, G = sd.G;
			int f[S], cg[S];
			bool live[S];
			u32x4 hs[S];
#pragma unroll
			for (int i = 0; i < S; i++)
			{
				FrameOf<GP, T>(cx, lane, i, f[i], cg[i], live[i]);
				hs[i] = SplitOf(cx, st, st.hd[i]);
				if (Geo<GP, T>::PARTIAL || GP == 4) hs[i] = (live[i] && cg[i] < G) ? hs[i] : u32x4{ 0, 0, 0, 0 };
			}
			f32x4 acc[S];
#pragma unroll
			for (int i = 0; i < S; i++) acc[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
			if (K > 1)
			{
				u32x4* imgNext = cx.img + (cur ^ 1) * cx.imgStride;
				const int pos0 = __builtin_amdgcn_readlane(cx.myPos, sd.ring_id);
#pragma unroll
				for (int i = 0; i < S; i++) Publish(cx, imgNext, hs[i], f[i], cg[i], live[i] && cg[i] < G, sd.out_ring_off, sd.out_ring_frames, sd.out_G, RingAdvance(pos0, cx.n, sd.out_ring_frames), cx.nSt);
				cur ^= 1;
				BlockBarrier<NTHREADS / 64>();
				for (int k = 0; k < K - 1; k++)
				{
					const u32x4 ah = wl[(2 * k) * 64], al = wl[(2 * k + 1) * 64];
					const int shift = sd.dilation * (K - 1 - k);
#pragma unroll
					for (int i = 0; i < S; i++)
					{
						const int lo = cx.F0 + 16 * P * i - shift;
						u32x4 b = TapOperandInline(cx, imgNext, sd.ring_off, sd.ring_frames, G, pos0, f[i], cg[i] < G ? cg[i] : 0, shift, lo, lo + 16 * P - 1);
						if (Geo<GP, T>::PARTIAL || GP == 4) b = (live[i] && cg[i] < G) ? b : u32x4{ 0, 0, 0, 0 };
						acc[i] = Mfma(ah, b, acc[i]);
						acc[i] = MfmaLo<8>(al, b, acc[i]);
					}
				}
			}
			{
				const u32x4 ah = wl[(2 * K - 2) * 64], al = wl[(2 * K - 1) * 64];
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					acc[i] = Mfma(ah, hs[i], acc[i]);
					acc[i] = MfmaLo<8>(al, hs[i], acc[i]);
					if (sd.flags & WN_FLAG_BIAS)
					{
						const u32x4 ax = AuxOf<PK>(cx, f[i], 0, 0); // bias only
						acc[i] = Mfma(wl[(2 * K) * 64], ax, acc[i]);
					}
					if constexpr (PK)
					{
						// one head row per packed stream: rows 0 .. pack-1 of the tile = the four values of the cg == 0 lane
						if (live[i] && cg[i] == 0 && f[i] < cx.nSt)
						{
							const float v[4] = { acc[i].x, acc[i].y, acc[i].z, acc[i].w };
#pragma unroll
							for (int q = 0; q < 4; q++)
								if (q < cx.pack && cx.outRow[q] >= 0) cx.out[cx.outRow[q] + f[i]] = cx.headScale * v[q];
						}
					}
					else if (live[i] && cg[i] == 0 && f[i] < cx.nSt) cx.out[cx.outBase + f[i]] = cx.headScale * acc[i].x;
				}
			}
			s++;
		}


		// grid = active streams / SPB; workgroup = SPB streams x WPS waves of T tiles (WPS * T * 16 >= n).
		// dynamic LDS: auxq[SPB][FRAMES] quads | img[SPB][2][maxG][PLANE] quads | wbuf[2][wstride] quads | idop[64] quads
		// PK: packed launch -- aux LDS is [SPB][4][FRAMES] x 8 bytes instead of [SPB][FRAMES] quads
		template <int T, int SPB, int WPS, bool GEN, bool PK>
		__global__ void __launch_bounds__(64 * WPS * SPB) __attribute__((amdgpu_waves_per_eu(T == 2 ? 4 : 2))) WaveNetSplitKernel(const LaunchArgs args, int maxGAll, int wstride, const float* __restrict__ in, float* __restrict__ out,
			long inStride, long outStride, int n, long long* __restrict__ trace, int traceBlock)
		{
			constexpr int NTHREADS = 64 * WPS * SPB;
			int gi = 0;
			for (int i = 1; i < args.numGroups; i++)
				if ((int)blockIdx.x >= args.g[i].firstBlock) gi = i;
			const GroupArgs& ga = args.g[gi];
			const int groupBlock = (int)blockIdx.x - ga.firstBlock;
			extern __shared__ __attribute__((aligned(16))) char smem[];

			const int lane = threadIdx.x & 63;
			const int waveAll = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			const int sub = waveAll / WPS;  // stream within the workgroup
			const int wave = waveAll % WPS; // part of the stream's block
			u32x4* auxAll = reinterpret_cast<u32x4*>(smem);
			u32x4* imgAll = auxAll + (PK ? SPB * 4 * FRAMES / 2 : SPB * FRAMES);
			const int imgStride = maxGAll * PLANE;
			u32x4* wbuf = imgAll + SPB * 2 * imgStride;

			// a partial last workgroup: the surplus waves shadow the last stream (they must keep staging weights and meeting barriers)
			int sidx = groupBlock * SPB + sub;
			const bool liveStream = sidx < ga.numStreams;
			if (!liveStream) sidx = ga.numStreams - 1;
			const int slot = ga.slots ? ga.slots[sidx] : ga.slot0 + sidx;
			const int row = PK ? 0 : (ga.slots ? ga.rows[sidx] : ga.row0 + sidx);
			u32x4* stt = ga.state + (size_t)slot * (size_t)ga.stateF4;
			int* header = reinterpret_cast<int*>(stt);

			Ctx cx;
			cx.stages = ga.stages;
			cx.nstages = ga.nstages;
			cx.wbuf = wbuf;
			cx.wstride = wstride;
			cx.wrsrc = MakeRsrc(ga.wsplit, (unsigned)ga.wsplitQuads * 16u);
			cx.img = imgAll + sub * 2 * imgStride;
			cx.imgStride = imgStride;
			cx.auxq = auxAll + sub * FRAMES;
			cx.auxp = reinterpret_cast<const u32x2*>(auxAll) + sub * 4 * FRAMES;
			cx.pack = PK ? ga.pack : 1;
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				// packed: rows[] always holds `pack` entries per virtual stream (the host fills -1 for the unused tail of the last one)
				const int r = (PK && liveStream && q < ga.pack) ? ga.rows[sidx * ga.pack + q] : -1;
				cx.outRow[q] = r >= 0 ? (long)r * outStride : -1;
			}
			cx.idop = wbuf + 2 * wstride;
			cx.srsrc = MakeRsrc(stt, (unsigned)ga.stateF4 * 16u);
			cx.myPos = header[lane]; // lane r holds the write cursor of ring r
			cx.n = n;
			cx.nSt = liveStream ? n : 0;
			cx.F0 = wave * 16 * T;
			cx.lane = lane;
			cx.j = lane & 15;
			cx.q = lane >> 4;
			cx.waveAll = waveAll;
			cx.out = out;
			cx.outBase = (size_t)row * outStride;
			cx.headScale = ga.headScale;
			cx.saturate = ga.saturate;
			cx.trace = ((int)blockIdx.x == traceBlock) ? trace : nullptr;
			cx.nwaves = NTHREADS / 64;
#ifdef NA_SP_TRACE
			if (cx.trace != nullptr && lane == 0) cx.trace[((ga.nstages * 8 + 0) * cx.nwaves) + waveAll] = (long long)__builtin_readcyclecounter();
#endif

			// input row (WaveNet.h:770 input -> condition) -> the aux operand of every frame: split quad of (cond, 1, 0, 0); cond = 0 beyond n
			if constexpr (PK)
			{
				u32x2* auxp = reinterpret_cast<u32x2*>(auxAll) + sub * 4 * FRAMES;
				for (int q = 0; q < ga.pack; q++)
				{
					const int r = liveStream ? ga.rows[sidx * ga.pack + q] : -1;
					for (int i = wave * 64 + lane; i < FRAMES; i += WPS * 64)
					{
						const float c = (r >= 0 && i < n) ? ClampCond(in[(size_t)r * inStride + i], ga.condLimit) : 0.0f;
						const _Float16 ch = (_Float16)c, cl = (_Float16)(c - (float)ch);
						const f16x2 a = { ch, (_Float16)1.0f }, b = { cl, (_Float16)1.0f };
						auxp[q * FRAMES + i] = u32x2{ __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b) };
					}
				}
			}
			else
			{
				u32x4* auxq = auxAll + sub * FRAMES;
				for (int i = wave * 64 + lane; i < FRAMES; i += WPS * 64)
				{
					// [cond_h, 1 | cond_l, 1 | cond_h, 0 | 0, 0] (see FillSplitAux in wavenet_plan.cpp)
					const float c = (i < n) ? ClampCond(in[(size_t)row * inStride + i], ga.condLimit) : 0.0f;
					const _Float16 ch = (_Float16)c, cl = (_Float16)(c - (float)ch);
					const f16x2 a = { ch, (_Float16)1.0f }, b = { cl, (_Float16)1.0f }, d = { ch, (_Float16)0.0f };
					auxq[i] = u32x4{ __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, d), 0u };
				}
			}
			// zero quad in front of frame 0 of every plane of both block images
			for (int i = threadIdx.x; i < SPB * 2 * maxGAll; i += NTHREADS) imgAll[i * PLANE + GUARD - 1] = u32x4{ 0, 0, 0, 0 };
			// identity A operand: row i x k-block q = i / 4: 1.0 against the h AND the l half of channel i % 4
			if (threadIdx.x < 64)
			{
				const int i = lane & 15, q = lane >> 4;
				const unsigned one = 0x3c00u; // f16 1.0
				const unsigned lo = (q == (i >> 2)) ? (((i & 3) == 0) ? one : ((i & 3) == 1) ? (one << 16) : 0u) : 0u;
				const unsigned hi = (q == (i >> 2)) ? (((i & 3) == 2) ? one : ((i & 3) == 3) ? (one << 16) : 0u) : 0u;
				(wbuf + 2 * wstride)[lane] = u32x4{ lo, hi, lo, hi };
			}
			Stage sd = LoadStage(cx.stages, 0);
			for (int i = threadIdx.x; i < sd.a_ops * 64; i += NTHREADS) wbuf[i] = BufLoad(cx.wrsrc, (sd.a_off + i) * 16);
			BlockBarrier<NTHREADS / 64>();

			State<T> st;
			st.hit = 0;
			int cur = 0;
			int s = 0;
			int mode = sd.Gp;
			while (s < cx.nstages)
			{
				if (sd.type == WN_ST_LAYER)
				{
					// K = 3 models: both shifted taps' history is requested a layer ahead; larger kernels (A2: 6 / 15): the first 2 as well,
					// the rest in line
					if (mode == 4) RunLayers<4, T, NTHREADS, 2, GEN, PK>(cx, s, sd, cur, st);
					else if (mode == 2) RunLayers<2, T, NTHREADS, 2, GEN, PK>(cx, s, sd, cur, st);
					else RunLayers<1, T, NTHREADS, 2, GEN, PK>(cx, s, sd, cur, st);
				}
				else if (sd.type == WN_ST_RECHANNEL_COND)
				{
					if (mode == 4) RechannelStage<4, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st);
					else if (mode == 2) RechannelStage<2, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st);
					else RechannelStage<1, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st);
				}
				else if (sd.type == WN_ST_ARRAY_LINK)
				{
					const int next = sd.ksize;
					const int key = mode * 8 + next;
					switch (key)
					{
					case 4 * 8 + 4: LinkStage<4, 4, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					case 4 * 8 + 2: LinkStage<4, 2, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					case 4 * 8 + 1: LinkStage<4, 1, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					case 2 * 8 + 4: LinkStage<2, 4, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					case 2 * 8 + 2: LinkStage<2, 2, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					case 2 * 8 + 1: LinkStage<2, 1, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					case 1 * 8 + 4: LinkStage<1, 4, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					case 1 * 8 + 2: LinkStage<1, 2, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					default: LinkStage<1, 1, T, NTHREADS, GEN, PK>(cx, s, sd, cur, st); break;
					}
					mode = next;
				}
				else
				{
					if (mode == 4) HeadStage<4, T, NTHREADS, PK>(cx, s, sd, cur, st);
					else if (mode == 2) HeadStage<2, T, NTHREADS, PK>(cx, s, sd, cur, st);
					else HeadStage<1, T, NTHREADS, PK>(cx, s, sd, cur, st);
				}
			}

#ifdef NA_SP_TRACE
			if (cx.trace != nullptr && lane == 0) cx.trace[((ga.nstages * 8 + 1) * cx.nwaves) + waveAll] = (long long)__builtin_readcyclecounter();
#endif
			if (cx.saturate) CountRangeEvent(header, st.hit, lane, liveStream);
			// advance every ring cursor by n (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
			if (wave == 0 && liveStream && lane < ga.nrings)
			{
				header[lane] = RingAdvance(cx.myPos, n, ga.ringFrames[lane]);
			}
		}

		template <int T, int SPB, int WPS, bool GEN, bool PK>
		static hipError_t Launch(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream)
		{
			LaunchArgs args = {};
			args.numGroups = numGroups;
			int blocks = 0, maxG = 1, maxOps = 16;
			for (int i = 0; i < numGroups; i++)
			{
				const WnFrameGroup& g = groups[i];
				const WnModelDev& m = *g.model;
				GroupArgs& a = args.g[i];
				a.stages = m.sstages; a.wsplit = m.wsplit; a.ringFrames = m.ring_frames;
				a.state = reinterpret_cast<u32x4*>(g.state); a.slots = g.slots; a.rows = g.rows;
				a.nstages = m.nstages; a.nrings = m.nrings; a.stateF4 = m.state_f4; a.wsplitQuads = m.wsplit_quads;
				a.headScale = m.head_scale;
				a.condLimit = m.cond_limit;
				a.saturate = m.saturate;
				a.numStreams = g.numStreams; a.slot0 = g.slot0; a.row0 = g.row0;
				a.maxG = m.max_G;
				a.firstBlock = blocks;
				a.pack = g.pack > 1 ? g.pack : 1;
				if (a.pack > 1 && !PK) return hipErrorInvalidValue; // the plain flavour cannot run packed groups (a plain group in a packed launch is pack = 1)
				if (PK && g.slots == nullptr) return hipErrorInvalidValue; // packed launches always pass the index lists
				blocks += (g.numStreams + SPB - 1) / SPB;
				maxG = std::max(maxG, m.max_G);
				maxOps = std::max(maxOps, m.max_split_ops);
			}
			const int wstride = maxOps * 64; // quads per LDS weight buffer (the LDS-DMA staging always writes its fixed 16 KB part)
			const size_t lds = (PK ? (size_t)SPB * 4 * FRAMES * 8 : (size_t)SPB * FRAMES * 16) + (size_t)SPB * 2 * maxG * PLANE * 16 + (size_t)2 * wstride * 16 + 1024;
			if (lds > 160 * 1024) return hipErrorInvalidValue;
			auto kernel = WaveNetSplitKernel<T, SPB, WPS, GEN, PK>;
			if (lds > 64 * 1024)
			{
				// per instantiation and device: the whole LDS of a CU once (granting it does not change what a launch uses)
				static PerDeviceOnce attr;
				const hipError_t e = attr.Run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
				if (e != hipSuccess) return e;
			}
			hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * WPS * SPB), lds, stream, args, maxG, wstride, in, out, inStride, outStride, n, GetWaveNetTraceBuffer(),
				Tuning::Get().traceBlock);
			return hipGetLastError();
		}
	}

	hipError_t LaunchWaveNetSplitFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, int sharing)
	{
		if (n <= 0 || numGroups <= 0) return hipSuccess;
		if (n > WN_MAX_FRAMES || numGroups > WN_FRAME_MAX_GROUPS) return hipErrorInvalidValue;
		{
			// the official architectures run compile-time specialised chains (wavenet_spec_kernels.hip); this file's stage interpreter takes
			// everything else: runtime-shaped models, other block lengths, mixed launches
			const hipError_t e = LaunchWaveNetSpecFused(groups, numGroups, in, out, inStride, outStride, n, stream, sharing);
			if (e != hipErrorNotSupported) return e;
		}
		int total = 0;
		for (int i = 0; i < numGroups; i++)
		{
			if (groups[i].numStreams <= 0) return hipErrorInvalidValue;
			total += groups[i].numStreams;
		}
		const int tEnv = Tuning::Get().spT;       // tuning knob: tiles per wave (2, 4)
		const int spbEnv = Tuning::Get().spSpb;   // tuning knob: streams per workgroup (1, 2)
		const bool genEnv = Tuning::Get().spGen;  // tuning knob: always the generic instantiation
		const int spb = spbEnv > 0 ? spbEnv : (total >= 512 ? 2 : 1);
		const int tiles = (n + 15) / 16;
		// fast instantiation: every layer of every group has K == 3 and fills its lane mode (the official A1 architectures except Lite);
		// split_fast_T = fewest tiles per wave it can run with (4 when an array has <= 4 channels), 0 = needs the generic one
		bool gen = genEnv;
		int t = (tEnv == 2 || tEnv == 4) ? tEnv : 2;
		for (int i = 0; i < numGroups; i++)
		{
			if (groups[i].model->split_fast_T == 0) gen = true;
			else t = std::max(t, groups[i].model->split_fast_T);
		}
		// packed groups (several real streams per virtual stream, WaveNetPlan::pack): their own launch, fast instantiation, 2 tiles per wave
		bool packed = false;
		for (int i = 0; i < numGroups; i++) packed = packed || groups[i].pack > 1;
		if (packed)
		{
			for (int i = 0; i < numGroups; i++)
				if (groups[i].model->split_fast_T != 2) return hipErrorInvalidValue; // plain groups may ride along (pack = 1) if their plan is a fast one
#ifndef NA_SP_QUICK
#define NA_SP_LAUNCH_PK(SS, WW) return sp::Launch<2, SS, WW, false, true>(groups, numGroups, in, out, inStride, outStride, n, stream)
			if (tiles > 4) { if (spb >= 2) NA_SP_LAUNCH_PK(2, 4); NA_SP_LAUNCH_PK(1, 4); }
			if (tiles > 2) { if (spb >= 2) NA_SP_LAUNCH_PK(2, 2); NA_SP_LAUNCH_PK(1, 2); }
			if (spb >= 2) NA_SP_LAUNCH_PK(2, 1);
			NA_SP_LAUNCH_PK(1, 1);
#undef NA_SP_LAUNCH_PK
#else
			return hipErrorInvalidValue;
#endif
		}
#ifdef NA_SP_QUICK
#define NA_SP_LAUNCH(TT, SS, WW) do { return sp::Launch<2, 2, 4, false, false>(groups, numGroups, in, out, inStride, outStride, n, stream); } while (0)
#else
#define NA_SP_LAUNCH(TT, SS, WW) do { return gen ? sp::Launch<TT, SS, WW, true, false>(groups, numGroups, in, out, inStride, outStride, n, stream) \
	: sp::Launch<TT, SS, WW, false, false>(groups, numGroups, in, out, inStride, outStride, n, stream); } while (0)
#endif
#ifdef NA_SP_QUICK // experiment builds: only the headline instantiation
		NA_SP_LAUNCH(2, 2, 4);
#endif
		if (t == 4)
		{
			if (tiles > 4) { if (spb >= 2) NA_SP_LAUNCH(4, 2, 2); NA_SP_LAUNCH(4, 1, 2); }
			if (spb >= 2) NA_SP_LAUNCH(4, 2, 1);
			NA_SP_LAUNCH(4, 1, 1);
		}
		if (tiles > 4) { if (spb >= 4) NA_SP_LAUNCH(2, 4, 4); if (spb >= 2) NA_SP_LAUNCH(2, 2, 4); NA_SP_LAUNCH(2, 1, 4); }
		if (tiles > 2) { if (spb >= 2) NA_SP_LAUNCH(2, 2, 2); NA_SP_LAUNCH(2, 1, 2); }
		if (spb >= 2) NA_SP_LAUNCH(2, 2, 1);
		NA_SP_LAUNCH(2, 1, 1);
#undef NA_SP_LAUNCH
	}
}
