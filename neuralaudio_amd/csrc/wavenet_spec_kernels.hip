// wavenet_spec_kernels.hip -- the f16-split MFMA WaveNet block kernel as COMPILE-TIME layer chains for the official architectures.
//
// Reference arithmetic being replaced: WaveNetModelT / LayerArrayT / LayerT::Process, Conv1DT::Process, DenseLayerT::Process
// (NeuralAudio/WaveNet.h:768-799, 632-661, 462-494, 139-290, 336-383) with FastMath::Tanh (Activation.h:83-91).  The reference itself
// instantiates its templates once per official architecture (InternalModel.h:12-20, 152-159; WaveNet.h:503-661); so does this file.
//
// Same mapping, same operand images, same stream-state format and the same order of floating-point operations as the stage
// interpreter in wavenet_split_kernels.hip (so the two are interchangeable on a running stream and agree bit for bit), but nothing
// about the model is looked up at run time: the layer chain is unrolled, and dilation, ring offset and length, operand offsets in the
// weight image, image buffer parity, and the classification of every conv tap of every wave (all of its frames before the block:
// prefetched ring history only; all inside the block: LDS image only; straddling: both) are template constants.  What remains dynamic
// is per stream (ring cursors, state / row pointers).  Per layer and wave this removes the stage-descriptor load, ~80 scalar and ~40
// vector address instructions, the tap-classification branches and their basic blocks, the LDS round trip of the unshifted tap (the
// layer input's split quad stays in registers), the second aux read, the LDS publish when the next layer has no in-block tap
// (d >= block length), and the history loads / ring stores no wave of the block needs.
//
// Blocks must be exactly NF = 128, 64 or 32 frames (what audio hosts use); other sizes run on the interpreter -- same state, so a
// stream may alternate between the two.
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_split_dev.h"

namespace na
{
	namespace spk
	{
		using namespace sp;

		typedef __attribute__((address_space(3))) char* LdsPtr;
		__device__ __forceinline__ u32x4 LdsRead16(unsigned addr) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>((LdsPtr)(size_t)addr); }
		__device__ __forceinline__ u32x2 LdsRead8(unsigned addr) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x2*>((LdsPtr)(size_t)addr); }
		__device__ __forceinline__ void LdsWrite16(unsigned addr, u32x4 v) { *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((LdsPtr)(size_t)addr) = v; }

		// ---- the architectures (virtual models: after padding / stream packing, wavenet_plan.cpp) ---------------------------------
		// NeuralModel.cpp:71-76 dilation tables; channels are those of the lane modes the plans fill completely
		struct ArchStd // A1 Standard (16 -> 8)
		{
			static constexpr int NA = 2;
			static constexpr int CH[2] = { 16, 8 };
			static constexpr int NLA[2] = { 10, 10 };
			static constexpr int DIL[2][16] = { { 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 }, { 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 } };
		};
		struct ArchLite // the "lite" dilation lists at 16 / 8 channels: A1 Lite padded (12 / 6), two Feather streams packed (8 / 4 each)
		{
			static constexpr int NA = 2;
			static constexpr int CH[2] = { 16, 8 };
			static constexpr int NLA[2] = { 7, 13 };
			static constexpr int DIL[2][16] = { { 1, 2, 4, 8, 16, 32, 64 }, { 128, 256, 512, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 } };
		};
		struct ArchLite16 // ... at 16 / 16 channels: four Nano streams packed (4 / 2 each, the second array padded to 4 per stream)
		{
			static constexpr int NA = 2;
			static constexpr int CH[2] = { 16, 16 };
			static constexpr int NLA[2] = { 7, 13 };
			static constexpr int DIL[2][16] = { { 1, 2, 4, 8, 16, 32, 64 }, { 128, 256, 512, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 } };
		};

		// architectures that may share one launch (same stage count, same LDS map): GroupArgs::arch picks the member per workgroup
		struct FamStd { typedef ArchStd A0; typedef ArchStd A1; static constexpr int N = 1; };
		struct FamLite { typedef ArchLite A0; typedef ArchLite A1; static constexpr int N = 1; };
		struct FamLitePacked { typedef ArchLite A0; typedef ArchLite16 A1; static constexpr int N = 2; };

		enum TapClass { TAP_LDS = 0, TAP_HIST = 1, TAP_BOTH = 2 };

		template <class A>
		struct Tab
		{
			static constexpr int NA = A::NA;
			static constexpr int TotalLayers() { int n = 0; for (int a = 0; a < NA; a++) n += A::NLA[a]; return n; }
			static constexpr int NL = TotalLayers();
			static constexpr int NSTAGES = NL + NA + 1; // rechannel, layers, NA - 1 links, head
			static constexpr int ArrOf(int L) { int a = 0; while (L >= A::NLA[a]) { L -= A::NLA[a]; a++; } return a; }
			static constexpr int InArr(int L) { int a = 0; while (L >= A::NLA[a]) { L -= A::NLA[a]; a++; } return L; }
			static constexpr int Dil(int L) { return A::DIL[ArrOf(L)][InArr(L)]; }
			static constexpr int GPof(int a) { return A::CH[a] / 4; } // lane mode == channel groups (full modes only)
			static constexpr bool FirstOfArr(int L) { return InArr(L) == 0; }
			static constexpr bool LastOfArr(int L) { return InArr(L) == A::NLA[ArrOf(L)] - 1; }
			static constexpr int RingFrames(int L) { return (2 * Dil(L) + 15) / 16 * 16 + FRAMES; } // K = 3; wavenet_plan.cpp AddRing
			static constexpr int RingOff(int L) // quads
			{
				int o = WN_HEADER_F4;
				for (int l = 0; l < L; l++) o += RingFrames(l) * GPof(ArrOf(l));
				return o;
			}
			static constexpr int StateF4 = (RingOff(NL) + 15) / 16 * 16;
			static constexpr int StageOfLayer(int L) { return 1 + L + ArrOf(L); }
			static constexpr int FirstLayerOfArr(int a) { int L = 0; for (int i = 0; i < a; i++) L += A::NLA[i]; return L; }
			static constexpr int LinkStage(int a) { return FirstLayerOfArr(a) + a; } // the link in front of array a >= 1
			static constexpr int LinkNC(int a) { const int Po = 4 / GPof(a - 1), Pn = 4 / GPof(a); return Po > Pn ? Po : Pn; }
			static constexpr int StageOps(int s)
			{
				if (s == 0) return 1;
				if (s == NSTAGES - 1) return 3;
				for (int a = 1; a < NA; a++)
					if (s == LinkStage(a)) return 4 * LinkNC(a) + 1;
				return 10; // 2 K + 4, K = 3
			}
			static constexpr int MaxOps() { int m = 0; for (int s = 0; s < NSTAGES; s++) m = StageOps(s) > m ? StageOps(s) : m; return m; }
			static constexpr int AOff(int s) { int o = 0; for (int i = 0; i < s; i++) o += StageOps(i) * 64; return o; } // quads
			static constexpr int WsplitQuads = AOff(NSTAGES);
		};

		// launch shape: NF frames per block, T = 2 tiles per wave, SPB streams per workgroup sharing the staged weights
		template <class A_, int NF_, int SPB_, bool PK_>
		struct Cfg
		{
			typedef A_ A;
			typedef Tab<A_> TB;
			static constexpr int NF = NF_, SPB = SPB_, T = 2, WPS = NF_ / 32, NTHREADS = 64 * WPS * SPB;
			static constexpr bool PK = PK_;
			static constexpr int MAXOPS = 10; // the same LDS map for every architecture (members of a family share a launch)
			static_assert(TB::MaxOps() <= MAXOPS, "stage operand block");
			// LDS map (bytes)
			static constexpr int AUX_OFF = 0;                                   // [SPB][FRAMES] quads, PK: [SPB][4][FRAMES] x 8 bytes
			static constexpr int IMG_OFF = AUX_OFF + SPB * FRAMES * (PK_ ? 32 : 16); // [SPB][2][4][PLANE] quads
			static constexpr int IMG_ONE = 4 * PLANE * 16;                      // one image: 4 planes
			static constexpr int WBUF_OFF = IMG_OFF + SPB * 2 * IMG_ONE;        // [2][MAXOPS] operands of 1 KB
			static constexpr int WBUF_ONE = MAXOPS * 1024;
			static constexpr int IDOP_OFF = WBUF_OFF + 2 * WBUF_ONE;            // identity operand
			static constexpr int DUMP_OFF = IDOP_OFF + 1024;                    // where the LDS-DMA of a wave with nothing to stage lands
			static constexpr int LDS_BYTES = DUMP_OFF + 1024;
			static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
		};

		// geometry of lane mode GP with T = 2 tiles per wave: P tiles share one MFMA ("set"), S sets per wave
		template <int GP>
		struct Geo
		{
			static_assert(GP == 4 || GP == 2, "full lane modes of 16 / 8 channels");
			static constexpr int P = 4 / GP;
			static constexpr int S = 2 / P;
		};

		// Where do the frames [F0 + 16 P i, + 16 P) of set i of the wave starting at F0 lie relative to the block start, `shift` frames back?
		constexpr int TapClassOf(int F0, int P, int i, int shift)
		{
			const int lo = F0 + 16 * P * i - shift, hi = lo + 16 * P - 1;
			return hi < 0 ? TAP_HIST : (lo >= 0 ? TAP_LDS : TAP_BOTH);
		}

		// everything about layer L that depends on the wave: classes of its two shifted taps and of the next layer's (history prefetch)
		template <class C, int L>
		struct LayerSig
		{
			typedef typename C::TB TB;
			static constexpr int GP = TB::GPof(TB::ArrOf(L)), P = Geo<GP>::P, S = Geo<GP>::S, d = TB::Dil(L);
			static constexpr bool NEXT = !TB::LastOfArr(L); // a layer of the same array follows (its history is requested during this one)
			static constexpr int dn = NEXT ? TB::Dil(NEXT ? L + 1 : L) : 0;
			static constexpr unsigned Of(int w)
			{
				unsigned s = 0;
				for (int k = 0; k < 2; k++)
					for (int i = 0; i < S; i++)
					{
						s = s * 4 + (unsigned)TapClassOf(32 * w, P, i, d * (2 - k));
						s = s * 4 + (unsigned)(NEXT ? TapClassOf(32 * w, P, i, dn * (2 - k)) : 0);
					}
				return s;
			}
			static constexpr int Rep(int w) { int r = w; for (int v = w - 1; v >= 0; v--) if (Of(v) == Of(w)) r = v; return r; }
			static constexpr unsigned MaskOf(int r) { unsigned m = 0; for (int w = 0; w < C::WPS; w++) if (Rep(w) == r) m |= 1u << w; return m; }
			static constexpr bool LastRep(int r) { for (int w = r + 1; w < C::WPS; w++) if (Rep(w) == w) return false; return true; }
		};

		// per-wave values that do not change from stage to stage
		struct Ctx
		{
			__amdgpu_buffer_rsrc_t srsrc; // this stream's state (zero-sized for the shadow waves of a partial last workgroup: loads give 0, stores are dropped)
			__amdgpu_buffer_rsrc_t wrsrc; // split weight image
			int myPos;                    // lane r: write cursor of ring r
			int wave, sub, waveAll;       // wave within the stream's block, stream within the workgroup (wave-uniform)
			int lane;
			int gs0, gs1;                 // packed launches: log2(channel groups per real stream) of array 0 / the other arrays
#ifdef NA_SP_TRACE
			long long* trace;             // tuning aid (make SUFFIX=_trace EXTRA=-DNA_SP_TRACE, tools/trace_split_timeline.py): nullptr unless this is the traced workgroup
			int nwaves;
#endif
		};

		// shader-clock stamps of one workgroup, trace[(stage * 8 + point) * waves + wave]; the scheduling barriers pin the stamp between the phases
#ifdef NA_SP_TRACE
#define SPK_STAMP(stage, point) do { __builtin_amdgcn_sched_barrier(0); if (cx.trace != nullptr && cx.lane == 0) cx.trace[(((stage) * 8 + (point)) * cx.nwaves) + cx.waveAll] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SPK_STAMP(stage, point) (void)0
#endif

		// per-lane values of one lane mode (recomputed at an array link)
		template <class C, int GP>
		struct Lanes
		{
			int fl;          // frame within the wave's 32, set 0: 16 p + j
			int cg;          // channel group
			unsigned img;    // LDS byte address of (plane cg, frame F0 + fl) in image 0 of this stream
			unsigned aux;    // LDS byte address of the aux entry of frame F0 + fl (PK: of the stream owning channel group cg)
			unsigned ring;   // byte offset of (frame fl, group cg) within a frame-major ring of this mode: (fl * GP + cg) * 16

			__device__ __forceinline__ void Init(const Ctx& cx, int gs)
			{
				const int q = cx.lane >> 4, j = cx.lane & 15;
				const int p = q / GP;
				cg = q % GP;
				fl = 16 * p + j;
				const int f = 32 * cx.wave + fl;
				img = (unsigned)(C::IMG_OFF + cx.sub * 2 * C::IMG_ONE + (cg * PLANE + GUARD + f) * 16);
				ring = (unsigned)((fl * GP + cg) * 16);
				if constexpr (C::PK) aux = (unsigned)(C::AUX_OFF + ((cx.sub * 4 + (cg >> gs)) * FRAMES + f) * 8);
				else aux = (unsigned)(C::AUX_OFF + (cx.sub * FRAMES + f) * 16);
			}
		};

		template <class C, int GP>
		__device__ __forceinline__ u32x4 AuxRead(const Lanes<C, GP>& ln, int i)
		{
			constexpr int P = Geo<GP>::P;
			if constexpr (C::PK)
			{
				const u32x2 v = LdsRead8(ln.aux + (unsigned)(16 * P * i * 8));
				return u32x4{ v.x, v.y, v.x & 0xffffu, 0u };
			}
			else return LdsRead16(ln.aux + (unsigned)(16 * P * i * 16));
		}

		struct State
		{
			f32x4 xc[2];       // layer input (residual stream), f32
			f32x4 hd[2];       // head accumulator
			u32x4 xs[2];       // split quad of xc: the unshifted conv tap's operand
			u32x4 hist[2][2];  // ring history of the current layer's shifted taps [tap][set]
		};

		// Byte offset of ring position (base + fl) mod R, channel group cg, relative to the ring's start: `base` in [0, R) is wave-uniform, the
		// lane part fl < 32 is folded into Lanes::ring, so the wrap is one unsigned min on the byte offset (3 VALU per access); the
		// ring's start rides in the instruction's scalar offset.
		template <int GP, int R>
		__device__ __forceinline__ int RingWrap(unsigned laneRing, int base)
		{
			unsigned a = laneRing + (unsigned)base * (unsigned)(GP * 16);
			return (int)__builtin_elementwise_min(a, a - (unsigned)(R * GP * 16));
		}

		// Ring history of layer L's shifted tap k for set i (frames before the block start).  One load instruction whatever the class --
		// every wave issues the same number of VMEM operations per stage, so the vmcnt waits can be counted -- with an out-of-range
		// offset where this wave (TAP_LDS) or this lane (TAP_BOTH, frames inside the block) needs nothing: such a load returns zeros.
		// (The scalar offset of a buffer instruction is not part of its range check: an out-of-range vector offset drops the access, and
		// so does the zero-sized resource of a shadow wave, whatever the scalar offset.)
		template <class C, int L, int WR>
		__device__ __forceinline__ u32x4 HistLoad(const Ctx& cx, unsigned laneRing, int fl, int k, int i)
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::GPof(TB::ArrOf(L)), P = Geo<GP>::P, R = TB::RingFrames(L), OFF = TB::RingOff(L);
			const int shift = TB::Dil(L) * (2 - k);
			const int cls = TapClassOf(32 * WR, P, i, shift);
			if ((NA_ABL & 4) || cls == TAP_LDS) return RingLoad(cx.srsrc, OOB);
			const int pos0 = __builtin_amdgcn_readlane(cx.myPos, L);
			int base = pos0 - shift + 32 * cx.wave + 16 * P * i; // wave-uniform; in (-R, 2R)
			if (base < 0) base += R;
			if (base >= R) base -= R;
			const int addr = RingWrap<GP, R>(laneRing, base);
			if (cls == TAP_HIST) return RingLoad(cx.srsrc, addr, OFF * 16);
			return RingLoad(cx.srsrc, (32 * cx.wave + 16 * P * i + fl < shift) ? addr : OOB, OFF * 16);
		}

		// does any wave of the block need the ring history of (layer L, tap k, set i)?  (wave 0 has the earliest frames)
		template <class C, int L>
		constexpr bool HistNeeded(int k, int i)
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::GPof(TB::ArrOf(L)), P = Geo<GP>::P;
			return TapClassOf(0, P, i, TB::Dil(L) * (2 - k)) != TAP_LDS;
		}
		template <class C, int L>
		constexpr int HistLoadsOf()
		{
			typedef typename C::TB TB;
			constexpr int S = Geo<TB::GPof(TB::ArrOf(L))>::S;
			int n = 0;
			for (int k = 0; k < 2; k++)
				for (int i = 0; i < S; i++) n += HistNeeded<C, L>(k, i) ? 1 : 0;
			return n;
		}

		template <class C, int L, int WR>
		__device__ __forceinline__ void HistPrefetch(const Ctx& cx, unsigned laneRing, int fl, State& st)
		{
			constexpr int S = Geo<C::TB::GPof(C::TB::ArrOf(L))>::S;
#pragma unroll
			for (int k = 0; k < 2; k++)
#pragma unroll
				for (int i = 0; i < S; i++)
					if (HistNeeded<C, L>(k, i)) st.hist[k][i] = HistLoad<C, L, WR>(cx, laneRing, fl, k, i);
		}

		// The input of layer LN (produced by the stage in front of it) -> the LDS image (in-block taps of LN, if it has any) and LN's HBM
		// ring (history for LATER blocks: only the last R - 128 frames of a block are ever read back).  Ring stores that no wave of the
		// block needs are not issued at all; the others are one instruction on every wave (out-of-range offset where nothing is kept).
		template <class C, int LN>
		constexpr bool StoreNeeded(int i)
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::GPof(TB::ArrOf(LN)), P = Geo<GP>::P, KEEP = TB::RingFrames(LN) - FRAMES;
			return (C::NF - 32) + 16 * P * (i + 1) - 1 >= C::NF - KEEP; // the last wave's last frame of set i
		}
		template <class C, int LN>
		constexpr int StoresOf()
		{
			constexpr int S = Geo<C::TB::GPof(C::TB::ArrOf(LN))>::S;
			int n = 0;
			for (int i = 0; i < S; i++) n += StoreNeeded<C, LN>(i) ? 1 : 0;
			return n;
		}

		template <class C, int LN, int GP>
		__device__ __forceinline__ void Publish(const Ctx& cx, const Lanes<C, GP>& ln, u32x4 v, int i, int imgWrite)
		{
			typedef typename C::TB TB;
			constexpr int P = Geo<GP>::P, R = TB::RingFrames(LN), OFF = TB::RingOff(LN), KEEP = R - FRAMES;
			static_assert(GP == TB::GPof(TB::ArrOf(LN)), "lane mode of the receiving layer");
			if (!(NA_ABL & 64) && TB::Dil(LN) < C::NF) // the next layer reads in-block frames of other lanes
				LdsWrite16(ln.img + (unsigned)(imgWrite * C::IMG_ONE + 16 * P * i * 16), v);
			if ((NA_ABL & 4) || !StoreNeeded<C, LN>(i)) return;
			const int pos0 = __builtin_amdgcn_readlane(cx.myPos, LN);
			int base = pos0 + 32 * cx.wave + 16 * P * i; // < 2R
			if (base >= R) base -= R;
			const int addr = RingWrap<GP, R>(ln.ring, base);
			if (KEEP >= C::NF) RingStore(cx.srsrc, v, addr, OFF * 16);
			else RingStore(cx.srsrc, v, (32 * cx.wave + 16 * P * i + ln.fl >= C::NF - KEEP) ? addr : OOB, OFF * 16);
		}

		// Stage s + 1's A operands -> the other LDS weight buffer by LDS-DMA (lane l's 16 bytes land at base + 16 l; no VGPRs, no ds_write),
		// issued at the start of stage s, awaited just before its closing barrier.  Operand count and offsets are constants: exactly
		// ceil(ops * 64 / NTHREADS) loads per thread, the last one partly out of range.
		template <class C, int SN>
		struct Stager
		{
			static constexpr int QUADS = C::TB::StageOps(SN) * 64;
			static constexpr int NCOPY = (QUADS + C::NTHREADS - 1) / C::NTHREADS;
			static __device__ __forceinline__ void Begin(const Ctx& cx)
			{
				if (NA_ABL & 16) return;
#pragma unroll
				for (int c = 0; c < NCOPY; c++)
				{
					// operands are 64 quads: a wave's 1 KB slice is one whole operand or lies beyond the block.  An out-of-range LDS-DMA load still
					// WRITES (zeros), so a wave with nothing to stage aims at the dump slot -- same instruction count on every wave.
					const int i0 = c * C::NTHREADS + cx.waveAll * 64; // first quad of this wave's slice (wave-uniform)
					const bool mine = i0 < QUADS;
					const unsigned dst = mine ? (unsigned)(C::WBUF_OFF + (SN & 1) * C::WBUF_ONE) + (unsigned)i0 * 16u : (unsigned)C::DUMP_OFF;
					__builtin_amdgcn_raw_ptr_buffer_load_lds(cx.wrsrc, (__attribute__((address_space(3))) void*)(LdsPtr)(size_t)dst, 16,
						mine ? (C::TB::AOff(SN) + i0 + cx.lane) * 16 : OOB, 0, 0, 0);
				}
			}
			// LATER = VMEM operations this wave issued after Begin() (they may stay in flight)
			template <int LATER>
			static __device__ __forceinline__ void End()
			{
				// gfx9 s_waitcnt: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "don't wait"
				__builtin_amdgcn_s_waitcnt((LATER & 15) | ((LATER >> 4) << 14) | (7 << 4) | (15 << 8));
			}
		};

		template <class C>
		__device__ __forceinline__ u32x4 WOp(const Ctx& cx, int s, int m)
		{
			return LdsRead16((unsigned)(C::WBUF_OFF + (s & 1) * C::WBUF_ONE + ((NA_ABL & 128) ? 0 : m) * 1024) + (unsigned)cx.lane * 16u);
		}

		// ---- one layer (WaveNetLayerT::Process, WaveNet.h:462-494) for the waves whose tap classes are those of wave WR ----------------
		template <class C, int L, int WR>
		__device__ __forceinline__ void LayerBody(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::ArrOf(L))>& ln, State& st)
		{
			typedef typename C::TB TB;
			typedef LayerSig<C, L> SG;
			constexpr int GP = SG::GP, P = SG::P, S = SG::S, d = SG::d, s = TB::StageOfLayer(L);
			constexpr int imgRead = s & 1, imgWrite = (s + 1) & 1;
			SPK_STAMP(s, 0);
			Stager<C, s + 1>::Begin(cx);

			u32x4 ax[S];
#pragma unroll
			for (int i = 0; i < S; i++) ax[i] = AuxRead<C, GP>(ln, i);

			// dilated conv (WaveNet.h:139-290): tap k reads the frame d (2 - k) back; bias and mix-in arrive through the aux operand
			f32x4 acc[S];
#pragma unroll
			for (int i = 0; i < S; i++) acc[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
			for (int k = 0; k < 2; k++)
			{
				constexpr int dd = d;
				const int shift = dd * (2 - k);
				const u32x4 ah = WOp<C>(cx, s, 2 * k), al = WOp<C>(cx, s, 2 * k + 1);
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					const int cls = TapClassOf(32 * WR, P, i, shift);
					if (cls != TAP_LDS)
					{
						acc[i] = Mfma(ah, st.hist[k][i], acc[i]);
						acc[i] = Mfma(al, st.hist[k][i], acc[i]);
					}
					if (cls != TAP_HIST)
					{
						u32x4 b;
						if (NA_ABL & 256) b = u32x4{ (unsigned)shift, 0, 0, 0 };
						else if (cls == TAP_LDS) b = LdsRead16(ln.img + (unsigned)(imgRead * C::IMG_ONE) + (unsigned)((16 * P * i - shift) * 16));
						else
						{
							// straddling (wave 0, shift < 16 P): lanes whose frame lies before the block read the zero guard quad in front of frame 0
							int off = 16 * P * i + ln.fl - shift; // the wave is wave 0: F0 = 0
							off = off < -1 ? -1 : off;
							b = LdsRead16((unsigned)(C::IMG_OFF + imgRead * C::IMG_ONE + GUARD * 16) + (unsigned)(cx.sub * 2 * C::IMG_ONE) + (unsigned)((ln.cg * PLANE + off) * 16));
						}
						acc[i] = Mfma(ah, b, acc[i]);
						acc[i] = Mfma(al, b, acc[i]);
					}
				}
			}
			// history of the NEXT layer's shifted taps (the registers are free again)
			if constexpr (SG::NEXT) HistPrefetch<C, SG::NEXT ? L + 1 : L, WR>(cx, ln.ring, ln.fl, st);
			{
				// unshifted tap = the layer input itself (registers) and the aux operand: (mix-in, conv bias) * (cond, 1)   (:288-289, :471)
				const u32x4 ah = WOp<C>(cx, s, 4), al = WOp<C>(cx, s, 5), xa = WOp<C>(cx, s, 6);
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					acc[i] = Mfma(ah, st.xs[i], acc[i]);
					acc[i] = Mfma(al, st.xs[i], acc[i]);
					acc[i] = Mfma(xa, ax[i], acc[i]);
				}
			}
			SPK_STAMP(s, 1);
			// activation (:473-480)
			f32x4 z[S];
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
			SPK_STAMP(s, 2);
			// head accumulate (:482) on the matrix pipe: head += I (zh + zl); 1x1 + bias + residual (:486-491)
			{
				const u32x4 idop = LdsRead16((unsigned)C::IDOP_OFF + (unsigned)cx.lane * 16u);
				const u32x4 w1h = WOp<C>(cx, s, 7), w1l = WOp<C>(cx, s, 8), b1a = WOp<C>(cx, s, 9);
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					const u32x4 zs = SplitQuad(z[i]);
					st.hd[i] = Mfma(idop, zs, st.hd[i]);
					if constexpr (!(TB::LastOfArr(L) && TB::ArrOf(L) == TB::NA - 1)) // NeedOutput (WaveNet.h:643,785): the very last layer's 1x1 is dead
					{
						f32x4 y = st.xc[i];
						y = Mfma(w1h, zs, y);
						y = Mfma(w1l, zs, y);
						y = Mfma(b1a, ax[i], y);
						st.xc[i] = y;
						if constexpr (SG::NEXT)
						{
							st.xs[i] = SplitQuad(y);
							Publish<C, SG::NEXT ? L + 1 : L, GP>(cx, ln, st.xs[i], i, imgWrite);
						}
					}
				}
			}
			SPK_STAMP(s, 3);
			// the DMA data must be in LDS before the closing barrier lets other waves read it
			Stager<C, s + 1>::template End<(SG::NEXT ? HistLoadsOf<C, SG::NEXT ? L + 1 : L>() + StoresOf<C, SG::NEXT ? L + 1 : L>() : 0)>();
			SPK_STAMP(s, 4);
			BlockBarrier<C::NTHREADS / 64>();
			SPK_STAMP(s, 5);
		}

		template <class C, int L, int W>
		__device__ __forceinline__ void LayerDispatch(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::ArrOf(L))>& ln, State& st)
		{
			typedef LayerSig<C, L> SG;
			if constexpr (W < C::WPS)
			{
				if constexpr (SG::Rep(W) != W) LayerDispatch<C, L, W + 1>(cx, ln, st);
				else if constexpr (SG::LastRep(W)) LayerBody<C, L, W>(cx, ln, st);
				else
				{
					if ((SG::MaskOf(W) >> cx.wave) & 1u) LayerBody<C, L, W>(cx, ln, st);
					else LayerDispatch<C, L, W + 1>(cx, ln, st);
				}
			}
		}

		// history prefetch of the first layer of an array (issued by the rechannel / link stage in front of it): per wave class
		template <class C, int L, int W>
		__device__ __forceinline__ void FirstHistDispatch(const Ctx& cx, unsigned laneRing, int fl, State& st)
		{
			typedef typename C::TB TB;
			constexpr int P = Geo<TB::GPof(TB::ArrOf(L))>::P, S = Geo<TB::GPof(TB::ArrOf(L))>::S;
			if constexpr (W < C::WPS)
			{
				constexpr bool same = [] {
					bool r = true;
					for (int w = W + 1; w < C::WPS; w++)
						for (int k = 0; k < 2; k++)
							for (int i = 0; i < S; i++) r = r && TapClassOf(32 * w, P, i, TB::Dil(L) * (2 - k)) == TapClassOf(32 * W, P, i, TB::Dil(L) * (2 - k));
					return r;
				}();
				if constexpr (same) HistPrefetch<C, L, W>(cx, laneRing, fl, st);
				else
				{
					if (cx.wave == W) HistPrefetch<C, L, W>(cx, laneRing, fl, st);
					else FirstHistDispatch<C, L, W + 1>(cx, laneRing, fl, st);
				}
			}
		}

		// array 0 rechannel: x = w_re * cond (WaveNet.h:637 with InputSize == 1) -- the aux operand against (w_re, 0)
		template <class C>
		__device__ __forceinline__ void RechStage(const Ctx& cx, const Lanes<C, C::TB::GPof(0)>& ln, State& st)
		{
			constexpr int GP = C::TB::GPof(0), S = Geo<GP>::S;
			SPK_STAMP(0, 0);
			Stager<C, 1>::Begin(cx);
			const u32x4 ra = WOp<C>(cx, 0, 0);
#pragma unroll
			for (int i = 0; i < S; i++)
			{
				const u32x4 ax = AuxRead<C, GP>(ln, i);
				f32x4 x = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				x = Mfma(ra, ax, x);
				st.xc[i] = x;
				st.hd[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; // WaveNet.h:772 headArray.SetZero()
				st.xs[i] = SplitQuad(x);
				Publish<C, 0, GP>(cx, ln, st.xs[i], i, 1);
			}
			FirstHistDispatch<C, 0, 0>(cx, ln.ring, ln.fl, st);
			SPK_STAMP(0, 1); SPK_STAMP(0, 2); SPK_STAMP(0, 3);
			Stager<C, 1>::template End<StoresOf<C, 0>() + HistLoadsOf<C, 0>()>();
			SPK_STAMP(0, 4);
			BlockBarrier<C::NTHREADS / 64>();
			SPK_STAMP(0, 5);
		}

		// array link: previous array's head rechannel (K = 1, WaveNet.h:658-660) and this array's rechannel (:637), tile by tile; the
		// operand of tile t writes the rows of tile slot t % Pn of the new mode from the k-blocks of slot t % Po of the old one
		template <class C, int AN>
		__device__ __forceinline__ void LinkStage(const Ctx& cx, const Lanes<C, C::TB::GPof(AN)>& ln, State& st)
		{
			typedef typename C::TB TB;
			constexpr int GPO = TB::GPof(AN - 1), GPN = TB::GPof(AN), Po = 4 / GPO, Pn = 4 / GPN, NC = Po > Pn ? Po : Pn;
			constexpr int So = Geo<GPO>::S, Sn = Geo<GPN>::S, s = TB::LinkStage(AN), LN = TB::FirstLayerOfArr(AN);
			SPK_STAMP(s, 0);
			Stager<C, s + 1>::Begin(cx);
			u32x4 hs[So], xq[So];
#pragma unroll
			for (int i = 0; i < So; i++)
			{
				hs[i] = SplitQuad(st.hd[i]);
				xq[i] = SplitQuad(st.xc[i]);
			}
			f32x4 hn[Sn], xn[Sn];
#pragma unroll
			for (int i = 0; i < Sn; i++)
			{
				hn[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				xn[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				// head bias of the previous array (a zero operand when it has none): any stream's aux operand carries the ones it multiplies
				hn[i] = Mfma(WOp<C>(cx, s, 4 * NC), AuxRead<C, GPN>(ln, i), hn[i]);
			}
#pragma unroll
			for (int t = 0; t < 2; t++)
			{
				const int u = t % NC, so = t / Po, sn = t / Pn;
				hn[sn] = Mfma(WOp<C>(cx, s, 4 * u), hs[so], hn[sn]);
				hn[sn] = Mfma(WOp<C>(cx, s, 4 * u + 1), hs[so], hn[sn]);
				xn[sn] = Mfma(WOp<C>(cx, s, 4 * u + 2), xq[so], xn[sn]);
				xn[sn] = Mfma(WOp<C>(cx, s, 4 * u + 3), xq[so], xn[sn]);
			}
#pragma unroll
			for (int i = 0; i < Sn; i++)
			{
				st.hd[i] = hn[i];
				st.xc[i] = xn[i];
				st.xs[i] = SplitQuad(xn[i]);
				Publish<C, LN, GPN>(cx, ln, st.xs[i], i, (s + 1) & 1);
			}
			FirstHistDispatch<C, LN, 0>(cx, ln.ring, ln.fl, st);
			SPK_STAMP(s, 1); SPK_STAMP(s, 2); SPK_STAMP(s, 3);
			Stager<C, s + 1>::template End<StoresOf<C, LN>() + HistLoadsOf<C, LN>()>();
			SPK_STAMP(s, 4);
			BlockBarrier<C::NTHREADS / 64>();
			SPK_STAMP(s, 5);
		}

		// last array's head: out = scale * (W_h head + b)[0]  (WaveNet.h:658-660, :793-798); one output row per tile slot (PK: per stream)
		template <class C>
		__device__ __forceinline__ void HeadStage(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::NA - 1)>& ln, State& st, float* __restrict__ out, size_t outBase,
			const long (&outRow)[4], int pack, float headScale, bool live)
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::GPof(TB::NA - 1), P = Geo<GP>::P, S = Geo<GP>::S, s = TB::NSTAGES - 1;
			const u32x4 ah = WOp<C>(cx, s, 0), al = WOp<C>(cx, s, 1), ba = WOp<C>(cx, s, 2);
#pragma unroll
			for (int i = 0; i < S; i++)
			{
				const u32x4 hs = SplitQuad(st.hd[i]);
				f32x4 acc = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				acc = Mfma(ah, hs, acc);
				acc = Mfma(al, hs, acc);
				acc = Mfma(ba, AuxRead<C, GP>(ln, i), acc);
				const int f = 32 * cx.wave + 16 * P * i + ln.fl;
				if (live && ln.cg == 0)
				{
					if constexpr (C::PK)
					{
						const float v[4] = { acc.x, acc.y, acc.z, acc.w };
#pragma unroll
						for (int q = 0; q < 4; q++)
							if (q < pack && outRow[q] >= 0) out[outRow[q] + f] = headScale * v[q];
					}
					else out[outBase + f] = headScale * acc.x;
				}
			}
		}

		template <class C, int L, int LEND>
		__device__ __forceinline__ void RunLayers(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::ArrOf(L))>& ln, State& st)
		{
			LayerDispatch<C, L, 0>(cx, ln, st);
			if constexpr (L + 1 < LEND) RunLayers<C, L + 1, LEND>(cx, ln, st);
		}

		template <class C, int AN>
		__device__ __forceinline__ void RunArrays(const Ctx& cx, State& st, float* __restrict__ out, size_t outBase, const long (&outRow)[4], int pack, float headScale, bool live)
		{
			typedef typename C::TB TB;
			Lanes<C, TB::GPof(AN)> ln;
			ln.Init(cx, AN == 0 ? cx.gs0 : cx.gs1);
			if constexpr (AN == 0) RechStage<C>(cx, ln, st);
			else LinkStage<C, AN>(cx, ln, st);
			RunLayers<C, TB::FirstLayerOfArr(AN), TB::FirstLayerOfArr(AN) + C::A::NLA[AN]>(cx, ln, st);
			if constexpr (AN + 1 < TB::NA) RunArrays<C, AN + 1>(cx, st, out, outBase, outRow, pack, headScale, live);
			else HeadStage<C>(cx, ln, st, out, outBase, outRow, pack, headScale, live);
		}

		// grid = active (virtual) streams / SPB; workgroup = SPB streams x WPS waves of 2 tiles; n == NF frames.  F = architecture family
		// (the groups of one launch may be different members of it).  Dynamic LDS = Cfg::LDS_BYTES, addressed absolutely from 0 (the kernel
		// has no static LDS), so every LDS offset of the chain is an instruction immediate.
		template <class F, int NF, int SPB, bool PK>
		__global__ void __launch_bounds__(64 * (NF / 32) * SPB) __attribute__((amdgpu_waves_per_eu(4))) WaveNetSpecKernel(const LaunchArgs args, const float* __restrict__ in,
			float* __restrict__ out, long inStride, long outStride
#ifdef NA_SP_TRACE
			, long long* __restrict__ trace, int traceBlock
#endif
			)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK> C;
			typedef Cfg<typename F::A1, NF, SPB, PK> C1;
			extern __shared__ __attribute__((aligned(16))) char dynSmem[];
			asm volatile("" : : "s"((unsigned)(size_t)(LdsPtr)dynSmem)); // the dynamic LDS segment is in use (and starts at 0)

			int gi = 0;
			for (int i = 1; i < args.numGroups; i++)
				if ((int)blockIdx.x >= args.g[i].firstBlock) gi = i;
			const GroupArgs& ga = args.g[gi];
			const int groupBlock = (int)blockIdx.x - ga.firstBlock;

			const int lane = threadIdx.x & 63;
			const int waveAll = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			const int sub = waveAll / C::WPS, wave = waveAll % C::WPS;

			// a partial last workgroup: the surplus waves shadow the last stream on a zero-sized state resource (they must keep staging
			// weights and meeting barriers; their loads return zeros, their stores go nowhere)
			int sidx = groupBlock * SPB + sub;
			const bool live = sidx < ga.numStreams;
			if (!live) sidx = ga.numStreams - 1;
			const int slot = ga.slots ? ga.slots[sidx] : ga.slot0 + sidx;
			const int row = PK ? 0 : (ga.slots ? ga.rows[sidx] : ga.row0 + sidx);
			u32x4* stt = ga.state + (size_t)slot * (size_t)ga.stateF4;
			int* header = reinterpret_cast<int*>(stt);

			Ctx cx;
			cx.srsrc = MakeRsrc(stt, live ? (unsigned)ga.stateF4 * 16u : 0u);
			cx.wrsrc = MakeRsrc(ga.wsplit, (unsigned)ga.wsplitQuads * 16u);
			cx.myPos = header[lane];
			cx.wave = wave; cx.sub = sub; cx.waveAll = waveAll; cx.lane = lane;
#ifdef NA_SP_TRACE
			cx.trace = ((int)blockIdx.x == traceBlock) ? trace : nullptr;
			cx.nwaves = C::NTHREADS / 64;
			if (cx.trace != nullptr && lane == 0) cx.trace[((C::TB::NSTAGES * 8 + 0) * cx.nwaves) + waveAll] = (long long)__builtin_readcyclecounter();
#endif
			// packed: channel groups per real stream = (channels / pack) / 4 -> shift (1, 2, 4 -> 0, 1, 2)
			const int pack = PK ? ga.pack : 1;
			cx.gs0 = PK ? ((ga.gps0 >> 1) & 3) : 0;
			cx.gs1 = PK ? ((ga.gps1 >> 1) & 3) : 0;
			long outRow[4];
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				const int r = (PK && live && q < ga.pack) ? ga.rows[sidx * ga.pack + q] : -1;
				outRow[q] = r >= 0 ? (long)r * outStride : -1;
			}

			// input row (WaveNet.h:770 input -> condition) -> the aux operand of every frame
			if constexpr (PK)
			{
				for (int q = 0; q < 4; q++)
				{
					const int r = (live && q < ga.pack) ? ga.rows[sidx * ga.pack + q] : -1;
					for (int i = wave * 64 + lane; i < FRAMES; i += C::WPS * 64)
					{
						const float c = (r >= 0 && i < NF) ? ClampCond(in[(size_t)r * inStride + i], ga.condLimit) : 0.0f;
						const _Float16 ch = (_Float16)c, cl = (_Float16)(c - (float)ch);
						const f16x2 a = { ch, (_Float16)1.0f }, b = { cl, (_Float16)1.0f };
						*reinterpret_cast<__attribute__((address_space(3))) u32x2*>((LdsPtr)(size_t)(unsigned)(C::AUX_OFF + ((sub * 4 + q) * FRAMES + i) * 8)) =
							u32x2{ __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b) };
					}
				}
			}
			else
			{
				for (int i = wave * 64 + lane; i < FRAMES; i += C::WPS * 64)
				{
					// [cond_h, 1 | cond_l, 1 | cond_h, 0 | 0, 0] (see FillSplitAux in wavenet_plan.cpp)
					const float c = (i < NF) ? ClampCond(in[(size_t)row * inStride + i], ga.condLimit) : 0.0f;
					const _Float16 ch = (_Float16)c, cl = (_Float16)(c - (float)ch);
					const f16x2 a = { ch, (_Float16)1.0f }, b = { cl, (_Float16)1.0f }, d = { ch, (_Float16)0.0f };
					LdsWrite16((unsigned)(C::AUX_OFF + (sub * FRAMES + i) * 16), u32x4{ __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, d), 0u });
				}
			}
			// zero quad in front of frame 0 of every plane of both block images
			for (int i = threadIdx.x; i < SPB * 2 * 4; i += C::NTHREADS) LdsWrite16((unsigned)(C::IMG_OFF + (i * PLANE + GUARD - 1) * 16), u32x4{ 0, 0, 0, 0 });
			// identity A operand: row i x k-block q = i / 4: 1.0 against the h AND the l half of channel i % 4
			if (threadIdx.x < 64)
			{
				const int i = lane & 15, q = lane >> 4;
				const unsigned one = 0x3c00u; // f16 1.0
				const unsigned lo = (q == (i >> 2)) ? (((i & 3) == 0) ? one : ((i & 3) == 1) ? (one << 16) : 0u) : 0u;
				const unsigned hi = (q == (i >> 2)) ? (((i & 3) == 2) ? one : ((i & 3) == 3) ? (one << 16) : 0u) : 0u;
				LdsWrite16((unsigned)C::IDOP_OFF + (unsigned)lane * 16u, u32x4{ lo, hi, lo, hi });
				// stage 0's single operand (offset 0 of every weight image)
				LdsWrite16((unsigned)C::WBUF_OFF + (unsigned)lane * 16u, BufLoad(cx.wrsrc, lane * 16));
			}
			BlockBarrier<C::NTHREADS / 64>();

			State st;
			if (F::N > 1 && ga.arch == 1) RunArrays<C1, 0>(cx, st, out, (size_t)row * outStride, outRow, pack, ga.headScale, live);
			else RunArrays<C, 0>(cx, st, out, (size_t)row * outStride, outRow, pack, ga.headScale, live);

#ifdef NA_SP_TRACE
			if (cx.trace != nullptr && lane == 0) cx.trace[((C::TB::NSTAGES * 8 + 1) * cx.nwaves) + waveAll] = (long long)__builtin_readcyclecounter();
#endif
			// advance every ring cursor by NF (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
			if (wave == 0 && live && lane < ga.nrings)
			{
				const int R = ga.ringFrames[lane];
				int p = cx.myPos + NF;
				if (p >= R) p -= R;
				header[lane] = p;
			}
		}

		// ---- host ----------------------------------------------------------------------------------------------------------------
		template <class A>
		static bool Matches(const WnSplitStage* st, int nstages, int stateF4, int wsplitQuads)
		{
			typedef Tab<A> TB;
			if (nstages != TB::NSTAGES || stateF4 != TB::StateF4 || wsplitQuads != TB::WsplitQuads) return false;
			for (int s = 0; s < nstages; s++)
				if (st[s].a_off != TB::AOff(s) || st[s].a_ops != TB::StageOps(s)) return false;
			if (st[0].type != WN_ST_RECHANNEL_COND || st[0].Gp != TB::GPof(0) || st[0].G != TB::GPof(0) || st[0].out_ring_id != 0) return false;
			for (int L = 0; L < TB::NL; L++)
			{
				const WnSplitStage& d = st[TB::StageOfLayer(L)];
				const int a = TB::ArrOf(L);
				if (d.type != WN_ST_LAYER || d.Gp != TB::GPof(a) || d.G != TB::GPof(a) || d.ksize != 3 || d.dilation != TB::Dil(L)) return false;
				if (d.ring_id != L || d.ring_off != TB::RingOff(L) || d.ring_frames != TB::RingFrames(L)) return false;
				if (d.flags & (WN_FLAG_LEAKY | WN_FLAG_STD_TANH)) return false;
				if (!TB::LastOfArr(L) && (d.out_ring_id != L + 1 || !(d.flags & WN_FLAG_PUBLISH))) return false;
			}
			for (int a = 1; a < TB::NA; a++)
			{
				const WnSplitStage& d = st[TB::LinkStage(a)];
				if (d.type != WN_ST_ARRAY_LINK || d.Gp != TB::GPof(a - 1) || d.ksize != TB::GPof(a) || d.out_ring_id != TB::FirstLayerOfArr(a)) return false;
			}
			const WnSplitStage& h = st[nstages - 1];
			return h.type == WN_ST_HEAD_DENSE_OUT && h.ksize == 1 && h.Gp == TB::GPof(TB::NA - 1);
		}

		template <class F, int NF, int SPB, bool PK>
		static hipError_t Launch(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, hipStream_t stream)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK> C;
			LaunchArgs args = {};
			args.numGroups = numGroups;
			int blocks = 0;
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
				a.numStreams = g.numStreams; a.slot0 = g.slot0; a.row0 = g.row0;
				a.maxG = m.max_G;
				a.firstBlock = blocks;
				a.pack = g.pack > 1 ? g.pack : 1;
				a.arch = (F::N > 1 && m.spec_arch == WN_SPEC_LITE16) ? 1 : 0;
				// channel groups per real stream of the first / the last array (packed launches: which stream's condition a channel group sees)
				const int c0 = a.arch == 1 ? F::A1::CH[0] : F::A0::CH[0], c1 = a.arch == 1 ? F::A1::CH[1] : F::A0::CH[1];
				a.gps0 = std::max(1, c0 / 4 / a.pack);
				a.gps1 = std::max(1, c1 / 4 / a.pack);
				if (a.pack > 1 && !PK) return hipErrorInvalidValue;
				if (PK && g.slots == nullptr) return hipErrorInvalidValue;
				blocks += (g.numStreams + SPB - 1) / SPB;
			}
			if (C::LDS_BYTES > 64 * 1024)
			{
				static bool granted = false; // per instantiation
				if (!granted)
				{
					const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(WaveNetSpecKernel<F, NF, SPB, PK>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
					if (e != hipSuccess) return e;
					granted = true;
				}
			}
			hipLaunchKernelGGL((WaveNetSpecKernel<F, NF, SPB, PK>), dim3((unsigned)blocks), dim3(64 * (NF / 32) * SPB), C::LDS_BYTES, stream, args, in, out, inStride, outStride
#ifdef NA_SP_TRACE
				, GetWaveNetTraceBuffer(), []() { const char* e = getenv("NA_TRACE_BLOCK"); return e ? atoi(e) : 0; }()
#endif
				);
			return hipGetLastError();
		}

		template <class F, bool PK>
		static hipError_t LaunchNF(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, hipStream_t stream)
		{
#ifdef NA_SP_QUICK
			(void)spb; (void)n;
			return Launch<F, 128, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream);
#else
			if (n == 128) return spb >= 2 ? Launch<F, 128, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream) : Launch<F, 128, 1, PK>(groups, numGroups, in, out, inStride, outStride, stream);
			if (n == 64) return spb >= 2 ? Launch<F, 64, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream) : Launch<F, 64, 1, PK>(groups, numGroups, in, out, inStride, outStride, stream);
			return spb >= 2 ? Launch<F, 32, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream) : Launch<F, 32, 1, PK>(groups, numGroups, in, out, inStride, outStride, stream);
#endif
		}
	}

	// NA_WN_SPEC=0 (environment) or SetWaveNetSpecEnabled(false) (tests: both kernels on one stream in one process) turn the chains off
	static int& SpecSwitch()
	{
		// (NA_SP_T / NA_SP_GEN select interpreter variants: they imply it)
		static int on = ((getenv("NA_WN_SPEC") != nullptr && atoi(getenv("NA_WN_SPEC")) == 0) || getenv("NA_SP_T") != nullptr || getenv("NA_SP_GEN") != nullptr) ? 0 : 1;
		return on;
	}
	bool WaveNetSpecEnabled() { return SpecSwitch() != 0; }
	void SetWaveNetSpecEnabled(bool on) { SpecSwitch() = on ? 1 : 0; }

	int WaveNetSpecArchId(const WnSplitStage* stages, int nstages, int stateF4, int wsplitQuads)
	{
		if (spk::Matches<spk::ArchStd>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_STD;
		if (spk::Matches<spk::ArchLite>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_LITE;
		if (spk::Matches<spk::ArchLite16>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_LITE16;
		return WN_SPEC_NONE;
	}

	// Runs the launch on a specialised chain when every group is the SAME official architecture and the block is 128 / 64 / 32 frames;
	// returns hipErrorNotSupported otherwise (the caller then uses the stage interpreter).
	hipError_t LaunchWaveNetSpecFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream)
	{
		if (numGroups <= 0 || numGroups > WN_FRAME_MAX_GROUPS || (n != 128 && n != 64 && n != 32)) return hipErrorNotSupported;
		if (!WaveNetSpecEnabled()) return hipErrorNotSupported; // tuning / tests: the interpreter for everything
		const int arch = groups[0].model->spec_arch;
		if (arch == WN_SPEC_NONE) return hipErrorNotSupported;
		const bool liteFamily = arch == WN_SPEC_LITE || arch == WN_SPEC_LITE16;
		int total = 0;
		bool packed = false;
		for (int i = 0; i < numGroups; i++)
		{
			const int a = groups[i].model->spec_arch;
			if (groups[i].numStreams <= 0 || (liteFamily ? (a != WN_SPEC_LITE && a != WN_SPEC_LITE16) : a != arch)) return hipErrorNotSupported;
			total += groups[i].numStreams;
			packed = packed || groups[i].pack > 1;
		}
		// a packed launch reads the index lists of every group (a plain group riding along is pack = 1)
		if (packed)
			for (int i = 0; i < numGroups; i++)
				if (groups[i].slots == nullptr) return hipErrorNotSupported;
		static const int spbEnv = getenv("NA_SP_SPB") ? atoi(getenv("NA_SP_SPB")) : 0;
		const int spb = spbEnv > 0 ? spbEnv : (total >= 512 ? 2 : 1);
#ifdef NA_SP_QUICK
		if (arch != WN_SPEC_STD || packed) return hipErrorNotSupported;
		return spk::LaunchNF<spk::FamStd, false>(groups, numGroups, in, out, inStride, outStride, n, spb, stream);
#else
		if (!liteFamily) return packed ? hipErrorNotSupported : spk::LaunchNF<spk::FamStd, false>(groups, numGroups, in, out, inStride, outStride, n, spb, stream);
		if (packed) return spk::LaunchNF<spk::FamLitePacked, true>(groups, numGroups, in, out, inStride, outStride, n, spb, stream);
		for (int i = 0; i < numGroups; i++)
			if (groups[i].model->spec_arch != WN_SPEC_LITE) return hipErrorNotSupported; // (16 / 16 only exists packed)
		return spk::LaunchNF<spk::FamLite, false>(groups, numGroups, in, out, inStride, outStride, n, spb, stream);
#endif
	}
}
