// wavenet_spec_impl.h -- the compile-time specialised WaveNet layer chains (device code + launch templates); see wavenet_spec_kernels.hip
// for the description.  Included by one translation unit per architecture family (their kernels are compiled with different scheduler
// options, and in parallel):
//   wavenet_spec_kernels.hip       FamStd (+ the host-side dispatch)
//   wavenet_spec_lite_kernels.hip  FamLite, FamLitePacked
//   wavenet_spec_a2_kernels.hip    FamA2
#pragma once

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "device_once.h"
#include <cstring>
#include <type_traits>
#include <vector>

#include "tuning.h"
#include "wavenet_split_dev.h"

namespace na
{
	namespace spk
	{
		using namespace sp;

// tuning builds: cache policy of the ring traffic of layers with a dilation >= NA_SPK_NT_DIL -- NA_SPK_NT_LD / NA_SPK_NT_ST are the
// policy immediates of their history loads / ring stores (0 = default policy; 2 = nt; 16 = sc1; 18 = both)
#ifndef NA_SPK_NT_DIL
#define NA_SPK_NT_DIL 256
#endif
#ifndef NA_SPK_NT_LD
#define NA_SPK_NT_LD 0
#endif
#ifndef NA_SPK_NT_ST
#define NA_SPK_NT_ST 0
#endif
#ifndef NA_SPK_DELAY
#define NA_SPK_DELAY 0
#endif
#ifndef NA_SPK_AUX2
#define NA_SPK_AUX2 0 // tuning builds: 1 = read the aux operand again for the 1x1 instead of keeping it in registers across the layer
#endif
		typedef __attribute__((address_space(3))) char* LdsPtr;
		__device__ __forceinline__ u32x4 LdsRead16(unsigned addr) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>((LdsPtr)(size_t)addr); }
		__device__ __forceinline__ u32x2 LdsRead8(unsigned addr) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x2*>((LdsPtr)(size_t)addr); }
		__device__ __forceinline__ void LdsWrite16(unsigned addr, u32x4 v) { *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((LdsPtr)(size_t)addr) = v; }

		// input / output rows of a launch: plain accesses between kernel boundaries; at system scope inside a resident launch
		template <bool COH>
		__device__ __forceinline__ float LoadIn(const float* p)
		{
			if constexpr (COH) return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
			else return *p;
		}
		template <bool COH>
		__device__ __forceinline__ void StoreOut(float* p, float v)
		{
			if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			else *p = v;
		}

		// ---- the architectures (virtual models: after padding / stream packing, wavenet_plan.cpp) ---------------------------------
		// NeuralModel.cpp:71-76 dilation tables; channels are those of the lane modes the plans fill completely
		// every A1 architecture: kernel size 3 everywhere, dense head, tanh, 2 tiles per wave, stage operand blocks of <= 10 KB
		struct ArchA1Base
		{
			static constexpr int K(int, int) { return 3; }
			static constexpr int HEADK = 1;
			static constexpr bool LEAKY = false;
			static constexpr int T = 2, CHUNK = 10;
			static constexpr bool COARSE = false; // (dilations are powers of two: at most three wave classes per layer, 53 KB of code)
			static constexpr bool GUARDHIST = false;
			static constexpr int SKEW = 0;
			static constexpr bool COMPACT = true; // K <= 3 everywhere, dense heads: histories of <= 32 frames live in compact rings (wavenet_plan.cpp AddRing)
#ifndef NA_A1_NT_DIL
#define NA_A1_NT_DIL 128
#endif
			static constexpr int NT_DIL = NA_A1_NT_DIL; // Cfg::NT: layers from this dilation on move their ring traffic non-temporally
		};
#ifndef NA_SPK_SKEW
#define NA_SPK_SKEW 0
#endif
		struct ArchStd : ArchA1Base // A1 Standard (16 -> 8)
		{
			// Tuning experiment (make KEXTRA=-DNA_SPK_SKEW=5; NOT the default): the two streams of a workgroup run SKEW stages apart
			// (Cfg::SKEW).  The layers of an array are light (d <= 32: bound by instruction issue) and then heavy (d >= 64: both taps of
			// every frame come from HBM), and in lock-step every workgroup of the launch is in the same phase -- the memory system idles
			// through the light layers and the SIMDs through the heavy ones; five stages apart, one stream's heavy layers fall on the other's
			// light ones.  Measured: 44.3 us against 43.3 in lock-step (profiles/r03_ablation.txt).  The memory system is then loaded all the
			// time, and the few ring loads of a LIGHT stage -- prefetched one stage ahead, all the 128 VGPRs allow -- come back after
			// ~1.5 us instead of ~0.5: every slot waits for its light stream (per-slot timeline: both streams done after ~2700 cycles,
			// barrier released after ~4400).
			static constexpr int SKEW = NA_SPK_SKEW;
			static constexpr int NA = 2;
			static constexpr int CH[2] = { 16, 8 };
			static constexpr int NLA[2] = { 10, 10 };
			static constexpr int DIL[2][16] = { { 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 }, { 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 } };
		};
		struct ArchLite : ArchA1Base // the "lite" dilation lists at 16 / 8 channels: A1 Lite padded (12 / 6), two Feather streams packed (8 / 4 each)
		{
			static constexpr int NA = 2;
			static constexpr int CH[2] = { 16, 8 };
			static constexpr int NLA[2] = { 7, 13 };
			static constexpr int DIL[2][16] = { { 1, 2, 4, 8, 16, 32, 64 }, { 128, 256, 512, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 } };
		};
		struct ArchLite16 : ArchA1Base // ... at 16 / 16 channels: four Nano streams packed (4 / 2 each, the second array padded to 4 per stream)
		{
			static constexpr int NA = 2;
			static constexpr int CH[2] = { 16, 16 };
			static constexpr int NLA[2] = { 7, 13 };
			static constexpr int DIL[2][16] = { { 1, 2, 4, 8, 16, 32, 64 }, { 128, 256, 512, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 } };
		};

		// The same with ONE tile per wave: eight waves per 128-frame stream instead of four.  For launches of at most one virtual stream per
		// CU (Nano x <= 1024): such a launch is a chain of 23 short stages bound by what its few waves can issue, and half the work per
		// wave halves the stage (both arrays are 16 channels wide: every tile has an MFMA of its own, so nothing is shared between the
		// tiles of a wave that splitting them would lose).  Same state, same arithmetic per frame.
		struct ArchLite16T1 : ArchLite16 { static constexpr int T = 1; };

		// A2 (NeuralModel.cpp:389-421, InternalModel.h:18-20): one array of 23 layers, kernel sizes 6 / 15, a conv head of 16 taps with bias,
		// LeakyReLU.  "Full": 8 channels (lane mode 2, 2 tiles per wave).  "Lite": 3 channels padded to 4 (lane mode 1: four tiles share
		// an MFMA, so a wave owns 4 tiles = 64 frames and a stream takes half the waves).  Operand blocks move through LDS in chunks of
		// <= 16 KB (a K = 6 layer is one chunk, a K = 15 layer three).
		struct ArchA2Base
		{
			static constexpr int NA = 1;
			static constexpr int NLA[2] = { 23, 0 };
			static constexpr int DIL[2][32] = { { 1, 3, 7, 17, 41, 101, 239, 1, 3, 7, 17, 41, 101, 239, 1, 13, 1, 3, 7, 17, 41, 101, 239 }, { 0 } };
			static constexpr int K(int, int l) { return (l == 14 || l == 15) ? 15 : 6; }
			static constexpr int HEADK = 16;
			static constexpr bool LEAKY = true;
			static constexpr int CHUNK = 16;
			static constexpr bool COARSE = true;
			static constexpr bool GUARDHIST = true;
			static constexpr int SKEW = 0;
			static constexpr bool COMPACT = false;
			static constexpr int NT_DIL = 100;    // (d = 101 and 239)
		};
		struct ArchA2Full : ArchA2Base { static constexpr int CH[2] = { 8, 0 }; static constexpr int T = 2; };
		struct ArchA2Lite : ArchA2Base { static constexpr int CH[2] = { 4, 0 }; static constexpr int T = 4; };

		// architectures that may share one launch (same stage count, same LDS map): GroupArgs::arch picks the member per workgroup
		struct FamStd { typedef ArchStd A0; typedef ArchStd A1; static constexpr int N = 1; };
		struct FamLite { typedef ArchLite A0; typedef ArchLite A1; static constexpr int N = 1; };
		struct FamLitePacked { typedef ArchLite A0; typedef ArchLite16 A1; static constexpr int N = 2; };
		struct FamA2 { typedef ArchA2Full A0; typedef ArchA2Lite A1; static constexpr int N = 2; };
		struct FamLite16T1 { typedef ArchLite16T1 A0; typedef ArchLite16T1 A1; static constexpr int N = 1; };

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
			static constexpr int KS(int L) { return A::K(ArrOf(L), InArr(L)); }
			static constexpr int HEADK = A::HEADK;
			static constexpr int NRINGS = NL + (HEADK > 1 ? 1 : 0); // one per layer (+ the conv head's)
			// ring L < NL: input history of layer L; ring NL: the head accumulator's (wavenet_plan.cpp AddRing: roundup16((K - 1) d) + 128)
			// (wavenet_plan.cpp AddRing: a dilation of at least a whole block -> exactly (K - 1) d frames, else roundup16((K - 1) d) + 128)
			static constexpr bool ExactRing(int L) { return L < NL && !FirstOfArr(L) && Dil(L) >= FRAMES && (KS(L) - 1) * Dil(L) >= 2 * FRAMES && ((KS(L) - 1) * Dil(L)) % 16 == 0; }
			static constexpr int HistFrames(int L) { return L < NL ? ((KS(L) - 1) * Dil(L) + 15) / 16 * 16 : (HEADK - 1 + 15) / 16 * 16; } // roundup16((K - 1) d)
			static constexpr bool CompactRing(int L) { return A::COMPACT && L < NL && KS(L) > 1 && HistFrames(L) <= WN_COMPACT_MAX_HISTORY; }
			static constexpr int RingFrames(int L)
			{
				if (ExactRing(L)) return (KS(L) - 1) * Dil(L);
				if (CompactRing(L)) return 3 * HistFrames(L);
				return HistFrames(L) + FRAMES;
			}
			// frames of a block a later block can still read: the whole history of an exact ring, R - 128 of the others
			static constexpr int RingKeep(int L) { return ExactRing(L) ? RingFrames(L) : (CompactRing(L) ? HistFrames(L) : RingFrames(L) - FRAMES); }
			static constexpr int RingG(int L) { return GPof(L < NL ? ArrOf(L) : NA - 1); }
			static constexpr int RingOff(int L) // quads
			{
				int o = WN_HEADER_F4;
				for (int l = 0; l < L; l++) o += RingFrames(l) * RingG(l);
				return o;
			}
			static constexpr int StateF4 = (RingOff(NRINGS) + 15) / 16 * 16;
			static constexpr int StageOfLayer(int L) { return 1 + L + ArrOf(L); }
			static constexpr int FirstLayerOfArr(int a) { int L = 0; for (int i = 0; i < a; i++) L += A::NLA[i]; return L; }
			static constexpr int LinkStage(int a) { return FirstLayerOfArr(a) + a; } // the link in front of array a >= 1
			static constexpr int LinkNC(int a) { const int Po = 4 / GPof(a - 1), Pn = 4 / GPof(a); return Po > Pn ? Po : Pn; }
			static constexpr int LayerOfStage(int s) // -1: not a layer stage
			{
				for (int L = 0; L < NL; L++)
					if (StageOfLayer(L) == s) return L;
				return -1;
			}
			static constexpr int StageOps(int s)
			{
				if (s == 0) return 1;
				if (s == NSTAGES - 1) return 2 * HEADK + 1;
				for (int a = 1; a < NA; a++)
					if (s == LinkStage(a)) return 4 * LinkNC(a) + 1;
				return 2 * KS(LayerOfStage(s)) + 4;
			}
			static constexpr int AOff(int s) { int o = 0; for (int i = 0; i < s; i++) o += StageOps(i) * 64; return o; } // quads
			static constexpr int WsplitQuads = AOff(NSTAGES);
			// A stage's operands move through LDS in chunks of <= CHUNK operands (a 1 KB operand each).  A block that fits is one chunk;
			// a larger one (K = 15 layer: 34, conv head: 33) is cut into tap chunks -- whole (hi, lo) pairs, evenly sized -- and a tail
			// chunk with what follows the taps (aux / 1x1 operands, head bias).
			static constexpr int CHUNK = A::CHUNK;
			static constexpr int TapOps(int s) { return s == NSTAGES - 1 ? 2 * HEADK : 2 * KS(LayerOfStage(s)); }
			static constexpr bool Chunked(int s) { return StageOps(s) > CHUNK; }
			static constexpr int TapChunks(int s) { return (TapOps(s) + CHUNK - 1) / CHUNK; }
			static constexpr int NumChunks(int s) { return Chunked(s) ? TapChunks(s) + 1 : 1; }
			static constexpr int ChunkBegin(int s, int c)
			{
				if (!Chunked(s)) return 0;
				if (c >= TapChunks(s)) return TapOps(s);
				const int pairs = TapOps(s) / 2, n = TapChunks(s);
				return 2 * ((pairs * c + n - 1) / n); // pairs split as evenly as possible
			}
			static constexpr int ChunkEnd(int s, int c) { return !Chunked(s) ? StageOps(s) : (c >= TapChunks(s) ? StageOps(s) : ChunkBegin(s, c + 1)); }
			static constexpr int ChunkIndex(int s, int c) { int n = 0; for (int i = 0; i < s; i++) n += NumChunks(i); return n + c; } // LDS buffer = index & 1
			static constexpr int MaxChunkOps()
			{
				int m = 0;
				for (int s = 0; s < NSTAGES; s++)
					for (int c = 0; c < NumChunks(s); c++) m = (ChunkEnd(s, c) - ChunkBegin(s, c)) > m ? (ChunkEnd(s, c) - ChunkBegin(s, c)) : m;
				return m;
			}
			static constexpr int MaxGP() { int m = 0; for (int a = 0; a < NA; a++) m = GPof(a) > m ? GPof(a) : m; return m; }
		};

		// launch shape: NF frames per block, T tiles per wave (FW = 16 T frames), SPB streams per workgroup sharing the staged weights.
		// SPB_ counts streams of 4 waves-per-128-frames (T = 2): an architecture with T = 4 takes half the waves per stream and puts twice
		// the streams into the workgroup, so every member of a family launches the same number of threads.
		// COH_: the resident launch (WaveNetSpecResidentKernel) -- input rows are read and output rows written at system scope, because
		// their producer / consumer runs while this launch is on the chip (no kernel boundary orders the caches)
		// NT_: the launch of a batch whose stream state does not fit the 256 MB Infinity Cache -- the ring traffic of the d >= 128 layers is
		// marked non-temporal (nothing of it is read again before the whole state has passed through the cache: 38.9 -> 38.3 us per 1024
		// streams with 8192 of them, profiles/r05_rotation_cache_policy.txt; inside the cache the same bits cost 2 - 3 us)
		template <class A_, int NF_, int SPB_, bool PK_, bool COH_ = false, bool NT_ = false>
		struct Cfg
		{
			static constexpr bool COH = COH_;
			static constexpr bool NT = NT_;
			typedef A_ A;
			typedef Tab<A_> TB;
			static constexpr int NF = NF_, T = A_::T, FW = 16 * T, WPS = NF_ / FW, SPB = SPB_ * T / 2, NTHREADS = 64 * WPS * SPB;
			static_assert(WPS >= 1 && WPS * FW == NF_, "block length is a whole number of waves");
			static constexpr bool PK = PK_;
			static constexpr int MAXOPS = A_::CHUNK; // operands per LDS weight buffer
			static_assert(TB::MaxChunkOps() <= MAXOPS, "stage operand chunk");
			static constexpr int HPF = 5;            // shifted taps whose ring history is prefetched a layer ahead (K = 3: both, K = 6: all five)
			// stages between the two streams of a workgroup (0: lock-step, one set of weight buffers for the workgroup); skewed streams stage
			// their own operands with their own waves into their own pair of buffers
			static constexpr int SKEW = (SPB == 2 && !PK_ && T == 2 && A_::SKEW > 0) ? A_::SKEW : 0;
			static constexpr int NWB = SKEW > 0 ? SPB : 1;          // weight buffer pairs
			static constexpr int STG_THREADS = NTHREADS / NWB;      // threads that stage one pair
			// LDS map (bytes)
			static constexpr bool AUX16 = !PK_ && SKEW == 0;                    // aux entries as whole quads (one LDS read, no unpacking)
			static constexpr int AUX_OFF = 0;                                   // [SPB][FRAMES] quads; PK: [SPB][4][FRAMES] x 8 bytes; skewed: [SPB][FRAMES] x 8
			static constexpr int IMG_OFF = AUX_OFF + SPB * FRAMES * (PK_ ? 32 : (AUX16 ? 16 : 8)); // [SPB][2][planes][PLANE] quads
			static constexpr int IMG_ONE = TB::MaxGP() * PLANE * 16;            // one image: a plane per channel group
			static constexpr int WBUF_OFF = IMG_OFF + SPB * 2 * IMG_ONE;        // [NWB][2][MAXOPS] operands of 1 KB
			static constexpr int WBUF_ONE = MAXOPS * 1024;
			static constexpr int IDOP_OFF = WBUF_OFF + NWB * 2 * WBUF_ONE;      // identity operand
			static constexpr int DUMP_OFF = IDOP_OFF + 1024;                    // where the LDS-DMA of a wave with nothing to stage lands
			static constexpr int FLAG_OFF = DUMP_OFF + 1024;                    // [SPB] x 16 bytes: "a value of this stream was saturated" (LeakyReLU chains)
			static constexpr int BCAST_OFF = FLAG_OFF + 16 * SPB;                // resident launch: the command wave 0 read, for the other waves (64 bytes)
			static constexpr int LDS_BYTES = BCAST_OFF + (COH_ ? 64 : 0);
			static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
		};

		// geometry of lane mode GP with T tiles per wave: P tiles share one MFMA ("set"), S sets per wave
		template <int GP, int T>
		struct Geo
		{
			static_assert(GP == 4 || GP == 2 || GP == 1, "full lane modes of 16 / 8 / 4 channels");
			static constexpr int P = 4 / GP;
			static_assert(T % P == 0, "every tile slot of a set is a tile of the wave");
			static constexpr int S = T / P;
		};

		// Where do the frames [F0 + 16 P i, + 16 P) of set i of the wave starting at F0 lie relative to the block start, `shift` frames back?
		constexpr int TapClassOf(int F0, int P, int i, int shift)
		{
			const int lo = F0 + 16 * P * i - shift, hi = lo + 16 * P - 1;
			return hi < 0 ? TAP_HIST : (lo >= 0 ? TAP_LDS : TAP_BOTH);
		}

		// The class a wave uses for a tap.  Exact per wave by default.  Architectures whose dilations are not tile multiples (A2) would get a
		// different tap pattern -- hence a different unrolled body -- on nearly every wave of a layer: four bodies per layer, 166 KB of code
		// against a 64 KB instruction cache shared by two CUs (measured: waves waiting thousands of cycles for their body's first fetch,
		// A2 "Full" 164 us per 1024 streams).  With COARSE wave 0 keeps its exact classes and waves 1 .. share one body: a tap they do not
		// agree on runs as TAP_BOTH on all of them (always correct: predicated ring load + clamped LDS read, two more MFMAs).
		// A ring whose reader looks at most GUARD = 16 frames back (d (K - 1) <= 16: the first layers of every array, the conv head): the
		// stage in front of the reader copies those 16 frames from the ring into the guard quads in front of frame 0 of the LDS image, so
		// EVERY tap of EVERY wave is a plain LDS read -- no per-tap history loads, no straddling taps, one body for all waves.
		// Only where a stage is long enough to hide the copy: it reads frames the PREVIOUS launch stored (always an HBM access) and has to be
		// in LDS by the end of the stage in front of the reader -- one stage of latency hiding instead of the two a register prefetch gets.
		// Measured: A2 (K = 6 / 15 layers) config 5 104.6 -> 87.9 us; A1 Standard (K = 3: short stages) 41.6 -> 48.3 us.  So A::GUARDHIST.
		template <class C, int RG>
		constexpr bool SmallRing() { return C::A::GUARDHIST && C::TB::RingKeep(RG) <= GUARD; }

		template <class C>
		constexpr int WaveTapClass(int w, int P, int i, int shift, bool small)
		{
			if (small) return TAP_LDS;
			if (!C::A::COARSE || w == 0 || C::WPS <= 2) return TapClassOf(C::FW * w, P, i, shift);
			const int first = TapClassOf(C::FW * 1, P, i, shift);
			for (int v = 2; v < C::WPS; v++)
				if (TapClassOf(C::FW * v, P, i, shift) != first) return TAP_BOTH;
			return first;
		}

		// everything about layer L that depends on the wave: classes of its shifted taps and of the next layer's prefetched ones
		template <class C, int L>
		struct LayerSig
		{
			typedef typename C::TB TB;
			static constexpr int GP = TB::GPof(TB::ArrOf(L)), P = Geo<GP, C::T>::P, S = Geo<GP, C::T>::S, d = TB::Dil(L), K = TB::KS(L);
			static constexpr bool NEXT = !TB::LastOfArr(L); // a layer of the same array follows (its history is requested during this one)
			static constexpr int LN = NEXT ? L + 1 : L;
			static constexpr int dn = NEXT ? TB::Dil(LN) : 0, Kn = NEXT ? TB::KS(LN) : 1;
			static constexpr unsigned long long Of(int w)
			{
				unsigned long long s = 0;
				for (int k = 0; k < K - 1; k++)
					for (int i = 0; i < S; i++) s = s * 3 + (unsigned)WaveTapClass<C>(w, P, i, d * (K - 1 - k), SmallRing<C, L>());
				for (int k = 0; k < Kn - 1 && k < C::HPF; k++)
					for (int i = 0; i < S; i++) s = s * 3 + (unsigned)WaveTapClass<C>(w, P, i, dn * (Kn - 1 - k), SmallRing<C, LN>());
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
			unsigned wbuf;                // LDS byte address of this wave's weight buffer pair (skewed streams: the stream's own)
			int stgWave;                  // wave index among the waves that stage into that pair
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
			int fl;          // frame within the wave's FW, set 0: 16 p + j
			int cg;          // channel group
			unsigned img;    // LDS byte address of (plane cg, frame F0 + fl) in image 0 of this stream
			unsigned aux;    // LDS byte address of the aux entry of frame F0 + fl (PK: of the stream owning channel group cg; dense: of the first of its two streams)
			unsigned ring;   // byte offset of (frame fl, group cg) within a frame-major ring of this mode: (fl * GP + cg) * 16
			bool dual;       // PK, dense pack (gs < 0): two streams share a channel group -- the aux operand carries both conditions (wave-uniform)

			__device__ __forceinline__ void Init(const Ctx& cx, int gs)
			{
				const int q = cx.lane >> 4, j = cx.lane & 15;
				const int p = q / GP;
				cg = q % GP;
				fl = 16 * p + j;
				const int f = C::FW * cx.wave + fl;
				img = (unsigned)(C::IMG_OFF + cx.sub * 2 * C::IMG_ONE + (cg * PLANE + GUARD + f) * 16);
				ring = (unsigned)((fl * GP + cg) * 16);
				dual = C::PK && gs < 0;
				if constexpr (C::PK) aux = (unsigned)(C::AUX_OFF + ((cx.sub * 4 + (gs < 0 ? 2 * cg : (cg >> gs))) * FRAMES + f) * 8);
				else aux = (unsigned)(C::AUX_OFF + (cx.sub * FRAMES + f) * (C::AUX16 ? 16 : 8));
			}
		};

		template <class C, int GP>
		__device__ __forceinline__ u32x4 AuxRead(const Lanes<C, GP>& ln, int i)
		{
			constexpr int P = Geo<GP, C::T>::P;
			// [cond_h, 1 | cond_l, 1 | cond_h, 0 | 0, 0] (see FillSplitAux in wavenet_plan.cpp): the whole quad, or from the 8 bytes kept per frame
			if constexpr (C::AUX16) return LdsRead16(ln.aux + (unsigned)(16 * P * i * 16));
			else
			{
				const u32x2 v = LdsRead8(ln.aux + (unsigned)(16 * P * i * 8));
				if constexpr (C::PK)
				{
					if (ln.dual)
					{
						// [cA_h, 1 | cA_l, 1 | cA_h, cB_h | cB_l, cB_h]: the second stream of the pair is FRAMES entries on (FillSplitAux, dense packs)
						const u32x2 w = LdsRead8(ln.aux + (unsigned)(16 * P * i * 8) + (unsigned)(FRAMES * 8));
						return u32x4{ v.x, v.y, (v.x & 0xffffu) | (w.x << 16), (w.y & 0xffffu) | (w.x << 16) };
					}
				}
				return u32x4{ v.x, v.y, v.x & 0xffffu, 0u };
			}
		}

		struct State
		{
			f32x4 xc[2];       // layer input (residual stream), f32
			f32x4 hd[2];       // head accumulator
			u32x4 xs[2];       // split quad of xc: the unshifted conv tap's operand
			u32x4 hist[5][2];  // ring history of the current layer's first HPF shifted taps [tap][set]
		};

		// f32 quad -> split quad: plain for the tanh chains (covered by the static range proof, wavenet_plan.cpp), saturating for the
		// LeakyReLU ones
		template <class C>
		__device__ __forceinline__ u32x4 Split(f32x4 v, const Ctx& cx)
		{
#ifdef NA_NO_SAT // tuning builds: what the saturating split costs the A2 chains (make SUFFIX=_nosat KEXTRA=-DNA_NO_SAT)
			return SplitQuad(v);
#else
			if constexpr (C::A::LEAKY) return SplitQuadSat(v, (unsigned)(C::FLAG_OFF + 16 * cx.sub));
			else return SplitQuad(v);
#endif
		}

		// Byte offset of ring position (base + fl) mod R, channel group cg, relative to the ring's start: `base` in [0, R) is wave-uniform, the
		// lane part fl < FW is folded into Lanes::ring, so the wrap is one unsigned min on the byte offset (3 VALU per access); the
		// ring's start rides in the instruction's scalar offset.
		template <int GP, int R>
		__device__ __forceinline__ int RingWrap(unsigned laneRing, int base)
		{
			unsigned a = laneRing + (unsigned)base * (unsigned)(GP * 16);
			return (int)__builtin_elementwise_min(a, a - (unsigned)(R * GP * 16));
		}

		// Ring history of ring RG (a layer's input ring, or the conv head's) for the frames `shift` before set i's (frames before the block
		// start).  One load instruction whatever the class -- every wave issues the same number of VMEM operations per stage, so the vmcnt
		// waits can be counted -- with an out-of-range offset where this wave (TAP_LDS) or this lane (TAP_BOTH, frames inside the block)
		// needs nothing: such a load returns zeros.  (The scalar offset of a buffer instruction is not part of its range check: an
		// out-of-range vector offset drops the access, and so does the zero-sized resource of a shadow wave, whatever the scalar offset.)
		template <class C, int RG, int WR>
		__device__ __forceinline__ u32x4 HistLoadAt(const Ctx& cx, unsigned laneRing, int fl, int shift, int i)
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::RingG(RG), P = Geo<GP, C::T>::P, R = TB::RingFrames(RG), OFF = TB::RingOff(RG);
			const int cls = WaveTapClass<C>(WR, P, i, shift, SmallRing<C, RG>());
			if ((NA_ABL & 4) || cls == TAP_LDS) return RingLoad(cx.srsrc, OOB);
			const int pos0 = __builtin_amdgcn_readlane(cx.myPos, RG);
			int base = pos0 - shift + C::FW * cx.wave + 16 * P * i; // wave-uniform; in (-R, 2R)
			if (base < 0) base += R;
			if (base >= R) base -= R;
			const int addr = RingWrap<GP, R>(laneRing, base);
			constexpr int NT_LD = C::NT ? 2 : NA_SPK_NT_LD, NT_DIL = C::NT ? C::A::NT_DIL : NA_SPK_NT_DIL;
			constexpr bool LONG = RG < TB::NL && TB::Dil(RG < TB::NL ? RG : 0) >= NT_DIL;
			if constexpr (LONG && NT_LD != 0)
			{
				if (cls == TAP_HIST) return RingLoadAux<NT_LD>(cx.srsrc, addr, OFF * 16);
				return RingLoadAux<NT_LD>(cx.srsrc, (C::FW * cx.wave + 16 * P * i + fl < shift) ? addr : OOB, OFF * 16);
			}
			if (cls == TAP_HIST) return RingLoad(cx.srsrc, addr, OFF * 16);
			return RingLoad(cx.srsrc, (C::FW * cx.wave + 16 * P * i + fl < shift) ? addr : OOB, OFF * 16);
		}

		// shift of layer L's tap k
		template <class C, int L>
		constexpr int ShiftOf(int k) { return C::TB::Dil(L) * (C::TB::KS(L) - 1 - k); }

		// does any wave of the block need the ring history of a tap `shift` back for set i?  (wave 0 has the earliest frames)
		template <class C, int GP>
		constexpr bool HistNeededAt(int shift, int i, bool small) { return !small && TapClassOf(0, Geo<GP, C::T>::P, i, shift) != TAP_LDS; }

		// prefetched taps of layer L: k < min(K - 1, HPF)
		template <class C, int L>
		constexpr int PrefetchTaps() { return (C::TB::KS(L) - 1) < C::HPF ? (C::TB::KS(L) - 1) : C::HPF; }
		template <class C, int L>
		constexpr int HistLoadsOf()
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::GPof(TB::ArrOf(L)), S = Geo<GP, C::T>::S;
			int n = 0;
			for (int k = 0; k < PrefetchTaps<C, L>(); k++)
				for (int i = 0; i < S; i++) n += HistNeededAt<C, GP>(ShiftOf<C, L>(k), i, SmallRing<C, L>()) ? 1 : 0;
			return n;
		}

		template <class C, int L, int WR>
		__device__ __forceinline__ void HistPrefetch(const Ctx& cx, unsigned laneRing, int fl, State& st)
		{
			constexpr int GP = C::TB::GPof(C::TB::ArrOf(L)), S = Geo<GP, C::T>::S;
#pragma unroll
			for (int k = 0; k < PrefetchTaps<C, L>(); k++)
#pragma unroll
				for (int i = 0; i < S; i++)
					if (HistNeededAt<C, GP>(ShiftOf<C, L>(k), i, SmallRing<C, L>())) st.hist[k][i] = HistLoadAt<C, L, WR>(cx, laneRing, fl, ShiftOf<C, L>(k), i);
		}

		// A stage's output -> the LDS image (in-block taps of the reader, if it has any) and ring RG (history for LATER blocks: only the
		// last R - 128 frames of a block are ever read back).  Ring stores that no wave of the block needs are not issued at all; the
		// others are one instruction on every wave (out-of-range offset where nothing is kept).  MINSHIFT: the reader's smallest tap shift.
		template <class C, int RG>
		constexpr bool StoreNeeded(int i)
		{
			typedef typename C::TB TB;
			constexpr int P = Geo<TB::RingG(RG), C::T>::P, KEEP = TB::RingKeep(RG);
			return (C::NF - C::FW) + 16 * P * (i + 1) - 1 >= C::NF - KEEP; // the last wave's last frame of set i
		}
		template <class C, int RG>
		constexpr int StoresOf()
		{
			constexpr int S = Geo<C::TB::RingG(RG), C::T>::S;
			int n = 0;
			for (int i = 0; i < S; i++) n += StoreNeeded<C, RG>(i) ? 1 : 0;
			return n;
		}

		template <class C, int RG, int GP, int MINSHIFT>
		__device__ __forceinline__ void Publish(const Ctx& cx, const Lanes<C, GP>& ln, u32x4 v, int i, int imgWrite)
		{
			typedef typename C::TB TB;
			constexpr int P = Geo<GP, C::T>::P, R = TB::RingFrames(RG), OFF = TB::RingOff(RG), KEEP = TB::RingKeep(RG);
			static_assert(GP == TB::RingG(RG), "lane mode of the receiving ring");
			if (!(NA_ABL & 64) && MINSHIFT < C::NF) // the reader takes in-block frames of other lanes
				LdsWrite16(ln.img + (unsigned)(imgWrite * C::IMG_ONE + 16 * P * i * 16), v);
			if ((NA_ABL & 4) || !StoreNeeded<C, RG>(i)) return;
			const int pos0 = __builtin_amdgcn_readlane(cx.myPos, RG);
			int base;
			if constexpr (TB::CompactRing(RG))
			{
				// a ring shorter than the block: the kept frames (the last KEEP of the block) are counted back from the cursor AFTER the
				// block, (pos0 + NF) mod R; frame f sits NF - f <= KEEP frames behind it.  (Only the sets that keep something use `base`.)
				int end = pos0 + C::NF % R;
				if (end >= R) end -= R;
				base = end - C::NF + C::FW * cx.wave + 16 * P * i + R; // position of the set's first frame + R: in [0, 2R) for a keeping set
			}
			else
			{
				base = pos0 + C::FW * cx.wave + 16 * P * i; // < 2R
				if (base >= R) base -= R;
			}
			const int addr = RingWrap<GP, R>(ln.ring, base);
			constexpr int NT_ST = C::NT ? 2 : NA_SPK_NT_ST, NT_DIL = C::NT ? C::A::NT_DIL : NA_SPK_NT_DIL;
			constexpr bool LONG = RG < TB::NL && TB::Dil(RG < TB::NL ? RG : 0) >= NT_DIL;
			if constexpr (LONG && NT_ST != 0)
			{
				if (KEEP >= C::NF) RingStoreAux<NT_ST>(cx.srsrc, v, addr, OFF * 16);
				else RingStoreAux<NT_ST>(cx.srsrc, v, (C::FW * cx.wave + 16 * P * i + ln.fl >= C::NF - KEEP) ? addr : OOB, OFF * 16);
				return;
			}
			if (KEEP >= C::NF) RingStore(cx.srsrc, v, addr, OFF * 16);
			else RingStore(cx.srsrc, v, (C::FW * cx.wave + 16 * P * i + ln.fl >= C::NF - KEEP) ? addr : OOB, OFF * 16);
		}

		// The 16 frames in front of the block from ring RG (the last 16 the previous block stored) -> guard quads of the reader's LDS image,
		// by LDS-DMA (no registers: a load into VGPRs was sunk next to its use by the scheduler and its latency paid in every layer), issued
		// at the START of the stage in front of the reader -- that image is not read during this stage -- and awaited with the stage's weight
		// DMA before the closing barrier.  Wave 0 of the stream moves them: lanes 0..15 = frames, one instruction per channel group plane.
		// A reader that is not "small" needs the quad in front of frame 0 to be ZERO instead (straddling taps clamp to it).
		template <class C, int RG, int GP>
		__device__ __forceinline__ void GuardStage(const Ctx& cx, const Lanes<C, GP>& ln, int imgBuf)
		{
			typedef typename C::TB TB;
			static_assert(GP == TB::RingG(RG), "lane mode of the ring");
			constexpr int R = TB::RingFrames(RG), OFF = TB::RingOff(RG);
			if constexpr (!C::A::GUARDHIST) return; // (the prologue's zero guard quads stay zero)
			if (cx.wave != 0) return;
			if constexpr (SmallRing<C, RG>())
			{
				if (NA_ABL & 4) return;
				const int pos0 = __builtin_amdgcn_readlane(cx.myPos, RG);
				int base = pos0 - GUARD;
				if (base < 0) base += R;
				if (cx.lane < 16)
				{
#pragma unroll
					for (int cg = 0; cg < GP; cg++)
					{
						const int addr = RingWrap<GP, R>((unsigned)((cx.lane * GP + cg) * 16), base);
						const unsigned dst = (unsigned)(C::IMG_OFF + imgBuf * C::IMG_ONE + cg * PLANE * 16) + (unsigned)(cx.sub * 2 * C::IMG_ONE);
						__builtin_amdgcn_raw_ptr_buffer_load_lds(cx.srsrc, (__attribute__((address_space(3))) void*)(LdsPtr)(size_t)dst, 16, addr, OFF * 16, 0, 0);
					}
				}
			}
			else
			{
				if ((cx.lane >> 4) < GP && (cx.lane & 15) == 15) LdsWrite16(ln.img + (unsigned)(imgBuf * C::IMG_ONE) - 16u * 16u, u32x4{ 0, 0, 0, 0 }); // frame 15 - 16 = -1
			}
		}

		// The next chunk of A operands (chunk CN of stage SN) -> the other LDS weight buffer by LDS-DMA (lane l's 16 bytes land at base +
		// 16 l; no VGPRs, no ds_write), issued at the start of the chunk before it, awaited just before that one's closing barrier.
		// Operand count and offsets are constants: exactly ceil(ops * 64 / NTHREADS) loads per thread.
		template <class C, int SN, int CN>
		struct Stager
		{
			typedef typename C::TB TB;
			static constexpr int QUADS = (TB::ChunkEnd(SN, CN) - TB::ChunkBegin(SN, CN)) * 64;
			static constexpr int SRC = TB::AOff(SN) + TB::ChunkBegin(SN, CN) * 64;
			static constexpr int BUF = TB::ChunkIndex(SN, CN) & 1;
			static constexpr int NCOPY = (QUADS + C::STG_THREADS - 1) / C::STG_THREADS;
			static __device__ __forceinline__ void Begin(const Ctx& cx)
			{
				if (NA_ABL & 16) return;
#pragma unroll
				for (int c = 0; c < NCOPY; c++)
				{
					// operands are 64 quads: a wave's 1 KB slice is one whole operand or lies beyond the block.  An out-of-range LDS-DMA load still
					// WRITES (zeros), so a wave with nothing to stage aims at the dump slot -- same instruction count on every wave.
					const int i0 = c * C::STG_THREADS + cx.stgWave * 64; // first quad of this wave's slice (wave-uniform)
					const bool mine = i0 < QUADS;
					const unsigned dst = mine ? cx.wbuf + (unsigned)(BUF * C::WBUF_ONE) + (unsigned)i0 * 16u : (unsigned)C::DUMP_OFF;
					__builtin_amdgcn_raw_ptr_buffer_load_lds(cx.wrsrc, (__attribute__((address_space(3))) void*)(LdsPtr)(size_t)dst, 16,
						mine ? (SRC + i0 + cx.lane) * 16 : OOB, 0, 0, 0);
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
		// the chunk after (s, c)
		template <class C, int s, int c>
		struct NextChunk
		{
			static constexpr bool SAME = c + 1 < C::TB::NumChunks(s);
			static constexpr int S = SAME ? s : s + 1, CN = SAME ? c + 1 : 0;
			typedef Stager<C, S, CN> St;
		};

		// operand m of stage s (must lie in the chunk that is in LDS: chunk c)
		template <class C>
		__device__ __forceinline__ u32x4 WOp(const Ctx& cx, int s, int c, int m)
		{
			const int buf = C::TB::ChunkIndex(s, c) & 1, local = m - C::TB::ChunkBegin(s, c);
			return LdsRead16(cx.wbuf + (unsigned)(buf * C::WBUF_ONE + ((NA_ABL & 128) ? 0 : local) * 1024) + (unsigned)cx.lane * 16u);
		}

		// PACKED: two values per v_pk_* instruction (half the instructions -- what a launch of few waves wants, every instruction costs a
		// lone wave its ~5 cycles); unpacked: plain VALU instructions, which issue beside the MFMAs of the SIMD's other waves where packed
		// f32 math does not -- what the full-size workgroups want at four waves per SIMD (A1 Standard 41.8 -> 40.9 us,
		// Lite 35.0 -> 34.5; Nano x 1024 on the half-size workgroups 23.4 -> 23.5 the other way).  Same expression tree, same results.
		template <bool PACKED>
		__device__ __forceinline__ f32x4 ActivateTanh(f32x4 a)
		{
			if (PACKED)
			{
				const f32x2 lo = FastTanh2(f32x2{ a.x, a.y }), hi = FastTanh2(f32x2{ a.z, a.w });
				return f32x4{ lo.x, lo.y, hi.x, hi.y };
			}
			return f32x4{ FastTanh(a.x), FastTanh(a.y), FastTanh(a.z), FastTanh(a.w) };
		}

		// One shifted conv tap of ring RG's reader for set i: the frames `shift` back, as classified for wave WR -- ring history (registers),
		// LDS image, or both (the conv is linear in the operand: the ring part is zero for in-block lanes and vice versa).
		template <class C, int RG, int GP, int WR>
		__device__ __forceinline__ f32x4 ConvTap(const Ctx& cx, const Lanes<C, GP>& ln, int imgRead, int shift, int i, u32x4 ah, u32x4 al, u32x4 hist, f32x4 acc)
		{
			constexpr int P = Geo<GP, C::T>::P;
			const int cls = WaveTapClass<C>(WR, P, i, shift, SmallRing<C, RG>());
			if (cls != TAP_LDS)
			{
				acc = Mfma(ah, hist, acc);
				acc = MfmaLo<1>(al, hist, acc);
			}
			if (cls != TAP_HIST)
			{
				u32x4 b;
				if (NA_ABL & 256) b = u32x4{ (unsigned)shift, 0, 0, 0 };
				else if (cls == TAP_LDS) b = LdsRead16(ln.img + (unsigned)(imgRead * C::IMG_ONE) + (unsigned)((16 * P * i - shift) * 16));
				else
				{
					// straddling: lanes whose frame lies before the block read the zero guard quad in front of frame 0
					int off = C::FW * cx.wave + 16 * P * i + ln.fl - shift;
					off = off < -1 ? -1 : off;
					b = LdsRead16((unsigned)(C::IMG_OFF + imgRead * C::IMG_ONE + GUARD * 16) + (unsigned)(cx.sub * 2 * C::IMG_ONE) + (unsigned)((ln.cg * PLANE + off) * 16));
				}
				acc = Mfma(ah, b, acc);
				acc = MfmaLo<1>(al, b, acc);
			}
			return acc;
		}

		// ---- one layer (WaveNetLayerT::Process, WaveNet.h:462-494) for the waves whose tap classes are those of wave WR ----------------
		// Operands of the stage: taps 0 .. K-1 as (hi, lo) pairs (tap K-1 is the unshifted one), aux = (mix-in, conv bias), 1x1 (hi, lo), its
		// bias.  They are in LDS chunk by chunk (Tab::ChunkBegin / ChunkEnd): chunk c + 1 (or the next stage's first) is staged while
		// chunk c is used, and every chunk is closed by a barrier.  A K <= 6 layer is ONE chunk.
		template <class C, int L, int WR, int c>
		__device__ __forceinline__ void LayerChunk(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::ArrOf(L))>& ln, State& st, u32x4 (&ax)[LayerSig<C, L>::S],
			f32x4 (&acc)[LayerSig<C, L>::S])
		{
			typedef typename C::TB TB;
			typedef LayerSig<C, L> SG;
			constexpr int GP = SG::GP, S = SG::S, K = SG::K, s = TB::StageOfLayer(L), NCH = TB::NumChunks(s), CB = TB::ChunkBegin(s, c), CE = TB::ChunkEnd(s, c);
			constexpr int imgRead = s & 1, imgWrite = (s + 1) & 1, LN = SG::LN;
			constexpr bool LASTLAYER = TB::LastOfArr(L) && TB::ArrOf(L) == TB::NA - 1;
			constexpr int NEXTMINSHIFT = SG::NEXT ? TB::Dil(LN) : (1 << 20); // the reader's smallest tap shift: is the LDS image needed?
			constexpr bool OWN = 2 * (K - 1) >= CB && 2 * (K - 1) < CE;  // the unshifted tap's operands are in this chunk (every shifted tap is consumed by then)
			constexpr bool TAIL = 2 * K >= CB && 2 * K < CE;             // aux / 1x1 operands are in this chunk
			constexpr int LATER = ((OWN && SG::NEXT) ? HistLoadsOf<C, LN>() : 0) + ((TAIL && SG::NEXT && !LASTLAYER) ? StoresOf<C, LN>() : 0);
			typedef typename NextChunk<C, s, c>::St NextStager;
			if constexpr (c == 0)
			{
				// what the next reader (the next layer, or the conv head after the last layer) finds in front of frame 0 of its image
				if constexpr (SG::NEXT) GuardStage<C, LN, GP>(cx, ln, imgWrite);
				else if constexpr (LASTLAYER && TB::HEADK > 1) GuardStage<C, TB::NL, GP>(cx, ln, (TB::NSTAGES - 1 + 1) & 1);
			}
			NextStager::Begin(cx);
			if constexpr (c == 0)
			{
#pragma unroll
				for (int i = 0; i < S; i++) ax[i] = AuxRead<C, GP>(ln, i);
			}
			// dilated conv (WaveNet.h:139-290): tap k reads the frame d (K-1-k) back; bias and mix-in arrive through the aux operand.
			// History of taps beyond the prefetched ones (K = 15 layers): all loads of the chunk first, then the MFMAs
			constexpr int KLO = CB / 2, KHI = (CE / 2 < K - 1) ? CE / 2 : K - 1; // shifted taps of this chunk: [KLO, KHI)
			constexpr int NX = (KHI > C::HPF) ? KHI - (KLO > C::HPF ? KLO : C::HPF) : 0;
			u32x4 hx[NX > 0 ? NX : 1][S];
#pragma unroll
			for (int k = KLO; k < KHI; k++)
			{
				if (k < C::HPF) continue;
#pragma unroll
				for (int i = 0; i < S; i++)
					if (WaveTapClass<C>(WR, SG::P, i, ShiftOf<C, L>(k), SmallRing<C, L>()) != TAP_LDS) hx[k - (KLO > C::HPF ? KLO : C::HPF)][i] = HistLoadAt<C, L, WR>(cx, ln.ring, ln.fl, ShiftOf<C, L>(k), i);
			}
#pragma unroll
			for (int k = KLO; k < KHI; k++)
			{
				const u32x4 ah = WOp<C>(cx, s, c, 2 * k), al = WOp<C>(cx, s, c, 2 * k + 1);
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					u32x4 h = u32x4{ 0, 0, 0, 0 };
					if (WaveTapClass<C>(WR, SG::P, i, ShiftOf<C, L>(k), SmallRing<C, L>()) != TAP_LDS)
						h = (k < C::HPF) ? st.hist[k < C::HPF ? k : 0][i] : hx[k >= C::HPF ? k - (KLO > C::HPF ? KLO : C::HPF) : 0][i];
					acc[i] = ConvTap<C, L, GP, WR>(cx, ln, imgRead, ShiftOf<C, L>(k), i, ah, al, h, acc[i]);
				}
			}
			if constexpr (OWN)
			{
				// history of the NEXT layer's prefetched taps (the registers are free again)
				if constexpr (SG::NEXT) HistPrefetch<C, LN, WR>(cx, ln.ring, ln.fl, st);
				// unshifted tap = the layer input itself (registers)
				const u32x4 ah = WOp<C>(cx, s, c, 2 * K - 2), al = WOp<C>(cx, s, c, 2 * K - 1);
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					acc[i] = Mfma(ah, st.xs[i], acc[i]);
					acc[i] = MfmaLo<1>(al, st.xs[i], acc[i]);
				}
			}
			if constexpr (TAIL)
			{
				// aux operand: (mix-in, conv bias) * (cond, 1)   (:288-289, :471); activation (:473-480); head accumulate (:482) on the
				// matrix pipe: head += I (zh + zl); 1x1 + bias + residual (:486-491)
				const u32x4 xa = WOp<C>(cx, s, c, 2 * K);
#pragma unroll
				for (int i = 0; i < S; i++) acc[i] = Mfma(xa, ax[i], acc[i]);
				SPK_STAMP(s, 1);
				f32x4 z[S];
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					if constexpr (C::A::LEAKY) z[i] = f32x4{ LeakyReLU(acc[i].x), LeakyReLU(acc[i].y), LeakyReLU(acc[i].z), LeakyReLU(acc[i].w) };
					else z[i] = ActivateTanh<(NA_PK_TANH != 0) && (C::NTHREADS < 512)>(acc[i]);
				}
				SPK_STAMP(s, 2);
				const u32x4 idop = LdsRead16((unsigned)C::IDOP_OFF + (unsigned)cx.lane * 16u);
				const u32x4 w1h = WOp<C>(cx, s, c, 2 * K + 1), w1l = WOp<C>(cx, s, c, 2 * K + 2), b1a = WOp<C>(cx, s, c, 2 * K + 3);
#pragma unroll
				for (int i = 0; i < S; i++)
				{
					const u32x4 zs = SplitQuad(z[i]);
					st.hd[i] = Mfma(idop, zs, st.hd[i]);
					if constexpr (!LASTLAYER) // NeedOutput (WaveNet.h:643,785): the very last layer's 1x1 is dead
					{
						f32x4 y = st.xc[i];
						y = Mfma(w1h, zs, y);
						y = MfmaLo<2>(w1l, zs, y);
						y = Mfma(b1a, NA_SPK_AUX2 ? AuxRead<C, GP>(ln, i) : ax[i], y);
						st.xc[i] = y;
						if constexpr (SG::NEXT)
						{
							st.xs[i] = Split<C>(y, cx);
							Publish<C, LN, GP, NEXTMINSHIFT>(cx, ln, st.xs[i], i, imgWrite);
						}
					}
				}
				SPK_STAMP(s, 3);
			}
			// the DMA data must be in LDS before the closing barrier lets other waves read it
			NextStager::template End<LATER>();
			SPK_STAMP(s, 4);
			BlockBarrier<C::NTHREADS / 64>();
			if constexpr (c + 1 < NCH) LayerChunk<C, L, WR, c + 1>(cx, ln, st, ax, acc);
		}

		template <class C, int L, int WR>
		__device__ __forceinline__ void LayerBody(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::ArrOf(L))>& ln, State& st)
		{
			constexpr int S = LayerSig<C, L>::S;
			SPK_STAMP(C::TB::StageOfLayer(L), 0);
			u32x4 ax[S];
			f32x4 acc[S];
#pragma unroll
			for (int i = 0; i < S; i++) acc[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
			LayerChunk<C, L, WR, 0>(cx, ln, st, ax, acc);
			SPK_STAMP(C::TB::StageOfLayer(L), 5);
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
					constexpr unsigned MASK = SG::MaskOf(W); // (a constant expression: evaluated by the compiler, not by the wave)
					if ((MASK >> cx.wave) & 1u) LayerBody<C, L, W>(cx, ln, st);
					else LayerDispatch<C, L, W + 1>(cx, ln, st);
				}
			}
		}

		// history prefetch of the first layer of an array (issued by the rechannel / link stage in front of it): per wave class
		template <class C, int L, int W>
		__device__ __forceinline__ void FirstHistDispatch(const Ctx& cx, unsigned laneRing, int fl, State& st)
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::GPof(TB::ArrOf(L)), P = Geo<GP, C::T>::P, S = Geo<GP, C::T>::S;
			if constexpr (W < C::WPS)
			{
				constexpr bool same = [] {
					bool r = true;
					for (int w = W + 1; w < C::WPS; w++)
						for (int k = 0; k < PrefetchTaps<C, L>(); k++)
							for (int i = 0; i < S; i++) r = r && WaveTapClass<C>(w, P, i, ShiftOf<C, L>(k), SmallRing<C, L>()) == WaveTapClass<C>(W, P, i, ShiftOf<C, L>(k), SmallRing<C, L>());
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
			constexpr int GP = C::TB::GPof(0), S = Geo<GP, C::T>::S;
			SPK_STAMP(0, 0);
			GuardStage<C, 0, GP>(cx, ln, 1);
			// (stage 1's operands were set out in the prologue, RunWorkgroup; this stage awaits them at its end)
			const u32x4 ra = WOp<C>(cx, 0, 0, 0);
#pragma unroll
			for (int i = 0; i < S; i++)
			{
				const u32x4 ax = AuxRead<C, GP>(ln, i);
				f32x4 x = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				x = Mfma(ra, ax, x);
				st.xc[i] = x;
				st.hd[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; // WaveNet.h:772 headArray.SetZero()
				st.xs[i] = Split<C>(x, cx);
				Publish<C, 0, GP, C::TB::Dil(0)>(cx, ln, st.xs[i], i, 1);
			}
			FirstHistDispatch<C, 0, 0>(cx, ln.ring, ln.fl, st);
			SPK_STAMP(0, 1); SPK_STAMP(0, 2); SPK_STAMP(0, 3);
			Stager<C, 1, 0>::template End<StoresOf<C, 0>() + HistLoadsOf<C, 0>()>();
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
			constexpr int So = Geo<GPO, C::T>::S, Sn = Geo<GPN, C::T>::S, s = TB::LinkStage(AN), LN = TB::FirstLayerOfArr(AN);
			static_assert(C::T == 1 || C::T == 2, "array links are written for one or two tiles per wave");
			SPK_STAMP(s, 0);
			GuardStage<C, LN, GPN>(cx, ln, (s + 1) & 1);
			Stager<C, s + 1, 0>::Begin(cx);
			u32x4 hs[So], xq[So];
#pragma unroll
			for (int i = 0; i < So; i++)
			{
				hs[i] = Split<C>(st.hd[i], cx);
				xq[i] = Split<C>(st.xc[i], cx);
			}
			f32x4 hn[Sn], xn[Sn];
#pragma unroll
			for (int i = 0; i < Sn; i++)
			{
				hn[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				xn[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				// head bias of the previous array (a zero operand when it has none): any stream's aux operand carries the ones it multiplies
				hn[i] = Mfma(WOp<C>(cx, s, 0, 4 * NC), AuxRead<C, GPN>(ln, i), hn[i]);
			}
#pragma unroll
			for (int t = 0; t < C::T; t++)
			{
				const int u = t % NC, so = t / Po, sn = t / Pn;
hn[sn] = Mfma(WOp<C>(cx, s, 0, 4 * u), hs[so], hn[sn]);
				hn[sn] = MfmaLo<4>(WOp<C>(cx, s, 0, 4 * u + 1), hs[so], hn[sn]);
				xn[sn] = Mfma(WOp<C>(cx, s, 0, 4 * u + 2), xq[so], xn[sn]);
				xn[sn] = MfmaLo<4>(WOp<C>(cx, s, 0, 4 * u + 3), xq[so], xn[sn]);
			}
#pragma unroll
			for (int i = 0; i < Sn; i++)
			{
				st.hd[i] = hn[i];
				st.xc[i] = xn[i];
				st.xs[i] = Split<C>(xn[i], cx);
				Publish<C, LN, GPN, TB::Dil(LN)>(cx, ln, st.xs[i], i, (s + 1) & 1);
			}
			FirstHistDispatch<C, LN, 0>(cx, ln.ring, ln.fl, st);
			SPK_STAMP(s, 1); SPK_STAMP(s, 2); SPK_STAMP(s, 3);
			Stager<C, s + 1, 0>::template End<StoresOf<C, LN>() + HistLoadsOf<C, LN>()>();
			SPK_STAMP(s, 4);
			BlockBarrier<C::NTHREADS / 64>();
			SPK_STAMP(s, 5);
		}

		// last array's head: out = scale * (conv_K(head) + b)[0]  (WaveNet.h:658-660, :793-798); K = 1 (A1) straight from the registers,
		// K > 1 (A2: 16) through the LDS image / the head ring like a layer conv, the operands chunk by chunk.  One output row per tile
		// slot (PK: per stream).
		template <class C, int WR>
		__device__ __forceinline__ void HeadBody(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::NA - 1)>& ln, State& st, float* __restrict__ out, size_t outBase,
			const long (&outRow)[4], int pack, float headScale, bool live)
		{
			typedef typename C::TB TB;
			constexpr int GP = TB::GPof(TB::NA - 1), P = Geo<GP, C::T>::P, S = Geo<GP, C::T>::S, s = TB::NSTAGES - 1, K = TB::HEADK, NCH = TB::NumChunks(s), RG = TB::NL;
			constexpr int imgHead = (s + 1) & 1;
			u32x4 hs[S];
			f32x4 acc[S];
#pragma unroll
			for (int i = 0; i < S; i++)
			{
				hs[i] = Split<C>(st.hd[i], cx);
				acc[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
			}
#pragma unroll
			for (int c = 0; c < NCH; c++)
			{
				if constexpr (K > 1)
				{
					if (c + 1 < NCH)
					{
						if (c == 0) Stager<C, s, 1>::Begin(cx);
						else if (c + 1 < NCH) Stager<C, s, (NCH > 2 ? 2 : 1)>::Begin(cx);
					}
					if (c == 0)
					{
						// the head accumulator of this block -> LDS image + head ring (and the 16 frames before the block into the image's guard),
						// then every wave may read its neighbours' frames
#pragma unroll
						for (int i = 0; i < S; i++) Publish<C, RG, GP, 1>(cx, ln, hs[i], i, imgHead);
						BlockBarrier<C::NTHREADS / 64>();
					}
#pragma unroll
					for (int k = 0; k < K - 1; k++)
					{
						if (2 * k < TB::ChunkBegin(s, c) || 2 * k >= TB::ChunkEnd(s, c)) continue;
						const u32x4 ah = WOp<C>(cx, s, c, 2 * k), al = WOp<C>(cx, s, c, 2 * k + 1);
#pragma unroll
						for (int i = 0; i < S; i++)
						{
							u32x4 h = u32x4{ 0, 0, 0, 0 };
							if (WaveTapClass<C>(WR, P, i, K - 1 - k, SmallRing<C, RG>()) != TAP_LDS) h = HistLoadAt<C, RG, WR>(cx, ln.ring, ln.fl, K - 1 - k, i);
							acc[i] = ConvTap<C, RG, GP, WR>(cx, ln, imgHead, K - 1 - k, i, ah, al, h, acc[i]);
						}
					}
				}
				if (2 * (K - 1) >= TB::ChunkBegin(s, c) && 2 * (K - 1) < TB::ChunkEnd(s, c))
				{
					const u32x4 ah = WOp<C>(cx, s, c, 2 * K - 2), al = WOp<C>(cx, s, c, 2 * K - 1);
#pragma unroll
					for (int i = 0; i < S; i++)
					{
						acc[i] = Mfma(ah, hs[i], acc[i]);
						acc[i] = MfmaLo<8>(al, hs[i], acc[i]);
					}
				}
				if (2 * K >= TB::ChunkBegin(s, c) && 2 * K < TB::ChunkEnd(s, c))
				{
					// bias (a zero operand when the head has none), scale, output row(s)
					const u32x4 ba = WOp<C>(cx, s, c, 2 * K);
#pragma unroll
					for (int i = 0; i < S; i++)
					{
						acc[i] = Mfma(ba, AuxRead<C, GP>(ln, i), acc[i]);
						const int f = C::FW * cx.wave + 16 * P * i + ln.fl;
						if (live && ln.cg == 0)
						{
							if constexpr (C::PK)
							{
								const float v[4] = { acc[i].x, acc[i].y, acc[i].z, acc[i].w };
#pragma unroll
								for (int q = 0; q < 4; q++)
									if (q < pack && outRow[q] >= 0) StoreOut<C::COH>(out + outRow[q] + f, headScale * v[q]);
							}
							else StoreOut<C::COH>(out + outBase + f, headScale * acc[i].x);
						}
					}
				}
				if (c + 1 < NCH)
				{
					__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8)); // the next chunk has landed (vmcnt 0: nothing else is worth keeping in flight here)
					BlockBarrier<C::NTHREADS / 64>();
				}
			}
		}

		template <class C>
		__device__ __forceinline__ void HeadDispatch(const Ctx& cx, const Lanes<C, C::TB::GPof(C::TB::NA - 1)>& ln, State& st, float* __restrict__ out, size_t outBase,
			const long (&outRow)[4], int pack, float headScale, bool live)
		{
			// a dense head is the same on every wave; a conv head's taps (shifts 1 .. K - 1 < FW) straddle the block start on wave 0 and lie
			// inside the block on every other wave
			if constexpr (C::TB::HEADK == 1 || C::WPS == 1 || SmallRing<C, C::TB::NL>()) HeadBody<C, 0>(cx, ln, st, out, outBase, outRow, pack, headScale, live);
			else
			{
				if (cx.wave == 0) HeadBody<C, 0>(cx, ln, st, out, outBase, outRow, pack, headScale, live);
				else HeadBody<C, C::WPS - 1>(cx, ln, st, out, outBase, outRow, pack, headScale, live);
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
			else HeadDispatch<C>(cx, ln, st, out, outBase, outRow, pack, headScale, live);
		}

		// One workgroup of architecture C: prologue (aux operands of the block, zero guards, identity operand, stage 0's operand), the
		// layer chain, cursor update.
		template <class C>
		__device__ __forceinline__ void RunWorkgroup(const GroupArgs& ga, int groupBlock, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride
#ifdef NA_SP_TRACE
			, long long* trace
#endif
			, const int tid = (int)threadIdx.x // (the resident launch hands in an opaque copy: nothing derived from it may be hoisted out of its loop)
			)
		{
			typedef typename C::TB TB;
			constexpr bool PK = C::PK;
			constexpr int SPB = C::SPB, NF = C::NF;
			const int lane = tid & 63;
			const int waveAll = __builtin_amdgcn_readfirstlane(tid >> 6);
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
			cx.wbuf = (unsigned)(C::WBUF_OFF + (C::SKEW > 0 ? sub : 0) * 2 * C::WBUF_ONE);
			cx.stgWave = C::SKEW > 0 ? wave : waveAll;
#ifdef NA_SP_TRACE
			cx.trace = trace;
			cx.nwaves = C::NTHREADS / 64;
			if (cx.trace != nullptr && lane == 0) cx.trace[((C::TB::NSTAGES * 8 + 0) * cx.nwaves) + waveAll] = (long long)__builtin_readcyclecounter();
#endif
			// stage 1's operands set out NOW (LDS-DMA into the second weight buffer; stage 0 awaits them at its end, RechStage) rather than at
			// the start of stage 0, whose own work is one MFMA (measured: 0.1 us per Standard step, noise level; kept)
			Stager<C, 1, 0>::Begin(cx);
			// packed: channel groups per real stream = (channels / pack) / 4 -> shift (1, 2, 4 -> 0, 1, 2)
			const int pack = PK ? ga.pack : 1;
			cx.gs0 = PK ? ((ga.gps0 >> 1) & 3) : 0;
			cx.gs1 = PK ? (ga.gps1 == 0 ? -1 : ((ga.gps1 >> 1) & 3)) : 0; // (0 channel groups per stream: a dense pack, two streams per group)
			long outRow[4];
			int rowOf[4]; // PK: the rows of the (up to four) real streams of this virtual stream, -1: none
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				const int r = (PK && live && q < ga.pack) ? ga.rows[sidx * ga.pack + q] : -1;
				rowOf[q] = r;
				outRow[q] = r >= 0 ? (long)r * outStride : -1;
			}

			// input row (WaveNet.h:770 input -> condition) -> the aux operand of every frame
			if constexpr (PK)
			{
				// (the four rows' samples of a frame are loaded TOGETHER: one after the other -- row index, then sample, four times -- the
				// prologue of a packed workgroup was eight dependent memory round trips, 6 500 cycles before stage 0 of Nano x 1024)
				for (int i = wave * 64 + lane; i < FRAMES; i += C::WPS * 64)
				{
					float c[4];
#pragma unroll
					for (int q = 0; q < 4; q++) c[q] = (rowOf[q] >= 0 && i < NF) ? LoadIn<C::COH>(in + (size_t)rowOf[q] * inStride + i) : 0.0f;
#pragma unroll
					for (int q = 0; q < 4; q++)
					{
						const float cc = ClampCond(c[q], ga.condLimit);
						const _Float16 ch = (_Float16)cc, cl = (_Float16)(cc - (float)ch);
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
					// [cond_h, 1 | cond_l, 1 | cond_h, 0 | 0, 0] (see FillSplitAux in wavenet_plan.cpp; 8-byte entries: AuxRead rebuilds the second half)
					const float c = (i < NF) ? ClampCond(LoadIn<C::COH>(in + (size_t)row * inStride + i), ga.condLimit) : 0.0f;
					const _Float16 ch = (_Float16)c, cl = (_Float16)(c - (float)ch);
					const f16x2 a = { ch, (_Float16)1.0f }, b = { cl, (_Float16)1.0f }, d = { ch, (_Float16)0.0f };
					if constexpr (C::AUX16)
						LdsWrite16((unsigned)(C::AUX_OFF + (sub * FRAMES + i) * 16), u32x4{ __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, d), 0u });
					else
						*reinterpret_cast<__attribute__((address_space(3))) u32x2*>((LdsPtr)(size_t)(unsigned)(C::AUX_OFF + (sub * FRAMES + i) * 8)) =
							u32x2{ __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b) };
				}
			}
			// zero quad in front of frame 0 of every plane of both block images
			for (int i = tid; i < SPB * 2 * TB::MaxGP(); i += C::NTHREADS) LdsWrite16((unsigned)(C::IMG_OFF + (i * PLANE + GUARD - 1) * 16), u32x4{ 0, 0, 0, 0 });
			// identity A operand: row i x k-block q = i / 4: 1.0 against the h AND the l half of channel i % 4
			if (tid < 64)
			{
				const int i = lane & 15, q = lane >> 4;
				const unsigned one = 0x3c00u; // f16 1.0
				const unsigned lo = (q == (i >> 2)) ? (((i & 3) == 0) ? one : ((i & 3) == 1) ? (one << 16) : 0u) : 0u;
				const unsigned hi = (q == (i >> 2)) ? (((i & 3) == 2) ? one : ((i & 3) == 3) ? (one << 16) : 0u) : 0u;
				LdsWrite16((unsigned)C::IDOP_OFF + (unsigned)lane * 16u, u32x4{ lo, hi, lo, hi });
			}
			if constexpr (C::A::LEAKY)
			{
				if (tid < SPB) *reinterpret_cast<__attribute__((address_space(3))) unsigned*>((LdsPtr)(size_t)(unsigned)(C::FLAG_OFF + 16 * tid)) = 0u;
			}
			// stage 0's single operand (offset 0 of every weight image), into every pair of weight buffers
			if (cx.stgWave == 0) LdsWrite16(cx.wbuf + (unsigned)lane * 16u, BufLoad(cx.wrsrc, lane * 16));
			BlockBarrier<C::NTHREADS / 64>();

			// skewed streams: stream 1 starts SKEW stages (= barriers) after stream 0 and ends as many after it
			if constexpr (C::SKEW > 0)
			{
				if (sub != 0)
					for (int k = 0; k < C::SKEW; k++) BlockBarrier<C::NTHREADS / 64>();
			}
			State st;
			RunArrays<C, 0>(cx, st, out, (size_t)row * outStride, outRow, pack, ga.headScale, live);
			if constexpr (C::SKEW > 0)
			{
				if (sub == 0)
					for (int k = 0; k < C::SKEW; k++) BlockBarrier<C::NTHREADS / 64>();
			}

#ifdef NA_SP_TRACE
			if (cx.trace != nullptr && lane == 0) cx.trace[((C::TB::NSTAGES * 8 + 1) * cx.nwaves) + waveAll] = (long long)__builtin_readcyclecounter();
#endif
			if constexpr (C::A::LEAKY)
			{
				// every wave of the stream is through its last split: the closing wave reads the stream's flag word
				BlockBarrier<C::NTHREADS / 64>();
				if (wave == 0) CountRangeEvent(header, (int)*reinterpret_cast<__attribute__((address_space(3))) unsigned*>((LdsPtr)(size_t)(unsigned)(C::FLAG_OFF + 16 * sub)), lane, live);
			}
			// advance every ring cursor by NF (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
			if (wave == 0 && live && lane < ga.nrings)
			{
				const int R = ga.ringFrames[lane];
				int p = cx.myPos + NF;
				if (p >= R) p -= R;
				if (p >= R) p -= R;
				if (p >= R) p -= R; // (compact rings are shorter than a block: R >= 48)
				header[lane] = p;
			}
		}

		// grid = sum over the groups of ceil(active (virtual) streams / SPB of the group's architecture); workgroup = SPB streams x WPS
		// waves of T tiles; n == NF frames.  F = architecture family (the groups of one launch may be different members of it; all members
		// launch the same number of threads).  Dynamic LDS = the largest Cfg::LDS_BYTES, addressed absolutely from 0 (the kernel has no
		// static LDS), so every LDS offset of the chain is an instruction immediate.
		// waves per SIMD the kernel is compiled for = what two resident workgroups per CU put on a SIMD: 2 NF SPB threads per workgroup -> 4
		// for the full-size workgroups (128 VGPRs), 2 for the half-size ones and for 64-frame blocks (256 VGPRs), 1 below (see the launcher)
#ifndef NA_SPK_OCC
#define NA_SPK_OCC(nf, spb) (((nf) * (spb) / 64) < 1 ? 1 : ((nf) * (spb) / 64))
#endif
		// (one tile per wave: launched with at most one workgroup per CU, i.e. half the waves per SIMD of the same thread count)
		template <class F, int NF, int SPB>
		constexpr int OccOf() { return F::A0::T == 1 ? (NA_SPK_OCC(NF, SPB) / 2 < 1 ? 1 : NA_SPK_OCC(NF, SPB) / 2) : NA_SPK_OCC(NF, SPB); }
		template <class F, int NF, int SPB, bool PK, bool NT = false>
		__global__ void __launch_bounds__(64 * (NF / (16 * F::A0::T)) * (SPB * F::A0::T / 2)) __attribute__((amdgpu_waves_per_eu(OccOf<F, NF, SPB>()))) WaveNetSpecKernel(const LaunchArgs args,
			const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride
#ifdef NA_SP_TRACE
			, long long* __restrict__ trace, int traceBlock
#endif
			)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK, false, NT> C;
			typedef Cfg<typename F::A1, NF, SPB, PK, false, NT> C1;
			static_assert(C::NTHREADS == C1::NTHREADS, "members of a family launch the same workgroup");
			extern __shared__ __attribute__((aligned(16))) char dynSmem[];
			asm volatile("" : : "s"((unsigned)(size_t)(LdsPtr)dynSmem)); // the dynamic LDS segment is in use (and starts at 0)

			int gi = 0;
			for (int i = 1; i < args.numGroups; i++)
				if ((int)blockIdx.x >= args.g[i].firstBlock) gi = i;
			const GroupArgs& ga = args.g[gi];
			const int groupBlock = (int)blockIdx.x - ga.firstBlock;
#if NA_SPK_DELAY > 0
			// tuning builds: the second half of the grid (the workgroups that arrive SECOND on their CUs when the launch is one round of two
			// workgroups per CU) starts NA_SPK_DELAY x 64 cycles late, so that the two workgroups of a CU are out of phase: one in its
			// issue-bound stages while the other waits for memory in its d >= 64 stages
			if (blockIdx.x >= gridDim.x / 2)
				for (int k = 0; k < NA_SPK_DELAY; k += 100) __builtin_amdgcn_s_sleep(100);
#endif
#ifdef NA_SP_TRACE
			long long* tr = ((int)blockIdx.x == traceBlock) ? trace : nullptr;
			if (F::N > 1 && ga.arch == 1) RunWorkgroup<C1>(ga, groupBlock, in, out, inStride, outStride, tr);
			else RunWorkgroup<C>(ga, groupBlock, in, out, inStride, outStride, tr);
#else
			if (F::N > 1 && ga.arch == 1) RunWorkgroup<C1>(ga, groupBlock, in, out, inStride, outStride);
			else RunWorkgroup<C>(ga, groupBlock, in, out, inStride, outStride);
#endif
		}

		// ---- the resident launch --------------------------------------------------------------------------------------------------
		// The same chain, but the launch stays on the chip and walks consecutive buffers itself (wavenet_split_dev.h ResidentCtrl): per
		// command, workgroup b runs the stream blocks b, b + grid, ... of the launch list, then counts itself done.  Wave 0 polls the
		// command ring in pinned host memory; the other waves sleep at the barrier behind it.  Nothing here waits for another
		// workgroup, so the launch makes progress with any part of its grid resident.
		template <class F, int NF, int SPB, bool PK>
		__global__ void __launch_bounds__(64 * (NF / (16 * F::A0::T)) * (SPB * F::A0::T / 2)) __attribute__((amdgpu_waves_per_eu(OccOf<F, NF, SPB>()))) WaveNetSpecResidentKernel(const LaunchArgs args,
			const ResidentArgs ra
#ifdef NA_SP_TRACE
			, long long* __restrict__ trace, int traceBlock // (row NSTAGES of the stamps: 0 / 1 block entry / exit, 2 closing barrier passed, 3 command taken, 4 closing barrier of the block before)
#endif
			)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK, true> C;
			typedef Cfg<typename F::A1, NF, SPB, PK, true> C1;
			static_assert(C::NTHREADS == C1::NTHREADS && C::BCAST_OFF == C1::BCAST_OFF, "members of a family launch the same workgroup");
			extern __shared__ __attribute__((aligned(16))) char dynSmem[];
			asm volatile("" : : "s"((unsigned)(size_t)(LdsPtr)dynSmem));
			typedef __attribute__((address_space(3))) unsigned long long* Lds64;
			Lds64 const bc = reinterpret_cast<Lds64>((LdsPtr)(size_t)(unsigned)C::BCAST_OFF);
			// Nothing is carried in registers from one block to the next (the chain has none to spare: 128 VGPRs, no spills): the
			// workgroup's position -- commands done, block in progress -- lives in LDS next to the command, and thread 0 reads the launch
			// arguments afresh (`ra` behind an opaque copy of its address) every time round.
			//   bc[0] go | bc[1..4] in, out, inStride, outStride | bc[5] commands done by this workgroup | bc[6] block in progress (-1: none)
			if (threadIdx.x == 0)
			{
				bc[5] = ra.base + ra.wgDone[blockIdx.x];
				bc[6] = ~0ull;
				// the second workgroup of every CU starts late, once: next to a partner in another phase of the chain a workgroup runs faster
				// (its issue-bound stages beside the partner's memory-bound ones), and inside ONE launch the offset is paid once, not per buffer
				if (ra.startDelay > 0 && 2 * blockIdx.x >= gridDim.x)
				{
					const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
					while (__builtin_amdgcn_s_memrealtime() - t0 < ra.startDelay) __builtin_amdgcn_s_sleep(16);
				}
			}
			for (;;)
			{
				if (threadIdx.x == 0)
				{
					int z = 0; // an opaque zero: the arguments are read from the kernarg segment afresh, nothing of them stays in registers
					asm volatile("" : "+v"(z));
					const ResidentArgs& r = (&ra)[__builtin_amdgcn_readfirstlane(z)];
					unsigned long long k = bc[5];
					long blk = (long)bc[6];
					unsigned long long go = 1;
					if (blk >= 0) blk += (long)gridDim.x;
					if (blk < 0 || blk >= (long)r.numBlocks)
					{
						// (the last block of command k + 1 is done: every wave's stores have left it -- the closing wait + barrier below)
						const bool finished = blk >= 0;
						if (finished) k++;
						const unsigned slot = (unsigned)(k % RESIDENT_RING);
						if (finished) __hip_atomic_store(&r.wgDone[blockIdx.x], (unsigned)(k - r.base), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						// The first look at the next command and the done count of the last one travel together: one round trip to device
						// memory.  A command is six words of one line written by the host through the BAR; it is taken when its sequence
						// number and its check word agree with what was read (a torn or stale line does not pass and is read again).
						const ResidentCmd* cmd = &r.ctrl->cmd[(k + 1) % RESIDENT_RING];
						auto word = [](const void* p) { return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
						unsigned long long cIn = word(&cmd->in), cOut = word(&cmd->out), cIs = word(&cmd->inStride), cOs = word(&cmd->outStride);
						unsigned long long cSeq = word(&cmd->seq), cChk = word(&cmd->check);
						if (finished)
						{
							// Ordering of the output rows before the count, and of every workgroup's rows before `completed`: done BY HAND, not by
							// release / acquire orders on these atomics.  The rows are stored write-through at system scope (sc0 sc1) and every wave
							// waits for its stores (s_waitcnt vmcnt(0)) before the workgroup barrier in front of this block, so they have left the
							// chip when the count is incremented; the count and `completed` are device / system scope atomics that bypass the
							// non-coherent cache levels.  An ACQ_REL order here makes the compiler write back the WHOLE dirty L2 of the XCD
							// (buffer_wbl2: the ring state of every stream, which the host never reads) per workgroup and command -- measured
							// 38.4 -> 49.9 us per 1024 x 128 step, 43 -> 54 us per lone buffer (r06), for rows that are already out.
							const unsigned before = __hip_atomic_fetch_add(&r.doneCount[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							if (before == gridDim.x - 1)
							{
								// the last workgroup of command k: the slot's counter is free again (the host does not reuse the slot before it
								// has seen `completed`), and the host may read the output rows
								__hip_atomic_store(&r.doneCount[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
								__hip_atomic_store(&r.status->completed, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (see above; the host loads it with acquire)
							}
						}
						const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
						bool leaving = false;
						while (cSeq != k + 1 || cChk != ResidentCmdCheck(cIn, cOut, cIs, cOs, cSeq))
						{
							if (leaving) { go = 0; break; } // (a command posted from now on is not lost: the relaunch resumes at wgDone)
							// nothing to do: told to leave behind command k, or idle for too long (the host relaunches when there is work) -- after
							// one more look
							leaving = word(&r.ctrl->exitAfter) <= k || __builtin_amdgcn_s_memrealtime() - t0 > r.idleTicks;
							if (!leaving) __builtin_amdgcn_s_sleep(4);
							cIn = word(&cmd->in), cOut = word(&cmd->out), cIs = word(&cmd->inStride), cOs = word(&cmd->outStride);
							cSeq = word(&cmd->seq), cChk = word(&cmd->check);
						}
						if (go)
						{
							bc[1] = cIn;
							bc[2] = cOut;
							bc[3] = cIs;
							bc[4] = cOs;
						}
						blk = (long)blockIdx.x;
					}
					bc[0] = go;
					bc[5] = k;
					bc[6] = (unsigned long long)blk;
				}
				BlockBarrier<C::NTHREADS / 64>();
				if (__builtin_amdgcn_readfirstlane((int)bc[0]) == 0) break;
				{
					// (wave-uniform: scalar registers, like kernel arguments)
					auto uni = [](unsigned long long v) {
						return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
					};
					const float* in = reinterpret_cast<const float*>(uni(bc[1]));
					float* out = reinterpret_cast<float*>(uni(bc[2]));
					const long inStride = (long)uni(bc[3]), outStride = (long)uni(bc[4]);
					const int blk = __builtin_amdgcn_readfirstlane((int)bc[6]);
					int gi = 0;
					for (int i = 1; i < args.numGroups; i++)
						if (blk >= args.g[i].firstBlock) gi = i;
					// (opaque: what the chain derives from the thread index and from the group's tables is derived afresh per block, exactly
					// like in the one-shot kernel -- hoisted out of this loop it would stay live across the whole chain)
					asm volatile("" : "+v"(gi));
					const GroupArgs& ga = args.g[__builtin_amdgcn_readfirstlane(gi)];
					int tid = (int)threadIdx.x;
					asm volatile("" : "+v"(tid));
					const int groupBlock = blk - ga.firstBlock;
#ifdef NA_SP_TRACE
					long long* tr = ((int)blockIdx.x == traceBlock) ? trace : nullptr;
					if (tr != nullptr && threadIdx.x == 0)
					{
						tr[(C::TB::NSTAGES * 8 + 4) * (C::NTHREADS / 64)] = tr[(C::TB::NSTAGES * 8 + 2) * (C::NTHREADS / 64)]; // (the closing barrier of the block before)
						tr[(C::TB::NSTAGES * 8 + 3) * (C::NTHREADS / 64)] = (long long)__builtin_readcyclecounter();
					}
					if (F::N > 1 && ga.arch == 1) RunWorkgroup<C1>(ga, groupBlock, in, out, inStride, outStride, tr, tid);
					else RunWorkgroup<C>(ga, groupBlock, in, out, inStride, outStride, tr, tid);
#else
					if (F::N > 1 && ga.arch == 1) RunWorkgroup<C1>(ga, groupBlock, in, out, inStride, outStride, tid);
					else RunWorkgroup<C>(ga, groupBlock, in, out, inStride, outStride, tid);
#endif
				}
				// every wave's ring stores, cursors and output rows have left the wave before the next block / the done count
				__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
				BlockBarrier<C::NTHREADS / 64>();
#ifdef NA_SP_TRACE
				if ((int)blockIdx.x == traceBlock && trace != nullptr && threadIdx.x == 0) trace[(C::TB::NSTAGES * 8 + 2) * (C::NTHREADS / 64)] = (long long)__builtin_readcyclecounter();
#endif
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
				if (d.type != WN_ST_LAYER || d.Gp != TB::GPof(a) || d.G != TB::GPof(a) || d.ksize != TB::KS(L) || d.dilation != TB::Dil(L)) return false;
				if (d.ring_id != L || d.ring_off != TB::RingOff(L) || d.ring_frames != TB::RingFrames(L)) return false;
				if (((d.flags & WN_FLAG_LEAKY) != 0) != A::LEAKY || (d.flags & WN_FLAG_STD_TANH)) return false;
				if (!TB::LastOfArr(L) && (d.out_ring_id != L + 1 || !(d.flags & WN_FLAG_PUBLISH))) return false;
			}
			for (int a = 1; a < TB::NA; a++)
			{
				const WnSplitStage& d = st[TB::LinkStage(a)];
				if (d.type != WN_ST_ARRAY_LINK || d.Gp != TB::GPof(a - 1) || d.ksize != TB::GPof(a) || d.out_ring_id != TB::FirstLayerOfArr(a)) return false;
			}
			const WnSplitStage& h = st[nstages - 1];
			if (h.ksize != TB::HEADK || h.Gp != TB::GPof(TB::NA - 1)) return false;
			if (TB::HEADK == 1) return h.type == WN_ST_HEAD_DENSE_OUT;
			return h.type == WN_ST_HEAD_CONV_OUT && h.dilation == 1 && h.ring_id == TB::NL && h.ring_off == TB::RingOff(TB::NL) && h.ring_frames == TB::RingFrames(TB::NL);
		}

		// the kernarg tables of a launch list; *blocksOut = workgroups it takes
		template <class F, int NF, int SPB, bool PK>
		static hipError_t FillGroupArgs(const WnFrameGroup* groups, int numGroups, GroupArgs* out, int* blocksOut)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK> C;
			typedef Cfg<typename F::A1, NF, SPB, PK> C1;
			int blocks = 0;
			const bool reverse = Tuning::Get().spReverse; // tuning: the groups' workgroups in the opposite dispatch order
			for (int i = 0; i < numGroups; i++)
			{
				const WnFrameGroup& g = groups[reverse ? numGroups - 1 - i : i];
				const WnModelDev& m = *g.model;
				out[i] = {};
				GroupArgs& a = out[i];
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
				a.arch = (F::N > 1 && (m.spec_arch == WN_SPEC_LITE16 || m.spec_arch == WN_SPEC_A2LITE)) ? 1 : 0;
				// channel groups per real stream of the first / the last array (packed launches: which stream's condition a channel group sees)
				const int c0 = a.arch == 1 ? F::A1::CH[0] : F::A0::CH[0], c1 = a.arch == 1 ? F::A1::CH[1] : F::A0::CH[1];
				a.gps0 = std::max(1, c0 / 4 / a.pack);
				a.gps1 = (a.pack > 1 && c1 < 4 * a.pack) ? 0 : std::max(1, c1 / 4 / a.pack); // (0: a dense pack -- two streams per channel group)
				if (a.pack > 1 && !PK) return hipErrorInvalidValue;
				if (PK && g.slots == nullptr) return hipErrorInvalidValue;
				const int spbArch = a.arch == 1 ? C1::SPB : C::SPB; // streams per workgroup of this group's architecture
				blocks += (g.numStreams + spbArch - 1) / spbArch;
			}
			*blocksOut = blocks;
			return hipSuccess;
		}

		template <class F, int NF, int SPB, bool PK>
		static hipError_t FillLaunchArgs(const WnFrameGroup* groups, int numGroups, LaunchArgs& args, int* blocksOut)
		{
			args = {};
			args.numGroups = numGroups;
			return FillGroupArgs<F, NF, SPB, PK>(groups, numGroups, args.g, blocksOut);
		}

		// ---- table launches: any number of model groups in ONE launch ---------------------------------------------------------------
		// The kernarg segment holds WN_FRAME_MAX_GROUPS groups; a batch of hundreds of DIFFERENT models (every stream its own capture: what a
		// server sees) would be cut into launches of eight groups each, one after the other -- 1024 models x 1 stream: 128 launches, 2.9 ms
		// per 128-frame buffer.  Here the group table lives in device memory (rebuilt and uploaded when the batch's topology changes): a
		// workgroup finds its group by binary search over GroupArgs::firstBlock with scalar loads and copies the entry, the chain is the
		// same code.  (Table kernels exist for 128- and 64-frame blocks; other lengths keep the launches of eight.)
		template <class F, int NF, int SPB, bool PK>
		__global__ void __launch_bounds__(64 * (NF / (16 * F::A0::T)) * (SPB * F::A0::T / 2)) __attribute__((amdgpu_waves_per_eu(OccOf<F, NF, SPB>()))) WaveNetSpecTableKernel(
			const GroupArgs* __restrict__ table, int numGroups, const float* __restrict__ in, float* __restrict__ out, long inStride, long outStride)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK> C;
			typedef Cfg<typename F::A1, NF, SPB, PK> C1;
			extern __shared__ __attribute__((aligned(16))) char dynSmem[];
			asm volatile("" : : "s"((unsigned)(size_t)(LdsPtr)dynSmem)); // the dynamic LDS segment is in use (and starts at 0)
			// (the table is read-only for the launch: constant address space = scalar loads)
			typedef const __attribute__((address_space(4))) GroupArgs* TablePtr;
			TablePtr tab = (TablePtr)(size_t)table;
			int lo = 0, hi = numGroups - 1;
			while (lo < hi)
			{
				const int mid = (lo + hi + 1) >> 1;
				if (tab[mid].firstBlock <= (int)blockIdx.x) lo = mid;
				else hi = mid - 1;
			}
			// the entry, field by field out of the constant address space: scalar loads into SGPRs, like kernel arguments (a word-wise copy of
			// the struct kept it on the stack -- pointers in VGPRs, 19 - 30 spills in the chain)
			GroupArgs ga;
			{
				const auto& e = tab[lo];
				ga.stages = e.stages; ga.wsplit = e.wsplit; ga.ringFrames = e.ringFrames; ga.state = e.state; ga.slots = e.slots; ga.rows = e.rows;
				ga.nstages = e.nstages; ga.nrings = e.nrings; ga.stateF4 = e.stateF4; ga.wsplitQuads = e.wsplitQuads;
				ga.headScale = e.headScale;
				ga.numStreams = e.numStreams; ga.slot0 = e.slot0; ga.row0 = e.row0;
				ga.maxG = e.maxG; ga.firstBlock = e.firstBlock; ga.pack = e.pack; ga.condLimit = e.condLimit; ga.arch = e.arch;
				ga.gps0 = e.gps0; ga.gps1 = e.gps1; ga.saturate = e.saturate;
			}
			const int groupBlock = (int)blockIdx.x - ga.firstBlock;
#ifdef NA_SP_TRACE
			if (F::N > 1 && ga.arch == 1) RunWorkgroup<C1>(ga, groupBlock, in, out, inStride, outStride, nullptr);
			else RunWorkgroup<C>(ga, groupBlock, in, out, inStride, outStride, nullptr);
#else
			if (F::N > 1 && ga.arch == 1) RunWorkgroup<C1>(ga, groupBlock, in, out, inStride, outStride);
			else RunWorkgroup<C>(ga, groupBlock, in, out, inStride, outStride);
#endif
		}

		// host side of a table launch: the table is rebuilt every call (it is a few stores per group) and looked up among the device
		// copies the batch holds (WnLaunchTable::Ensure uploads a new one -- never inside a stream capture)
		template <class F, int NF, int SPB, bool PK>
		static hipError_t LaunchTable(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, hipStream_t stream, WnLaunchTable& t)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK> C;
			typedef Cfg<typename F::A1, NF, SPB, PK> C1;
			std::vector<GroupArgs> fresh((size_t)numGroups);
			int blocks = 0;
			const hipError_t fe = FillGroupArgs<F, NF, SPB, PK>(groups, numGroups, fresh.data(), &blocks);
			if (fe != hipSuccess) return fe;
			const void* dev = nullptr;
			const hipError_t ee = t.Ensure(fresh.data(), fresh.size() * sizeof(GroupArgs), stream, &dev);
			if (ee != hipSuccess) return ee;
			if (t.prepareOnly) return hipSuccess;
			constexpr int LDS_BYTES = C::LDS_BYTES > C1::LDS_BYTES ? C::LDS_BYTES : C1::LDS_BYTES;
			if (LDS_BYTES > 64 * 1024)
			{
				static PerDeviceOnce attr; // per instantiation and device
				const hipError_t e = attr.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(WaveNetSpecTableKernel<F, NF, SPB, PK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); });
				if (e != hipSuccess) return e;
			}
			hipLaunchKernelGGL((WaveNetSpecTableKernel<F, NF, SPB, PK>), dim3((unsigned)blocks), dim3(C::NTHREADS), LDS_BYTES, stream,
				reinterpret_cast<const GroupArgs*>(dev), numGroups, in, out, inStride, outStride);
			return hipGetLastError();
		}

		template <class F, int NF, int SPB, bool PK, bool NT = false>
		static hipError_t Launch(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, hipStream_t stream)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK> C;
			typedef Cfg<typename F::A1, NF, SPB, PK> C1;
			LaunchArgs args;
			int blocks = 0;
			const hipError_t fe = FillLaunchArgs<F, NF, SPB, PK>(groups, numGroups, args, &blocks);
			if (fe != hipSuccess) return fe;
			constexpr int LDS_BYTES = C::LDS_BYTES > C1::LDS_BYTES ? C::LDS_BYTES : C1::LDS_BYTES;
			if (LDS_BYTES > 64 * 1024)
			{
				static PerDeviceOnce attr; // per instantiation and device
				const hipError_t e = attr.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(WaveNetSpecKernel<F, NF, SPB, PK, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); });
				if (e != hipSuccess) return e;
			}
			hipLaunchKernelGGL((WaveNetSpecKernel<F, NF, SPB, PK, NT>), dim3((unsigned)blocks), dim3(C::NTHREADS), LDS_BYTES, stream, args, in, out, inStride, outStride
#ifdef NA_SP_TRACE
				, GetWaveNetTraceBuffer(), Tuning::Get().traceBlock
#endif
				);
			return hipGetLastError();
		}

		// The resident launch of the same list (WaveNetSpecResidentKernel): as many workgroups as are resident at the kernel's occupancy,
		// at most one per block of the list.  ra.numBlocks is filled in here.  stream == nullptr: only report the grid.
		template <class F, int NF, int SPB, bool PK>
		static hipError_t LaunchResident(const WnFrameGroup* groups, int numGroups, ResidentArgs ra, hipStream_t stream, int* gridOut)
		{
			typedef Cfg<typename F::A0, NF, SPB, PK, true> C;
			typedef Cfg<typename F::A1, NF, SPB, PK, true> C1;
			LaunchArgs args;
			int blocks = 0;
			const hipError_t fe = FillLaunchArgs<F, NF, SPB, PK>(groups, numGroups, args, &blocks);
			if (fe != hipSuccess) return fe;
			constexpr int LDS_BYTES = C::LDS_BYTES > C1::LDS_BYTES ? C::LDS_BYTES : C1::LDS_BYTES;
			static PerDeviceOnce attr;
			static std::atomic<int> perCU[kMaxHipDevices];
			int device = 0;
			hipError_t e = hipGetDevice(&device);
			if (e != hipSuccess) return e;
			if (device < 0 || device >= kMaxHipDevices) return hipErrorInvalidDevice;
			e = attr.Run([device] {
				hipError_t r = hipSuccess;
				if (LDS_BYTES > 64 * 1024)
					r = hipFuncSetAttribute(reinterpret_cast<const void*>(WaveNetSpecResidentKernel<F, NF, SPB, PK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
				if (r != hipSuccess) return r;
				int nb = 0;
				r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(WaveNetSpecResidentKernel<F, NF, SPB, PK>), C::NTHREADS, LDS_BYTES);
				perCU[device].store(nb);
				return r;
			});
			if (e != hipSuccess) return e;
			const int resident = perCU[device].load() * CurrentDeviceCUs();
			int grid = std::min(blocks, resident);
			if (Tuning::Get().residentGrid > 0) grid = std::min(grid, Tuning::Get().residentGrid);
			*gridOut = grid;
			if (grid < 1) return hipErrorInvalidValue;
			if (stream == nullptr) return hipSuccess;
			ra.numBlocks = blocks;
			hipLaunchKernelGGL((WaveNetSpecResidentKernel<F, NF, SPB, PK>), dim3((unsigned)grid), dim3(C::NTHREADS), LDS_BYTES, stream, args, ra
#ifdef NA_SP_TRACE
				, GetWaveNetTraceBuffer(), Tuning::Get().traceBlock
#endif
				);
			return hipGetLastError();
		}

		template <class F, bool PK>
		static hipError_t LaunchNF(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, hipStream_t stream,
			bool beyondCache = false)
		{
#ifdef NA_SP_QUICK
			(void)spb; (void)n; (void)beyondCache;
			return Launch<F, 128, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream);
#else
			// (the non-temporal variant exists for full-size workgroups of 128-frame blocks: the regime it is for -- more state than the
			// Infinity Cache holds -- is thousands of streams)
			if constexpr (F::A0::T != 1)
				if (beyondCache && n == 128 && spb >= 2) return Launch<F, 128, 2, PK, true>(groups, numGroups, in, out, inStride, outStride, stream);
			if constexpr (F::A0::T == 1)
			{
				// one tile per wave: one stream per workgroup (SPB_ = 2 in units of four-wave streams), every block length
				if (n == 128) return Launch<F, 128, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream);
				if (n == 64) return Launch<F, 64, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream);
				return Launch<F, 32, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream);
			}
			else
			{
			if (n == 128) return spb >= 2 ? Launch<F, 128, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream) : Launch<F, 128, 1, PK>(groups, numGroups, in, out, inStride, outStride, stream);
			if (n == 64) return spb >= 2 ? Launch<F, 64, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream) : Launch<F, 64, 1, PK>(groups, numGroups, in, out, inStride, outStride, stream);
			if constexpr (F::A0::T == 2 && F::A1::T == 2)
				return spb >= 2 ? Launch<F, 32, 2, PK>(groups, numGroups, in, out, inStride, outStride, stream) : Launch<F, 32, 1, PK>(groups, numGroups, in, out, inStride, outStride, stream);
			else return hipErrorNotSupported; // (a wave of a 4-tile architecture covers 64 frames)
			}
#endif
		}

		// (defined in the family's translation unit)
		hipError_t LaunchSpecLite(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, bool packed,
			hipStream_t stream, bool oneTilePerWave = false, bool beyondCache = false);
		hipError_t LaunchSpecA2(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, hipStream_t stream,
			bool beyondCache = false);
		hipError_t LaunchSpecA2Table(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, hipStream_t stream,
			WnLaunchTable& table);
		hipError_t LaunchSpecLiteTable(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, bool packed,
			hipStream_t stream, WnLaunchTable& table);
	}
}
