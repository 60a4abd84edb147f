// wavenet_generic_kernels.hip -- the runtime-shaped WaveNet block kernel: any channel count up to 64, any kernel sizes / dilations /
// layer counts, dense or conv heads -- what the reference's dynamic engine accepts beyond the official architectures
// (NeuralAudio/WaveNetDynamic.h:67-83, 229-254, 445-468, InternalModel.h:177-248; same arithmetic as WaveNet.h:768-799, 632-661, 462-494).
//
// It runs what the shaped kernels cannot -- layer arrays wider than 16 channels -- and these are exactly the layers north_star wants on
// the matrix pipe ("MFMA only for the widest WaveNet layer GEMMs"): every mat-mul of a layer (conv taps, 1x1, rechannel, dense head) is
// v_mfma_f32_16x16x32_f16 with the three-product f16 split of the shaped kernels (wavenet_split_dev.h: W x = Wh xh + Wh xl + Wl xh,
// f32 accumulation).  It walks the natural-layout tensor table the prewarm kernel uses (WnPrewarmLayer: offsets into the flat weight
// array in the reference's order, WaveNet.h:700-719) and keeps the frame kernel's stream-state format (f32 quads, tile layout), so
// prewarm / reset are shared.
//
//   workgroup = one stream = 512 threads = 8 waves; wave w owns the 16-frame tile w of the 128-frame block for the mat-muls
//   (lane = (frame j, channel group q): result registers = 4 channels of the lane's own frame); for gathers and ring traffic thread
//   (f = tid & 127, cq = tid >> 7) moves the channel groups g = cq, cq + 4, ... of frame f;
//   LDS: X[G][128] layer input, HEAD[G][128] head accumulator -- float4 per (4-channel group,
//   frame) -- and the A operands of ONE matrix, split to f16 hi / lo ONCE per workgroup while they are staged ([row block][k block][hi,
//   lo][64 lanes] x 16 bytes);
//   a layer = publish X to its ring, load the ring history its taps need -> per tap: stage the tap's matrix, read the tap's input where
//   the mat-mul wants it (X for in-block frames, the prefetched ring history for earlier ones), MFMA into register accumulators -> bias + mix-in + activation + head accumulate in the result lanes (the
//   activation's split quad IS the 1x1's B operand: no LDS round trip) -> stage the 1x1 -> MFMA -> residual into X.
#include "device_once.h"
#include <algorithm>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"
#include "wavenet_split_dev.h"

namespace na
{
	namespace gn
	{
		constexpr int FRAMES = WN_MAX_FRAMES;
		constexpr int NTHREADS = 512;
		typedef float f32x4 __attribute__((ext_vector_type(4)));
		using sp::u32x4;

		// Activation.h:83-91
		__device__ __forceinline__ float FastTanh(float x)
		{
			const float ax = fabsf(x);
			const float x2 = x * x;
			return (x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2)) *
				__builtin_amdgcn_rcpf(2.44506634652299f + (2.44506634652299f + x2) * (ax + 0.814642734961073f * x2));
		}
		__device__ __forceinline__ float Activate(float v, int act)
		{
			if (act == 1) return v > 0.0f ? v : 0.01f * v;                                                                       // Activation.h:110-118
			if (act == 2) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v * 2.885390081777927f) + 1.0f); // StdMath tanh
			return FastTanh(v);
		}

		// float4 index of (ring position p, channel group g) in the tile layout with G channel groups: ((p >> 4) G + g) 16 + (p & 15)
		__device__ __forceinline__ size_t RingQuad(int ringOffF4, int G, int p, int g) { return (size_t)ringOffF4 + (size_t)(((p >> 4) * G + g) * 16 + (p & 15)); }

		struct Args
		{
			const WnPrewarmLayer* layers;
			int numLayers;
			const float* w; // flat weights, reference order
			const int* ringOffF4;
			const int* ringFrames;
			const int* ringG;
			int nrings, stateF4, maxC;
			float headScale;
			float* state;
			const int* slots; // nullptr: contiguous (slot0 + i, row0 + i)
			const int* rows;
			int slot0, row0;
		};

		// W[o][c] = w[off + o * cin + c] -> the A operands of the [cout x cin] matrix: operand (rb, kb, hi | lo), lane (i = row 16 rb + i, q = channels 16 kb + 4 q .. + 3).  Split once per
		// workgroup.  Two halves so that the weights of the NEXT mat-mul travel (global -> registers: StageLoad) while the current one
		// runs, and are split and written to LDS (StageCommit) once its operands are free: a layer is 4 .. K + 1 dependent mat-muls, and
		// with the load in front of each of them 30 % of the kernel was weight-load latency (64 / 32 channels: 306 -> 213 us without loads).
		// workgroup barrier that orders LDS traffic only: the weight loads of the next mat-mul (StageLoad) stay in flight across it --
		// __syncthreads() would wait for them.  (No thread of the block reads global memory another one wrote in this launch: ring
		// stores are history for LATER blocks.)
		__device__ __forceinline__ void LdsBarrier()
		{
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
			__builtin_amdgcn_s_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
		}
		// a row-major [cout x cin] matrix at w[off] (the weight copy of this kernel keeps layer convs tap-major: tap k of a conv at wconv
		// is the matrix at wconv + k cout cin; gpu_batch.cpp)
		struct MatRef
		{
			int off, cout, cin, nbo, nbk;
			int ld = 0; // floats between rows (0: cin) -- a sub-matrix of a wider one
		};
		typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4))); // a 16-byte load from a 4-byte aligned address
		template <int NB>
		struct StagedRegs
		{
			static constexpr int ITEMS = (NB * NB * 64 + NTHREADS - 1) / NTHREADS;
			sp::f32x4 v[ITEMS];
		};
		template <int NB>
		__device__ __forceinline__ void StageLoad(StagedRegs<NB>& r, const float* __restrict__ w, const MatRef& m)
		{
#pragma unroll
			for (int it = 0; it < StagedRegs<NB>::ITEMS; it++)
			{
				const int idx = threadIdx.x + it * NTHREADS;
				const int lane = idx & 63, blk = idx >> 6, rb = blk / m.nbk, kb = blk % m.nbk;
				const int o = 16 * rb + (lane & 15), c0 = 16 * kb + 4 * (lane >> 4);
				const bool on = idx < m.nbo * m.nbk * 64 && o < m.cout;
				const float* p = w + m.off + (size_t)o * (m.ld ? m.ld : m.cin) + c0;
				sp::f32x4 v = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				if (on && c0 + 3 < m.cin)
				{
					const f32x4u t = *reinterpret_cast<const f32x4u*>(p);
					v = sp::f32x4{ t.x, t.y, t.z, t.w };
				}
				else if (on)
				{
					v.x = (c0 + 0 < m.cin) ? p[0] : 0.0f;
					v.y = (c0 + 1 < m.cin) ? p[1] : 0.0f;
					v.z = (c0 + 2 < m.cin) ? p[2] : 0.0f;
				}
				r.v[it] = v;
			}
		}
		template <int NB>
		__device__ __forceinline__ void StageCommit(u32x4* ops, const StagedRegs<NB>& r, const MatRef& m)
		{
#pragma unroll
			for (int it = 0; it < StagedRegs<NB>::ITEMS; it++)
			{
				const int idx = threadIdx.x + it * NTHREADS;
				if (idx < m.nbo * m.nbk * 64)
				{
					const int lane = idx & 63, blk = idx >> 6;
					const u32x4 s = sp::SplitQuad(r.v[it]);                     // [h01 | h23 | l01 | l23]
					ops[(blk * 2 + 0) * 64 + lane] = u32x4{ s.x, s.y, s.x, s.y }; // Wh against the h AND the l half of the operand
					ops[(blk * 2 + 1) * 64 + lane] = u32x4{ s.z, s.w, 0u, 0u };   // Wl against the h half
				}
			}
		}
		template <int NB>
		__device__ __forceinline__ void StageMatrix(u32x4* ops, const float* __restrict__ w, int off, int cout, int cin, int nbo, int nbk)
		{
			const MatRef m = { off, cout, cin, nbo, nbk };
			StagedRegs<NB> r;
			StageLoad<NB>(r, w, m);
			StageCommit<NB>(ops, r, m);
		}

		// acc[rb] += M[16 rb .. + 16][all k blocks] * b[kb]   (b[kb] = split quad of channels 16 kb + 4 q .. of the lane's frame)
		template <int NB>
		__device__ __forceinline__ void MatMul(const u32x4* ops, int nbo, int nbk, int lane, const u32x4 (&b)[NB], sp::f32x4 (&acc)[NB])
		{
#pragma unroll
			for (int rb = 0; rb < NB; rb++)
			{
				if (rb >= nbo) break;
#pragma unroll
				for (int kb = 0; kb < NB; kb++)
				{
					if (kb >= nbk) break;
					const int blk = rb * nbk + kb;
					acc[rb] = sp::Mfma(ops[(blk * 2 + 0) * 64 + lane], b[kb], acc[rb]);
					acc[rb] = sp::Mfma(ops[(blk * 2 + 1) * 64 + lane], b[kb], acc[rb]);
				}
			}
		}

		__device__ __forceinline__ f32x4 Load4(const float* __restrict__ w, int off, int c0, int count)
		{
			f32x4 v;
			v.x = (c0 + 0 < count) ? w[off + c0 + 0] : 0.0f;
			v.y = (c0 + 1 < count) ? w[off + c0 + 1] : 0.0f;
			v.z = (c0 + 2 < count) ? w[off + c0 + 2] : 0.0f;
			v.w = (c0 + 3 < count) ? w[off + c0 + 3] : 0.0f;
			return v;
		}

		// OCC = waves per SIMD the kernel is compiled for.  2 (256 VGPRs) is the faster code for one workgroup per CU; models of up to 32
		// channels fit two workgroups into a CU's LDS, and a batch with more streams than CUs wants the 128-VGPR build (4) so that two
		// ARE resident (32 / 16 channels: 256 streams 107 vs 122 us, 512 streams 208 vs 157, 1024 streams 423 vs 308).
		template <int NB, int OCC>
		__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(OCC))) WaveNetGenericKernel(const Args a, const float* __restrict__ in, float* __restrict__ out, long inStride,
			long outStride, int n)
		{
			extern __shared__ __attribute__((aligned(16))) float lds[];
			const int GQ = (a.maxC + 3) / 4;                          // channel groups of the widest array
			f32x4* X = reinterpret_cast<f32x4*>(lds);                 // [GQ][FRAMES]
			f32x4* HEAD = X + (size_t)GQ * FRAMES;
			// two operand buffers, used in turn by consecutive mat-muls: a matrix is committed into the one the mat-mul before the last read,
			// and the barrier between every commit and its mat-mul is the only one the operands need
			constexpr int OPS_ONE = NB * NB * 2 * 64;
			u32x4* ops0 = reinterpret_cast<u32x4*>(HEAD + (size_t)GQ * FRAMES); // [2][NB * NB][2][64]
			float* condL = reinterpret_cast<float*>(ops0 + 2 * OPS_ONE);      // [FRAMES]
			int par = 0; // mat-muls so far (workgroup-uniform)
			const int tid = threadIdx.x;
			const int f = tid & (FRAMES - 1), cq = tid >> 7;          // gather / ring role
			const int lane = tid & 63, wave = tid >> 6;               // mat-mul role: tile = wave
			const int j = lane & 15, q = lane >> 4, tf = 16 * wave + j; // the lane's frame
			const int sidx = blockIdx.x;
			const int slot = a.slots ? a.slots[sidx] : a.slot0 + sidx;
			const int row = a.slots ? a.rows[sidx] : a.row0 + sidx;
			float* st = a.state + (size_t)slot * (size_t)a.stateF4 * 4;
			f32x4* stq = reinterpret_cast<f32x4*>(st);
			int* header = reinterpret_cast<int*>(st);
			if (cq == 0) condL[f] = (f < n) ? in[(size_t)row * inStride + f] : 0.0f; // WaveNet.h:770 (input -> condition)
			for (int g = cq; g < GQ; g += 4)
			{
				X[g * FRAMES + f] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
				HEAD[g * FRAMES + f] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; // :772 headArray.SetZero()
			}
			__syncthreads();
			const float cond = condL[tf];

			const float* __restrict__ w = a.w;
			StagedRegs<NB> pre;   // weights of the next mat-mul, under way while the current one runs
			bool havePre = false; // (workgroup-uniform)
			for (int li = 0; li < a.numLayers; li++)
			{
				const WnPrewarmLayer L = a.layers[li];
				if (L.kind == 0)
				{
					const int cin = L.cin; // == cout
					const int Gl = (cin + 3) / 4, nb = (cin + 15) / 16;
					// the layer's bias / mix-in vectors for the rows this lane will hold, loaded now: they are used behind the mat-muls, and a load
					// issued there is a memory round trip on the critical path of every layer
					f32x4 bcv[NB], wmv[NB], b1v[NB];
#pragma unroll
					for (int rb = 0; rb < NB; rb++)
					{
						const int g = 4 * rb + q;
						const bool on = rb < nb && g < Gl;
						bcv[rb] = on ? Load4(w, L.bconv, 4 * g, cin) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						wmv[rb] = on ? Load4(w, L.wmix, 4 * g, cin) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						b1v[rb] = on ? Load4(w, L.b1, 4 * g, cin) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
					}
					if (L.rechannel >= 0)
					{
						// rechannel (:637): array 0 from the condition (input_size == 1), later arrays from the previous array's output (in X)
						if (li == 0 && L.rech_in == 1)
						{
							for (int rb = 0; rb < nb; rb++)
							{
								const int g = 4 * rb + q;
								if (g < Gl) X[g * FRAMES + tf] = Load4(w, L.rechannel, 4 * g, cin) * cond;
							}
						}
						else
						{
							const int nbk = (L.rech_in + 15) / 16, Gin = (L.rech_in + 3) / 4;
							__syncthreads(); // the operand buffer is free (the previous array's head mat-mul is done everywhere)
							u32x4* ops = ops0 + (par++ & 1) * OPS_ONE;
							StageMatrix<NB>(ops, w, L.rechannel, cin, L.rech_in, nb, nbk);
							u32x4 b[NB];
#pragma unroll
							for (int kb = 0; kb < NB; kb++)
								b[kb] = (kb < nbk && 4 * kb + q < Gin) ? sp::SplitQuad(X[(4 * kb + q) * FRAMES + tf]) : u32x4{ 0, 0, 0, 0 };
							__syncthreads(); // operands staged; every wave has read its own frames of the old X
							sp::f32x4 acc[NB];
#pragma unroll
							for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
							MatMul<NB>(ops, nb, nbk, lane, b, acc);
#pragma unroll
							for (int rb = 0; rb < NB; rb++)
								if (rb < nb && 4 * rb + q < Gl) X[(4 * rb + q) * FRAMES + tf] = acc[rb];
						}
					}
					__syncthreads(); // X of every frame is complete; the operand buffer is free
					// the layer input of this block -> its ring (history for LATER blocks: only the last R - FRAMES frames can be read back)
					const int R = a.ringFrames[L.ring_id], G = a.ringG[L.ring_id], roff = a.ringOffF4[L.ring_id];
					const int pos0 = header[L.ring_id];
					{
						int p = pos0 + f;
						if (p >= R) p -= R;
						if (f < n && f >= n - (R - FRAMES))
							for (int g = cq; g < Gl; g += 4) stq[RingQuad(roff, G, p, g)] = X[g * FRAMES + f];
					}
					// dilated conv (:139-290), tap by tap: stage the tap's [cin x cin] matrix as split A operands, gather the tap's input
					// (LDS for in-block frames, the ring for earlier ones), accumulate on the matrix pipe
					const int K = L.ksize;
					sp::f32x4 acc[NB];
#pragma unroll
					for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
					// A tap's B operand is the layer input `shift` frames back, read where the mat-mul wants it (lane = frame of the wave's tile,
					// channel group 4 kb + q): from X inside the block, from the ring before it.  The ring part of up to three taps (K <= 3: every
					// A1-shaped model) is loaded HERE, ahead of the whole tap loop, so that its HBM round trip runs behind the first mat-muls.
					auto histOf = [&](int k, f32x4 (&h)[NB]) {
						const int src = tf - L.dilation * (K - 1 - k);
						int p = pos0 + src; // src >= -(R - FRAMES): one wrap
						if (p < 0) p += R;
#pragma unroll
						for (int kb = 0; kb < NB; kb++)
						{
							const int g = 4 * kb + q;
							h[kb] = (src < 0 && kb < nb && g < Gl) ? stq[RingQuad(roff, G, p, g)] : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						}
					};
					f32x4 hist0[NB], hist1[NB], hist2[NB];
					if (K <= 3)
					{
						histOf(0, hist0);
						if (K > 1) histOf(1, hist1);
						if (K > 2) histOf(2, hist2);
					}
					auto tap = [&](int k, const f32x4 (&h)[NB]) {
						u32x4* ops = ops0 + (par++ & 1) * OPS_ONE;
						const MatRef mk = { L.wconv + k * cin * cin, cin, cin, nb, nb };
						if (!havePre) StageLoad<NB>(pre, w, mk); // (the first tap behind a rechannel / head stage: nothing was under way)
						StageCommit<NB>(ops, pre, mk);
						// the next mat-mul's weights set out now: the next tap's, or the 1x1's behind the last tap
						const MatRef mn = (k + 1 < K) ? MatRef{ L.wconv + (k + 1) * cin * cin, cin, cin, nb, nb } : MatRef{ L.w1, cin, cin, nb, nb };
						StageLoad<NB>(pre, w, mn);
						havePre = true;
						LdsBarrier();
						const int src = tf - L.dilation * (K - 1 - k); // tap k reads the frame d (K-1-k) back
						u32x4 b[NB];
#pragma unroll
						for (int kb = 0; kb < NB; kb++)
						{
							const int g = 4 * kb + q;
							b[kb] = (kb < nb && g < Gl) ? sp::SplitQuad(src >= 0 ? X[g * FRAMES + src] : h[kb]) : u32x4{ 0, 0, 0, 0 };
						}
						MatMul<NB>(ops, nb, nb, lane, b, acc);
					};
					if (K <= 3)
					{
						tap(0, hist0);
						if (K > 1) tap(1, hist1);
						if (K > 2) tap(2, hist2);
					}
					else
					{
						for (int k = 0; k < K; k++)
						{
							histOf(k, hist0);
							tap(k, hist0);
						}
					}
					// bias + mix-in (:288-289, :471), activation (:473-480), head accumulate (:482) -- in the lanes that hold the results; the
					// activation's split quad is the 1x1's B operand (result rows 16 rb + 4 q .. of the lane's frame = k block rb, group q)
					u32x4 zs[NB];
#pragma unroll
					for (int rb = 0; rb < NB; rb++)
					{
						zs[rb] = u32x4{ 0, 0, 0, 0 };
						const int g = 4 * rb + q;
						if (rb < nb && g < Gl)
						{
							const f32x4 bc = bcv[rb], wm = wmv[rb];
							f32x4 zv;
							zv.x = (4 * g + 0 < cin) ? Activate(acc[rb].x + bc.x + wm.x * cond, L.act) : 0.0f;
							zv.y = (4 * g + 1 < cin) ? Activate(acc[rb].y + bc.y + wm.y * cond, L.act) : 0.0f;
							zv.z = (4 * g + 2 < cin) ? Activate(acc[rb].z + bc.z + wm.z * cond, L.act) : 0.0f;
							zv.w = (4 * g + 3 < cin) ? Activate(acc[rb].w + bc.w + wm.w * cond, L.act) : 0.0f;
							HEAD[g * FRAMES + tf] += zv;
							zs[rb] = sp::SplitQuad(sp::f32x4{ zv.x, zv.y, zv.z, zv.w });
						}
					}
					// 1x1 + bias + residual (:486-491); the last layer's output feeds the next array's rechannel (or nothing)
					u32x4* ops = ops0 + (par++ & 1) * OPS_ONE;
					StageCommit<NB>(ops, pre, MatRef{ L.w1, cin, cin, nb, nb });
					havePre = false;
					if (li + 1 < a.numLayers)
					{
						// the first tap of the next layer, when nothing else is staged in between (same array: no rechannel, no head)
						const WnPrewarmLayer N = a.layers[li + 1];
						if (N.kind == 0 && N.rechannel < 0)
						{
							const int nbn = (N.cin + 15) / 16;
							StageLoad<NB>(pre, w, MatRef{ N.wconv, N.cin, N.cin, nbn, nbn });
							havePre = true;
						}
					}
					LdsBarrier();
#pragma unroll
					for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
					MatMul<NB>(ops, nb, nb, lane, zs, acc);
#pragma unroll
					for (int rb = 0; rb < NB; rb++)
					{
						const int g = 4 * rb + q;
						if (rb < nb && g < Gl)
						{
							const f32x4 b1 = b1v[rb];
							X[g * FRAMES + tf] += f32x4{ acc[rb].x + b1.x, acc[rb].y + b1.y, acc[rb].z + b1.z, acc[rb].w + b1.w };
						}
					}
					// (the next stage starts with a barrier: X complete, operand buffer free)
				}
				else
				{
					// head rechannel (:658-660): becomes the next array's head accumulator (:785-789) or, for the last array, the output.
					// K = 1: a dense layer on the frame's own head column (matrix pipe); K > 1 (a conv head like A2's, on the last array):
					// the head accumulator has its own ring -- publish, then every tap reads the frame (K-1-k) head_dilation back, from LDS
					// inside the block and from the ring before it (thread = frame; the head has few output channels).
					const bool last = (li == a.numLayers - 1);
					const int Gin = (L.cin + 3) / 4, nbk = (L.cin + 15) / 16, nbo = (L.cout + 15) / 16, Gout = (L.cout + 3) / 4;
					__syncthreads(); // HEAD of every frame is complete, the operand buffer is free
					if (L.ksize > 1)
					{
						const int R = a.ringFrames[L.ring_id], G = a.ringG[L.ring_id], roff = a.ringOffF4[L.ring_id];
						const int pos0 = header[L.ring_id];
						int p = pos0 + f;
						if (p >= R) p -= R;
						if (f < n && f >= n - (R - FRAMES))
							for (int g = cq; g < Gin; g += 4) stq[RingQuad(roff, G, p, g)] = HEAD[g * FRAMES + f];
						float res[4] = { 0.0f, 0.0f, 0.0f, 0.0f }; // up to 4 head outputs per pass (conv heads have one)
						if (cq == 0)
						{
							for (int o = 0; o < L.cout && o < 4; o++)
							{
								float accv = (L.bconv >= 0) ? w[L.bconv + o] : 0.0f;
								for (int k = 0; k < L.ksize; k++)
								{
									const int off = f - L.dilation * (L.ksize - 1 - k);
									int pq = pos0 + off;
									if (pq < 0) pq += R;
									for (int g = 0; g < Gin; g++)
									{
										const f32x4 hv = (off >= 0) ? HEAD[g * FRAMES + off] : stq[RingQuad(roff, G, pq, g)];
										const size_t wo = (size_t)L.wconv + ((size_t)o * L.cin + 4 * g) * L.ksize + k;
										accv += w[wo] * hv.x;
										if (4 * g + 1 < L.cin) accv += w[wo + L.ksize] * hv.y;
										if (4 * g + 2 < L.cin) accv += w[wo + 2 * L.ksize] * hv.z;
										if (4 * g + 3 < L.cin) accv += w[wo + 3 * L.ksize] * hv.w;
									}
								}
								res[o] = accv;
							}
						}
						__syncthreads(); // every tap read: HEAD may be overwritten
						if (cq == 0)
						{
							if (last) { if (f < n) out[(size_t)row * outStride + f] = a.headScale * res[0]; } // :793-798: head channel 0
							else HEAD[f] = f32x4{ res[0], res[1], res[2], res[3] };
						}
						if (!last) // (ValidateWaveNetDesc only admits a conv head on the last array; kept total for completeness)
							for (int g = cq; g < GQ; g += 4)
								if (g > 0) HEAD[g * FRAMES + f] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
					}
					else
					{
						u32x4* ops = ops0 + (par++ & 1) * OPS_ONE;
						StageMatrix<NB>(ops, w, L.wconv, L.cout, L.cin, nbo, nbk);
						u32x4 b[NB];
#pragma unroll
						for (int kb = 0; kb < NB; kb++)
							b[kb] = (kb < nbk && 4 * kb + q < Gin) ? sp::SplitQuad(HEAD[(4 * kb + q) * FRAMES + tf]) : u32x4{ 0, 0, 0, 0 };
						__syncthreads();
						sp::f32x4 acc[NB];
#pragma unroll
						for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						MatMul<NB>(ops, nbo, nbk, lane, b, acc);
						if (last)
						{
							if (q == 0 && tf < n) out[(size_t)row * outStride + tf] = a.headScale * (acc[0].x + ((L.bconv >= 0) ? w[L.bconv] : 0.0f)); // :793-798
						}
						else
						{
#pragma unroll
							for (int rb = 0; rb < NB; rb++)
							{
								const int g = 4 * rb + q;
								if (g < GQ)
								{
									f32x4 v = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
									if (rb < nbo && g < Gout)
									{
										const f32x4 bb = (L.bconv >= 0) ? Load4(w, L.bconv, 4 * g, L.cout) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
										v = f32x4{ acc[rb].x + bb.x, acc[rb].y + bb.y, acc[rb].z + bb.z, acc[rb].w + bb.w };
										if (4 * g + 1 >= L.cout) v.y = 0.0f;
										if (4 * g + 2 >= L.cout) v.z = 0.0f;
										if (4 * g + 3 >= L.cout) v.w = 0.0f;
									}
									HEAD[g * FRAMES + tf] = v; // the lane owns (group, frame): read above, written here
								}
							}
						}
					}
				}
			}
			__syncthreads();
			// advance every ring cursor by n (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
			if (tid < a.nrings)
			{
				const int R = a.ringFrames[tid];
				int p = header[tid] + n;
				if (p >= R) p -= R;
				header[tid] = p;
			}
		}

		// ------------------------------------------------------------------------------------------------------------------------------
		// Layer arrays of 65 .. 128 channels (WaveNetDynamic.h:229-254 takes any width).  The split A operands of a 128 x 128 matrix are
		// 128 KB, so a mat-mul runs as 64 x 64 sub-matrices -- output half oh, input half ih -- staged one at a time into the two operand
		// buffers of the 64-channel kernel (2 x 32 KB), X[<= 32 groups][128] stays in LDS (64 KB), and the head accumulator, which a lane
		// only ever touches at its own (frame, channel group) positions, lives in registers.  Same state format, same per-lane layouts,
		// same arithmetic order per output as the kernel above; dense heads only.  No weight prefetch across mat-muls: a stage is load ->
		// split -> commit -> barrier -> MFMA (the 64-channel kernel hides the load behind the previous mat-mul; here a 128-channel model is
		// 4 x the matrix work of a 64-channel one and the loads are a smaller share).
		template <int OCC>
		__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(OCC))) WaveNetWideKernel(const Args a, const float* __restrict__ in, float* __restrict__ out,
			long inStride, long outStride, int n)
		{
			constexpr int NB = 4, HB = 2; // 16-channel blocks per half, halves
			extern __shared__ __attribute__((aligned(16))) float lds[];
			const int GQ = (a.maxC + 3) / 4;
			f32x4* X = reinterpret_cast<f32x4*>(lds); // [GQ][FRAMES]
			constexpr int OPS_ONE = NB * NB * 2 * 64;
			u32x4* ops0 = reinterpret_cast<u32x4*>(X + (size_t)GQ * FRAMES); // [2][NB * NB][2][64]
			float* condL = reinterpret_cast<float*>(ops0 + 2 * OPS_ONE);    // [FRAMES]
			int par = 0; // sub-mat-muls so far (workgroup-uniform)
			const int tid = threadIdx.x;
			const int f = tid & (FRAMES - 1), cq = tid >> 7;
			const int lane = tid & 63, wave = tid >> 6;
			const int j = lane & 15, q = lane >> 4, tf = 16 * wave + j;
			const int sidx = blockIdx.x;
			const int slot = a.slots ? a.slots[sidx] : a.slot0 + sidx;
			const int row = a.slots ? a.rows[sidx] : a.row0 + sidx;
			float* st = a.state + (size_t)slot * (size_t)a.stateF4 * 4;
			f32x4* stq = reinterpret_cast<f32x4*>(st);
			int* header = reinterpret_cast<int*>(st);
			if (cq == 0) condL[f] = (f < n) ? in[(size_t)row * inStride + f] : 0.0f;
			for (int g = cq; g < GQ; g += 4) X[g * FRAMES + f] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
			__syncthreads();
			const float cond = condL[tf];
			const float* __restrict__ w = a.w;
			f32x4 head[HB][NB]; // the lane's head accumulator: group 16 h + 4 rb + q of frame tf (:772 headArray.SetZero())
#pragma unroll
			for (int h = 0; h < HB; h++)
#pragma unroll
				for (int rb = 0; rb < NB; rb++) head[h][rb] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

			// acc[rb] (rows 64 oh + 16 rb ..) += M[those rows][columns of half ih] * b[kb]; M row-major [cout x cin] at w[off]
			auto subMatMul = [&](int off, int cout, int cin, int oh, int ih, const u32x4 (&b)[NB], sp::f32x4 (&acc)[NB]) {
				const int co = min(64, cout - 64 * oh), ci = min(64, cin - 64 * ih);
				if (co <= 0 || ci <= 0) return; // (workgroup-uniform)
				const int nbo = (co + 15) / 16, nbk = (ci + 15) / 16;
				u32x4* ops = ops0 + (par++ & 1) * OPS_ONE;
				const MatRef m = { off + 64 * oh * cin + 64 * ih, co, ci, nbo, nbk, cin };
				StagedRegs<NB> r;
				StageLoad<NB>(r, w, m);
				StageCommit<NB>(ops, r, m);
				LdsBarrier(); // staged; and every wave is done with the mat-mul before the last, which read the OTHER buffer's predecessor
				MatMul<NB>(ops, nbo, nbk, lane, b, acc);
			};

			for (int li = 0; li < a.numLayers; li++)
			{
				const WnPrewarmLayer L = a.layers[li];
				if (L.kind == 0)
				{
					const int cin = L.cin; // == cout
					const int Gl = (cin + 3) / 4, nh = (cin + 63) / 64;
					if (L.rechannel >= 0)
					{
						if (li == 0 && L.rech_in == 1)
						{
#pragma unroll
							for (int oh = 0; oh < HB; oh++)
#pragma unroll
								for (int rb = 0; rb < NB; rb++)
								{
									const int g = 16 * oh + 4 * rb + q;
									if (g < Gl) X[g * FRAMES + tf] = Load4(w, L.rechannel, 4 * g, cin) * cond;
								}
						}
						else
						{
							// from the previous array's output (in X): every lane reads and writes its own frame only
							const int Gin = (L.rech_in + 3) / 4, nhi = (L.rech_in + 63) / 64;
							__syncthreads();
							u32x4 bq[HB][NB];
#pragma unroll
							for (int ih = 0; ih < HB; ih++)
#pragma unroll
								for (int kb = 0; kb < NB; kb++)
								{
									const int g = 16 * ih + 4 * kb + q;
									bq[ih][kb] = (g < Gin) ? sp::SplitQuad(X[g * FRAMES + tf]) : u32x4{ 0, 0, 0, 0 };
								}
#pragma unroll
							for (int oh = 0; oh < HB; oh++)
							{
								if (oh >= nh) break;
								sp::f32x4 acc[NB];
#pragma unroll
								for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
								for (int ih = 0; ih < HB; ih++)
									if (ih < nhi) subMatMul(L.rechannel, cin, L.rech_in, oh, ih, bq[ih], acc);
#pragma unroll
								for (int rb = 0; rb < NB; rb++)
								{
									const int g = 16 * oh + 4 * rb + q;
									if (g < Gl) X[g * FRAMES + tf] = acc[rb];
								}
							}
							for (int g = Gl + cq; g < GQ; g += 4) X[g * FRAMES + f] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; // (a narrower array behind a wider one)
						}
					}
					__syncthreads(); // X of every frame is complete
					const int R = a.ringFrames[L.ring_id], G = a.ringG[L.ring_id], roff = a.ringOffF4[L.ring_id];
					const int pos0 = header[L.ring_id];
					{
						int p = pos0 + f;
						if (p >= R) p -= R;
						if (f < n && f >= n - (R - FRAMES))
							for (int g = cq; g < Gl; g += 4) stq[RingQuad(roff, G, p, g)] = X[g * FRAMES + f];
					}
					const int K = L.ksize;
					u32x4 zq[HB][NB];
#pragma unroll
					for (int oh = 0; oh < HB; oh++)
					{
#pragma unroll
						for (int rb = 0; rb < NB; rb++) zq[oh][rb] = u32x4{ 0, 0, 0, 0 };
						if (oh >= nh) continue;
						sp::f32x4 acc[NB];
#pragma unroll
						for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						for (int k = 0; k < K; k++)
						{
							const int src = tf - L.dilation * (K - 1 - k); // tap k reads the frame d (K-1-k) back: X inside the block, the ring before it
							int p = pos0 + src;
							if (p < 0) p += R;
#pragma unroll
							for (int ih = 0; ih < HB; ih++)
							{
								if (ih >= nh) break;
								u32x4 b[NB];
#pragma unroll
								for (int kb = 0; kb < NB; kb++)
								{
									const int g = 16 * ih + 4 * kb + q;
									b[kb] = (g < Gl) ? sp::SplitQuad(src >= 0 ? X[g * FRAMES + src] : stq[RingQuad(roff, G, p, g)]) : u32x4{ 0, 0, 0, 0 };
								}
								subMatMul(L.wconv + k * cin * cin, cin, cin, oh, ih, b, acc);
							}
						}
						// bias + mix-in, activation, head accumulate (:288-289, :471-482): in the lanes that hold the results
#pragma unroll
						for (int rb = 0; rb < NB; rb++)
						{
							const int g = 16 * oh + 4 * rb + q;
							if (g < Gl)
							{
								const f32x4 bc = Load4(w, L.bconv, 4 * g, cin), wm = Load4(w, L.wmix, 4 * g, cin);
								f32x4 zv;
								zv.x = (4 * g + 0 < cin) ? Activate(acc[rb].x + bc.x + wm.x * cond, L.act) : 0.0f;
								zv.y = (4 * g + 1 < cin) ? Activate(acc[rb].y + bc.y + wm.y * cond, L.act) : 0.0f;
								zv.z = (4 * g + 2 < cin) ? Activate(acc[rb].z + bc.z + wm.z * cond, L.act) : 0.0f;
								zv.w = (4 * g + 3 < cin) ? Activate(acc[rb].w + bc.w + wm.w * cond, L.act) : 0.0f;
								head[oh][rb] += zv;
								zq[oh][rb] = sp::SplitQuad(sp::f32x4{ zv.x, zv.y, zv.z, zv.w });
							}
						}
					}
					// 1x1 + bias + residual (:486-491): every wave is past its taps (the barriers of the 1x1's own stages), X may change
#pragma unroll
					for (int oh = 0; oh < HB; oh++)
					{
						if (oh >= nh) break;
						sp::f32x4 acc[NB];
#pragma unroll
						for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
						for (int ih = 0; ih < HB; ih++)
							if (ih < nh) subMatMul(L.w1, cin, cin, oh, ih, zq[ih], acc);
#pragma unroll
						for (int rb = 0; rb < NB; rb++)
						{
							const int g = 16 * oh + 4 * rb + q;
							if (g < Gl)
							{
								const f32x4 b1 = Load4(w, L.b1, 4 * g, cin);
								X[g * FRAMES + tf] += f32x4{ acc[rb].x + b1.x, acc[rb].y + b1.y, acc[rb].z + b1.z, acc[rb].w + b1.w };
							}
						}
					}
				}
				else
				{
					// head rechannel (:658-660), dense (K = 1): the next array's head accumulator (:785-789) or, for the last array, the output
					const bool last = (li == a.numLayers - 1);
					const int Gin = (L.cin + 3) / 4, Gout = (L.cout + 3) / 4, nhi = (L.cin + 63) / 64, nho = (L.cout + 63) / 64;
					u32x4 bq[HB][NB];
#pragma unroll
					for (int ih = 0; ih < HB; ih++)
#pragma unroll
						for (int kb = 0; kb < NB; kb++)
							bq[ih][kb] = (16 * ih + 4 * kb + q < Gin) ? sp::SplitQuad(sp::f32x4{ head[ih][kb].x, head[ih][kb].y, head[ih][kb].z, head[ih][kb].w }) : u32x4{ 0, 0, 0, 0 };
#pragma unroll
					for (int oh = 0; oh < HB; oh++)
					{
						sp::f32x4 acc[NB];
#pragma unroll
						for (int rb = 0; rb < NB; rb++) acc[rb] = sp::f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
						if (oh < nho)
						{
#pragma unroll
							for (int ih = 0; ih < HB; ih++)
								if (ih < nhi) subMatMul(L.wconv, L.cout, L.cin, oh, ih, bq[ih], acc);
						}
						if (last)
						{
							if (oh == 0 && q == 0 && tf < n) out[(size_t)row * outStride + tf] = a.headScale * (acc[0].x + ((L.bconv >= 0) ? w[L.bconv] : 0.0f)); // :793-798
						}
						else
						{
#pragma unroll
							for (int rb = 0; rb < NB; rb++)
							{
								const int g = 16 * oh + 4 * rb + q;
								f32x4 v = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
								if (oh < nho && g < Gout)
								{
									const f32x4 bb = (L.bconv >= 0) ? Load4(w, L.bconv, 4 * g, L.cout) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
									v = f32x4{ acc[rb].x + bb.x, acc[rb].y + bb.y, acc[rb].z + bb.z, acc[rb].w + bb.w };
									if (4 * g + 1 >= L.cout) v.y = 0.0f;
									if (4 * g + 2 >= L.cout) v.z = 0.0f;
									if (4 * g + 3 >= L.cout) v.w = 0.0f;
								}
								head[oh][rb] = v;
							}
						}
					}
				}
			}
			__syncthreads();
			if (tid < a.nrings)
			{
				const int R = a.ringFrames[tid];
				int p = header[tid] + n;
				if (p >= R) p -= R;
				header[tid] = p;
			}
		}
	}

	hipError_t LaunchWaveNetGeneric(const WnPrewarmLayer* layers, int numLayers, const float* weights, const int* ringOffF4, const int* ringFrames,
		const int* ringG, int nrings, int stateF4, int maxChannels, float headScale, float* state, const int* slots, const int* rows, int numStreams,
		int slot0, int row0, const float* in, float* out, long inStride, long outStride, int n, hipStream_t stream)
	{
		if (numStreams <= 0 || n <= 0) return hipSuccess;
		if (n > WN_MAX_FRAMES || maxChannels > WN_GENERIC_MAX_CHANNELS) return hipErrorInvalidValue;
		gn::Args a;
		a.layers = layers;
		a.numLayers = numLayers;
		a.w = weights;
		a.ringOffF4 = ringOffF4;
		a.ringFrames = ringFrames;
		a.ringG = ringG;
		a.nrings = nrings;
		a.stateF4 = stateF4;
		a.maxC = maxChannels;
		a.headScale = headScale;
		a.state = state;
		a.slots = slots;
		a.rows = rows;
		a.slot0 = slot0;
		a.row0 = row0;
		if (maxChannels > 64)
		{
			// 65 .. 128 channels: X + two 64 x 64 operand buffers + the condition row (128 channels: 64 + 64 KB)
			const size_t wideBytes = (size_t)((maxChannels + 3) / 4) * gn::FRAMES * 16 + (size_t)2 * 4 * 4 * 2 * 64 * 16 + gn::FRAMES * sizeof(float);
			static PerDeviceOnce attrWide;
			(void)attrWide.Run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gn::WaveNetWideKernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
			hipLaunchKernelGGL((gn::WaveNetWideKernel<2>), dim3((unsigned)numStreams), dim3(gn::NTHREADS), wideBytes, stream, a, in, out, inStride, outStride, n);
			return hipGetLastError();
		}
		// LDS: two [G][128] float4 arrays + the split A operands of two matrices ([nb x nb][hi, lo][64] x 16 B each) + the condition row
		// (64 channels: 64 + 64 KB, 48 channels: 48 + 36 KB, 32 channels: 32 + 16 KB -> three workgroups per CU)
		const int nb = (maxChannels + 15) / 16, gq = (maxChannels + 3) / 4;
		const size_t ldsBytes = (size_t)2 * gq * gn::FRAMES * 16 + (size_t)2 * nb * nb * 2 * 64 * 16 + gn::FRAMES * sizeof(float);
		const int numCUs = CurrentDeviceCUs();
		static PerDeviceOnce attr;
		(void)attr.Run([] {
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn::WaveNetGenericKernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn::WaveNetGenericKernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn::WaveNetGenericKernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn::WaveNetGenericKernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			return hipFuncSetAttribute(reinterpret_cast<const void*>(&gn::WaveNetGenericKernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		});
		// NB = 16-channel blocks per matrix side
		const dim3 grid((unsigned)numStreams), block(gn::NTHREADS);
		if (nb <= 1) hipLaunchKernelGGL((gn::WaveNetGenericKernel<1, 4>), grid, block, ldsBytes, stream, a, in, out, inStride, outStride, n);
		else if (nb == 2 && numStreams > numCUs) hipLaunchKernelGGL((gn::WaveNetGenericKernel<2, 4>), grid, block, ldsBytes, stream, a, in, out, inStride, outStride, n);
		else if (nb == 2) hipLaunchKernelGGL((gn::WaveNetGenericKernel<2, 2>), grid, block, ldsBytes, stream, a, in, out, inStride, outStride, n);
		else if (nb == 3) hipLaunchKernelGGL((gn::WaveNetGenericKernel<3, 2>), grid, block, ldsBytes, stream, a, in, out, inStride, outStride, n);
		else hipLaunchKernelGGL((gn::WaveNetGenericKernel<4, 2>), grid, block, ldsBytes, stream, a, in, out, inStride, outStride, n);
		return hipGetLastError();
	}
}
