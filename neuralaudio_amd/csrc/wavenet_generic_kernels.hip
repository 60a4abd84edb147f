// wavenet_generic_kernels.hip -- the runtime-shaped WaveNet block kernel: any channel count up to 64, any kernel sizes / dilations /
// layer counts, dense heads -- what the reference's dynamic engine accepts beyond the official architectures
// (NeuralAudio/WaveNetDynamic.h:229-254,445-468, InternalModel.h:177-248; same arithmetic as WaveNet.h:768-799, 632-661, 462-494).
//
// It is the counterpart of the runtime-shaped LSTM / GRU kernels: slow next to the shaped kernels (every weight is a scalar load per
// use, no matrix pipe) but it runs what they cannot -- layer arrays wider than 16 channels -- instead of a load error.  It walks the
// natural-layout tensor table the prewarm kernel uses (WnPrewarmLayer: offsets into the flat weight array in the reference's order,
// WaveNet.h:700-719) and keeps the frame kernel's stream-state format (f32 quads, tile layout), so prewarm / reset are shared.
//
//   workgroup = one stream, thread = one frame of the 128-frame block;
//   LDS: x[C][128] layer input (updated in place), z[C][128] accumulators / activations, head[C][128] head accumulator, t[C][128] the
//   tap being accumulated, and a [C][C] weight buffer (one tap's matrix, then the 1x1: the inner loops read four weights per
//   broadcast ds_read_b128 instead of a scalar load per term);
//   a layer = publish x to its ring -> per tap: stage its weight matrix, gather the tap's input column of every frame once (LDS for
//   in-block frames, the layer's HBM ring for earlier ones), accumulate all outputs -> activation, head += z -> 1x1 + residual into x.
#include <algorithm>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "wavenet_dev.h"
#include "wavenet_launch.h"

namespace na
{
	namespace gn
	{
		constexpr int FRAMES = WN_MAX_FRAMES;
		typedef float f32x4 __attribute__((ext_vector_type(4)));

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

		// float index of (ring position p, channel c) in the tile layout with G channel groups: float4 ((p >> 4) G + c / 4) 16 + (p & 15)
		__device__ __forceinline__ size_t RingElem(int ringOffF4, int G, int p, int c)
		{
			return ((size_t)ringOffF4 + (size_t)(((p >> 4) * G + (c >> 2)) * 16 + (p & 15))) * 4 + (size_t)(c & 3);
		}

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

		__global__ void __launch_bounds__(FRAMES) WaveNetGenericKernel(const Args a, const float* __restrict__ in, float* __restrict__ out, long inStride,
			long outStride, int n)
		{
			extern __shared__ __attribute__((aligned(16))) float lds[];
			const int C = a.maxC;
			float* x = lds;                  // [C][FRAMES]
			float* z = x + (size_t)C * FRAMES;
			float* head = z + (size_t)C * FRAMES;
			float* t = head + (size_t)C * FRAMES;  // [C][FRAMES]: the tap being accumulated, gathered per frame
			float* wl = t + (size_t)C * FRAMES;    // [C * C]: one tap's weight matrix / the 1x1 matrix
			const int f = threadIdx.x;
			const int sidx = blockIdx.x;
			const int slot = a.slots ? a.slots[sidx] : a.slot0 + sidx;
			const int row = a.slots ? a.rows[sidx] : a.row0 + sidx;
			float* st = a.state + (size_t)slot * (size_t)a.stateF4 * 4;
			int* header = reinterpret_cast<int*>(st);
			const float cond = (f < n) ? in[(size_t)row * inStride + f] : 0.0f; // WaveNet.h:770 (input -> condition)
			for (int c = 0; c < C; c++)
			{
				x[c * FRAMES + f] = 0.0f;
				z[c * FRAMES + f] = 0.0f;
				head[c * FRAMES + f] = 0.0f; // :772 headArray.SetZero()
			}
			__syncthreads();

			const float* __restrict__ w = a.w;
			for (int li = 0; li < a.numLayers; li++)
			{
				const WnPrewarmLayer L = a.layers[li];
				if (L.kind == 0)
				{
					const int cin = L.cin; // == cout
					if (L.rechannel >= 0)
					{
						// rechannel (:637): array 0 from the condition (input_size == 1), later arrays from the previous array's output (in x)
						for (int o = 0; o < cin; o++)
						{
							float v = 0.0f;
							if (L.rech_in == 1 && li == 0) v = w[L.rechannel + o] * cond;
							else
								for (int c = 0; c < L.rech_in; c++) v += w[L.rechannel + o * L.rech_in + c] * x[c * FRAMES + f];
							z[o * FRAMES + f] = v;
						}
						for (int o = 0; o < cin; o++) x[o * FRAMES + f] = z[o * FRAMES + f]; // own frame only: no barrier needed in between
					}
					// the layer input of this block -> its ring (history for LATER blocks: only the last R - FRAMES frames can be read back)
					const int R = a.ringFrames[L.ring_id], G = a.ringG[L.ring_id], roff = a.ringOffF4[L.ring_id];
					const int pos0 = header[L.ring_id];
					{
						int p = pos0 + f;
						if (p >= R) p -= R;
						if (f < n && f >= n - (R - FRAMES))
							for (int c = 0; c < cin; c++) st[RingElem(roff, G, p, c)] = x[c * FRAMES + f];
					}
					// dilated conv + bias + mix-in (:139-290, :288-289, :471), tap by tap: the tap's weight matrix goes to LDS ([out][in], from
					// [(o cin + c) K + k]), every thread gathers the tap's input column of ITS frame once (LDS for in-block frames, the ring
					// for earlier ones) into t, then accumulates all outputs in its own column of z
					const int K = L.ksize;
					for (int o = 0; o < cin; o++) z[o * FRAMES + f] = w[L.bconv + o] + w[L.wmix + o] * cond;
					for (int k = 0; k < K; k++)
					{
						__syncthreads(); // k == 0: every thread's x is complete; k > 0: the previous tap's weights are no longer read
						for (int i = f; i < cin * cin; i += FRAMES) wl[i] = w[L.wconv + (size_t)i * K + k];
						const int off = f - L.dilation * (K - 1 - k); // tap k reads the frame d (K-1-k) back
						if (off >= 0)
							for (int c = 0; c < cin; c++) t[c * FRAMES + f] = x[c * FRAMES + off];
						else
						{
							int p = pos0 + off; // off >= -(R - FRAMES): one wrap
							if (p < 0) p += R;
							for (int c = 0; c < cin; c++) t[c * FRAMES + f] = st[RingElem(roff, G, p, c)];
						}
						__syncthreads(); // the tap's weights are staged (t is read by its own thread only)
						for (int o = 0; o < cin; o++)
						{
							const float* wr = wl + o * cin;
							float acc = z[o * FRAMES + f];
							int c = 0;
							if ((cin & 3) == 0) // rows of the weight buffer are 16-byte aligned: four weights per (broadcast) LDS read
								for (; c < cin; c += 4)
								{
									const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + c);
									acc = __builtin_fmaf(w4.x, t[c * FRAMES + f], acc);
									acc = __builtin_fmaf(w4.y, t[(c + 1) * FRAMES + f], acc);
									acc = __builtin_fmaf(w4.z, t[(c + 2) * FRAMES + f], acc);
									acc = __builtin_fmaf(w4.w, t[(c + 3) * FRAMES + f], acc);
								}
							for (; c < cin; c++) acc = __builtin_fmaf(wr[c], t[c * FRAMES + f], acc);
							z[o * FRAMES + f] = acc;
						}
					}
					// activation (:473-480), head accumulate (:482)
					for (int o = 0; o < cin; o++)
					{
						const float zv = Activate(z[o * FRAMES + f], L.act);
						z[o * FRAMES + f] = zv;
						head[o * FRAMES + f] += zv;
					}
					__syncthreads(); // every tap gathered: x may be overwritten; the weight buffer is free
					// 1x1 matrix -> LDS (natural [out][in]); 1x1 + bias + residual (:486-491); the last layer's output feeds the next array's
					// rechannel (or nothing)
					for (int i = f; i < cin * cin; i += FRAMES) wl[i] = w[L.w1 + i];
					__syncthreads();
					for (int o = 0; o < cin; o++)
					{
						const float* wr = wl + o * cin;
						float y = w[L.b1 + o] + x[o * FRAMES + f];
						int c = 0;
						if ((cin & 3) == 0)
							for (; c < cin; c += 4)
							{
								const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + c);
								y = __builtin_fmaf(w4.x, z[c * FRAMES + f], y);
								y = __builtin_fmaf(w4.y, z[(c + 1) * FRAMES + f], y);
								y = __builtin_fmaf(w4.z, z[(c + 2) * FRAMES + f], y);
								y = __builtin_fmaf(w4.w, z[(c + 3) * FRAMES + f], y);
							}
						for (; c < cin; c++) y = __builtin_fmaf(wr[c], z[c * FRAMES + f], y);
						x[o * FRAMES + f] = y;
					}
					__syncthreads(); // the weight buffer is rewritten by the next layer's staging
					// (own-frame accesses only from here to the next publish: no barrier)
				}
				else
				{
					// head rechannel (K = 1, :658-660): becomes the next array's head accumulator (:785-789) or, for the last array, the output
					const bool last = (li == a.numLayers - 1);
					for (int o = 0; o < L.cout; o++)
					{
						float acc = (L.bconv >= 0) ? w[L.bconv + o] : 0.0f;
						for (int c = 0; c < L.cin; c++) acc += w[L.wconv + o * L.cin + c] * head[c * FRAMES + f];
						z[o * FRAMES + f] = acc;
					}
					if (last)
					{
						if (f < n) out[(size_t)row * outStride + f] = a.headScale * z[f]; // :793-798: head channel 0
					}
					else
					{
						for (int o = 0; o < C; o++) head[o * FRAMES + f] = (o < L.cout) ? z[o * FRAMES + f] : 0.0f;
					}
				}
			}
			__syncthreads();
			// advance every ring cursor by n (ChannelHistoryBuffer::AdvanceFrames, WaveNet.h:59-65, as a true modulo ring)
			if (f < a.nrings)
			{
				const int R = a.ringFrames[f];
				int p = header[f] + n;
				if (p >= R) p -= R;
				header[f] = p;
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
		// LDS: four [C][128] float arrays + one [C][C] weight matrix (C = 64: 144 KB)
		const size_t ldsBytes = ((size_t)4 * maxChannels * gn::FRAMES + (size_t)maxChannels * maxChannels) * sizeof(float);
		static bool attrSet = false;
		if (!attrSet)
		{
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn::WaveNetGenericKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			attrSet = true;
		}
		hipLaunchKernelGGL(gn::WaveNetGenericKernel, dim3((unsigned)numStreams), dim3(gn::FRAMES), ldsBytes, stream, a, in, out, inStride, outStride, n);
		return hipGetLastError();
	}
}
