// wavenet_plan.cpp -- WaveNetDesc -> stage program + packed MFMA operand tables.
//
// Weight walk order follows the reference's SetWeights chain exactly:
//   WaveNetModelT::SetWeights (WaveNet.h:700-719) -> per array WaveNetLayerArrayT::SetWeights (:570-580):
//   rechannel [out][in]; per layer (:420-425) conv [out][in][k] + bias, input mix-in [out][cond],
//   1x1 [out][in] + bias; head conv [out][in][k] (+ bias); last float = head scale.
#include "wavenet_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <sstream>
#include <stdexcept>

namespace na
{
	static int CeilDiv(int a, int b) { return (a + b - 1) / b; }

	size_t WaveNetDesc::ExpectedNumWeights() const
	{
		size_t n = 0;
		for (const auto& a : arrays)
		{
			const size_t c = (size_t)a.channels;
			n += c * a.inputSize;
			for (size_t l = 0; l < a.kernelSizes.size(); l++)
			{
				n += c * c * a.kernelSizes[l] + c; // conv + bias
				n += c * a.conditionSize;         // mix-in
				n += c * c + c;                   // 1x1 + bias
			}
			n += (size_t)a.headSize * c * a.headKernelSize + (a.hasHeadBias ? a.headSize : 0);
		}
		return n + 1; // head scale
	}

	int WaveNetDesc::ReceptiveFieldSize() const
	{
		int rf = 0;
		for (const auto& a : arrays)
		{
			for (size_t l = 0; l < a.kernelSizes.size(); l++) rf += (a.kernelSizes[l] - 1) * a.dilations[l];
			rf += (a.headKernelSize - 1) * a.headDilation;
		}
		return rf;
	}

	// Everything that can be wrong with a WaveNet description, checked at LOAD time (the reference throws "Wrong number of weights"
	// from SetWeights inside CreateFromJson, WaveNet.h:704-709) together with the limits of the gfx950 kernels.
	void ValidateWaveNetDesc(const WaveNetDesc& desc)
	{
		if (desc.arrays.empty()) throw std::runtime_error("WaveNet without layer arrays");
		const int numArrays = (int)desc.arrays.size();
		int rings = 0;
		for (int a = 0; a < numArrays; a++)
		{
			const WnArrayCfg& cfg = desc.arrays[a];
			if (cfg.channels < 1 || cfg.headSize < 1 || cfg.inputSize < 1) throw std::runtime_error("WaveNet layer array with a zero-sized dimension");
			if (cfg.channels > WN_GENERIC_MAX_CHANNELS || cfg.headSize > WN_GENERIC_MAX_CHANNELS || cfg.inputSize > WN_GENERIC_MAX_CHANNELS)
				throw std::runtime_error("WaveNet channels > 128 are not supported");
			if (cfg.headKernelSize > 1 && (cfg.channels > 64 || cfg.headSize > 64))
				throw std::runtime_error("WaveNet: a conv head on a layer array wider than 64 channels is not supported");
			if (cfg.conditionSize != 1) throw std::runtime_error("WaveNet condition_size != 1 is not supported");
			if (cfg.kernelSizes.size() != cfg.dilations.size() || cfg.kernelSizes.empty())
				throw std::runtime_error("WaveNet kernel_sizes/dilations mismatch");
			for (size_t l = 0; l < cfg.kernelSizes.size(); l++)
				if (cfg.kernelSizes[l] < 1 || cfg.dilations[l] < 1) throw std::runtime_error("WaveNet kernel_size / dilation must be >= 1");
			if (cfg.headKernelSize < 1 || cfg.headDilation < 1) throw std::runtime_error("WaveNet head kernel_size / dilation must be >= 1");
			rings += (int)cfg.kernelSizes.size() + (cfg.headKernelSize > 1 ? 1 : 0);
			if (a > 0)
			{
				// head accumulation continues in place in the previous array's head outputs (WaveNet.h:785-789)
				if (desc.arrays[a - 1].headSize != cfg.channels || cfg.inputSize != desc.arrays[a - 1].channels)
					throw std::runtime_error("WaveNet layer arrays do not chain (head_size/input_size mismatch)");
				if (desc.arrays[a - 1].headKernelSize != 1)
					throw std::runtime_error("WaveNet: head kernel > 1 is only supported on the last layer array");
			}
		}
		if (desc.arrays[0].inputSize != 1) throw std::runtime_error("WaveNet first layer array must have input_size 1");
		if (rings > WN_MAX_RINGS) throw std::runtime_error("WaveNet has more than 64 conv layers (unsupported)");
		const size_t expected = desc.ExpectedNumWeights();
		if (expected != desc.weights.size())
		{
			std::stringstream str;
			str << "Wrong number of weights. Expected " << expected << " but got " << desc.weights.size();
			throw std::runtime_error(str.str());
		}
	}


	// ---- f16-split kernel: A-operand image -------------------------------------------------------------------------------
	// float -> IEEE binary16 bit pattern, round to nearest even, subnormals kept (what v_cvt_pk_f16_f32 does for the activations)
	static uint16_t FloatToHalfBits(float f)
	{
		uint32_t x;
		memcpy(&x, &f, 4);
		const uint32_t sign = (x >> 16) & 0x8000u;
		x &= 0x7fffffffu;
		if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
		if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // rounds to >= 65520 -> inf
		if (x < 0x33000001u) return (uint16_t)sign;              // < 2^-25 (or exactly, ties to even 0)
		int e = (int)(x >> 23) - 127;
		uint32_t m = (x & 0x7fffffu) | 0x800000u;
		int shift = (e < -14) ? (13 + (-14 - e)) : 13; // subnormal: shift further
		uint32_t half = m >> shift;
		const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
		if (rem > halfway || (rem == halfway && (half & 1u))) half++;
		if (e < -14) return (uint16_t)(sign | half); // subnormal (a carry into the exponent field is the right bit pattern)
		half += (uint32_t)(e + 15 - 1) << 10;       // hidden bit adds 1 to the exponent field
		return (uint16_t)(sign | half);
	}

	static float HalfBitsToFloat(uint16_t h)
	{
		const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
		const int e = (h >> 10) & 0x1f;
		const uint32_t m = h & 0x3ffu;
		float v;
		if (e == 0) v = std::ldexp((float)m, -24);
		else if (e == 31) v = m ? NAN : INFINITY;
		else v = std::ldexp((float)(m | 0x400u), e - 25);
		return sign ? -v : v;
	}

	// lane mode of a channel count: channel groups per 16-frame tile, rounded up to a power of two
	static int LaneMode(int channels) { return channels <= 4 ? 1 : (channels <= 8 ? 2 : 4); }

	namespace
	{
		struct Builder
		{
			const WaveNetDesc& desc;
			WaveNetPlan plan;
			size_t cursor = 0; // into desc.weights

			int pack = 1; // > 1: `desc` is a packed virtual model (PackWaveNetDesc) of `pack` streams
			bool exactRings = false; // rings of the f16-split kernels' state format (see AddRing)
			bool compactOk = false;  // ... and every layer has K <= 3 with dense heads: short histories get compact rings

			explicit Builder(const WaveNetDesc& d, int packStreams = 1, bool exact = false) : desc(d), pack(packStreams), exactRings(exact) {}

			int Take(size_t n)
			{
				const size_t off = cursor;
				cursor += n;
				return (int)off;
			}

			float W(int off) const { return desc.weights[(size_t)off]; }

			// reserve `nF4` float4 slots in wpack, zero-filled; returns float4 index
			int AllocF4(int nF4)
			{
				const int off = (int)(plan.wpack.size() / 4);
				plan.wpack.resize(plan.wpack.size() + (size_t)nF4 * 4, 0.0f);
				return off;
			}

			// A layer's input history: a true modulo ring.  A block of n <= 128 frames reads the `history` = (K - 1) d frames in front of it and
			// writes its own; with roundup16(history) + 128 frames the two never touch the same position inside one launch, whatever n.
			// Rings of the f16-split kernels' state format (`exactRings`) with a dilation of at least a whole block are EXACTLY `history`
			// frames long instead: the only position a block both reads and writes is then the one frame f reads for its most-shifted tap
			// (p + f - history = p + f mod R) and overwrites with its own value -- the same lane of the same wave, load before store in
			// program order (every other tap of frame f lies d >= 128 > n frames behind a position this block writes).  Not for the first
			// layer of an array: its ring is written by the rechannel / link stage BEFORE the layer's history is requested.  A1 Standard:
			// 36 KB less state per stream (315 -> 279 KB): 896 streams' address footprint then fits the 256 MB Infinity Cache.
			int AddRing(int channels, int history, int dilation, bool firstOfArray = true)
			{
				WnRingInfo r;
				r.channels = channels;
				r.G = CeilDiv(channels, 4);
				// (history >= 2 blocks: the stage interpreter keeps WnRingKeep(R) = R - 128 frames of a block, which is the whole block only then --
			// a K = 2, d = 128 .. 240 ring of exactly its history would lose frames; such a layer keeps the roomy ring)
			const bool exact = exactRings && !firstOfArray && dilation >= WN_MAX_FRAMES && history >= 2 * WN_MAX_FRAMES && history % WN_TILE == 0;
				// Short histories (A1's d = 1 .. 16 layers: 16 or 32 frames) in rings of three times their length instead of + 128: for the
				// block lengths the host then restricts itself to, the frames a block reads and the ones it writes never share a position
				// (wavenet_dev.h WnRingKeep).  A1 Standard: another 43 KB per stream (279 -> 236 KB): 1024 streams = 242 MB.
				const int H = CeilDiv(history, WN_TILE) * WN_TILE;
				const bool compact = compactOk && history > 0 && H <= WN_COMPACT_MAX_HISTORY;
				if (compact) plan.compactRings = true;
				r.frames = exact ? history : (compact ? 3 * H : H + WN_MAX_FRAMES);
				r.offF4 = plan.stateF4;
				plan.stateF4 += (r.frames / WN_TILE) * r.G * WN_TILE; // tiles * G * 16 float4
				plan.rings.push_back(r);
				if ((int)plan.rings.size() > WN_MAX_RINGS) throw std::runtime_error("WaveNet has more than 64 conv layers (unsupported)");
				return (int)plan.rings.size() - 1;
			}

			int AllocPk(int nFloats)
			{
				const int off = (int)plan.wpk.size();
				plan.wpk.resize(plan.wpk.size() + (size_t)((nFloats + 15) & ~15), 0.0f); // 64-byte granules: s_load_dwordx16 friendly
				return off;
			}

			// head conv of the frame kernel (plain FMAs over scalar-loaded weights): [tap][in c][out o], channel counts padded; flat source [(o*cin + c)*K + tap]
			int PackConvPk(int wOff, int cin, int cout, int ksize, int CPin, int CPout)
			{
				const int off = AllocPk(ksize * CPin * CPout);
				for (int k = 0; k < ksize; k++)
					for (int c = 0; c < cin; c++)
						for (int o = 0; o < cout; o++) plan.wpk[(size_t)off + ((size_t)k * CPin + c) * CPout + o] = W(wOff + (o * cin + c) * ksize + k);
				return off;
			}

			// head dense of the frame kernel: [in c][out o] padded; flat source [o*cin + c]
			int PackDensePk(int wOff, int cin, int cout, int CPin, int CPout)
			{
				const int off = AllocPk(CPin * CPout);
				for (int c = 0; c < cin; c++)
					for (int o = 0; o < cout; o++) plan.wpk[(size_t)off + (size_t)c * CPout + o] = W(wOff + o * cin + c);
				return off;
			}

			// A operands of v_mfma_f32_4x4x1_16b_f32 for the frame kernel.  The kernel issues the MFMA with CBSZ=4 / ABID=c: the 4 lanes of
			// block c supply the A operand of ALL 16 blocks, so ONE register per (tap, out group) carries the weights of 16 input channels:
			// lane l holds W[4*og + (l & 3)][c = l >> 2].  Image: [tap][lane 0..63][og 0..G-1] floats (a lane reads its G floats as one
			// ds_read_b128/b64/b32).  Flat source conv [(o*cin + c)*K + tap], dense [o*cin + c].
			int PackConvA4(int wOff, int cin, int cout, int ksize, int G)
			{
				const int off = AllocPk(ksize * 64 * G);
				for (int k = 0; k < ksize; k++)
					for (int lane = 0; lane < 64; lane++)
						for (int og = 0; og < G; og++)
						{
							const int o = 4 * og + (lane & 3), c = lane >> 2;
							if (o < cout && c < cin) plan.wpk[(size_t)off + ((size_t)k * 64 + lane) * G + og] = W(wOff + (o * cin + c) * ksize + k);
						}
				return off;
			}

			int PackDenseA4(int wOff, int cin, int cout, int G)
			{
				const int off = AllocPk(64 * G);
				for (int lane = 0; lane < 64; lane++)
					for (int og = 0; og < G; og++)
					{
						const int o = 4 * og + (lane & 3), c = lane >> 2;
						if (o < cout && c < cin) plan.wpk[(size_t)off + (size_t)lane * G + og] = W(wOff + o * cin + c);
					}
				return off;
			}

			void SetVec(const WnStage& st, int slot, int wOff, int n)
			{
				for (int i = 0; i < n && i < 16; i++) plan.wpack[((size_t)st.vec_off + (size_t)slot * 4) * 4 + i] = W(wOff + i);
			}

			static WnStage EmptyStage(int type)
			{
				WnStage st = {};
				st.type = type;
				st.ring_id = -1;
				st.out_ring_id = -1;
				return st;
			}

			void SetRing(WnStage& st, int ringId)
			{
				st.ring_id = ringId;
				st.ring_off = plan.rings[ringId].offF4;
				st.ring_frames = plan.rings[ringId].frames;
			}

			void SetOutRing(WnStage& st, int ringId)
			{
				st.out_ring_id = ringId;
				st.out_ring_off = plan.rings[ringId].offF4;
				st.out_ring_frames = plan.rings[ringId].frames;
				st.out_G = plan.rings[ringId].G;
			}


			// ---- f16-split kernel ------------------------------------------------------------------------------------------
			// One MFMA A operand (v_mfma_f32_16x16x32_f16): lane (row i = lane & 15, k-block q = lane >> 4) holds 8 halfs, the weights
			// that multiply the 8 halfs of lane (column, q)'s B operand = one split quad [h0..h3 | l0..l3].  W*x ~= Wh*(xh + xl) + Wl*xh:
			// the "hi" operand carries Wh in all 8 slots, the "lo" operand Wl in the four h slots (zeros against l).
			// Rows [rowBase, rowBase + cout) x k-blocks [kbBase, kbBase + ceil(cin/4)) receive W(o, c); everything else stays zero, so
			// several tiles share one MFMA through block-diagonal operands (mode Gp: tile slot p owns rows 4*Gp*p.. and k-blocks Gp*p..).
			// Static range proof of the f16-split kernels.  With the input clamped to +-L every value that is ever split
			// (residual stream x, activations z, head accumulator) is bounded by G * L + A; the pairs collected here give the largest L for
			// which all of them stay below half the f16 range.  tanh layers add a bounded amount (|FastMath tanh| <= 1.0081), LeakyReLU layers
			// (|f(v)| <= |v|) multiply: for them G grows with the worst-case row sums of every layer.
			double rangeGain = 1.0, rangeAdd = 0.0;       // |x| <= rangeGain * L + rangeAdd
			double rangeHeadGain = 0.0, rangeHeadAdd = 0.0; // |head accumulator| likewise
			double linearGain = 1.0, linearGainMax = 0.0;   // the rechannel path alone (input -> array inputs)
			std::vector<std::pair<double, double>> rangeCons;
			double weightAbsMax = 0.0;      // largest |w| of the model
			double matrixPeakMin = 1e300;   // smallest max|w| over its (non-zero) weight matrices
			void RangeNote(double G, double A) { rangeCons.push_back({ G, A }); }
			void NoteTensor(int off, size_t count, bool matrix)
			{
				double peak = 0.0;
				for (size_t i = 0; i < count; i++) peak = std::max(peak, std::fabs((double)W(off + (int)i)));
				weightAbsMax = std::max(weightAbsMax, peak);
				if (matrix && peak > 0.0) matrixPeakMin = std::min(matrixPeakMin, peak);
			}

			int NewSplitOps(int count)
			{
				const int first = (int)(plan.wsplit.size() / 512);
				plan.wsplit.resize(plan.wsplit.size() + (size_t)count * 512, 0);
				return first;
			}

			void FillSplitBlock(int opHi, int opLo, int rowBase, int kbBase, int cout, int cin, const std::function<float(int, int)>& w)
			{
				for (int o = 0; o < cout; o++)
					for (int c = 0; c < cin; c++)
					{
						const int row = rowBase + o, q = kbBase + c / 4, r = c % 4;
						if (row > 15 || q > 3) throw std::runtime_error("internal: split operand block out of range");
						const float v = w(o, c);
						const uint16_t hi = FloatToHalfBits(v);
						const uint16_t lo = FloatToHalfBits(v - HalfBitsToFloat(hi));
						const size_t lane = (size_t)q * 16 + row;
						plan.wsplit[(size_t)opHi * 512 + lane * 8 + r] = hi;
						plan.wsplit[(size_t)opHi * 512 + lane * 8 + 4 + r] = hi;
						plan.wsplit[(size_t)opLo * 512 + lane * 8 + r] = lo;
					}
			}

			// the same weights for every tile slot of mode Gp (merged block-diagonal operand pair)
			void FillSplitMerged(int opHi, int Gp, int cout, int cin, const std::function<float(int, int)>& w)
			{
				for (int p = 0; p < 4 / Gp; p++) FillSplitBlock(opHi, opHi + 1, 4 * Gp * p, Gp * p, cout, cin, w);
			}

			// "aux" operand: bias and input mix-in ride in the MFMA too.  The kernel's aux B operand of a frame is the 8 halfs
			// [cond_h, 1, cond_l, 1, cond_h, 0, 0, 0]; against the row [wc_h, w1_h, wc_h, w1_l, wc_l, 0, 0, 0] it contributes
			// wc * cond + w1 to that row (same three-product split, ONE operand).  It sits in the cg = 0 k-block of each tile slot.
			// Packed plans (several streams of a narrow model as the channel groups of one virtual stream, see PackWaveNetDesc): the aux B
			// operand of k-block q carries the condition of the stream that owns channel group q, so row o's weights sit in the first
			// k-block of ITS stream (cpad = padded channels per stream; unpacked: cpad >= cout, i.e. always the cg = 0 k-block).
			// Dense packs (cpad == 2: two streams A, B share a channel group): the B operand of that k-block is
			// [cA_h, 1, cA_l, 1, cA_h, cB_h, cB_l, cB_h] -- the rows of the first stream of the pair keep the usual five slots, the rows of the
			// second one take their ones from slots 1 / 3 and their condition from slots 5 / 6 / 7: the eight k-slots are exactly enough.
			void FillSplitAux(int op, int Gp, int cout, int condOff, int oneOff, int cpad = 1 << 20)
			{
				for (int p = 0; p < 4 / Gp; p++)
					for (int o = 0; o < cout; o++)
					{
						const float wc = condOff >= 0 ? W(condOff + o) : 0.0f, w1 = oneOff >= 0 ? W(oneOff + o) : 0.0f;
						const uint16_t wch = FloatToHalfBits(wc), wcl = FloatToHalfBits(wc - HalfBitsToFloat(wch));
						const uint16_t w1h = FloatToHalfBits(w1), w1l = FloatToHalfBits(w1 - HalfBitsToFloat(w1h));
						const bool halfGroups = cpad == 2;
						const size_t kblock = (size_t)Gp * p + (halfGroups ? (size_t)(o / 4) : (size_t)((o / cpad) * (cpad / 4)));
						const size_t lane = kblock * 16 + (size_t)(4 * Gp * p + o);
						uint16_t* e = &plan.wsplit[(size_t)op * 512 + lane * 8];
						if (halfGroups && ((o / 2) & 1)) { e[1] = w1h; e[3] = w1l; e[5] = wch; e[6] = wch; e[7] = wcl; }
						else { e[0] = wch; e[1] = w1h; e[2] = wch; e[3] = w1l; e[4] = wcl; }
					}
			}

			static WnSplitStage EmptySplit(int type)
			{
				WnSplitStage st = {};
				st.type = type;
				st.ring_off = -1; st.ring_id = -1;
				st.out_ring_off = -1; st.out_ring_id = -1;
				return st;
			}

			void SplitRing(WnSplitStage& st, int ringId)
			{
				st.ring_id = ringId;
				st.ring_off = plan.rings[ringId].offF4;
				st.ring_frames = plan.rings[ringId].frames;
			}

			void SplitOutRing(WnSplitStage& st, int ringId)
			{
				st.out_ring_id = ringId;
				st.out_ring_off = plan.rings[ringId].offF4;
				st.out_ring_frames = plan.rings[ringId].frames;
				st.out_G = plan.rings[ringId].G;
			}

			// second walk over the flat weights (same order as Build): stage program + A-operand image of the f16-split kernel
			void BuildSplit(const std::vector<std::vector<int>>& layerRing, const std::vector<int>& headRing)
			{
				cursor = 0;
				const int numArrays = (int)desc.arrays.size();
				int prevHeadW = -1, prevHeadB = -1;
				for (int a = 0; a < numArrays; a++)
				{
					const WnArrayCfg& cfg = desc.arrays[a];
					const int C = cfg.channels, G = CeilDiv(C, 4), Gp = LaneMode(C);
					const int numLayers = (int)cfg.kernelSizes.size();
					const bool lastArray = (a == numArrays - 1);
					plan.maxG = std::max(plan.maxG, G);
					const int cpad = pack > 1 ? C / pack : (1 << 20); // channels per packed stream (a multiple of 4; 2 in a dense pack)
					const int groupsPerStream = pack > 1 ? cpad / 4 : 4; // (0: two streams per channel group)
					const int rechOff = Take((size_t)C * cfg.inputSize);
					{
						// range bookkeeping for condLimit: |residual stream| <= gain * |cond| + add through this array's rechannel
						double rowMax = 0.0;
						for (int o = 0; o < C; o++)
						{
							double row = 0.0;
							for (int c = 0; c < cfg.inputSize; c++) row += std::fabs((double)W(rechOff + o * cfg.inputSize + c));
							rowMax = std::max(rowMax, row);
						}
						rangeGain = (a == 0 ? 1.0 : rangeGain) * rowMax;
						rangeAdd = (a == 0 ? 0.0 : rangeAdd) * rowMax;
						RangeNote(rangeGain, rangeAdd);
						linearGain = (a == 0 ? 1.0 : linearGain) * rowMax;
						linearGainMax = std::max(linearGainMax, linearGain);
						NoteTensor(rechOff, (size_t)C * cfg.inputSize, true);
					}
					if (a == 0)
					{
						// x = w_re * cond (WaveNet.h:637, input_size == 1): the aux operand with weights (w_re, 0)
						WnSplitStage st = EmptySplit(WN_ST_RECHANNEL_COND);
						st.G = G; st.Gp = Gp;
						st.a_ops = 1;
						st.a_off = NewSplitOps(st.a_ops) * 64;
						FillSplitAux(st.a_off / 64, Gp, C, rechOff, -1, cpad);
						st.reserved = groupsPerStream;
						SplitOutRing(st, layerRing[a][0]);
						st.flags = WN_FLAG_PUBLISH;
						plan.sstages.push_back(st);
					}
					else
					{
						// previous array's head rechannel (K = 1, WaveNet.h:658-660) and this array's rechannel (:637), one tile at a time: the
						// operand for tile t places its output rows in tile slot t % Pn of the NEW lane mode and takes its k-blocks from tile
						// slot t % Po of the OLD one, so the MFMA itself re-lays the data out between the two modes.
						const WnArrayCfg& prev = desc.arrays[a - 1];
						const int GpO = LaneMode(prev.channels), Po = 4 / GpO, Pn = 4 / Gp, NC = std::max(Po, Pn);
						WnSplitStage st = EmptySplit(WN_ST_ARRAY_LINK);
						st.G = CeilDiv(prev.channels, 4); st.Gp = GpO; st.ksize = Gp;
						st.a_ops = 4 * NC + 1;
						st.a_off = NewSplitOps(st.a_ops) * 64;
						const int op0 = st.a_off / 64;
						for (int u = 0; u < NC; u++)
						{
							const int pn = u % Pn, po = u % Po;
							FillSplitBlock(op0 + 4 * u, op0 + 4 * u + 1, 4 * Gp * pn, GpO * po, prev.headSize, prev.channels,
								[&](int o, int c) { return W(prevHeadW + o * prev.channels + c); });
							FillSplitBlock(op0 + 4 * u + 2, op0 + 4 * u + 3, 4 * Gp * pn, GpO * po, C, cfg.inputSize,
								[&](int o, int c) { return W(rechOff + o * cfg.inputSize + c); });
						}
						if (prev.hasHeadBias)
						{
							st.flags |= WN_FLAG_BIAS;
							FillSplitAux(op0 + 4 * NC, Gp, prev.headSize, -1, prevHeadB);
						}
						SplitOutRing(st, layerRing[a][0]);
						st.flags |= WN_FLAG_PUBLISH;
						plan.sstages.push_back(st);
					}

					for (int l = 0; l < numLayers; l++)
					{
						const int K = cfg.kernelSizes[l];
						const int wconv = Take((size_t)C * C * K);
						const int bconv = Take((size_t)C);
						const int wmix = Take((size_t)C * cfg.conditionSize);
						const int w1 = Take((size_t)C * C);
						const int b1 = Take((size_t)C);
						const bool lastLayer = (l == numLayers - 1);
						const bool needOutput = !(lastLayer && lastArray);
						{
							NoteTensor(wconv, (size_t)C * C * K, true);
							NoteTensor(bconv, (size_t)C, false);
							NoteTensor(wmix, (size_t)C * cfg.conditionSize, true);
							NoteTensor(w1, (size_t)C * C, needOutput); // (the very last 1x1 is dead, WaveNet.h:643,785: trained files hold denormals there)
							NoteTensor(b1, (size_t)C, false);
							// |z| <= zG * L + zA: 1.0081 for the rational tanh (the StdMath one: 1); LeakyReLU passes the conv's worst case on
							double zG = 0.0, zA = 1.0081;
							if (cfg.activation == ACT_LEAKYRELU)
							{
								double convRow = 0.0, mixMax = 0.0, biasMax = 0.0;
								for (int o = 0; o < C; o++)
								{
									double row = 0.0;
									for (int c = 0; c < C * K; c++) row += std::fabs((double)W(wconv + o * C * K + c));
									convRow = std::max(convRow, row);
									double mix = 0.0;
									for (int c = 0; c < cfg.conditionSize; c++) mix += std::fabs((double)W(wmix + o * cfg.conditionSize + c));
									mixMax = std::max(mixMax, mix);
									biasMax = std::max(biasMax, std::fabs((double)W(bconv + o)));
								}
								zG = convRow * rangeGain + mixMax;
								zA = convRow * rangeAdd + biasMax;
							}
							RangeNote(zG, zA);
							rangeHeadGain = (l == 0 && a == 0 ? 0.0 : rangeHeadGain) + zG;
							rangeHeadAdd = (l == 0 && a == 0 ? 0.0 : rangeHeadAdd) + zA;
							RangeNote(rangeHeadGain, rangeHeadAdd);
							// the 1x1 adds at most max_o (sum_c |w1[o][c]| * |z| + |b1[o]|) to the residual stream
							double rowMax = 0.0, b1Max = 0.0;
							for (int o = 0; o < C; o++)
							{
								double row = 0.0;
								for (int c = 0; c < C; c++) row += std::fabs((double)W(w1 + o * C + c));
								rowMax = std::max(rowMax, row);
								b1Max = std::max(b1Max, std::fabs((double)W(b1 + o)));
							}
							rangeGain += rowMax * zG;
							rangeAdd += rowMax * zA + b1Max;
							RangeNote(rangeGain, rangeAdd);
						}

						WnSplitStage st = EmptySplit(WN_ST_LAYER);
						st.G = G; st.Gp = Gp; st.ksize = K; st.dilation = cfg.dilations[l];
						// operands: taps 0..K-1 (hi, lo each; tap K-1 is the unshifted one), aux = (mix-in, conv bias), then 1x1 (hi, lo) and its bias
						// (the very last layer's 1x1 output is dead -- NeedOutput == false, WaveNet.h:643,785 -- but it still gets its operands:
						// the kernel runs ONE code path for every layer)
						st.a_ops = 2 * K + 4;
						st.a_off = NewSplitOps(st.a_ops) * 64;
						const int op0 = st.a_off / 64;
						for (int k = 0; k < K; k++)
							FillSplitMerged(op0 + 2 * k, Gp, C, C, [&](int o, int c) { return W(wconv + (o * C + c) * K + k); });
						FillSplitAux(op0 + 2 * K, Gp, C, wmix, bconv, cpad);
						st.reserved = groupsPerStream;
						FillSplitMerged(op0 + 2 * K + 1, Gp, C, C, [&](int o, int c) { return W(w1 + o * C + c); });
						FillSplitAux(op0 + 2 * K + 3, Gp, C, -1, b1);
						if (needOutput) st.flags |= WN_FLAG_NEED_OUTPUT;
						SplitRing(st, layerRing[a][l]);
						if (cfg.activation == ACT_LEAKYRELU) st.flags |= WN_FLAG_LEAKY;
						else if (desc.mathMode == MATH_STD) st.flags |= WN_FLAG_STD_TANH;
						if (!lastLayer)
						{
							st.flags |= WN_FLAG_PUBLISH;
							SplitOutRing(st, layerRing[a][l + 1]);
						}
						plan.sstages.push_back(st);
					}

					const int wh = Take((size_t)cfg.headSize * C * cfg.headKernelSize);
					const int bh = cfg.hasHeadBias ? Take((size_t)cfg.headSize) : -1;
					NoteTensor(wh, (size_t)cfg.headSize * C * cfg.headKernelSize, true);
					if (bh >= 0) NoteTensor(bh, (size_t)cfg.headSize, false);
					if (!lastArray)
					{
						// the head rechannel's output is the next array's head accumulator (WaveNet.h:785-789)
						double rowMax = 0.0, bMax = 0.0;
						for (int o = 0; o < cfg.headSize; o++)
						{
							double row = 0.0;
							for (int c = 0; c < C * cfg.headKernelSize; c++) row += std::fabs((double)W(wh + o * C * cfg.headKernelSize + c));
							rowMax = std::max(rowMax, row);
							if (bh >= 0) bMax = std::max(bMax, std::fabs((double)W(bh + o)));
						}
						rangeHeadGain *= rowMax;
						rangeHeadAdd = rangeHeadAdd * rowMax + bMax;
						RangeNote(rangeHeadGain, rangeHeadAdd);
					}
					if (lastArray)
					{
						// only head channel 0 reaches the output (WaveNet.h:793-798): one output row per tile slot
						const int Kh = cfg.headKernelSize;
						WnSplitStage st = EmptySplit(Kh == 1 ? WN_ST_HEAD_DENSE_OUT : WN_ST_HEAD_CONV_OUT);
						st.G = G; st.Gp = Gp; st.ksize = Kh; st.dilation = cfg.headDilation;
						st.a_ops = 2 * Kh + 1;
						st.a_off = NewSplitOps(st.a_ops) * 64;
						const int op0 = st.a_off / 64;
						// (a packed plan: one output row per packed stream, head channel q = stream q, see PackWaveNetDesc)
						for (int k = 0; k < Kh; k++)
							FillSplitMerged(op0 + 2 * k, Gp, pack > 1 ? cfg.headSize : 1, C, [&](int o, int c) { return W(wh + (o * C + c) * Kh + k); });
						if (cfg.hasHeadBias)
						{
							st.flags |= WN_FLAG_BIAS;
							FillSplitAux(op0 + 2 * Kh, Gp, pack > 1 ? cfg.headSize : 1, -1, bh);
						}
						if (Kh > 1)
						{
							SplitRing(st, headRing[a]);
							SplitOutRing(st, headRing[a]);
						}
						plan.sstages.push_back(st);
					}
					else
					{
						prevHeadW = wh;
						prevHeadB = bh;
					}
				}
				for (const WnSplitStage& st : plan.sstages) plan.maxSplitOps = std::max(plan.maxSplitOps, st.a_ops);
				// Range contract of the f16-split kernels: the hi half of every value is an f16 (|v| <= 65504).  With the input clamped to
				// +-condLimit every split value stays below half of that: G * limit + A <= 32752 for every (G, A) collected above.  A model
				// for which that leaves less than kSplitMinInputLimit -- or whose weights do not fit the operand format -- is not run by
				// the f16-split kernels at all (splitRangeProven / splitWeightsOk -> the f32 frame kernel, gpu_batch.cpp FamilyFor).
				{
					double limit = 32752.0;
					for (const auto& c : rangeCons)
					{
						const double room = 32752.0 - c.second;
						limit = std::min(limit, room <= 0.0 ? 0.0 : (c.first > 0.0 ? room / c.first : 32752.0));
					}
					plan.splitRangeProven = limit >= kSplitMinInputLimit;
					// no proof (in practice: LeakyReLU models, whose worst case grows with the product of every layer's row sums): the limit
					// then covers the rechannel path only, and the kernels that still run such a model -- the official A2 chains -- saturate
					// at the f16 range and count the event instead of overflowing (wavenet_split_dev.h SplitQuadSat)
					plan.condLimit = (float)(plan.splitRangeProven ? limit : std::min(32752.0, 32752.0 / std::max(1e-6, linearGainMax)));
					// weights: |w| <= half the f16 range (hi + lo never overflow); every live weight MATRIX has its largest entry above
					// 2^-12 -- a split value carries an absolute error of 2^-25 (f16 subnormals), so the entries that matter keep >= 13
					// bits and nothing that matters is flushed to zero (|w| < 2^-25 is)
					plan.splitWeightsOk = weightAbsMax <= 32752.0 && (matrixPeakMin >= 1.0 / 4096.0 || matrixPeakMin == 1e300);
				}
				// fast instantiation of the kernel: K == 3 everywhere, every array fills its lane mode (G == Gp), 1x1 heads, weight blocks
				// within the fixed 16 KB staging part
				plan.splitFastT = 2;
				for (const WnSplitStage& st : plan.sstages)
				{
					if (st.a_ops > 16 || st.G != st.Gp || st.type == WN_ST_HEAD_CONV_OUT || (st.type == WN_ST_LAYER && st.ksize != 3)) plan.splitFastT = 0;
					if (plan.splitFastT && (st.Gp == 1 || (st.type == WN_ST_ARRAY_LINK && st.ksize == 1))) plan.splitFastT = 4; // 4 tiles share an MFMA
				}
			}

			// Models wider than the shaped kernels take (> 16 channels): rings and the natural-layout tensor table only -- what the prewarm
			// kernel and the runtime-shaped block kernel (wavenet_generic_kernels.hip) walk.  Same weight order as Build() (WaveNet.h:700-719).
			void BuildGenericOnly()
			{
				const int numArrays = (int)desc.arrays.size();
				cursor = 0;
				for (int a = 0; a < numArrays; a++)
				{
					const WnArrayCfg& cfg = desc.arrays[a];
					const int C = cfg.channels;
					const int numLayers = (int)cfg.kernelSizes.size();
					const int rechOff = Take((size_t)C * cfg.inputSize);
					for (int l = 0; l < numLayers; l++)
					{
						const int K = cfg.kernelSizes[l];
						WnPrewarmLayer pw = {};
						pw.kind = 0;
						pw.cin = C; pw.cout = C; pw.ksize = K;
						pw.act = (cfg.activation == ACT_LEAKYRELU) ? 1 : (desc.mathMode == MATH_STD ? 2 : 0);
						pw.wconv = Take((size_t)C * C * K);
						pw.bconv = Take((size_t)C);
						pw.wmix = Take((size_t)C * cfg.conditionSize);
						pw.w1 = Take((size_t)C * C);
						pw.b1 = Take((size_t)C);
						pw.ring_id = AddRing(C, (K - 1) * cfg.dilations[l], cfg.dilations[l], l == 0);
						pw.need_output = 1;
						pw.last_of_array = (l == numLayers - 1) ? 1 : 0;
						pw.rechannel = (l == 0) ? rechOff : -1;
						pw.rech_in = cfg.inputSize;
						pw.dilation = cfg.dilations[l];
						plan.prewarm.push_back(pw);
					}
					WnPrewarmLayer pw = {};
					pw.kind = 1;
					pw.cin = C; pw.cout = cfg.headSize; pw.ksize = cfg.headKernelSize;
					pw.wconv = Take((size_t)cfg.headSize * C * cfg.headKernelSize);
					pw.bconv = cfg.hasHeadBias ? Take((size_t)cfg.headSize) : -1;
					pw.wmix = -1; pw.w1 = -1; pw.b1 = -1;
					pw.ring_id = (cfg.headKernelSize > 1) ? AddRing(C, (cfg.headKernelSize - 1) * cfg.headDilation, cfg.headDilation) : -1; // a conv head keeps its own history
					pw.rechannel = -1;
					pw.dilation = cfg.headDilation;
					plan.prewarm.push_back(pw);
				}
				plan.headScale = W(Take(1));
				plan.stateF4 = CeilDiv(plan.stateF4, 16) * 16;
			}

			void Build()
			{
				ValidateWaveNetDesc(desc);

				plan.arrays = desc.arrays;
				plan.receptiveField = desc.ReceptiveFieldSize();
				compactOk = exactRings;
				for (const WnArrayCfg& cfg : desc.arrays)
				{
					if (cfg.headKernelSize != 1) compactOk = false;
					for (int k : cfg.kernelSizes)
						if (k > 3) compactOk = false;
				}
				plan.stateF4 = WN_HEADER_F4;
				plan.genericOk = true;
				for (const WnArrayCfg& cfg : desc.arrays)
				{
					plan.maxChannels = std::max(plan.maxChannels, std::max(cfg.channels, std::max(cfg.headSize, cfg.inputSize)));
					(void)cfg; // (conv heads run on the runtime-shaped kernel too since round 3)
				}
				if (plan.maxChannels > 16)
				{
					plan.genericOnly = true;
					BuildGenericOnly();
					return;
				}

				const int numArrays = (int)desc.arrays.size();

				// Pass 1: rings (one per conv layer, plus a head ring when the head conv has K > 1)
				std::vector<std::vector<int>> layerRing(numArrays);
				std::vector<int> headRing(numArrays, -1);
				for (int a = 0; a < numArrays; a++)
				{
					const WnArrayCfg& cfg = desc.arrays[a];
					for (size_t l = 0; l < cfg.kernelSizes.size(); l++)
						layerRing[a].push_back(AddRing(cfg.channels, (cfg.kernelSizes[l] - 1) * cfg.dilations[l], cfg.dilations[l], l == 0));
					if (cfg.headKernelSize > 1)
						headRing[a] = AddRing(cfg.channels, (cfg.headKernelSize - 1) * cfg.headDilation, cfg.headDilation);
				}

				// Pass 2: walk the flat weights and emit stages
				int prevHeadW = -1, prevHeadB = -1;
				for (int a = 0; a < numArrays; a++)
				{
					const WnArrayCfg& cfg = desc.arrays[a];
					const int C = cfg.channels;
					const int numLayers = (int)cfg.kernelSizes.size();
					const bool lastArray = (a == numArrays - 1);

					const int rechOff = Take((size_t)C * cfg.inputSize);

					if (a == 0)
					{
						WnStage st = EmptyStage(WN_ST_RECHANNEL_COND);
						st.vec_off = AllocF4(16);
						SetVec(st, 3, rechOff, C); // aux slot: w_re[c] (input_size == 1)
						SetOutRing(st, layerRing[a][0]);
						st.flags = WN_FLAG_PUBLISH;
						plan.stages.push_back(st);
					}
					else
					{
						const WnArrayCfg& prev = desc.arrays[a - 1];
						WnStage st = EmptyStage(WN_ST_ARRAY_LINK);
						st.vec_off = AllocF4(16);
						if (prev.hasHeadBias)
						{
							st.flags |= WN_FLAG_BIAS;
							SetVec(st, 0, prevHeadB, prev.headSize);
						}
						st.a4_off = PackDenseA4(prevHeadW, prev.channels, prev.headSize, 4); // [head dense | rechannel], padded to 16x16 each
						PackDenseA4(rechOff, cfg.inputSize, C, 4);
						st.a4_floats = 2 * 256;
						SetOutRing(st, layerRing[a][0]);
						st.flags |= WN_FLAG_PUBLISH;
						plan.stages.push_back(st);
					}

					for (int l = 0; l < numLayers; l++)
					{
						const int K = cfg.kernelSizes[l];
						const int d = cfg.dilations[l];
						const int wconv = Take((size_t)C * C * K);
						const int bconv = Take((size_t)C);
						const int wmix = Take((size_t)C * cfg.conditionSize);
						const int w1 = Take((size_t)C * C);
						const int b1 = Take((size_t)C);

						const bool lastLayer = (l == numLayers - 1);
						WnStage st = EmptyStage(WN_ST_LAYER);
						st.G = CeilDiv(C, 4);
						st.vec_off = AllocF4(16);
						SetVec(st, 0, bconv, C);
						SetVec(st, 1, wmix, C);
						SetVec(st, 2, b1, C);
						st.ksize = K;
						st.dilation = d;
						st.a4_off = PackConvA4(wconv, C, C, K, st.G);       // [conv taps | 1x1 | vectors], one contiguous block
						PackDenseA4(w1, C, C, st.G);
						{
							// tail of the block: conv bias | mixin | 1x1 bias, 4G floats each, read back from LDS as broadcast float4s
							const int CP = 4 * st.G;
							const int tail = AllocPk(3 * CP);
							for (int c = 0; c < C; c++)
							{
								plan.wpk[(size_t)tail + c] = W(bconv + c);
								plan.wpk[(size_t)tail + CP + c] = W(wmix + c);
								plan.wpk[(size_t)tail + 2 * CP + c] = W(b1 + c);
							}
						}
						st.a4_floats = (K + 1) * 64 * st.G + 12 * st.G;
						SetRing(st, layerRing[a][l]);
						if (cfg.activation == ACT_LEAKYRELU) st.flags |= WN_FLAG_LEAKY;
						else if (desc.mathMode == MATH_STD) st.flags |= WN_FLAG_STD_TANH;
						// NeedOutput=false for the very last layer (WaveNet.h:643,785); for a single-array model the
						// reference still computes it but nothing reads it.
						if (!(lastLayer && lastArray)) st.flags |= WN_FLAG_NEED_OUTPUT;
						if (!lastLayer)
						{
							st.flags |= WN_FLAG_PUBLISH;
							SetOutRing(st, layerRing[a][l + 1]);
						}
						plan.stages.push_back(st);

						WnPrewarmLayer pw = {};
						pw.kind = 0;
						pw.cin = C; pw.cout = C; pw.ksize = K;
						pw.act = (cfg.activation == ACT_LEAKYRELU) ? 1 : (desc.mathMode == MATH_STD ? 2 : 0);
						pw.wconv = wconv; pw.bconv = bconv; pw.wmix = wmix; pw.w1 = w1; pw.b1 = b1;
						pw.ring_id = layerRing[a][l];
						pw.need_output = 1;
						pw.last_of_array = lastLayer ? 1 : 0;
						pw.rechannel = (l == 0) ? rechOff : -1;
						pw.rech_in = cfg.inputSize;
						pw.dilation = d;
						plan.prewarm.push_back(pw);
					}

					const int wh = Take((size_t)cfg.headSize * C * cfg.headKernelSize);
					const int bh = cfg.hasHeadBias ? Take((size_t)cfg.headSize) : -1;

					WnPrewarmLayer pw = {};
					pw.kind = 1;
					pw.cin = C; pw.cout = cfg.headSize; pw.ksize = cfg.headKernelSize;
					pw.wconv = wh; pw.bconv = bh; pw.wmix = -1; pw.w1 = -1; pw.b1 = -1;
					pw.ring_id = headRing[a];
					pw.rechannel = -1;
					pw.dilation = cfg.headDilation;
					plan.prewarm.push_back(pw);

					if (lastArray)
					{
						if (cfg.headKernelSize == 1)
						{
							WnStage st = EmptyStage(WN_ST_HEAD_DENSE_OUT);
							st.vec_off = AllocF4(16);
							st.G = CeilDiv(C, 4);
							st.pk_w1_off = PackDensePk(wh, C, 1, 16, 1); // only head channel 0 reaches the output (WaveNet.h:793-798)
							if (cfg.hasHeadBias)
							{
								st.flags |= WN_FLAG_BIAS;
								SetVec(st, 0, bh, cfg.headSize);
							}
							plan.stages.push_back(st);
						}
						else
						{
							WnStage st = EmptyStage(WN_ST_HEAD_CONV_OUT);
							st.G = CeilDiv(C, 4);
							st.vec_off = AllocF4(16);
							st.ksize = cfg.headKernelSize;
							st.dilation = cfg.headDilation;
							st.pk_conv_off = PackConvPk(wh, C, 1, cfg.headKernelSize, 4 * st.G, 1);
							if (cfg.hasHeadBias)
							{
								st.flags |= WN_FLAG_BIAS;
								SetVec(st, 0, bh, cfg.headSize);
							}
							SetRing(st, headRing[a]);
							SetOutRing(st, headRing[a]);
							plan.stages.push_back(st);
						}
					}
					else
					{
						prevHeadW = wh;
						prevHeadB = bh;
					}
				}

				plan.headScale = W(Take(1));
				BuildSplit(layerRing, headRing);
				for (const WnStage& st : plan.stages) plan.maxA4Floats = std::max(plan.maxA4Floats, st.a4_floats);
				// round the state up to a 256-byte multiple so every stream's state starts float4/line aligned
				plan.stateF4 = CeilDiv(plan.stateF4, 16) * 16;
			}
		};
	}

	WaveNetPlan BuildWaveNetPlan(const WaveNetDesc& desc, bool splitStateFormat)
	{
		Builder b(desc, 1, splitStateFormat);
		b.Build();
		return std::move(b.plan);
	}

	// ---- stream packing ------------------------------------------------------------------------------------------------------
	// A narrow model leaves most of a 16x16 MFMA tile empty (4 channels: one channel group of four), and what a layer costs on the
	// f16-split kernel hardly depends on its width.  So P streams of such a model run as ONE virtual stream whose channel groups belong
	// to different streams: channels C_v = P * pad4(C) per layer array, every weight matrix block-diagonal (stream q's block at rows /
	// columns q * pad4(C) ...), bias / mix-in / rechannel vectors replicated, the last head with one output row per stream.  The
	// only thing that is per stream in the arithmetic is the condition (= input sample): the kernel's aux operand carries stream q's
	// condition in the k-blocks of stream q's channel groups (FillSplitAux).  Same flat weight order as the reference (WaveNet.h:700-719).
	int WaveNetPackFactor(const WaveNetDesc& desc)
	{
		int maxPad = 0;
		for (const WnArrayCfg& cfg : desc.arrays)
		{
			maxPad = std::max(maxPad, CeilDiv(cfg.channels, 4) * 4);
			if (cfg.headKernelSize != 1 || cfg.conditionSize != 1) return 1;
			for (int k : cfg.kernelSizes)
				if (k != 3) return 1; // the fast instantiation of the kernel
		}
		if (desc.arrays.back().headSize != 1) return 1;
		return maxPad <= 4 ? 4 : (maxPad <= 8 ? 2 : 1);
	}

	// P == 1: no packing, only padding -- every layer array is widened to the channel count that fills its lane mode (12 -> 16, 6 -> 8,
	// 3 -> 4), so that a model like A1 Lite runs the fast flavour of the split kernel (zero rows / columns cost nothing there: the
	// cost of a layer is its skeleton, not its width).
	// Dense packs: four streams whose LAST array has two channels (Nano: 4 / 2) keep that array at 2 channels per stream -- 8 virtual
	// channels, two streams per channel group -- instead of padding it to 4: half the ring traffic of the array's thirteen layers and
	// the 8-channel lane mode (two tiles per MFMA).  Only the last array of a two-array model (its head has one output per stream; an
	// array in front of another one would hand half channel groups to the link's lane-mode change).
	bool WaveNetPackCanBeDense(const WaveNetDesc& desc, int P)
	{
		return P == 4 && desc.arrays.size() == 2 && desc.arrays[0].channels == 4 && desc.arrays[1].channels == 2;
	}

	WaveNetDesc PackWaveNetDesc(const WaveNetDesc& desc, int P, bool dense)
	{
		WaveNetDesc v;
		v.mathMode = desc.mathMode;
		const int numArrays = (int)desc.arrays.size();
		if (dense && !WaveNetPackCanBeDense(desc, P)) throw std::runtime_error("internal: this WaveNet has no dense pack");
		std::vector<int> cpad((size_t)numArrays);
		for (int a = 0; a < numArrays; a++)
			cpad[(size_t)a] = (dense && a == numArrays - 1) ? desc.arrays[(size_t)a].channels
				: (P > 1 ? CeilDiv(desc.arrays[(size_t)a].channels, 4) * 4 : 4 * LaneMode(desc.arrays[(size_t)a].channels));
		size_t pos = 0;
		auto take = [&](size_t n) { const size_t at = pos; pos += n; return at; };
		for (int a = 0; a < numArrays; a++)
		{
			const WnArrayCfg& cfg = desc.arrays[(size_t)a];
			const bool last = (a == numArrays - 1);
			const int C = cfg.channels, Cp = cpad[(size_t)a], Cv = P * Cp;
			const int In = cfg.inputSize, Inp = (a == 0) ? 1 : cpad[(size_t)a - 1], Inv = (a == 0) ? 1 : P * Inp;
			const int Hs = cfg.headSize, Hp = last ? 1 : cpad[(size_t)a + 1], Hv = P * Hp;
			WnArrayCfg vc = cfg;
			vc.channels = Cv;
			vc.inputSize = Inv;
			vc.headSize = Hv;
			v.arrays.push_back(vc);
			// rechannel W[out][in]
			{
				const size_t w = take((size_t)C * In);
				std::vector<float> m((size_t)Cv * Inv, 0.0f);
				for (int q = 0; q < P; q++)
					for (int o = 0; o < C; o++)
						for (int c = 0; c < In; c++)
							m[(size_t)(q * Cp + o) * Inv + (a == 0 ? 0 : q * Inp + c)] = desc.weights[w + (size_t)o * In + c];
				v.weights.insert(v.weights.end(), m.begin(), m.end());
			}
			for (size_t l = 0; l < cfg.kernelSizes.size(); l++)
			{
				const int K = cfg.kernelSizes[l];
				const size_t wconv = take((size_t)C * C * K), bconv = take((size_t)C), wmix = take((size_t)C), w1 = take((size_t)C * C), b1 = take((size_t)C);
				std::vector<float> m((size_t)Cv * Cv * K, 0.0f), vb((size_t)Cv, 0.0f), vm((size_t)Cv, 0.0f), m1((size_t)Cv * Cv, 0.0f), vb1((size_t)Cv, 0.0f);
				for (int q = 0; q < P; q++)
					for (int o = 0; o < C; o++)
					{
						const int vo = q * Cp + o;
						vb[(size_t)vo] = desc.weights[bconv + o];
						vm[(size_t)vo] = desc.weights[wmix + o];
						vb1[(size_t)vo] = desc.weights[b1 + o];
						for (int c = 0; c < C; c++)
						{
							const int vcn = q * Cp + c;
							m1[(size_t)vo * Cv + vcn] = desc.weights[w1 + (size_t)o * C + c];
							for (int k = 0; k < K; k++) m[((size_t)vo * Cv + vcn) * K + k] = desc.weights[wconv + ((size_t)o * C + c) * K + k];
						}
					}
				v.weights.insert(v.weights.end(), m.begin(), m.end());
				v.weights.insert(v.weights.end(), vb.begin(), vb.end());
				v.weights.insert(v.weights.end(), vm.begin(), vm.end());
				v.weights.insert(v.weights.end(), m1.begin(), m1.end());
				v.weights.insert(v.weights.end(), vb1.begin(), vb1.end());
			}
			// head rechannel (K = 1): W[out][in], bias[out]
			{
				const size_t wh = take((size_t)Hs * C), bh = cfg.hasHeadBias ? take((size_t)Hs) : 0;
				std::vector<float> m((size_t)Hv * Cv, 0.0f), vb((size_t)Hv, 0.0f);
				for (int q = 0; q < P; q++)
					for (int o = 0; o < Hs; o++)
					{
						const int vo = q * Hp + o;
						if (cfg.hasHeadBias) vb[(size_t)vo] = desc.weights[bh + o];
						for (int c = 0; c < C; c++) m[(size_t)vo * Cv + (q * Cp + c)] = desc.weights[wh + (size_t)o * C + c];
					}
				v.weights.insert(v.weights.end(), m.begin(), m.end());
				if (cfg.hasHeadBias) v.weights.insert(v.weights.end(), vb.begin(), vb.end());
			}
		}
		v.weights.push_back(desc.weights[take(1)]); // head scale
		if (pos != desc.weights.size()) throw std::runtime_error("internal: PackWaveNetDesc walked a different number of weights");
		return v;
	}

	// true: the model qualifies for the fast split-kernel flavour except that some layer array does not fill its lane mode
	bool WaveNetWantsPadding(const WaveNetDesc& desc)
	{
		bool partial = false;
		for (const WnArrayCfg& cfg : desc.arrays)
		{
			if (cfg.channels > 16 || cfg.headKernelSize != 1 || cfg.conditionSize != 1) return false;
			for (int k : cfg.kernelSizes)
				if (k != 3) return false;
			if (cfg.channels != 4 * LaneMode(cfg.channels)) partial = true;
		}
		if (desc.arrays.back().headSize != 1) return false;
		// arrays of <= 4 channels want 4 tiles per wave when they run alone (packing is what helps them), leave those to PackFor
		for (const WnArrayCfg& cfg : desc.arrays)
			if (cfg.channels <= 4) return false;
		return partial;
	}

	WaveNetPlan BuildPackedWaveNetPlan(const WaveNetDesc& desc, int P, bool dense)
	{
		if (P < 1) return BuildWaveNetPlan(desc);
		const WaveNetDesc v = PackWaveNetDesc(desc, P, dense);
		Builder b(v, P, true); // (packed / padded plans only ever run on the f16-split kernels)
		b.Build();
		b.plan.pack = P;
		b.plan.packDense = dense;
		b.plan.packedWeights = v.weights; // the prewarm kernel walks the natural layout of the VIRTUAL model
		return std::move(b.plan);
	}


	// B(arch) = 8 + (4/N) * sum_layers C_l * [ sum_{j=1}^{K_l-1} min(j*d_l, N) + min(N, (K_l-1)*d_l) ]
	// (compulsory ring traffic with perfect in-block reuse, weights amortised; includes a head conv with K > 1)
	double WaveNetPlan::AlgorithmicBytesPerSample(int N) const
	{
		double sum = 0.0;
		for (const auto& a : arrays)
		{
			for (size_t l = 0; l < a.kernelSizes.size(); l++)
			{
				const int K = a.kernelSizes[l], d = a.dilations[l];
				double t = 0.0;
				for (int j = 1; j <= K - 1; j++) t += std::min(j * d, N);
				t += std::min(N, (K - 1) * d);
				sum += a.channels * t;
			}
			if (a.headKernelSize > 1)
			{
				const int K = a.headKernelSize, d = a.headDilation;
				double t = 0.0;
				for (int j = 1; j <= K - 1; j++) t += std::min(j * d, N);
				t += std::min(N, (K - 1) * d);
				sum += a.channels * t;
			}
		}
		return 8.0 + 4.0 * sum / N;
	}

	double WaveNetPlan::MacsPerSample() const
	{
		double macs = 0.0;
		const int numArrays = (int)arrays.size();
		for (int ai = 0; ai < numArrays; ai++)
		{
			const auto& a = arrays[ai];
			const double c = a.channels;
			macs += c * a.inputSize;
			for (size_t l = 0; l < a.kernelSizes.size(); l++)
			{
				const bool last = (ai == numArrays - 1) && (l == a.kernelSizes.size() - 1) && numArrays > 1;
				macs += c * c * a.kernelSizes[l] + c * a.conditionSize + (last ? 0.0 : c * c);
			}
			macs += (double)a.headSize * c * a.headKernelSize;
		}
		return macs;
	}
}
