// wavenet_dev.h -- device-side data model shared by the host packer (wavenet_plan.cpp) and the gfx950 WaveNet kernels
// (wavenet_split_kernels.hip and wavenet_frame_kernels.hip, one per model, see FamilyFor() in gpu_batch.cpp; the tile / packed-FMA
// fields below belong to the round-1 alternatives kept under tools/alternates/, not built into the library).
//
// A WaveNet model (reference: NeuralAudio/WaveNet.h) is lowered at load time into a short "stage program" (WnStage: one per
// rechannel / layer / array link / head) plus per-kernel weight images.  A workgroup interprets the program for one or two
// audio streams and one block of up to 128 frames.
//
// Tile layout (identical for the LDS block image and the HBM rings):
//     a tile is 16 frames x 4*G channels; the float4 holding channels 4g..4g+3 of frame j has index (tile*G + g)*16 + j, i.e.
//     16 frames x 4 channels per 256-byte row.  The frame kernel's lanes (lane = frame) read and write full rows; for the tile
//     kernel the same image is the C/D fragment of v_mfma_f32_16x16x4_f32 (row = 4*(lane>>4)+reg, col = lane&15) and, with the
//     weight operand permuted on the host, also a valid B fragment for the next mat-mul.
//
// HBM state per stream (float4 units):
//     [0, 16)            header: 64 ints, header[r] = write cursor (frame index) of ring r
//     ring r             ring_frames[r]/16 tiles of G[r] channel groups, true modulo ring of the
//                        INPUT of conv layer r (what ChannelHistoryBuffer holds in the reference,
//                        WaveNet.h:30-83), ring_frames = roundup16((K-1)*dilation) + 128 ("roomy"; the f16-split state
//                        format also has exact and compact rings, see WnRingKeep below).
#pragma once

#include <cstdint>

namespace na
{
	constexpr int WN_TILE = 16;            // frames per MFMA tile
	constexpr int WN_MAX_TILES = 8;        // tiles per launch
	constexpr int WN_MAX_FRAMES = WN_TILE * WN_MAX_TILES; // 128 frames per launch
	constexpr int WN_MAX_RINGS = 64;
	constexpr int WN_HEADER_F4 = WN_MAX_RINGS / 4; // header size in float4 units
	// f16-split kernels: header[63] counts "range events" of the stream -- (wave, block) pairs in which a value left the f16 range and was
	// saturated (only chains without a static range proof can get there; models with 64 rings do not run on those kernels)
	constexpr int WN_RANGE_EVENT_SLOT = WN_MAX_RINGS - 1;

	// Ring geometry (wavenet_plan.cpp AddRing).  H = roundup16 of a layer's history (K - 1) d:
	//   roomy    R = H + 128     every state format; a block of any n <= 128 frames never reads a position it writes
	//   exact    R = (K - 1) d   f16-split format, d >= 128, not the first layer of an array: the one shared position is read and
	//                            overwritten by the same lane, load first
	//   compact  R = 3 H         f16-split format, H <= 32 (A1-style models: K <= 3, dense heads): a block reads [p - H, p) and writes
	//                            its last H frames [p + n - H, p + n); for n in {1 .. 32, 64, 128} the two never meet modulo 3 H --
	//                            the host cuts other buffer lengths into such pieces (WaveNetPlan::compactRings)
	// How many frames at the end of a block a later block can still read: everything else of a block is never stored.
	constexpr int WN_COMPACT_MAX_HISTORY = 32;
	constexpr int WnRingKeep(int ringFrames) { return ringFrames < WN_MAX_FRAMES ? ringFrames / 3 : ringFrames - WN_MAX_FRAMES; }
	// the buffer lengths a model with compact rings takes in one launch
	constexpr bool WnCompactSafeFrames(int n) { return (n >= 1 && n <= 32) || n == 64 || n == 128; }

	enum WnStageType : int
	{
		WN_ST_RECHANNEL_COND = 0, // x = w_re * cond                   (array 0 rechannel, WaveNet.h:637)
		WN_ST_LAYER = 1,          // WaveNetLayerT::Process             (WaveNet.h:462-494)
		WN_ST_ARRAY_LINK = 2,     // head = Wh*head (+b); x = Wre*x     (A1 array i -> i+1, WaveNet.h:658-660,637)
		WN_ST_HEAD_DENSE_OUT = 3, // out = scale*(Wh*head + b)[0]       (A1 last array head, K=1)
		WN_ST_HEAD_CONV_OUT = 4,  // out = scale*(conv_K(head) + b)[0]  (A2 head, K=16)
	};

	enum WnStageFlags : int
	{
		WN_FLAG_LEAKY = 1,       // LeakyReLU(0.01) instead of FastMath tanh (Activation.h:83-118)
		WN_FLAG_NEED_OUTPUT = 2, // compute the 1x1 + residual (NeedOutput, WaveNet.h:486-491)
		WN_FLAG_PUBLISH = 4,     // write the layer output to LDS + the next layer's ring
		WN_FLAG_BIAS = 8,        // dense/head stage has a bias
		WN_FLAG_STD_TANH = 16,   // StdMath policy: std::tanh instead of the rational FastMath tanh (Activation.h:37-40)
	};

	struct WnStage
	{
		// ---- first 16 ints: everything the frame kernel reads per stage (one s_load_dwordx16; a stage is 128 bytes)
		int type;
		int flags;
		int G;               // channel groups of this stage's conv input
		int ksize;           // conv kernel size / dilation of this stage (0 for stages without a conv)
		int dilation;
		int ring_id;         // ring read by this stage's conv (-1: none)
		int ring_off;        // float4 offset of that ring in the stream state
		int ring_frames;
		int out_ring_id;     // ring receiving this stage's output block (-1: none)
		int out_ring_off;
		int out_ring_frames;
		int out_G;
		// frame kernel (lane = frame, v_mfma_f32_4x4x1_16b_f32): this stage's A-operand block in wpk, staged into LDS one stage ahead
		int a4_off;          // float offset into wpk ([conv taps | 1x1 | vectors] or [head dense | rechannel])
		int a4_floats;       // 0: stage has no MFMA weights
		int vec_off;         // float4 index into wpack: [0..3] conv/dense bias, [4..7] mix-in w, [8..11] 1x1 bias, [12..15] aux
		int pk_conv_off;     // head conv stage: [tap][in c] weights, float offset into wpk
		// ---- the rest
		int pk_w1_off;       // head dense [in c] (only head channel 0 reaches the output), float offset into wpk
		int reserved[15];
	};
	static_assert(sizeof(WnStage) == 128, "stage descriptors are 128-byte records");


	// ---- the f16-split MFMA kernel (wavenet_split_kernels.hip, the shipped path) ------------------------------------------
	// Its own, compact stage record (16 ints = one scalar load).  Activations live in HBM rings / LDS as "split quads": 4 consecutive
	// channels of one frame = 16 bytes = [h0 h1 | h2 h3 | l0 l1 | l2 l3] f16, h = f16(x), l = f16(x - h) -- the same 4 bytes per
	// element as f32, 22 mantissa bits.  Ring image: frame-major [frame][channel group] quads (a wave's 64 lanes = 16 frames x 4
	// groups, or 32 x 2, or 64 x 1, always touch 1 KB of consecutive bytes).
	struct WnSplitStage
	{
		int type;            // WnStageType
		int flags;           // WnStageFlags
		int G;               // channel groups (of 4) of this stage's input
		int Gp;              // lane mode: 1, 2 or 4 channel groups per 16-frame tile (G rounded up to a power of two)
		int ksize;           // conv kernel size; WN_ST_ARRAY_LINK: lane mode of the NEXT array
		int dilation;
		int ring_off;        // quad (16 B) offset of the ring this stage's conv reads, -1: none
		int ring_frames;
		int ring_id;
		int out_ring_off;    // ring receiving this stage's output, -1: none
		int out_ring_frames;
		int out_ring_id;
		int out_G;
		int a_off;           // this stage's A-operand block in the split weight image: quad offset (64 quads = 1 KB per MFMA operand)
		int a_ops;           // number of 1 KB operands
		int reserved;
	};
	static_assert(sizeof(WnSplitStage) == 64, "split stage descriptors are 64-byte records");

	constexpr int WN_GENERIC_MAX_CHANNELS = 128; // runtime-shaped block kernels (wavenet_generic_kernels.hip): channels per layer array (65 .. 128: WaveNetWideKernel)
	constexpr int WN_COL_STRIDE = 128;          // floats per ring in the steady-state column table

	// Natural-layout tensor table used by the prewarm kernel and the runtime-shaped block kernel (one entry per conv ring).
	struct WnPrewarmLayer
	{
		int kind;       // 0: layer, 1: head conv (K may be 1)
		int cin, cout, ksize;
		int act;        // 0 FastMath tanh, 1 leaky, 2 StdMath tanh
		int wconv;      // offsets into the flat reference-order weight array
		int bconv;      // -1 if none
		int wmix;       // -1 for head
		int w1, b1;     // -1 for head
		int ring_id;    // ring whose steady-state column is the INPUT of this conv (-1: K==1 head, no ring)
		int need_output;
		int last_of_array;
		int rechannel;  // >=0: before this layer apply rechannel weights at this offset (first layer of an array)
		int rech_in;
		int dilation;   // layer conv / head conv dilation (the generic block kernel walks this table; the prewarm kernel ignores it)
	};

	// Everything the kernels need about one model; passed by value.
	struct WnModelDev
	{
		const WnStage* stages;
		const float* wpack;   // per-stage bias / mix-in vectors (float4 aligned)
		const float* wpk;       // frame kernel: per-stage A images + head weights (LDS-DMA / scalar-load friendly)
		const int* ring_frames; // [nrings]
		int nstages;
		int wpack_f4;         // size of wpack in float4 units
		int max_a4_floats;    // largest per-stage A-operand block of the frame kernel
		int max_ksize;        // largest conv kernel size over all layers
		int wpk_floats;
		int nrings;
		int state_f4;         // per-stream state size in float4 units (header + rings)
		float head_scale;
		// f16-split kernel
		const WnSplitStage* sstages;
		const void* wsplit;   // A-operand image (f16), 16-byte units
		int wsplit_quads;
		int max_split_ops;    // largest per-stage operand count (sizes the LDS weight buffers)
		int max_G;            // largest channel-group count of any ring / stage (sizes the LDS block image)
		int split_fast_T;     // fewest tiles per wave the fast instantiation can run this model with (2 or 4); 0: needs the generic one
		int spec_arch;        // WnSpecArch: the compile-time specialised chain that runs this model (wavenet_spec_kernels.hip), 0: none
		float cond_limit;     // f16-split kernels: input samples are clamped to +-cond_limit (WaveNetPlan::condLimit)
		int compact_rings;    // f16-split state format: some ring is a compact one -- launches take WnCompactSafeFrames() lengths only
		int saturate;         // f16-split kernels: 1 = no static range proof (WaveNetPlan::splitRangeProven): saturating split + range events
	};

	enum WnSpecArch : int
	{
		WN_SPEC_NONE = 0,
		WN_SPEC_STD = 1,    // A1 Standard: 16 / 8 channels, dilations 1..512 twice
		WN_SPEC_LITE = 2,   // the lite dilation lists at 16 / 8 channels (A1 Lite padded, two packed Feather streams)
		WN_SPEC_LITE16 = 3, // ... at 16 / 16 channels (four packed Nano streams)
		WN_SPEC_A2FULL = 4, // A2 "Full": one array of 23 layers (K = 6 / 15), 8 channels, conv head of 16 taps, LeakyReLU
		WN_SPEC_A2LITE = 5, // A2 "Lite": the same with 3 channels (padded to 4; four tiles share an MFMA)
	};
}
