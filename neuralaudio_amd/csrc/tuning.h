// tuning.h -- every environment knob of the library in ONE struct, parsed once per process (first use, thread-safe) and read-only after.
// None is needed in normal use: they select kernel variants for A/B measurements and for the test suite (tests/test_gpu_families.py runs
// every kernel family over the parity suites).  A build with -DNA_NO_TUNING ignores the environment altogether: Get() returns the
// defaults below.  DESIGN.md 6a lists what each knob does and the measurement behind its default.
#pragma once

namespace na
{
	struct Tuning
	{
		// WaveNet: kernel family and layout
		int wnKernel = 0;          // NA_WN_KERNEL=split|frame|generic -> 1 | 2 | 3; 0: per model (FamilyFor)
		int wnPack = -1;           // NA_WN_PACK=0: no stream packing; -1: default
		int wnDense = -1;          // NA_WN_DENSE=0: four Nano streams packed at 16 / 16 virtual channels (default: 16 / 8, two streams per channel group of the second array)
		int wnNtFromMB = 400;      // NA_WN_NT_MB: stream state (MiB) from which a batch's long-dilation ring traffic is non-temporal (wavenet_launch.h WN_BEYOND_CACHE_BYTES)
		bool wnNtOff = false;      // NA_WN_NT=0: no non-temporal ring traffic for batches whose state exceeds the Infinity Cache
		bool wnPadOff = false;     // NA_WN_PAD=0: no padding of partly filled lane modes
		bool wnSpecOff = false;    // NA_WN_SPEC=0 (implied by NA_SP_T / NA_SP_GEN): the stage interpreter for everything
		int spT = 0;               // NA_SP_T=2|4: stage interpreter, tiles per wave
		int spSpb = 0;             // NA_SP_SPB=1|2(|4): streams per workgroup of the f16-split kernels
		bool spGen = false;        // NA_SP_GEN: stage interpreter, always the generic flavour
		bool spNoT1 = false;       // NA_SP_NO_T1: packed Nano never on one tile per wave
		bool spReverse = false;    // NA_SP_REVERSE: groups of a fused launch dispatched in the opposite order
		int frPrefetch = 1;        // NA_FR_PF=0|1|2: frame kernel history prefetch (none / registers / LDS-DMA)
		int frSpb = 0;             // NA_FR_SPB=1|2|4: frame kernel streams per workgroup
		int traceBlock = 0;        // NA_TRACE_BLOCK: workgroup stamped by the trace builds
		int traceChain = 0;        // NA_TRACE_CHAIN: which half-batch chain is stamped
		// recurrent kernels
		bool lstmNoDpp = false;    // NA_LSTM_NO_DPP
		bool gruNoDpp = false;     // NA_GRU_NO_DPP
		bool lstmLaneKernel = false; // NA_LSTM_LANE_KERNEL
		bool lstmNoWaveRt = false; // NA_LSTM_NO_WAVE_RT
		bool recL2w = false;       // NA_REC_L2W=1
		int recQuadMin = -1;       // NA_REC_QUAD_MIN (streams of a launch from which one-layer recurrent models run four streams per wave; -1: two waves per SIMD + 1, 0: never)
		bool recNoDpp32 = false;   // NA_REC_NO_DPP32
		bool recNoSkew = false;    // NA_REC_NOSKEW
		bool recNoPipe = false;    // NA_REC_NOPIPE: two-layer 16-unit LSTMs on the one-wave body instead of one wave per layer (recurrent_dpp_kernels.hip LstmDppPipeBody)
		int recPipeMax = 0;        // NA_REC_PIPE_MAX: > 0 = launches of more waves than this (two per pipelined stream, one per other) keep one wave per stream; 0 = UsePipe's rule
		int recRpl = 1;            // NA_REC_RPL=1|2|4|8
		// host side
		int hostChains = 2;        // NA_HOST_CHAINS=2..4
		bool hostHalvesOff = false; // NA_HOST_HALVES=0
		bool hostDirect = true;    // NA_HOST_DIRECT=0: copy engines instead of kernels on the pinned block
		bool batchSerial = false;  // NA_BATCH_SERIAL
		bool batchNoGraph = false; // NA_BATCH_NO_GRAPH: multi-unit batches issue their fork / join sequence directly every buffer instead of replaying a captured hipGraph
		bool residentOn = false;   // NA_RESIDENT=1: batches start with the resident launch enabled (NA_BatchSetResidentLaunch; default: free-running chains)
		bool residentHostRing = false; // NA_RESIDENT_HOST_RING=1: the command ring in pinned host memory even where the BAR maps device memory
		int residentDelayUs = 0;   // NA_RESIDENT_DELAY_US: start offset of the second workgroup of every CU inside the resident launch
		int residentGrid = 0;      // NA_RESIDENT_GRID: at most this many workgroups in the resident launch (0: as many as are resident)
		int residentIdleUs = 200;  // NA_RESIDENT_IDLE_US: a resident workgroup leaves after this long without a command

		static const Tuning& Get();
	};
}
