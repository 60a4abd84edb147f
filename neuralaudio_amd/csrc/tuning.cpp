// tuning.cpp -- see tuning.h
#include "tuning.h"

#include <cstdlib>
#include <cstring>

namespace na
{
	namespace
	{
#ifndef NA_NO_TUNING
		bool Has(const char* name) { return getenv(name) != nullptr; }
		int Int(const char* name, int dflt)
		{
			const char* e = getenv(name);
			return e ? atoi(e) : dflt;
		}
		bool IsZero(const char* name)
		{
			const char* e = getenv(name);
			return e != nullptr && atoi(e) == 0;
		}
#endif
		Tuning Parse()
		{
			Tuning t;
#ifndef NA_NO_TUNING
			if (const char* e = getenv("NA_WN_KERNEL")) t.wnKernel = !strcmp(e, "split") ? 1 : (!strcmp(e, "frame") ? 2 : (!strcmp(e, "generic") ? 3 : 0));
			t.wnPack = Int("NA_WN_PACK", -1);
			t.wnPadOff = IsZero("NA_WN_PAD");
			t.wnNtOff = IsZero("NA_WN_NT");
			t.wnNtFromMB = Int("NA_WN_NT_MB", 400);
			t.wnDense = Int("NA_WN_DENSE", -1);
			t.spT = Int("NA_SP_T", 0);
			t.spSpb = Int("NA_SP_SPB", 0);
			t.spGen = Has("NA_SP_GEN");
			t.wnSpecOff = IsZero("NA_WN_SPEC") || Has("NA_SP_T") || Has("NA_SP_GEN");
			t.spNoT1 = Has("NA_SP_NO_T1");
			t.spReverse = Has("NA_SP_REVERSE");
			t.frPrefetch = Int("NA_FR_PF", 1);
			t.frSpb = Int("NA_FR_SPB", 0);
			t.traceBlock = Int("NA_TRACE_BLOCK", 0);
			t.traceChain = Int("NA_TRACE_CHAIN", 0);
			t.lstmNoDpp = Has("NA_LSTM_NO_DPP");
			t.gruNoDpp = Has("NA_GRU_NO_DPP");
			t.lstmLaneKernel = Has("NA_LSTM_LANE_KERNEL");
			t.lstmNoWaveRt = Has("NA_LSTM_NO_WAVE_RT");
			t.recL2w = Int("NA_REC_L2W", 0) != 0;
			t.recQuadMin = Int("NA_REC_QUAD_MIN", -1);
			t.recNoDpp32 = Has("NA_REC_NO_DPP32");
			t.recNoSkew = Has("NA_REC_NOSKEW");
			t.recNoPipe = Has("NA_REC_NOPIPE");
			t.recPipeMax = Int("NA_REC_PIPE_MAX", 0);
			t.recRpl = Int("NA_REC_RPL", 1);
			t.hostChains = Int("NA_HOST_CHAINS", 2);
			t.hostHalvesOff = IsZero("NA_HOST_HALVES");
			t.hostDirect = !IsZero("NA_HOST_DIRECT");
			t.batchSerial = Has("NA_BATCH_SERIAL");
			t.batchNoGraph = Has("NA_BATCH_NO_GRAPH");
			t.residentOn = Int("NA_RESIDENT", 0) != 0;
			t.residentHostRing = Int("NA_RESIDENT_HOST_RING", 0) != 0;
			t.residentDelayUs = Int("NA_RESIDENT_DELAY_US", 0);
			t.residentGrid = Int("NA_RESIDENT_GRID", 0);
			t.residentIdleUs = Int("NA_RESIDENT_IDLE_US", 200);
#endif
			return t;
		}
	}

	const Tuning& Tuning::Get()
	{
		static const Tuning t = Parse(); // (C++11: initialised once, thread-safe)
		return t;
	}
}
