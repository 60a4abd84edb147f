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
#include "wavenet_spec_impl.h"

namespace na
{
	// NA_WN_SPEC=0 (environment) or SetWaveNetSpecEnabled(false) (tests: both kernels on one stream in one process) turn the chains off
	static int& SpecSwitch()
	{
		// (NA_SP_T / NA_SP_GEN select interpreter variants: they imply it)
		static int on = Tuning::Get().wnSpecOff ? 0 : 1;
		return on;
	}
	bool WaveNetSpecEnabled() { return SpecSwitch() != 0; }
	void SetWaveNetSpecEnabled(bool on) { SpecSwitch() = on ? 1 : 0; }

	int WaveNetSpecArchId(const WnSplitStage* stages, int nstages, int stateF4, int wsplitQuads)
	{
		if (spk::Matches<spk::ArchStd>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_STD;
		if (spk::Matches<spk::ArchLite>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_LITE;
		if (spk::Matches<spk::ArchLite16>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_LITE16;
		if (spk::Matches<spk::ArchA2Full>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_A2FULL;
		if (spk::Matches<spk::ArchA2Lite>(stages, nstages, stateF4, wsplitQuads)) return WN_SPEC_A2LITE;
		return WN_SPEC_NONE;
	}

	// More groups than the kernarg segment holds, in one launch (wavenet_launch.h).  Streams per workgroup: two share the staged weights,
	// so a group with an odd stream count leaves half a workgroup idle -- with mostly one-stream groups (every stream its own model)
	// the half-size workgroups waste nothing.
	hipError_t LaunchWaveNetSpecTable(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, WnLaunchTable& table)
	{
#ifdef NA_SP_QUICK
		return hipErrorNotSupported;
#else
		if (numGroups <= WN_FRAME_MAX_GROUPS || (n != 128 && n != 64) || !WaveNetSpecEnabled() || Tuning::Get().spSpb > 0) return hipErrorNotSupported;
		const int arch = groups[0].model->spec_arch;
		const bool lite = arch == WN_SPEC_LITE || arch == WN_SPEC_LITE16, a2 = arch == WN_SPEC_A2FULL || arch == WN_SPEC_A2LITE;
		if (arch != WN_SPEC_STD && !lite && !a2) return hipErrorNotSupported;
		bool packed = false;
		long streams = 0, slots2 = 0;
		for (int i = 0; i < numGroups; i++)
		{
			const int a = groups[i].model->spec_arch;
			const bool same = lite ? (a == WN_SPEC_LITE || a == WN_SPEC_LITE16) : (a2 ? (a == WN_SPEC_A2FULL || a == WN_SPEC_A2LITE) : a == WN_SPEC_STD);
			if (groups[i].numStreams <= 0 || !same) return hipErrorNotSupported;
			packed = packed || groups[i].pack > 1;
			const int per2 = a == WN_SPEC_A2LITE ? 4 : 2; // streams per full-size workgroup of this architecture
			streams += groups[i].numStreams;
			slots2 += (groups[i].numStreams + per2 - 1) / per2 * per2;
		}
		if (!lite && packed) return hipErrorNotSupported;
		if (packed)
			for (int i = 0; i < numGroups; i++)
				if (groups[i].slots == nullptr) return hipErrorNotSupported;
		if (!packed)
			for (int i = 0; i < numGroups; i++)
				if (groups[i].model->spec_arch == WN_SPEC_LITE16) return hipErrorNotSupported; // (16 / 16 only exists packed)
		const int spb = (slots2 * 4 > streams * 5) ? 1 : 2; // more than a quarter of the full-size workgroups' stream slots would idle
		if (a2) return spk::LaunchSpecA2Table(groups, numGroups, in, out, inStride, outStride, n, spb, stream, table);
		if (!lite)
		{
			if (n == 64) return spb == 2 ? spk::LaunchTable<spk::FamStd, 64, 2, false>(groups, numGroups, in, out, inStride, outStride, stream, table)
										 : spk::LaunchTable<spk::FamStd, 64, 1, false>(groups, numGroups, in, out, inStride, outStride, stream, table);
			return spb == 2 ? spk::LaunchTable<spk::FamStd, 128, 2, false>(groups, numGroups, in, out, inStride, outStride, stream, table)
							: spk::LaunchTable<spk::FamStd, 128, 1, false>(groups, numGroups, in, out, inStride, outStride, stream, table);
		}
		return spk::LaunchSpecLiteTable(groups, numGroups, in, out, inStride, outStride, n, spb, packed, stream, table);
#endif
	}

	// Runs the launch on a specialised chain when every group is the SAME official architecture and the block is 128 / 64 / 32 frames;
	// returns hipErrorNotSupported otherwise (the caller then uses the stage interpreter).
	hipError_t LaunchWaveNetSpecFused(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n,
		hipStream_t stream, int sharing)
	{
		if (numGroups <= 0 || numGroups > WN_FRAME_MAX_GROUPS || (n != 128 && n != 64 && n != 32)) return hipErrorNotSupported;
		if (!WaveNetSpecEnabled()) return hipErrorNotSupported; // tuning / tests: the interpreter for everything
		const int arch = groups[0].model->spec_arch;
		if (arch == WN_SPEC_NONE) return hipErrorNotSupported;
		auto familyOf = [](int a) { return (a == WN_SPEC_LITE || a == WN_SPEC_LITE16) ? 1 : ((a == WN_SPEC_A2FULL || a == WN_SPEC_A2LITE) ? 2 : 0); };
		const int fam = familyOf(arch);
		bool packed = false, lite16 = false;
		for (int i = 0; i < numGroups; i++)
		{
			const int a = groups[i].model->spec_arch;
			if (groups[i].numStreams <= 0 || a == WN_SPEC_NONE || familyOf(a) != fam || (fam == 0 && a != arch)) return hipErrorNotSupported;
			packed = packed || groups[i].pack > 1;
			lite16 = lite16 || a == WN_SPEC_LITE16;
		}
		// a packed launch reads the index lists of every group (a plain group riding along is pack = 1)
		if (packed)
			for (int i = 0; i < numGroups; i++)
				if (groups[i].slots == nullptr) return hipErrorNotSupported;
		// Streams per workgroup.  The half-size workgroups (SPB = 1: four waves) are compiled for two waves per SIMD, i.e. with 256 VGPRs
		// (NA_SPK_OCC): a launch that cannot fill the chip anyway is bound by the latency of its few waves, and the registers let the
		// compiler keep a layer's LDS reads in flight (Nano x 1024 = 256 packed streams 25.1 -> 22.6 us, Feather x 1024 24.3 -> 22.2,
		// Standard x 512 25.4 -> 23.4).  They are used while every workgroup is resident at that occupancy: at most two per CU --
		// counting the workgroups of the launches that share the chip with this one (two free-running half-batch chains of 512 Standard
		// streams each: 37.1 us per step with the half-size workgroups, 36.7 with the full-size ones; Feather x 1024 = 2 x 256: 21.5 vs 24.8).
		const int spbEnv = Tuning::Get().spSpb;
		const int residentHalf = 2 * CurrentDeviceCUs();
		int halfGroups = 0; // workgroups of the launch at SPB = 1 (an A2-Lite workgroup holds two streams there: T = 4)
		for (int i = 0; i < numGroups; i++)
		{
			const int per = groups[i].model->spec_arch == WN_SPEC_A2LITE ? 2 : 1;
			halfGroups += (groups[i].numStreams + per - 1) / per;
		}
		const bool beyondCache = (sharing & WN_SHARING_BEYOND_CACHE) != 0; // (the batch's stream state does not fit the Infinity Cache)
		sharing &= ~WN_SHARING_BEYOND_CACHE;
		const int spb = spbEnv > 0 ? spbEnv : (halfGroups * std::max(sharing, 1) > residentHalf ? 2 : 1);
#ifdef NA_SP_QUICK
		if (arch != WN_SPEC_STD || packed) return hipErrorNotSupported;
		return spk::LaunchNF<spk::FamStd, false>(groups, numGroups, in, out, inStride, outStride, n, spb, stream);
#else
		if (fam == 0) return packed ? hipErrorNotSupported : spk::LaunchNF<spk::FamStd, false>(groups, numGroups, in, out, inStride, outStride, n, spb, stream, beyondCache);
		if (fam == 2) return packed ? hipErrorNotSupported : spk::LaunchSpecA2(groups, numGroups, in, out, inStride, outStride, n, spb, stream, beyondCache);
		if (!packed && lite16) return hipErrorNotSupported; // (16 / 16 only exists packed)
		// a launch of 16 / 16 virtual streams only (packed Nano) with at most one of them per CU: one tile per wave, eight waves per stream
		// (Nano x 1024 = 256 virtual streams: a chain of 23 short stages, bound by what its few waves can issue)
		const bool noT1 = Tuning::Get().spNoT1; // tuning knob
		bool all16 = packed && !noT1;
		int virtualStreams = 0;
		for (int i = 0; i < numGroups; i++)
		{
			all16 = all16 && groups[i].model->spec_arch == WN_SPEC_LITE16;
			virtualStreams += groups[i].numStreams;
		}
		const bool t1 = all16 && spbEnv == 0 && virtualStreams <= CurrentDeviceCUs();
		return spk::LaunchSpecLite(groups, numGroups, in, out, inStride, outStride, n, spb, packed, stream, t1, beyondCache);
#endif
	}
}
