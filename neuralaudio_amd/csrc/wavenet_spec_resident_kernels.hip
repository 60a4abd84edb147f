// wavenet_spec_resident_kernels.hip -- the resident ("persistent") launch of the specialised A1 Standard chain (wavenet_spec_impl.h
// WaveNetSpecResidentKernel; protocol: wavenet_launch.h ResidentCtrl; host side: gpu_batch_chains.cpp).  Its own translation unit because
// it is compiled WITHOUT machine-level loop-invariant code motion (Makefile): the chain sits in the launch's command loop, and LICM
// hoists ~30 scalar and ~8 vector constants of the unrolled chain (LDS offsets, tanh coefficients, zero accumulators) out of that loop
// and keeps them live across all of it -- 106 SGPRs and 11 spills, three of them vector memory operations in the middle of the chain's
// counted vmcnt waits (one-shot kernel: 69 SGPRs, none).  See wavenet_spec_kernels.hip for the chains themselves.
#include "wavenet_spec_impl.h"

namespace na
{
	// The resident launch (wavenet_launch.h ResidentCtrl) of a list that is ONE launch of full-size workgroups: 128-frame blocks of A1
	// Standard streams (the headline workload; the other families keep their ordinary launches for now).  stream == nullptr: grid only.
	static hipError_t ResidentDispatch(const WnFrameGroup* groups, int numGroups, int n, const ResidentArgs& ra, hipStream_t stream, int* gridOut)
	{
		*gridOut = 0;
		if (numGroups <= 0 || numGroups > WN_FRAME_MAX_GROUPS || n != 128 || !WaveNetSpecEnabled()) return hipErrorNotSupported;
		for (int i = 0; i < numGroups; i++)
			if (groups[i].model->spec_arch != WN_SPEC_STD || groups[i].pack > 1 || groups[i].numStreams <= 0) return hipErrorNotSupported;
		return spk::LaunchResident<spk::FamStd, 128, 2, false>(groups, numGroups, ra, stream, gridOut);
	}
	hipError_t LaunchWaveNetSpecResident(const WnFrameGroup* groups, int numGroups, int n, const ResidentArgs& ra, hipStream_t stream, int* gridOut)
	{
		if (stream == nullptr) return hipErrorInvalidValue; // (the null stream would serialise the launch behind everything)
		return ResidentDispatch(groups, numGroups, n, ra, stream, gridOut);
	}
	int WaveNetSpecResidentGrid(const WnFrameGroup* groups, int numGroups, int n)
	{
		int grid = 0;
		ResidentArgs none = {};
		return ResidentDispatch(groups, numGroups, n, none, nullptr, &grid) == hipSuccess ? grid : 0;
	}

}
