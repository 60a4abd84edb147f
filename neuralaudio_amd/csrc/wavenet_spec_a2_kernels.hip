// wavenet_spec_a2_kernels.hip -- the specialised chains of the A2 slimmable model (NeuralModel.cpp:389-421: 23 layers of kernel size 6 / 15,
// conv head of 16 taps, LeakyReLU; 8 or 3 -> 4 channels): kernels of FamA2.  See wavenet_spec_kernels.hip.  (Compiled without
// -amdgpu-use-amdgpu-trackers: these long stages schedule better with the generic pressure trackers, 49.2 vs 50.3 us for A2 "Full".)
#include "wavenet_spec_impl.h"

namespace na
{
	namespace spk
	{
		hipError_t LaunchSpecA2(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, hipStream_t stream, bool beyondCache)
		{
#ifdef NA_SP_QUICK
			return hipErrorNotSupported;
#else
			return LaunchNF<FamA2, false>(groups, numGroups, in, out, inStride, outStride, n, spb, stream, beyondCache);
#endif
		}

		// ... as a table launch (wavenet_launch.h LaunchWaveNetSpecTable): 128-frame blocks
		hipError_t LaunchSpecA2Table(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, hipStream_t stream,
			WnLaunchTable& table)
		{
#ifdef NA_SP_QUICK
			return hipErrorNotSupported;
#else
			if (n == 64)
				return spb >= 2 ? LaunchTable<FamA2, 64, 2, false>(groups, numGroups, in, out, inStride, outStride, stream, table)
								: LaunchTable<FamA2, 64, 1, false>(groups, numGroups, in, out, inStride, outStride, stream, table);
			return spb >= 2 ? LaunchTable<FamA2, 128, 2, false>(groups, numGroups, in, out, inStride, outStride, stream, table)
							: LaunchTable<FamA2, 128, 1, false>(groups, numGroups, in, out, inStride, outStride, stream, table);
#endif
		}
	}
}
