// wavenet_spec_lite_kernels.hip -- the specialised chains of the "lite" dilation lists (A1 Lite padded to 16 / 8 channels; two Feather or
// four Nano streams packed into one virtual stream): kernels of FamLite and FamLitePacked.  See wavenet_spec_kernels.hip.
#include "wavenet_spec_impl.h"

namespace na
{
	namespace spk
	{
		hipError_t LaunchSpecLite(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, bool packed,
			hipStream_t stream, bool oneTilePerWave, bool beyondCache)
		{
#ifdef NA_SP_QUICK
			return hipErrorNotSupported;
#else
			// (16 / 16 packed streams, at most one per CU: eight waves of one tile each, see ArchLite16T1)
			if (packed && oneTilePerWave) return LaunchNF<FamLite16T1, true>(groups, numGroups, in, out, inStride, outStride, n, 2, stream);
			if (packed) return LaunchNF<FamLitePacked, true>(groups, numGroups, in, out, inStride, outStride, n, spb, stream, beyondCache);
			return LaunchNF<FamLite, false>(groups, numGroups, in, out, inStride, outStride, n, spb, stream, beyondCache);
#endif
		}

		// the same families as table launches (wavenet_launch.h LaunchWaveNetSpecTable): 128-frame blocks
		hipError_t LaunchSpecLiteTable(const WnFrameGroup* groups, int numGroups, const float* in, float* out, long inStride, long outStride, int n, int spb, bool packed,
			hipStream_t stream, WnLaunchTable& table)
		{
#ifdef NA_SP_QUICK
			return hipErrorNotSupported;
#else
#define NA_LITE_TABLE(NFR) \
			if (packed) \
				return spb >= 2 ? LaunchTable<FamLitePacked, NFR, 2, true>(groups, numGroups, in, out, inStride, outStride, stream, table) \
								: LaunchTable<FamLitePacked, NFR, 1, true>(groups, numGroups, in, out, inStride, outStride, stream, table); \
			return spb >= 2 ? LaunchTable<FamLite, NFR, 2, false>(groups, numGroups, in, out, inStride, outStride, stream, table) \
							: LaunchTable<FamLite, NFR, 1, false>(groups, numGroups, in, out, inStride, outStride, stream, table);
			if (n == 64) { NA_LITE_TABLE(64) }
			NA_LITE_TABLE(128)
#undef NA_LITE_TABLE
#endif
		}
	}
}
