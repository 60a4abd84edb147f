// Probe the operand / result layouts of v_mfma_f32_16x16x1_4b_f32 and v_mfma_f32_4x4x1_16b_f32 on gfx950.
// Hypothesis (lane = frame mapping used by wavenet kernels):
//   16x16x1_4b : lane l supplies A[block l/16][row l%16], B[block l/16][col l%16];  D: 16 VGPRs, lane l holds D_block(l/16)[row r][col l%16] in VGPR r
//   4x4x1_16b  : lane l supplies A[block l/4][row l%4],  B[block l/4][col l%4];    D: 4 VGPRs,  lane l holds D_block(l/4)[row r][col l%4]  in VGPR r
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float* out16, float* out4)
{
	const int l = threadIdx.x;
	const float a = 1.0f + l;          // A value of this lane
	const float b = 100.0f + 3.0f * l; // B value of this lane
	f32x16 c16;
	for (int i = 0; i < 16; i++) c16[i] = 0.0f;
	c16 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c16, 0, 0, 0);
	for (int r = 0; r < 16; r++) out16[l * 16 + r] = c16[r];
	f32x4 c4 = {0, 0, 0, 0};
	c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0);
	for (int r = 0; r < 4; r++) out4[l * 4 + r] = c4[r];
}

int main()
{
	float *d16, *d4, h16[64 * 16], h4[64 * 4];
	hipMalloc(&d16, sizeof(h16)); hipMalloc(&d4, sizeof(h4));
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d16, d4);
	hipMemcpy(h16, d16, sizeof(h16), hipMemcpyDeviceToHost); hipMemcpy(h4, d4, sizeof(h4), hipMemcpyDeviceToHost);
	int bad16 = 0, bad4 = 0;
	for (int l = 0; l < 64; l++)
		for (int r = 0; r < 16; r++)
		{
			// expected: A of lane (block*16 + r) times B of lane l
			const int blk = l / 16;
			const float want = (1.0f + (blk * 16 + r)) * (100.0f + 3.0f * l);
			if (h16[l * 16 + r] != want) { if (bad16 < 4) printf("16x16x1: lane %d reg %d got %g want %g\n", l, r, h16[l * 16 + r], want); bad16++; }
		}
	for (int l = 0; l < 64; l++)
		for (int r = 0; r < 4; r++)
		{
			const int blk = l / 4;
			const float want = (1.0f + (blk * 4 + r)) * (100.0f + 3.0f * l);
			if (h4[l * 4 + r] != want) { if (bad4 < 4) printf("4x4x1: lane %d reg %d got %g want %g\n", l, r, h4[l * 4 + r], want); bad4++; }
		}
	printf("16x16x1_4b mismatches: %d   4x4x1_16b mismatches: %d\n", bad16, bad4);
	printf("16x16x1 lane 17 regs: "); for (int r = 0; r < 16; r++) printf("%g ", h16[17 * 16 + r]); printf("\n");
	printf("4x4x1 lane 5 regs: "); for (int r = 0; r < 4; r++) printf("%g ", h4[5 * 4 + r]); printf("\n");
	return 0;
}
