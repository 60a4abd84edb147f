// trapsts_probe.hip -- does a wave's sticky IEEE exception status (TRAPSTS.EXCP, bits 8:0: invalid, denormal, div0, OVERFLOW, underflow,
// inexact, ...) record an f32 -> f16 conversion overflow when the kernel runs with the default exception mask (no traps enabled)?
// It would be a free "a value left the f16 range" flag for the saturating split of the A2 chains (wavenet_split_dev.h).
//   hipcc --offload-arch=gfx950 -O3 -o bin/trapsts_probe trapsts_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void Probe(const float* in, unsigned* out)
{
	const float v = in[threadIdx.x];
	unsigned before, afterSmall, afterBig, afterManual;
	__builtin_amdgcn_s_setreg((3 | (0 << 6) | ((9 - 1) << 11)), 0);
	before = __builtin_amdgcn_s_getreg(3 | (0 << 6) | ((9 - 1) << 11));
	float t;
	asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(t) : "v"(v));        // 1.0 -> exact
	afterSmall = __builtin_amdgcn_s_getreg(3 | (0 << 6) | ((9 - 1) << 11));
	asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(t) : "v"(v * 1e6f)); // 1e6 -> inf: overflow + inexact
	asm volatile("s_nop 7\n\ts_nop 7");
	afterBig = __builtin_amdgcn_s_getreg(3 | (0 << 6) | ((9 - 1) << 11));
	__builtin_amdgcn_s_setreg((3 | (3 << 6) | ((1 - 1) << 11)), 1);       // set the overflow bit by hand
	afterManual = __builtin_amdgcn_s_getreg(3 | (0 << 6) | ((9 - 1) << 11));
	if (threadIdx.x == 0)
	{
		out[0] = before; out[1] = afterSmall; out[2] = afterBig; out[3] = afterManual;
		out[4] = __builtin_amdgcn_s_getreg(1 | (0 << 6) | ((32 - 1) << 11)); // MODE
		out[5] = __builtin_bit_cast(unsigned, t);
	}
}

int main()
{
	float* in; unsigned* out;
	(void)hipMalloc(&in, 256); (void)hipMalloc(&out, 64);
	float h[64]; for (int i = 0; i < 64; i++) h[i] = 1.0f;
	(void)hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(Probe, dim3(1), dim3(64), 0, 0, in, out);
	unsigned r[6];
	(void)hipMemcpy(r, out, 24, hipMemcpyDeviceToHost);
	printf("TRAPSTS.EXCP cleared %#x | after cvt(1.0) %#x | after cvt(1e6) %#x | after setting bit 3 by hand %#x | MODE %#x | cvt result bits %#x\n", r[0], r[1], r[2], r[3], r[4], r[5]);
	return 0;
}
