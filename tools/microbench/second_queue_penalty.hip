// second_queue_penalty.hip -- does a kernel chain on stream A slow down once the process has used a second / third HIP stream (its own
// hardware queue)?  Seen in round 4: the one-launch 1024 x Standard step went 40 -> 46 us as soon as the batch had recorded events on
// two more (idle) streams.  Here: a read-modify-write walk of a 240 MB set (Infinity Cache resident, ~75 us) and a compute-only kernel,
// back-to-back launches on A, timed (a) alone, (b) after creating B and C without using them, (c) after one event record on each,
// (d) after a kernel on each, (e) after destroying them.
//   hipcc --offload-arch=gfx950 -O3 -o bin/second_queue_penalty second_queue_penalty.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) Walk(u32x4* __restrict__ p, size_t quads)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < quads; i += 4 * stride)
	{
		u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
		a += 1u; b += 1u; c += 1u; d += 1u;
		p[i] = a; p[i + stride] = b; p[i + 2 * stride] = c; p[i + 3 * stride] = d;
	}
}

__global__ void __launch_bounds__(256) Spin(float* out, int trips)
{
	float a = threadIdx.x * 1e-3f, b = 1.0001f;
	for (int i = 0; i < trips; i++) { a = a * b + 0.5f; b = b * 0.9999f + 1e-4f; }
	if (a == 123.456f) out[0] = a + b;
}

__global__ void Tiny(float* out) { if (threadIdx.x == 9999) out[0] = 1.0f; }

static hipStream_t A;
static u32x4* buf; static float* sink;

static void Report(const char* what)
{
	const size_t quads = ((size_t)240 << 20) / 16;
	double us[2];
	for (int k = 0; k < 2; k++)
	{
		auto launch = [&] { if (k == 0) Walk<<<2048, 256, 0, A>>>(buf, quads); else Spin<<<2048, 256, 0, A>>>(sink, 6000); };
		for (int i = 0; i < 50; i++) launch();
		hipStreamSynchronize(A);
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < 400; i++) launch();
		hipStreamSynchronize(A);
		us[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 400 * 1e6;
	}
	printf("%-58s walk %7.2f us   spin %7.2f us\n", what, us[0], us[1]);
}

int main()
{
	hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
	hipMalloc(&buf, (size_t)240 << 20); hipMalloc(&sink, 64);
	hipMemset(buf, 0, (size_t)240 << 20);
	hipDeviceSynchronize();
	Report("A alone");
	Report("A alone (again)");
	hipStream_t B, C;
	hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
	hipStreamCreateWithFlags(&C, hipStreamNonBlocking);
	Report("B, C created, unused");
	hipEvent_t e; hipEventCreate(&e);
	hipEventRecord(e, B); hipEventSynchronize(e);
	Report("event recorded on B");
	hipEventRecord(e, C); hipEventSynchronize(e);
	Report("event recorded on B and C");
	Tiny<<<1, 64, 0, B>>>(sink); Tiny<<<1, 64, 0, C>>>(sink);
	hipDeviceSynchronize();
	Report("a kernel ran on B and C");
	hipStreamDestroy(B); hipStreamDestroy(C);
	Report("B, C destroyed");
	// the null stream as the second queue
	Tiny<<<1, 64>>>(sink);
	hipDeviceSynchronize();
	Report("a kernel ran on the null stream");
	return 0;
}
