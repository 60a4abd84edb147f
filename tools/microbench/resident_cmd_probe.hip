// resident_cmd_probe.hip -- what does the command protocol of a resident ("persistent") launch cost per command on gfx950, apart from the
// work itself?  512 workgroups (two per CU, like the headline chain) loop over commands the host posts into a ring in pinned, coherent
// host memory; a workgroup's "work" is a spin of `work` ticks (s_memrealtime, 100 MHz).  Ways to FETCH a command:
//   F0  every workgroup polls the host ring over PCIe (seq, then four fields) at system scope
//   F1  workgroup 0 polls the host ring and forwards the command into a ring in device memory; the others poll that (agent scope)
//   F2  every workgroup polls the device ring; the host itself writes it (fine-grained device memory mapped into the host, if it maps)
// and to COUNT a command done:
//   C0  one counter per command slot, fetch_add by every workgroup at agent scope; the last one stores `completed` to the host
//   C1  eight counters per slot (workgroup % 8 = its XCD), the last of each bumps a top counter, the last of those stores `completed`
//   C2  no counting: a workgroup stores its own progress word; workgroup 0 (one wave) scans the 512 words when it comes by
//   hipcc --offload-arch=gfx950 -O3 -o bin/resident_cmd_probe resident_cmd_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int RING = 64;
struct Cmd { unsigned long long f[4]; unsigned long long seq; unsigned long long pad[3]; };
struct Ctrl { unsigned long long exitAfter, pad0[7], completed, pad1[7]; Cmd cmd[RING]; };

struct Args
{
	Ctrl* host;              // pinned host block (device address)
	Cmd* dev;                // device-memory ring (F1 / F2)
	unsigned* count;         // [RING][16]
	unsigned* progress;      // [grid]
	unsigned long long* sink;
	unsigned long long last; // commands to run
	int fetch, done, work;
};

template <typename T> __device__ T LoadSys(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <typename T> __device__ T LoadDev(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(512) Resident(const Args a)
{
	__shared__ unsigned long long bc[8];
	unsigned long long acc = 0;
	for (unsigned long long k = 1; k <= a.last; k++)
	{
		if (threadIdx.x == 0)
		{
			const unsigned slot = (unsigned)(k % RING);
			if (a.fetch == 0 || (a.fetch == 1 && blockIdx.x == 0))
			{
				const Cmd* c = &a.host->cmd[slot];
				while (LoadSys(&c->seq) != k) __builtin_amdgcn_s_sleep(8);
				unsigned long long f0 = LoadSys(&c->f[0]), f1 = LoadSys(&c->f[1]), f2 = LoadSys(&c->f[2]), f3 = LoadSys(&c->f[3]);
				if (a.fetch == 1)
				{
					Cmd* d = &a.dev[slot];
					__hip_atomic_store(&d->f[0], f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(&d->f[1], f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(&d->f[2], f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(&d->f[3], f3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(&d->seq, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
				}
				acc += f0 + f1 + f2 + f3;
			}
			else
			{
				const Cmd* d = &a.dev[slot];
				while (LoadDev(&d->seq) != k) __builtin_amdgcn_s_sleep(4);
				acc += LoadDev(&d->f[0]) + LoadDev(&d->f[1]) + LoadDev(&d->f[2]) + LoadDev(&d->f[3]);
			}
			bc[0] = acc;
		}
		__syncthreads();
		acc += bc[0];
		if (a.work > 0)
		{
			const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
			while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.work) __builtin_amdgcn_s_sleep(16);
		}
		__syncthreads();
		if (threadIdx.x == 0)
		{
			const unsigned slot = (unsigned)(k % RING);
			if (a.done == 0)
			{
				const unsigned before = __hip_atomic_fetch_add(&a.count[slot * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (before == gridDim.x - 1)
				{
					__hip_atomic_store(&a.count[slot * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(&a.host->completed, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			}
			else if (a.done == 1)
			{
				const unsigned x = blockIdx.x & 7u, members = (gridDim.x - x + 7u) / 8u;
				const unsigned before = __hip_atomic_fetch_add(&a.count[slot * 16 + 1 + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (before == members - 1)
				{
					__hip_atomic_store(&a.count[slot * 16 + 1 + x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const unsigned top = __hip_atomic_fetch_add(&a.count[slot * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (top == (gridDim.x < 8u ? gridDim.x : 8u) - 1)
					{
						__hip_atomic_store(&a.count[slot * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						__hip_atomic_store(&a.host->completed, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					}
				}
			}
			else
				__hip_atomic_store(&a.progress[blockIdx.x], (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (a.done == 2 && blockIdx.x == 0 && threadIdx.x < 64)
		{
			// one wave scans everybody's progress word: the minimum is what is complete
			unsigned m = ~0u;
			for (unsigned i = threadIdx.x; i < gridDim.x; i += 64) { const unsigned v = LoadDev(&a.progress[i]); m = v < m ? v : m; }
			for (int o = 32; o > 0; o >>= 1) { const unsigned v = (unsigned)__shfl_xor((int)m, o); m = v < m ? v : m; }
			if (threadIdx.x == 0 && m > 0) __hip_atomic_store(&a.host->completed, (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	if (a.done == 2 && blockIdx.x == 0 && threadIdx.x == 0)
	{
		// closing scan: wait for the others
		for (;;)
		{
			unsigned m = ~0u;
			for (unsigned i = 0; i < gridDim.x; i++) { const unsigned v = LoadDev(&a.progress[i]); m = v < m ? v : m; }
			__hip_atomic_store(&a.host->completed, (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (m >= a.last) break;
			__builtin_amdgcn_s_sleep(32);
		}
	}
	if (acc == 0x1234567) a.sink[0] = acc;
}

int main(int argc, char** argv)
{
	const int grid = argc > 1 ? atoi(argv[1]) : 512;
	const int N = argc > 2 ? atoi(argv[2]) : 2000;
	const bool tryFine = argc > 3; // (a host write to memory that is not mapped into the host ends the process: its own run)
	Ctrl* host; Ctrl* dHost;
	CHECK(hipHostMalloc((void**)&host, sizeof(Ctrl), hipHostMallocMapped | hipHostMallocCoherent));
	CHECK(hipHostGetDevicePointer((void**)&dHost, host, 0));
	Cmd* dev; unsigned* count; unsigned* progress; unsigned long long* sink;
	// F2: fine-grained device memory the host can write through the BAR
	Cmd* devFine = nullptr;
	const bool fineOk = hipExtMallocWithFlags((void**)&devFine, sizeof(Cmd) * RING, hipDeviceMallocFinegrained) == hipSuccess;
	CHECK(hipMalloc((void**)&dev, sizeof(Cmd) * RING));
	CHECK(hipMalloc((void**)&count, RING * 16 * 4));
	CHECK(hipMalloc((void**)&progress, 4096 * 4));
	CHECK(hipMalloc((void**)&sink, 8));
	hipStream_t s;
	CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	printf("grid %d, %d commands per run; fine-grained device ring: %s\n", grid, N, fineOk ? "allocated" : "no");
	for (int work : {0, 3000})
		for (int fetch = 0; fetch < 3; fetch++)
			for (int done = 0; done < 3; done++)
			{
				if ((fetch == 2) != tryFine || (fetch == 2 && !fineOk)) continue;
				Cmd* ring = fetch == 2 ? devFine : dev;
				memset(host, 0, sizeof(Ctrl));
				CHECK(hipMemset(ring, 0, sizeof(Cmd) * RING));
				CHECK(hipMemset(count, 0, RING * 16 * 4));
				CHECK(hipMemset(progress, 0, 4096 * 4));
				CHECK(hipDeviceSynchronize());
				Args a = {dHost, ring, count, progress, sink, (unsigned long long)N, fetch, done, work};
				const auto t0 = std::chrono::steady_clock::now();
				hipLaunchKernelGGL(Resident, dim3(grid), dim3(512), 0, s, a);
				bool hostWriteFailed = false;
				for (unsigned long long k = 1; k <= (unsigned long long)N; k++)
				{
					while (k - __atomic_load_n(&host->completed, __ATOMIC_ACQUIRE) >= RING - 1) { }
					Cmd* c = fetch == 2 ? &devFine[k % RING] : &host->cmd[k % RING];
					c->f[0] = k; c->f[1] = 2 * k; c->f[2] = 3; c->f[3] = 4;
					__atomic_store_n(&c->seq, k, __ATOMIC_RELEASE);
				}
				CHECK(hipStreamSynchronize(s));
				const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
				printf("work %5.1f us  fetch F%d  done C%d : %8.2f us per command (completed %llu)%s\n", work / 100.0, fetch, done, us / N,
					(unsigned long long)host->completed, hostWriteFailed ? " host write failed" : "");
				fflush(stdout);
			}
	return 0;
}
