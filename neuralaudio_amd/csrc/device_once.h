// device_once.h -- per-device, thread-safe lazy initialisation for the kernel launchers.
//
// The launchers are reached concurrently from the worker threads of MultiGpuBatch (multi_gpu.cpp: one thread and one device per shard), and
// both things they set up lazily belong to ONE device: hipFuncSetAttribute applies to the current device's copy of the kernel, and the
// compute-unit count is the current device's.  So the "done" flags / cached counts are arrays indexed by device, read and written with
// atomics.  Running an initialiser twice (two threads racing on the same device) is harmless: both set the same attribute.
#pragma once

#include <atomic>

#include <hip/hip_runtime_api.h>

namespace na
{
	constexpr int kMaxHipDevices = 64;

	inline int CurrentHipDevice()
	{
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess) return -1;
		return dev;
	}

	struct PerDeviceOnce
	{
		std::atomic<unsigned char> done[kMaxHipDevices] = {};
		// f() -> hipError_t; runs until it has succeeded once on the current device
		template <class F>
		hipError_t Run(F&& f)
		{
			const int dev = CurrentHipDevice();
			if (dev < 0 || dev >= kMaxHipDevices) return f();
			if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
			const hipError_t e = f();
			if (e == hipSuccess) done[dev].store(1, std::memory_order_release);
			return e;
		}
	};

	// compute units of the current device (256 when the query fails: MI355X)
	inline int CurrentDeviceCUs()
	{
		static std::atomic<int> cached[kMaxHipDevices] = {};
		const int dev = CurrentHipDevice();
		if (dev >= 0 && dev < kMaxHipDevices)
		{
			const int c = cached[dev].load(std::memory_order_relaxed);
			if (c > 0) return c;
		}
		int cus = 0;
		if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
		if (dev >= 0 && dev < kMaxHipDevices) cached[dev].store(cus, std::memory_order_relaxed);
		return cus;
	}
}
