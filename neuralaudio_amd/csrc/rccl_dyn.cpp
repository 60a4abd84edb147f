// rccl_dyn.cpp -- see rccl_dyn.h
#include "rccl_dyn.h"

#include <dlfcn.h>

#include <atomic>
#include <mutex>

namespace na
{
	namespace rccl
	{
		namespace
		{
			const char* const kSymbols[] = { "ncclGetVersion", "ncclCommInitAll", "ncclCommDestroy", "ncclGetErrorString", "ncclBroadcast",
				"ncclAllGather", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd" };
			std::once_flag gOnce;
			Api gApi = {};
			bool gOk = false;
			std::string gError;

			void LoadOnce()
			{
				// the versioned name first (what the ROCm runtime packages install), then the development link, then the ROCm prefix
				const char* const candidates[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
				void* lib = nullptr;
				for (const char* name : candidates)
				{
					lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
					if (lib) break;
				}
				if (!lib)
				{
					const char* e = dlerror();
					gError = std::string("neuralaudio_amd: RCCL is not available (dlopen librccl.so.1: ") + (e ? e : "not found") + ")";
					return;
				}
				void* fn[sizeof(kSymbols) / sizeof(kSymbols[0])];
				for (size_t i = 0; i < sizeof(kSymbols) / sizeof(kSymbols[0]); i++)
				{
					fn[i] = dlsym(lib, kSymbols[i]);
					if (!fn[i])
					{
						gError = std::string("neuralaudio_amd: librccl.so lacks ") + kSymbols[i];
						return; // (the handle stays open: harmless, and nothing else will try again)
					}
				}
				gApi.GetVersion = reinterpret_cast<decltype(gApi.GetVersion)>(fn[0]);
				gApi.CommInitAll = reinterpret_cast<decltype(gApi.CommInitAll)>(fn[1]);
				gApi.CommDestroy = reinterpret_cast<decltype(gApi.CommDestroy)>(fn[2]);
				gApi.GetErrorString = reinterpret_cast<decltype(gApi.GetErrorString)>(fn[3]);
				gApi.Broadcast = reinterpret_cast<decltype(gApi.Broadcast)>(fn[4]);
				gApi.AllGather = reinterpret_cast<decltype(gApi.AllGather)>(fn[5]);
				gApi.Send = reinterpret_cast<decltype(gApi.Send)>(fn[6]);
				gApi.Recv = reinterpret_cast<decltype(gApi.Recv)>(fn[7]);
				gApi.GroupStart = reinterpret_cast<decltype(gApi.GroupStart)>(fn[8]);
				gApi.GroupEnd = reinterpret_cast<decltype(gApi.GroupEnd)>(fn[9]);
				gOk = true;
			}
		}

		const Api* Load(std::string& error)
		{
			std::call_once(gOnce, LoadOnce);
			if (!gOk)
			{
				error = gError;
				return nullptr;
			}
			return &gApi;
		}

		namespace
		{
			std::atomic<const Api*> gOverride{ nullptr };
		}
		void SetOverride(const Api* api) { gOverride.store(api); }
		bool OverrideActive() { return gOverride.load() != nullptr; }
		const Api* Active(std::string& error)
		{
			if (const Api* o = gOverride.load()) return o;
			return Load(error);
		}

		const char* const* SymbolNames(int& count)
		{
			count = (int)(sizeof(kSymbols) / sizeof(kSymbols[0]));
			return kSymbols;
		}
	}
}
