// rccl_dyn.h -- RCCL (librccl.so, the ROCm build of the NCCL API) bound at RUN time with dlopen: the product library links only
// libamdhip64, single-GPU hosts never load RCCL, and the multi-GPU host (multi_gpu.cpp) uses it only when asked to
// (MultiGpuBatch::SetFanIn(FanIn::Rccl)): weight images replicated from the first device that holds a model over xGMI, and the
// shards' output rows gathered into one device buffer -- the "embarrassingly-parallel fan-out / fan-in" of BASELINE.json's north star.
// The data path between the kernels needs no collective (streams are independent).
//
// Declarations restate the public NCCL API (rccl.h of ROCm 7.2: ncclCommInitAll :236, ncclCommDestroy :260, ncclGetErrorString :339,
// ncclBroadcast :591, ncclAllGather :678, ncclSend / ncclRecv, ncclGroupStart :923, ncclGroupEnd :933; ncclDataType_t :459-466).
#pragma once

#include <cstddef>
#include <string>

#include <hip/hip_runtime_api.h>

namespace na
{
	namespace rccl
	{
		typedef struct ncclComm* Comm;
		typedef int Result;                         // ncclResult_t: 0 = ncclSuccess
		enum DataType { kUint8 = 1, kFloat32 = 7 }; // ncclUint8, ncclFloat32

		struct Api
		{
			Result (*GetVersion)(int* version);
			Result (*CommInitAll)(Comm* comms, int ndev, const int* devlist);
			Result (*CommDestroy)(Comm comm);
			const char* (*GetErrorString)(Result r);
			Result (*Broadcast)(const void* sendbuff, void* recvbuff, size_t count, int datatype, int root, Comm comm, hipStream_t stream);
			Result (*AllGather)(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, Comm comm, hipStream_t stream);
			Result (*Send)(const void* sendbuff, size_t count, int datatype, int peer, Comm comm, hipStream_t stream);
			Result (*Recv)(void* recvbuff, size_t count, int datatype, int peer, Comm comm, hipStream_t stream);
			Result (*GroupStart)();
			Result (*GroupEnd)();
		};

		// The process-wide binding: nullptr (and `error` says why) when librccl.so cannot be loaded or lacks a symbol.  Thread-safe.
		const Api* Load(std::string& error);
		// The table the multi-GPU host uses: the override if one is set, else Load().  The one override that exists is LoopbackApi()
		// (rccl_loopback.cpp): every rank on the same device, transfers are device-to-device copies -- so that the multi-rank
		// orchestration executes on a one-GPU box (tests; NA_DebugSetRcclApi).  Not for use while a multi batch is being committed.
		const Api* Active(std::string& error);
		void SetOverride(const Api* api); // nullptr: back to librccl.so
		bool OverrideActive();
		const Api* LoopbackApi();
		void LoopbackConfigure(int failSendAt, int rendezvousMs); // fault injection / rendezvous time-out of the loopback table (tests)
		// names of the symbols Load() resolves (tests check them against the installed library without a GPU)
		const char* const* SymbolNames(int& count);
	}
}
