// rccl_loopback.cpp -- a loopback implementation of the rccl::Api table (rccl_dyn.h) for boxes with ONE GPU: every "rank" lives on the
// same device and a transfer between two ranks is a device-to-device hipMemcpyAsync between their buffers, ordered against both ranks'
// streams with events.  It exists so that the multi-rank orchestration of the multi-GPU host (multi_gpu.cpp: communicator set-up, the
// weight fan-out with ncclSend / ncclRecv, the all-gather of unequal parts with one ncclBroadcast per shard, failure teardown) EXECUTES
// on the hardware this project is tested on; with two or more GPUs the same host code runs over librccl.so.  Selected by
// NA_DebugSetRcclApi(1) (tests only); it moves no byte over xGMI and says nothing about RCCL's performance.
//
// Semantics restated from the public NCCL API (rccl.h of ROCm 7.2: ncclSend / ncclRecv match in posting order per (sender, receiver)
// pair; ncclBroadcast :591 copies `count` elements from the root's sendbuff to every rank's recvbuff; operations between
// ncclGroupStart :923 and ncclGroupEnd :933 are issued together at the closing call, which is where this implementation rendezvouses).
#include "rccl_dyn.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace na
{
	namespace rccl
	{
		namespace
		{
			constexpr Result kOk = 0, kInternalError = 3, kInvalidArgument = 4; // ncclSuccess, ncclInternalError, ncclInvalidArgument

			size_t TypeBytes(int datatype) { return datatype == kFloat32 ? 4 : 1; }

			// one posted transfer: the sender's buffer and an event on the sender's stream behind which the bytes are final
			struct Parcel
			{
				const void* src = nullptr;
				size_t bytes = 0;
				hipEvent_t ready = nullptr;    // recorded on the sender's stream when the parcel is posted
				hipEvent_t consumed = nullptr; // recorded on the receiver's stream behind its copy
				bool taken = false;            // the receiver has enqueued its copy (consumed is recorded)
			};

			struct World // the ranks of one ncclCommInitAll call
			{
				int n = 0;
				std::mutex m;
				std::condition_variable cv;
				std::map<std::pair<int, int>, std::deque<std::shared_ptr<Parcel>>> mail; // (sender, receiver) -> parcels in posting order
				bool aborted = false;
			};

			struct LoopComm
			{
				std::shared_ptr<World> world;
				int rank = 0;
			};

			struct Op
			{
				enum Kind { SEND, RECV } kind;
				LoopComm* comm;
				int peer;
				const void* src;
				void* dst;
				size_t bytes;
				hipStream_t stream;
			};

			thread_local int tGroupDepth = 0;
			thread_local std::vector<Op> tOps;

			// a rank waits this long for its peer to post before it gives up (a peer that failed before or inside its group: multi_gpu.cpp
			// tears the shards down / marks the object broken) -- the real library would hang there
			std::atomic<int> gRendezvousMs{ 20000 };
			// fault injection (tests of the failure paths): the gFailSendAt-th ncclSend of the process fails (0: never)
			std::atomic<int> gFailSendAt{ 0 }, gSends{ 0 };

			Result Execute(std::vector<Op>& ops)
			{
				Result result = kOk;
				struct Posted { std::shared_ptr<Parcel> parcel; hipStream_t stream; World* world; };
				std::vector<Posted> posted;
				// 1. post everything this rank sends
				for (const Op& op : ops)
				{
					if (op.kind != Op::SEND) continue;
					auto p = std::make_shared<Parcel>();
					p->src = op.src;
					p->bytes = op.bytes;
					if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->consumed, hipEventDisableTiming) != hipSuccess) return kInternalError;
					if (hipEventRecord(p->ready, op.stream) != hipSuccess) return kInternalError;
					World& w = *op.comm->world;
					{
						std::lock_guard<std::mutex> lock(w.m);
						w.mail[{ op.comm->rank, op.peer }].push_back(p);
					}
					w.cv.notify_all();
					posted.push_back({ p, op.stream, &w });
				}
				// 2. take everything this rank receives, in posting order per sender
				for (const Op& op : ops)
				{
					if (op.kind != Op::RECV) continue;
					World& w = *op.comm->world;
					std::shared_ptr<Parcel> p;
					{
						std::unique_lock<std::mutex> lock(w.m);
						auto& q = w.mail[{ op.peer, op.comm->rank }];
						if (!w.cv.wait_for(lock, std::chrono::milliseconds(gRendezvousMs.load()), [&] { return !q.empty() || w.aborted; }) || w.aborted)
						{
							w.aborted = true;
							w.cv.notify_all();
							result = kInternalError;
							continue;
						}
						p = q.front();
						q.pop_front();
					}
					if (p->bytes != op.bytes) result = kInvalidArgument;
					else if (hipStreamWaitEvent(op.stream, p->ready, 0) != hipSuccess || hipMemcpyAsync(op.dst, p->src, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) result = kInternalError;
					(void)hipEventRecord(p->consumed, op.stream);
					{
						std::lock_guard<std::mutex> lock(w.m);
						p->taken = true;
					}
					w.cv.notify_all();
				}
				// 3. the sender's stream goes on only behind the receivers' copies (its buffers may be rewritten by the next launch)
				for (auto& ps : posted)
				{
					Parcel& p = *ps.parcel;
					World* w = ps.world;
					std::unique_lock<std::mutex> lock(w->m);
					if (!w->cv.wait_for(lock, std::chrono::milliseconds(gRendezvousMs.load()), [&] { return p.taken || w->aborted; }) || w->aborted)
					{
						w->aborted = true;
						w->cv.notify_all();
						result = kInternalError;
						continue;
					}
					lock.unlock();
					if (hipStreamWaitEvent(ps.stream, p.consumed, 0) != hipSuccess) result = kInternalError;
				}
				// (events are destroyed once nothing refers to them: HIP defers the destruction of an event with pending work)
				for (auto& ps : posted)
				{
					if (!ps.parcel->taken) continue;
					(void)hipEventDestroy(ps.parcel->ready);
					(void)hipEventDestroy(ps.parcel->consumed);
				}
				return result;
			}

			Result Enqueue(const Op& op)
			{
				tOps.push_back(op);
				if (tGroupDepth > 0) return kOk;
				std::vector<Op> ops;
				ops.swap(tOps);
				return Execute(ops);
			}

			Result LbGetVersion(int* version)
			{
				if (version) *version = 0; // (not a librccl build)
				return kOk;
			}

			Result LbCommInitAll(Comm* comms, int ndev, const int* devlist)
			{
				(void)devlist; // any device list, repeated indices included: every rank's buffers live wherever its shard put them
				if (!comms || ndev < 1) return kInvalidArgument;
				auto world = std::make_shared<World>();
				world->n = ndev;
				for (int i = 0; i < ndev; i++)
				{
					LoopComm* c = new LoopComm();
					c->world = world;
					c->rank = i;
					comms[i] = reinterpret_cast<Comm>(c);
				}
				return kOk;
			}

			Result LbCommDestroy(Comm comm)
			{
				delete reinterpret_cast<LoopComm*>(comm);
				return kOk;
			}

			const char* LbGetErrorString(Result r) { return r == kOk ? "no error" : (r == kInvalidArgument ? "loopback: invalid argument / size mismatch" : "loopback: a peer never posted its side"); }

			Result LbSend(const void* sendbuff, size_t count, int datatype, int peer, Comm comm, hipStream_t stream)
			{
				LoopComm* c = reinterpret_cast<LoopComm*>(comm);
				if (!c || peer < 0 || peer >= c->world->n || peer == c->rank) return kInvalidArgument;
				if (gFailSendAt.load() > 0 && ++gSends == gFailSendAt.load()) return kInternalError; // (injected)
				return Enqueue({ Op::SEND, c, peer, sendbuff, nullptr, count * TypeBytes(datatype), stream });
			}

			Result LbRecv(void* recvbuff, size_t count, int datatype, int peer, Comm comm, hipStream_t stream)
			{
				LoopComm* c = reinterpret_cast<LoopComm*>(comm);
				if (!c || peer < 0 || peer >= c->world->n || peer == c->rank) return kInvalidArgument;
				return Enqueue({ Op::RECV, c, peer, nullptr, recvbuff, count * TypeBytes(datatype), stream });
			}

			Result LbGroupStart()
			{
				tGroupDepth++;
				return kOk;
			}

			Result LbGroupEnd()
			{
				if (tGroupDepth <= 0) return kInvalidArgument;
				if (--tGroupDepth > 0) return kOk;
				std::vector<Op> ops;
				ops.swap(tOps);
				return Execute(ops);
			}

			// root -> every other rank; the root's own sendbuff -> recvbuff copy when they differ
			Result LbBroadcast(const void* sendbuff, void* recvbuff, size_t count, int datatype, int root, Comm comm, hipStream_t stream)
			{
				LoopComm* c = reinterpret_cast<LoopComm*>(comm);
				if (!c || root < 0 || root >= c->world->n) return kInvalidArgument;
				const size_t bytes = count * TypeBytes(datatype);
				if (c->rank != root) return Enqueue({ Op::RECV, c, root, nullptr, recvbuff, bytes, stream });
				if (sendbuff != recvbuff && hipMemcpyAsync(recvbuff, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return kInternalError;
				tGroupDepth++; // (the n - 1 sends of one broadcast are one group)
				Result r = kOk;
				for (int peer = 0; peer < c->world->n; peer++)
					if (peer != root)
					{
						const Result e = Enqueue({ Op::SEND, c, peer, sendbuff, nullptr, bytes, stream });
						if (e != kOk) r = e;
					}
				const Result e = LbGroupEnd();
				return r != kOk ? r : e;
			}

			Result LbAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, Comm comm, hipStream_t stream)
			{
				LoopComm* c = reinterpret_cast<LoopComm*>(comm);
				if (!c) return kInvalidArgument;
				const size_t bytes = sendcount * TypeBytes(datatype);
				tGroupDepth++;
				Result r = kOk;
				for (int root = 0; root < c->world->n; root++)
				{
					char* part = static_cast<char*>(recvbuff) + (size_t)root * bytes;
					const Result e = LbBroadcast(root == c->rank ? sendbuff : part, part, sendcount, datatype, root, comm, stream);
					if (e != kOk) r = e;
				}
				const Result e = LbGroupEnd();
				return r != kOk ? r : e;
			}

			const Api kLoopback = { LbGetVersion, LbCommInitAll, LbCommDestroy, LbGetErrorString, LbBroadcast, LbAllGather, LbSend, LbRecv, LbGroupStart, LbGroupEnd };
		}

		const Api* LoopbackApi() { return &kLoopback; }
		void LoopbackConfigure(int failSendAt, int rendezvousMs)
		{
			gSends.store(0);
			gFailSendAt.store(failSendAt > 0 ? failSendAt : 0);
			gRendezvousMs.store(rendezvousMs > 0 ? rendezvousMs : 20000);
		}
	}
}
