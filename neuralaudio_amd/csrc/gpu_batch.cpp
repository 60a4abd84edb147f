// gpu_batch.cpp -- see gpu_batch.h.  Host side only: device memory, tables, launches.
#include "gpu_batch.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <hip/hip_runtime.h>

#include "lstm_launch.h"
#include "wavenet_launch.h"
#include "wavenet_plan.h"

namespace na
{
	void CheckHip(hipError_t e, const char* what)
	{
		if (e != hipSuccess) throw HipError(e, what);
	}

	int VisibleDeviceCount()
	{
		int count = 0;
		const hipError_t e = hipGetDeviceCount(&count);
		if (e != hipSuccess)
		{
			(void)hipGetLastError();
			return 0;
		}
		return count;
	}

	namespace
	{
		template <typename T>
		class DevArray
		{
		public:
			DevArray() = default;
			~DevArray() { Free(); }
			DevArray(const DevArray&) = delete;
			DevArray& operator=(const DevArray&) = delete;

			void Alloc(size_t n)
			{
				Free();
				if (n == 0) return;
				CheckHip(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)), "hipMalloc");
				count = n;
			}

			void Upload(const std::vector<T>& host, hipStream_t s)
			{
				if (host.size() > count) Alloc(std::max(host.size(), count * 2));
				if (!host.empty())
				{
					CheckHip(hipMemcpyAsync(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, s), "hipMemcpyAsync H2D");
					// host vectors are pageable and may be reused right away
					CheckHip(hipStreamSynchronize(s), "hipStreamSynchronize");
				}
			}

			void Free()
			{
				if (ptr) (void)hipFree(ptr);
				ptr = nullptr;
				count = 0;
			}

			void Swap(DevArray& o)
			{
				std::swap(ptr, o.ptr);
				std::swap(count, o.count);
			}

			T* Get() const { return ptr; }
			size_t Count() const { return count; }

		private:
			T* ptr = nullptr;
			size_t count = 0;
		};
	}

	// ------------------------------------------------------------------------------------------ groups

	class ModelGroup
	{
	public:
		ModelGroup(const std::shared_ptr<const ModelDesc>& d, hipStream_t s) : desc(d), stream(s) {}
		virtual ~ModelGroup()
		{
			if (sideStream) (void)hipStreamDestroy(sideStream);
			if (doneEvent) (void)hipEventDestroy(doneEvent);
			for (int b = 0; b < 2; b++)
			{
				if (pinnedLists[b]) (void)hipHostFree(pinnedLists[b]);
				if (listEvent[b]) (void)hipEventDestroy(listEvent[b]);
			}
		}

		// created on first use: lets independent model groups of a mixed batch run concurrently
		hipStream_t SideStream()
		{
			if (!sideStream) CheckHip(hipStreamCreateWithFlags(&sideStream, hipStreamNonBlocking), "hipStreamCreate");
			return sideStream;
		}

		hipEvent_t DoneEvent()
		{
			if (!doneEvent) CheckHip(hipEventCreateWithFlags(&doneEvent, hipEventDisableTiming), "hipEventCreate");
			return doneEvent;
		}

		const std::shared_ptr<const ModelDesc> desc;

		// a state slot for a new stream: the lowest freed one, else a new one
		int AddMember()
		{
			if (!freeMembers.empty())
			{
				const int member = freeMembers.front();
				freeMembers.erase(freeMembers.begin());
				memberInUse[(size_t)member] = 1;
				return member;
			}
			const int member = (int)memberRow.size();
			EnsureCapacity(member + 1);
			EnsureListCapacity((size_t)member + 1);
			memberRow.push_back(-1);
			memberInUse.push_back(1);
			return member;
		}

		// the stream is gone: its slot goes inactive and may be handed to a later AddMember (which resets it)
		void RemoveMember(int member)
		{
			SetActive(member, -1);
			memberInUse[(size_t)member] = 0;
			freeMembers.insert(std::lower_bound(freeMembers.begin(), freeMembers.end(), member), member);
		}

		bool InUse(int member) const { return member >= 0 && (size_t)member < memberInUse.size() && memberInUse[(size_t)member] != 0; }
		int NumInUse() const { return (int)memberRow.size() - (int)freeMembers.size(); }

		// row >= 0: active, reads/writes that row of the batch arrays; row < 0: inactive (state frozen)
		void SetActive(int member, int row)
		{
			memberRow[(size_t)member] = row;
			activeDirty = true;
		}

		int NumMembers() const { return (int)memberRow.size(); }
		bool IsContiguous() const { return contiguous; }

		// fresh (never prewarmed) state: zero history / the model's initial h,c
		virtual void Reset(const std::vector<int>& members) = 0;
		virtual void Prewarm(const std::vector<int>& members) = 0;
		// launches on `launchStream` (the batch's main stream, or this group's side stream when several groups run concurrently)
		virtual void Process(const float* dIn, float* dOut, long inStride, long outStride, size_t n, hipStream_t launchStream) = 0;
		virtual double AlgorithmicBytesPerSample(int blockFrames) const = 0;
		virtual double MacsPerSample() const = 0;
		virtual size_t StateBytesPerStream() const = 0;
		// Which launch of a buffer this group's streams ride in (GpuBatch::ProcessDevice): 0 frame kernel, 1 f16-split kernel, 2 f16-split
		// kernel with packed streams, -1 split kernel, joins list 2 when the batch has one (else 1), 3 the fused LDS-free recurrent launch,
		// -2 a launch of its own
		virtual int LaunchClass() const { return -2; }
		virtual int PackFactor() const { return 1; } // real streams per kernel-level stream (WaveNet stream packing)
		// f16-split kernels without a static range proof: (wave, block) pairs in which a value of this member's stream was saturated
		virtual int RangeEvents(int member) { (void)member; return 0; }
		// device buffers that hold nothing but the model's (re-laid-out) weights: identical on every device that runs the model
		virtual void WeightImages(std::vector<std::pair<void*, size_t>>& out) const { (void)out; }
		virtual float InputLimit() const { return INFINITY; } // samples beyond +-limit are clamped by the kernel (f16-split WaveNet kernels)
		virtual const char* KernelName() const = 0;  // the kernel that runs this group's streams (rocprof name, without template arguments)
		// WaveNet groups on the frame kernel can share ONE launch with other such groups (a heterogeneous batch without stream
		// fork/join); fills `out` with this group's part of that launch.  Other groups return false.
		// `launchList`: which fused launch it joins (0 frame kernel, 1 f16-split kernel, 2 f16-split kernel with packed streams)
		virtual bool FusedLaunchArgs(WnFrameGroup& out, int& launchList)
		{
			(void)out;
			(void)launchList;
			return false;
		}

		// LSTM / GRU groups with an LDS-free kernel instance likewise share one launch (recurrent_dpp_kernels.hip)
		virtual bool FusedRecurrentArgs(RecurrentGroup& out)
		{
			(void)out;
			return false;
		}

		bool ListsDirty() const { return activeDirty; }

		int NumActive() const
		{
			int c = 0;
			for (int r : memberRow) c += (r >= 0);
			return c;
		}

		// Upload the active-stream lists if they changed.  Real-time safe: everything it touches was allocated when the members were
		// added (AddMember is the non-real-time side); the copy is asynchronous on the batch stream from one of two pinned staging
		// buffers, so a quality switch costs the audio thread two small enqueues and no synchronisation (the reference switches an
		// atomic index, CompositeModel.h:49-63).  Never called inside a graph capture.
		// returns true when the lists were re-uploaded
		virtual bool SyncActiveLists()
		{
			if (!activeDirty) return false;
			hSlots.clear();
			hRows.clear();
			for (size_t m = 0; m < memberRow.size(); m++)
			{
				if (memberRow[m] >= 0)
				{
					hSlots.push_back((int)m);
					hRows.push_back(memberRow[m]);
				}
			}
			if (!hSlots.empty())
			{
				UploadLists();
			}
			contiguous = !hSlots.empty();
			for (size_t i = 1; i < hSlots.size() && contiguous; i++)
				contiguous = hSlots[i] == hSlots[0] + (int)i && hRows[i] == hRows[0] + (int)i;
			activeDirty = false;
			return true;
		}

	protected:
		virtual void EnsureCapacity(int members) = 0;

		// hSlots / hRows -> the device lists through one of two pinned staging buffers, asynchronously on the batch stream.  Real-time
		// safe: buffers and both events were created on the AddStreams side (EnsureListCapacity).
		void UploadLists()
		{
			listFlip ^= 1;
			int* pin = pinnedLists[listFlip];
			// the copy issued from this buffer two switches ago: long finished unless the host is far ahead of the device
			if (listUsed[listFlip]) CheckHip(hipEventSynchronize(listEvent[listFlip]), "hipEventSynchronize");
			memcpy(pin, hSlots.data(), hSlots.size() * sizeof(int));
			memcpy(pin + listCapacity, hRows.data(), hRows.size() * sizeof(int));
			CheckHip(hipMemcpyAsync(dSlots.Get(), pin, hSlots.size() * sizeof(int), hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
			CheckHip(hipMemcpyAsync(dRows.Get(), pin + listCapacity, hRows.size() * sizeof(int), hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
			CheckHip(hipEventRecord(listEvent[listFlip], stream), "hipEventRecord");
			listUsed[listFlip] = true;
		}

		// index lists (device + two pinned staging buffers + their events) sized for every member: grown here, on the AddStreams side only
		void EnsureListCapacity(size_t members)
		{
			for (int b = 0; b < 2; b++)
				if (!listEvent[b]) CheckHip(hipEventCreateWithFlags(&listEvent[b], hipEventDisableTiming), "hipEventCreate");
			if (members <= listCapacity) return;
			const size_t cap = std::max<size_t>(members, std::max<size_t>(listCapacity * 2, 64));
			CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
			dSlots.Alloc(cap);
			dRows.Alloc(cap);
			for (int b = 0; b < 2; b++)
			{
				if (pinnedLists[b]) (void)hipHostFree(pinnedLists[b]);
				pinnedLists[b] = nullptr;
				CheckHip(hipHostMalloc(reinterpret_cast<void**>(&pinnedLists[b]), 2 * cap * sizeof(int), hipHostMallocDefault), "hipHostMalloc");
			}
			hSlots.reserve(cap);
			hRows.reserve(cap);
			listCapacity = cap;
			listUsed[0] = listUsed[1] = false;
			activeDirty = true;
		}

		hipStream_t stream;
		hipStream_t sideStream = nullptr;
		hipEvent_t doneEvent = nullptr;
		std::vector<int> memberRow; // member == state slot
		std::vector<char> memberInUse; // 0: slot is on the free list
		std::vector<int> freeMembers;  // sorted
		std::vector<int> hSlots, hRows;
		DevArray<int> dSlots, dRows;
		int* pinnedLists[2] = { nullptr, nullptr }; // [slots | rows], listCapacity ints each
		hipEvent_t listEvent[2] = { nullptr, nullptr };
		bool listUsed[2] = { false, false };
		size_t listCapacity = 0;
		int listFlip = 0;
		bool contiguous = false; // active streams are slot0+i / row0+i: kernels may skip the index arrays
		bool activeDirty = true;
	};

	namespace
	{
		// WaveNet kernel families: "split" = the f16-split MFMA kernel (wavenet_split_kernels.hip), "frame" = the f32 4x4x1-MFMA kernel
		// (wavenet_frame_kernels.hip), "generic" = the runtime-shaped kernel for layer arrays wider than 16 channels
		// (wavenet_generic_kernels.hip; frame-kernel state format).  NA_WN_KERNEL=split|frame|generic forces one for every model it can
		// run (tuning / tests); default: chosen per model (FamilyFor).
		enum WnFamily { WN_FAMILY_AUTO, WN_FAMILY_SPLIT, WN_FAMILY_FRAME, WN_FAMILY_GENERIC };
		WnFamily WaveNetFamilyOverride()
		{
			static const WnFamily fam = []() {
				const char* e = getenv("NA_WN_KERNEL");
				const std::string w = e ? e : "auto";
				if (w == "split") return WN_FAMILY_SPLIT;
				if (w == "frame") return WN_FAMILY_FRAME;
				if (w == "generic") return WN_FAMILY_GENERIC;
				return WN_FAMILY_AUTO;
			}();
			return fam;
		}

		// Which kernel family runs a model (fixed for the life of its group: the families keep different stream-state formats).
		// Measured on MI355X, 1024 streams x 128 frames: the f16-split kernel wins where its fast instantiation applies with 2 tiles
		// per wave (every array has 5..8 or 13..16 channels, K = 3: Standard 50 vs 60 us); narrow (Feather, Nano: <= 4-channel
		// arrays) and large-kernel (A2) models are faster on the frame kernel (33 / 30 / 71 us vs 44 / 44 / 133 us); a 12-channel model (Lite)
		// is too as it is (46 vs 50 us), but padded to 16 / 8 channels it runs the fast split flavour (PadFor below: 42.6 us).
		// May the f16-split kernels run this plan at all?  Their values are (hi, lo) pairs of f16: the plan builder proves statically that
		// with inputs inside +-condLimit (>= kSplitMinInputLimit) nothing leaves the f16 range and that the weights fit the operand format
		// (wavenet_plan.cpp).  A model that fails the proof runs on the f32 frame kernel -- no clamp, no overflow, the
		// reference's own number format -- and NA_BatchStreamKernelName says so.  One exception: the official A2 shapes (LeakyReLU: the
		// worst-case bound grows with the product of 23 layers' row sums and fails for every trained model) stay on their chains, which
		// saturate instead of overflowing and count the event (wavenet_split_dev.h SplitQuadSat, NA_BatchStreamRangeEvents).
		bool SplitAllowed(const WaveNetPlan& plan)
		{
			if (plan.genericOnly || !plan.splitWeightsOk || plan.rings.size() > (size_t)WN_RANGE_EVENT_SLOT) return false;
			if (plan.splitRangeProven) return true;
			const int spec = WaveNetSpecArchId(plan.sstages.data(), (int)plan.sstages.size(), plan.stateF4, (int)(plan.wsplit.size() / 8));
			return spec == WN_SPEC_A2FULL || spec == WN_SPEC_A2LITE;
		}

		WnFamily FamilyFor(const WaveNetPlan& plan)
		{
			if (plan.genericOnly) return WN_FAMILY_GENERIC; // > 16 channels: the runtime-shaped kernel is the only one that runs it
			const WnFamily o = WaveNetFamilyOverride();
			if (o == WN_FAMILY_GENERIC && !plan.genericOk) return WN_FAMILY_FRAME; // (conv heads: not in the runtime-shaped kernel)
			if (o == WN_FAMILY_SPLIT && !SplitAllowed(plan)) return WN_FAMILY_FRAME; // (the range proof outranks the tuning knob)
			if (o != WN_FAMILY_AUTO) return o;
			if (!SplitAllowed(plan)) return WN_FAMILY_FRAME;
			if (plan.splitFastT == 2) return WN_FAMILY_SPLIT;
			// the A2 submodels have compile-time specialised chains on the split kernels' state format (wavenet_spec_kernels.hip; round 3:
			// 2048-stream quality sweep 120 us on the frame kernel); blocks that are not 128 / 64 frames fall to the stage interpreter
			const int spec = WaveNetSpecArchId(plan.sstages.data(), (int)plan.sstages.size(), plan.stateF4, (int)(plan.wsplit.size() / 8));
			return (spec == WN_SPEC_A2FULL || spec == WN_SPEC_A2LITE) ? WN_FAMILY_SPLIT : WN_FAMILY_FRAME;
		}

		// Frames of the next launch of a buffer with `left` frames to go.  Models with compact rings (wavenet_dev.h) take 128, 64 or at
		// most 32 frames per launch -- the lengths for which a block never reads a ring position it writes; everything else 128 at a time
		// (the reference chunks at 64, InternalModel.h:104-117; results do not depend on the chunking).
		int NextWaveNetChunk(size_t left, bool compactRings)
		{
			if (left >= (size_t)WN_MAX_FRAMES) return WN_MAX_FRAMES;
			if (!compactRings) return (int)left;
			return left >= 64 ? 64 : (left >= 32 ? 32 : (int)left);
		}

		class WaveNetGroup : public ModelGroup
		{
		public:
			// Stream packing (wavenet_plan.cpp PackWaveNetDesc): several streams of a NARROW model share one virtual stream of the f16-split
			// kernel -- 4 streams for <= 4-channel arrays (Nano), 2 for <= 8 (Feather).  With the compile-time specialised chains it wins at
			// every batch size (measured, 128-frame blocks, us per step packed / f32 frame kernel: Nano 64 streams 19.5 / 24.5, 1024: 26.8 /
			// 30.6, 4096: 61 / 113; Feather 64: 15.4 / 26.5, 1024: 24.3 / 33.6), so every static narrow model that is not a submodel of a
			// slimmable container (`packHint` > 0: its members are always active) runs packed, whatever the AddStreams call pattern -- the
			// state layout of a group never depends on how its streams arrived.  NA_WN_PACK=0 turns packing off.
			static int PackFor(const WaveNetDesc& wn, int packHint)
			{
				static const int mode = getenv("NA_WN_PACK") ? atoi(getenv("NA_WN_PACK")) : -1;
				const WnFamily o = WaveNetFamilyOverride();
				if (packHint <= 0 || mode == 0 || (o != WN_FAMILY_AUTO && o != WN_FAMILY_SPLIT)) return 1;
				ValidateWaveNetDesc(wn);
				for (const WnArrayCfg& cfg : wn.arrays)
					if (cfg.channels > 16) return 1;
				const int P = WaveNetPackFactor(wn);
				if (P < 2) return 1;
				// packing means the f16-split kernels: only for a model that passes their range proof (block-diagonal packing keeps every
				// row sum, so the real model's proof is the virtual model's)
				return SplitAllowed(BuildWaveNetPlan(wn, true)) ? P : 1;
			}

			// Padding without packing (wavenet_plan.cpp WaveNetWantsPadding): a model whose arrays do not fill their lane mode (A1 Lite:
			// 12 / 6 channels) is widened to 16 / 8 and runs the fast flavour of the split kernel (1024 streams: 43.8 vs 46.0 us on the
			// frame kernel, 1365: 66.4 vs 76.0).  NA_WN_PAD=0 turns it off.
			static bool PadFor(const WaveNetDesc& wn)
			{
				static const bool off = getenv("NA_WN_PAD") != nullptr && atoi(getenv("NA_WN_PAD")) == 0;
				const WnFamily o = WaveNetFamilyOverride();
				if (off || (o != WN_FAMILY_AUTO && o != WN_FAMILY_SPLIT)) return false;
				ValidateWaveNetDesc(wn);
				return WaveNetWantsPadding(wn) && SplitAllowed(BuildWaveNetPlan(wn, true));
			}

			// The two state formats size their rings differently (wavenet_plan.cpp AddRing): the plan is built for the f16-split kernels
			// first, and once more for the others when the family choice (which looks at the stage program, not at the rings) says so
			static WaveNetPlan PlanForItsFamily(const WaveNetDesc& wn)
			{
				WaveNetPlan p = BuildWaveNetPlan(wn, true);
				if (FamilyFor(p) == WN_FAMILY_SPLIT) return p;
				return BuildWaveNetPlan(wn, false);
			}

			// packHint: 0 = never pack (submodel of a container), otherwise the number of streams the creating AddStreams call brings
			WaveNetGroup(const std::shared_ptr<const ModelDesc>& d, hipStream_t s, int packHint = 0)
				: ModelGroup(d, s), pack(PackFor(d->wavenet, packHint)),
				  plan((pack > 1 || PadFor(d->wavenet)) ? BuildPackedWaveNetPlan(d->wavenet, pack) : PlanForItsFamily(d->wavenet)),
				  family(plan.isVirtual() ? WN_FAMILY_SPLIT : FamilyFor(plan))
			{
				if (plan.isVirtual())
				{
					if (plan.splitFastT != 2) throw std::runtime_error("internal: packed / padded WaveNet plan is not a fast split-kernel plan");
					realPlan = BuildWaveNetPlan(d->wavenet); // bookkeeping (bytes / MACs per REAL stream)
				}
				dStages.Upload(plan.stages, stream);
				dWpack.Upload(plan.wpack, stream);
				dWpk.Upload(plan.wpk, stream);
				dPrewarm.Upload(plan.prewarm, stream);
				dWeights.Upload(plan.isVirtual() ? plan.packedWeights : d->wavenet.weights, stream);
				if (family == WN_FAMILY_GENERIC)
				{
					// the runtime-shaped kernel reads a layer conv tap by tap as a [cout x cin] matrix: its copy of the weights keeps every
					// layer conv tap-major ([k][out][in] instead of the reference's [out][in][k], WaveNet.h:99-111), so that a lane's four
					// input channels are one 16-byte load and a row is contiguous
					std::vector<float> wg = d->wavenet.weights;
					for (const WnPrewarmLayer& pw : plan.prewarm)
					{
						if (pw.kind != 0 || pw.ksize <= 1) continue;
						const size_t base = (size_t)pw.wconv, K = (size_t)pw.ksize, CO = (size_t)pw.cout, CI = (size_t)pw.cin;
						for (size_t o = 0; o < CO; o++)
							for (size_t c = 0; c < CI; c++)
								for (size_t k = 0; k < K; k++) wg[base + (k * CO + o) * CI + c] = d->wavenet.weights[base + (o * CI + c) * K + k];
					}
					dWeightsGen.Upload(wg, stream);
				}
				dSStages.Upload(plan.sstages, stream);
				dWsplit.Upload(plan.wsplit, stream);

				std::vector<int> ringOff, ringFrames, ringG;
				for (const auto& r : plan.rings)
				{
					ringOff.push_back(r.offF4);
					ringFrames.push_back(r.frames);
					ringG.push_back(r.G);
				}
				dRingOff.Upload(ringOff, stream);
				dRingFrames.Upload(ringFrames, stream);
				dRingG.Upload(ringG, stream);

				// steady-state columns: once per model (WaveNet.h:746-766)
				dCols.Alloc(plan.rings.size() * WN_COL_STRIDE);
				CheckHip(LaunchWaveNetPrewarmColumns(dPrewarm.Get(), (int)plan.prewarm.size(), dWeights.Get(), dCols.Get(), stream),
					"WaveNetPrewarmColumnsKernel");

				dev.stages = dStages.Get();
				dev.wpack = dWpack.Get();
				dev.wpk = dWpk.Get();
				dev.ring_frames = dRingFrames.Get();
				dev.nstages = (int)plan.stages.size();
				dev.wpack_f4 = (int)(plan.wpack.size() / 4);
				dev.max_a4_floats = plan.maxA4Floats;
				dev.max_ksize = 1;
				for (const WnStage& st : plan.stages)
					if (st.type == WN_ST_LAYER) dev.max_ksize = std::max(dev.max_ksize, st.ksize);
				dev.wpk_floats = (int)plan.wpk.size();
				dev.nrings = (int)plan.rings.size();
				dev.state_f4 = plan.stateF4;
				dev.head_scale = plan.headScale;
				dev.sstages = dSStages.Get();
				dev.wsplit = dWsplit.Get();
				dev.wsplit_quads = (int)(plan.wsplit.size() / 8);
				dev.max_split_ops = plan.maxSplitOps;
				dev.max_G = plan.maxG;
				dev.split_fast_T = plan.splitFastT;
				dev.cond_limit = plan.condLimit;
				dev.saturate = plan.splitRangeProven ? 0 : 1;
				dev.compact_rings = (family == WN_FAMILY_SPLIT && plan.compactRings) ? 1 : 0;
				dev.spec_arch = family == WN_FAMILY_SPLIT ? WaveNetSpecArchId(plan.sstages.data(), (int)plan.sstages.size(), plan.stateF4, (int)(plan.wsplit.size() / 8)) : WN_SPEC_NONE;
			}

			// ChannelHistoryBuffer::AllocBuffer zero-fills (WaveNet.h:38-40)
			void Reset(const std::vector<int>& members) override
			{
				if (pack > 1)
				{
					// a member in position 0 opens a fresh virtual stream (cursors and every ring zero); the others only clear their own
					// channel groups of a virtual stream that is already running
					std::vector<int> slots, subs;
					// (`members` is ascending: freed slots are handed out lowest first, new ones follow)
					auto isNew = [&](int o) { return std::binary_search(members.begin(), members.end(), o); };
					for (int m : members)
					{
						// a virtual stream none of whose other members is running starts fresh: cursors and every ring zero (once, by its
						// first new member); a member joining -- or recycling a position of -- a running virtual stream only clears its own
						// channel groups and leaves cursors and neighbours alone
						const int v0 = (m / pack) * pack;
						bool fresh = true;
						int firstNew = m;
						for (int q = 0; q < pack; q++)
						{
							const int o = v0 + q;
							if (o == m) continue;
							if (isNew(o)) firstNew = std::min(firstNew, o);
							else if (InUse(o)) fresh = false;
						}
						if (fresh)
						{
							if (m == firstNew)
								CheckHip(hipMemsetAsync(state.Get() + (size_t)(m / pack) * (size_t)plan.stateF4 * 4, 0, (size_t)plan.stateF4 * 16, stream), "hipMemsetAsync");
						}
						else
						{
							slots.push_back(m / pack);
							subs.push_back(m % pack);
						}
					}
					FillPacked(slots, subs, true);
					return;
				}
				// one memset per run of consecutive slots (a batch add is a single run)
				for (size_t i = 0; i < members.size();)
				{
					size_t k = i + 1;
					while (k < members.size() && members[k] == members[k - 1] + 1) k++;
					CheckHip(hipMemsetAsync(state.Get() + (size_t)members[i] * (size_t)plan.stateF4 * 4, 0, (k - i) * (size_t)plan.stateF4 * 16, stream),
						"hipMemsetAsync");
					i = k;
				}
			}

			void Prewarm(const std::vector<int>& members) override
			{
				if (members.empty()) return;
				if (pack > 1)
				{
					std::vector<int> slots, subs;
					for (int m : members)
					{
						slots.push_back(m / pack);
						subs.push_back(m % pack);
					}
					FillPacked(slots, subs, false);
					return;
				}
				DevArray<int> list;
				list.Upload(members, stream);
				CheckHip(LaunchWaveNetFillRings(state.Get(), plan.stateF4, list.Get(), (int)members.size(), (int)plan.rings.size(),
					dRingOff.Get(), dRingFrames.Get(), dRingG.Get(), dCols.Get(), stream, family == WN_FAMILY_SPLIT), "WaveNetFillRingsKernel");
				CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize"); // `list` is freed on return
			}

			void Process(const float* dIn, float* dOut, long inStride, long outStride, size_t n, hipStream_t launchStream) override
			{
				SyncActiveLists();
				const int numActive = (int)hSlots.size();
				if (numActive == 0) return;
				size_t offset = 0;
				while (n > 0)
				{
					const int chunk = NextWaveNetChunk(n, dev.compact_rings != 0);
					const WnFamily which = family;
					if (which == WN_FAMILY_SPLIT)
					{
						const WnFrameGroup g = { &dev, state.Get(), contiguous ? nullptr : dSlots.Get(), dRows.Get(), numActive, contiguous ? hSlots[0] : 0, contiguous ? hRows[0] : 0, pack };
						CheckHip(LaunchWaveNetSplitFused(&g, 1, dIn + offset, dOut + offset, inStride, outStride, chunk, launchStream), "WaveNetSplitKernel");
					}
					else if (which == WN_FAMILY_GENERIC)
						CheckHip(LaunchWaveNetGeneric(dPrewarm.Get(), (int)plan.prewarm.size(), dWeightsGen.Get(), dRingOff.Get(), dRingFrames.Get(), dRingG.Get(),
							(int)plan.rings.size(), plan.stateF4, plan.maxChannels, plan.headScale, state.Get(), contiguous ? nullptr : dSlots.Get(), dRows.Get(), numActive,
							contiguous ? hSlots[0] : 0, contiguous ? hRows[0] : 0, dIn + offset, dOut + offset, inStride, outStride, chunk, launchStream), "WaveNetGenericKernel");
					else
						CheckHip(LaunchWaveNetFrame(dev, state.Get(), contiguous ? nullptr : dSlots.Get(), dRows.Get(), numActive, dIn + offset, dOut + offset,
							inStride, outStride, chunk, launchStream, contiguous ? hSlots[0] : 0, contiguous ? hRows[0] : 0), "WaveNetFrameKernel");
					offset += (size_t)chunk;
					n -= (size_t)chunk;
				}
			}

			bool FusedLaunchArgs(WnFrameGroup& out, int& launchList) override
			{
				if (family == WN_FAMILY_GENERIC) return false; // its own launch
				// list 2 = the packed flavour of the split kernel; a plain group whose plan runs the fast flavour may join it (negative list:
				// "1, or 2 if a packed group is in the batch" -- then it passes its index lists even when its streams are contiguous)
				launchList = LaunchClass();
				out.pack = pack;
				SyncActiveLists();
				out.model = &dev;
				out.state = state.Get();
				out.slots = contiguous ? nullptr : dSlots.Get();
				out.rows = dRows.Get();
				out.numStreams = (int)hSlots.size();
				out.slot0 = contiguous ? hSlots[0] : 0;
				out.row0 = contiguous ? hRows[0] : 0;
				listSlots = dSlots.Get();
				return out.numStreams > 0;
			}

			double AlgorithmicBytesPerSample(int blockFrames) const override { return (plan.isVirtual() ? realPlan : plan).AlgorithmicBytesPerSample(blockFrames); }
			double MacsPerSample() const override { return (plan.isVirtual() ? realPlan : plan).MacsPerSample(); }
			size_t StateBytesPerStream() const override { return (size_t)plan.stateF4 * 16 / (size_t)pack; }
			int LaunchClass() const override
			{
				if (family == WN_FAMILY_GENERIC) return -2;
				return family == WN_FAMILY_SPLIT ? (pack > 1 ? 2 : (plan.splitFastT == 2 ? -1 : 1)) : 0;
			}
			int PackFactor() const override { return pack; }
			float InputLimit() const override { return family == WN_FAMILY_SPLIT ? plan.condLimit : INFINITY; }
			void WeightImages(std::vector<std::pair<void*, size_t>>& out) const override
			{
				auto add = [&](void* p, size_t bytes) { if (p && bytes) out.push_back({ p, bytes }); };
				add(dWpack.Get(), dWpack.Count() * sizeof(float));
				add(dWpk.Get(), dWpk.Count() * sizeof(float));
				add(dWeights.Get(), dWeights.Count() * sizeof(float));
				add(dWeightsGen.Get(), dWeightsGen.Count() * sizeof(float));
				add(dWsplit.Get(), dWsplit.Count() * sizeof(uint16_t));
			}
			int RangeEvents(int member) override
			{
				if (family != WN_FAMILY_SPLIT || !dev.saturate || !InUse(member)) return 0;
				int count = 0;
				const float* slot = state.Get() + (size_t)(member / pack) * (size_t)plan.stateF4 * 4;
				CheckHip(hipMemcpyAsync(&count, reinterpret_cast<const int*>(slot) + WN_RANGE_EVENT_SLOT, sizeof(int), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H");
				CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
				return count;
			}
			const char* KernelName() const override
			{
				// (a model with a specialised chain runs it for blocks of 128 / 64 / 32 frames, the interpreter for other lengths)
				if (family == WN_FAMILY_SPLIT) return (dev.spec_arch != WN_SPEC_NONE && WaveNetSpecEnabled()) ? "WaveNetSpecKernel" : "WaveNetSplitKernel";
				return family == WN_FAMILY_GENERIC ? (plan.maxChannels > 64 ? "WaveNetWideKernel" : "WaveNetGenericKernel") : "WaveNetFrameKernel";
			}

			// Packed groups: the launch lists name VIRTUAL streams -- slot = member / pack -- and hold `pack` rows each (-1: no member in
			// that position yet).  Members of a static model are always active, so a virtual stream runs as soon as it has one member.
			bool SyncActiveLists() override
			{
				if (pack <= 1) return ModelGroup::SyncActiveLists();
				if (!activeDirty) return false;
				hSlots.clear();
				hRows.clear();
				const size_t numSlots = (memberRow.size() + (size_t)pack - 1) / (size_t)pack;
				for (size_t v = 0; v < numSlots; v++)
				{
					bool any = false;
					int rows[4] = { -1, -1, -1, -1 };
					for (int q = 0; q < pack; q++)
					{
						const size_t m = v * (size_t)pack + (size_t)q;
						if (m < memberRow.size() && memberRow[m] >= 0)
						{
							rows[q] = memberRow[m];
							any = true;
						}
					}
					if (!any) continue;
					hSlots.push_back((int)v);
					for (int q = 0; q < pack; q++) hRows.push_back(rows[q]);
				}
				if (!hSlots.empty()) UploadLists();
				contiguous = false; // the packed kernel always reads the lists
				activeDirty = false;
				return true;
			}

		protected:
			void EnsureCapacity(int numMembers) override
			{
				EnsureListCapacity((size_t)numMembers + (size_t)pack); // the row list holds `pack` entries per virtual stream
				const int members = (numMembers + pack - 1) / pack;      // state slots = virtual streams
				if ((size_t)members <= capacity) return;
				const size_t newCap = std::max<size_t>((size_t)members, std::max<size_t>(capacity * 2, 16));
				DevArray<float> bigger;
				bigger.Alloc(newCap * (size_t)plan.stateF4 * 4);
				if (capacity > 0)
				{
					CheckHip(hipMemcpyAsync(bigger.Get(), state.Get(), capacity * (size_t)plan.stateF4 * 16, hipMemcpyDeviceToDevice, stream),
						"hipMemcpyAsync D2D");
					CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
				}
				state.Swap(bigger);
				capacity = newCap;
			}

		private:
			void FillPacked(const std::vector<int>& slots, const std::vector<int>& subs, bool zero)
			{
				if (slots.empty()) return;
				DevArray<int> dS, dQ;
				dS.Upload(slots, stream);
				dQ.Upload(subs, stream);
				CheckHip(LaunchWaveNetFillRings(state.Get(), plan.stateF4, dS.Get(), (int)slots.size(), (int)plan.rings.size(), dRingOff.Get(), dRingFrames.Get(),
					dRingG.Get(), dCols.Get(), stream, true, dQ.Get(), pack, zero), "WaveNetFillRingsKernel");
				CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize"); // the lists are freed on return
			}

		public:
			const int* listSlots = nullptr; // device slot list of the last FusedLaunchArgs (always uploaded, also for contiguous groups)
		private:
			const int pack;       // real streams per virtual stream (1: no packing)
			WaveNetPlan plan;     // pack > 1: of the VIRTUAL model
			WaveNetPlan realPlan; // pack > 1: of the real model (bookkeeping only)
			const WnFamily family;
			WnModelDev dev = {};
			DevArray<WnStage> dStages;
			DevArray<float> dWpack;
			DevArray<float> dWpk;
			DevArray<WnPrewarmLayer> dPrewarm;
			DevArray<float> dWeights;
			DevArray<float> dWeightsGen; // WN_FAMILY_GENERIC: layer convs tap-major
			DevArray<int> dRingOff, dRingFrames, dRingG;
			DevArray<float> dCols;
			DevArray<WnSplitStage> dSStages;
			DevArray<uint16_t> dWsplit;
			DevArray<float> state;
			size_t capacity = 0;
		};

		class LstmGroup : public ModelGroup
		{
		public:
			LstmGroup(const std::shared_ptr<const ModelDesc>& d, hipStream_t s) : ModelGroup(d, s)
			{
				const LSTMDesc& lstm = d->lstm;
				ValidateRecurrentDesc(lstm); // the loader already did; descs built by hand get the same message
				std::vector<float> w;
				for (int l = 0; l < lstm.numLayers; l++)
				{
					dev.layerOff[l] = (int)w.size();
					w.insert(w.end(), lstm.layers[(size_t)l].w.begin(), lstm.layers[(size_t)l].w.end());
					w.insert(w.end(), lstm.layers[(size_t)l].bias.begin(), lstm.layers[(size_t)l].bias.end());
					init.insert(init.end(), lstm.layers[(size_t)l].h0.begin(), lstm.layers[(size_t)l].h0.end());
					init.insert(init.end(), lstm.layers[(size_t)l].c0.begin(), lstm.layers[(size_t)l].c0.end());
				}
				dev.headOff = (int)w.size();
				w.insert(w.end(), lstm.headWeights.begin(), lstm.headWeights.begin() + lstm.hiddenSize);
				w.push_back(lstm.headBias);
				dev.tailLayers = (int)lstm.tail.size(); // generic keras stack: a chain of dense layers instead of the head
				dev.tailWidth = 0;
				for (size_t t = 0; t < lstm.tail.size(); t++)
				{
					const DenseLayerDesc& dl = lstm.tail[t];
					dev.tailOff[t] = (int)w.size();
					dev.tailIn[t] = dl.in;
					dev.tailOut[t] = dl.out;
					dev.tailAct[t] = dl.activation;
					dev.tailWidth = std::max(dev.tailWidth, dl.out);
					w.insert(w.end(), dl.w.begin(), dl.w.end());
					w.insert(w.end(), dl.b.begin(), dl.b.end());
					tailMacs += (double)dl.in * dl.out;
				}
				dW.Upload(w, stream);
				dInit.Upload(init, stream);
				dev.w = dW.Get();
				{
					// the gate matrices once more, transposed into [quad of inputs][row][4] (lstm_dev.h: LstmModelDev::wT)
					const int H = lstm.hiddenSize, gateRows = ((lstm.cell == CELL_GRU) ? 3 : 4) * H;
					dev.waves = RecurrentWaveWaves(gateRows);
					dev.rowsPad = (gateRows + 64 * dev.waves - 1) / (64 * dev.waves) * (64 * dev.waves);
					std::vector<float> wt;
					for (int l = 0; l < lstm.numLayers; l++)
					{
						const int I = (l == 0) ? 1 : H, W = I + H, Qi = (I + 3) / 4, Qh = (H + 3) / 4;
						dev.layerOffT[l] = (int)wt.size();
						wt.resize(wt.size() + (size_t)(Qi + Qh) * dev.rowsPad * 4, 0.0f);
						float* dst = wt.data() + dev.layerOffT[l];
						const std::vector<float>& src = lstm.layers[(size_t)l].w; // row-major [gateRows][W]
						for (int r = 0; r < gateRows; r++)
						{
							for (int k = 0; k < I; k++) dst[((size_t)(k / 4) * dev.rowsPad + r) * 4 + (k % 4)] = src[(size_t)r * W + k];
							for (int k = 0; k < H; k++) dst[((size_t)(Qi + k / 4) * dev.rowsPad + r) * 4 + (k % 4)] = src[(size_t)r * W + I + k];
						}
					}
					dWT.Upload(wt, stream);
					dev.wT = dWT.Get();
				}
				dev.cell = (lstm.cell == CELL_GRU) ? LSTM_CELL_GRU : LSTM_CELL_LSTM;
				dev.numLayers = lstm.numLayers;
				dev.hidden = lstm.hiddenSize;
				dev.math = (lstm.mathMode == MATH_STD) ? LSTM_MATH_STD : LSTM_MATH_FAST;
				numElems = lstm.numLayers * 2 * lstm.hiddenSize;
				dZeros.Alloc(LSTM_MAX_FRAMES);
				CheckHip(hipMemsetAsync(dZeros.Get(), 0, LSTM_MAX_FRAMES * sizeof(float), stream), "hipMemsetAsync");
			}

			// InternalLSTMModelT::Prewarm -> NeuralModelImpl::Prewarm(2048, 64) (InternalModel.h:368-371):
			// run 2048 zeros through the recurrence from the CURRENT state (the initial h/c right after load;
			// a later Prewarm() call continues from wherever the stream is, exactly like the reference).
			void Reset(const std::vector<int>& members) override
			{
				if (members.empty()) return;
				DevArray<int> list;
				list.Upload(members, stream);
				CheckHip(LaunchLstmInitState(state.Get(), (int)capacity, list.Get(), (int)members.size(), dInit.Get(), numElems, stream),
					"LstmInitStateKernel");
				CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
			}

			void Prewarm(const std::vector<int>& members) override
			{
				if (members.empty()) return;
				DevArray<int> list, rows;
				list.Upload(members, stream);
				std::vector<int> zeroRows(members.size(), 0);
				rows.Upload(zeroRows, stream);
				DevArray<float> sink;
				sink.Alloc(LSTM_MAX_FRAMES);
				for (int done = 0; done < 2048; done += LSTM_MAX_FRAMES)
					CheckHip(Launch(list.Get(), rows.Get(), (int)members.size(), dZeros.Get(), sink.Get(), 0, 0, LSTM_MAX_FRAMES, stream), "recurrent kernel (prewarm)");
				CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
			}

			void Process(const float* dIn, float* dOut, long inStride, long outStride, size_t n, hipStream_t launchStream) override
			{
				SyncActiveLists();
				const int numActive = (int)hSlots.size();
				if (numActive == 0) return;
				size_t offset = 0;
				RecurrentGroup fused;
				const bool dpp = FusedRecurrentArgs(fused); // the LDS-free kernel, as a launch of one group (with the contiguous-streams shortcut)
				while (n > 0)
				{
					const int chunk = (int)std::min<size_t>(n, (size_t)LSTM_MAX_FRAMES);
					if (dpp) CheckHip(LaunchRecurrentDpp(&fused, 1, dIn + offset, dOut + offset, inStride, outStride, chunk, launchStream), "RecurrentDppKernel");
					else CheckHip(Launch(dSlots.Get(), dRows.Get(), numActive, dIn + offset, dOut + offset, inStride, outStride, chunk, launchStream), "recurrent kernel");
					offset += (size_t)chunk;
					n -= (size_t)chunk;
				}
			}

			bool FusedRecurrentArgs(RecurrentGroup& out) override
			{
				static const bool noDpp = getenv("NA_LSTM_NO_DPP") != nullptr || getenv("NA_GRU_NO_DPP") != nullptr || getenv("NA_LSTM_LANE_KERNEL") != nullptr;
				if (noDpp || !RecurrentDppSupported(dev)) return false;
				SyncActiveLists();
				out.model = dev;
				out.state = state.Get();
				out.capacity = (int)capacity;
				out.slots = contiguous ? nullptr : dSlots.Get();
				out.rows = dRows.Get();
				out.numStreams = (int)hSlots.size();
				out.slot0 = contiguous ? hSlots[0] : 0;
				out.row0 = contiguous ? hRows[0] : 0;
				return out.numStreams > 0;
			}

			hipError_t Launch(const int* slots, const int* rows, int count, const float* dIn, float* dOut, long inStride, long outStride, int n, hipStream_t s)
			{
				if (dev.cell == LSTM_CELL_GRU) return LaunchGruBlock(dev, state.Get(), (int)capacity, slots, rows, count, dIn, dOut, inStride, outStride, n, s);
				return LaunchLstmBlock(dev, state.Get(), (int)capacity, slots, rows, count, dIn, dOut, inStride, outStride, n, s);
			}

			// 8 + 2*4*(state floats)/N bytes per sample (a GRU has no cell state: half of it)
			double AlgorithmicBytesPerSample(int blockFrames) const override
			{
				return 8.0 + 8.0 * (dev.cell == LSTM_CELL_GRU ? numElems / 2 : numElems) / blockFrames;
			}

			double MacsPerSample() const override
			{
				const LSTMDesc& lstm = desc->lstm;
				double macs = 0.0;
				const double gates = (lstm.cell == CELL_GRU) ? 3.0 : 4.0;
				for (int l = 0; l < lstm.numLayers; l++) macs += gates * lstm.hiddenSize * ((l == 0 ? 1 : lstm.hiddenSize) + lstm.hiddenSize);
				return macs + (lstm.tail.empty() ? lstm.hiddenSize : tailMacs);
			}

			void WeightImages(std::vector<std::pair<void*, size_t>>& out) const override
			{
				if (dW.Get()) out.push_back({ dW.Get(), dW.Count() * sizeof(float) });
				if (dWT.Get()) out.push_back({ dWT.Get(), dWT.Count() * sizeof(float) });
			}
			size_t StateBytesPerStream() const override { return (size_t)numElems * sizeof(float); }
			int LaunchClass() const override
			{
				static const bool noDpp = getenv("NA_LSTM_NO_DPP") != nullptr || getenv("NA_GRU_NO_DPP") != nullptr || getenv("NA_LSTM_LANE_KERNEL") != nullptr;
				return (!noDpp && RecurrentDppSupported(dev)) ? 3 : -2;
			}
			const char* KernelName() const override
			{
				// (four streams per wave from RecurrentQuadMinStreams() streams in ONE launch: a batch of several recurrent models decides on
				// their total, this name on the group's own count)
				if (RecurrentDppSupported(dev))
					return (RecurrentQuadSupported(dev) && RecurrentQuadMinStreams() > 0 &&
						NumActive() >= (dev.cell == LSTM_CELL_GRU ? RecurrentQuadMinStreams() * 2 / 3 : RecurrentQuadMinStreams())) ? "RecurrentQuadKernel" : "RecurrentDppKernel";
				return dev.cell == LSTM_CELL_GRU ? "GruWaveKernel / RecurrentWaveRtKernel / GruGenericKernel" : "LstmWaveKernel / RecurrentWaveRtKernel / LstmBlockKernel / LstmGenericKernel";
			}

		protected:
			void EnsureCapacity(int members) override
			{
				if ((size_t)members <= capacity) return;
				const size_t newCap = std::max<size_t>((size_t)members, std::max<size_t>(capacity * 2, 64));
				DevArray<float> bigger;
				bigger.Alloc(newCap * (size_t)numElems);
				if (capacity > 0)
				{
					// [elem][capacity] -> [elem][newCap]
					CheckHip(hipMemcpy2DAsync(bigger.Get(), newCap * sizeof(float), state.Get(), capacity * sizeof(float), capacity * sizeof(float),
						(size_t)numElems, hipMemcpyDeviceToDevice, stream), "hipMemcpy2DAsync");
					CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
				}
				state.Swap(bigger);
				capacity = newCap;
			}

		private:
			LstmModelDev dev = {};
			DevArray<float> dW, dWT, dInit, dZeros;
			DevArray<float> state;
			std::vector<float> init;
			int numElems = 0;
			double tailMacs = 0.0;
			size_t capacity = 0;
		};
	}
}

namespace na
{
	// Host side only (no device): which kernel family a batch would pick for `streams` streams of this submodel -- the constructor logic
	// of WaveNetGroup (pack / pad / range proof / FamilyFor) -- and the facts of the range proof behind the choice.
	ModelKernelInfo PredictModelKernel(const LoadedModel& model, float quality, int streams)
	{
		ModelKernelInfo info;
		if (model.subModels.empty()) return info;
		const int idx = model.isComposite ? model.ModelIndexFromQuality(quality) : 0;
		const ModelDesc& d = *model.subModels[(size_t)idx].desc;
		if (d.kind != MODEL_WAVENET)
		{
			info.kernel = "recurrent";
			return info;
		}
		const WaveNetPlan real = BuildWaveNetPlan(d.wavenet, true);
		info.inputLimit = real.condLimit;
		info.rangeProven = real.splitRangeProven;
		info.weightsOk = real.splitWeightsOk;
		const int packHint = (!model.isComposite && model.subModels.size() == 1) ? streams : 0;
		const int pack = WaveNetGroup::PackFor(d.wavenet, packHint);
		const bool isVirtual = pack > 1 || WaveNetGroup::PadFor(d.wavenet);
		const WnFamily fam = isVirtual ? WN_FAMILY_SPLIT : FamilyFor(real);
		info.pack = pack;
		info.kernel = fam == WN_FAMILY_SPLIT ? "f16-split" : (fam == WN_FAMILY_GENERIC ? "generic" : "frame");
		if (fam != WN_FAMILY_SPLIT) info.inputLimit = INFINITY;
		return info;
	}

	// Cost of one stream for the multi-GPU sharder: time = per-launch skeleton + bytes (WaveNet) / multiply-accumulates (recurrent),
	// two-point fits per kernel family to the round-3 measurements (us per 1024 streams x 128 frames): specialised / split chains
	// 22.3 + 0.0143 B (Standard 41.6, Lite 33.7), f32 frame kernel 31.6 + 0.0164 B (A2-Lite 46.6, A2-Full 71.4), LDS-free recurrent
	// kernel 8.4 + 0.0107 MAC (LSTM 1x16 20.2, 2x16 42.1), runtime-shaped kernels by their measured per-block times.
	double EstimateStreamCost(const LoadedModel& model, float quality)
	{
		if (model.subModels.empty()) return 1.0;
		const int idx = model.isComposite ? model.ModelIndexFromQuality(quality) : 0;
		const ModelDesc& d = *model.subModels[(size_t)idx].desc;
		if (d.kind == MODEL_WAVENET)
		{
			const WaveNetPlan plan = BuildWaveNetPlan(d.wavenet, true);
			const double B = plan.AlgorithmicBytesPerSample(WN_MAX_FRAMES);
			if (plan.genericOnly) return 1000.0 * plan.maxChannels / 32.0; // 1.0 ms per block at 32 channels (<= 256 streams)
			const bool composite = model.isComposite;
			const bool split = plan.splitFastT == 2 || (!composite && (WaveNetPackFactor(d.wavenet) > 1 || WaveNetWantsPadding(d.wavenet)));
			return split ? 22.3 + 0.0143 * B : 31.6 + 0.0164 * B;
		}
		const LSTMDesc& l = d.lstm;
		const double gates = (l.cell == CELL_GRU) ? 3.0 : 4.0;
		double macs = 0.0;
		for (int k = 0; k < l.numLayers; k++) macs += gates * l.hiddenSize * ((k == 0 ? 1 : l.hiddenSize) + l.hiddenSize);
		macs += l.hiddenSize;
		const bool dpp = l.tail.empty() && l.numLayers <= 2 && (l.hiddenSize <= 16 || (l.numLayers == 1 && l.hiddenSize <= 32));
		return dpp ? 8.4 + 0.0107 * macs : 60.0 + 0.06 * macs;
	}

	// ------------------------------------------------------------------------------------------ GpuBatch

	GpuBatch::GpuBatch(int dev, hipStream_t borrowedStream) : device(dev)
	{
		const int count = VisibleDeviceCount();
		if (count <= 0) throw std::runtime_error("neuralaudio_amd: no HIP device is visible; this library has no CPU fallback");
		if (dev < 0 || dev >= count) throw std::runtime_error("neuralaudio_amd: invalid HIP device index");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		if (borrowedStream)
		{
			stream = borrowedStream;
			ownsStream = false;
			streamObserved = true; // the caller orders its own work on it
		}
		else
		{
			CheckHip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
		}
		if (const char* e = getenv("NA_HOST_CHAINS")) numChains = std::min(std::max(atoi(e), 2), kMaxChains); // tuning knob
	}

	GpuBatch::~GpuBatch()
	{
		(void)hipSetDevice(device);
		for (PipeSlot& p : pipe)
			if (p.own) (void)hipStreamSynchronize(p.own);
		for (hipStream_t hs : halfStream)
			if (hs) (void)hipStreamSynchronize(hs);
		if (stream) (void)hipStreamSynchronize(stream);
		groups.clear();
		if (hostStage) (void)hipHostFree(hostStage);
		if (devStage) (void)hipFree(devStage);
		for (PipeSlot& p : pipe)
		{
			if (p.hostIn) (void)hipHostFree(p.hostIn);
			if (p.hostOut) (void)hipHostFree(p.hostOut);
			if (p.dev) (void)hipFree(p.dev);
			if (p.uploaded) (void)hipEventDestroy(p.uploaded);
			if (p.computed) (void)hipEventDestroy(p.computed);
			if (p.downloaded) (void)hipEventDestroy(p.downloaded);
			for (hipEvent_t e : p.halfDone)
				if (e) (void)hipEventDestroy(e);
			if (p.own) (void)hipStreamDestroy(p.own);
		}
		if (mainDone) (void)hipEventDestroy(mainDone);
		for (hipStream_t hs : halfStream)
			if (hs) (void)hipStreamDestroy(hs);
		for (auto& m : marks)
			for (hipEvent_t e : m)
				if (e) (void)hipEventDestroy(e);

		if (copyIn) (void)hipStreamDestroy(copyIn);
		if (copyOut) (void)hipStreamDestroy(copyOut);
		for (auto& e : graphCache) (void)hipGraphExecDestroy(e.exec);
		if (forkEvent) (void)hipEventDestroy(forkEvent);
		if (stream && ownsStream) (void)hipStreamDestroy(stream);
	}

	ModelGroup* GpuBatch::GroupFor(const std::shared_ptr<const ModelDesc>& desc, int packHint)
	{
		for (auto& g : groups)
			if (g->desc.get() == desc.get()) return g.get();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		std::unique_ptr<ModelGroup> g;
		if (desc->kind == MODEL_WAVENET) g.reset(new WaveNetGroup(desc, stream, packHint));
		else if (desc->kind == MODEL_LSTM) g.reset(new LstmGroup(desc, stream));
		else throw std::runtime_error("neuralaudio_amd: unsupported model kind");
		groups.push_back(std::move(g));
		return groups.back().get();
	}

	int GpuBatch::AddStream(const std::shared_ptr<const LoadedModel>& model, float quality, bool prewarm, bool onDemand)
	{
		return AddStreams(model, quality, 1, prewarm, onDemand);
	}

	// ids for `count` new streams: retired ones first (the lowest for a single stream, a run of consecutive ones for several), else new rows
	int GpuBatch::AllocateIds(int count)
	{
		for (size_t i = 0; i + (size_t)count <= retired.size(); i++)
		{
			if (retired[i + (size_t)count - 1] == retired[i] + count - 1)
			{
				const int first = retired[i];
				retired.erase(retired.begin() + (long)i, retired.begin() + (long)i + count);
				return first;
			}
		}
		const int first = (int)streams.size();
		streams.resize(streams.size() + (size_t)count);
		for (int i = 0; i < count; i++) streams[(size_t)(first + i)].live = false;
		return first;
	}

	int GpuBatch::AddStreams(const std::shared_ptr<const LoadedModel>& model, float quality, int count, bool prewarm, bool onDemand)
	{
		if (!model || model->subModels.empty()) throw std::runtime_error("neuralaudio_amd: AddStream with an empty model");
		if (count < 1) throw std::runtime_error("neuralaudio_amd: AddStreams with count < 1");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline(); // (state arrays may be re-allocated below)
		topologyVersion++;
		const int active = model->isComposite ? model->ModelIndexFromQuality(quality) : 0;
		const size_t numSub = model->subModels.size();
		std::vector<ModelGroup*> subGroups(numSub);
		std::vector<std::vector<int>> newMembers(numSub);
		for (size_t k = 0; k < numSub; k++) subGroups[k] = GroupFor(model->subModels[k].desc, (!model->isComposite && numSub == 1) ? count : 0);
		const int first = AllocateIds(count);
		// Everything below may throw (hipMalloc inside AddMember / Reset / Prewarm).  A failed call must leave the batch as it was: the
		// members it created are removed again, the ids go back to `retired` (rows appended by AllocateIds leave the arrays).
		int built = 0; // rows [first, first + built) are complete StreamRefs
		std::vector<std::pair<ModelGroup*, int>> partial; // members of the row under construction
		try
		{
			for (int i = 0; i < count; i++)
			{
				StreamRef ref;
				ref.model = model;
				ref.quality = quality;
				ref.active = active;
				ref.onDemand = onDemand;
				ref.live = true;
				ref.prewarmed.assign(numSub, 0);
				const int row = first + i;
				partial.clear();
				for (size_t k = 0; k < numSub; k++)
				{
					const int member = subGroups[k]->AddMember();
					partial.push_back({ subGroups[k], member });
					newMembers[k].push_back(member);
				}
				ref.members = partial;
				partial.clear();
				ref.members[(size_t)active].first->SetActive(ref.members[(size_t)active].second, row);
				streams[(size_t)row] = ref;
				built = i + 1;
			}
			// fresh state for every new member, then prewarm: every submodel (LoadAll, CompositeModel.h:111-118) or only the active one
			// (OnDemand, :104-109 -- the others are prewarmed when a quality change first selects them, :52-60)
			for (size_t k = 0; k < numSub; k++)
			{
				std::sort(newMembers[k].begin(), newMembers[k].end());
				subGroups[k]->Reset(newMembers[k]);
				const bool now = prewarm && (!onDemand || (int)k == active);
				if (now) subGroups[k]->Prewarm(newMembers[k]);
				for (int i = 0; i < count; i++) streams[(size_t)(first + i)].prewarmed[k] = now ? 1 : 0;
			}
		}
		catch (...)
		{
			for (auto& gm : partial) gm.first->RemoveMember(gm.second);
			for (int i = 0; i < count; i++)
			{
				StreamRef& ref = streams[(size_t)(first + i)];
				if (i < built)
					for (auto& gm : ref.members) gm.first->RemoveMember(gm.second);
				ref = StreamRef();
				ref.live = false;
				retired.insert(std::lower_bound(retired.begin(), retired.end(), first + i), first + i);
			}
			DropTrailingRetiredRows();
			throw;
		}
		// the half-batch chains' streams on this, the set-up side (creating a HIP stream takes ~13 ms: not inside the first buffer)
		if (ownsStream && !streamObserved && streams.size() >= 512)
			for (int h = 0; h < numChains; h++)
				if (!halfStream[h]) CheckHip(hipStreamCreateWithFlags(&halfStream[h], hipStreamNonBlocking), "hipStreamCreate");
		return first;
	}

	// trailing retired rows leave the [streams][n] arrays altogether (their ids are the largest entries of the sorted `retired` list)
	void GpuBatch::DropTrailingRetiredRows()
	{
		while (!streams.empty() && !streams.back().live)
		{
			if (!retired.empty() && retired.back() == (int)streams.size() - 1) retired.pop_back();
			streams.pop_back();
		}
	}

	// every buffer still in flight on a slot stream (pipelined interface) is done after this
	void GpuBatch::DrainPipeline()
	{
		for (PipeSlot& p : pipe)
			if (p.own) CheckHip(hipStreamSynchronize(p.own), "hipStreamSynchronize");
		for (hipStream_t hs : halfStream)
			if (hs) CheckHip(hipStreamSynchronize(hs), "hipStreamSynchronize");
		halfChainsUsed = false; // (the next half-batch launches wait for the batch stream first: LaunchHalves)
	}

	hipStream_t GpuBatch::GetStream()
	{
		if (!streamObserved)
		{
			(void)hipSetDevice(device);
			JoinHalves();
			streamObserved = true;
		}
		return stream;
	}

	void GpuBatch::JoinHalves()
	{
		if (!halfChainsUsed) return;
		for (hipStream_t hs : halfStream)
			if (hs) CheckHip(hipStreamSynchronize(hs), "hipStreamSynchronize");
		halfChainsUsed = false;
	}

	struct GpuBatch::HalfLists
	{
		std::vector<WnFrameGroup> part[GpuBatch::kMaxChains];
		bool listsUploaded = false; // an index list went to the device on the batch stream while the lists were built
		bool compact = false;       // some group has compact rings: chunk lengths of WnCompactSafeFrames() only
		// every part is a contiguous range of rows (no index lists, no packed streams): the host can stage and collect a half by itself
		bool RowRangesOnly() const
		{
			for (const auto& list : part)
				for (const WnFrameGroup& g : list)
					if (g.slots != nullptr || g.pack > 1) return false;
			return true;
		}
	};

	// The buffer as two launch lists of half of every group's streams each (see halfStream); false: it runs as ordered launches.
	bool GpuBatch::PrepareHalves(size_t n)
	{
		static const bool off = getenv("NA_HOST_HALVES") != nullptr && atoi(getenv("NA_HOST_HALVES")) == 0; // tuning knob
		if (off) return false;
		bool dirty = false, packed = false, plain = false;
		int active = 0, kernelStreams = 0;
		for (const auto& g : groups)
		{
			const int members = g->NumActive();
			if (members == 0) continue;
			const int c = g->LaunchClass(); // 1 / 2 / -1: the f16-split kernels' plain launch / packed launch / either (gpu_batch.cpp LaunchClass)
			if (c != 1 && c != 2 && c != -1) return false;
			packed = packed || c == 2;
			plain = plain || c == 1;
			dirty = dirty || g->ListsDirty();
			kernelStreams += (members + g->PackFactor() - 1) / g->PackFactor();
			active++;
		}
		if (active == 0 || active > WN_FRAME_MAX_GROUPS || (packed && plain)) return false; // (two launches per buffer: not split)
		if (kernelStreams < 512) return false; // (a small batch: nothing below is worth its host time; the exact count is checked at the end)
		// changed index lists are re-uploaded below (asynchronously, on the batch stream): nothing in flight may still read the old ones
		if (dirty && (halfChainsUsed || pipelineUsed)) DrainPipeline();
		if (!halfLists) halfLists.reset(new HalfLists());
		HalfLists& hl = *halfLists;
		for (auto& part : hl.part) part.clear();
		hl.listsUploaded = dirty;
		int total = 0;
		bool compact = false;
		for (const auto& g : groups)
		{
			if (g->NumActive() == 0) continue;
			WnFrameGroup a = {};
			int list = 0;
			if (!g->FusedLaunchArgs(a, list)) return false;
			WaveNetGroup* wg = static_cast<WaveNetGroup*>(g.get());
			if (list < 0 && packed) a.slots = wg->listSlots; // a plain group in the packed launch passes its index lists
			total += a.numStreams;
			compact = compact || a.model->compact_rings != 0;
			// contiguous parts of whole workgroups (two streams each)
			int first = 0;
			for (int c = 0; c < numChains; c++)
			{
				const int end = c + 1 == numChains ? a.numStreams : std::min(a.numStreams, (int)(((long)a.numStreams * (c + 1) / numChains + 1) & ~1L));
				if (end <= first) continue;
				WnFrameGroup part = a;
				part.numStreams = end - first;
				part.slot0 += first;
				part.row0 += first;
				if (part.slots) part.slots += first;
				if (part.slots || a.pack > 1) part.rows += (size_t)first * (size_t)a.pack;
				hl.part[c].push_back(part);
				first = end;
			}
		}
		// (below 512 kernel-level streams a launch does not fill the chip anyway: nothing to overlap)
		hl.compact = compact;
		(void)n; // (any buffer length: a chain runs the chunks of a long buffer one after the other, LaunchChain)
		return total >= 512;
	}

	// the chains are about to take launches: whatever else is in flight for this batch comes first, and their streams exist
	void GpuBatch::BeginHalves()
	{
		if (!halfChainsUsed || submitTopology != topologyVersion || halfLists->listsUploaded)
		{
			// whatever the batch stream (state resets, prewarms of new streams, index lists) or a slot stream still has in flight comes first
			DrainPipeline();
			CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
			submitTopology = topologyVersion;
		}
		for (int h = 0; h < numChains; h++)
		{
			if (halfStream[h]) continue;
			CheckHip(hipStreamCreateWithFlags(&halfStream[h], hipStreamNonBlocking), "hipStreamCreate");
			if (markOpen)
			{
				if (!marks[1 + h][0]) CheckHip(hipEventCreate(&marks[1 + h][0]), "hipEventCreate");
				CheckHip(hipEventRecord(marks[1 + h][0], halfStream[h]), "hipEventRecord");
			}
		}
		halfChainsUsed = true;
		lastStepHalves = true;
	}

	// list h of PrepareHalves on its own stream, behind that chain's previous launch
	void GpuBatch::LaunchChain(int h, const float* dIn, float* dOut, size_t n, long inStride, long outStride, bool hostRows)
	{
		// Workgroup shape: for rows in HBM, sized for what is resident with both chains on the chip (full-size workgroups: 36.7 vs 37.1 us
		// per 1024 x 128 Standard step); for rows in pinned host memory the half-size workgroups win (43.5-44.4 vs 45.1-46.2 us per buffer
		// host to host: twice the workgroups keep more PCIe reads in flight)
		const std::vector<WnFrameGroup>& part = halfLists->part[h];
		// (trace builds, tools/trace_split_timeline.py: the traced workgroup index exists in every chain's launch -- only chain NA_TRACE_CHAIN stamps)
		long long* const trace = GetWaveNetTraceBuffer();
		static const int traceChain = getenv("NA_TRACE_CHAIN") ? atoi(getenv("NA_TRACE_CHAIN")) : 0;
		if (trace != nullptr && h != traceChain) SetWaveNetTraceBuffer(nullptr);
		if (!part.empty())
		{
			// (a buffer longer than a launch takes: the chunks one after the other on this chain -- the chains still never wait for each other)
			size_t offset = 0, left = n;
			while (left > 0)
			{
				const int chunk = NextWaveNetChunk(left, halfLists->compact);
				CheckHip(LaunchWaveNetSplitFused(part.data(), (int)part.size(), dIn + offset, dOut + offset, inStride, outStride, chunk, halfStream[h],
					hostRows ? 1 : numChains), "WaveNet kernel (half batch)");
				offset += (size_t)chunk;
				left -= (size_t)chunk;
			}
		}
		if (trace != nullptr) SetWaveNetTraceBuffer(trace);
	}

	// the lists of PrepareHalves, each on its own stream; `done`: events to record
	void GpuBatch::LaunchHalves(const float* dIn, float* dOut, size_t n, long inStride, long outStride, hipEvent_t* done, bool hostRows)
	{
		BeginHalves();
		for (int h = 0; h < numChains; h++)
		{
			LaunchChain(h, dIn, dOut, n, inStride, outStride, hostRows);
			if (done) CheckHip(hipEventRecord(done[h], halfStream[h]), "hipEventRecord");
		}
	}

	void GpuBatch::MarkTime(int which)
	{
		if (which < 0 || which > 1) throw std::runtime_error("neuralaudio_amd: MarkTime(0 | 1)");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		// (a chain stream that does not exist yet is created by the first launch that needs it -- 13 ms, not inside a timed window if
		// nothing will run on it -- and gets its start mark then: LaunchHalves)
		markOpen = which == 0;
		for (int i = 0; i <= kMaxChains; i++)
		{
			hipStream_t s = i == 0 ? stream : halfStream[i - 1];
			if (!s || (which == 1 && !marks[i][0])) continue;
			if (!marks[i][which]) CheckHip(hipEventCreate(&marks[i][which]), "hipEventCreate");
			CheckHip(hipEventRecord(marks[i][which], s), "hipEventRecord");
		}
	}

	// polls the closing marks (a benchmark's closing wait should not pay the wake-up latency of a blocking synchronisation)
	void GpuBatch::WaitMarks()
	{
		CheckHip(hipSetDevice(device), "hipSetDevice");
		for (int i = 0; i <= kMaxChains; i++)
			if (marks[i][0] && marks[i][1])
				while (hipEventQuery(marks[i][1]) == hipErrorNotReady) {}
	}

	float GpuBatch::ElapsedMs()
	{
		CheckHip(hipSetDevice(device), "hipSetDevice");
		float longest = 0.0f;
		for (int i = 0; i <= kMaxChains; i++)
		{
			if (!marks[i][0] || !marks[i][1]) continue;
			// (polled: a benchmark's closing wait should not pay the wake-up latency of a blocking synchronisation -- ~25 us of a 20-step run)
			while (hipEventQuery(marks[i][1]) == hipErrorNotReady) {}
			CheckHip(hipEventSynchronize(marks[i][1]), "hipEventSynchronize");
			float ms = 0.0f;
			CheckHip(hipEventElapsedTime(&ms, marks[i][0], marks[i][1]), "hipEventElapsedTime");
			longest = std::max(longest, ms);
		}
		return longest;
	}

	void GpuBatch::RemoveStreams(int first, int count)
	{
		if (count < 1 || first < 0 || (size_t)first + (size_t)count > streams.size()) throw std::runtime_error("neuralaudio_amd: RemoveStreams: id range outside the batch");
		for (int i = 0; i < count; i++)
			if (!streams[(size_t)(first + i)].live) throw std::runtime_error("neuralaudio_amd: RemoveStreams: stream was already removed");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		// the slots may be handed out again right away: nothing of theirs may still be in flight
		DrainPipeline();
		CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
		topologyVersion++;
		for (int i = 0; i < count; i++)
		{
			StreamRef& ref = streams[(size_t)(first + i)];
			for (auto& gm : ref.members) gm.first->RemoveMember(gm.second);
			ref = StreamRef();
			ref.live = false;
			retired.insert(std::lower_bound(retired.begin(), retired.end(), first + i), first + i);
		}
		DropTrailingRetiredRows();
	}

	unsigned GpuBatch::StreamPrewarmedMask(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		unsigned mask = 0;
		for (size_t k = 0; k < ref.prewarmed.size() && k < 32; k++) mask |= ref.prewarmed[k] ? (1u << k) : 0u;
		return mask;
	}

	void GpuBatch::ZeroRetiredRows(float* hostRows, size_t n, size_t rows) const
	{
		for (int id : retired)
			if ((size_t)id < rows) memset(hostRows + (size_t)id * n, 0, n * sizeof(float));
	}

	void GpuBatch::SetQuality(int s, float quality)
	{
		StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) throw std::runtime_error("neuralaudio_amd: stream was removed");
		ref.quality = quality;
		if (!ref.model->isComposite) return;
		const int idx = ref.model->ModelIndexFromQuality(quality);
		if (idx == ref.active) return;
		ref.members[(size_t)ref.active].first->SetActive(ref.members[(size_t)ref.active].second, -1);
		ref.active = idx;
		ref.members[(size_t)idx].first->SetActive(ref.members[(size_t)idx].second, s);
		topologyVersion++;
		if (ref.onDemand && !ref.prewarmed[(size_t)idx])
		{
			// CompositeModel::SetCurrentModelIndex (CompositeModel.h:52-60): first use of this submodel -- NOT real-time safe, which
			// IsQualityChangeRealtimeSafe() reports beforehand
			CheckHip(hipSetDevice(device), "hipSetDevice");
			ref.members[(size_t)idx].first->Prewarm({ ref.members[(size_t)idx].second });
			ref.prewarmed[(size_t)idx] = 1;
		}
	}

	// How many launches a buffer takes once `leaving` has lost / `entering` has gained an active stream (ProcessDevice's grouping: the
	// WaveNet launch lists, the fused recurrent launch, everything else on its own)
	int GpuBatch::LaunchUnitsAfterSwitch(const ModelGroup* leaving, const ModelGroup* entering) const
	{
		bool lists[3] = { false, false, false }, joiner = false, rec = false;
		int singles = 0;
		for (const auto& g : groups)
		{
			int active = g->NumActive();
			if (g.get() == leaving) active -= 1;
			if (g.get() == entering) active += 1;
			if (active <= 0) continue;
			const int c = g->LaunchClass();
			if (c >= 0 && c <= 2) lists[c] = true;
			else if (c == -1) joiner = true;
			else if (c == 3) rec = true;
			else singles++;
		}
		if (joiner && !lists[2]) lists[1] = true; // plain fast-flavour split groups ride in the packed launch when there is one
		return (int)lists[0] + (int)lists[1] + (int)lists[2] + (int)rec + singles;
	}

	// CompositeModel::IsModelChangeRealtimeSafe (CompositeModel.h:44-50, HadInitialPrewarm): false when the target submodel never had its
	// prewarm -- the switch would prewarm it (OnDemand) or run it cold (a stream added without prewarm) -- and false when the batch
	// would take several launches per buffer afterwards: those run as a captured hipGraph, which a switch re-captures.  Otherwise a
	// switch only re-uploads two pinned index lists asynchronously.
	bool GpuBatch::IsQualityChangeRealtimeSafe(int s, float quality) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return false;
		if (!ref.model->isComposite) return true;
		const int idx = ref.model->ModelIndexFromQuality(quality);
		if (idx == ref.active) return true;
		if (!ref.prewarmed[(size_t)idx]) return false;
		return LaunchUnitsAfterSwitch(ref.members[(size_t)ref.active].first, ref.members[(size_t)idx].first) <= 1;
	}

	float GpuBatch::GetQuality(int s) const { return streams.at((size_t)s).quality; }
	int GpuBatch::GetActiveSubModel(int s) const { return streams.at((size_t)s).active; }

	void GpuBatch::Prewarm(int s)
	{
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline();
		StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return;
		// LoadAll: every submodel is prewarmed (CompositeModel.h:111-118); OnDemand: the current one (:104-109)
		for (size_t k = 0; k < ref.members.size(); k++)
		{
			if (ref.onDemand && (int)k != ref.active) continue;
			ref.members[k].first->Prewarm({ ref.members[k].second });
			ref.prewarmed[k] = 1;
		}
	}

	void GpuBatch::ProcessDevice(const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		// A batch on its own stream that nobody has seen: a buffer of one contiguous WaveNet group runs as two free-running half-batch
		// launches (the order of work on the internal streams is not observable from outside; Synchronize() and the host-buffer entry
		// points wait for all of them).  1024 x A1 Standard x 128 frames: 40.1 -> 37.4 us per step.
		if (ownsStream && !streamObserved)
		{
			if (PrepareHalves(n))
			{
				LaunchHalves(dIn, dOut, n, inStride, outStride, nullptr, false);
				return;
			}
		}
		ProcessDeviceOrdered(dIn, dOut, n, inStride, outStride);
	}

	// on the batch stream, behind everything launched so far (the host-buffer entry points: their copies are on that stream; a lone
	// blocking buffer gains nothing from two half launches -- 57 vs 60 us)
	void GpuBatch::ProcessDeviceOrdered(const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		lastStepHalves = false;
		JoinHalves(); // earlier buffers ran as two half-batch chains: this call's kernels come after both
		if (pipelineUsed)
		{
			// buffers submitted through the pipelined interface run on per-slot streams: this call's kernels come after theirs ...
			if (lastKernelStream != stream && lastKernelEvent) CheckHip(hipStreamWaitEvent(stream, lastKernelEvent, 0), "hipStreamWaitEvent");
		}
		ProcessDeviceOn(stream, dIn, dOut, n, inStride, outStride);
		if (pipelineUsed)
		{
			// ... and the next submitted buffer after this call's
			if (!mainDone) CheckHip(hipEventCreateWithFlags(&mainDone, hipEventDisableTiming), "hipEventCreate");
			CheckHip(hipEventRecord(mainDone, stream), "hipEventRecord");
			lastKernelEvent = mainDone;
			lastKernelStream = stream;
		}
	}

	// `launch` != the batch stream is only used for a batch that runs as ONE launch per buffer (Submit checks)
	void GpuBatch::ProcessDeviceOn(hipStream_t launch, const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		int activeGroups = 0;
		for (auto& g : groups) activeGroups += (g->NumActive() > 0);
		if (activeGroups <= 1)
		{
			for (auto& g : groups) g->Process(dIn, dOut, inStride, outStride, n, launch);
			return;
		}
		// Mixed batch.  Groups that can share a launch are fused: all WaveNet groups on the frame kernel into one launch, all LSTM / GRU
		// groups with an LDS-free kernel instance into another (the workgroups of all architectures share the chip, no fork/join per
		// group).  What remains are independent "units" (disjoint rows, disjoint state); one unit runs directly on the batch stream.
		constexpr int NUM_WN_LISTS = 3; // frame kernel | f16-split kernel | f16-split kernel, packed streams: one launch each
		std::vector<WnFrameGroup> fusedWn[NUM_WN_LISTS];
		std::vector<RecurrentGroup> fusedRec;
		std::vector<ModelGroup*> singles;
		ModelGroup* wnOwner[NUM_WN_LISTS] = {}; // lends its side stream / event to the fused unit
		std::vector<std::pair<WnFrameGroup, ModelGroup*>> joiners;
		ModelGroup* recOwner = nullptr;
		for (auto& g : groups)
		{
			if (g->NumActive() == 0) continue;
			WnFrameGroup a = {};
			RecurrentGroup r;
			int list = 0;
			if (g->FusedLaunchArgs(a, list))
			{
				if (list < 0)
				{
					// a plain fast-flavour split group: joins the packed launch if there is one (decided below), else the plain split launch
					a.slots = static_cast<WaveNetGroup*>(g.get())->listSlots;
					joiners.push_back({ a, g.get() });
					continue;
				}
				fusedWn[list].push_back(a);
				if (!wnOwner[list]) wnOwner[list] = g.get();
			}
			else if (g->FusedRecurrentArgs(r))
			{
				fusedRec.push_back(r);
				if (!recOwner) recOwner = g.get();
			}
			else
			{
				g->SyncActiveLists();
				singles.push_back(g.get());
			}
		}
		for (auto& j : joiners)
		{
			const int list = fusedWn[2].empty() ? 1 : 2;
			WnFrameGroup a = j.first;
			if (list == 1 && static_cast<WaveNetGroup*>(j.second)->IsContiguous()) a.slots = nullptr; // the plain kernel's shortcut
			fusedWn[list].push_back(a);
			if (!wnOwner[list]) wnOwner[list] = j.second;
		}
		auto launchWnList = [&](int which, hipStream_t s) {
			const std::vector<WnFrameGroup>& list = fusedWn[which];
			bool compact = false;
			for (const WnFrameGroup& g : list) compact = compact || g.model->compact_rings != 0;
			size_t offset = 0, left = n;
			while (left > 0)
			{
				const int chunk = NextWaveNetChunk(left, compact);
				for (size_t first = 0; first < list.size(); first += WN_FRAME_MAX_GROUPS)
				{
					const int count = (int)std::min<size_t>(list.size() - first, (size_t)WN_FRAME_MAX_GROUPS);
					CheckHip(which == 0 ? LaunchWaveNetFrameFused(list.data() + first, count, dIn + offset, dOut + offset, inStride, outStride, chunk, s)
										: LaunchWaveNetSplitFused(list.data() + first, count, dIn + offset, dOut + offset, inStride, outStride, chunk, s),
						"WaveNet kernel (fused)");
				}
				offset += (size_t)chunk;
				left -= (size_t)chunk;
			}
		};
		auto launchRec = [&](hipStream_t s) {
			size_t offset = 0, left = n;
			while (left > 0)
			{
				const int chunk = (int)std::min<size_t>(left, (size_t)LSTM_MAX_FRAMES);
				for (size_t first = 0; first < fusedRec.size(); first += RECURRENT_MAX_GROUPS)
					CheckHip(LaunchRecurrentDpp(fusedRec.data() + first, (int)std::min<size_t>(fusedRec.size() - first, (size_t)RECURRENT_MAX_GROUPS),
						dIn + offset, dOut + offset, inStride, outStride, chunk, s), "RecurrentDppKernel (fused)");
				offset += (size_t)chunk;
				left -= (size_t)chunk;
			}
		};
		size_t units = (fusedRec.empty() ? 0 : 1) + singles.size();
		for (int l = 0; l < NUM_WN_LISTS; l++) units += fusedWn[l].empty() ? 0 : 1;
		if (units == 1)
		{
			for (int l = 0; l < NUM_WN_LISTS; l++)
				if (!fusedWn[l].empty())
				{
					launchWnList(l, launch);
					return;
				}
			if (!fusedRec.empty()) launchRec(launch);
			else singles[0]->Process(dIn, dOut, inStride, outStride, n, launch);
			return;
		}
		{
			// tuning knob: the units one after the other on the batch stream instead of concurrently on side streams
			static const bool serial = getenv("NA_BATCH_SERIAL") != nullptr;
			if (serial)
			{
				for (int l = 0; l < NUM_WN_LISTS; l++)
					if (!fusedWn[l].empty()) launchWnList(l, stream);
				if (!fusedRec.empty()) launchRec(stream);
				for (ModelGroup* g : singles) g->Process(dIn, dOut, inStride, outStride, n, stream);
				return;
			}
		}
		// Several units: fork onto side streams so their kernels share the GPU, then join back into the batch stream.  The fork/join
		// costs ~5 HIP calls per unit, which would make a buffer host-bound, so the sequence is captured once into a hipGraph and
		// replayed while the call signature (pointers, n, strides) and the active-stream lists stay the same -- the steady state of a
		// real-time host.
		if (!graphCache.empty() && graphCache.front().key.version != topologyVersion)
		{
			for (auto& e : graphCache) (void)hipGraphExecDestroy(e.exec);
			graphCache.clear();
		}
		hipGraphExec_t graphExec = nullptr;
		for (auto& e : graphCache)
			if (e.key.dIn == dIn && e.key.dOut == dOut && e.key.n == n && e.key.inStride == inStride && e.key.outStride == outStride) graphExec = e.exec;
		if (!graphExec)
		{
			if (graphCache.size() >= 16)
			{
				(void)hipGraphExecDestroy(graphCache.front().exec);
				graphCache.erase(graphCache.begin());
			}
			hipGraph_t graph = nullptr;
			CheckHip(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture");
			try
			{
				if (!forkEvent) CheckHip(hipEventCreateWithFlags(&forkEvent, hipEventDisableTiming), "hipEventCreate");
				CheckHip(hipEventRecord(forkEvent, stream), "hipEventRecord");
				auto branch = [&](ModelGroup* owner, const std::function<void(hipStream_t)>& work) {
					hipStream_t side = owner->SideStream();
					CheckHip(hipStreamWaitEvent(side, forkEvent, 0), "hipStreamWaitEvent");
					work(side);
					CheckHip(hipEventRecord(owner->DoneEvent(), side), "hipEventRecord");
					CheckHip(hipStreamWaitEvent(stream, owner->DoneEvent(), 0), "hipStreamWaitEvent");
				};
				for (int l = 0; l < NUM_WN_LISTS; l++)
					if (!fusedWn[l].empty()) branch(wnOwner[l], [&, l](hipStream_t s) { launchWnList(l, s); });
				if (!fusedRec.empty()) branch(recOwner, launchRec);
				for (ModelGroup* g : singles) branch(g, [&](hipStream_t s) { g->Process(dIn, dOut, inStride, outStride, n, s); });
			}
			catch (...)
			{
				(void)hipStreamEndCapture(stream, &graph);
				if (graph) (void)hipGraphDestroy(graph);
				throw;
			}
			CheckHip(hipStreamEndCapture(stream, &graph), "hipStreamEndCapture");
			const hipError_t e = hipGraphInstantiate(&graphExec, graph, nullptr, nullptr, 0);
			(void)hipGraphDestroy(graph);
			CheckHip(e, "hipGraphInstantiate");
			graphCache.push_back({ { dIn, dOut, n, inStride, outStride, topologyVersion }, graphExec });
		}
		CheckHip(hipGraphLaunch(graphExec, stream), "hipGraphLaunch");
	}

	void GpuBatch::EnsureStaging(size_t floats)
	{
		if (floats <= stageFloats) return;
		if (hostStage) (void)hipHostFree(hostStage);
		if (devStage) (void)hipFree(devStage);
		hostStage = nullptr;
		devStage = nullptr;
		stageFloats = 0;
		CheckHip(hipHostMalloc(reinterpret_cast<void**>(&hostStage), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
		CheckHip(hipMalloc(reinterpret_cast<void**>(&devStage), floats * sizeof(float)), "hipMalloc");
		stageFloats = floats;
	}

	// Direct mode (the default; NA_HOST_DIRECT=0 selects the copy engines): the kernels read the block straight from the pinned host buffer
	// and write their output straight into it (every kernel touches `in` once in its prologue and `out` once in its head), so a buffer
	// is ONE launch instead of copy + launch + copy.  The 2 x 512 KB of a 1024 x 128 block still cross PCIe, inside the kernel, but the
	// two asynchronous copies (each ~10 us of latency before its first byte moves) and the waits between them are gone.  Measured on
	// MI355X with this round's kernels (tools/HostPipeBench, 1024 streams x 128 frames; round 2 had it the other way round for the
	// pipelined path, 66.8 vs 61.9 us, and kept the copies):
	//   A1 Standard  Submit..Collect in place 91.8 -> 59.4 us (p50), pipelined 52.9 -> 51.4 us per buffer;  64 streams: 44.9 -> 28.9 us
	//   Nano / Feather / LSTM 1x16 / 2x8      75.8 / 74.6 / 69.4 / 71.2 -> 54.0 / 52.4 / 49.9 / 50.5 us;  A2 96.9 -> 68.4;  4096 Standard 262 -> 180
	static bool HostDirect()
	{
		static const bool direct = getenv("NA_HOST_DIRECT") == nullptr || atoi(getenv("NA_HOST_DIRECT")) != 0;
		return direct;
	}

	// ---- registered host blocks ----
	namespace
	{
		struct HostBlock
		{
			char* host;
			char* dev;
			size_t bytes;
		};
		std::mutex gHostBlocksMutex;
		std::vector<HostBlock> gHostBlocks;
	}
	bool RegisterHostBuffer(void* p, size_t bytes, std::string& error)
	{
		if (!p || bytes == 0) { error = "neuralaudio_amd: RegisterHostBuffer with an empty block"; return false; }
		hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
		if (e != hipSuccess) { error = std::string("neuralaudio_amd: hipHostRegister: ") + hipGetErrorString(e); return false; }
		void* d = nullptr;
		e = hipHostGetDevicePointer(&d, p, 0);
		if (e != hipSuccess || !d)
		{
			(void)hipHostUnregister(p);
			error = std::string("neuralaudio_amd: hipHostGetDevicePointer: ") + hipGetErrorString(e);
			return false;
		}
		std::lock_guard<std::mutex> lock(gHostBlocksMutex);
		gHostBlocks.push_back({ static_cast<char*>(p), static_cast<char*>(d), bytes });
		return true;
	}
	bool UnregisterHostBuffer(void* p)
	{
		std::lock_guard<std::mutex> lock(gHostBlocksMutex);
		for (size_t i = 0; i < gHostBlocks.size(); i++)
		{
			if (gHostBlocks[i].host != p) continue;
			gHostBlocks.erase(gHostBlocks.begin() + (long)i);
			return hipHostUnregister(p) == hipSuccess;
		}
		return false;
	}
	void* RegisteredDevicePointer(const void* p, size_t bytes)
	{
		const char* c = static_cast<const char*>(p);
		std::lock_guard<std::mutex> lock(gHostBlocksMutex);
		for (const HostBlock& b : gHostBlocks)
			if (c >= b.host && c + bytes <= b.host + b.bytes) return b.dev + (c - b.host);
		return nullptr;
	}

	// (Splitting the buffer into chunks so that host copies overlap the DMA was measured and dropped: every extra asynchronous copy /
	// event costs more than it hides -- 1024 x 128 frames: 114 us per call as one piece, 134 / 186 / 282 us in 2 / 4 / 8 chunks.)
	void GpuBatch::ProcessHost(const float* in, float* out, size_t n)
	{
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		const size_t total = streams.size() * n;
		if (HostDirect())
		{
			// blocks the caller registered: the kernels run on them as they are (no staging copies: 81 -> ~62 us for 1024 x 128)
			float* dIn = static_cast<float*>(RegisteredDevicePointer(in, total * sizeof(float)));
			float* dOut = static_cast<float*>(RegisteredDevicePointer(out, total * sizeof(float)));
			if (dIn && dOut)
			{
				ProcessDeviceOrdered(dIn, dOut, n, (long)n, (long)n);
				CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
				ZeroRetiredRows(out, n, streams.size());
				return;
			}
		}
		EnsureStaging(total);
		float* dStage = nullptr;
		const bool direct = HostDirect() && hipHostGetDevicePointer(reinterpret_cast<void**>(&dStage), hostStage, 0) == hipSuccess && dStage != nullptr;
		if (direct && PrepareHalves(n) && halfLists->RowRangesOnly())
		{
			// The blocking call in two halves: the rows of the first half are staged and launched, the second half is staged while the
			// first runs, and the first half's result is copied out while the second still runs -- the two 512 KB host copies of a
			// 1024 x 128 buffer (2 x 10 us) hide behind the kernels: p50 75-77 -> 60 us.
			BeginHalves();
			for (int h = 0; h < numChains; h++)
			{
				for (const WnFrameGroup& g : halfLists->part[h])
					memcpy(hostStage + (size_t)g.row0 * n, in + (size_t)g.row0 * n, (size_t)g.numStreams * n * sizeof(float));
				LaunchChain(h, dStage, dStage, n, (long)n, (long)n, true);
			}
			for (int h = 0; h < numChains; h++)
			{
				CheckHip(hipStreamSynchronize(halfStream[h]), "hipStreamSynchronize");
				for (const WnFrameGroup& g : halfLists->part[h])
					memcpy(out + (size_t)g.row0 * n, hostStage + (size_t)g.row0 * n, (size_t)g.numStreams * n * sizeof(float));
			}
			halfChainsUsed = false; // (both chains are idle again)
			ZeroRetiredRows(out, n, streams.size());
			return;
		}
		memcpy(hostStage, in, total * sizeof(float));
		// (a pinned block the device cannot address -- not seen on MI355X -- goes through the copy engines instead of failing)
		if (direct)
			ProcessDeviceOrdered(dStage, dStage, n, (long)n, (long)n);
		else
		{
			CheckHip(hipMemcpyAsync(devStage, hostStage, total * sizeof(float), hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
			ProcessDeviceOrdered(devStage, devStage, n, (long)n, (long)n);
			CheckHip(hipMemcpyAsync(hostStage, devStage, total * sizeof(float), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H");
		}
		JoinHalves();
		CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
		memcpy(out, hostStage, total * sizeof(float));
		ZeroRetiredRows(out, n, streams.size());
	}

	void GpuBatch::ProcessHostToDevice(const float* in, float* dOut, size_t n, long outStride)
	{
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		const size_t total = streams.size() * n;
		// the previous call's kernels may still be reading the pinned block
		CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
		EnsureStaging(total);
		memcpy(hostStage, in, total * sizeof(float));
		float* dStage = nullptr;
		if (HostDirect() && hipHostGetDevicePointer(reinterpret_cast<void**>(&dStage), hostStage, 0) == hipSuccess && dStage != nullptr)
			ProcessDeviceOrdered(dStage, dOut, n, (long)n, outStride);
		else
		{
			CheckHip(hipMemcpyAsync(devStage, hostStage, total * sizeof(float), hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
			ProcessDeviceOrdered(devStage, dOut, n, (long)n, outStride);
		}
	}

	void GpuBatch::WeightImages(const LoadedModel& model, std::vector<std::pair<void*, size_t>>& out) const
	{
		for (const auto& sub : model.subModels)
			for (const auto& g : groups)
				if (g->desc.get() == sub.desc.get()) g->WeightImages(out);
	}

	void GpuBatch::EnsurePipeSlot(PipeSlot& p, size_t floats)
	{
		if (!p.uploaded)
		{
			CheckHip(hipEventCreateWithFlags(&p.uploaded, hipEventDisableTiming), "hipEventCreate");
			CheckHip(hipEventCreateWithFlags(&p.computed, hipEventDisableTiming), "hipEventCreate");
			CheckHip(hipEventCreateWithFlags(&p.downloaded, hipEventDisableTiming), "hipEventCreate");
			// the set-up side of the pipelined interface: the half-batch chains' streams too (creating a HIP stream takes ~13 ms)
			for (int h = 0; h < numChains; h++)
				if (!halfStream[h]) CheckHip(hipStreamCreateWithFlags(&halfStream[h], hipStreamNonBlocking), "hipStreamCreate");
		}
		if (floats <= p.floats) return;
		if (p.hostIn) (void)hipHostFree(p.hostIn);
		if (p.hostOut) (void)hipHostFree(p.hostOut);
		if (p.dev) (void)hipFree(p.dev);
		p.hostIn = p.hostOut = p.dev = nullptr;
		p.floats = 0;
		CheckHip(hipHostMalloc(reinterpret_cast<void**>(&p.hostIn), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
		CheckHip(hipHostMalloc(reinterpret_cast<void**>(&p.hostOut), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
		CheckHip(hipMalloc(reinterpret_cast<void**>(&p.dev), floats * sizeof(float)), "hipMalloc");
		p.floats = floats;
	}

	int GpuBatch::Submit(const float* in, size_t n)
	{
		if (n == 0 || streams.empty()) throw std::runtime_error("neuralaudio_amd: Submit on an empty batch / buffer");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		const int ticket = nextSlot;
		PipeSlot& p = pipe[ticket];
		if (p.busy) throw std::runtime_error("neuralaudio_amd: Submit with every pipeline slot in flight (Collect the oldest ticket first)");
		const size_t total = streams.size() * n;
		EnsurePipeSlot(p, total);
		p.n = n;
		p.rows = streams.size(); // Collect sizes its copy by THIS (AddStreams / RemoveStreams may run while the ticket is in flight)
		if (in) memcpy(p.hostIn, in, total * sizeof(float)); // nullptr: the caller filled NextInput() in place
		const bool direct = HostDirect(); // (see ProcessHost)
		float *dIn = nullptr, *dOut = nullptr;
		if (direct && hipHostGetDevicePointer(reinterpret_cast<void**>(&dIn), p.hostIn, 0) == hipSuccess && dIn != nullptr &&
			hipHostGetDevicePointer(reinterpret_cast<void**>(&dOut), p.hostOut, 0) == hipSuccess && dOut != nullptr)
		{
			// (a lone buffer gains nothing from being split -- 55-58 vs 60 us Submit .. Collect -- so only with another ticket in flight)
			bool othersInFlight = false;
			for (const PipeSlot& o : pipe) othersInFlight = othersInFlight || (&o != &p && o.busy);
			// (nor does a submission whose caller has the library copy its rows: the host thread is the bottleneck there, 2 x 512 KB of
			// memcpy per buffer, and a second launch only adds to it -- 48-50 vs 52-59 us per buffer)
			if (in == nullptr && (othersInFlight || halfChainsUsed) && PrepareHalves(n))
			{
				// two free-running half-batch chains (see halfStream): each half in submission order on its own stream
				for (int h = 0; h < numChains; h++)
					if (!p.halfDone[h]) CheckHip(hipEventCreateWithFlags(&p.halfDone[h], hipEventDisableTiming), "hipEventCreate");
				LaunchHalves(dIn, dOut, n, (long)n, (long)n, p.halfDone, true);
				halfChainsUsed = true;
				pipelineUsed = true;
				lastKernelEvent = nullptr; // (ProcessDevice after this drains the half streams itself)
				lastKernelStream = nullptr;
				p.onOwnStream = false;
				p.onHalfStreams = true;
				p.busy = true;
				nextSlot = (nextSlot + 1) % kPipelineSlots;
				return ticket;
			}
			JoinHalves(); // back on the batch stream: the half chains first
			ProcessDeviceOn(stream, dIn, dOut, n, (long)n, (long)n);
			CheckHip(hipEventRecord(p.downloaded, stream), "hipEventRecord");
			p.onOwnStream = false;
			p.onHalfStreams = false;
			p.busy = true;
			nextSlot = (nextSlot + 1) % kPipelineSlots;
			return ticket;
		}
		// One launch per buffer (the usual case): the whole buffer -- upload, kernel, download -- rides on the slot's OWN stream, in order,
		// with no event between them; the only cross-stream edge is the stream state: this buffer's kernel waits for the previous
		// buffer's.  The upload of buffer k + 1 (its stream's first operation) overlaps the kernel of buffer k, the download of buffer k
		// (behind its kernel) overlaps the kernel of buffer k + 1.  Per buffer: 2 copies, 1 launch, 1 event wait, 1 event record --
		// the round-2 path cost 2 more waits and 2 more records on the compute stream, 15 us per buffer (tools/microbench/host_pipe_probe.cpp).
		if (LaunchUnitsAfterSwitch(nullptr, nullptr) <= 1)
		{
			JoinHalves(); // (device-pointer steps may have run as half-batch chains: this buffer's kernel comes after both)
			if (!p.own) CheckHip(hipStreamCreateWithFlags(&p.own, hipStreamNonBlocking), "hipStreamCreate");
			bool listsChanged = false, dirty = false;
			for (auto& g : groups) dirty = dirty || (g->NumActive() > 0 && g->ListsDirty());
			if (dirty)
			{
				// the index lists are re-uploaded on the batch stream: not before the kernels still reading the old ones are done
				if (lastKernelEvent && lastKernelStream != stream) CheckHip(hipStreamWaitEvent(stream, lastKernelEvent, 0), "hipStreamWaitEvent");
				for (auto& g : groups)
					if (g->NumActive() > 0) listsChanged = g->SyncActiveLists() || listsChanged;
			}
			if (listsChanged || submitTopology != topologyVersion || !pipelineUsed)
			{
				// everything the batch stream still has in flight for this batch (state resets of new streams, index lists) comes first
				if (!mainDone) CheckHip(hipEventCreateWithFlags(&mainDone, hipEventDisableTiming), "hipEventCreate");
				CheckHip(hipEventRecord(mainDone, stream), "hipEventRecord");
				CheckHip(hipStreamWaitEvent(p.own, mainDone, 0), "hipStreamWaitEvent");
				submitTopology = topologyVersion;
			}
			pipelineUsed = true;
			CheckHip(hipMemcpyAsync(p.dev, p.hostIn, total * sizeof(float), hipMemcpyHostToDevice, p.own), "hipMemcpyAsync H2D");
			if (lastKernelEvent && lastKernelStream != p.own) CheckHip(hipStreamWaitEvent(p.own, lastKernelEvent, 0), "hipStreamWaitEvent");
			ProcessDeviceOn(p.own, p.dev, p.dev, n, (long)n, (long)n);
			CheckHip(hipEventRecord(p.computed, p.own), "hipEventRecord");
			lastKernelEvent = p.computed;
			lastKernelStream = p.own;
			CheckHip(hipMemcpyAsync(p.hostOut, p.dev, total * sizeof(float), hipMemcpyDeviceToHost, p.own), "hipMemcpyAsync D2H");
			p.onOwnStream = true;
			p.onHalfStreams = false;
			p.busy = true;
			nextSlot = (nextSlot + 1) % kPipelineSlots;
			return ticket;
		}
		// several launch units per buffer (a captured hipGraph on the batch stream): copies on the copy streams, events in between
		if (!copyIn)
		{
			CheckHip(hipStreamCreateWithFlags(&copyIn, hipStreamNonBlocking), "hipStreamCreate");
			CheckHip(hipStreamCreateWithFlags(&copyOut, hipStreamNonBlocking), "hipStreamCreate");
		}
		CheckHip(hipMemcpyAsync(p.dev, p.hostIn, total * sizeof(float), hipMemcpyHostToDevice, copyIn), "hipMemcpyAsync H2D");
		CheckHip(hipEventRecord(p.uploaded, copyIn), "hipEventRecord");
		CheckHip(hipStreamWaitEvent(stream, p.uploaded, 0), "hipStreamWaitEvent");
		ProcessDeviceOrdered(p.dev, p.dev, n, (long)n, (long)n);
		CheckHip(hipEventRecord(p.computed, stream), "hipEventRecord");
		CheckHip(hipStreamWaitEvent(copyOut, p.computed, 0), "hipStreamWaitEvent");
		CheckHip(hipMemcpyAsync(p.hostOut, p.dev, total * sizeof(float), hipMemcpyDeviceToHost, copyOut), "hipMemcpyAsync D2H");
		CheckHip(hipEventRecord(p.downloaded, copyOut), "hipEventRecord");
		p.onOwnStream = false;
		p.onHalfStreams = false;
		p.busy = true;
		nextSlot = (nextSlot + 1) % kPipelineSlots;
		return ticket;
	}

	void GpuBatch::Collect(int ticket, float* out)
	{
		if (ticket < 0 || ticket >= kPipelineSlots || !pipe[ticket].busy) throw std::runtime_error("neuralaudio_amd: Collect with an invalid ticket");
		PipeSlot& p = pipe[ticket];
		if (p.onHalfStreams)
		{
			for (int h = 0; h < numChains; h++) CheckHip(hipEventSynchronize(p.halfDone[h]), "hipEventSynchronize");
		}
		else if (p.onOwnStream) CheckHip(hipStreamSynchronize(p.own), "hipStreamSynchronize"); // the download is the stream's last operation
		else CheckHip(hipEventSynchronize(p.downloaded), "hipEventSynchronize");
		// the slot holds the rows the batch had at Submit: ids retired since then are zeroed only inside that block
		if (!retired.empty()) ZeroRetiredRows(p.hostOut, p.n, p.rows);
		if (out) memcpy(out, p.hostOut, p.rows * p.n * sizeof(float)); // nullptr: the caller reads OutputView() in place
		p.busy = false;
	}

	float* GpuBatch::NextInput(size_t n)
	{
		if (n == 0 || streams.empty()) throw std::runtime_error("neuralaudio_amd: NextInput on an empty batch / buffer");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		PipeSlot& p = pipe[nextSlot];
		if (p.busy) throw std::runtime_error("neuralaudio_amd: NextInput with every pipeline slot in flight (Collect the oldest ticket first)");
		EnsurePipeSlot(p, streams.size() * n);
		return p.hostIn;
	}

	const float* GpuBatch::OutputView(int ticket) const
	{
		if (ticket < 0 || ticket >= kPipelineSlots) throw std::runtime_error("neuralaudio_amd: OutputView with an invalid ticket");
		return pipe[ticket].hostOut;
	}

	void GpuBatch::Synchronize()
	{
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline();
		JoinHalves();
		CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
	}

	double GpuBatch::AlgorithmicBytesPerSample(int blockFrames) const
	{
		double sum = 0.0;
		int n = 0;
		for (const auto& g : groups)
		{
			sum += g->AlgorithmicBytesPerSample(blockFrames) * g->NumActive();
			n += g->NumActive();
		}
		return n ? sum / n : 0.0;
	}

	double GpuBatch::MacsPerSample() const
	{
		double sum = 0.0;
		int n = 0;
		for (const auto& g : groups)
		{
			sum += g->MacsPerSample() * g->NumActive();
			n += g->NumActive();
		}
		return n ? sum / n : 0.0;
	}

	const char* GpuBatch::StreamKernelName(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return "";
		return ref.members[(size_t)ref.active].first->KernelName();
	}

	float GpuBatch::StreamInputLimit(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return 0.0f;
		return ref.members[(size_t)ref.active].first->InputLimit();
	}

	int GpuBatch::StreamRangeEvents(int s)
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return 0;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		DrainPipeline();
		int total = 0;
		for (const auto& gm : ref.members) total += gm.first->RangeEvents(gm.second);
		return total;
	}

	int GpuBatch::StreamPackFactor(int s) const
	{
		const StreamRef& ref = streams.at((size_t)s);
		if (!ref.live) return 0;
		return ref.members[(size_t)ref.active].first->PackFactor();
	}

	size_t GpuBatch::StateBytes() const
	{
		size_t total = 0;
		for (const auto& g : groups) total += g->StateBytesPerStream() * (size_t)g->NumInUse();
		return total;
	}
}
