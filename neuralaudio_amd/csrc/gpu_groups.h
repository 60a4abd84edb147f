// gpu_groups.h -- model groups of the batched engine (internal to the gpu_batch*.cpp translation units): one group per distinct model of
// a batch holds the model's device tables (weights deduplicated over its streams) and the state slots of all its streams, and knows the
// kernel family that runs them.  WaveNetGroup: the WaveNet kernels (f16-split / frame / generic families, stream packing, padding);
// LstmGroup: the recurrent kernels (LSTM / GRU cells, dense tails).  Reference counterparts: one InternalWaveNetModelT / InternalLSTMModelT
// instance per stream (NeuralAudio/InternalModel.h:84-160, 300-372); the batch axis is this library's.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "gpu_batch.h"
#include "lstm_launch.h"
#include "tuning.h"
#include "wavenet_launch.h"
#include "wavenet_plan.h"

namespace na
{
	inline namespace groups
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

			// a weight image of a model group whose content arrives from a peer device (multi-GPU host, RCCL fan-out): sized, not filled
			void UploadUnless(bool fromPeer, const std::vector<T>& host, hipStream_t s)
			{
				if (!fromPeer) Upload(host, s);
				else if (host.size() > count) Alloc(host.size());
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
		// a group created with its weight images left for a peer device to fill (GpuBatch::SetPeerWeights): they are in place now --
		// whatever the constructor derives from them on the device (the WaveNet prewarm columns) is computed here
		virtual void WeightsArrived() {}
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

	inline namespace groups
	{
		// WaveNet kernel families: "split" = the f16-split MFMA kernel (wavenet_split_kernels.hip), "frame" = the f32 4x4x1-MFMA kernel
		// (wavenet_frame_kernels.hip), "generic" = the runtime-shaped kernel for layer arrays wider than 16 channels
		// (wavenet_generic_kernels.hip; frame-kernel state format).  NA_WN_KERNEL=split|frame|generic forces one for every model it can
		// run (tuning / tests); default: chosen per model (FamilyFor).
		enum WnFamily { WN_FAMILY_AUTO, WN_FAMILY_SPLIT, WN_FAMILY_FRAME, WN_FAMILY_GENERIC };
		inline WnFamily WaveNetFamilyOverride()
		{
			const int k = Tuning::Get().wnKernel;
			return k == 1 ? WN_FAMILY_SPLIT : (k == 2 ? WN_FAMILY_FRAME : (k == 3 ? WN_FAMILY_GENERIC : WN_FAMILY_AUTO));
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
		inline bool SplitAllowed(const WaveNetPlan& plan)
		{
			if (plan.genericOnly || !plan.splitWeightsOk || plan.rings.size() > (size_t)WN_RANGE_EVENT_SLOT) return false;
			if (plan.splitRangeProven) return true;
			const int spec = WaveNetSpecArchId(plan.sstages.data(), (int)plan.sstages.size(), plan.stateF4, (int)(plan.wsplit.size() / 8));
			return spec == WN_SPEC_A2FULL || spec == WN_SPEC_A2LITE;
		}

		inline WnFamily FamilyFor(const WaveNetPlan& plan)
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
		inline int NextWaveNetChunk(size_t left, bool compactRings)
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
				const int mode = Tuning::Get().wnPack;
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
				const bool off = Tuning::Get().wnPadOff;
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

			// Dense packs (wavenet_plan.cpp WaveNetPackCanBeDense: four Nano streams at 16 / 8 instead of 16 / 16 virtual channels -- half the
			// ring traffic of the second array, two tiles per MFMA in its thirteen layers).  Faster at every batch size (round 5, us per
			// 128-frame step dense / 16 x 16: 16 streams 15.3 / 16.0, 1024: 17.7 / 20.0 -- the 16 / 16 layout's one-tile-per-wave chain
			// included --, 4096: 32.4 / 48.2, 8192: 67.6 / 97.0), so it is the layout of every such group.  NA_WN_DENSE=0: the 16 / 16 layout.
			static bool DenseFor(const WaveNetDesc& wn, int pack)
			{
				return pack > 1 && Tuning::Get().wnDense != 0 && WaveNetPackCanBeDense(wn, pack);
			}

			// packHint: 0 = never pack (submodel of a container), otherwise the number of streams the creating AddStreams call brings
			WaveNetGroup(const std::shared_ptr<const ModelDesc>& d, hipStream_t s, int packHint = 0, bool peerWeights = false)
				: ModelGroup(d, s), columnsPending(peerWeights), pack(PackFor(d->wavenet, packHint)), dense(DenseFor(d->wavenet, pack)),
				  plan((pack > 1 || PadFor(d->wavenet)) ? BuildPackedWaveNetPlan(d->wavenet, pack, dense) : PlanForItsFamily(d->wavenet)),
				  family(plan.isVirtual() ? WN_FAMILY_SPLIT : FamilyFor(plan))
			{
				if (plan.isVirtual())
				{
					if (plan.splitFastT != 2) throw std::runtime_error("internal: packed / padded WaveNet plan is not a fast split-kernel plan");
					realPlan = BuildWaveNetPlan(d->wavenet); // bookkeeping (bytes / MACs per REAL stream)
				}
				dStages.Upload(plan.stages, stream);
				// (the weight images -- WeightImages() below -- may be left for a peer device to fill: the multi-GPU host's RCCL fan-out)
				dWpack.UploadUnless(peerWeights, plan.wpack, stream);
				dWpk.UploadUnless(peerWeights, plan.wpk, stream);
				dPrewarm.Upload(plan.prewarm, stream);
				dWeights.UploadUnless(peerWeights, plan.isVirtual() ? plan.packedWeights : d->wavenet.weights, stream);
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
					dWeightsGen.UploadUnless(peerWeights, wg, stream);
				}
				dSStages.Upload(plan.sstages, stream);
				dWsplit.UploadUnless(peerWeights, plan.wsplit, stream);

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
				if (!columnsPending)
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
						// (sharing 1; the state of this group's streams alone beyond the Infinity Cache: non-temporal ring traffic for the long dilations)
						const bool beyondCache = !Tuning::Get().wnNtOff && (size_t)numActive * StateBytesPerStream() > ((size_t)Tuning::Get().wnNtFromMB << 20);
						CheckHip(LaunchWaveNetSplitFused(&g, 1, dIn + offset, dOut + offset, inStride, outStride, chunk, launchStream, 1 | (beyondCache ? WN_SHARING_BEYOND_CACHE : 0)),
							"WaveNetSplitKernel");
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
			void WeightsArrived() override
			{
				if (!columnsPending) return;
				CheckHip(LaunchWaveNetPrewarmColumns(dPrewarm.Get(), (int)plan.prewarm.size(), dWeights.Get(), dCols.Get(), stream),
					"WaveNetPrewarmColumnsKernel");
				columnsPending = false;
			}
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
			bool columnsPending;  // the weight images arrive from a peer device: the prewarm columns are computed then (WeightsArrived)
			const int pack;       // real streams per virtual stream (1: no packing)
			const bool dense;     // ... two streams per channel group in the last array (DenseFor)
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
			LstmGroup(const std::shared_ptr<const ModelDesc>& d, hipStream_t s, bool peerWeights = false) : ModelGroup(d, s)
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
				dev.tailLayers = (int)lstm.tail.size(); // generic keras stack: a chain of dense / conv1d layers instead of the head
				dev.tailWidth = 0;
				dev.tailHistMax = 0;
				for (const DenseLayerDesc& dl : lstm.tail) dev.tailHistMax = std::max(dev.tailHistMax, dl.History());
				int convRows = 0; // rows of conv1d input history behind the recurrent state's rows (zero at reset, like RTNeural's model->reset())
				for (size_t t = 0; t < lstm.tail.size(); t++)
				{
					const DenseLayerDesc& dl = lstm.tail[t];
					dev.tailOff[t] = (int)w.size();
					dev.tailIn[t] = dl.in;
					dev.tailOut[t] = dl.out;
					dev.tailAct[t] = dl.activation;
					dev.tailK[t] = dl.ksize;
					dev.tailDil[t] = dl.dilation;
					dev.tailHistRow[t] = lstm.numLayers * 2 * lstm.hiddenSize + convRows;
					convRows += dl.History() * dl.in;
					dev.tailWidth = std::max(dev.tailWidth, dev.tailHistMax > 0 ? std::max(dl.in, dl.out) : dl.out);
					w.insert(w.end(), dl.w.begin(), dl.w.end());
					w.insert(w.end(), dl.b.begin(), dl.b.end());
					tailMacs += (double)dl.RowLen() * dl.out;
				}
				init.insert(init.end(), (size_t)convRows, 0.0f);
				dW.UploadUnless(peerWeights, w, stream); // (weight images: may be left for a peer device to fill)
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
					dWT.UploadUnless(peerWeights, wt, stream);
					dev.wT = dWT.Get();
				}
				dev.cell = (lstm.cell == CELL_GRU) ? LSTM_CELL_GRU : LSTM_CELL_LSTM;
				dev.numLayers = lstm.numLayers;
				dev.hidden = lstm.hiddenSize;
				dev.math = (lstm.mathMode == MATH_STD) ? LSTM_MATH_STD : LSTM_MATH_FAST;
				numElems = lstm.numLayers * 2 * lstm.hiddenSize + convRows;
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
				const bool noDpp = Tuning::Get().lstmNoDpp || Tuning::Get().gruNoDpp || Tuning::Get().lstmLaneKernel;
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
				const bool noDpp = Tuning::Get().lstmNoDpp || Tuning::Get().gruNoDpp || Tuning::Get().lstmLaneKernel;
				return (!noDpp && RecurrentDppSupported(dev)) ? 3 : -2;
			}
			const char* KernelName() const override
			{
				// (four streams per wave from RecurrentQuadMinStreams() streams in ONE launch: a batch of several recurrent models decides on
				// their total, this name on the group's own count)
				if (RecurrentDppSupported(dev))
					return (RecurrentQuadSupported(dev) && RecurrentQuadMinStreams() > 0 &&
						NumActive() >= RecurrentQuadMinStreams()) ? "RecurrentQuadKernel" : "RecurrentDppKernel";
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
