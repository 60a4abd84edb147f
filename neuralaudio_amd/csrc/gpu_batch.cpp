// gpu_batch.cpp -- see gpu_batch.h.  Host side only: device memory, tables, launches.
#include "gpu_batch.h"

#include <algorithm>
#include <functional>
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

		int AddMember()
		{
			const int member = (int)memberRow.size();
			EnsureCapacity(member + 1);
			memberRow.push_back(-1);
			return member;
		}

		// row >= 0: active, reads/writes that row of the batch arrays; row < 0: inactive (state frozen)
		void SetActive(int member, int row)
		{
			memberRow[(size_t)member] = row;
			activeDirty = true;
		}

		int NumMembers() const { return (int)memberRow.size(); }

		// fresh (never prewarmed) state: zero history / the model's initial h,c
		virtual void Reset(const std::vector<int>& members) = 0;
		virtual void Prewarm(const std::vector<int>& members) = 0;
		// launches on `launchStream` (the batch's main stream, or this group's side stream when several groups run concurrently)
		virtual void Process(const float* dIn, float* dOut, long inStride, long outStride, size_t n, hipStream_t launchStream) = 0;
		virtual double AlgorithmicBytesPerSample(int blockFrames) const = 0;
		virtual double MacsPerSample() const = 0;
		virtual size_t StateBytesPerStream() const = 0;
		// WaveNet groups on the frame kernel can share ONE launch with other such groups (a heterogeneous batch without stream
		// fork/join); fills `out` with this group's part of that launch.  Other groups return false.
		virtual bool FusedLaunchArgs(WnFrameGroup& out)
		{
			(void)out;
			return false;
		}

		// LSTM / GRU groups with an LDS-free kernel instance likewise share one launch (recurrent_dpp_kernels.hip)
		virtual bool FusedRecurrentArgs(RecurrentGroup& out)
		{
			(void)out;
			return false;
		}

		int NumActive() const
		{
			int c = 0;
			for (int r : memberRow) c += (r >= 0);
			return c;
		}

		// upload the active-stream lists if they changed (host -> device copy + sync: never inside a graph capture)
		void SyncActiveLists()
		{
			if (!activeDirty) return;
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
			dSlots.Upload(hSlots, stream);
			dRows.Upload(hRows, stream);
			contiguous = !hSlots.empty();
			for (size_t i = 1; i < hSlots.size() && contiguous; i++)
				contiguous = hSlots[i] == hSlots[0] + (int)i && hRows[i] == hRows[0] + (int)i;
			activeDirty = false;
		}

	protected:
		virtual void EnsureCapacity(int members) = 0;

		hipStream_t stream;
		hipStream_t sideStream = nullptr;
		hipEvent_t doneEvent = nullptr;
		std::vector<int> memberRow; // member == state slot
		std::vector<int> hSlots, hRows;
		DevArray<int> dSlots, dRows;
		bool contiguous = false; // active streams are slot0+i / row0+i: kernels may skip the index arrays
		bool activeDirty = true;
	};

	namespace
	{
		// WaveNet kernel family (process-wide, read once): "split" = the f16-split MFMA kernel (default), "frame" = the f32 4x4x1-MFMA
		// kernel of round 1, "tile" / "pk" = the older measured alternatives.  The families differ in their stream-state format.
		enum WnFamily { WN_FAMILY_SPLIT, WN_FAMILY_FRAME, WN_FAMILY_TILE, WN_FAMILY_PK };
		WnFamily WaveNetFamily()
		{
			static const WnFamily fam = []() {
				const char* e = getenv("NA_WN_KERNEL");
				const std::string w = e ? e : "split";
				if (w == "frame") return WN_FAMILY_FRAME;
				if (w == "tile") return WN_FAMILY_TILE;
				if (w == "pk") return WN_FAMILY_PK;
				return WN_FAMILY_SPLIT;
			}();
			return fam;
		}

		class WaveNetGroup : public ModelGroup
		{
		public:
			WaveNetGroup(const std::shared_ptr<const ModelDesc>& d, hipStream_t s) : ModelGroup(d, s), plan(BuildWaveNetPlan(d->wavenet))
			{
				dStages.Upload(plan.stages, stream);
				dWpack.Upload(plan.wpack, stream);
				dQdesc.Upload(plan.qdesc, stream);
				dWpk.Upload(plan.wpk, stream);
				dPrewarm.Upload(plan.prewarm, stream);
				dWeights.Upload(d->wavenet.weights, stream);
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
				dCols.Alloc(plan.rings.size() * 16);
				CheckHip(LaunchWaveNetPrewarmColumns(dPrewarm.Get(), (int)plan.prewarm.size(), dWeights.Get(), dCols.Get(), stream),
					"WaveNetPrewarmColumnsKernel");

				dev.stages = dStages.Get();
				dev.wpack = dWpack.Get();
				dev.qdesc = dQdesc.Get();
				dev.wpk = dWpk.Get();
				dev.ring_frames = dRingFrames.Get();
				dev.nstages = (int)plan.stages.size();
				dev.nqdesc = (int)plan.qdesc.size();
				dev.wpack_f4 = (int)(plan.wpack.size() / 4);
				dev.max_stage_f4 = plan.maxStageF4;
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
			}

			// ChannelHistoryBuffer::AllocBuffer zero-fills (WaveNet.h:38-40)
			void Reset(const std::vector<int>& members) override
			{
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
				DevArray<int> list;
				list.Upload(members, stream);
				CheckHip(LaunchWaveNetFillRings(state.Get(), plan.stateF4, list.Get(), (int)members.size(), (int)plan.rings.size(),
					dRingOff.Get(), dRingFrames.Get(), dRingG.Get(), dCols.Get(), stream, WaveNetFamily() == WN_FAMILY_SPLIT), "WaveNetFillRingsKernel");
				CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize"); // `list` is freed on return
			}

			void Process(const float* dIn, float* dOut, long inStride, long outStride, size_t n, hipStream_t launchStream) override
			{
				SyncActiveLists();
				const int numActive = (int)hSlots.size();
				if (numActive == 0) return;
				// any n: chunks of <= 128 frames per launch (the reference chunks at 64, InternalModel.h:104-117;
				// results do not depend on the chunking)
				size_t offset = 0;
				while (n > 0)
				{
					const int chunk = (int)std::min<size_t>(n, (size_t)WN_MAX_FRAMES);
					const WnFamily which = WaveNetFamily();
					if (which == WN_FAMILY_SPLIT)
					{
						const WnFrameGroup g = { &dev, state.Get(), contiguous ? nullptr : dSlots.Get(), dRows.Get(), numActive, contiguous ? hSlots[0] : 0, contiguous ? hRows[0] : 0 };
						CheckHip(LaunchWaveNetSplitFused(&g, 1, dIn + offset, dOut + offset, inStride, outStride, chunk, launchStream), "WaveNetSplitKernel");
					}
					else if (which == WN_FAMILY_FRAME)
						CheckHip(LaunchWaveNetFrame(dev, state.Get(), contiguous ? nullptr : dSlots.Get(), dRows.Get(), numActive, dIn + offset, dOut + offset,
							inStride, outStride, chunk, launchStream, contiguous ? hSlots[0] : 0, contiguous ? hRows[0] : 0), "WaveNetFrameKernel");
					else if (which != WN_FAMILY_PK)
						CheckHip(LaunchWaveNetBlock(dev, state.Get(), dSlots.Get(), dRows.Get(), numActive, dIn + offset, dOut + offset, inStride,
							outStride, chunk, launchStream), "WaveNetBlockKernel");
					else
						CheckHip(LaunchWaveNetPk(dev, state.Get(), dSlots.Get(), dRows.Get(), numActive, dIn + offset, dOut + offset, inStride,
							outStride, chunk, launchStream), "WaveNetPkKernel");
					offset += (size_t)chunk;
					n -= (size_t)chunk;
				}
			}

			bool FusedLaunchArgs(WnFrameGroup& out) override
			{
				if (WaveNetFamily() != WN_FAMILY_SPLIT && WaveNetFamily() != WN_FAMILY_FRAME) return false;
				SyncActiveLists();
				out.model = &dev;
				out.state = state.Get();
				out.slots = contiguous ? nullptr : dSlots.Get();
				out.rows = dRows.Get();
				out.numStreams = (int)hSlots.size();
				out.slot0 = contiguous ? hSlots[0] : 0;
				out.row0 = contiguous ? hRows[0] : 0;
				return out.numStreams > 0;
			}

			double AlgorithmicBytesPerSample(int blockFrames) const override { return plan.AlgorithmicBytesPerSample(blockFrames); }
			double MacsPerSample() const override { return plan.MacsPerSample(); }
			size_t StateBytesPerStream() const override { return (size_t)plan.stateF4 * 16; }

		protected:
			void EnsureCapacity(int members) override
			{
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
			WaveNetPlan plan;
			WnModelDev dev = {};
			DevArray<WnStage> dStages;
			DevArray<float> dWpack;
			DevArray<WnQuad> dQdesc;
			DevArray<float> dWpk;
			DevArray<WnPrewarmLayer> dPrewarm;
			DevArray<float> dWeights;
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
				dW.Upload(w, stream);
				dInit.Upload(init, stream);
				dev.w = dW.Get();
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
				while (n > 0)
				{
					const int chunk = (int)std::min<size_t>(n, (size_t)LSTM_MAX_FRAMES);
					CheckHip(Launch(dSlots.Get(), dRows.Get(), numActive, dIn + offset, dOut + offset, inStride, outStride, chunk, launchStream), "recurrent kernel");
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
				out.slots = dSlots.Get();
				out.rows = dRows.Get();
				out.numStreams = (int)hSlots.size();
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
				return macs + lstm.hiddenSize;
			}

			size_t StateBytesPerStream() const override { return (size_t)numElems * sizeof(float); }

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
			DevArray<float> dW, dInit, dZeros;
			DevArray<float> state;
			std::vector<float> init;
			int numElems = 0;
			size_t capacity = 0;
		};
	}
}

namespace na
{
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
		}
		else
		{
			CheckHip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
		}
	}

	GpuBatch::~GpuBatch()
	{
		(void)hipSetDevice(device);
		if (stream) (void)hipStreamSynchronize(stream);
		groups.clear();
		if (hostStage) (void)hipHostFree(hostStage);
		if (devStage) (void)hipFree(devStage);
		for (auto& e : graphCache) (void)hipGraphExecDestroy(e.exec);
		if (forkEvent) (void)hipEventDestroy(forkEvent);
		if (stream && ownsStream) (void)hipStreamDestroy(stream);
	}

	ModelGroup* GpuBatch::GroupFor(const std::shared_ptr<const ModelDesc>& desc)
	{
		for (auto& g : groups)
			if (g->desc.get() == desc.get()) return g.get();
		CheckHip(hipSetDevice(device), "hipSetDevice");
		std::unique_ptr<ModelGroup> g;
		if (desc->kind == MODEL_WAVENET) g.reset(new WaveNetGroup(desc, stream));
		else if (desc->kind == MODEL_LSTM) g.reset(new LstmGroup(desc, stream));
		else throw std::runtime_error("neuralaudio_amd: unsupported model kind");
		groups.push_back(std::move(g));
		return groups.back().get();
	}

	int GpuBatch::AddStream(const std::shared_ptr<const LoadedModel>& model, float quality, bool prewarm)
	{
		return AddStreams(model, quality, 1, prewarm);
	}

	int GpuBatch::AddStreams(const std::shared_ptr<const LoadedModel>& model, float quality, int count, bool prewarm)
	{
		if (!model || model->subModels.empty()) throw std::runtime_error("neuralaudio_amd: AddStream with an empty model");
		if (count < 1) throw std::runtime_error("neuralaudio_amd: AddStreams with count < 1");
		CheckHip(hipSetDevice(device), "hipSetDevice");
		topologyVersion++;
		const int first = (int)streams.size();
		const int active = model->isComposite ? model->ModelIndexFromQuality(quality) : 0;
		const size_t numSub = model->subModels.size();
		std::vector<ModelGroup*> subGroups(numSub);
		std::vector<std::vector<int>> newMembers(numSub);
		for (size_t k = 0; k < numSub; k++) subGroups[k] = GroupFor(model->subModels[k].desc);
		for (int i = 0; i < count; i++)
		{
			StreamRef ref;
			ref.model = model;
			ref.quality = quality;
			ref.active = active;
			const int row = first + i;
			for (size_t k = 0; k < numSub; k++)
			{
				const int member = subGroups[k]->AddMember();
				newMembers[k].push_back(member);
				ref.members.push_back({ subGroups[k], member });
			}
			ref.members[(size_t)active].first->SetActive(ref.members[(size_t)active].second, row);
			streams.push_back(ref);
		}
		// fresh state for every new member, then (LoadAll semantics, CompositeModel.h:111-118) prewarm every submodel
		for (size_t k = 0; k < numSub; k++)
		{
			subGroups[k]->Reset(newMembers[k]);
			if (prewarm) subGroups[k]->Prewarm(newMembers[k]);
		}
		return first;
	}

	void GpuBatch::SetQuality(int s, float quality)
	{
		StreamRef& ref = streams.at((size_t)s);
		ref.quality = quality;
		if (!ref.model->isComposite) return;
		const int idx = ref.model->ModelIndexFromQuality(quality);
		if (idx == ref.active) return;
		ref.members[(size_t)ref.active].first->SetActive(ref.members[(size_t)ref.active].second, -1);
		ref.active = idx;
		ref.members[(size_t)idx].first->SetActive(ref.members[(size_t)idx].second, s);
		topologyVersion++;
	}

	float GpuBatch::GetQuality(int s) const { return streams.at((size_t)s).quality; }
	int GpuBatch::GetActiveSubModel(int s) const { return streams.at((size_t)s).active; }

	void GpuBatch::Prewarm(int s)
	{
		CheckHip(hipSetDevice(device), "hipSetDevice");
		StreamRef& ref = streams.at((size_t)s);
		// LoadAll semantics (CompositeModel.h:111-118): every submodel is prewarmed
		for (auto& m : ref.members) m.first->Prewarm({ m.second });
	}

	void GpuBatch::ProcessDevice(const float* dIn, float* dOut, size_t n, long inStride, long outStride)
	{
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		int activeGroups = 0;
		for (auto& g : groups) activeGroups += (g->NumActive() > 0);
		if (activeGroups <= 1)
		{
			for (auto& g : groups) g->Process(dIn, dOut, inStride, outStride, n, stream);
			return;
		}
		// Mixed batch.  Groups that can share a launch are fused: all WaveNet groups on the frame kernel into one launch, all LSTM / GRU
		// groups with an LDS-free kernel instance into another (the workgroups of all architectures share the chip, no fork/join per
		// group).  What remains are independent "units" (disjoint rows, disjoint state); one unit runs directly on the batch stream.
		std::vector<WnFrameGroup> fusedWn;
		std::vector<RecurrentGroup> fusedRec;
		std::vector<ModelGroup*> singles;
		ModelGroup* wnOwner = nullptr;  // lends its side stream / event to the fused unit
		ModelGroup* recOwner = nullptr;
		for (auto& g : groups)
		{
			if (g->NumActive() == 0) continue;
			WnFrameGroup a;
			RecurrentGroup r;
			if (g->FusedLaunchArgs(a))
			{
				fusedWn.push_back(a);
				if (!wnOwner) wnOwner = g.get();
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
		auto launchWn = [&](hipStream_t s) {
			size_t offset = 0, left = n;
			while (left > 0)
			{
				const int chunk = (int)std::min<size_t>(left, (size_t)WN_MAX_FRAMES);
				for (size_t first = 0; first < fusedWn.size(); first += WN_FRAME_MAX_GROUPS)
					CheckHip((WaveNetFamily() == WN_FAMILY_SPLIT ? LaunchWaveNetSplitFused : LaunchWaveNetFrameFused)(fusedWn.data() + first,
						(int)std::min<size_t>(fusedWn.size() - first, (size_t)WN_FRAME_MAX_GROUPS), dIn + offset, dOut + offset, inStride, outStride, chunk, s),
						"WaveNet kernel (fused)");
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
		const size_t units = (fusedWn.empty() ? 0 : 1) + (fusedRec.empty() ? 0 : 1) + singles.size();
		if (units == 1)
		{
			if (!fusedWn.empty()) launchWn(stream);
			else if (!fusedRec.empty()) launchRec(stream);
			else singles[0]->Process(dIn, dOut, inStride, outStride, n, stream);
			return;
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
				if (!fusedWn.empty()) branch(wnOwner, launchWn);
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

	void GpuBatch::ProcessHost(const float* in, float* out, size_t n)
	{
		if (n == 0 || streams.empty()) return;
		CheckHip(hipSetDevice(device), "hipSetDevice");
		const size_t total = streams.size() * n;
		EnsureStaging(total);
		memcpy(hostStage, in, total * sizeof(float));
		CheckHip(hipMemcpyAsync(devStage, hostStage, total * sizeof(float), hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D");
		ProcessDevice(devStage, devStage, n, (long)n, (long)n);
		CheckHip(hipMemcpyAsync(hostStage, devStage, total * sizeof(float), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H");
		CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
		memcpy(out, hostStage, total * sizeof(float));
	}

	void GpuBatch::Synchronize()
	{
		CheckHip(hipSetDevice(device), "hipSetDevice");
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

	size_t GpuBatch::StateBytes() const
	{
		size_t total = 0;
		for (const auto& g : groups) total += g->StateBytesPerStream() * (size_t)g->NumMembers();
		return total;
	}
}
