// neural_model.cpp -- NeuralAudio::NeuralModel / NeuralModelLoader on top of na::GpuBatch.
//
// Counterpart of the reference's L2 adapters + loader entry points:
//   InternalWaveNetModelT / InternalLSTMModelT / *Dyn   NeuralAudio/InternalModel.h:54-126,251-375,177-248,417-538
//   ScalableCompositeModel                              NeuralAudio/CompositeModel.h:127-214
//   NeuralModelLoader::CreateFromFile/Stream/Json       NeuralAudio/NeuralModel.cpp:319-581
// One GpuModel is one stream; its device state is created lazily on the first Process()/Prewarm() so
// that a model can be loaded (and used as a template for a many-stream na::GpuBatch) without a GPU.
#include "neural_model_impl.h"

#include <fstream>
#include <sstream>

namespace NeuralAudio
{
	GpuModel::GpuModel(std::shared_ptr<const na::LoadedModel> loaded, NeuralModelLoader* loader, bool doPrewarm)
		: model(std::move(loaded)), device(loader->GetDevice()), prewarmPending(doPrewarm)
	{
		// NeuralModelImpl::SetModelLoader (NeuralModelImpl.h:12-17)
		SetAudioInputLevelDBu(loader->GetAudioInputLevelDBu());

		const na::ModelInfo& info = model->info;
		modelInputLevelDBu = info.modelInputLevelDBu;
		modelOutputLevelDBu = info.modelOutputLevelDBu;
		modelLoudnessDB = info.modelLoudnessDB;
		sampleRate = info.sampleRate;
		modelVersion = info.modelVersion;
		metadata = info.metadata;

		// ScalableCompositeModel::CreateModelFromNAMJson ends with SetQualityScaleFactor(default) (CompositeModel.h:155)
		quality = model->isComposite ? loader->GetDefaultQualityScaleFactor() : 1.0f;
		activeIndex = model->isComposite ? model->ModelIndexFromQuality(quality) : 0;
		appliedQuality = quality;
		onDemand = loader->GetCompositeModelLoadMode() == ECompositeModelLoadMode::OnDemand; // CompositeModel.h:139
	}

	GpuModel::~GpuModel() {}

	void GpuModel::EnsureDeviceState()
	{
		if (batch) return;
		// build aside and publish only once the stream exists: a throwing AddStream (no device, HIP error) must not leave a batch
		// with zero streams behind, or every later Process() would silently write nothing
		std::unique_ptr<na::GpuBatch> fresh(new na::GpuBatch(device));
		appliedQuality = quality.load();
		fresh->AddStream(model, appliedQuality, prewarmPending, onDemand);
		batch = std::move(fresh);
		prewarmPending = false;
		PublishPrewarmState();
	}

	void GpuModel::PublishPrewarmState() { prewarmedMask.store(batch->StreamPrewarmedMask(0)); }

	bool GpuModel::HasQualityScaling() { return model->isComposite; }

	float GpuModel::GetQualityScaleFactor() { return model->isComposite ? quality.load() : 1.0f; }

	// CompositeModel::IsModelChangeRealtimeSafe (CompositeModel.h:44-50): safe when the target submodel already had its prewarm
	// (HadInitialPrewarm) -- then the switch only re-uploads two pinned index lists asynchronously on the next Process(): no allocation,
	// no synchronisation, and a one-stream batch always runs as one launch.  In OnDemand mode the first switch to a submodel prewarms it;
	// a model created without prewarm has no prewarmed submodel at all.  Like the setter this may be called from any thread: it reads
	// atomics only (the audio thread publishes the prewarm state after every call that can change it).
	bool GpuModel::IsQualityChangeRealtimeSafe(float newScaleFactor)
	{
		if (!model->isComposite) return true;
		const int idx = model->ModelIndexFromQuality(newScaleFactor);
		if (idx == activeIndex.load()) return true;
		return idx >= 0 && idx < 32 && ((prewarmedMask.load() >> idx) & 1u) != 0;
	}

	// May be called from another thread than Process(): stores only (see neural_model_impl.h)
	void GpuModel::SetQualityScaleFactor(float scaleFactor)
	{
		if (!model->isComposite) return;
		quality.store(scaleFactor);
		activeIndex.store(model->ModelIndexFromQuality(scaleFactor));
	}

	void GpuModel::ApplyPendingQuality()
	{
		const float q = quality.load();
		if (q == appliedQuality) return;
		appliedQuality = q;
		batch->SetQuality(0, q);
		PublishPrewarmState();
	}

	bool GpuModel::IsStatic()
	{
		const na::ModelDesc& d = *model->subModels[(size_t)activeIndex.load()].desc;
		return d.kind == na::MODEL_WAVENET ? d.wavenet.isStatic : d.lstm.isStatic;
	}

	int GpuModel::GetReceptiveFieldSize()
	{
		const na::ModelDesc& d = *model->subModels[(size_t)activeIndex.load()].desc;
		return d.kind == na::MODEL_WAVENET ? d.wavenet.ReceptiveFieldSize() : -1;
	}

	void GpuModel::Process(float* input, float* output, size_t numSamples)
	{
		if (numSamples == 0) return;
		EnsureDeviceState();
		ApplyPendingQuality();
		batch->ProcessHost(input, output, numSamples);
	}

	void GpuModel::Prewarm()
	{
		if (!batch)
		{
			prewarmPending = true;
			EnsureDeviceState();
			return;
		}
		ApplyPendingQuality();
		batch->Prewarm(0);
		PublishPrewarmState();
	}

	// ------------------------------------------------------------------------------------------ loader

	bool NeuralModelLoader::SupportsWaveNetLoadMode(EModelLoadMode mode) { return mode == EModelLoadMode::Internal; }

	bool NeuralModelLoader::SupportsLSTMLoadMode(EModelLoadMode mode) { return mode == EModelLoadMode::Internal; }

	NeuralModel* NeuralModelLoader::CreateFromFile(const std::filesystem::path& modelPath, bool doPrewarm)
	{
		if (!std::filesystem::exists(modelPath)) return nullptr; // ref NeuralModel.cpp:321-322
		std::ifstream jsonStream(modelPath, std::ifstream::binary);
		return CreateFromStream(jsonStream, modelPath.extension(), doPrewarm);
	}

	NeuralModel* NeuralModelLoader::CreateFromStream(std::basic_istream<char>& stream, const std::filesystem::path& extension, bool doPrewarm)
	{
		std::stringstream ss;
		ss << stream.rdbuf();
		return CreateFromString(ss.str(), extension, doPrewarm);
	}

	NeuralModel* NeuralModelLoader::CreateFromString(const std::string& jsonText, const std::filesystem::path& extension, bool doPrewarm)
	{
		na::LoaderOptions opts;
		opts.externalSampleRate = externalSampleRate;
		opts.wavenetMath = (wavenetMath == EMathMode::StdMath) ? na::MATH_STD : na::MATH_FAST;
		opts.lstmMath = (lstmMath == EMathMode::StdMath) ? na::MATH_STD : na::MATH_FAST;
		std::shared_ptr<na::LoadedModel> loaded = na::LoadModelFromText(jsonText, extension.string(), opts);
		if (!loaded) return nullptr;
		GpuModel* m = new GpuModel(loaded, this, doPrewarm);
		if (doPrewarm)
		{
			// the reference prewarms inside the factory (NeuralModel.cpp:575-578); do the same when a device is present so
			// the first Process() call is real-time safe.  Without a device the model stays a host-side template.
			if (na::VisibleDeviceCount() > 0)
			{
				try
				{
					m->Prewarm();
				}
				catch (...)
				{
					delete m;
					throw;
				}
			}
		}
		return m;
	}
}
