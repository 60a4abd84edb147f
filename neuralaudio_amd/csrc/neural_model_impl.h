// neural_model_impl.h -- concrete NeuralModel of this implementation (internal header).
#pragma once

#include <atomic>
#include <memory>

#include <NeuralAudio/NeuralModel.h>

#include "gpu_batch.h"
#include "model_loader.h"

namespace NeuralAudio
{
	class GpuModel : public NeuralModel
	{
	public:
		GpuModel(std::shared_ptr<const na::LoadedModel> loaded, NeuralModelLoader* loader, bool doPrewarm);
		~GpuModel() override;

		bool HasQualityScaling() override;
		float GetQualityScaleFactor() override;
		bool IsQualityChangeRealtimeSafe(float newScaleFactor) override;
		void SetQualityScaleFactor(float scaleFactor) override;
		bool IsStatic() override;
		int GetReceptiveFieldSize() override;
		void Process(float* input, float* output, size_t numSamples) override;
		void Prewarm() override;

		// host-side model, shareable with a many-stream na::GpuBatch
		const std::shared_ptr<const na::LoadedModel>& GetLoadedModel() const { return model; }
		int GetDevice() const { return device; }
		bool IsOnDemand() const { return onDemand; }

	private:
		void EnsureDeviceState();

		std::shared_ptr<const na::LoadedModel> model;
		std::unique_ptr<na::GpuBatch> batch;
		void ApplyPendingQuality(); // audio thread: hand the latest requested quality to the batch

		int device;
		// SetQualityScaleFactor may come from another thread than Process (CompositeModel.h:122,196-197 keeps atomics for that): the
		// setter only stores; the audio thread applies the switch at the top of its next Process().
		std::atomic<float> quality{ 1.0f };
		std::atomic<int> activeIndex{ 0 };
		// bit k: submodel k had its prewarm (CompositeModel::HadInitialPrewarm).  Written by the audio thread after it touched the batch,
		// read by IsQualityChangeRealtimeSafe() from any thread -- which therefore never looks at the batch itself.
		std::atomic<unsigned> prewarmedMask{ 0 };
		void PublishPrewarmState(); // audio thread
		float appliedQuality = 1.0f; // audio thread only
		bool onDemand = false;
		bool prewarmPending;
	};
}
