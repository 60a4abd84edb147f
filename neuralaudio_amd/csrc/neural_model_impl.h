// neural_model_impl.h -- concrete NeuralModel of this implementation (internal header).
#pragma once

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

	private:
		void EnsureDeviceState();

		std::shared_ptr<const na::LoadedModel> model;
		std::unique_ptr<na::GpuBatch> batch;
		int device;
		float quality = 1.0f;
		int activeIndex = 0;
		bool prewarmPending;
	};
}
