// model_loader.h -- model file (.nam / keras .json / .aidax) -> host-side LoadedModel.
//
// Mirrors the dispatch of NeuralModelLoader::CreateFromJson (NeuralAudio/NeuralModel.cpp:338-581) and
// the metadata readers of NeuralModelImpl (NeuralAudio/NeuralModelImpl.h:30-94).  No device work here.
#pragma once

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "model_desc.h"
#include "na_json.h"

namespace na
{
	struct ModelInfo
	{
		// defaults == NeuralModel's protected members (NeuralAudio/NeuralModel.h:137-145)
		float modelInputLevelDBu = 12.0f;
		float modelOutputLevelDBu = 12.0f;
		float modelLoudnessDB = -18.0f;
		float sampleRate = 48000.0f;
		std::string modelVersion;
		std::vector<std::pair<std::string, std::string>> metadata;
	};

	struct SubModel
	{
		float maxValue = 1.0f;                 // "max_value" of a SlimmableContainer entry
		std::shared_ptr<const ModelDesc> desc;
		ModelInfo info;
	};

	struct LoadedModel
	{
		ModelInfo info;
		bool isComposite = false;              // architecture == "SlimmableContainer" (NeuralModel.cpp:350-358)
		std::vector<SubModel> subModels;       // size 1 when !isComposite; file order
		std::vector<std::pair<float, int>> qualityLevels; // sorted by max_value (CompositeModel.h:183-194)

		// CompositeModel.h:200-213: first sorted level with quality <= max_value, else the last
		int ModelIndexFromQuality(float quality) const;
	};

	struct LoaderOptions
	{
		int externalSampleRate = 48000; // NeuralModel.h:229
		int wavenetMath = MATH_FAST;    // WAVENET_MATH (NeuralAudio/CMakeLists.txt:82-84)
		int lstmMath = MATH_FAST;       // LSTM_MATH (:94-96)
	};

	// `extension` is ".nam", ".json" or ".aidax".  Returns nullptr when no Internal-path engine accepts the
	// model (e.g. keras GRU); throws std::runtime_error / std::out_of_range on malformed files, like the
	// reference (nlohmann exceptions, "Wrong number of weights").
	std::shared_ptr<LoadedModel> LoadModelFromJson(const Json& modelJson, const std::string& extension, const LoaderOptions& opts);
	std::shared_ptr<LoadedModel> LoadModelFromText(const std::string& text, const std::string& extension, const LoaderOptions& opts);
	// the reference's engine-selection predicates for A2-format files (NeuralModel.cpp:159-168, 188-317)
	bool NAMIsA2(const std::string& version);
	bool NAMIsA2Standard(const Json& modelJson);

	// returns nullptr if the file does not exist (NeuralModel.cpp:321-322)
	std::shared_ptr<LoadedModel> LoadModelFromFile(const std::string& path, const LoaderOptions& opts);
}
