// NeuralAudio/NeuralModel.h -- public C++ API of the MI355X-native implementation.
//
// API-compatible with the reference's public header (class / method names, argument meaning, defaults
// and ownership follow NeuralAudio/NeuralModel.h:20-231 of mikeoliphant/NeuralAudio): a host that
// compiles against the reference header compiles against this one.  Factories return a raw `new`-ed
// pointer that the caller deletes.  Behind the interface, Process() runs hand-written gfx950 kernels
// through a one-stream na::GpuBatch.  There is NO CPU fallback: creating device state without a usable
// HIP device throws std::runtime_error.
//
// Deviations (additive, or forced by absent third-party code):
//  * CreateFromJson(nlohmann::json&, ...) (ref :153) would put a third-party type in the ABI.  It exists
//    here only as an inline forwarder when the host has already included nlohmann/json; the portable
//    entry point is CreateFromString().
//  * Load modes RTNeural / NAMCore are not available; Set*LoadMode() returns false for them, which is
//    what the reference does when those back-ends are compiled out (ref NeuralModel.cpp:132-157).
//  * New loader knobs: SetDevice() / GetDevice().
#pragma once

#include <cstddef>
#include <filesystem>
#include <istream>
#include <string>
#include <utility>
#include <vector>

// Both classes are exported from libNeuralAudioCAPI.so (the library itself is built with -fvisibility=hidden): a C++ host links
// against it exactly as it would against the reference's static NeuralAudio library (tools/ModelTest is such a host).
#ifndef NEURALAUDIO_API
#define NEURALAUDIO_API __attribute__((visibility("default")))
#endif

#ifndef DEFAULT_QUALITY_SCALE
#define DEFAULT_QUALITY_SCALE 1.0
#endif
#ifndef DEFAULT_INPUT_DBU
#define DEFAULT_INPUT_DBU 12
#endif

namespace NeuralAudio
{
	enum EModelLoadMode { Internal, RTNeural, NAMCore };          // ref :20-25 (values 0,1,2 cross the C ABI)
	enum ECompositeModelLoadMode { LoadAll, OnDemand };           // ref :27-31
	// The reference picks its activation arithmetic at build time (-DWAVENET_MATH / -DLSTM_MATH = FastMath | StdMath,
	// NeuralAudio/CMakeLists.txt:82-96, Activation.h:12-118); here it is a load-time knob with the same two policies.
	// FastMath (the reference default) = the rational tanh of Activation.h:83-96; StdMath = std::tanh / 1/(1+exp(-x)) (:20-45).
	enum EMathMode { FastMath, StdMath };

	// Runtime interface: one instance == one mono audio stream.  Not thread-safe; call Process() from one
	// thread.  The base class is a silent no-op model, exactly as in the reference.
	class NEURALAUDIO_API NeuralModel
	{
	public:
		virtual ~NeuralModel() {}

		// The virtuals are declared in the reference's order (NeuralModel.h:40-134), so the vtable slots agree with a host compiled against
		// the reference header (Itanium ABI: slots follow declaration order; tests/vtable_slots.cpp checks them).
		virtual EModelLoadMode GetLoadMode() { return EModelLoadMode::Internal; }

		// --- quality scaling (A2 slimmable containers) ------------------------------------------------
		virtual bool HasQualityScaling() { return false; }
		virtual float GetQualityScaleFactor() { return 1.0f; }
		virtual bool IsQualityChangeRealtimeSafe(float newScaleFactor) { (void)newScaleFactor; return true; }
		virtual void SetQualityScaleFactor(float scaleFactor) { (void)scaleFactor; }

		virtual bool IsStatic() { return false; }                  // true for the official fixed architectures
		virtual void SetMaxAudioBufferSize(const int maxSize) { (void)maxSize; }

		// --- level calibration --------------------------------------------------------------------------
		virtual void SetAudioInputLevelDBu(float audioDBu) { audioInputLevelDBu = audioDBu; }
		virtual float GetAudioInputLevelDBu() { return audioInputLevelDBu; }
		virtual float GetRecommendedInputDBAdjustment() { return audioInputLevelDBu - modelInputLevelDBu; }
		virtual float GetRecommendedOutputDBAdjustment() { return -18 - modelLoudnessDB; }

		// --- identity ---------------------------------------------------------------------------------------
		virtual float GetSampleRate() { return sampleRate; }
		virtual int GetReceptiveFieldSize() { return -1; }         // -1: unbounded memory (LSTM)
		virtual std::string GetModelVersion() { return modelVersion; }
		// value is the JSON text of the metadata field ("" when absent)
		virtual std::string GetMetadata(const std::string& fieldName)
		{
			for (const auto& kv : metadata)
				if (kv.first == fieldName) return kv.second;
			return "";
		}

		// --- audio ----------------------------------------------------------------------------------------
		// input/output: caller-owned, >= numSamples floats each; input == output is allowed.
		virtual void Process(float* input, float* output, size_t numSamples) { (void)input; (void)output; (void)numSamples; }
		// (re-)establish the zero-input steady state
		virtual void Prewarm() {}

	protected:
		float audioInputLevelDBu = (float)DEFAULT_INPUT_DBU;
		float modelInputLevelDBu = 12;
		float modelOutputLevelDBu = 12;
		float modelLoudnessDB = -18;
		float sampleRate = 48000;
		std::string modelVersion = "";
		std::vector<std::pair<std::string, std::string>> metadata;
	};

	// Factory + load-time settings.  Loading is not real-time safe.
	class NEURALAUDIO_API NeuralModelLoader
	{
	public:
		// nullptr when the file does not exist or no engine accepts the model; throws on malformed files
		NeuralModel* CreateFromFile(const std::filesystem::path& modelPath, bool doPrewarm = true);
		NeuralModel* CreateFromStream(std::basic_istream<char>& stream, const std::filesystem::path& extension, bool doPrewarm = true);
		NeuralModel* CreateFromString(const std::string& jsonText, const std::filesystem::path& extension, bool doPrewarm = true);
#ifdef NLOHMANN_JSON_VERSION_MAJOR
		NeuralModel* CreateFromJson(nlohmann::json& modelJson, const std::filesystem::path& extension, bool doPrewarm = true)
		{
			return CreateFromString(modelJson.dump(), extension, doPrewarm);
		}
#endif

		bool SupportsWaveNetLoadMode(EModelLoadMode mode);
		bool SupportsLSTMLoadMode(EModelLoadMode mode);
		bool SetLSTMLoadMode(EModelLoadMode val) { if (!SupportsLSTMLoadMode(val)) return false; lstmLoadMode = val; return true; }
		bool SetWaveNetLoadMode(EModelLoadMode val) { if (!SupportsWaveNetLoadMode(val)) return false; wavenetLoadMode = val; return true; }

		ECompositeModelLoadMode GetCompositeModelLoadMode() { return compositeLoadMode; }
		void SetCompositeModelLoadMode(ECompositeModelLoadMode loadMode) { compositeLoadMode = loadMode; }

		void SetAudioInputLevelDBu(float audioDBu) { audioInputLevelDBu = audioDBu; }
		float GetAudioInputLevelDBu() { return audioInputLevelDBu; }
		void SetDefaultMaxAudioBufferSize(int maxSize) { defaultMaxAudioBufferSize = maxSize; }
		int GetDefaultMaxAudioBufferSize() { return defaultMaxAudioBufferSize; }
		void SetDefaultQualityScaleFactor(float scaleFactor) { defaultQualityScaleFactor = scaleFactor; }
		float GetDefaultQualityScaleFactor() { return defaultQualityScaleFactor; }
		void SetExternalSampleRate(int sampleRate) { this->externalSampleRate = sampleRate; }

		// MI355X additions: HIP device the created models run on (default 0)
		void SetDevice(int deviceIndex) { device = deviceIndex; }
		int GetDevice() { return device; }
		// math policy of the models created next (the reference's WAVENET_MATH / LSTM_MATH build options)
		void SetWaveNetMathMode(EMathMode mode) { wavenetMath = mode; }
		EMathMode GetWaveNetMathMode() { return wavenetMath; }
		void SetLSTMMathMode(EMathMode mode) { lstmMath = mode; }
		EMathMode GetLSTMMathMode() { return lstmMath; }

	protected:
		EModelLoadMode lstmLoadMode = EModelLoadMode::Internal;
		EModelLoadMode wavenetLoadMode = EModelLoadMode::Internal;
		ECompositeModelLoadMode compositeLoadMode = ECompositeModelLoadMode::LoadAll;
		float audioInputLevelDBu = (float)DEFAULT_INPUT_DBU;
		int defaultMaxAudioBufferSize = 128;
		float defaultQualityScaleFactor = (float)DEFAULT_QUALITY_SCALE;
		int externalSampleRate = 48000;
		int device = 0;
		EMathMode wavenetMath = EMathMode::FastMath;
		EMathMode lstmMath = EMathMode::FastMath;
	};
}
