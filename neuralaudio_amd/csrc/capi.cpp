// capi.cpp -- extern "C" surface of libNeuralAudioCAPI.so: the 15 legacy symbols of the reference's
// NeuralAudioCAPI (NeuralAudioCAPI/NeuralAudioCApi.cpp:4-97) plus the additive NA_* batch API
// (include/neuralaudio_amd.h).  No exception crosses this boundary.
#include <cstring>
#include <cwchar>
#include <string>

#include "neuralaudio_amd.h"
#include "multi_gpu.h"
#include "neural_model_impl.h"
#include "lstm_launch.h"
#include "wavenet_launch.h"
#include "wavenet_plan.h"
#include "tuning.h"

struct NeuralModel
{
	NeuralAudio::NeuralModel* model;
};

struct NeuralModelLoader
{
	NeuralAudio::NeuralModelLoader* loader;
};

struct NA_Batch
{
	na::GpuBatch* batch;
};

struct NA_MultiBatch
{
	na::MultiGpuBatch* multi;
};

namespace
{
	thread_local std::string g_lastError;

	void SetError(const char* what) { g_lastError = what ? what : "unknown error"; }

	template <typename F>
	int Guard(F&& f)
	{
		try
		{
			f();
			return 0;
		}
		catch (const std::exception& e)
		{
			SetError(e.what());
		}
		catch (...)
		{
			SetError("unknown exception");
		}
		return -1;
	}

	// wchar_t is UTF-32 on Linux and UTF-16 on Windows (the C# caller marshals LPWStr, NativeApi.cs:17)
	std::string WideToUtf8(const wchar_t* w)
	{
		std::string out;
		if (!w) return out;
		for (; *w; ++w)
		{
			unsigned long cp = (unsigned long)*w;
			if (sizeof(wchar_t) == 2 && cp >= 0xD800 && cp <= 0xDBFF && w[1] >= 0xDC00 && w[1] <= 0xDFFF)
			{
				cp = 0x10000 + ((cp - 0xD800) << 10) + ((unsigned long)w[1] - 0xDC00);
				++w;
			}
			if (cp < 0x80) out.push_back((char)cp);
			else if (cp < 0x800)
			{
				out.push_back((char)(0xC0 | (cp >> 6)));
				out.push_back((char)(0x80 | (cp & 0x3F)));
			}
			else if (cp < 0x10000)
			{
				out.push_back((char)(0xE0 | (cp >> 12)));
				out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
				out.push_back((char)(0x80 | (cp & 0x3F)));
			}
			else
			{
				out.push_back((char)(0xF0 | (cp >> 18)));
				out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
				out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
				out.push_back((char)(0x80 | (cp & 0x3F)));
			}
		}
		return out;
	}

	NeuralModel* Wrap(NeuralAudio::NeuralModel* m)
	{
		if (!m) return nullptr;
		NeuralModel* w = new NeuralModel();
		w->model = m;
		return w;
	}

	int CopyOut(const std::string& s, char* buf, int bufSize)
	{
		if (buf && bufSize > 0)
		{
			const size_t n = std::min((size_t)bufSize - 1, s.size());
			memcpy(buf, s.data(), n);
			buf[n] = 0;
		}
		return (int)s.size();
	}
}

extern "C" {

// ---------------------------------------------------------------- legacy symbols (reference NeuralAudioCApi.h:18-46)

NeuralModelLoader* CreateLoader(void)
{
	NeuralModelLoader* loader = nullptr;
	Guard([&] {
		loader = new NeuralModelLoader();
		loader->loader = new NeuralAudio::NeuralModelLoader();
	});
	return loader;
}

void DeleteLoader(NeuralModelLoader* loader)
{
	if (!loader) return;
	delete loader->loader;
	delete loader;
}

NeuralModel* CreateModelFromFile(NeuralModelLoader* loader, const wchar_t* modelPath)
{
	NeuralModel* result = nullptr;
	Guard([&] {
		if (!loader || !modelPath) throw std::runtime_error("CreateModelFromFile: null argument");
		NeuralAudio::NeuralModel* m = loader->loader->CreateFromFile(std::filesystem::path(WideToUtf8(modelPath)));
		if (!m) SetError("model file not found or not supported");
		result = Wrap(m);
	});
	return result;
}

void DeleteModel(NeuralModel* model)
{
	if (!model) return;
	delete model->model;
	delete model;
}

void SetLSTMLoadMode(NeuralModelLoader* loader, int loadMode)
{
	if (loader) loader->loader->SetLSTMLoadMode((NeuralAudio::EModelLoadMode)loadMode);
}

void SetWaveNetLoadMode(NeuralModelLoader* loader, int loadMode)
{
	if (loader) loader->loader->SetWaveNetLoadMode((NeuralAudio::EModelLoadMode)loadMode);
}

void SetAudioInputLevelDBu(NeuralModelLoader* loader, float audioDBu)
{
	if (loader) loader->loader->SetAudioInputLevelDBu(audioDBu);
}

void SetDefaultMaxAudioBufferSize(NeuralModelLoader* loader, int maxSize)
{
	if (loader) loader->loader->SetDefaultMaxAudioBufferSize(maxSize);
}

int GetLoadMode(NeuralModel* model) { return model ? (int)model->model->GetLoadMode() : 0; }

bool IsStatic(NeuralModel* model) { return model ? model->model->IsStatic() : false; }

void SetMaxAudioBufferSize(NeuralModel* model, int maxSize)
{
	if (model) model->model->SetMaxAudioBufferSize(maxSize);
}

float GetRecommendedInputDBAdjustment(NeuralModel* model) { return model ? model->model->GetRecommendedInputDBAdjustment() : 0.0f; }

float GetRecommendedOutputDBAdjustment(NeuralModel* model) { return model ? model->model->GetRecommendedOutputDBAdjustment() : 0.0f; }

float GetSampleRate(NeuralModel* model) { return model ? model->model->GetSampleRate() : 0.0f; }

void Process(NeuralModel* model, float* input, float* output, size_t numSamples)
{
	(void)NA_ProcessChecked(model, input, output, numSamples);
}

// ---------------------------------------------------------------- additive API (include/neuralaudio_amd.h)

const char* NA_GetLastError(void) { return g_lastError.c_str(); }

int NA_GetDeviceCount(void) { return na::VisibleDeviceCount(); }

const char* NA_GetVersion(void) { return "neuralaudio_amd 0.1.0 (gfx950)"; }

NeuralModel* NA_CreateModelFromFileUtf8(NeuralModelLoader* loader, const char* utf8Path, int doPrewarm)
{
	NeuralModel* result = nullptr;
	Guard([&] {
		if (!loader || !utf8Path) throw std::runtime_error("NA_CreateModelFromFileUtf8: null argument");
		NeuralAudio::NeuralModel* m = loader->loader->CreateFromFile(std::filesystem::path(std::string(utf8Path)), doPrewarm != 0);
		if (!m) SetError("model file not found or not supported");
		result = Wrap(m);
	});
	return result;
}

NeuralModel* NA_CreateModelFromString(NeuralModelLoader* loader, const char* jsonText, const char* extension, int doPrewarm)
{
	NeuralModel* result = nullptr;
	Guard([&] {
		if (!loader || !jsonText || !extension) throw std::runtime_error("NA_CreateModelFromString: null argument");
		NeuralAudio::NeuralModel* m = loader->loader->CreateFromString(jsonText, std::filesystem::path(std::string(extension)), doPrewarm != 0);
		if (!m) SetError("model not supported");
		result = Wrap(m);
	});
	return result;
}

// The legacy Process() cannot report failure; an audio host must never be handed uninitialised memory, so a failed call
// (no device, HIP error) returns silence and leaves the reason in NA_GetLastError().
int NA_ProcessChecked(NeuralModel* model, float* input, float* output, size_t numSamples)
{
	if (!model || !output)
	{
		SetError("Process: null argument");
		return -1;
	}
	const int rc = Guard([&] { model->model->Process(input, output, numSamples); });
	if (rc != 0) memset(output, 0, numSamples * sizeof(float));
	return rc;
}

void NA_SetWaveNetMathMode(NeuralModelLoader* loader, int mathMode)
{
	if (loader) loader->loader->SetWaveNetMathMode(mathMode == 1 ? NeuralAudio::EMathMode::StdMath : NeuralAudio::EMathMode::FastMath);
}

void NA_SetLSTMMathMode(NeuralModelLoader* loader, int mathMode)
{
	if (loader) loader->loader->SetLSTMMathMode(mathMode == 1 ? NeuralAudio::EMathMode::StdMath : NeuralAudio::EMathMode::FastMath);
}

void NA_SetCompositeModelLoadMode(NeuralModelLoader* loader, int loadMode)
{
	if (loader) loader->loader->SetCompositeModelLoadMode(loadMode == 1 ? NeuralAudio::ECompositeModelLoadMode::OnDemand : NeuralAudio::ECompositeModelLoadMode::LoadAll);
}

int NA_IsQualityChangeRealtimeSafe(NeuralModel* model, float newQuality)
{
	return (model && model->model->IsQualityChangeRealtimeSafe(newQuality)) ? 1 : 0;
}

// bit 0: NAMIsA2(version), bit 1: NAMIsA2Standard(model) -- the reference's engine-selection predicates, exposed for tests
#ifndef NA_RELEASE
int NA_DebugClassifyNam(const char* jsonText)
{
	int r = -1;
	Guard([&] {
		const na::Json j = na::Json::Parse(jsonText ? jsonText : "");
		const std::string v = (j.IsObject() && j.Contains("version") && j.At("version").IsString()) ? j.At("version").AsString() : "";
		r = (na::NAMIsA2(v) ? 1 : 0) | (na::NAMIsA2Standard(j) ? 2 : 0);
	});
	return r;
}
#endif

// Stream packing, host side only (no GPU needed; tests): the pack factor the model would run with in a large batch (1: none) and, when
// `out` is given, the flat weights of the packed virtual model (wavenet_plan.cpp PackWaveNetDesc); returns their count, -1 on failure.
#ifndef NA_RELEASE
int NA_DebugPackedWeights(NeuralModel* model, int* packFactor, float* out, int capacity)
{
	int r = -1;
	Guard([&] {
		NeuralAudio::GpuModel* gm = model ? dynamic_cast<NeuralAudio::GpuModel*>(model->model) : nullptr;
		if (!gm) throw std::runtime_error("NA_DebugPackedWeights: not a model of this library");
		const auto& lm = gm->GetLoadedModel();
		if (lm->subModels.size() != 1 || lm->subModels[0].desc->kind != na::MODEL_WAVENET)
		{
			if (packFactor) *packFactor = 1;
			r = 0;
			return;
		}
		const na::WaveNetDesc& wn = lm->subModels[0].desc->wavenet;
		const int P = na::WaveNetPackFactor(wn);
		if (packFactor) *packFactor = P;
		if (P < 2)
		{
			r = 0;
			return;
		}
		const bool dense = na::Tuning::Get().wnDense != 0 && na::WaveNetPackCanBeDense(wn, P); // (the layout a batch would give it: four Nano streams at 16 / 8 channels)
		const na::WaveNetDesc v = na::PackWaveNetDesc(wn, P, dense);
		na::ValidateWaveNetDesc(v); // weight count and chaining of the virtual model
		const na::WaveNetPlan plan = na::BuildPackedWaveNetPlan(wn, P, dense);
		if (plan.pack != P || plan.splitFastT != 2) throw std::runtime_error("NA_DebugPackedWeights: the packed plan is not a fast split-kernel plan");
		r = (int)v.weights.size();
		if (out)
			for (int i = 0; i < r && i < capacity; i++) out[i] = v.weights[(size_t)i];
	});
	return r;
}
#endif

void NA_SetDevice(NeuralModelLoader* loader, int device)
{
	if (loader) loader->loader->SetDevice(device);
}

void NA_SetDefaultQualityScaleFactor(NeuralModelLoader* loader, float quality)
{
	if (loader) loader->loader->SetDefaultQualityScaleFactor(quality);
}

void NA_SetExternalSampleRate(NeuralModelLoader* loader, int sampleRate)
{
	if (loader) loader->loader->SetExternalSampleRate(sampleRate);
}

int NA_HasQualityScaling(NeuralModel* model) { return (model && model->model->HasQualityScaling()) ? 1 : 0; }

float NA_GetQualityScaleFactor(NeuralModel* model) { return model ? model->model->GetQualityScaleFactor() : 1.0f; }

void NA_SetQualityScaleFactor(NeuralModel* model, float quality)
{
	if (model) Guard([&] { model->model->SetQualityScaleFactor(quality); });
}

int NA_GetReceptiveFieldSize(NeuralModel* model) { return model ? model->model->GetReceptiveFieldSize() : -1; }

int NA_Prewarm(NeuralModel* model)
{
	if (!model) return -1;
	return Guard([&] { model->model->Prewarm(); });
}

int NA_GetMetadata(NeuralModel* model, const char* fieldName, char* buf, int bufSize)
{
	if (!model || !fieldName) return -1;
	return CopyOut(model->model->GetMetadata(fieldName), buf, bufSize);
}

int NA_GetModelVersion(NeuralModel* model, char* buf, int bufSize)
{
	if (!model) return -1;
	return CopyOut(model->model->GetModelVersion(), buf, bufSize);
}

NA_Batch* NA_BatchCreate(int device, void* hipStream)
{
	NA_Batch* b = nullptr;
	Guard([&] {
		na::GpuBatch* gb = new na::GpuBatch(device, reinterpret_cast<hipStream_t>(hipStream));
		b = new NA_Batch();
		b->batch = gb;
	});
	return b;
}

void NA_BatchDestroy(NA_Batch* batch)
{
	if (!batch) return;
	Guard([&] { delete batch->batch; });
	delete batch;
}

int NA_BatchAddStreams(NA_Batch* batch, NeuralModel* model, float quality, int count, int doPrewarm)
{
	int first = -1;
	const int rc = Guard([&] {
		if (!batch || !model || count < 1) throw std::runtime_error("NA_BatchAddStreams: bad argument");
		NeuralAudio::GpuModel* gm = dynamic_cast<NeuralAudio::GpuModel*>(model->model);
		if (!gm) throw std::runtime_error("NA_BatchAddStreams: model was not created by this library");
		first = batch->batch->AddStreams(gm->GetLoadedModel(), quality, count, doPrewarm != 0, gm->IsOnDemand());
	});
	return rc == 0 ? first : -1;
}

int NA_BatchNumStreams(NA_Batch* batch) { return batch ? batch->batch->NumStreams() : -1; }

int NA_BatchNumLiveStreams(NA_Batch* batch) { return batch ? batch->batch->NumLiveStreams() : -1; }

int NA_BatchIsLive(NA_Batch* batch, int stream) { return (batch && batch->batch->IsLive(stream)) ? 1 : 0; }

int NA_BatchRemoveStreams(NA_Batch* batch, int first, int count)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->RemoveStreams(first, count); });
}

int NA_BatchSetQuality(NA_Batch* batch, int stream, float quality)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->SetQuality(stream, quality); });
}

int NA_BatchIsQualityChangeRealtimeSafe(NA_Batch* batch, int stream, float quality)
{
	int safe = 0;
	if (!batch) return 0;
	Guard([&] { safe = batch->batch->IsQualityChangeRealtimeSafe(stream, quality) ? 1 : 0; });
	return safe;
}

int NA_BatchGetActiveSubModel(NA_Batch* batch, int stream)
{
	int idx = -1;
	if (!batch) return -1;
	Guard([&] { idx = batch->batch->GetActiveSubModel(stream); });
	return idx;
}

int NA_BatchPrewarm(NA_Batch* batch, int stream)
{
	if (!batch) return -1;
	return Guard([&] {
		if (stream >= 0) batch->batch->Prewarm(stream);
		else
			for (int s = 0; s < batch->batch->NumStreams(); s++) batch->batch->Prewarm(s);
	});
}

int NA_BatchProcess(NA_Batch* batch, const float* in, float* out, size_t n)
{
	if (!batch || !out)
	{
		SetError("NA_BatchProcess: null argument");
		return -1;
	}
	const int rc = Guard([&] { batch->batch->ProcessHost(in, out, n); });
	// (an audio host must never be handed stale or uninitialised rows: a failed buffer is silence, like the legacy Process)
	if (rc != 0) memset(out, 0, (size_t)batch->batch->NumStreams() * n * sizeof(float));
	return rc;
}

int NA_RegisterHostBuffer(void* ptr, size_t bytes)
{
	return Guard([&] {
		std::string error;
		if (!na::RegisterHostBuffer(ptr, bytes, error)) throw std::runtime_error(error);
	});
}

int NA_UnregisterHostBuffer(void* ptr) { return na::UnregisterHostBuffer(ptr) ? 0 : -1; }

int NA_BatchSubmit(NA_Batch* batch, const float* in, size_t n)
{
	int ticket = -1;
	if (!batch) return -1;
	const int rc = Guard([&] { ticket = batch->batch->Submit(in, n); });
	return rc == 0 ? ticket : -1;
}

int NA_BatchCollect(NA_Batch* batch, int ticket, float* out)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->Collect(ticket, out); });
}

float* NA_BatchNextInput(NA_Batch* batch, size_t n)
{
	float* ptr = nullptr;
	if (!batch) return nullptr;
	Guard([&] { ptr = batch->batch->NextInput(n); });
	return ptr;
}

const float* NA_BatchOutputView(NA_Batch* batch, int ticket)
{
	const float* ptr = nullptr;
	if (!batch) return nullptr;
	Guard([&] { ptr = batch->batch->OutputView(ticket); });
	return ptr;
}

int NA_BatchProcessDevice(NA_Batch* batch, const float* dIn, float* dOut, size_t n, long inStride, long outStride)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->ProcessDevice(dIn, dOut, n, inStride, outStride); });
}

int NA_BatchSynchronize(NA_Batch* batch)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->Synchronize(); });
}

int NA_BatchSetWaitLimitMs(NA_Batch* batch, double milliseconds)
{
	if (!batch) return -1;
	batch->batch->SetWaitLimitMs(milliseconds);
	return 0;
}

double NA_BatchGetWaitLimitMs(NA_Batch* batch) { return batch ? batch->batch->GetWaitLimitMs() : 0.0; }

int NA_BatchIsBroken(NA_Batch* batch) { return (batch && batch->batch->IsBroken()) ? 1 : 0; }

#ifndef NA_RELEASE
int NA_DebugStallDevice(NA_Batch* batch, double milliseconds)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->DebugStallDevice(milliseconds); });
}
#endif

void* NA_BatchGetHipStream(NA_Batch* batch) { return batch ? reinterpret_cast<void*>(batch->batch->GetStream()) : nullptr; }

int NA_BatchMarkTime(NA_Batch* batch, int which)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->MarkTime(which); });
}

int NA_BatchWaitMarks(NA_Batch* batch)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->WaitMarks(); });
}

float NA_BatchElapsedMs(NA_Batch* batch)
{
	if (!batch) return -1.0f;
	float ms = -1.0f;
	return Guard([&] { ms = batch->batch->ElapsedMs(); }) == 0 ? ms : -1.0f;
}

int NA_BatchUsesHalfLaunches(NA_Batch* batch) { return batch && batch->batch->UsesHalfLaunches() ? 1 : 0; }
int NA_BatchSetResidentLaunch(NA_Batch* batch, int on)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->SetResidentLaunch(on != 0); });
}
int NA_BatchUsesResidentLaunch(NA_Batch* batch) { return batch && batch->batch->UsesResidentLaunch() ? 1 : 0; }

int NA_BatchWaitOutputs(NA_Batch* batch)
{
	if (!batch) return -1;
	return Guard([&] { batch->batch->WaitOutputs(); });
}

double NA_BatchAlgorithmicBytesPerSample(NA_Batch* batch, int blockFrames)
{
	return batch ? batch->batch->AlgorithmicBytesPerSample(blockFrames) : 0.0;
}

double NA_BatchMacsPerSample(NA_Batch* batch) { return batch ? batch->batch->MacsPerSample() : 0.0; }

float NA_BatchStreamInputLimit(NA_Batch* batch, int stream)
{
	try { return batch ? batch->batch->StreamInputLimit(stream) : 0.0f; }
	catch (...) { return 0.0f; }
}

// ---- multi-GPU host (multi_gpu.h) ---------------------------------------------------------------------------------------------
int NA_ShardByCost(const double* cost, int n, int parts, int* bounds)
{
	return Guard([&] {
		if (!cost && n > 0) throw std::runtime_error("NA_ShardByCost: bad argument");
		if (!bounds) throw std::runtime_error("NA_ShardByCost: bad argument");
		const std::vector<int> b = na::ShardByCost(cost, n, parts);
		for (size_t i = 0; i < b.size(); i++) bounds[i] = b[i];
	});
}

double NA_ModelStreamCost(NeuralModel* model, float quality)
{
	double c = -1.0;
	Guard([&] {
		NeuralAudio::GpuModel* gm = model ? dynamic_cast<NeuralAudio::GpuModel*>(model->model) : nullptr;
		if (!gm) throw std::runtime_error("NA_ModelStreamCost: model was not created by this library");
		c = na::EstimateStreamCost(*gm->GetLoadedModel(), quality);
	});
	return c;
}

// Host side only (no GPU needed): which WaveNet kernel family a batch of `streams` streams of the model would run on, and the static
// range proof of the f16-split kernels behind the choice (wavenet_plan.cpp)
int NA_ModelKernelInfo(NeuralModel* model, float quality, int streams, char* kernelBuf, int bufSize, float* inputLimit, int* rangeProven, int* weightsOk, int* packFactor)
{
	int r = -1;
	Guard([&] {
		NeuralAudio::GpuModel* gm = model ? dynamic_cast<NeuralAudio::GpuModel*>(model->model) : nullptr;
		if (!gm) throw std::runtime_error("NA_ModelKernelInfo: model was not created by this library");
		const na::ModelKernelInfo info = na::PredictModelKernel(*gm->GetLoadedModel(), quality, streams);
		if (kernelBuf && bufSize > 0)
		{
			strncpy(kernelBuf, info.kernel, (size_t)bufSize - 1);
			kernelBuf[bufSize - 1] = 0;
		}
		if (inputLimit) *inputLimit = info.inputLimit;
		if (rangeProven) *rangeProven = info.rangeProven ? 1 : 0;
		if (weightsOk) *weightsOk = info.weightsOk ? 1 : 0;
		if (packFactor) *packFactor = info.pack;
		r = 0;
	});
	return r;
}

int NA_BatchStreamRangeEvents(NA_Batch* b, int stream)
{
	int r = -1;
	Guard([&] {
		if (!b) throw std::runtime_error("NA_BatchStreamRangeEvents: null batch");
		r = b->batch->StreamRangeEvents(stream);
	});
	return r;
}

NA_MultiBatch* NA_MultiCreate(const int* devices, int numDevices)
{
	NA_MultiBatch* mb = nullptr;
	Guard([&] {
		if (!devices || numDevices < 1) throw std::runtime_error("NA_MultiCreate: bad argument");
		na::MultiGpuBatch* m = new na::MultiGpuBatch(std::vector<int>(devices, devices + numDevices));
		mb = new NA_MultiBatch{ m };
	});
	return mb;
}

void NA_MultiDestroy(NA_MultiBatch* mb)
{
	if (!mb) return;
	Guard([&] { delete mb->multi; });
	delete mb;
}

int NA_MultiAddStreams(NA_MultiBatch* mb, NeuralModel* model, float quality, int count, int doPrewarm)
{
	int first = -1;
	const int rc = Guard([&] {
		NeuralAudio::GpuModel* gm = (mb && model) ? dynamic_cast<NeuralAudio::GpuModel*>(model->model) : nullptr;
		if (!gm || count < 1) throw std::runtime_error("NA_MultiAddStreams: bad argument");
		first = mb->multi->AddStreams(gm->GetLoadedModel(), quality, count, doPrewarm != 0, gm->IsOnDemand());
	});
	return rc == 0 ? first : -1;
}

int NA_MultiCommit(NA_MultiBatch* mb)
{
	if (!mb) return -1;
	return Guard([&] { mb->multi->Commit(); });
}

int NA_MultiNumStreams(NA_MultiBatch* mb) { return mb ? mb->multi->NumStreams() : -1; }
int NA_MultiNumShards(NA_MultiBatch* mb) { return mb ? mb->multi->NumShards() : -1; }

int NA_MultiShardRange(NA_MultiBatch* mb, int shard, int* begin, int* end, int* device)
{
	if (!mb) return -1;
	return Guard([&] {
		int b = 0, e = 0, d = 0;
		mb->multi->ShardRange(shard, b, e, d);
		if (begin) *begin = b;
		if (end) *end = e;
		if (device) *device = d;
	});
}

int NA_MultiProcess(NA_MultiBatch* mb, const float* in, float* out, size_t n)
{
	if (!mb || !in || !out) return -1;
	return Guard([&] { mb->multi->Process(in, out, n); });
}

int NA_MultiSubmit(NA_MultiBatch* mb, const float* in, size_t n)
{
	int ticket = -1;
	if (!mb || !in) return -1;
	const int rc = Guard([&] { ticket = mb->multi->Submit(in, n); });
	return rc == 0 ? ticket : -1;
}

int NA_MultiCollect(NA_MultiBatch* mb, int ticket, float* out)
{
	if (!mb) return -1;
	return Guard([&] { mb->multi->Collect(ticket, out); });
}

// 0 = host rows (default), 1 = RCCL (see multi_gpu.h FanIn); before NA_MultiCommit / the first NA_MultiProcess
int NA_MultiSetFanIn(NA_MultiBatch* mb, int mode)
{
	if (!mb) return -1;
	return Guard([&] {
		if (mode != 0 && mode != 1) throw std::runtime_error("NA_MultiSetFanIn: mode must be 0 (host rows) or 1 (RCCL)");
		mb->multi->SetFanIn(mode == 1 ? na::MultiGpuBatch::FanIn::Rccl : na::MultiGpuBatch::FanIn::HostRows);
	});
}

const float* NA_MultiGatheredOutput(NA_MultiBatch* mb, int shard)
{
	const float* p = nullptr;
	if (!mb) return nullptr;
	Guard([&] { p = mb->multi->GatheredOutput(shard); });
	return p;
}

// 1 when librccl.so can be loaded and exports every entry point the multi-GPU host binds (rccl_dyn.cpp); no GPU needed
int NA_RcclAvailable(void)
{
	int ok = 0;
	Guard([&] {
		std::string error;
		if (na::rccl::Load(error) == nullptr) throw std::runtime_error(error);
		ok = 1;
	});
	return ok;
}

#ifndef NA_RELEASE
void NA_DebugSetRcclApi(int mode, int failSendAt, int rendezvousMs)
{
	na::rccl::LoopbackConfigure(failSendAt, rendezvousMs);
	na::rccl::SetOverride(mode == 1 ? na::rccl::LoopbackApi() : nullptr);
}
#endif

int NA_MultiSetQuality(NA_MultiBatch* mb, int stream, float quality)
{
	if (!mb) return -1;
	return Guard([&] { mb->multi->SetQuality(stream, quality); });
}

#ifndef NA_RELEASE
void NA_DebugSetWaveNetSpec(int on) { na::SetWaveNetSpecEnabled(on != 0); }
#endif

#ifndef NA_RELEASE
int NA_DebugSetRecurrentQuadMin(int streams) { return na::SetRecurrentQuadMinStreams(streams); }
#endif

#ifndef NA_RELEASE
long long NA_DebugRecurrentQuadLaunches(void) { return (long long)na::RecurrentQuadLaunches(); }
#endif

#ifndef NA_RELEASE
void NA_DebugSetTraceBuffer(void* deviceBuffer) { na::SetWaveNetTraceBuffer(reinterpret_cast<long long*>(deviceBuffer)); }
#endif

double NA_BatchStateBytes(NA_Batch* batch) { return batch ? (double)batch->batch->StateBytes() : 0.0; }

const char* NA_BatchStreamKernelName(NA_Batch* batch, int stream)
{
	try { return batch ? batch->batch->StreamKernelName(stream) : ""; }
	catch (...) { return ""; }
}

int NA_BatchStreamPackFactor(NA_Batch* batch, int stream)
{
	try { return batch ? batch->batch->StreamPackFactor(stream) : 0; }
	catch (...) { return 0; }
}

} // extern "C"
