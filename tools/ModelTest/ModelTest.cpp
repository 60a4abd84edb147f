// ModelTest -- command-line host of the C++ API (include/NeuralAudio/NeuralModel.h) linked against libNeuralAudioCAPI.so.
//
// Counterpart of the reference's Utils/ModelTest (ModelTest.cpp:59-79 BenchModel, :81-118 ComputeError, :120-123 PrintBench,
// :220-267 default model set, :269-337 command line): same command line (`ModelTest [model_file] -b block -q quality`), same
// measurement protocol (4096*64 zero samples through Process() in `block`-sized calls, seconds and x real time at 48 kHz) and the
// same "<engine>: <seconds> (<x>xRT)" line.  The reference compares its three CPU engines with each other; this library has one
// engine, so the RMS line compares two instances of it driven with DIFFERENT call sizes over sin(0.01 n) -- the chunk-invariance
// the reference's Internal path has (its results do not depend on the 64-frame split, InternalModel.h:104-117).
// `--streams N` additionally times N copies of the model as one batch (the data-parallel entry point the GPU path adds).
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <NeuralAudio/NeuralModel.h>
#include <neuralaudio_amd.h>

namespace fs = std::filesystem;
// (the C ABI header declares opaque global structs called NA::NeuralModel / NA::NeuralModelLoader too: the C++ classes stay qualified)
namespace NA = NeuralAudio;
using NA::EModelLoadMode;

namespace
{
	constexpr int kDataSize = 4096 * 64; // samples per timing run (ModelTest.cpp:129)
	const char* kLoadModeNames[] = { "Internal", "RTNeural", "NAMCore" };

	struct Options
	{
		fs::path modelFile;
		int blockSize = 64;
		float quality = 1.0f;
		int streams = 0;
	};

	void Usage()
	{
		std::cerr << "Usage: ModelTest [model_file] [-b|--block_size N] [-q|--quality_scale Q] [--streams N]\n"
					 "  model_file        .nam / .json / .aidax model; default: the sample models of a \"Models\" folder up the path\n"
					 "  -b, --block_size  samples per Process() call (default 64)\n"
					 "  -q, --quality_scale  0.0 fastest .. 1.0 best, for slimmable models (default 1.0)\n"
					 "  --streams N       also time N copies of the model as one GPU batch\n";
	}

	bool ParseArgs(int argc, char** argv, Options& o)
	{
		for (int i = 1; i < argc; i++)
		{
			const std::string a = argv[i];
			auto value = [&](const char* name) -> const char* {
				if (i + 1 >= argc)
				{
					std::cerr << name << ": missing value" << std::endl;
					return nullptr;
				}
				return argv[++i];
			};
			if (a == "-b" || a == "--block_size")
			{
				const char* v = value("--block_size");
				if (!v) return false;
				o.blockSize = atoi(v);
			}
			else if (a == "-q" || a == "--quality_scale")
			{
				const char* v = value("--quality_scale");
				if (!v) return false;
				o.quality = (float)atof(v);
			}
			else if (a == "--streams")
			{
				const char* v = value("--streams");
				if (!v) return false;
				o.streams = atoi(v);
			}
			else if (a == "-h" || a == "--help") return false;
			else if (!a.empty() && a[0] == '-')
			{
				std::cerr << "Unknown argument: " << a << std::endl;
				return false;
			}
			else o.modelFile = a;
		}
		if (o.blockSize < 1)
		{
			std::cerr << "--block_size must be >= 1" << std::endl;
			return false;
		}
		return true;
	}

	std::unique_ptr<NA::NeuralModel> Load(const fs::path& path, NA::NeuralModelLoader& loader, EModelLoadMode mode)
	{
		if (!loader.SetWaveNetLoadMode(mode) || !loader.SetLSTMLoadMode(mode)) return nullptr; // engine not part of this build
		if (!fs::exists(path))
		{
			std::cout << "Model file does not exist: " << path << std::endl;
			return nullptr;
		}
		try
		{
			std::unique_ptr<NA::NeuralModel> model(loader.CreateFromFile(path));
			if (!model)
			{
				std::cout << "Unable to load model from: " << path << std::endl;
				return nullptr;
			}
			if (model->GetLoadMode() != mode) return nullptr;
			if (!model->IsStatic())
				std::cout << "**Warning: " << kLoadModeNames[model->GetLoadMode()] << " model is not using a static architecture" << std::endl;
			return model;
		}
		catch (const std::exception& e)
		{
			std::cout << "Error loading model: " << e.what() << std::endl;
		}
		return nullptr;
	}

	double Seconds(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b)
	{
		return std::chrono::duration<double>(b - a).count();
	}

	// zeros through Process(), `block` samples per call
	double TimeSilence(NA::NeuralModel& model, int block, int numBlocks)
	{
		std::vector<float> in((size_t)block, 0.0f), out((size_t)block, 0.0f);
		const auto t0 = std::chrono::steady_clock::now();
		for (int b = 0; b < numBlocks; b++) model.Process(in.data(), out.data(), (size_t)block);
		return Seconds(t0, std::chrono::steady_clock::now());
	}

	// the same sin(0.01 n) signal through two instances that are called with different block sizes
	double RmsBetween(NA::NeuralModel& a, int blockA, NA::NeuralModel& b, int blockB, int total)
	{
		std::vector<float> x((size_t)total), ya((size_t)total), yb((size_t)total);
		for (int i = 0; i < total; i++) x[(size_t)i] = (float)std::sin(i * 0.01);
		a.Prewarm();
		b.Prewarm();
		for (int pos = 0; pos < total; pos += blockA) a.Process(x.data() + pos, ya.data() + pos, (size_t)std::min(blockA, total - pos));
		for (int pos = 0; pos < total; pos += blockB) b.Process(x.data() + pos, yb.data() + pos, (size_t)std::min(blockB, total - pos));
		double sum = 0.0;
		for (int i = 0; i < total; i++)
		{
			const double d = (double)ya[(size_t)i] - (double)yb[(size_t)i];
			sum += d * d;
		}
		return std::sqrt(sum / total);
	}

	void PrintBench(const std::string& name, double seconds, double samples)
	{
		std::cout << name << ": " << seconds << " (" << (samples / 48000.0) / seconds << "xRT)" << std::endl;
	}

	// N copies of the model as one batch through the C ABI's NA_Batch* entry points (host buffers in, host buffers out)
	void TimeBatch(const fs::path& path, const Options& o, int streams)
	{
		::NeuralModelLoader* loader = CreateLoader();
		NA_SetDefaultQualityScaleFactor(loader, o.quality);
		::NeuralModel* model = NA_CreateModelFromFileUtf8(loader, path.string().c_str(), 0);
		NA_Batch* batch = model ? NA_BatchCreate(0, nullptr) : nullptr;
		if (!batch || NA_BatchAddStreams(batch, model, o.quality, streams, 1) < 0)
		{
			std::cout << "Batch: " << NA_GetLastError() << std::endl;
		}
		else
		{
			const int numBlocks = std::max(1, kDataSize / o.blockSize / 16);
			std::vector<float> in((size_t)streams * o.blockSize, 0.0f), out(in.size());
			NA_BatchProcess(batch, in.data(), out.data(), (size_t)o.blockSize); // first call: tables, staging buffers
			const auto t0 = std::chrono::steady_clock::now();
			for (int b = 0; b < numBlocks; b++) NA_BatchProcess(batch, in.data(), out.data(), (size_t)o.blockSize);
			const double t = Seconds(t0, std::chrono::steady_clock::now());
			PrintBench("Batch x" + std::to_string(streams), t, (double)numBlocks * o.blockSize * streams);
			std::cout << "  per " << o.blockSize << "-sample buffer: " << 1e3 * t / numBlocks << " ms" << std::endl;
		}
		if (batch) NA_BatchDestroy(batch);
		if (model) DeleteModel(model);
		DeleteLoader(loader);
	}

	// false when the model could not be loaded or run
	bool RunModel(const fs::path& path, NA::NeuralModelLoader& loader, const Options& o)
	{
		bool ok = true;
		std::cout << "Model: " << path << std::endl << std::endl;
		const int numBlocks = kDataSize / o.blockSize;
		loader.SetDefaultMaxAudioBufferSize(o.blockSize);

		// the engines the reference tries (ModelTest.cpp:134-136); only Internal exists in this library
		for (EModelLoadMode mode : { EModelLoadMode::RTNeural, EModelLoadMode::NAMCore })
			if (Load(path, loader, mode)) std::cout << kLoadModeNames[mode] << ": unexpectedly available" << std::endl;

		std::unique_ptr<NA::NeuralModel> internal = Load(path, loader, EModelLoadMode::Internal);
		if (!internal)
		{
			std::cout << "Model can't be loaded as internal model" << std::endl << std::endl;
			return false;
		}
		try
		{
			TimeSilence(*internal, o.blockSize, 8); // first calls create the device state
			const double t = TimeSilence(*internal, o.blockSize, numBlocks);
			PrintBench("Internal", t, (double)numBlocks * o.blockSize);

			// two FRESH instances (an LSTM's Prewarm continues from its current state, and `internal` has the timing run behind it)
			std::unique_ptr<NA::NeuralModel> first = Load(path, loader, EModelLoadMode::Internal);
			std::unique_ptr<NA::NeuralModel> second = Load(path, loader, EModelLoadMode::Internal);
			const int otherBlock = (o.blockSize == 37) ? 53 : 37;
			if (first && second)
			{
				const double rms = RmsBetween(*first, o.blockSize, *second, otherBlock, 16384);
				std::cout << "Internal (block " << o.blockSize << ") vs Internal (block " << otherBlock << ") RMS err: " << rms << std::endl;
			}
			if (o.streams > 0) TimeBatch(path, o, o.streams);
		}
		catch (const std::exception& e)
		{
			std::cout << "Error running model: " << e.what() << std::endl;
			ok = false;
		}
		std::cout << std::endl;
		return ok;
	}

	// ModelTest.cpp:220-267: a "Models" folder in the current directory or up the path (here also tests/golden/models of this tree)
	int RunDefaultSet(NA::NeuralModelLoader& loader, Options o)
	{
		fs::path dir = fs::current_path();
		fs::path models;
		for (;;)
		{
			if (fs::exists(dir / "Models")) { models = dir / "Models"; break; }
			if (fs::exists(dir / "tests" / "golden" / "models")) { models = dir / "tests" / "golden" / "models"; break; }
			if (dir == dir.root_path()) break;
			dir = dir.parent_path();
		}
		if (models.empty())
		{
			std::cout << "Unable to find Models: " << fs::current_path() << std::endl;
			std::cout << "ModelTest looks for a \"Models\" folder in current folder or up the path." << std::endl;
			std::cout << "You can also specify a specific model to test by passing the path on the commandline." << std::endl;
			return -1;
		}
		std::cout << "Loading models from: " << models << std::endl << std::endl;
		struct Case { const char* title; const char* file; float quality; };
		const Case cases[] = {
			{ "WaveNet (A2 Full) Test", "BossWN-a2.nam", 1.0f },
			{ "WaveNet (A2 Lite) Test", "BossWN-a2.nam", 0.0f },
			{ "WaveNet (A1 Standard) Test", "BossWN-standard.nam", o.quality },
			{ "LSTM (1x16) Test", "BossLSTM-1x16.nam", o.quality },
		};
		for (const Case& c : cases)
		{
			std::cout << c.title << std::endl;
			loader.SetDefaultQualityScaleFactor(c.quality);
			o.quality = c.quality;
			if (!RunModel(models / c.file, loader, o)) return 2;
		}
		return 0;
	}
}

int main(int argc, char** argv)
{
	Options o;
	if (!ParseArgs(argc, argv, o))
	{
		Usage();
		return 1;
	}
	std::cout << std::endl;
	NA::NeuralModelLoader loader;
	loader.SetDefaultQualityScaleFactor(o.quality);
	std::cout << "Block size: " << o.blockSize << "  Quality Scale: " << o.quality << std::endl;
	if (!o.modelFile.empty())
	{
		return RunModel(o.modelFile, loader, o) ? 0 : 2;
	}
	return RunDefaultSet(loader, o);
}
