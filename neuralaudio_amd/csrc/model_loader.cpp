// model_loader.cpp -- see model_loader.h.
#include "model_loader.h"
#include "lstm_dev.h"

#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <cmath>
#include <sstream>
#include <stdexcept>

namespace na
{
	int LoadedModel::ModelIndexFromQuality(float quality) const
	{
		int modelIndex = 0;
		for (const auto& level : qualityLevels)
		{
			modelIndex = level.second;
			if (quality <= level.first) break;
		}
		return modelIndex;
	}

	namespace
	{
		const std::vector<int> kStdDilations = { 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 };              // NeuralModel.cpp:71
		const std::vector<int> kLiteDilations = { 1, 2, 4, 8, 16, 32, 64 };                            // :72
		const std::vector<int> kLiteDilations2 = { 128, 256, 512, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512 }; // :73
		const std::vector<int> kA2KernelSizes = { 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 15, 15, 6, 6, 6, 6, 6, 6, 6 }; // :75
		const std::vector<int> kA2Dilations = { 1, 3, 7, 17, 41, 101, 239, 1, 3, 7, 17, 41, 101, 239, 1, 13, 1, 3, 7, 17, 41, 101, 239 }; // :76

		std::vector<int> IntArray(const Json& j)
		{
			std::vector<int> v;
			for (size_t i = 0; i < j.Size(); i++) v.push_back(j.At(i).AsInt());
			return v;
		}

		bool Truthy(const Json& j)
		{
			if (j.IsBool()) return j.AsBool();
			if (j.IsNumber()) return j.AsDouble() != 0.0;
			return !j.IsNull();
		}

		// NeuralModelImpl.h:30-60
		void ReadNAMConfig(const Json& modelJson, ModelInfo& info)
		{
			info.modelVersion = modelJson.At("version").AsString();
			if (modelJson.Contains("sample_rate") && modelJson.At("sample_rate").IsNumber())
				info.sampleRate = modelJson.At("sample_rate").AsFloat();
			if (modelJson.Contains("metadata") && modelJson.At("metadata").IsObject())
			{
				const Json& md = modelJson.At("metadata");
				for (const auto& key : md.Keys())
				{
					const Json& value = md.At(key);
					if (!value.IsNull()) info.metadata.push_back({ key, value.Dump() }); // :85-94
				}
				if (md.Contains("loudness") && md.At("loudness").IsNumber()) info.modelLoudnessDB = md.At("loudness").AsFloat();
				if (md.Contains("input_level_dbu") && md.At("input_level_dbu").IsNumber())
					info.modelInputLevelDBu = md.At("input_level_dbu").AsFloat();
				if (md.Contains("output_level_dbu") && md.At("output_level_dbu").IsNumber())
					info.modelOutputLevelDBu = md.At("output_level_dbu").AsFloat();
			}
		}

		// NeuralModelImpl.h:62-78
		void ReadKerasConfig(const Json& modelJson, ModelInfo& info)
		{
			if (modelJson.Contains("samplerate") && modelJson.At("samplerate").IsNumber())
				info.sampleRate = modelJson.At("samplerate").AsFloat();
			if (modelJson.Contains("in_gain") && modelJson.At("in_gain").IsNumber())
				info.modelInputLevelDBu = modelJson.At("in_gain").AsFloat();
			if (modelJson.Contains("out_gain") && modelJson.At("out_gain").IsNumber())
				info.modelLoudnessDB = -18.0f - modelJson.At("out_gain").AsFloat();
		}

		int ActivationFromJson(const Json& act)
		{
			// A1: "Tanh"; A2: [{type:"LeakyReLU", negative_slope:0.01}, ...] (one per layer, all equal in supported files)
			const Json* a = &act;
			if (act.IsArray())
			{
				if (act.Size() == 0) return ACT_TANH;
				a = &act.At(0);
			}
			std::string name;
			if (a->IsString()) name = a->AsString();
			else if (a->IsObject() && a->Contains("type")) name = a->At("type").AsString();
			if (name == "Tanh") return ACT_TANH;
			if (name == "LeakyReLU") return ACT_LEAKYRELU;
			throw std::runtime_error("WaveNet activation '" + name + "' is not supported by the Internal path");
		}

		// A2-format features the Internal path has no arithmetic for.  The reference tests them in NAMIsA2Standard
		// (NeuralModel.cpp:188-317) and hands such files to the NAM Core back-end; without that back-end they cannot be evaluated, so
		// they are rejected loudly here instead of being mis-evaluated.  (Unlike the reference, a MISSING optional block counts as
		// inactive: a block that is not described has no weights in the file either.)
		bool BlockActive(const Json& lc, const char* name)
		{
			if (!lc.Contains(name) || !lc.At(name).IsObject()) return false;
			const Json& b = lc.At(name);
			return b.Contains("active") && Truthy(b.At("active"));
		}

		void RejectUnsupportedA2Features(const Json& lc, int channels)
		{
			auto fail = [](const std::string& what) {
				throw std::runtime_error("WaveNet feature not supported by the Internal path (needs the NAM Core back-end): " + what);
			};
			if (lc.Contains("bottleneck") && lc.At("bottleneck").IsNumber() && lc.At("bottleneck").AsInt() != channels) fail("bottleneck != channels");
			if (lc.Contains("secondary_activation") && lc.At("secondary_activation").IsArray())
				for (size_t i = 0; i < lc.At("secondary_activation").Size(); i++)
					if (!lc.At("secondary_activation").At(i).IsNull()) fail("secondary_activation");
			if (lc.Contains("gating_mode") && lc.At("gating_mode").IsArray())
				for (size_t i = 0; i < lc.At("gating_mode").Size(); i++)
				{
					const Json& g = lc.At("gating_mode").At(i);
					if (!g.IsNull() && !(g.IsString() && g.AsString() == "none")) fail("gating_mode");
				}
			if (lc.Contains("layer1x1") && lc.At("layer1x1").IsObject())
			{
				const Json& l = lc.At("layer1x1");
				if (l.Contains("active") && !Truthy(l.At("active"))) fail("layer1x1 inactive");
				if (l.Contains("groups") && l.At("groups").AsInt() != 1) fail("layer1x1 groups != 1");
			}
			for (const char* key : { "head1x1", "conv_pre_film", "conv_post_film", "input_mixin_pre_film", "input_mixin_post_film", "activation_pre_film",
					 "activation_post_film", "layer1x1_post_film", "head1x1_post_film" })
				if (BlockActive(lc, key)) fail(key);
			if (lc.Contains("groups_input") && lc.At("groups_input").AsInt() != 1) fail("groups_input != 1");
			if (lc.Contains("groups_input_mixin") && lc.At("groups_input_mixin").AsInt() != 1) fail("groups_input_mixin != 1");
			if (lc.Contains("slimmable") && !lc.At("slimmable").IsNull()) fail("slimmable layers");
			if (lc.Contains("activation") && lc.At("activation").IsArray())
				for (size_t i = 0; i < lc.At("activation").Size(); i++)
				{
					const Json& a = lc.At("activation").At(i);
					if (!a.IsObject()) continue;
					if (a.Contains("type") && a.At("type").AsString() != lc.At("activation").At(0).At("type").AsString()) fail("per-layer activation types differ");
					// the kernels hard-wire Activation.h:110-118 (slope 0.01)
					if (a.Contains("negative_slope") && std::fabs(a.At("negative_slope").AsFloat() - 0.01f) > 1e-5f) fail("LeakyReLU negative_slope != 0.01");
				}
		}

		// One "layers" entry of a WaveNet config -> WnArrayCfg.
		// A1 keys (InternalModel.h:204-209): input_size, condition_size, head_size, channels, kernel_size, head_bias, dilations
		// A2 keys: kernel_sizes[], head{out_channels,kernel_size,bias}, activation[] ...
		WnArrayCfg ReadLayerArray(const Json& lc, int headDilationOverride)
		{
			WnArrayCfg cfg;
			cfg.inputSize = lc.At("input_size").AsInt();
			cfg.conditionSize = lc.At("condition_size").AsInt();
			cfg.channels = lc.At("channels").AsInt();
			cfg.dilations = IntArray(lc.At("dilations"));
			if (lc.Contains("gated") && Truthy(lc.At("gated")))
				throw std::runtime_error("gated WaveNet layers are not supported by the Internal path");
			if (lc.Contains("activation")) cfg.activation = ActivationFromJson(lc.At("activation"));
			if (lc.Contains("kernel_sizes"))
			{
				RejectUnsupportedA2Features(lc, cfg.channels);
				cfg.kernelSizes = IntArray(lc.At("kernel_sizes"));
				const Json& head = lc.At("head");
				cfg.headSize = head.At("out_channels").AsInt();
				cfg.headKernelSize = head.At("kernel_size").AsInt();
				cfg.hasHeadBias = Truthy(head.At("bias"));
				cfg.headDilation = headDilationOverride; // OversampleNAMConfig sets head["head_dilation"], :124-127
			}
			else
			{
				const int k = lc.At("kernel_size").AsInt();
				cfg.kernelSizes.assign(cfg.dilations.size(), k);
				cfg.headSize = lc.At("head_size").AsInt();
				cfg.headKernelSize = 1;
				cfg.hasHeadBias = Truthy(lc.At("head_bias"));
				cfg.headDilation = 1;
			}
			if (cfg.kernelSizes.size() != cfg.dilations.size()) throw std::runtime_error("kernel_sizes / dilations length mismatch");
			return cfg;
		}

		// NeuralModel.cpp:389-465: which configurations the reference runs on its static (templated) engines
		bool IsOfficialArchitecture(const std::vector<WnArrayCfg>& arrays)
		{
			if (arrays.size() == 1)
			{
				const WnArrayCfg& a = arrays[0];
				return a.dilations == kA2Dilations && a.kernelSizes == kA2KernelSizes && (a.channels == 3 || a.channels == 8) &&
					a.headKernelSize == 16 && a.headSize == 1 && a.hasHeadBias && a.activation == ACT_LEAKYRELU;
			}
			if (arrays.size() == 2)
			{
				const WnArrayCfg& a = arrays[0];
				const WnArrayCfg& b = arrays[1];
				if (a.hasHeadBias || !b.hasHeadBias) return false;
				if (a.headKernelSize != 1 || b.headKernelSize != 1) return false;
				bool dil = false;
				if (a.channels == 16) dil = (a.dilations == kStdDilations && b.dilations == kStdDilations);
				else dil = (a.dilations == kLiteDilations && b.dilations == kLiteDilations2);
				if (!dil) return false;
				// InternalA1WaveNetDefinitionT<16,8> / <12,6> / <8,4> / <4,2> (NeuralModel.cpp:25-28)
				const int c = a.channels, h = a.headSize;
				return (c == 16 && h == 8) || (c == 12 && h == 6) || (c == 8 && h == 4) || (c == 4 && h == 2);
			}
			return false;
		}

		std::shared_ptr<ModelDesc> ReadNAMWaveNet(const Json& modelJson, int oversampleFactor, const LoaderOptions& opts)
		{
			auto desc = std::make_shared<ModelDesc>();
			desc->kind = MODEL_WAVENET;
			const Json& config = modelJson.At("config");
			// model-level blocks of the A2 format that only the NAM Core back-end evaluates (NeuralModel.cpp:200-207)
			if (config.Contains("head") && !config.At("head").IsNull())
				throw std::runtime_error("WaveNet feature not supported by the Internal path (needs the NAM Core back-end): model-level head");
			if (config.Contains("condition_dsp"))
				throw std::runtime_error("WaveNet feature not supported by the Internal path (needs the NAM Core back-end): condition_dsp");
			if (config.Contains("in_channels") && config.At("in_channels").IsNumber() && config.At("in_channels").AsInt() != 1)
				throw std::runtime_error("WaveNet feature not supported by the Internal path (needs the NAM Core back-end): in_channels != 1");
			const Json& layers = config.At("layers");
			for (size_t i = 0; i < layers.Size(); i++)
			{
				WnArrayCfg cfg = ReadLayerArray(layers.At(i), oversampleFactor);
				// OversampleNAMConfig (NeuralModel.cpp:92-130): integer oversampling multiplies every dilation
				if (oversampleFactor != 1)
					for (auto& d : cfg.dilations) d *= oversampleFactor;
				desc->wavenet.arrays.push_back(cfg);
			}
			modelJson.At("weights").FlattenNumbers(desc->wavenet.weights);
			desc->wavenet.isStatic = (oversampleFactor == 1) && IsOfficialArchitecture(desc->wavenet.arrays);
			// an A2-format file runs on the static engine only if it is the standard architecture (NeuralModel.cpp:365-380)
			if (desc->wavenet.isStatic && layers.Size() == 1 && layers.At(0).Contains("kernel_sizes") && !NAMIsA2Standard(modelJson)) desc->wavenet.isStatic = false;
			desc->wavenet.mathMode = opts.wavenetMath;
			// the reference throws "Wrong number of weights" inside CreateFromJson (WaveNet.h:704-709); so does this loader, together with
			// the limits of the gfx950 kernels, instead of deferring them to the first device use on the audio thread
			ValidateWaveNetDesc(desc->wavenet);
			return desc;
		}

	}

	// ---- NAMIsA2 / NAMIsA2Standard, restated rule for rule (NeuralModel.cpp:159-168, 188-317) -----------------------------------
	// The reference uses them to decide which engine evaluates an A2-format file: the static Internal engine for the standard
	// architecture, NAM Core for everything else.  Here they decide IsStatic() for A2 files (the rejection of features without
	// arithmetic on this path is RejectUnsupportedA2Features above).  Note the reference's IsActive(): a MISSING block counts as active.
	// "A2 format" = file version newer than 0.5.4 (the rule of NeuralModel.cpp:159-168): the dotted version is read as up to three
	// numbers (a missing or malformed field counts as 0, like a failed stream extraction there) and compared as a tuple
	bool NAMIsA2(const std::string& version)
	{
		long field[3] = { 0, 0, 0 };
		const char* p = version.c_str();
		for (int i = 0; i < 3; i++)
		{
			char* end = nullptr;
			const long v = std::strtol(p, &end, 10);
			if (end == p) break; // no number here: this field and the rest stay 0
			field[i] = v;
			if (*end == 0) break;
			p = end + 1; // the reference skips exactly one separator character, whatever it is
		}
		// newer than 0.5.4 -- the reference tests major > 0 || minor > 5 || (minor == 5 && patch > 4), which a tuple comparison
		// reproduces for every version with major >= 0
		if (field[0] != 0) return field[0] > 0;
		if (field[1] != 5) return field[1] > 5;
		return field[2] > 4;
	}

	namespace
	{
		int ValueOr(const Json& j, const char* key, int dflt) { return (j.Contains(key) && j.At(key).IsNumber()) ? j.At(key).AsInt() : dflt; }
		bool RefIsActive(const Json& j, const char* name)
		{
			if (!j.Contains(name)) return true;
			const Json& b = j.At(name);
			return b.IsObject() && b.Contains("active") && b.At("active").IsBool() && b.At("active").AsBool();
		}
		bool HasNonNull(const Json& j, const char* name) { return j.Contains(name) && !j.At(name).IsNull(); }
		bool SequenceIs(const Json& j, const std::vector<int>& want)
		{
			if (!j.IsArray() || j.Size() != want.size()) return false;
			for (size_t i = 0; i < want.size(); i++)
				if (!j.At(i).IsNumber() || j.At(i).AsInt() != want[i]) return false;
			return true;
		}
	}

	bool NAMIsA2Standard(const Json& modelJson)
	{
		if (!modelJson.IsObject() || !modelJson.Contains("architecture")) return false;
		if (!modelJson.At("architecture").IsString() || modelJson.At("architecture").AsString() != "WaveNet") return false;
		if (!modelJson.Contains("config")) return false;
		const Json& config = modelJson.At("config");
		if (HasNonNull(config, "head")) return false;
		if (config.Contains("condition_dsp")) return false;
		if (ValueOr(config, "in_channels", 1) != 1) return false;
		if (!config.Contains("layers") || config.At("layers").Size() != 1) return false;
		const Json& lc = config.At("layers").At(0);
		if (ValueOr(lc, "input_size", 0) != 1) return false;
		if (ValueOr(lc, "condition_size", 0) != 1) return false;
		const int channels = ValueOr(lc, "channels", 0);
		if (channels != 3 && channels != 8) return false;
		if (ValueOr(lc, "bottleneck", channels) != channels) return false;
		if (!lc.Contains("kernel_sizes") || !SequenceIs(lc.At("kernel_sizes"), kA2KernelSizes)) return false;
		if (!lc.Contains("dilations") || !SequenceIs(lc.At("dilations"), kA2Dilations)) return false;
		if (!lc.Contains("activation") || !lc.At("activation").IsArray()) return false;
		for (size_t i = 0; i < lc.At("activation").Size(); i++)
		{
			const Json& a = lc.At("activation").At(i);
			if (!a.IsObject() || !a.Contains("type") || !a.At("type").IsString() || a.At("type").AsString() != "LeakyReLU") return false;
			const float slope = (a.Contains("negative_slope") && a.At("negative_slope").IsNumber()) ? a.At("negative_slope").AsFloat() : 0.01f;
			if (std::fabs(slope - 0.01f) > 1e-5f) return false;
		}
		if (lc.Contains("secondary_activation") && lc.At("secondary_activation").IsArray())
			for (size_t i = 0; i < lc.At("secondary_activation").Size(); i++)
				if (!lc.At("secondary_activation").At(i).IsNull()) return false;
		if (lc.Contains("gating_mode") && lc.At("gating_mode").IsArray())
			for (size_t i = 0; i < lc.At("gating_mode").Size(); i++)
			{
				const Json& g = lc.At("gating_mode").At(i);
				if (!g.IsNull() && !(g.IsString() && g.AsString() == "none")) return false;
			}
		if (!lc.Contains("head") || !lc.At("head").IsObject()) return false;
		const Json& head = lc.At("head");
		if (ValueOr(head, "out_channels", 1) != 1) return false;
		if (ValueOr(head, "kernel_size", 16) != 16) return false;
		if (ValueOr(head, "head_dilation", 1) != 1) return false;
		if (head.Contains("bias") && !Truthy(head.At("bias"))) return false;
		if (!RefIsActive(lc, "layer1x1")) return false;
		if (!lc.Contains("layer1x1")) return false; // the reference's .at("layer1x1") throws here; a file without the block is not standard
		if (ValueOr(lc.At("layer1x1"), "groups", 1) != 1) return false;
		for (const char* key : { "head1x1", "conv_pre_film", "conv_post_film", "input_mixin_pre_film", "input_mixin_post_film", "activation_pre_film",
				 "activation_post_film", "layer1x1_post_film", "head1x1_post_film" })
			if (RefIsActive(lc, key)) return false;
		if (ValueOr(lc, "groups_input", 1) != 1) return false;
		if (ValueOr(lc, "groups_input_mixin", 1) != 1) return false;
		if (HasNonNull(lc, "slimmable")) return false;
		return true;
	}

	void ValidateRecurrentDesc(const LSTMDesc& d)
	{
		int tailWidth = 0, tailHistory = 0;
		if (!d.tail.empty())
		{
			if ((int)d.tail.size() > LSTM_MAX_TAIL) throw std::runtime_error("keras model with more than " + std::to_string(LSTM_MAX_TAIL) + " dense layers is not supported");
			int in = d.numLayers > 0 ? d.hiddenSize : 1;
			for (const DenseLayerDesc& t : d.tail) tailHistory = std::max(tailHistory, t.History());
			for (const DenseLayerDesc& t : d.tail)
			{
				if (t.in != in || t.out < 1 || t.ksize < 1 || t.dilation < 1 || (int)t.w.size() != t.RowLen() * t.out || (int)t.b.size() != t.out)
					throw std::runtime_error("keras dense / conv1d layer has unexpected weight shapes");
				if (t.out > LSTM_MAX_TAIL_WIDTH) throw std::runtime_error("keras dense / conv1d layer wider than " + std::to_string(LSTM_MAX_TAIL_WIDTH) + " units is not supported");
				if (t.History() > LSTM_MAX_TAIL_HISTORY)
					throw std::runtime_error("keras conv1d layer with more than " + std::to_string(LSTM_MAX_TAIL_HISTORY) + " samples of history ((kernel_size - 1) x dilation) is not supported");
				tailWidth = std::max(tailWidth, tailHistory > 0 ? std::max(t.in, t.out) : t.out);
				in = t.out;
			}
		}
		if (d.cell == CELL_GRU)
		{
			if (!GruShapeSupported(d.hiddenSize, d.numLayers, tailWidth, tailHistory))
				throw std::runtime_error("GRU " + std::to_string(d.numLayers) + "x" + std::to_string(d.hiddenSize) +
					" is not supported (1-8 layers of up to 1024 units; a dense tail behind more than 128 units must fit the 160 KB LDS with its [samples][units] buffer; a tail with conv1d layers with its two [widest layer][history + 128] buffers)");
		}
		else if (!LstmShapeSupported(d.hiddenSize, d.numLayers, tailWidth, tailHistory))
			throw std::runtime_error("LSTM " + std::to_string(d.numLayers) + "x" + std::to_string(d.hiddenSize) +
				" is not supported (1-8 layers of up to 1024 units; a dense tail behind more than 128 units must fit the 160 KB LDS with its [samples][units] buffer; a tail with conv1d layers with its two [widest layer][history + 128] buffers)");
	}

	namespace
	{
		// LSTM.h:42-56,130-147
		std::shared_ptr<ModelDesc> ReadNAMLSTM(const Json& modelJson, const LoaderOptions& opts)
		{
			auto desc = std::make_shared<ModelDesc>();
			desc->kind = MODEL_LSTM;
			const Json& config = modelJson.At("config");
			LSTMDesc& lstm = desc->lstm;
			lstm.numLayers = config.At("num_layers").AsInt();
			lstm.hiddenSize = config.At("hidden_size").AsInt();
			if (config.Contains("input_size") && config.At("input_size").AsInt() != 1)
				throw std::runtime_error("LSTM input_size != 1 is not supported");
			std::vector<float> w;
			modelJson.At("weights").FlattenNumbers(w);
			const int H = lstm.hiddenSize;
			size_t expected = 0;
			for (int l = 0; l < lstm.numLayers; l++) expected += (size_t)4 * H * ((l == 0 ? 1 : H) + H) + 4 * H + 2 * H;
			expected += (size_t)H + 1;
			if (expected != w.size())
			{
				std::stringstream str;
				str << "Wrong number of weights. Expected " << expected << " but got " << w.size();
				throw std::runtime_error(str.str());
			}
			size_t it = 0;
			for (int l = 0; l < lstm.numLayers; l++)
			{
				LSTMLayerDesc layer;
				layer.inputSize = (l == 0) ? 1 : H;
				const size_t nW = (size_t)4 * H * (layer.inputSize + H);
				layer.w.assign(w.begin() + it, w.begin() + it + nW); it += nW;
				layer.bias.assign(w.begin() + it, w.begin() + it + 4 * H); it += (size_t)4 * H;
				layer.h0.assign(w.begin() + it, w.begin() + it + H); it += (size_t)H;
				layer.c0.assign(w.begin() + it, w.begin() + it + H); it += (size_t)H;
				lstm.layers.push_back(std::move(layer));
			}
			lstm.headWeights.assign(w.begin() + it, w.begin() + it + H); it += (size_t)H;
			lstm.headBias = w[it++];
			// InternalLSTMDefinitionT list, NeuralModel.cpp:31-38 (only when BUILD_INTERNAL_STATIC_LSTM)
			lstm.isStatic = false;
			lstm.mathMode = opts.lstmMath;
			ValidateRecurrentDesc(lstm);
			return desc;
		}

		// InternalModel.h:311-356 / :473-519 and LSTM.h:58-85
		std::shared_ptr<ModelDesc> ReadKerasLSTM(const Json& modelJson, const LoaderOptions& opts)
		{
			const Json& layers = modelJson.At("layers");
			const size_t numLayers = layers.Size();
			if (numLayers < 2) return nullptr;
			const Json& lastLayer = layers.At(numLayers - 1);
			if (lastLayer.At("type").AsString() != "dense") return nullptr;

			auto desc = std::make_shared<ModelDesc>();
			desc->kind = MODEL_LSTM;
			LSTMDesc& lstm = desc->lstm;
			lstm.numLayers = (int)numLayers - 1;
			lstm.hiddenSize = layers.At(0).At("shape").Back().AsInt();
			const int H = lstm.hiddenSize;

			lastLayer.At("weights").At(0).FlattenNumbers(lstm.headWeights);
			lstm.headBias = lastLayer.At("weights").At(1).At(0).AsFloat();
			if ((int)lstm.headWeights.size() < H) throw std::runtime_error("keras dense head has too few weights");

			for (int l = 0; l < lstm.numLayers; l++)
			{
				const Json& layer = layers.At((size_t)l);
				if (layer.At("type").AsString() != "lstm") return nullptr;
				std::vector<float> kernel, recurrent, bias;
				layer.At("weights").At(0).FlattenNumbers(kernel);    // [I][4H]
				layer.At("weights").At(1).FlattenNumbers(recurrent); // [H][4H]
				layer.At("weights").At(2).FlattenNumbers(bias);      // [4H]
				LSTMLayerDesc ld;
				ld.inputSize = (l == 0) ? 1 : H;
				const int I = ld.inputSize, W = I + H, R = 4 * H;
				if ((int)kernel.size() != I * R || (int)recurrent.size() != H * R || (int)bias.size() < R)
					throw std::runtime_error("keras lstm layer has unexpected weight shapes");
				ld.w.assign((size_t)R * W, 0.0f);
				for (int j = 0; j < I; j++)
					for (int i = 0; i < R; i++) ld.w[(size_t)i * W + j] = kernel[(size_t)j * R + i];
				for (int j = 0; j < H; j++)
					for (int i = 0; i < R; i++) ld.w[(size_t)i * W + I + j] = recurrent[(size_t)j * R + i];
				ld.bias.assign(bias.begin(), bias.begin() + R);
				ld.h0.assign((size_t)H, 0.0f);
				ld.c0.assign((size_t)H, 0.0f);
				lstm.layers.push_back(std::move(ld));
			}
			lstm.mathMode = opts.lstmMath;
			ValidateRecurrentDesc(lstm);
			return desc;
		}

		int KerasActivation(const std::string& act, const char* what)
		{
			if (act.empty() || act == "linear") return DENSE_LINEAR;
			if (act == "tanh") return DENSE_TANH;
			if (act == "relu") return DENSE_RELU;
			if (act == "sigmoid") return DENSE_SIGMOID;
			if (act == "elu") return DENSE_ELU;
			if (act == "softmax") return DENSE_SOFTMAX;
			throw std::runtime_error(std::string("keras ") + what + " activation '" + act + "' is not supported");
		}

		// One keras dense layer: weights [in][out], bias [out], optional activation (RTNeural json_parser: a Dense layer followed by an
		// activation layer).  Throws on an activation this library has no kernel for.
		DenseLayerDesc ReadKerasDense(const Json& layer, int in)
		{
			DenseLayerDesc d;
			d.in = in;
			d.out = layer.At("shape").Back().AsInt();
			std::vector<float> kernel;
			layer.At("weights").At(0).FlattenNumbers(kernel); // [in][out]
			layer.At("weights").At(1).FlattenNumbers(d.b);
			if (d.out < 1 || (int)kernel.size() != d.in * d.out || (int)d.b.size() != d.out) throw std::runtime_error("keras dense layer has unexpected weight shapes");
			d.w.assign(kernel.size(), 0.0f);
			for (int k = 0; k < d.in; k++)
				for (int o = 0; o < d.out; o++) d.w[(size_t)o * d.in + k] = kernel[(size_t)k * d.out + o];
			d.activation = KerasActivation(layer.Contains("activation") ? layer.At("activation").AsString() : std::string(), "dense");
			return d;
		}

		// One keras conv1d layer (RTNeural json_parser: Conv1D(in, out, kernel_size, dilation) + an activation layer): causal, stride 1, one
		// group; weights [kernel_size][in][out], bias [out]; "kernel_size" / "dilation" as numbers or one-element lists.
		DenseLayerDesc ReadKerasConv1D(const Json& layer, int in)
		{
			auto lastInt = [&](const char* key, int dflt) {
				if (!layer.Contains(key)) return dflt;
				const Json& v = layer.At(key);
				return v.IsNumber() ? v.AsInt() : v.Back().AsInt();
			};
			DenseLayerDesc d;
			d.in = in;
			d.out = layer.At("shape").Back().AsInt();
			d.ksize = lastInt("kernel_size", 1);
			d.dilation = lastInt("dilation", 1);
			if (lastInt("strides", 1) != 1 || lastInt("groups", 1) != 1) throw std::runtime_error("keras conv1d layer with a stride or groups is not supported");
			if (d.ksize < 1 || d.dilation < 1) throw std::runtime_error("keras conv1d layer has unexpected kernel_size / dilation");
			std::vector<float> kernel;
			layer.At("weights").At(0).FlattenNumbers(kernel); // [k][in][out]
			layer.At("weights").At(1).FlattenNumbers(d.b);
			if (d.out < 1 || (int)kernel.size() != d.ksize * d.in * d.out || (int)d.b.size() != d.out) throw std::runtime_error("keras conv1d layer has unexpected weight shapes");
			d.w.assign(kernel.size(), 0.0f);
			for (int k = 0; k < d.ksize; k++)
				for (int i = 0; i < d.in; i++)
					for (int o = 0; o < d.out; o++) d.w[((size_t)o * d.ksize + k) * d.in + i] = kernel[((size_t)k * d.in + i) * d.out + o];
			d.activation = KerasActivation(layer.Contains("activation") ? layer.At("activation").AsString() : std::string(), "conv1d");
			return d;
		}

		// The layer types behind the recurrent part that this library evaluates with its dense-chain kernels.  Besides "dense" (RTNeural
		// json_parser: a Dense layer + an activation layer) the element-wise layers of RTNeural's parser are LOWERED at load time to dense
		// layers, so that no kernel has to know them:
		//   "activation"  joins the dense layer in front of it when that one is linear, else an identity layer carries it
		//   "batchnorm"   y = gamma (x - mean) / sqrt(var + epsilon) + beta (weights [gamma, beta, mean, var], or [mean, var] without the
		//                 affine part; epsilon 1e-3 when the file has none): folded into a linear dense layer in front of it, else a
		//                 diagonal layer
		//   "prelu"       y = max(x, 0) + alpha min(x, 0) (weights [alpha], per unit or one value) = relu(x) - alpha relu(-x): the rows
		//                 [W; -W] of a linear dense layer in front of it with relu, then the layer [I | -diag(alpha)] -- twice the
		//                 width in between (<= 64)
		//   "conv1d"      a causal convolution over time (round 5): a dense layer over `kernel_size` delayed copies of its input; the input
		//                 history is stream state (recurrent_tail.h ConvTail).  softmax runs across the units of its layer.
		// Third-party arithmetic (RTNeural is absent): parity unpinned, checked by the test suite against a float64 restatement.
		bool IsKerasTailType(const std::string& t) { return t == "dense" || t == "time-distributed-dense" || t == "conv1d" || t == "activation" || t == "batchnorm" || t == "prelu"; }

		DenseLayerDesc DiagonalDense(const std::vector<float>& scale, const std::vector<float>& shift)
		{
			DenseLayerDesc d;
			d.in = d.out = (int)scale.size();
			d.w.assign((size_t)d.in * d.out, 0.0f);
			for (int i = 0; i < d.in; i++) d.w[(size_t)i * d.in + i] = scale[(size_t)i];
			d.b = shift;
			d.activation = DENSE_LINEAR;
			return d;
		}

		void AppendKerasTailLayer(const Json& layer, int in, std::vector<DenseLayerDesc>& tail)
		{
			const std::string type = layer.At("type").AsString();
			if (type == "dense" || type == "time-distributed-dense")
			{
				tail.push_back(ReadKerasDense(layer, in));
				return;
			}
			if (type == "conv1d")
			{
				tail.push_back(ReadKerasConv1D(layer, in));
				return;
			}
			const bool foldable = !tail.empty() && tail.back().activation == DENSE_LINEAR; // a linear dense layer right in front
			if (type == "activation")
			{
				const int act = KerasActivation(layer.Contains("activation") ? layer.At("activation").AsString() : std::string(), "activation layer");
				if (foldable) tail.back().activation = act;
				else
				{
					tail.push_back(DiagonalDense(std::vector<float>((size_t)in, 1.0f), std::vector<float>((size_t)in, 0.0f)));
					tail.back().activation = act;
				}
				return;
			}
			if (type == "batchnorm")
			{
				const Json& w = layer.At("weights");
				std::vector<float> gamma((size_t)in, 1.0f), beta((size_t)in, 0.0f), mean, var;
				if (w.Size() >= 4)
				{
					gamma.clear(); beta.clear();
					w.At(0).FlattenNumbers(gamma); w.At(1).FlattenNumbers(beta); w.At(2).FlattenNumbers(mean); w.At(3).FlattenNumbers(var);
				}
				else if (w.Size() == 2) { w.At(0).FlattenNumbers(mean); w.At(1).FlattenNumbers(var); }
				else throw std::runtime_error("keras batchnorm layer has unexpected weight shapes");
				if ((int)gamma.size() != in || (int)beta.size() != in || (int)mean.size() != in || (int)var.size() != in)
					throw std::runtime_error("keras batchnorm layer has unexpected weight shapes");
				const double eps = (layer.Contains("epsilon") && layer.At("epsilon").IsNumber()) ? layer.At("epsilon").AsDouble() : 1e-3;
				std::vector<float> scale((size_t)in), shift((size_t)in);
				for (int i = 0; i < in; i++)
				{
					const double s = (double)gamma[(size_t)i] / std::sqrt((double)var[(size_t)i] + eps);
					scale[(size_t)i] = (float)s;
					shift[(size_t)i] = (float)((double)beta[(size_t)i] - (double)mean[(size_t)i] * s);
				}
				if (foldable)
				{
					DenseLayerDesc& p = tail.back();
					for (int o = 0; o < p.out; o++)
					{
						for (int k = 0; k < p.RowLen(); k++) p.w[(size_t)o * p.RowLen() + k] *= scale[(size_t)o];
						p.b[(size_t)o] = p.b[(size_t)o] * scale[(size_t)o] + shift[(size_t)o];
					}
				}
				else tail.push_back(DiagonalDense(scale, shift));
				return;
			}
			if (type == "prelu")
			{
				std::vector<float> alpha;
				layer.At("weights").At(0).FlattenNumbers(alpha);
				if (alpha.size() == 1) alpha.assign((size_t)in, alpha[0]);
				if ((int)alpha.size() != in) throw std::runtime_error("keras prelu layer has unexpected weight shapes");
				if (2 * in > LSTM_MAX_TAIL_WIDTH) throw std::runtime_error("keras prelu layer wider than " + std::to_string(LSTM_MAX_TAIL_WIDTH / 2) + " units is not supported");
				DenseLayerDesc a; // [x; -x] (of the linear dense layer in front when there is one), relu
				if (foldable)
				{
					const DenseLayerDesc p = tail.back();
					tail.pop_back();
					a.in = p.in; a.out = 2 * in;
					a.ksize = p.ksize; a.dilation = p.dilation; // (a linear conv1d layer in front folds the same way: its rows are ksize x in long)
					const int RL = p.RowLen();
					a.w.assign((size_t)a.out * RL, 0.0f);
					a.b.assign((size_t)a.out, 0.0f);
					for (int o = 0; o < in; o++)
					{
						for (int k = 0; k < RL; k++)
						{
							a.w[(size_t)o * RL + k] = p.w[(size_t)o * RL + k];
							a.w[(size_t)(in + o) * RL + k] = -p.w[(size_t)o * RL + k];
						}
						a.b[(size_t)o] = p.b[(size_t)o];
						a.b[(size_t)(in + o)] = -p.b[(size_t)o];
					}
				}
				else
				{
					a.in = in; a.out = 2 * in;
					a.w.assign((size_t)a.out * a.in, 0.0f);
					a.b.assign((size_t)a.out, 0.0f);
					for (int o = 0; o < in; o++)
					{
						a.w[(size_t)o * in + o] = 1.0f;
						a.w[(size_t)(in + o) * in + o] = -1.0f;
					}
				}
				a.activation = DENSE_RELU;
				DenseLayerDesc b; // relu(x) - alpha relu(-x)
				b.in = 2 * in; b.out = in;
				b.w.assign((size_t)b.out * b.in, 0.0f);
				b.b.assign((size_t)b.out, 0.0f);
				for (int o = 0; o < in; o++)
				{
					b.w[(size_t)o * b.in + o] = 1.0f;
					b.w[(size_t)o * b.in + in + o] = -alpha[(size_t)o];
				}
				b.activation = DENSE_LINEAR;
				tail.push_back(std::move(a));
				tail.push_back(std::move(b));
				return;
			}
			throw std::runtime_error("keras layer type '" + type + "' is not supported");
		}

		// Generic keras stacks -- what the reference hands to RTNeural's json_parser (NeuralModel.cpp:565-572, RTNeuralModel.h:300):
		// zero or more recurrent layers of ONE kind (lstm | gru, equal sizes) followed by one or more dense layers with activations;
		// the output is unit 0 of the last layer (RTNeuralModelDyn::Process, RTNeuralModel.h:417-421).  Arithmetic as RTNeural's with the
		// reference's FastMathsProvider (RTNeuralModel.h:10-31): accurate tanh, sigmoid = (tanh(x/2)+1)/2 -- so an LSTM in such a
		// stack runs with the StdMath policy.  Parity unpinned (RTNeural is an absent submodule): tests compare against a numpy
		// restatement of the Keras definitions.  activation / batchnorm / prelu layers are lowered to dense layers (AppendKerasTailLayer);
		// conv1d layers make the tail a per-block evaluation (ConvTail); softmax runs across the units of its layer in both tail forms.
		std::shared_ptr<ModelDesc> ReadKerasStack(const Json& modelJson)
		{
			const Json& layers = modelJson.At("layers");
			const size_t total = layers.Size();
			size_t numRec = 0;
			std::string cell;
			while (numRec < total)
			{
				const std::string t = layers.At(numRec).At("type").AsString();
				if (t != "lstm" && t != "gru") break;
				if (cell.empty()) cell = t;
				else if (cell != t) return nullptr;
				numRec++;
			}
			if (numRec == total) return nullptr; // no dense layer at the end
			for (size_t i = numRec; i < total; i++)
				if (!IsKerasTailType(layers.At(i).At("type").AsString())) return nullptr;

			std::shared_ptr<ModelDesc> desc;
			if (numRec > 0)
			{
				// the recurrent part through the classic readers (they expect [recurrent..., dense]): parse a copy of the layer list cut
				// after the first dense layer, then replace its head by the real tail
				desc = std::make_shared<ModelDesc>();
				desc->kind = MODEL_LSTM;
				LSTMDesc& d = desc->lstm;
				d.cell = (cell == "gru") ? CELL_GRU : CELL_LSTM;
				d.numLayers = (int)numRec;
				d.hiddenSize = layers.At(0).At("shape").Back().AsInt();
				const int H = d.hiddenSize, gates = (d.cell == CELL_GRU) ? 3 : 4;
				for (size_t l = 0; l < numRec; l++)
				{
					const Json& layer = layers.At(l);
					if (layer.At("shape").Back().AsInt() != H) return nullptr;
					std::vector<float> kernel, recurrent, bias;
					layer.At("weights").At(0).FlattenNumbers(kernel);    // [I][gates H]
					layer.At("weights").At(1).FlattenNumbers(recurrent); // [H][gates H]
					layer.At("weights").At(2).FlattenNumbers(bias);      // lstm [4H]; gru [2][3H]
					LSTMLayerDesc ld;
					ld.inputSize = (l == 0) ? 1 : H;
					const int I = ld.inputSize, W = I + H, R = gates * H;
					if ((int)kernel.size() != I * R || (int)recurrent.size() != H * R || (int)bias.size() != (d.cell == CELL_GRU ? 2 * R : R))
						throw std::runtime_error("keras " + cell + " layer has unexpected weight shapes");
					ld.w.assign((size_t)R * W, 0.0f);
					for (int j = 0; j < I; j++)
						for (int i = 0; i < R; i++) ld.w[(size_t)i * W + j] = kernel[(size_t)j * R + i];
					for (int j = 0; j < H; j++)
						for (int i = 0; i < R; i++) ld.w[(size_t)i * W + I + j] = recurrent[(size_t)j * R + i];
					ld.bias = bias;
					ld.h0.assign((size_t)H, 0.0f);
					ld.c0.assign((size_t)H, 0.0f);
					d.layers.push_back(std::move(ld));
				}
				d.mathMode = MATH_STD; // RTNeural with the reference's FastMathsProvider: accurate tanh
			}
			else
			{
				desc = std::make_shared<ModelDesc>();
				desc->kind = MODEL_LSTM;
				desc->lstm.cell = CELL_LSTM;
				desc->lstm.numLayers = 0;
				desc->lstm.hiddenSize = 1;
				desc->lstm.mathMode = MATH_STD;
			}
			LSTMDesc& d = desc->lstm;
			int in = numRec > 0 ? d.hiddenSize : 1;
			for (size_t i = numRec; i < total; i++)
			{
				AppendKerasTailLayer(layers.At(i), in, d.tail);
				in = d.tail.back().out;
			}
			d.headWeights.assign((size_t)std::max(d.hiddenSize, 1), 0.0f); // unused with a tail
			ValidateRecurrentDesc(d);
			return desc;
		}

		// keras "gru" stacks + dense head.  The reference runs these on RTNeural (NeuralModel.cpp:565-572, RTNeuralModel.h:300);
		// weights per layer: kernel [I][3H], recurrent [H][3H], bias [2][3H], gate column blocks z | r | c (Keras reset_after form).
		std::shared_ptr<ModelDesc> ReadKerasGRU(const Json& modelJson)
		{
			const Json& layers = modelJson.At("layers");
			const size_t numLayers = layers.Size();
			if (numLayers < 2) return nullptr;
			const Json& lastLayer = layers.At(numLayers - 1);
			if (lastLayer.At("type").AsString() != "dense") return nullptr;

			auto desc = std::make_shared<ModelDesc>();
			desc->kind = MODEL_LSTM;
			LSTMDesc& gru = desc->lstm;
			gru.cell = CELL_GRU;
			gru.numLayers = (int)numLayers - 1;
			gru.hiddenSize = layers.At(0).At("shape").Back().AsInt();
			const int H = gru.hiddenSize;

			lastLayer.At("weights").At(0).FlattenNumbers(gru.headWeights);
			lastLayer.At("weights").At(1).FlattenNumbers(gru.headBiasVec);
			if ((int)gru.headWeights.size() != H || gru.headBiasVec.size() != 1) return nullptr; // a wider dense head is a generic RTNeural model
			gru.headBias = gru.headBiasVec[0];

			for (int l = 0; l < gru.numLayers; l++)
			{
				const Json& layer = layers.At((size_t)l);
				if (layer.At("type").AsString() != "gru") return nullptr;
				if (layer.At("shape").Back().AsInt() != H) return nullptr;
				std::vector<float> kernel, recurrent, bias;
				layer.At("weights").At(0).FlattenNumbers(kernel);    // [I][3H]
				layer.At("weights").At(1).FlattenNumbers(recurrent); // [H][3H]
				layer.At("weights").At(2).FlattenNumbers(bias);      // [2][3H]
				LSTMLayerDesc ld;
				ld.inputSize = (l == 0) ? 1 : H;
				const int I = ld.inputSize, W = I + H, R = 3 * H;
				if ((int)kernel.size() != I * R || (int)recurrent.size() != H * R || (int)bias.size() != 2 * R)
					throw std::runtime_error("keras gru layer has unexpected weight shapes");
				ld.w.assign((size_t)R * W, 0.0f);
				for (int j = 0; j < I; j++)
					for (int i = 0; i < R; i++) ld.w[(size_t)i * W + j] = kernel[(size_t)j * R + i];
				for (int j = 0; j < H; j++)
					for (int i = 0; i < R; i++) ld.w[(size_t)i * W + I + j] = recurrent[(size_t)j * R + i];
				ld.bias = bias;
				ld.h0.assign((size_t)H, 0.0f);
				ld.c0.assign((size_t)H, 0.0f);
				gru.layers.push_back(std::move(ld));
			}
			ValidateRecurrentDesc(gru);
			return desc;
		}

		int OversampleFactor(const Json& modelJson, int externalSampleRate)
		{
			// NeuralModel.cpp:92-114
			if (modelJson.At("architecture").AsString() != "WaveNet") return 1;
			int modelSampleRate = 48000;
			if (modelJson.Contains("sample_rate") && modelJson.At("sample_rate").IsNumber())
				modelSampleRate = (int)modelJson.At("sample_rate").AsFloat();
			if (modelSampleRate == externalSampleRate || modelSampleRate <= 0) return 1;
			if ((externalSampleRate % modelSampleRate) != 0) return 1;
			return externalSampleRate / modelSampleRate;
		}
	}

	std::shared_ptr<LoadedModel> LoadModelFromJson(const Json& modelJson, const std::string& extension, const LoaderOptions& opts)
	{
		auto model = std::make_shared<LoadedModel>();

		if (extension == ".nam")
		{
			const std::string arch = modelJson.At("architecture").AsString();
			ReadNAMConfig(modelJson, model->info);

			if (arch == "SlimmableContainer")
			{
				// ScalableCompositeModel::CreateModelFromNAMJson, CompositeModel.h:137-159
				model->isComposite = true;
				const Json& subModels = modelJson.At("config").At("submodels");
				for (size_t i = 0; i < subModels.Size(); i++)
				{
					const Json& sub = subModels.At(i);
					auto loaded = LoadModelFromJson(sub.At("model"), ".nam", opts);
					if (!loaded || loaded->isComposite || loaded->subModels.size() != 1)
						throw std::runtime_error("SlimmableContainer submodel could not be loaded");
					SubModel sm = loaded->subModels[0];
					sm.maxValue = sub.At("max_value").AsFloat();
					sm.info = loaded->info;
					model->subModels.push_back(sm);
					model->qualityLevels.push_back({ sm.maxValue, (int)model->subModels.size() - 1 });
					std::stable_sort(model->qualityLevels.begin(), model->qualityLevels.end(),
						[](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
				}
				if (model->subModels.empty()) throw std::runtime_error("SlimmableContainer without submodels");
				return model;
			}

			SubModel sm;
			sm.info = model->info;
			if (arch == "WaveNet") sm.desc = ReadNAMWaveNet(modelJson, OversampleFactor(modelJson, opts.externalSampleRate), opts);
			else if (arch == "LSTM") sm.desc = ReadNAMLSTM(modelJson, opts);
			else return nullptr;
			model->subModels.push_back(sm);
			model->qualityLevels.push_back({ 1.0f, 0 });
			return model;
		}
		else if (extension == ".json" || extension == ".aidax")
		{
			ReadKerasConfig(modelJson, model->info);
			const Json& layers = modelJson.At("layers");
			const std::string modelType = layers.At(0).At("type").AsString();
			// "lstm": Internal path (NeuralModel.cpp:526-563); "gru" and every other stack of lstm | gru and dense layers: RTNeural in
			// the reference (:565-572), restated here (parity unpinned); layer types without a kernel (conv1d, batchnorm ...): no model
			SubModel sm;
			sm.info = model->info;
			// the classic shapes first ([lstm..., dense(1)] / [gru..., dense(1)], no activation: the shaped kernels), then the generic stack
			bool classic = (modelType == "lstm" || modelType == "gru");
			if (classic)
			{
				const size_t nl = layers.Size();
				for (size_t i = 0; i + 1 < nl && classic; i++) classic = layers.At(i).At("type").AsString() == modelType;
				const Json& last = layers.At(nl - 1);
				classic = classic && nl >= 2 && last.At("type").AsString() == "dense" && last.At("shape").Back().AsInt() == 1 &&
					(!last.Contains("activation") || last.At("activation").AsString().empty() || last.At("activation").AsString() == "linear");
			}
			if (classic) sm.desc = (modelType == "gru") ? ReadKerasGRU(modelJson) : ReadKerasLSTM(modelJson, opts);
			if (!sm.desc) sm.desc = ReadKerasStack(modelJson);
			if (!sm.desc) return nullptr;
			model->subModels.push_back(sm);
			model->qualityLevels.push_back({ 1.0f, 0 });
			return model;
		}

		return nullptr;
	}

	std::shared_ptr<LoadedModel> LoadModelFromText(const std::string& text, const std::string& extension, const LoaderOptions& opts)
	{
		const Json j = Json::Parse(text);
		return LoadModelFromJson(j, extension, opts);
	}

	std::shared_ptr<LoadedModel> LoadModelFromFile(const std::string& path, const LoaderOptions& opts)
	{
		std::ifstream f(path, std::ifstream::binary);
		if (!f.good()) return nullptr;
		std::stringstream ss;
		ss << f.rdbuf();
		std::string ext;
		const size_t dot = path.find_last_of('.');
		const size_t slash = path.find_last_of("/\\");
		if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ext = path.substr(dot);
		return LoadModelFromText(ss.str(), ext, opts);
	}
}
