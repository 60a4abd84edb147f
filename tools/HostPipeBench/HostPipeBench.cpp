// HostPipeBench: PCIe-inclusive throughput of the batched engine from a plain C++ host, through the C ABI only
// (include/neuralaudio_amd.h): host buffers in, host buffers out, pipelined with NA_BatchSubmit / NA_BatchCollect.
//   HostPipeBench <model file> [streams=1024] [frames=128] [buffers=2000]
//   HostPipeBench <model file> [streams] [frames] [buffers] --gpus N [--devices 0,0,...] [--fan-in rccl] [--loopback]
//       the multi-GPU host (NA_Multi*: one batch + one host thread + one HIP stream per device, the global stream list sharded by
//       cost): `streams` is the GLOBAL count; --devices names the device of every shard explicitly (an index may repeat, e.g. 0,0 runs
//       two shards on one GPU -- the plumbing test on a single-GPU box).  A second model file may follow --mix: the global list is then
//       half / half (architecture-sorted), which exercises the cost-balanced cut.
// Prints one JSON object: microseconds per buffer for the copying entry points (caller-owned buffers) and for the zero-copy ones
// (NA_BatchNextInput / NA_BatchOutputView: the host produces into / consumes from the pinned staging buffers), two buffers in
// flight, plus the blocking NA_BatchProcess latency.  bench.py reports these as "pcie_inclusive" (never as `value`).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "neuralaudio_amd.h"

static double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "HostPipeBench: %s failed: %s\n", #cond, NA_GetLastError()); return 1; } } while (0)

static int RunMulti(NeuralModelLoader* loader, NeuralModel* model, const char* mixFile, const std::vector<int>& devices, int streams, int frames, int buffers, bool rcclFanIn)
{
	NA_MultiBatch* multi = NA_MultiCreate(devices.data(), (int)devices.size());
	CHECK(multi != nullptr);
	// --fan-in rccl: weights replicated and output rows gathered over RCCL (one rank per shard: distinct devices); blocking calls only
	if (rcclFanIn) CHECK(NA_MultiSetFanIn(multi, 1) == 0);
	NeuralModel* second = nullptr;
	if (mixFile)
	{
		second = NA_CreateModelFromFileUtf8(loader, mixFile, 0);
		CHECK(second != nullptr);
		CHECK(NA_MultiAddStreams(multi, model, 1.0f, streams / 2, 1) == 0);
		CHECK(NA_MultiAddStreams(multi, second, 1.0f, streams - streams / 2, 1) == streams / 2);
	}
	else CHECK(NA_MultiAddStreams(multi, model, 1.0f, streams, 1) == 0);
	CHECK(NA_MultiCommit(multi) == 0);
	const size_t count = (size_t)streams * frames;
	std::vector<float> in(count), out(count), ref(count);
	for (size_t i = 0; i < count; i++) in[i] = 0.25f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.125f;
	// one single-device batch with the same global list: the sharded host must reproduce it bit for bit
	{
		NA_Batch* one = NA_BatchCreate(devices[0], nullptr);
		CHECK(one != nullptr);
		if (second)
		{
			CHECK(NA_BatchAddStreams(one, model, 1.0f, streams / 2, 1) == 0);
			CHECK(NA_BatchAddStreams(one, second, 1.0f, streams - streams / 2, 1) == streams / 2);
		}
		else CHECK(NA_BatchAddStreams(one, model, 1.0f, streams, 1) == 0);
		CHECK(NA_BatchProcess(one, in.data(), ref.data(), (size_t)frames) == 0);
		NA_BatchDestroy(one);
	}
	CHECK(NA_MultiProcess(multi, in.data(), out.data(), (size_t)frames) == 0);
	const bool identical = std::memcmp(out.data(), ref.data(), count * sizeof(float)) == 0;
	std::vector<double> lat;
	for (int i = 0; i < 200; i++)
	{
		const double t0 = Now();
		CHECK(NA_MultiProcess(multi, in.data(), out.data(), (size_t)frames) == 0);
		if (i >= 30) lat.push_back((Now() - t0) * 1e6);
	}
	std::sort(lat.begin(), lat.end());
	double usPipe = 0.0;
	if (!rcclFanIn)
	{
		int pending = NA_MultiSubmit(multi, in.data(), (size_t)frames);
		CHECK(pending >= 0);
		const double t0 = Now();
		for (int i = 0; i < buffers; i++)
		{
			const int next = NA_MultiSubmit(multi, in.data(), (size_t)frames);
			CHECK(next >= 0);
			CHECK(NA_MultiCollect(multi, pending, out.data()) == 0);
			pending = next;
		}
		CHECK(NA_MultiCollect(multi, pending, out.data()) == 0);
		usPipe = (Now() - t0) * 1e6 / buffers;
	}
	else
	{
		const double t0 = Now();
		for (int i = 0; i < buffers; i++) CHECK(NA_MultiProcess(multi, in.data(), out.data(), (size_t)frames) == 0);
		usPipe = (Now() - t0) * 1e6 / buffers; // (blocking calls back to back: the gathered path has no pipelined form)
	}
	std::printf("{\"multi_gpu_host\": true, \"shards\": [");
	for (int s = 0; s < NA_MultiNumShards(multi); s++)
	{
		int b = 0, e = 0, d = 0;
		CHECK(NA_MultiShardRange(multi, s, &b, &e, &d) == 0);
		std::printf("%s{\"device\": %d, \"begin\": %d, \"end\": %d}", s ? ", " : "", d, b, e);
	}
	std::printf("], \"fan_in\": \"%s\", \"streams\": %d, \"frames\": %d, \"buffers\": %d, \"matches_single_batch\": %s, \"us_per_buffer_pipelined\": %.3f, \"Msamples_per_s\": %.1f, "
		"\"blocking_latency_us\": {\"p50\": %.1f, \"p99\": %.1f}}\n",
		rcclFanIn ? "rccl" : "host rows", streams, frames, buffers, identical ? "true" : "false", usPipe, (double)count / usPipe, lat[lat.size() / 2], lat[(size_t)(lat.size() * 0.99)]);
	NA_MultiDestroy(multi);
	if (second) DeleteModel(second);
	return identical ? 0 : 3;
}

int main(int argc, char** argv)
{
	if (argc < 2) { std::fprintf(stderr, "usage: HostPipeBench <model> [streams] [frames] [buffers] [--gpus N] [--devices a,b,...] [--mix <model 2>] [--fan-in rccl] [--loopback]\n"); return 2; }
	std::vector<const char*> pos;
	std::vector<int> devices;
	int gpus = 0;
	const char* mixFile = nullptr;
	bool rcclFanIn = false;
	for (int i = 1; i < argc; i++)
	{
		if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = std::atoi(argv[++i]);
		else if (!std::strcmp(argv[i], "--fan-in") && i + 1 < argc) rcclFanIn = !std::strcmp(argv[++i], "rccl");
		else if (!std::strcmp(argv[i], "--mix") && i + 1 < argc) mixFile = argv[++i];
		else if (!std::strcmp(argv[i], "--loopback"))
		{
			// rehearsal on a one-GPU box: the multi-GPU host bound to the library's loopback RCCL table (test build only), so that
			// `--devices 0,0 --fan-in rccl` runs the communicator set-up, the weight fan-out and the gathered fan-in with two ranks
			NA_DebugSetRcclApi(1, 0, 0);
		}
		else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc)
		{
			for (const char* p = argv[++i]; *p;)
			{
				devices.push_back(std::atoi(p));
				while (*p && *p != ',') p++;
				if (*p == ',') p++;
			}
		}
		else pos.push_back(argv[i]);
	}
	const int streams = pos.size() > 1 ? std::atoi(pos[1]) : 1024, frames = pos.size() > 2 ? std::atoi(pos[2]) : 128, buffers = pos.size() > 3 ? std::atoi(pos[3]) : 2000;
	NeuralModelLoader* loader = CreateLoader();
	CHECK(loader != nullptr);
	NeuralModel* model = NA_CreateModelFromFileUtf8(loader, pos[0], 0);
	CHECK(model != nullptr);
	if (gpus > 0 || !devices.empty())
	{
		if (devices.empty())
			for (int d = 0; d < gpus; d++) devices.push_back(d);
		const int rc = RunMulti(loader, model, mixFile, devices, streams, frames, buffers, rcclFanIn);
		DeleteModel(model);
		DeleteLoader(loader);
		return rc;
	}
	NA_Batch* batch = NA_BatchCreate(0, nullptr);
	CHECK(batch != nullptr);
	CHECK(NA_BatchAddStreams(batch, model, 1.0f, streams, 1) >= 0);
	const size_t count = (size_t)streams * frames;
	std::vector<float> in(count), out(count);
	for (size_t i = 0; i < count; i++) in[i] = 0.25f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.125f;

	// blocking call, one buffer at a time
	std::vector<double> lat;
	for (int i = 0; i < 300; i++)
	{
		const double t0 = Now();
		CHECK(NA_BatchProcess(batch, in.data(), out.data(), (size_t)frames) == 0);
		if (i >= 50) lat.push_back((Now() - t0) * 1e6);
	}
	std::sort(lat.begin(), lat.end());

	// blocking call on blocks the host registered once (NA_RegisterHostBuffer): the kernels run on the caller's memory, no staging copies
	std::vector<double> latReg;
	{
		std::vector<float> rin(in), rout(count);
		CHECK(NA_RegisterHostBuffer(rin.data(), count * sizeof(float)) == 0);
		CHECK(NA_RegisterHostBuffer(rout.data(), count * sizeof(float)) == 0);
		for (int i = 0; i < 300; i++)
		{
			const double t0 = Now();
			CHECK(NA_BatchProcess(batch, rin.data(), rout.data(), (size_t)frames) == 0);
			if (i >= 50) latReg.push_back((Now() - t0) * 1e6);
		}
		CHECK(NA_UnregisterHostBuffer(rin.data()) == 0);
		CHECK(NA_UnregisterHostBuffer(rout.data()) == 0);
		std::sort(latReg.begin(), latReg.end());
	}

	// the same blocking step through the in-place entry points: the producer writes the pinned input slot, the consumer reads the pinned
	// output slot (no host-side copy of the two 512 KB blocks); latency = Submit .. Collect
	std::vector<double> latInPlace;
	for (int i = 0; i < 300; i++)
	{
		float* slot = NA_BatchNextInput(batch, (size_t)frames);
		CHECK(slot != nullptr);
		std::memcpy(slot, in.data(), count * sizeof(float)); // the producer's write (not part of the latency: a producer writes its samples here anyway)
		const double t0 = Now();
		const int t = NA_BatchSubmit(batch, nullptr, (size_t)frames);
		CHECK(t >= 0);
		CHECK(NA_BatchCollect(batch, t, nullptr) == 0);
		if (i >= 50) latInPlace.push_back((Now() - t0) * 1e6);
	}
	std::sort(latInPlace.begin(), latInPlace.end());

	// copying entry points, two buffers in flight
	int pending = NA_BatchSubmit(batch, in.data(), (size_t)frames);
	CHECK(pending >= 0);
	double t0 = Now();
	for (int i = 0; i < buffers; i++)
	{
		const int next = NA_BatchSubmit(batch, in.data(), (size_t)frames);
		CHECK(next >= 0);
		CHECK(NA_BatchCollect(batch, pending, out.data()) == 0);
		pending = next;
	}
	CHECK(NA_BatchCollect(batch, pending, out.data()) == 0);
	const double usCopy = (Now() - t0) * 1e6 / buffers;

	// zero-copy entry points: the host writes the next input in place and reads the result in place (here: one pass over each);
	// `depth` buffers in flight (the engine has 3 slots)
	double usZero[2] = { 0.0, 0.0 };
	double checksum = 0.0;
	for (int depth = 2; depth <= 3; depth++)
	{
		std::vector<int> tickets;
		for (int k = 0; k < depth - 1; k++)
		{
			float* slot = NA_BatchNextInput(batch, (size_t)frames);
			CHECK(slot != nullptr);
			std::memcpy(slot, in.data(), count * sizeof(float));
			const int t = NA_BatchSubmit(batch, nullptr, (size_t)frames);
			CHECK(t >= 0);
			tickets.push_back(t);
		}
		t0 = Now();
		for (int i = 0; i < buffers; i++)
		{
			float* slot = NA_BatchNextInput(batch, (size_t)frames);
			CHECK(slot != nullptr);
			std::memcpy(slot, in.data(), count * sizeof(float)); // the producer's write
			const int next = NA_BatchSubmit(batch, nullptr, (size_t)frames);
			CHECK(next >= 0);
			tickets.push_back(next);
			const int done = tickets.front();
			tickets.erase(tickets.begin());
			CHECK(NA_BatchCollect(batch, done, nullptr) == 0);
			const float* y = NA_BatchOutputView(batch, done);
			CHECK(y != nullptr);
			for (size_t k = 0; k < count; k += 4096) checksum += y[k]; // the consumer's read (sparse: a real consumer reads all of it)
		}
		for (int t : tickets) CHECK(NA_BatchCollect(batch, t, nullptr) == 0);
		usZero[depth - 2] = (Now() - t0) * 1e6 / buffers;
	}

	std::printf("{\"streams\": %d, \"frames\": %d, \"buffers\": %d, \"us_per_buffer_zero_copy\": %.3f, \"us_per_buffer_zero_copy_3_in_flight\": %.3f, \"us_per_buffer_copying\": %.3f, "
		"\"blocking_latency_us\": {\"p50\": %.1f, \"p99\": %.1f, \"max\": %.1f}, \"in_place_latency_us\": {\"p50\": %.1f, \"p99\": %.1f}, \"registered_blocking_latency_us\": {\"p50\": %.1f, \"p99\": %.1f}, \"checksum\": %.6g}\n",
		streams, frames, buffers, usZero[0], usZero[1], usCopy, lat[lat.size() / 2], lat[(size_t)(lat.size() * 0.99)], lat.back(), latInPlace[latInPlace.size() / 2], latInPlace[(size_t)(latInPlace.size() * 0.99)], latReg[latReg.size() / 2], latReg[(size_t)(latReg.size() * 0.99)], checksum);
	NA_BatchDestroy(batch);
	DeleteModel(model);
	DeleteLoader(loader);
	return 0;
}
