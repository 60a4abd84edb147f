// host_pipe_probe.cpp -- where a pipelined host buffer's time goes (1024 x Standard x 128 samples, zero-copy entry points, two buffers in
// flight): g++ -std=c++17 -O2 -Iinclude tools/microbench/host_pipe_probe.cpp -o /tmp/host_pipe_probe -Lneuralaudio_amd -lNeuralAudioCAPI
// Measured on MI355X: 62-68 us per buffer = memcpy 6-7 + NA_BatchSubmit 12-27 + NA_BatchCollect wait 33-44, i.e. the GPU side takes ~62 us
// per buffer although the kernel is 48: every buffer's launch sits between a stream-wait on its upload event and an event record for its
// download, and those barrier / marker packets cost the compute stream ~14 us per buffer (HSA_ENABLE_SDMA=0 doubles it: 114 us, the
// copies then run as shader kernels behind the compute kernel).  Three buffers in flight change nothing (tools/HostPipeBench).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "neuralaudio_amd.h"
static double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
	NeuralModelLoader* loader = CreateLoader();
	NeuralModel* model = NA_CreateModelFromFileUtf8(loader, argv[1], 0);
	NA_Batch* batch = NA_BatchCreate(0, nullptr);
	NA_BatchAddStreams(batch, model, 1.0f, 1024, 1);
	const size_t count = 1024 * 128;
	std::vector<float> in(count, 0.1f);
	float* slot = NA_BatchNextInput(batch, 128); std::memcpy(slot, in.data(), count * 4);
	int pending = NA_BatchSubmit(batch, nullptr, 128);
	double tCopy = 0, tSub = 0, tCol = 0; const int N = 3000;
	const double t0 = Now();
	for (int i = 0; i < N; i++)
	{
		double a = Now();
		slot = NA_BatchNextInput(batch, 128); std::memcpy(slot, in.data(), count * 4);
		double b = Now();
		const int next = NA_BatchSubmit(batch, nullptr, 128);
		double c = Now();
		NA_BatchCollect(batch, pending, nullptr);
		double d = Now();
		tCopy += b - a; tSub += c - b; tCol += d - c; pending = next;
	}
	NA_BatchCollect(batch, pending, nullptr);
	std::printf("per buffer: total %.1f us = memcpy %.1f + submit %.1f + collect(wait) %.1f\n", (Now() - t0) * 1e6 / N, tCopy * 1e6 / N, tSub * 1e6 / N, tCol * 1e6 / N);
	return 0;
}
