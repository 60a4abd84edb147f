// vtable_slots.cpp -- prints the vtable slot of every virtual of NeuralAudio::NeuralModel as this library's header declares it.
// Itanium C++ ABI 2.3: a pointer to a virtual member function is { 1 + offset of its vtable entry in bytes, this-adjustment }, so the slot
// can be read from the representation without calling anything.  tests/test_host_cpu.py compares the list with the declaration order of
// the reference header (NeuralAudio/NeuralModel.h:40-134): a host compiled against THAT header calls through the same slots.
#include <NeuralAudio/NeuralModel.h>

#include <cstddef>
#include <cstdio>
#include <cstring>

template <class F>
static long SlotOf(F f)
{
	static_assert(sizeof(F) == 2 * sizeof(std::ptrdiff_t), "Itanium pointer to member function");
	std::ptrdiff_t rep[2];
	std::memcpy(rep, &f, sizeof(rep));
	return (rep[0] & 1) ? (long)((rep[0] - 1) / (std::ptrdiff_t)sizeof(void*)) : -1; // -1: not virtual
}

int main()
{
	using NeuralAudio::NeuralModel;
#define SLOT(name) std::printf("%s %ld\n", #name, SlotOf(&NeuralModel::name))
	SLOT(GetLoadMode); SLOT(HasQualityScaling); SLOT(GetQualityScaleFactor); SLOT(IsQualityChangeRealtimeSafe); SLOT(SetQualityScaleFactor);
	SLOT(IsStatic); SLOT(SetMaxAudioBufferSize); SLOT(SetAudioInputLevelDBu); SLOT(GetAudioInputLevelDBu); SLOT(GetRecommendedInputDBAdjustment);
	SLOT(GetRecommendedOutputDBAdjustment); SLOT(GetSampleRate); SLOT(GetReceptiveFieldSize); SLOT(GetModelVersion); SLOT(GetMetadata);
	SLOT(Process); SLOT(Prewarm);
	std::printf("sizeof %zu\n", sizeof(NeuralModel));
	return 0;
}
