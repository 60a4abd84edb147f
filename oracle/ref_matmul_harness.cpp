// ref_matmul_harness.cpp -- C-ABI harness around the reference's OWN NeuralAudio/MatMul.h.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes
// <NeuralAudio/MatMul.h> from where it lies under /root/reference (passed with -I by
// oracle/Makefile) and instantiates its templates behind extern "C" entry points.
// MatMul.h is the one arithmetic header of the hot path with no external dependency
// (WaveNet.h / LSTM.h / Activation.h need Eigen, which is absent from this image, so they
// are treated as unbuildable -- see DESIGN.md "Oracle").  Output goes to oracle/_ref/ only.
//
// Used by tests/test_oracle.py to pin the oracle's (3,3) / (3,1) / (8,1) / (1,3) tiny
// mat-muls (A2 "Lite" conv taps, A2 heads, A2 rechannel) against the reference itself.
#include <cstddef>
#include <NeuralAudio/MatMul.h>

using NeuralAudio::MatMul;

#define NA_REF_EXPORT(IN, OUT)                                                                                   \
	extern "C" void na_ref_matmul_init_zero_##IN##_##OUT(const float* in, float* out, const float* w, size_t n)  \
	{                                                                                                            \
		MatMul<float, IN, OUT>::MultiplyInitZero(in, out, w, n);                                                 \
	}                                                                                                            \
	extern "C" void na_ref_matmul_init_colwise_##IN##_##OUT(const float* in, float* out, const float* w,        \
		const float* init, size_t n)                                                                             \
	{                                                                                                            \
		MatMul<float, IN, OUT>::MultiplyInitColwise(in, out, w, init, n);                                        \
	}                                                                                                            \
	extern "C" void na_ref_matmul_accumulate_##IN##_##OUT(const float* in, float* out, const float* w, size_t n) \
	{                                                                                                            \
		MatMul<float, IN, OUT>::MultiplyAccumlulate(in, out, w, n);                                              \
	}

NA_REF_EXPORT(3, 3)
NA_REF_EXPORT(8, 1)
NA_REF_EXPORT(3, 1)
NA_REF_EXPORT(1, 3)
