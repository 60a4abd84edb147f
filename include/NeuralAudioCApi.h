/*
 * NeuralAudioCApi.h -- the legacy C ABI of libNeuralAudioCAPI.so, kept symbol-for-symbol so existing
 * FFI consumers (the reference's C# P/Invoke layer, NeuralAudioCSharp/NativeApi.cs:11-54) keep working
 * against the MI355X-native library.
 *
 * Each entry cites the reference declaration it replaces (NeuralAudioCAPI/NeuralAudioCApi.h:<line>) and
 * the definition whose behaviour it reproduces (NeuralAudioCAPI/NeuralAudioCApi.cpp:<line>).
 * Opaque handles wrap one C++ object each, as in the reference (NeuralAudioCApi.cpp:4-12).
 *
 * Behavioural differences (documented in INTEGRATION.md):
 *  - no C++ exception ever crosses this boundary: failures return NULL / leave outputs untouched and
 *    the message is available from NA_GetLastError() (neuralaudio_amd.h);
 *  - CreateModelFromFile returns NULL when loading fails (the reference returns a wrapper around a null
 *    model, NeuralAudioCApi.cpp:29-36; its C# caller already treats NULL as failure).
 */
#ifndef NEURALAUDIO_CAPI_H
#define NEURALAUDIO_CAPI_H

#include <stddef.h>
#ifndef __cplusplus
#include <stdbool.h>
#include <wchar.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_MSC_VER)
#define NA_EXTERN extern __declspec(dllexport)
#else
#define NA_EXTERN extern __attribute__((visibility("default")))
#endif

struct NeuralModel;
struct NeuralModelLoader;
typedef struct NeuralModel NeuralModel;
typedef struct NeuralModelLoader NeuralModelLoader;

/* ref .h:18 / .cpp:14-21 */
NA_EXTERN NeuralModelLoader* CreateLoader(void);
/* ref .h:20 / .cpp:23-27 */
NA_EXTERN void DeleteLoader(NeuralModelLoader* loader);
/* ref .h:22 / .cpp:29-36 -- wchar_t path (UTF-32 on Linux, UTF-16 on Windows); prewarms like CreateFromFile(path) */
NA_EXTERN NeuralModel* CreateModelFromFile(NeuralModelLoader* loader, const wchar_t* modelPath);
/* ref .h:24 / .cpp:38-42 */
NA_EXTERN void DeleteModel(NeuralModel* model);
/* ref .h:26 / .cpp:44-47 -- loadMode: 0 Internal, 1 RTNeural, 2 NAMCore (only 0 is accepted) */
NA_EXTERN void SetLSTMLoadMode(NeuralModelLoader* loader, int loadMode);
/* ref .h:28 / .cpp:49-52 */
NA_EXTERN void SetWaveNetLoadMode(NeuralModelLoader* loader, int loadMode);
/* ref .h:30 / .cpp:54-57 */
NA_EXTERN void SetAudioInputLevelDBu(NeuralModelLoader* loader, float audioDBu);
/* ref .h:32 / .cpp:59-62 */
NA_EXTERN void SetDefaultMaxAudioBufferSize(NeuralModelLoader* loader, int maxSize);
/* ref .h:34 / .cpp:64-67 */
NA_EXTERN int GetLoadMode(NeuralModel* model);
/* ref .h:36 / .cpp:69-72 */
NA_EXTERN bool IsStatic(NeuralModel* model);
/* ref .h:38 / .cpp:74-77 */
NA_EXTERN void SetMaxAudioBufferSize(NeuralModel* model, int maxSize);
/* ref .h:40 / .cpp:79-82 */
NA_EXTERN float GetRecommendedInputDBAdjustment(NeuralModel* model);
/* ref .h:42 / .cpp:84-87 */
NA_EXTERN float GetRecommendedOutputDBAdjustment(NeuralModel* model);
/* ref .h:44 / .cpp:89-92 */
NA_EXTERN float GetSampleRate(NeuralModel* model);
/* ref .h:46 / .cpp:94-97 -- host pointers, mono, numSamples floats each; input == output allowed */
NA_EXTERN void Process(NeuralModel* model, float* input, float* output, size_t numSamples);

#ifdef __cplusplus
}
#endif
#endif
