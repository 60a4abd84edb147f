/*
 * na_oracle.h -- CPU restatement of NeuralAudio's "Internal" per-sample inference path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (neuralaudio_amd/, include/,
 * libNeuralAudioCAPI.so) may include, link, call or execute this code.  It is the checker
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * What it restates (file:line relative to the reference tree):
 *   - NeuralAudio/Activation.h:83-118      FastMath tanh / sigmoid / LeakyReLU
 *   - NeuralAudio/Activation.h:20-66       StdMath (libm) variants, used only for the keras KAT
 *   - NeuralAudio/WaveNet.h:30-83          ChannelHistoryBuffer (linear history + rewind)
 *   - NeuralAudio/WaveNet.h:87-297         Conv1DT (dilated causal conv, tap k reads t-d*(K-1-k))
 *   - NeuralAudio/WaveNet.h:299-389        DenseLayerT (1x1)
 *   - NeuralAudio/WaveNet.h:391-494        WaveNetLayerT::Process
 *   - NeuralAudio/WaveNet.h:503-661        WaveNetLayerArrayT::{SetWeights,Prewarm,Process}
 *   - NeuralAudio/WaveNet.h:663-806        WaveNetModelT::{SetWeights,Prewarm,Process}
 *   - NeuralAudio/InternalModel.h:104-117  64-frame chunk loop
 *   - NeuralAudio/LSTM.h:42-100,130-191    LSTM layer / model (NAM and keras weight layouts)
 *   - NeuralAudio/InternalModel.h:368-371 + NeuralModelImpl.h:96-109  LSTM prewarm (2048 zeros, 64/blk)
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - keras LSTM layout / gate order / recurrence: PINNED by the reference's own known-answer
 *     vector (Utils/Models/tw40_blues_deluxe_deerinkstudios.json input_batch -> output_batch,
 *     exact tanh/sigmoid, zero state, no prewarm); tests/test_oracle.py checks <= 1e-6 RMS.
 *   - (3,3)/(3,1)/(8,1)/(1,3) tiny mat-muls: PINNED against the reference's MatMul.h compiled
 *     unmodified into oracle/_ref (it has no external dependencies).
 *   - weight inventory per architecture: PINNED by the reference's sample .nam files (the
 *     restatement consumes exactly weights.size() floats; WaveNet.h:704-709 is the same check).
 *   - WaveNet / NAM-LSTM FastMath output values: **parity unpinned**.  The reference holds no
 *     golden output for them, and WaveNet.h/LSTM.h cannot be built here (they need Eigen, an
 *     external library that is absent; no stand-in headers are written).  They are covered by
 *     an independent float64 restatement (tests/ref_np.py) and by structural properties
 *     (chunk-size invariance, prewarm == long zero lead-in).
 *   - keras GRU: **parity unpinned** -- evaluated by RTNeural in the reference (submodule absent); restated from the published
 *     algorithm and cross-checked against torch.nn.GRU.
 */
#ifndef NA_ORACLE_H
#define NA_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NA_ORACLE_MAX_LAYERS 64
#define NA_ORACLE_MAX_ARRAYS 4

enum { NA_ORACLE_ACT_TANH = 0, NA_ORACLE_ACT_LEAKYRELU = 1 };
enum { NA_ORACLE_MATH_FAST = 0, NA_ORACLE_MATH_STD = 1 };

/* One WaveNet layer-array == template parameters of WaveNetLayerArrayT (WaveNet.h:503). */
typedef struct na_oracle_wn_array_cfg {
	int input_size;
	int condition_size; /* always 1 in every supported model */
	int head_size;
	int head_kernel_size; /* 1 for A1, 16 for A2 */
	int head_dilation;    /* 1 */
	int channels;
	int has_head_bias;
	int activation; /* NA_ORACLE_ACT_* */
	int num_layers;
	int kernel_sizes[NA_ORACLE_MAX_LAYERS];
	int dilations[NA_ORACLE_MAX_LAYERS];
} na_oracle_wn_array_cfg;

typedef struct na_oracle_wavenet na_oracle_wavenet;
typedef struct na_oracle_lstm na_oracle_lstm;
typedef struct na_oracle_gru na_oracle_gru;

/* scalar math, exposed for unit tests */
float na_oracle_fast_tanh(float x);
float na_oracle_fast_sigmoid(float x);
float na_oracle_leaky_relu(float x);

/* test hook: W column-major [cin][cout]; see na_oracle.c */
void na_oracle_test_dense(int cin, int cout, const float* w_colmajor, const float* bias_or_null, const float* in, float* out,
	int frames, int acc);

/* number of weights the architecture consumes (incl. trailing head_scale) */
size_t na_oracle_wavenet_num_weights(int num_arrays, const na_oracle_wn_array_cfg* cfgs);

/* returns NULL if num_weights does not match (reference throws, WaveNet.h:704-709) */
na_oracle_wavenet* na_oracle_wavenet_create(int num_arrays, const na_oracle_wn_array_cfg* cfgs,
	const float* weights, size_t num_weights, int math_mode);
void na_oracle_wavenet_free(na_oracle_wavenet* m);
int na_oracle_wavenet_receptive_field(const na_oracle_wavenet* m);
/* zero the history rings (fresh model, before prewarm) */
void na_oracle_wavenet_reset(na_oracle_wavenet* m);
void na_oracle_wavenet_prewarm(na_oracle_wavenet* m);
/* any num_samples; internally chunked to <= max_frames (64) like InternalModel.h:104-117 */
void na_oracle_wavenet_process(na_oracle_wavenet* m, const float* in, float* out, size_t num_samples);
/* change the chunk size (1..64) -- results must be bit-identical for any value */
void na_oracle_wavenet_set_max_frames(na_oracle_wavenet* m, int max_frames);

/* NAM flat layout (LSTM.h:42-56,130-147). input_size == 1. */
na_oracle_lstm* na_oracle_lstm_create_nam(int num_layers, int hidden_size, const float* weights,
	size_t num_weights, int math_mode);
/* keras layout (LSTM.h:58-85,149-162): per layer kernel[I][4H], recurrent[H][4H], bias[4H] */
na_oracle_lstm* na_oracle_lstm_create_keras(int num_layers, int hidden_size, const float* const* kernels,
	const float* const* recurrents, const float* const* biases, const float* head_weights, float head_bias,
	int math_mode);
void na_oracle_lstm_free(na_oracle_lstm* m);
void na_oracle_lstm_prewarm(na_oracle_lstm* m); /* 2048 zeros */
void na_oracle_lstm_process(na_oracle_lstm* m, const float* in, float* out, size_t num_samples);

/* keras GRU, reset_after form (RTNeural's arithmetic -- absent from the reference tree: PARITY UNPINNED, see na_oracle.c).
 * kernels[l]: [I][3H], recurrents[l]: [H][3H], biases[l]: [2][3H]; gate column blocks z | r | c; hidden <= 64. */
na_oracle_gru* na_oracle_gru_create_keras(int num_layers, int hidden_size, const float* const* kernels,
	const float* const* recurrents, const float* const* biases, const float* head_weights, float head_bias);
void na_oracle_gru_free(na_oracle_gru* m);
void na_oracle_gru_prewarm(na_oracle_gru* m); /* 2048 zeros */
void na_oracle_gru_process(na_oracle_gru* m, const float* in, float* out, size_t num_samples);

/* ModelTest-style timing helpers (Utils/ModelTest/ModelTest.cpp:59-79): run `num_blocks` blocks
 * of `block_size` zeros through `threads` independent copies (one per thread, pthreads);
 * returns wall seconds. Used only by bench.py's cpu_baseline leg. */
double na_oracle_wavenet_bench(int num_arrays, const na_oracle_wn_array_cfg* cfgs, const float* weights,
	size_t num_weights, int block_size, int num_blocks, int threads);
double na_oracle_lstm_bench(int num_layers, int hidden_size, const float* weights, size_t num_weights,
	int block_size, int num_blocks, int threads);
double na_oracle_gru_bench(int num_layers, int hidden_size, const float* const* kernels, const float* const* recurrents,
	const float* const* biases, const float* head_weights, float head_bias, int block_size, int num_blocks, int threads);

/* ---- na_oracle_simd.c: a vectorised variant of the WaveNet path (frames as the vector axis, 8-frame x 8-channel register tiles --
 * the idea of the reference's MULTIFRAME_8X8_CONVOLUTION, WaveNet.h:144-239), for bench.py's cpu_baseline leg only.  Validated against
 * the scalar restatement above (tests/test_oracle.py); it is NOT the parity checker.  condition_size == 1; num_samples a multiple of
 * 8 (process returns -1 otherwise); FastMath; a created model is prewarmed. */
typedef struct na_oracle_simd_wavenet na_oracle_simd_wavenet;
na_oracle_simd_wavenet* na_oracle_simd_wavenet_create(int num_arrays, const na_oracle_wn_array_cfg* cfgs, const float* weights, size_t num_weights);
void na_oracle_simd_wavenet_free(na_oracle_simd_wavenet* m);
int na_oracle_simd_wavenet_process(na_oracle_simd_wavenet* m, const float* in, float* out, size_t num_samples);
double na_oracle_simd_wavenet_bench(int num_arrays, const na_oracle_wn_array_cfg* cfgs, const float* weights, size_t num_weights,
	int block_size, int num_blocks, int threads);

#ifdef __cplusplus
}
#endif
#endif
