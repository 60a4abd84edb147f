/*
 * na_oracle_simd.c -- a vectorised variant of the oracle's WaveNet path, for bench.py's cpu_baseline leg ONLY.
 *
 * TEST INFRASTRUCTURE ONLY (like na_oracle.c): never linked into, imported by, or executed from the product path.  It is not the
 * parity checker either -- that stays the scalar restatement in na_oracle.c, against which this file is validated
 * (tests/test_oracle.py: <= 1e-6 RMS on every official architecture).
 *
 * Why it exists: the scalar port walks [time][channel] arrays one frame at a time and understates what the reference's CPU path
 * does on a modern x86 core.  The reference vectorises through Eigen and, with MULTIFRAME_8X8_CONVOLUTION, evaluates the dilated
 * convolution on 8-frame x 8-channel tiles (NeuralAudio/WaveNet.h:144-239).  This variant takes that idea without Eigen: the FRAME
 * axis is the vector axis (8 floats; GCC vector extensions, so -march=native picks AVX2 / AVX-512 and any other target still
 * compiles), activations are kept [channel][time], and a tile is 8 frames x up to 8 output channels held in registers while the
 * taps and input channels stream past with broadcast weights.  Same arithmetic per value as WaveNet.h:462-494 / na_oracle.c
 * layer_process (conv taps, + bias + mix-in, activation, head accumulation, 1x1 + bias + residual), another summation order
 * (bias first) -- f32 reassociation noise.
 *
 * Restrictions (it is a benchmark aid): condition_size == 1, chunks of 8 .. 64 frames in multiples of 8, FastMath tanh / LeakyReLU.
 */
#include "na_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXF 64
#define BUF_PADDING 24
#define VW 8

typedef float v8 __attribute__((vector_size(32), aligned(4)));   /* unaligned loads / stores */
typedef int v8i __attribute__((vector_size(32), aligned(4)));

static inline v8 splat(float x) { return (v8){ x, x, x, x, x, x, x, x }; }
static inline v8 vabs(v8 x) { return (v8)((v8i)x & (v8i){ 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff }); }

/* Activation.h:83-91, eight values at a time */
static inline v8 fast_tanh8(v8 x)
{
	const v8 ax = vabs(x);
	const v8 x2 = x * x;
	return (x * (splat(2.45550750702956f) + splat(2.45550750702956f) * ax + (splat(0.893229853513558f) + splat(0.821226666969744f) * ax) * x2)
		/ (splat(2.44506634652299f) + (splat(2.44506634652299f) + x2) * vabs(x + splat(0.814642734961073f) * x * ax)));
}

/* Activation.h:110-118 */
static inline v8 leaky8(v8 x)
{
	const v8i pos = x > splat(0.0f);
	const v8 scaled = splat(0.01f) * x;
	return (v8)(((v8i)x & pos) | ((v8i)scaled & ~pos));
}

/* a [channels][time] history: row r at buf + r * stride; the current block starts at column `start` (ChannelHistoryBuffer,
 * WaveNet.h:30-83, one row per channel instead of one column per frame) */
typedef struct {
	int rows, rf, stride, start;
	float* buf;
} hist;

static void hist_init(hist* h, int rows, int rf)
{
	h->rows = rows; h->rf = rf;
	h->stride = rf + (BUF_PADDING + 1) * MAXF;
	h->buf = (float*)calloc((size_t)rows * h->stride, sizeof(float));
	h->start = rf;
}

static void hist_advance(hist* h, int frames)
{
	h->start += frames;
	if (h->start + MAXF > h->stride) {
		for (int r = 0; r < h->rows; r++)
			memmove(h->buf + (size_t)r * h->stride, h->buf + (size_t)r * h->stride + (h->start - h->rf), (size_t)h->rf * sizeof(float));
		h->start = h->rf;
	}
}

typedef struct {
	int C, K, dil;
	float* wconv; /* [k][c][o] */
	float* bconv; /* [o] */
	float* wmix;  /* [o] */
	float* w1;    /* [c][o] */
	float* b1;    /* [o] */
	hist in;      /* the layer's input ring */
} s_layer;

typedef struct {
	na_oracle_wn_array_cfg cfg;
	s_layer* layers;
	float* wre;   /* [i][o] */
	int HK;
	float* whead; /* [k][c][h] */
	float* bhead; /* [h] */
	hist head;    /* head conv input ring (C rows) */
	float* outputs;      /* [C][MAXF]: the last layer's output */
	float* head_outputs; /* [head][MAXF] */
} s_array;

struct na_oracle_simd_wavenet {
	int num_arrays;
	s_array arrays[NA_ORACLE_MAX_ARRAYS];
	float* z;    /* [maxC][MAXF] */
	float* head; /* [C0][MAXF] */
	float head_scale;
	int receptive_field;
};
typedef struct na_oracle_simd_wavenet na_oracle_simd_wavenet;

/* One tile: 8 frames x NO output channels.  acc[o] starts at `init[o]` (a vector per channel), then every (tap, input channel)
 * adds w * x with x read from the rows of `src` at column col0 + tap offset.  NO is a literal at every call site. */
static inline __attribute__((always_inline)) void conv_tile(const float* w, int ldw, int ob, int NO, int K, int dil, int Cin, const float* src, int stride,
	int col0, v8* acc)
{
	for (int k = 0; k < K; k++) {
		const int off = dil * (k + 1 - K);
		const float* wk = w + (size_t)k * Cin * ldw + ob;
		for (int c = 0; c < Cin; c++) {
			const v8 x = *(const v8*)(src + (size_t)c * stride + col0 + off);
			const float* wc = wk + (size_t)c * ldw;
			for (int o = 0; o < NO; o++) acc[o] += splat(wc[o]) * x;
		}
	}
}

#define TILE_CASES(CALL) \
	switch (no) { \
	case 8: CALL(8); break; case 7: CALL(7); break; case 6: CALL(6); break; case 5: CALL(5); break; \
	case 4: CALL(4); break; case 3: CALL(3); break; case 2: CALL(2); break; default: CALL(1); break; }

/* WaveNet.h:462-494 for one chunk: z = act(conv(x) + b + wmix * cond); head += z; next = W1 z + b1 + x */
static void layer_chunk(s_layer* l, int act, const float* cond, float* z, float* head, float* next, int nstride, int ncol, int frames, int need_output)
{
	const int C = l->C;
	for (int f0 = 0; f0 < frames; f0 += VW) {
		const v8 cv = *(const v8*)(cond + f0);
		for (int ob = 0; ob < C; ob += 8) {
			const int no = C - ob < 8 ? C - ob : 8;
			v8 acc[8];
			for (int o = 0; o < no; o++) acc[o] = splat(l->bconv[ob + o]) + splat(l->wmix[ob + o]) * cv;
#define CONV_CALL(N) conv_tile(l->wconv, C, ob, N, l->K, l->dil, C, l->in.buf, l->in.stride, l->in.start + f0, acc)
			TILE_CASES(CONV_CALL)
#undef CONV_CALL
			for (int o = 0; o < no; o++) {
				const v8 a = act == NA_ORACLE_ACT_TANH ? fast_tanh8(acc[o]) : leaky8(acc[o]);
				*(v8*)(z + (size_t)(ob + o) * MAXF + f0) = a;
				*(v8*)(head + (size_t)(ob + o) * MAXF + f0) += a;
			}
		}
		if (!need_output) continue;
		for (int ob = 0; ob < C; ob += 8) {
			const int no = C - ob < 8 ? C - ob : 8;
			v8 acc[8];
			for (int o = 0; o < no; o++) acc[o] = splat(l->b1[ob + o]) + *(const v8*)(l->in.buf + (size_t)(ob + o) * l->in.stride + l->in.start + f0);
#define ONE_CALL(N) conv_tile(l->w1, C, ob, N, 1, 1, C, z, MAXF, f0, acc)
			TILE_CASES(ONE_CALL)
#undef ONE_CALL
			for (int o = 0; o < no; o++) *(v8*)(next + (size_t)(ob + o) * nstride + ncol + f0) = acc[o];
		}
	}
}

/* WaveNet.h:632-661 */
static void array_chunk(na_oracle_simd_wavenet* m, s_array* a, const float* inputs, int in_rows, const float* cond, float* head, int frames, int need_output)
{
	const int C = a->cfg.channels, nl = a->cfg.num_layers;
	/* rechannel into layer 0's ring (no bias) */
	{
		hist* h0 = &a->layers[0].in;
		for (int f0 = 0; f0 < frames; f0 += VW)
			for (int ob = 0; ob < C; ob += 8) {
				const int no = C - ob < 8 ? C - ob : 8;
				v8 acc[8];
				for (int o = 0; o < no; o++) acc[o] = splat(0.0f);
#define RE_CALL(N) conv_tile(a->wre, C, ob, N, 1, 1, in_rows, inputs, MAXF, f0, acc)
				TILE_CASES(RE_CALL)
#undef RE_CALL
				for (int o = 0; o < no; o++) *(v8*)(h0->buf + (size_t)(ob + o) * h0->stride + h0->start + f0) = acc[o];
			}
	}
	for (int i = 0; i < nl; i++) {
		s_layer* l = &a->layers[i];
		if (i == nl - 1) layer_chunk(l, a->cfg.activation, cond, m->z, head, a->outputs, MAXF, 0, frames, need_output);
		else layer_chunk(l, a->cfg.activation, cond, m->z, head, a->layers[i + 1].in.buf, a->layers[i + 1].in.stride, a->layers[i + 1].in.start, frames, 1);
		hist_advance(&l->in, frames);
	}
	/* head conv: copy the accumulated head into its ring, then K taps (+ bias) */
	for (int c = 0; c < C; c++) memcpy(a->head.buf + (size_t)c * a->head.stride + a->head.start, head + (size_t)c * MAXF, (size_t)frames * sizeof(float));
	const int H = a->cfg.head_size;
	for (int f0 = 0; f0 < frames; f0 += VW)
		for (int ob = 0; ob < H; ob += 8) {
			const int no = H - ob < 8 ? H - ob : 8;
			v8 acc[8];
			for (int o = 0; o < no; o++) acc[o] = splat(a->cfg.has_head_bias ? a->bhead[ob + o] : 0.0f);
#define HEAD_CALL(N) conv_tile(a->whead, H, ob, N, a->HK, a->cfg.head_dilation, C, a->head.buf, a->head.stride, a->head.start + f0, acc)
			TILE_CASES(HEAD_CALL)
#undef HEAD_CALL
			for (int o = 0; o < no; o++) *(v8*)(a->head_outputs + (size_t)(ob + o) * MAXF + f0) = acc[o];
		}
	hist_advance(&a->head, frames);
}

static void simd_chunk(na_oracle_simd_wavenet* m, const float* in, float* out, int frames)
{
	float cond[MAXF];
	memcpy(cond, in, (size_t)frames * sizeof(float));
	memset(m->head, 0, (size_t)MAXF * m->arrays[0].cfg.channels * sizeof(float));
	const int last = m->num_arrays - 1;
	for (int i = 0; i < m->num_arrays; i++) {
		if (i == 0) array_chunk(m, &m->arrays[0], cond, 1, cond, m->head, frames, 1);
		else array_chunk(m, &m->arrays[i], m->arrays[i - 1].outputs, m->arrays[i - 1].cfg.channels, cond, m->arrays[i - 1].head_outputs, frames, i != last);
	}
	const float* fh = m->arrays[last].head_outputs;
	for (int f = 0; f < frames; f++) out[f] = m->head_scale * fh[f];
}

void na_oracle_simd_wavenet_free(na_oracle_simd_wavenet* m)
{
	if (!m) return;
	for (int i = 0; i < m->num_arrays; i++) {
		s_array* a = &m->arrays[i];
		for (int l = 0; a->layers && l < a->cfg.num_layers; l++) {
			s_layer* L = &a->layers[l];
			free(L->wconv); free(L->bconv); free(L->wmix); free(L->w1); free(L->b1); free(L->in.buf);
		}
		free(a->layers); free(a->wre); free(a->whead); free(a->bhead); free(a->head.buf); free(a->outputs); free(a->head_outputs);
	}
	free(m->z); free(m->head);
	free(m);
}

/* any multiple of 8 samples; chunks of 64 like InternalModel.h:104-117 */
int na_oracle_simd_wavenet_process(na_oracle_simd_wavenet* m, const float* in, float* out, size_t num_samples)
{
	if (num_samples % VW != 0) return -1;
	size_t offset = 0;
	while (num_samples > 0) {
		const int n = (int)(num_samples < (size_t)MAXF ? num_samples : (size_t)MAXF);
		simd_chunk(m, in + offset, out + offset, n);
		offset += (size_t)n;
		num_samples -= (size_t)n;
	}
	return 0;
}

/* Weight order: WaveNet.h:570-580 (array), :420-425 (layer), :99-111 (conv: [o][c][k], k fastest), :308-319 (dense: [o][i]).
 * The model starts prewarmed: the zero-input steady state (WaveNetModelT::Prewarm, :746-766) is what a receptive field of zeros from
 * all-zero rings leaves behind -- without input every layer's values are constant in time, so running the zeros reproduces it. */
na_oracle_simd_wavenet* na_oracle_simd_wavenet_create(int num_arrays, const na_oracle_wn_array_cfg* cfgs, const float* weights, size_t num_weights)
{
	if (num_arrays < 1 || num_arrays > NA_ORACLE_MAX_ARRAYS) return NULL;
	if (na_oracle_wavenet_num_weights(num_arrays, cfgs) != num_weights) return NULL;
	for (int i = 0; i < num_arrays; i++)
		if (cfgs[i].condition_size != 1) return NULL;
	for (int i = 1; i < num_arrays; i++)
		if (cfgs[i - 1].head_size != cfgs[i].channels || cfgs[i].input_size != cfgs[i - 1].channels) return NULL;
	na_oracle_simd_wavenet* m = (na_oracle_simd_wavenet*)calloc(1, sizeof(*m));
	m->num_arrays = num_arrays;
	const float* it = weights;
	int maxC = 1;
	for (int ai = 0; ai < num_arrays; ai++) {
		s_array* a = &m->arrays[ai];
		a->cfg = cfgs[ai];
		const int C = a->cfg.channels, I = a->cfg.input_size, H = a->cfg.head_size;
		if (C > maxC) maxC = C;
		a->wre = (float*)calloc((size_t)I * C, sizeof(float));
		for (int o = 0; o < C; o++)
			for (int i = 0; i < I; i++) a->wre[(size_t)i * C + o] = *(it++);
		a->layers = (s_layer*)calloc((size_t)a->cfg.num_layers, sizeof(s_layer));
		for (int li = 0; li < a->cfg.num_layers; li++) {
			s_layer* l = &a->layers[li];
			l->C = C; l->K = a->cfg.kernel_sizes[li]; l->dil = a->cfg.dilations[li];
			l->wconv = (float*)calloc((size_t)l->K * C * C, sizeof(float));
			l->bconv = (float*)calloc((size_t)C, sizeof(float));
			l->wmix = (float*)calloc((size_t)C, sizeof(float));
			l->w1 = (float*)calloc((size_t)C * C, sizeof(float));
			l->b1 = (float*)calloc((size_t)C, sizeof(float));
			for (int o = 0; o < C; o++)
				for (int c = 0; c < C; c++)
					for (int k = 0; k < l->K; k++) l->wconv[((size_t)k * C + c) * C + o] = *(it++);
			for (int o = 0; o < C; o++) l->bconv[o] = *(it++);
			for (int o = 0; o < C; o++) l->wmix[o] = *(it++);
			for (int o = 0; o < C; o++)
				for (int c = 0; c < C; c++) l->w1[(size_t)c * C + o] = *(it++);
			for (int o = 0; o < C; o++) l->b1[o] = *(it++);
			hist_init(&l->in, C, (l->K - 1) * l->dil);
			m->receptive_field += (l->K - 1) * l->dil;
		}
		a->HK = a->cfg.head_kernel_size;
		a->whead = (float*)calloc((size_t)a->HK * C * H, sizeof(float));
		a->bhead = (float*)calloc((size_t)H, sizeof(float));
		for (int h = 0; h < H; h++)
			for (int c = 0; c < C; c++)
				for (int k = 0; k < a->HK; k++) a->whead[((size_t)k * C + c) * H + h] = *(it++);
		if (a->cfg.has_head_bias)
			for (int h = 0; h < H; h++) a->bhead[h] = *(it++);
		hist_init(&a->head, C, (a->HK - 1) * a->cfg.head_dilation);
		m->receptive_field += (a->HK - 1) * a->cfg.head_dilation;
		a->outputs = (float*)calloc((size_t)C * MAXF, sizeof(float));
		a->head_outputs = (float*)calloc((size_t)(H > 0 ? H : 1) * MAXF, sizeof(float));
	}
	m->head_scale = *(it++);
	m->z = (float*)calloc((size_t)maxC * MAXF, sizeof(float));
	m->head = (float*)calloc((size_t)m->arrays[0].cfg.channels * MAXF, sizeof(float));
	/* prewarm */
	float zeros[MAXF] = { 0 }, sink[MAXF];
	for (int done = 0; done < m->receptive_field + MAXF; done += MAXF) simd_chunk(m, zeros, sink, MAXF);
	return m;
}

/* ---- ModelTest-style timing (Utils/ModelTest/ModelTest.cpp:59-79), one model instance per thread ---- */
typedef struct {
	int num_arrays; const na_oracle_wn_array_cfg* cfgs; const float* weights; size_t num_weights;
	int block_size, num_blocks;
	pthread_barrier_t* barrier;
} sbench_arg;

static void* sbench_thread(void* p)
{
	sbench_arg* a = (sbench_arg*)p;
	float* in = (float*)calloc((size_t)a->block_size, sizeof(float));
	float* out = (float*)calloc((size_t)a->block_size, sizeof(float));
	na_oracle_simd_wavenet* m = na_oracle_simd_wavenet_create(a->num_arrays, a->cfgs, a->weights, a->num_weights);
	pthread_barrier_wait(a->barrier);
	for (int b = 0; m && b < a->num_blocks; b++) na_oracle_simd_wavenet_process(m, in, out, (size_t)a->block_size);
	pthread_barrier_wait(a->barrier);
	na_oracle_simd_wavenet_free(m);
	free(in); free(out);
	return NULL;
}

double na_oracle_simd_wavenet_bench(int num_arrays, const na_oracle_wn_array_cfg* cfgs, const float* weights, size_t num_weights, int block_size,
	int num_blocks, int threads)
{
	if (threads < 1) threads = 1;
	if (block_size % VW != 0) return -1.0;
	pthread_barrier_t barrier;
	pthread_barrier_init(&barrier, NULL, (unsigned)threads + 1);
	pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
	sbench_arg* args = (sbench_arg*)calloc((size_t)threads, sizeof(sbench_arg));
	for (int t = 0; t < threads; t++) {
		args[t] = (sbench_arg){ num_arrays, cfgs, weights, num_weights, block_size, num_blocks, &barrier };
		pthread_create(&tids[t], NULL, sbench_thread, &args[t]);
	}
	struct timespec t0, t1;
	pthread_barrier_wait(&barrier);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	pthread_barrier_wait(&barrier);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	for (int t = 0; t < threads; t++) pthread_join(tids[t], NULL);
	pthread_barrier_destroy(&barrier);
	free(tids); free(args);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
