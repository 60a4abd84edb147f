/*
 * na_oracle.c -- CPU restatement of NeuralAudio's Internal WaveNet / LSTM path (see na_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, smoke() and bench.py's cpu_baseline leg.
 * Never linked into, imported by, or executed from the product path.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference tree).
 * Data layout follows the reference: activations are [time][channel] (channels contiguous,
 * NeuralAudio/ChannelBuffer.h:116), weights are column-major ChannelBuffer<T,Out,In>
 * (ptr[j*Out + i] == W(i,j), ChannelBuffer.h:85-93).
 *
 * Build with -ffp-contract=off for the checker library so results do not depend on the host ISA.
 */
#include "na_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXF 64          /* WAVENET_MAX_NUM_FRAMES, WaveNet.h:14-16 */
#define BUF_PADDING 24   /* LAYER_ARRAY_BUFFER_PADDING, WaveNet.h:18-20 */

/* ---------------------------------------------------------------- math policies (Activation.h) */

/* Activation.h:83-91 -- same association, float constants (TCONST casts the double literal). */
float na_oracle_fast_tanh(float x)
{
	const float ax = fabsf(x);
	const float x2 = x * x;
	return (x * (2.45550750702956f + 2.45550750702956f * ax + (0.893229853513558f + 0.821226666969744f * ax) * x2)
		/ (2.44506634652299f + (2.44506634652299f + x2) * fabsf(x + 0.814642734961073f * x * ax)));
}

/* Activation.h:93-96 */
float na_oracle_fast_sigmoid(float x)
{
	return 0.5f * (na_oracle_fast_tanh(x * 0.5f) + 1.0f);
}

/* Activation.h:110-118 */
float na_oracle_leaky_relu(float x)
{
	return x > 0.0f ? x : 0.01f * x;
}

/* Activation.h:35-43 (StdMath) */
static float std_tanh(float x) { return tanhf(x); }
static float std_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/* ---------------------------------------------------------------- ChannelHistoryBuffer + Conv1DT */

typedef struct {
	int cin, cout, ksize, dil, has_bias;
	float* w;    /* [ksize][cin][cout]: w[(k*cin + j)*cout + i] = W_k(i,j)   (WaveNet.h:99-104) */
	float* bias; /* [cout] */
	int rf;      /* (ksize-1)*dil, WaveNet.h:91 */
	int bufsize; /* rf + (PADDING+1)*MAXF, WaveNet.h:34 */
	int start;   /* bufferStart */
	int alloc_num;
	float* buf;  /* [bufsize][cin] */
} conv1d;

static void conv1d_init(conv1d* c, int cin, int cout, int ksize, int dil, int has_bias)
{
	c->cin = cin; c->cout = cout; c->ksize = ksize; c->dil = dil; c->has_bias = has_bias;
	c->rf = (ksize - 1) * dil;
	c->bufsize = c->rf + (BUF_PADDING + 1) * MAXF;
	c->w = (float*)calloc((size_t)ksize * cin * cout, sizeof(float));
	c->bias = (float*)calloc((size_t)cout, sizeof(float));
	c->buf = (float*)calloc((size_t)c->bufsize * cin, sizeof(float));
	c->start = c->rf;
	c->alloc_num = 0;
}

static void conv1d_free(conv1d* c) { free(c->w); free(c->bias); free(c->buf); }

/* WaveNet.h:38-57 AllocBuffer: zero, stagger the start so rewinds of different layers fall in different blocks */
static void conv1d_alloc(conv1d* c, int alloc_num)
{
	c->alloc_num = alloc_num;
	memset(c->buf, 0, (size_t)c->bufsize * c->cin * sizeof(float));
	c->start = c->bufsize - (MAXF * ((alloc_num % BUF_PADDING) + 1));
}

static size_t conv1d_num_weights(int cin, int cout, int ksize, int has_bias)
{
	return (size_t)cout * cin * ksize + (has_bias ? cout : 0); /* WaveNet.h:94-97 */
}

/* WaveNet.h:99-111: for i<Out, j<In, k<K: weights[k](i,j) = *w++ ; then bias */
static const float* conv1d_set_weights(conv1d* c, const float* it)
{
	for (int i = 0; i < c->cout; i++)
		for (int j = 0; j < c->cin; j++)
			for (int k = 0; k < c->ksize; k++)
				c->w[((size_t)k * c->cin + j) * c->cout + i] = *(it++);
	if (c->has_bias)
		for (int i = 0; i < c->cout; i++)
			c->bias[i] = *(it++);
	return it;
}

/* WaveNet.h:59-72 AdvanceFrames / RewindBuffer */
static void conv1d_advance(conv1d* c, int frames)
{
	c->start += frames;
	if (c->start + MAXF > c->bufsize) {
		memmove(c->buf, c->buf + (size_t)(c->start - c->rf) * c->cin, (size_t)c->rf * c->cin * sizeof(float));
		c->start = c->rf;
	}
}

/* WaveNet.h:74-82 CopyBuffer: replicate the column at bufferStart over the whole receptive field */
static void conv1d_copy_buffer(conv1d* c)
{
	const float* src = c->buf + (size_t)c->start * c->cin;
	for (int off = 1; off < c->rf + 1; off++)
		memcpy(c->buf + (size_t)(c->start - off) * c->cin, src, (size_t)c->cin * sizeof(float));
}

static float* conv1d_input(conv1d* c) { return c->buf + (size_t)c->start * c->cin; } /* WaveNet.h:134-137 */

/* WaveNet.h:139-290: out[f][o] = sum_k sum_c W_k(o,c) * buf[start + f + d*(k+1-K)][c]  (+ bias)
 * Summation order: taps outermost, then input channel (the generic path :242-286 / tile path :144-239);
 * bias is folded into tap 0 when a MatMul kernel exists (:264) and added last otherwise (:288-289) --
 * we add it last; the difference is f32 reassociation noise (~1e-7). */
static void conv1d_process(const conv1d* c, float* out, int frames)
{
	const int cin = c->cin, cout = c->cout;
	for (int f = 0; f < frames; f++) {
		float* o = out + (size_t)f * cout;
		for (int i = 0; i < cout; i++) o[i] = 0.0f;
		for (int k = 0; k < c->ksize; k++) {
			const int offset = c->dil * (k + 1 - c->ksize);
			const float* h = c->buf + (size_t)(c->start + offset + f) * cin;
			const float* wk = c->w + (size_t)k * cin * cout;
			for (int j = 0; j < cin; j++) {
				const float hv = h[j];
				const float* wcol = wk + (size_t)j * cout;
				for (int i = 0; i < cout; i++) o[i] += wcol[i] * hv;
			}
		}
		if (c->has_bias)
			for (int i = 0; i < cout; i++) o[i] += c->bias[i];
	}
}

/* ---------------------------------------------------------------- DenseLayerT (WaveNet.h:299-389) */

typedef struct {
	int cin, cout, has_bias;
	float* w;    /* [cin][cout] column-major */
	float* bias;
} dense;

static void dense_init(dense* d, int cin, int cout, int has_bias)
{
	d->cin = cin; d->cout = cout; d->has_bias = has_bias;
	d->w = (float*)calloc((size_t)cin * cout, sizeof(float));
	d->bias = (float*)calloc((size_t)cout, sizeof(float));
}

static void dense_free(dense* d) { free(d->w); free(d->bias); }

static size_t dense_num_weights(int cin, int cout, int has_bias) { return (size_t)cout * cin + (has_bias ? cout : 0); }

/* WaveNet.h:308-319 */
static const float* dense_set_weights(dense* d, const float* it)
{
	for (int i = 0; i < d->cout; i++)
		for (int j = 0; j < d->cin; j++)
			d->w[(size_t)j * d->cout + i] = *(it++);
	if (d->has_bias)
		for (int i = 0; i < d->cout; i++)
			d->bias[i] = *(it++);
	return it;
}

/* WaveNet.h:336-362 Process (acc==0) / :364-383 ProcessAcc (acc==1) */
static void dense_process(const dense* d, const float* in, float* out, int frames, int acc)
{
	const int cin = d->cin, cout = d->cout;
	float tmp[cout > 0 ? cout : 1]; /* (any width: the dynamic engine takes layer arrays wider than 64 channels) */
	for (int f = 0; f < frames; f++) {
		const float* x = in + (size_t)f * cin;
		float* o = out + (size_t)f * cout;
		for (int i = 0; i < cout; i++) tmp[i] = 0.0f;
		for (int j = 0; j < cin; j++) {
			const float xv = x[j];
			const float* wcol = d->w + (size_t)j * cout;
			for (int i = 0; i < cout; i++) tmp[i] += wcol[i] * xv;
		}
		if (d->has_bias)
			for (int i = 0; i < cout; i++) tmp[i] += d->bias[i];
		if (acc)
			for (int i = 0; i < cout; i++) o[i] += tmp[i];
		else
			for (int i = 0; i < cout; i++) o[i] = tmp[i];
	}
}

/* Test hook: the oracle's 1x1 / conv-tap arithmetic on caller-supplied column-major weights (W(i,j) = w[j*cout+i],
 * the reference's ChannelBuffer<T,Out,In> layout), so it can be pinned against the reference's MatMul.h vectors.
 * acc == 0: out = W*in (+bias);  acc == 1: out += W*in (+bias). */
void na_oracle_test_dense(int cin, int cout, const float* w_colmajor, const float* bias_or_null, const float* in, float* out,
	int frames, int acc)
{
	dense d;
	dense_init(&d, cin, cout, bias_or_null != NULL);
	memcpy(d.w, w_colmajor, (size_t)cin * cout * sizeof(float));
	if (bias_or_null) memcpy(d.bias, bias_or_null, (size_t)cout * sizeof(float));
	dense_process(&d, in, out, frames, acc);
	dense_free(&d);
}

/* ---------------------------------------------------------------- WaveNetLayerT (WaveNet.h:391-494) */

typedef struct {
	int channels, activation, math_mode;
	conv1d conv;
	dense mixin;     /* ConditionSize -> Channels, no bias */
	dense one_by_one; /* Channels -> Channels, bias */
	float* state;    /* [MAXF][channels] */
} wn_layer;

static void layer_init(wn_layer* l, int cond, int channels, int ksize, int dil, int activation, int math_mode)
{
	l->channels = channels; l->activation = activation; l->math_mode = math_mode;
	conv1d_init(&l->conv, channels, channels, ksize, dil, 1);
	dense_init(&l->mixin, cond, channels, 0);
	dense_init(&l->one_by_one, channels, channels, 1);
	l->state = (float*)calloc((size_t)MAXF * channels, sizeof(float));
}

static void layer_free(wn_layer* l)
{
	conv1d_free(&l->conv); dense_free(&l->mixin); dense_free(&l->one_by_one); free(l->state);
}

/* WaveNet.h:462-494 */
static void layer_process(wn_layer* l, const float* cond, float* head_input, float* output, int frames, int need_output)
{
	const int c = l->channels;
	float* block = l->state;
	conv1d_process(&l->conv, block, frames);                 /* :468 */
	dense_process(&l->mixin, cond, block, frames, 1);        /* :471 */
	const int n = frames * c;
	if (l->activation == NA_ORACLE_ACT_TANH) {               /* :473-480 */
		if (l->math_mode == NA_ORACLE_MATH_FAST)
			for (int p = 0; p < n; p++) block[p] = na_oracle_fast_tanh(block[p]);
		else
			for (int p = 0; p < n; p++) block[p] = std_tanh(block[p]);
	} else {
		for (int p = 0; p < n; p++) block[p] = na_oracle_leaky_relu(block[p]);
	}
	for (int p = 0; p < n; p++) head_input[p] += block[p];    /* :482 */
	if (need_output) {                                        /* :486-491 */
		const float* in = conv1d_input(&l->conv);
		/* output may alias nothing we read here: it is the next layer's ring or arrayOutputs */
		dense_process(&l->one_by_one, block, output, frames, 0);
		for (int p = 0; p < n; p++) output[p] += in[p];
	}
}

/* ---------------------------------------------------------------- WaveNetLayerArrayT (WaveNet.h:503-661) */

typedef struct {
	na_oracle_wn_array_cfg cfg;
	wn_layer* layers;
	dense rechannel;     /* InputSize -> Channels, no bias */
	conv1d head_rechannel; /* Channels -> HeadSize, K=head_kernel_size, bias per cfg */
	float* array_outputs; /* [MAXF][channels] */
	float* head_outputs;  /* [MAXF][head_size] */
	int receptive_field;
} wn_array;

static void array_init(wn_array* a, const na_oracle_wn_array_cfg* cfg, int math_mode)
{
	a->cfg = *cfg;
	a->layers = (wn_layer*)calloc((size_t)cfg->num_layers, sizeof(wn_layer));
	a->receptive_field = 0;
	for (int i = 0; i < cfg->num_layers; i++) {
		layer_init(&a->layers[i], cfg->condition_size, cfg->channels, cfg->kernel_sizes[i], cfg->dilations[i],
			cfg->activation, math_mode);
		a->receptive_field += a->layers[i].conv.rf; /* :534-542 */
	}
	dense_init(&a->rechannel, cfg->input_size, cfg->channels, 0);
	conv1d_init(&a->head_rechannel, cfg->channels, cfg->head_size, cfg->head_kernel_size, cfg->head_dilation,
		cfg->has_head_bias);
	a->receptive_field += a->head_rechannel.rf;
	a->array_outputs = (float*)calloc((size_t)MAXF * cfg->channels, sizeof(float));
	a->head_outputs = (float*)calloc((size_t)MAXF * cfg->head_size, sizeof(float));
}

static void array_free(wn_array* a)
{
	for (int i = 0; i < a->cfg.num_layers; i++) layer_free(&a->layers[i]);
	free(a->layers);
	dense_free(&a->rechannel); conv1d_free(&a->head_rechannel);
	free(a->array_outputs); free(a->head_outputs);
}

/* :544-554 */
static int array_alloc_buffers(wn_array* a, int alloc_num)
{
	for (int i = 0; i < a->cfg.num_layers; i++) conv1d_alloc(&a->layers[i].conv, alloc_num++);
	conv1d_alloc(&a->head_rechannel, alloc_num++);
	return alloc_num;
}

static size_t array_num_weights(const na_oracle_wn_array_cfg* cfg)
{
	size_t n = dense_num_weights(cfg->input_size, cfg->channels, 0);
	for (int i = 0; i < cfg->num_layers; i++) {
		n += conv1d_num_weights(cfg->channels, cfg->channels, cfg->kernel_sizes[i], 1);
		n += dense_num_weights(cfg->condition_size, cfg->channels, 0);
		n += dense_num_weights(cfg->channels, cfg->channels, 1);
	}
	n += conv1d_num_weights(cfg->channels, cfg->head_size, cfg->head_kernel_size, cfg->has_head_bias);
	return n;
}

/* :570-580 (array) and :420-425 (layer: conv1D, inputMixin, oneByOne) */
static const float* array_set_weights(wn_array* a, const float* it)
{
	it = dense_set_weights(&a->rechannel, it);
	for (int i = 0; i < a->cfg.num_layers; i++) {
		it = conv1d_set_weights(&a->layers[i].conv, it);
		it = dense_set_weights(&a->layers[i].mixin, it);
		it = dense_set_weights(&a->layers[i].one_by_one, it);
	}
	it = conv1d_set_weights(&a->head_rechannel, it);
	return it;
}

/* :607-630 -- one frame per layer, no cursor advance */
static void array_prewarm(wn_array* a, const float* layer_inputs, const float* cond, float* head_inputs)
{
	const int nl = a->cfg.num_layers;
	dense_process(&a->rechannel, layer_inputs, conv1d_input(&a->layers[0].conv), 1, 0);
	for (int i = 0; i < nl; i++) {
		conv1d_copy_buffer(&a->layers[i].conv);
		float* out = (i == nl - 1) ? a->array_outputs : conv1d_input(&a->layers[i + 1].conv);
		layer_process(&a->layers[i], cond, head_inputs, out, 1, 1);
	}
	memcpy(conv1d_input(&a->head_rechannel), head_inputs, (size_t)a->cfg.channels * sizeof(float));
	conv1d_copy_buffer(&a->head_rechannel);
	conv1d_process(&a->head_rechannel, a->head_outputs, 1);
}

/* :632-661 */
static void array_process(wn_array* a, const float* layer_inputs, const float* cond, float* head_inputs, int frames,
	int need_output)
{
	const int nl = a->cfg.num_layers;
	dense_process(&a->rechannel, layer_inputs, conv1d_input(&a->layers[0].conv), frames, 0); /* :637 */
	for (int i = 0; i < nl; i++) {
		if (i == nl - 1)
			layer_process(&a->layers[i], cond, head_inputs, a->array_outputs, frames, need_output);
		else
			layer_process(&a->layers[i], cond, head_inputs, conv1d_input(&a->layers[i + 1].conv), frames, 1);
		conv1d_advance(&a->layers[i].conv, frames); /* :650 */
	}
	memcpy(conv1d_input(&a->head_rechannel), head_inputs, (size_t)frames * a->cfg.channels * sizeof(float)); /* :658 */
	conv1d_process(&a->head_rechannel, a->head_outputs, frames);
	conv1d_advance(&a->head_rechannel, frames);
}

/* ---------------------------------------------------------------- WaveNetModelT (WaveNet.h:663-806) */

struct na_oracle_wavenet {
	int num_arrays;
	wn_array arrays[NA_ORACLE_MAX_ARRAYS];
	float condition[MAXF];
	float* head_array; /* [MAXF][channels of array 0] */
	float head_scale;
	int receptive_field;
	int max_frames;
};

size_t na_oracle_wavenet_num_weights(int num_arrays, const na_oracle_wn_array_cfg* cfgs)
{
	size_t n = 0;
	for (int i = 0; i < num_arrays; i++) n += array_num_weights(&cfgs[i]);
	return n + 1; /* headScale, :698 */
}

na_oracle_wavenet* na_oracle_wavenet_create(int num_arrays, const na_oracle_wn_array_cfg* cfgs, const float* weights,
	size_t num_weights, int math_mode)
{
	if (num_arrays < 1 || num_arrays > NA_ORACLE_MAX_ARRAYS) return NULL;
	if (na_oracle_wavenet_num_weights(num_arrays, cfgs) != num_weights) return NULL; /* :704-709 */
	for (int i = 1; i < num_arrays; i++)
		if (cfgs[i - 1].head_size != cfgs[i].channels || cfgs[i].input_size != cfgs[i - 1].channels)
			return NULL; /* head accumulation happens in place into the previous array's headOutputs, :785-789 */
	na_oracle_wavenet* m = (na_oracle_wavenet*)calloc(1, sizeof(*m));
	m->num_arrays = num_arrays;
	m->max_frames = MAXF;
	int alloc_num = 0;
	for (int i = 0; i < num_arrays; i++) {
		array_init(&m->arrays[i], &cfgs[i], math_mode);
		m->receptive_field += m->arrays[i].receptive_field;        /* :676-684 */
		alloc_num = array_alloc_buffers(&m->arrays[i], alloc_num);
	}
	m->head_array = (float*)calloc((size_t)MAXF * cfgs[0].channels, sizeof(float));
	const float* it = weights;
	for (int i = 0; i < num_arrays; i++) it = array_set_weights(&m->arrays[i], it); /* :711-716 */
	m->head_scale = *(it++);                                                         /* :718 */
	return m;
}

void na_oracle_wavenet_free(na_oracle_wavenet* m)
{
	if (!m) return;
	for (int i = 0; i < m->num_arrays; i++) array_free(&m->arrays[i]);
	free(m->head_array);
	free(m);
}

int na_oracle_wavenet_receptive_field(const na_oracle_wavenet* m) { return m->receptive_field; }

void na_oracle_wavenet_set_max_frames(na_oracle_wavenet* m, int max_frames)
{
	if (max_frames < 1) max_frames = 1;
	if (max_frames > MAXF) max_frames = MAXF;
	m->max_frames = max_frames;
}

void na_oracle_wavenet_reset(na_oracle_wavenet* m)
{
	int alloc_num = 0;
	for (int i = 0; i < m->num_arrays; i++) alloc_num = array_alloc_buffers(&m->arrays[i], alloc_num);
}

/* :746-766 */
void na_oracle_wavenet_prewarm(na_oracle_wavenet* m)
{
	m->condition[0] = 0.0f;
	memset(m->head_array, 0, (size_t)MAXF * m->arrays[0].cfg.channels * sizeof(float));
	for (int i = 0; i < m->num_arrays; i++) {
		if (i == 0)
			array_prewarm(&m->arrays[0], m->condition, m->condition, m->head_array);
		else
			array_prewarm(&m->arrays[i], m->arrays[i - 1].array_outputs, m->condition, m->arrays[i - 1].head_outputs);
	}
}

/* :768-799, F <= MAXF */
static void wavenet_process_chunk(na_oracle_wavenet* m, const float* in, float* out, int frames)
{
	memcpy(m->condition, in, (size_t)frames * sizeof(float));
	memset(m->head_array, 0, (size_t)MAXF * m->arrays[0].cfg.channels * sizeof(float));
	const int last = m->num_arrays - 1;
	for (int i = 0; i < m->num_arrays; i++) {
		if (i == 0)
			array_process(&m->arrays[0], m->condition, m->condition, m->head_array, frames, 1);
		else
			array_process(&m->arrays[i], m->arrays[i - 1].array_outputs, m->condition, m->arrays[i - 1].head_outputs,
				frames, i != last);
	}
	const float* final_head = m->arrays[last].head_outputs;
	const int hs = m->arrays[last].cfg.head_size;
	for (int f = 0; f < frames; f++) out[f] = m->head_scale * final_head[(size_t)f * hs];
}

/* InternalModel.h:104-117 */
void na_oracle_wavenet_process(na_oracle_wavenet* m, const float* in, float* out, size_t num_samples)
{
	size_t offset = 0;
	float tmp[MAXF];
	while (num_samples > 0) {
		const int n = (int)(num_samples < (size_t)m->max_frames ? num_samples : (size_t)m->max_frames);
		memcpy(tmp, in + offset, (size_t)n * sizeof(float)); /* in-place safe like WaveNet.h:770 */
		wavenet_process_chunk(m, tmp, out + offset, n);
		offset += (size_t)n;
		num_samples -= (size_t)n;
	}
}

/* ---------------------------------------------------------------- LSTM (LSTM.h) */

typedef struct {
	int input_size, hidden;
	float* w;     /* row-major [4H][I+H]  (inputHiddenWeights(i,j)), LSTM.h:27 */
	float* bias;  /* [4H] */
	float* state; /* [I+H]: input then hidden, LSTM.h:29,37 */
	float* gates; /* [4H] */
	float* cell;  /* [H] */
} lstm_layer;

struct na_oracle_lstm {
	int num_layers, hidden, math_mode;
	lstm_layer* layers;
	float* head_w;
	float head_b;
};

static void lstm_layer_init(lstm_layer* l, int input_size, int hidden)
{
	l->input_size = input_size; l->hidden = hidden;
	l->w = (float*)calloc((size_t)4 * hidden * (input_size + hidden), sizeof(float));
	l->bias = (float*)calloc((size_t)4 * hidden, sizeof(float));
	l->state = (float*)calloc((size_t)(input_size + hidden), sizeof(float));
	l->gates = (float*)calloc((size_t)4 * hidden, sizeof(float));
	l->cell = (float*)calloc((size_t)hidden, sizeof(float));
}

static void lstm_layer_free(lstm_layer* l)
{
	free(l->w); free(l->bias); free(l->state); free(l->gates); free(l->cell);
}

/* LSTM.h:87-100; gate row blocks i,f,g,o at 0,H,2H,3H (:33-36) */
static void lstm_layer_process(lstm_layer* l, const float* input, int math_mode)
{
	const int I = l->input_size, H = l->hidden, W = I + H;
	for (int i = 0; i < I; i++) l->state[i] = input[i];
	for (int r = 0; r < 4 * H; r++) {
		const float* row = l->w + (size_t)r * W;
		float acc = 0.0f;
		for (int j = 0; j < W; j++) acc += row[j] * l->state[j];
		l->gates[r] = acc + l->bias[r];
	}
	if (math_mode == NA_ORACLE_MATH_FAST) {
		for (int i = 0; i < H; i++)
			l->cell[i] = (na_oracle_fast_sigmoid(l->gates[i + H]) * l->cell[i])
				+ (na_oracle_fast_sigmoid(l->gates[i]) * na_oracle_fast_tanh(l->gates[i + 2 * H]));
		for (int i = 0; i < H; i++)
			l->state[i + I] = na_oracle_fast_sigmoid(l->gates[i + 3 * H]) * na_oracle_fast_tanh(l->cell[i]);
	} else {
		for (int i = 0; i < H; i++)
			l->cell[i] = (std_sigmoid(l->gates[i + H]) * l->cell[i]) + (std_sigmoid(l->gates[i]) * std_tanh(l->gates[i + 2 * H]));
		for (int i = 0; i < H; i++)
			l->state[i + I] = std_sigmoid(l->gates[i + 3 * H]) * std_tanh(l->cell[i]);
	}
}

static na_oracle_lstm* lstm_alloc(int num_layers, int hidden, int math_mode)
{
	na_oracle_lstm* m = (na_oracle_lstm*)calloc(1, sizeof(*m));
	m->num_layers = num_layers; m->hidden = hidden; m->math_mode = math_mode;
	m->layers = (lstm_layer*)calloc((size_t)num_layers, sizeof(lstm_layer));
	for (int l = 0; l < num_layers; l++) lstm_layer_init(&m->layers[l], l == 0 ? 1 : hidden, hidden);
	m->head_w = (float*)calloc((size_t)hidden, sizeof(float));
	return m;
}

/* LSTM.h:42-56 (layer) + :130-147 (model) */
na_oracle_lstm* na_oracle_lstm_create_nam(int num_layers, int hidden, const float* weights, size_t num_weights,
	int math_mode)
{
	size_t expect = 0;
	for (int l = 0; l < num_layers; l++) {
		const int I = l == 0 ? 1 : hidden;
		expect += (size_t)4 * hidden * (I + hidden) + 4 * hidden + 2 * hidden;
	}
	expect += (size_t)hidden + 1;
	if (expect != num_weights) return NULL;
	na_oracle_lstm* m = lstm_alloc(num_layers, hidden, math_mode);
	const float* it = weights;
	for (int l = 0; l < num_layers; l++) {
		lstm_layer* L = &m->layers[l];
		const int W = L->input_size + hidden;
		for (int i = 0; i < 4 * hidden; i++)
			for (int j = 0; j < W; j++) L->w[(size_t)i * W + j] = *(it++);
		for (int i = 0; i < 4 * hidden; i++) L->bias[i] = *(it++);
		for (int i = 0; i < hidden; i++) L->state[i + L->input_size] = *(it++); /* initial hidden */
		for (int i = 0; i < hidden; i++) L->cell[i] = *(it++);                  /* initial cell */
	}
	for (int i = 0; i < hidden; i++) m->head_w[i] = *(it++);
	m->head_b = *(it++);
	return m;
}

/* LSTM.h:58-85: W(:, j) = kernel[j][:] (j < I), W(:, I+j) = recurrent[j][:]; bias; zero state */
na_oracle_lstm* na_oracle_lstm_create_keras(int num_layers, int hidden, const float* const* kernels,
	const float* const* recurrents, const float* const* biases, const float* head_weights, float head_bias, int math_mode)
{
	na_oracle_lstm* m = lstm_alloc(num_layers, hidden, math_mode);
	for (int l = 0; l < num_layers; l++) {
		lstm_layer* L = &m->layers[l];
		const int I = L->input_size, W = I + hidden, R = 4 * hidden;
		for (int j = 0; j < I; j++)
			for (int i = 0; i < R; i++) L->w[(size_t)i * W + j] = kernels[l][(size_t)j * R + i];
		for (int j = 0; j < hidden; j++)
			for (int i = 0; i < R; i++) L->w[(size_t)i * W + I + j] = recurrents[l][(size_t)j * R + i];
		for (int i = 0; i < R; i++) L->bias[i] = biases[l][i];
	}
	for (int i = 0; i < hidden; i++) m->head_w[i] = head_weights[i];
	m->head_b = head_bias;
	return m;
}

void na_oracle_lstm_free(na_oracle_lstm* m)
{
	if (!m) return;
	for (int l = 0; l < m->num_layers; l++) lstm_layer_free(&m->layers[l]);
	free(m->layers); free(m->head_w); free(m);
}

/* LSTM.h:164-191 */
void na_oracle_lstm_process(na_oracle_lstm* m, const float* in, float* out, size_t num_samples)
{
	const int H = m->hidden;
	for (size_t s = 0; s < num_samples; s++) {
		lstm_layer_process(&m->layers[0], in + s, m->math_mode);
		for (int l = 1; l < m->num_layers; l++)
			lstm_layer_process(&m->layers[l], m->layers[l - 1].state + m->layers[l - 1].input_size, m->math_mode);
		const lstm_layer* last = &m->layers[m->num_layers - 1];
		const float* h = last->state + last->input_size;
		float acc = 0.0f;
		for (int i = 0; i < H; i++) acc += m->head_w[i] * h[i];
		out[s] = acc + m->head_b;
	}
}

/* InternalModel.h:368-371 -> NeuralModelImpl.h:96-109: Prewarm(2048, 64) */
void na_oracle_lstm_prewarm(na_oracle_lstm* m)
{
	float in[64], out[64];
	memset(in, 0, sizeof(in));
	for (int b = 0; b < 2048 / 64; b++) na_oracle_lstm_process(m, in, out, 64);
}

/* ---------------------------------------------------------------- keras GRU (config 4; third-party arithmetic)
 *
 * PARITY UNPINNED.  In the reference a keras "gru" model is not evaluated by NeuralAudio's own code but by RTNeural
 * (deps/RTNeural, an empty submodule here; call sites NeuralAudio/NeuralModel.cpp:565-572 -> RTNeuralModel.h:300,
 * 417-429: json_parser::parseJson<float, FastMathsProvider>, model->forward per sample, 2048-zero prewarm).  This is a
 * restatement of the PUBLISHED algorithm of RTNeural's GRULayer / Keras GRU(reset_after=True):
 *     z = sigma(W_z x + U_z h + b_z0 + b_z1)            kernel [I][3H] and recurrent [H][3H] hold the gate column
 *     r = sigma(W_r x + U_r h + b_r0 + b_r1)            blocks z | r | c, bias [2][3H] = input row, recurrent row
 *     c = tanh (W_c x + b_c0 + r o (U_c h + b_c1))
 *     h = (1 - z) o c + z o h                           zero initial state
 * with the reference's FastMathsProvider (RTNeuralModel.h:10-31): tanh = Eigen array tanh (an accurate float tanh; libm
 * tanhf here), sigmoid(x) = (tanh(x/2) + 1) / 2.  tests/test_oracle.py cross-checks it against torch.nn.GRU.
 */
struct na_oracle_gru {
	int num_layers, hidden;
	float** wi;   /* per layer: row-major [3H][I]  */
	float** wh;   /* per layer: row-major [3H][H]  */
	float** bi;   /* per layer: [3H] input bias    */
	float** bh;   /* per layer: [3H] recurrent bias */
	float** h;    /* per layer: [H] */
	float* head_w; float head_b;
};

static float gru_sigmoid(float x) { return (tanhf(x / 2.0f) + 1.0f) / 2.0f; }

na_oracle_gru* na_oracle_gru_create_keras(int num_layers, int hidden, const float* const* kernels, const float* const* recurrents,
	const float* const* biases, const float* head_weights, float head_bias)
{
	na_oracle_gru* m = (na_oracle_gru*)calloc(1, sizeof(*m));
	const int H = hidden, R = 3 * hidden;
	m->num_layers = num_layers; m->hidden = hidden;
	m->wi = (float**)calloc((size_t)num_layers, sizeof(float*)); m->wh = (float**)calloc((size_t)num_layers, sizeof(float*));
	m->bi = (float**)calloc((size_t)num_layers, sizeof(float*)); m->bh = (float**)calloc((size_t)num_layers, sizeof(float*));
	m->h = (float**)calloc((size_t)num_layers, sizeof(float*));
	for (int l = 0; l < num_layers; l++) {
		const int I = (l == 0) ? 1 : H;
		m->wi[l] = (float*)calloc((size_t)R * I, sizeof(float)); m->wh[l] = (float*)calloc((size_t)R * H, sizeof(float));
		m->bi[l] = (float*)calloc((size_t)R, sizeof(float)); m->bh[l] = (float*)calloc((size_t)R, sizeof(float));
		m->h[l] = (float*)calloc((size_t)H, sizeof(float));
		for (int j = 0; j < I; j++) for (int i = 0; i < R; i++) m->wi[l][(size_t)i * I + j] = kernels[l][(size_t)j * R + i];
		for (int j = 0; j < H; j++) for (int i = 0; i < R; i++) m->wh[l][(size_t)i * H + j] = recurrents[l][(size_t)j * R + i];
		for (int i = 0; i < R; i++) { m->bi[l][i] = biases[l][i]; m->bh[l][i] = biases[l][R + i]; }
	}
	m->head_w = (float*)calloc((size_t)H, sizeof(float));
	for (int i = 0; i < H; i++) m->head_w[i] = head_weights[i];
	m->head_b = head_bias;
	return m;
}

void na_oracle_gru_free(na_oracle_gru* m)
{
	if (!m) return;
	for (int l = 0; l < m->num_layers; l++) { free(m->wi[l]); free(m->wh[l]); free(m->bi[l]); free(m->bh[l]); free(m->h[l]); }
	free(m->wi); free(m->wh); free(m->bi); free(m->bh); free(m->h); free(m->head_w); free(m);
}

static void gru_layer_step(na_oracle_gru* m, int l, const float* x)
{
	const int H = m->hidden, I = (l == 0) ? 1 : H;
	float ai[3 * 256], ah[3 * 256]; /* H <= 256 */
	float* h = m->h[l];
	for (int r = 0; r < 3 * H; r++) {
		float a = 0.0f, b = 0.0f;
		for (int k = 0; k < I; k++) a += m->wi[l][(size_t)r * I + k] * x[k];
		for (int k = 0; k < H; k++) b += m->wh[l][(size_t)r * H + k] * h[k];
		ai[r] = a + m->bi[l][r];
		ah[r] = b + m->bh[l][r];
	}
	for (int u = 0; u < H; u++) {
		const float z = gru_sigmoid(ai[u] + ah[u]);
		const float rg = gru_sigmoid(ai[H + u] + ah[H + u]);
		const float c = tanhf(ai[2 * H + u] + rg * ah[2 * H + u]);
		ai[u] = (1.0f - z) * c + z * h[u]; /* new h, parked until every unit has read the old one */
	}
	for (int u = 0; u < H; u++) h[u] = ai[u];
}

void na_oracle_gru_process(na_oracle_gru* m, const float* in, float* out, size_t num_samples)
{
	const int H = m->hidden;
	for (size_t s = 0; s < num_samples; s++) {
		gru_layer_step(m, 0, in + s);
		for (int l = 1; l < m->num_layers; l++) gru_layer_step(m, l, m->h[l - 1]);
		const float* h = m->h[m->num_layers - 1];
		float acc = 0.0f;
		for (int i = 0; i < H; i++) acc += m->head_w[i] * h[i];
		out[s] = acc + m->head_b;
	}
}

/* RTNeuralModel.h:423-429 */
void na_oracle_gru_prewarm(na_oracle_gru* m)
{
	float in[64], out[64];
	memset(in, 0, sizeof(in));
	for (int b = 0; b < 2048 / 64; b++) na_oracle_gru_process(m, in, out, 64);
}

/* ---------------------------------------------------------------- ModelTest-style timing (cpu_baseline only) */

typedef struct {
	int is_lstm;
	int num_arrays; const na_oracle_wn_array_cfg* cfgs;
	int num_layers, hidden;
	const float* weights; size_t num_weights;
	int block_size, num_blocks;
	/* is_lstm == 2: keras GRU (na_oracle_gru_create_keras arguments) */
	const float* const* gru_kernels; const float* const* gru_recurrents; const float* const* gru_biases;
	const float* gru_head_w; float gru_head_b;
	pthread_barrier_t* barrier;
	double seconds;
} bench_arg;

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Utils/ModelTest/ModelTest.cpp:59-79 BenchModel: blocks of zeros after prewarm */
static void* bench_thread(void* p)
{
	bench_arg* a = (bench_arg*)p;
	float* in = (float*)calloc((size_t)a->block_size, sizeof(float));
	float* out = (float*)calloc((size_t)a->block_size, sizeof(float));
	na_oracle_wavenet* wn = NULL;
	na_oracle_lstm* ls = NULL;
	na_oracle_gru* gr = NULL;
	if (a->is_lstm == 2) {
		gr = na_oracle_gru_create_keras(a->num_layers, a->hidden, a->gru_kernels, a->gru_recurrents, a->gru_biases, a->gru_head_w, a->gru_head_b);
		if (gr) na_oracle_gru_prewarm(gr);
	} else if (a->is_lstm) {
		ls = na_oracle_lstm_create_nam(a->num_layers, a->hidden, a->weights, a->num_weights, NA_ORACLE_MATH_FAST);
		if (ls) na_oracle_lstm_prewarm(ls);
	} else {
		wn = na_oracle_wavenet_create(a->num_arrays, a->cfgs, a->weights, a->num_weights, NA_ORACLE_MATH_FAST);
		if (wn) na_oracle_wavenet_prewarm(wn);
	}
	pthread_barrier_wait(a->barrier);
	const double t0 = now_s();
	for (int b = 0; b < a->num_blocks; b++) {
		if (wn) na_oracle_wavenet_process(wn, in, out, (size_t)a->block_size);
		else if (ls) na_oracle_lstm_process(ls, in, out, (size_t)a->block_size);
		else if (gr) na_oracle_gru_process(gr, in, out, (size_t)a->block_size);
	}
	a->seconds = now_s() - t0;
	pthread_barrier_wait(a->barrier);
	na_oracle_wavenet_free(wn);
	na_oracle_lstm_free(ls);
	na_oracle_gru_free(gr);
	free(in); free(out);
	return NULL;
}

static double run_bench(bench_arg proto, int threads)
{
	if (threads < 1) threads = 1;
	pthread_barrier_t barrier;
	pthread_barrier_init(&barrier, NULL, (unsigned)threads + 1);
	pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
	bench_arg* args = (bench_arg*)calloc((size_t)threads, sizeof(bench_arg));
	for (int t = 0; t < threads; t++) {
		args[t] = proto;
		args[t].barrier = &barrier;
		pthread_create(&tids[t], NULL, bench_thread, &args[t]);
	}
	pthread_barrier_wait(&barrier);
	const double t0 = now_s();
	pthread_barrier_wait(&barrier);
	const double wall = now_s() - t0;
	for (int t = 0; t < threads; t++) pthread_join(tids[t], NULL);
	pthread_barrier_destroy(&barrier);
	free(tids); free(args);
	return wall;
}

double na_oracle_wavenet_bench(int num_arrays, const na_oracle_wn_array_cfg* cfgs, const float* weights,
	size_t num_weights, int block_size, int num_blocks, int threads)
{
	bench_arg a;
	memset(&a, 0, sizeof(a));
	a.is_lstm = 0; a.num_arrays = num_arrays; a.cfgs = cfgs; a.weights = weights; a.num_weights = num_weights;
	a.block_size = block_size; a.num_blocks = num_blocks;
	return run_bench(a, threads);
}

double na_oracle_lstm_bench(int num_layers, int hidden_size, const float* weights, size_t num_weights, int block_size,
	int num_blocks, int threads)
{
	bench_arg a;
	memset(&a, 0, sizeof(a));
	a.is_lstm = 1; a.num_layers = num_layers; a.hidden = hidden_size; a.weights = weights; a.num_weights = num_weights;
	a.block_size = block_size; a.num_blocks = num_blocks;
	return run_bench(a, threads);
}

double na_oracle_gru_bench(int num_layers, int hidden_size, const float* const* kernels, const float* const* recurrents,
	const float* const* biases, const float* head_weights, float head_bias, int block_size, int num_blocks, int threads)
{
	bench_arg a;
	memset(&a, 0, sizeof(a));
	a.is_lstm = 2; a.num_layers = num_layers; a.hidden = hidden_size;
	a.gru_kernels = kernels; a.gru_recurrents = recurrents; a.gru_biases = biases; a.gru_head_w = head_weights; a.gru_head_b = head_bias;
	a.block_size = block_size; a.num_blocks = num_blocks;
	return run_bench(a, threads);
}
