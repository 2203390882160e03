/*
 * enet_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked, imported or called by the product path).
 *
 * CPU restatement, in plain C, of the arithmetic on the reference's pool-scoring hot path
 * (alfrunesiq/SemanticSegmentationActiveLearning).  The reference itself is pure Python on
 * TensorFlow 1.13.2; the kernels it calls live in the un-vendored tensorflow-gpu wheel, which is
 * not present here, so every function below restates the *documented* TF-1.13 op semantics at the
 * reference call site it cites.
 *
 * PARITY STATUS: "parity unpinned" against real TensorFlow (the reference ships no golden
 * vectors, no known-answer tests and no fixtures for this path).  What pins
 * this file: (1) the reference's only test invariant, pool->unpool->pool identity
 * (models/util/test_xops.py:6-21), (2) agreement to <=1e-5 with an independent torch-CPU
 * restatement (oracle/torch_restatement.py), (3) committed self-generated fixtures (tests/golden).
 *
 * Accumulation order is FIXED so that the HIP path can be compared bit-for-bit:
 *   conv / conv-transpose output element = single fp32 fmaf chain, acc starts at +0, taps visited
 *   in (kh ascending, kw ascending, ci ascending) order; taps that fall in the SAME zero padding
 *   are skipped (== adding an exact zero product).
 *   batch-norm (inference) is applied in folded form  y = fmaf(x, s, t),
 *   s = gamma / sqrtf(var + 1e-3f), t = fmaf(-mean, s, beta)        (orc_bn_fold).
 *
 * Layouts: activations NHWC fp32; conv kernels HWIO; transposed-conv kernels HW-O-I (TF layout).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* tf.nn.fused_batch_norm(is_training=False), default epsilon 1e-3
 * (reference models/util/extra_ops.py:181-184), folded to one fma per element. */
ORC_API void orc_bn_fold(const float *mean, const float *var, const float *gamma, const float *beta,
                         int C, float *scale, float *shift)
{
    for (int c = 0; c < C; ++c) {
        float s = gamma[c] / sqrtf(var[c] + 1e-3f);
        scale[c] = s;
        shift[c] = fmaf(-mean[c], s, beta[c]);
    }
}

static inline void same_pad(int in, int k, int stride, int dil, int *out, int *pad_before)
{
    /* TF "SAME": out = ceil(in/stride); pad_total = max((out-1)*stride + (k-1)*dil + 1 - in, 0);
     * pad_before = pad_total / 2, remainder goes after. */
    int o = (in + stride - 1) / stride;
    int tot = (o - 1) * stride + (k - 1) * dil + 1 - in;
    if (tot < 0) tot = 0;
    *out = o;
    *pad_before = tot / 2;
}

/* tf.nn.conv2d (cross-correlation), NHWC x HWIO, padding="SAME"
 * call sites: reference models/enet/enet_modules.py:205,538,554,559,565,581,880,895,911,1236,1267,1285 */
ORC_API void orc_conv2d_same(const float *x, int N, int H, int W, int Cin,
                             const float *w, int KH, int KW, int Cout,
                             int stride, int dil, float *y)
{
    int Ho, Wo, pt, pl;
    same_pad(H, KH, stride, dil, &Ho, &pt);
    same_pad(W, KW, stride, dil, &Wo, &pl);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy) {
            float *acc = (float *)malloc(sizeof(float) * (size_t)Cout);
            for (int ox = 0; ox < Wo; ++ox) {
                for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
                for (int kh = 0; kh < KH; ++kh) {
                    int iy = oy * stride - pt + kh * dil;
                    if (iy < 0 || iy >= H) continue;
                    for (int kw = 0; kw < KW; ++kw) {
                        int ix = ox * stride - pl + kw * dil;
                        if (ix < 0 || ix >= W) continue;
                        const float *xp = x + (((size_t)n * H + iy) * W + ix) * Cin;
                        const float *wp = w + ((size_t)(kh * KW + kw) * Cin) * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            float xv = xp[ci];
                            const float *wr = wp + (size_t)ci * Cout;
                            for (int co = 0; co < Cout; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                        }
                    }
                }
                float *yp = y + (((size_t)n * Ho + oy) * Wo + ox) * Cout;
                for (int co = 0; co < Cout; ++co) yp[co] = acc[co];
            }
            free(acc);
        }
}

/* tf.nn.conv2d_transpose, 3x3 kernel [3,3,Cout,Cin], strides 2, padding="SAME", output 2H x 2W
 * (reference enet_modules.py:1251-1255, 1376-1380).  It is the input-gradient of a stride-2 SAME
 * conv whose forward padding on an even size is (0 before, 1 after):
 *     out[2i+kh, 2j+kw, o] += in[i, j, c] * W[kh, kw, o, c],  rows/cols with index 2H / 2W dropped.
 * Gather form per output element, taps in (kh, kw, c) ascending order. */
ORC_API void orc_conv2d_transpose_3x3_s2(const float *x, int N, int H, int W, int Cin,
                                         const float *w, int Cout, float *y)
{
    int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy) {
            for (int ox = 0; ox < Wo; ++ox) {
                float *yp = y + (((size_t)n * Ho + oy) * Wo + ox) * Cout;
                for (int co = 0; co < Cout; ++co) {
                    float acc = 0.0f;
                    for (int kh = 0; kh < 3; ++kh) {
                        int ty = oy - kh;
                        if (ty < 0 || (ty & 1)) continue;
                        int iy = ty >> 1;
                        if (iy >= H) continue;
                        for (int kw = 0; kw < 3; ++kw) {
                            int tx = ox - kw;
                            if (tx < 0 || (tx & 1)) continue;
                            int ix = tx >> 1;
                            if (ix >= W) continue;
                            const float *xp = x + (((size_t)n * H + iy) * W + ix) * Cin;
                            const float *wp = w + ((size_t)((kh * 3 + kw) * Cout + co)) * Cin;
                            for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
                        }
                    }
                    yp[co] = acc;
                }
            }
        }
}

/* folded batch-norm followed by an optional PReLU.
 * PReLU = relu(x) - alpha*relu(-x)  (reference extra_ops.py:21-26)  ==  x >= 0 ? x : alpha*x exactly. */
ORC_API void orc_affine_prelu(const float *x, size_t pixels, int C, const float *scale,
                              const float *shift, const float *alpha /* may be NULL */, float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < pixels; ++p)
        for (int c = 0; c < C; ++c) {
            float v = x[p * C + c];
            if (scale) v = fmaf(v, scale[c], shift[c]);
            if (alpha) v = (v >= 0.0f) ? v : alpha[c] * v;
            y[p * C + c] = v;
        }
}

/* tf.math.add(conv_out, residual) then PReLU (reference enet_modules.py:596-598, 935-937, 1290-1291).
 * The residual may have fewer channels than the main branch (tf.pad zero channels at the END,
 * enet_modules.py:706-707,931-933): missing channels add an exact zero. */
ORC_API void orc_add_prelu(const float *a, const float *res, size_t pixels, int C, int Cres,
                           const float *alpha, float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < pixels; ++p)
        for (int c = 0; c < C; ++c) {
            float v = a[p * C + c] + (c < Cres ? res[p * (size_t)Cres + c] : 0.0f);
            y[p * C + c] = (v >= 0.0f) ? v : alpha[c] * v;
        }
}

/* tf.nn.max_pool / max_pool_with_argmax, ksize 2x2, strides 2, SAME on even H,W (no padding)
 * (reference enet_modules.py:212, 927-929).  Strict '>' scan in (y, x) window order => the first
 * maximum wins ties.  Index convention:
 *   include_batch == 0 : (y*W + x)*C + c              (TF<=1.13 GPU kernel, per image)
 *   include_batch != 0 : ((b*H + y)*W + x)*C + c      (TF<=1.13 CPU kernel) */
ORC_API void orc_maxpool2x2_argmax(const float *x, int N, int H, int W, int C, float *y,
                                   int64_t *idx /* may be NULL */, int include_batch)
{
    int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox)
                for (int c = 0; c < C; ++c) {
                    float best = -FLT_MAX;
                    int64_t bi = -1;
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            int iy = 2 * oy + dy, ix = 2 * ox + dx;
                            float v = x[(((size_t)n * H + iy) * W + ix) * C + c];
                            if (v > best) {
                                best = v;
                                bi = ((int64_t)iy * W + ix) * C + c;
                            }
                        }
                    size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + c;
                    y[o] = best;
                    if (idx) idx[o] = bi + (include_batch ? (int64_t)n * H * W * C : 0);
                }
}

/* xops.unpool_2d (reference extra_ops.py:28-86): zeros([N,2H,2W,C]).flat[idx (+ b*2H*2W*C)] = in.flat
 * via tf.scatter_nd.  idx_has_batch != 0 means idx already contains the batch term (CPU branch,
 * extra_ops.py:80-81); otherwise the per-image offset is added here (GPU branch, :73-79). */
ORC_API void orc_unpool2d(const float *x, const int64_t *idx, int N, int H, int W, int C,
                          int idx_has_batch, float *y)
{
    size_t img_out = (size_t)4 * H * W * C, img_in = (size_t)H * W * C;
    memset(y, 0, sizeof(float) * img_out * N);
    for (int n = 0; n < N; ++n)
        for (size_t i = 0; i < img_in; ++i) {
            int64_t k = idx[n * img_in + i] + (idx_has_batch ? 0 : (int64_t)n * (int64_t)img_out);
            y[k] = x[n * img_in + i];
        }
}

/* Acquisition score (reference active_learning.py:39-40, 234-263):
 *   p = softmax(logits, -1)                       (exp(x - max) / sum)
 *   measure 0 "entropy"   : conf = 1 - (-sum_k p*log(p + FLT_MIN)) / log((float)K)
 *   measure 1 "margin"    : conf = top1(p) - top2(p)
 *   measure 2 "confidence": conf = max_k p
 *   per image: mean over H,W of (double)conf       (tf.reduce_mean on float64, :261-263)
 *   label = (uint8) argmax_k logits, first maximum wins (:234-236)
 * All per-pixel arithmetic is fp32 like the TF graph. */
ORC_API int orc_score(const float *logits, int N, int H, int W, int K, int measure,
                      float *conf_out /* [N,H,W] or NULL */, uint8_t *label_out /* or NULL */,
                      double *mean_out /* [N] */)
{
    if (measure < 0 || measure > 2) return -1; /* NotImplementedError, active_learning.py:259-260 */
    const float eps = FLT_MIN; /* np.finfo(np.float32).tiny */
    const float log_base = logf((float)K);
    size_t P = (size_t)H * W;
    for (int n = 0; n < N; ++n) {
        double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
        for (size_t i = 0; i < P; ++i) {
            const float *l = logits + ((size_t)n * P + i) * K;
            float m = l[0];
            int am = 0;
            for (int k = 1; k < K; ++k)
                if (l[k] > m) { m = l[k]; am = k; }
            float sum = 0.0f;
            float e[256];
            for (int k = 0; k < K; ++k) { e[k] = expf(l[k] - m); sum += e[k]; }
            float conf;
            if (measure == 0) {
                float ent = 0.0f;
                for (int k = 0; k < K; ++k) {
                    float p = e[k] / sum;
                    ent += -p * logf(p + eps);
                }
                conf = 1.0f - ent / log_base;
            } else if (measure == 1) {
                float v0 = -1.0f, v1 = -1.0f;
                for (int k = 0; k < K; ++k) {
                    float p = e[k] / sum;
                    if (p > v0) { v1 = v0; v0 = p; }
                    else if (p > v1) { v1 = p; }
                }
                conf = v0 - v1;
            } else {
                float v0 = 0.0f;
                for (int k = 0; k < K; ++k) { float p = e[k] / sum; if (p > v0) v0 = p; }
                conf = v0;
            }
            if (conf_out) conf_out[(size_t)n * P + i] = conf;
            if (label_out) label_out[(size_t)n * P + i] = (uint8_t)am;
            total += (double)conf;
        }
        mean_out[n] = total / (double)P;
    }
    return 0;
}

/* Initial block helper: concat([conv_out(Cc), pool_out(Cp)], -1)  (reference enet_modules.py:214-215). */
ORC_API void orc_concat2(const float *a, int Ca, const float *b, int Cb, size_t pixels, float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < pixels; ++p) {
        memcpy(y + p * (Ca + Cb), a + p * Ca, sizeof(float) * Ca);
        memcpy(y + p * (Ca + Cb) + Ca, b + p * Cb, sizeof(float) * Cb);
    }
}

/* tensortools.losses.masked_softmax_cross_entropy forward (reference tensortools/losses.py:3-74).
 * fp32 per-pixel cross entropy with label smoothing, mask, optional class weighting (weight > 1),
 * fp32 sum over the batch axis, float64 over the spatial axes, divided by (double)(float)sum(mask). */
ORC_API double orc_masked_softmax_xent(const float *logits, const uint8_t *labels, const float *mask,
                                       int N, int H, int W, int K, float weight, float label_smoothing)
{
    const float on_value = 1.0f - label_smoothing, off_value = label_smoothing / ((float)K - 1.0f);
    size_t P = (size_t)H * W;
    double total = 0.0;
    float msum = 0.0f;
    for (size_t i = 0; i < (size_t)N * P; ++i) msum += mask[i];
    for (size_t pos = 0; pos < P; ++pos) {
        float bsum = 0.0f;
        for (int n = 0; n < N; ++n) {
            const float *l = logits + ((size_t)n * P + pos) * K;
            int lab = labels[(size_t)n * P + pos];
            float m = l[0];
            for (int k = 1; k < K; ++k) if (l[k] > m) m = l[k];
            float S = 0.0f;
            for (int k = 0; k < K; ++k) S += expf(l[k] - m);
            float logS = logf(S), ce = 0.0f, pc = 0.0f;
            for (int k = 0; k < K; ++k) {
                float yk = (k == lab) ? on_value : off_value;
                ce += yk * (logS - (l[k] - m));
                pc += yk * (expf(l[k] - m) / S);
            }
            ce *= mask[(size_t)n * P + pos];
            if (weight > 1.0f) ce *= 1.0f / logf(weight + (1.718281828459045f - weight) * pc);
            bsum += ce;
        }
        total += (double)bsum;
    }
    return total / (double)msum;
}
