/*
 * icnet_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked, imported or called by the product path).
 *
 * Plain-C restatement of the operators ICNET_SPEC.md adds to the ones in enet_oracle.c.  The reference's
 * models/icnet/icnet.py:1-7 is an empty class: there is NO reference behaviour for these -- PARITY STATUS
 * "unpinned and undefined".  Each function restates the TF-1.13 op ICNET_SPEC.md names, with a fixed
 * evaluation order so that the HIP kernels can be compared bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* y = [relu]( fmaf(x, scale, shift) [+ res] )  -- folded batch-norm (extra_ops.py:181-184; ICNET_SPEC "batch-norm"),
 * optional shortcut / fusion add (ICNET_SPEC sections 1, 4), optional tf.nn.relu.  scale may be NULL (then
 * y = x + shift: the conv6_cls bias when shift != NULL). */
ORC_API void orc_affine_add_relu(const float *x, size_t pixels, int C, const float *scale, const float *shift,
                                 const float *res /* may be NULL */, int relu, float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < pixels; ++p)
        for (int c = 0; c < C; ++c) {
            float v = x[p * C + c];
            if (scale) v = fmaf(v, scale[c], shift[c]);
            else if (shift) v = v + shift[c];
            if (res) v = v + res[p * C + c];
            if (relu) v = v > 0.0f ? v : 0.0f;
            y[p * C + c] = v;
        }
}

/* tf.nn.max_pool(ksize 3x3, strides 2, "SAME") (ICNET_SPEC pool1_3x3_s2): out = ceil(in/2),
 * pad_total = max((out-1)*2 + 3 - in, 0), pad_before = pad_total/2; the maximum runs over in-image taps only. */
ORC_API void orc_maxpool3x3_s2_same(const float *x, int N, int H, int W, int C, float *y)
{
    int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    int th = (Ho - 1) * 2 + 3 - H, tw = (Wo - 1) * 2 + 3 - W;
    if (th < 0) th = 0;
    if (tw < 0) tw = 0;
    int pt = th / 2, pl = tw / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox)
                for (int c = 0; c < C; ++c) {
                    float best = -FLT_MAX;
                    for (int dy = 0; dy < 3; ++dy) {
                        int iy = oy * 2 - pt + dy;
                        if (iy < 0 || iy >= H) continue;
                        for (int dx = 0; dx < 3; ++dx) {
                            int ix = ox * 2 - pl + dx;
                            if (ix < 0 || ix >= W) continue;
                            float v = x[(((size_t)n * H + iy) * W + ix) * C + c];
                            if (v > best) best = v;
                        }
                    }
                    y[(((size_t)n * Ho + oy) * Wo + ox) * C + c] = best;
                }
}

/* tf.image.resize_bilinear(x, [OH, OW]), TF-1.13 defaults: align_corners=False, legacy mapping
 * src = dst * (in / out) computed in fp32 (reference inference.py:96-99):
 *   top = tl + (tr - tl) * xl;  bot = bl + (br - bl) * xl;  out = top + (bot - top) * yl   (no fused ops) */
ORC_API void orc_resize_bilinear(const float *x, int N, int H, int W, int C, int OH, int OW, float *y)
{
    const float hs = (float)H / (float)OH, ws = (float)W / (float)OW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < OH; ++oy) {
            const float fy = (float)oy * hs;
            const int y0 = (int)floorf(fy);
            const int y1 = y0 + 1 < H - 1 ? y0 + 1 : H - 1;
            const float ly = fy - (float)y0;
            for (int ox = 0; ox < OW; ++ox) {
                const float fx = (float)ox * ws;
                const int x0 = (int)floorf(fx);
                const int x1 = x0 + 1 < W - 1 ? x0 + 1 : W - 1;
                const float lx = fx - (float)x0;
                const float *img = x + (size_t)n * H * W * C;
                const float *ptl = img + ((size_t)y0 * W + x0) * C, *ptr = img + ((size_t)y0 * W + x1) * C;
                const float *pbl = img + ((size_t)y1 * W + x0) * C, *pbr = img + ((size_t)y1 * W + x1) * C;
                float *yp = y + (((size_t)n * OH + oy) * OW + ox) * C;
                for (int c = 0; c < C; ++c) {
                    const float top = ptl[c] + (ptr[c] - ptl[c]) * lx;
                    const float bot = pbl[c] + (pbr[c] - pbl[c]) * lx;
                    yp[c] = top + (bot - top) * ly;
                }
            }
        }
}

/* Pyramid pooling bin average (ICNET_SPEC conv5_3_pool{1,2,3,6}): bin (i, j) of a b x b grid covers rows
 * [floor(i*H/b), ceil((i+1)*H/b)) and columns likewise; value = (sum over the rows, ascending, of the row's sum over
 * the columns, ascending) / count, all fp32. */
ORC_API void orc_adaptive_avg_pool(const float *x, int N, int H, int W, int C, int b, float *y /* [N,b,b,C] */)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < b; ++i)
            for (int j = 0; j < b; ++j) {
                const int y0 = (i * H) / b, y1 = ((i + 1) * H + b - 1) / b;
                const int x0 = (j * W) / b, x1 = ((j + 1) * W + b - 1) / b;
                const float cnt = (float)((y1 - y0) * (x1 - x0));
                for (int c = 0; c < C; ++c) {
                    float s = 0.0f;
                    for (int yy = y0; yy < y1; ++yy) {
                        float rs = 0.0f;
                        for (int xx = x0; xx < x1; ++xx) rs += x[(((size_t)n * H + yy) * W + xx) * C + c];
                        s += rs;
                    }
                    y[(((size_t)n * b + i) * b + j) * C + c] = s / cnt;
                }
            }
}

/* y = a + b (tf.add_n accumulates left to right) */
ORC_API void orc_add(const float *a, const float *b, size_t count, float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < count; ++i) y[i] = a[i] + b[i];
}
