/*
 * ssal_icnet.h -- C ABI of libssal_hip.so for the ICNet row of the pool-scoring hot path (BASELINE config C4:
 * ICNet multi-scale 1/4, 1/2, 1 at 1024x2048, margin acquisition).
 *
 * The reference's models/icnet/icnet.py:1-7 is an EMPTY class (docstring :3 cites the ICNet paper): there is no
 * reference interface or behaviour to replace.  The network implemented here is pinned in ICNET_SPEC.md; each
 * operator uses the semantics the reference repository defines for it (SAME convolutions as
 * models/enet/enet_modules.py:205,538,565,581; batch-norm models/util/extra_ops.py:154-185; bilinear resize
 * inference.py:96-99; acquisition measures active_learning.py:239-263).  The entry points follow the pattern of
 * the ENet handle (include/ssal_enet.h): same conventions, status codes, ownership and threading rules (concurrent
 * forward / score calls on one committed handle need their own stream + workspace; one handle per device).
 */
#ifndef SSAL_ICNET_H
#define SSAL_ICNET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssal_icnet ssal_icnet;

/* Model handle: stands where `models.ICNet(classes)` would (models/icnet/icnet.py:1-7).  Tensor names are
 * "<layer>.<attr>" with the layer names of ICNET_SPEC.md (e.g. "conv1_1_3x3_s2.kernel", "conv4_3_3x3.gamma",
 * "conv6_cls.bias"); kernels HWIO, batch-norm vectors [C]. */
int ssal_icnet_create(int c_in, int classes, ssal_icnet **out);
int ssal_icnet_destroy(ssal_icnet *net);
int ssal_icnet_num_tensors(const ssal_icnet *net);
int ssal_icnet_tensor_info(const ssal_icnet *net, int i, const char **name, int *ndim, int64_t dims[4]);
int ssal_icnet_set_tensor(ssal_icnet *net, const char *name, const float *host, int64_t numel);
/* fold the batch-norm statistics (extra_ops.py:181-184, eps = 1e-3), re-layout the kernels for the matrix-core
 * convolution and upload */
int ssal_icnet_commit(ssal_icnet *net, void *stream);
int64_t ssal_icnet_workspace_bytes(const ssal_icnet *net, int n, int h, int w);

/* ICNet.call(inputs, training=False) -> logits [n,h,w,classes] (ICNET_SPEC section 4: conv6_interp);
 * x_dev [n,h,w,c_in] fp32 (or uint8 through the _u8 form: x * f32(1/255), tensortools/input.py:289-290);
 * h, w divisible by 32. */
int ssal_icnet_forward_nhwc(ssal_icnet *net, const float *x_dev, int n, int h, int w, float *logits_dev,
                            void *ws_dev, int64_t ws_bytes, void *stream);
int ssal_icnet_forward_nhwc_u8(ssal_icnet *net, const uint8_t *x_dev, int n, int h, int w, float *logits_dev,
                               void *ws_dev, int64_t ws_bytes, void *stream);

/* forward + softmax + acquisition measure + float64 per-image mean (active_learning.py:229-263), fused: the
 * full-resolution logits never reach HBM (the 4x bilinear conv6_interp is evaluated inside the score kernel).
 * Outputs as ssal_enet_score_nhwc: scores_dev [n] float64; optional label_dev / mask_dev uint8 [n,h,w] (4-byte
 * aligned: four pixels leave per store), conf_dev fp32 [n,h,w] (16-byte aligned). */
int ssal_icnet_score_nhwc(ssal_icnet *net, const float *x_dev, int n, int h, int w, int measure, float threshold,
                          double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev, float *conf_dev, void *ws_dev,
                          int64_t ws_bytes, void *stream);
int ssal_icnet_score_nhwc_u8(ssal_icnet *net, const uint8_t *x_dev, int n, int h, int w, int measure,
                             float threshold, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev,
                             float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream);

/* Named intermediate tensors of the LAST forward/score call on a workspace (every ICNET_SPEC layer output that is
 * materialised keeps its own buffer): byte offset into the workspace and NHWC dims.  SSAL_EINVAL for a name that
 * is not materialised (the 2x interpolations are evaluated inside the consuming convolution).  Every endpoint is valid
 * after a FORWARD call; a SCORE call runs conv2_2 / conv2_3 as one fused launch each and conv1_sub1 + conv2_sub1 as one
 * (k_front2, csrc/ssal_icnet_front.hip; knob "ic_front", include/ssal_enet.h), so the *_1x1_reduce / *_3x3 buffers of those
 * two blocks, conv1_sub1's and the four *_1x1_proj buffers (knob "ic_dual": the projection shortcut is evaluated inside the
 * increase launch) keep whatever an earlier call left there (stale) -- inspect layer outputs after ssal_icnet_forward_nhwc.
 * ssal_icnet_endpoint_valid_after_score answers, for one name and frame size and the knobs as they are now: 1 = a score
 * call writes it, 0 = a fused launch swallows it, -1 = unknown name / bad arguments (models.ICNet.endpoint raises on 0). */
int ssal_icnet_num_endpoints(const ssal_icnet *net);
int ssal_icnet_endpoint_name(const ssal_icnet *net, int i, const char **name);
int ssal_icnet_endpoint_info(const ssal_icnet *net, const char *name, int n, int h, int w, int64_t *offset,
                             int64_t dims[4]);
int ssal_icnet_endpoint_valid_after_score(const ssal_icnet *net, const char *name, int h, int w);

/* Stand-alone fused convolution (the operator every ICNet layer is built from; also the per-block parity hook):
 * y = [relu]( BN(conv2d(x, kernel HWIO, strides s, dilations d, "SAME")) [+ res] ), BN given as mean / variance /
 * gamma / beta (all NULL: no batch-norm; bias_dev optional).  upsample2x != 0 runs the conv on
 * tf.image.resize_bilinear(x, 2x) evaluated on the fly.  cin % 32 == 0 (matrix-core path) or cin in {1,3,4} with
 * a 3x3 / stride-2 / 32-channel kernel (first-layer path).  ws: ssal_conv_bn_workspace_bytes().  The kernel and the
 * batch-norm vectors are host arrays: they are folded / re-laid-out and uploaded on every call (the call synchronises
 * the stream once for that), which is what a per-operator test hook needs; the network handle does it once, at commit. */
int64_t ssal_conv_bn_workspace_bytes(int kh, int kw, int cin, int cout);
int ssal_conv_bn_act(const float *x_dev, int n, int h, int w, int cin, const float *kernel_host, int kh, int kw,
                     int cout, int stride, int dilation, const float *mean_host, const float *var_host,
                     const float *gamma_host, const float *beta_host, const float *bias_host,
                     const float *res_dev, int relu, int upsample2x, float *y_dev, void *ws_dev, int64_t ws_bytes,
                     void *stream);
/* tf.nn.max_pool(x, 3x3, strides 2, "SAME") -> [n, ceil(h/2), ceil(w/2), c];  c % 4 == 0 */
int ssal_max_pool_3x3_s2(const float *x_dev, int n, int h, int w, int c, float *y_dev, void *stream);
/* ICNET_SPEC pyramid pooling: y = x + sum over b in (1,2,3,6) of resize_bilinear(bin_average_b(x), h, w);
 * (bin average = sum over the bin's rows of the sum over its columns, divided by the count);
 * ws >= n * (50 + 12 * h) * c * 4 bytes; c % 4 == 0 */
int ssal_pyramid_pooling(const float *x_dev, int n, int h, int w, int c, float *y_dev, void *ws_dev, int64_t ws_bytes,
                         void *stream);
/* conv6_interp + score on materialised 1/4-resolution logits lq_dev [n,h,w,classes]: outputs at [n,4h,4w] */
int64_t ssal_upscore_workspace_bytes(int n, int h, int w);
int ssal_upscore_logits_nhwc(const float *lq_dev, int n, int h, int w, int classes, int measure, float threshold,
                             double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev, float *conf_dev,
                             void *ws_dev, int64_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSAL_ICNET_H */
