// ssal_icnet.h -- launch wrappers of the ICNet kernels (ssal_icnet_kernels.hip), shared with the ICNet handle
// (ssal_icnet_api.hip).  gfx950 (MI355X / CDNA4) only.  All tensors fp32 NHWC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssal_measure.h"

namespace ssal {

// One fused convolution of ICNET_SPEC.md: conv (SAME, no bias) -> folded batch-norm (scale, shift) -> [+ res] -> [relu].
struct IgemmArgs {
    const float *x;      // [N,H,W,Cin]  (UP2: the HALF-resolution source [N,H/2,W/2,Cin], see `up2`)
    const float *wt;     // re-laid-out kernel [KH*KW][Cin/32][CoutP][32]  (igemm_relayout)
    float *y;            // [N,Ho,Wo,Cout]
    const float *scale;  // [CoutP] (1.0 for the bias-only classifier)
    const float *shift;  // [CoutP]
    const float *res;    // [N,Ho,Wo,Cout] or NULL
    // second source (k_igemm<.., DUAL>; NULL otherwise): y += fmaf(conv1x1_stride2(x2, wt2), scale2, shift2) instead of + res
    const float *x2, *wt2, *scale2, *shift2;
    int Cin2, H2, W2, stride2;
    int N, H, W, Cin, Ho, Wo, Cout, CoutP;
    int KH, KW, stride, dil, pad_t, pad_l;
    int relu;
    int up2;             // 1: x is the half-resolution tensor; the conv runs on resize_bilinear(x, 2x) computed on the fly
    long M;              // N*Ho*Wo
    int tiles_m, tiles_n;
    int ntiles, xcd_chunk;  // XCD-aware tile order: tile = (b % 8) * xcd_chunk + b / 8 (xcd_chunk = 0: tile = b)
    unsigned long long *trace;  // phase-trace buffer (NULL unless a -DSSAL_PHASE_TRACE build is being traced)
#ifdef SSAL_MEASURE
    int ablate;  // measurement builds only: bit 0 = no MFMAs, bit 1 = no global loads, bit 2 = no LDS writes
#endif
};

// kernel [KH][KW][Cin][Cout] (HWIO) -> [KH*KW][Cin/32][CoutP][32], CoutP = Cout rounded up to 32 (zero rows)
size_t igemm_relayout_floats(int KH, int KW, int Cin, int Cout);
void igemm_relayout(const float *w_hwio, int KH, int KW, int Cin, int Cout, float *out);
bool igemm_supported(int Cin, int Cout, int KH, int KW);
// a bottleneck's last 1x1 convolution (x -> y, stride 1) with its PROJECTION shortcut evaluated in the same launch:
// y = [relu](fmaf(conv1x1(x, wt), scale, shift) + fmaf(conv1x1(xs[::stride_s, ::stride_s], wt_s), scale_s, shift_s));
// xs = [N, H * stride_s, W * stride_s, Cin_s].  Bit-identical to launch_igemm(xs ...) -> t, launch_igemm(x ..., res = t).
hipError_t launch_igemm_dual(const float *x, int N, int H, int W, int Cin, const float *wt, int Cout, const float *scale,
                             const float *shift, const float *xs, int Cin_s, int stride_s, const float *wt_s,
                             const float *scale_s, const float *shift_s, bool relu, float *y, hipStream_t s);
hipError_t launch_igemm(const float *x, int N, int H, int W, int Cin, const float *wt, int KH, int KW, int Cout,
                        int stride, int dil, const float *scale, const float *shift, const float *res, bool relu,
                        bool up2, float *y, hipStream_t s);

// first convolution of a branch: 3x3 / stride 2 / SAME on a 1-, 3- or 4-channel image -> 32 channels, BN, ReLU.
// sub = 1: plain;  sub = 2: the conv runs on resize_bilinear(x, H/2, W/2) (ICNET_SPEC data_sub2), which at the exact
// factor 2 of the legacy mapping is x[2y][2x] bit for bit.  x may be the decoded uint8 frame (x * f32(1/255)).
hipError_t launch_conv_first(const void *x, bool x_is_u8, int N, int H, int W, int Cin, int sub, const float *w,
                             const float *scale, const float *shift, float *y, hipStream_t s);

// the first TWO convolutions of a branch in one launch (ssal_icnet_front.hip): launch_conv_first(x, ..., sub, w1, s1, t1)
// followed by the 3x3 / stride `stride2` (1 or 2) / SAME convolution 32 -> 32 + BN + ReLU whose kernel is w2_igemm
// (igemm_relayout layout); y = [N, H/sub/2/stride2, W/sub/2/stride2, 32].  Bit-identical to the two launches.
bool front2_supported(int H, int W, int Cin, int sub, int stride2);
hipError_t launch_front2(const void *x, bool x_is_u8, int N, int H, int W, int Cin, int sub, const float *w1,
                         const float *s1, const float *t1, const float *w2_igemm, const float *s2, const float *t2,
                         int stride2, float *y, hipStream_t s);

// tf.nn.max_pool(3x3, stride 2, SAME); C % 4 == 0
hipError_t launch_maxpool3x3_s2(const float *x, int N, int H, int W, int C, float *y, hipStream_t s);

// launches of one layer the caller runs side by side (image-group chains): the convolution launchers size their
// "fill the chip" rules for 512 / concurrency workgroups per launch.  Thread-local.
void set_launch_concurrency(int g);
int launch_concurrency();

// pyramid pooling (ICNET_SPEC conv5_3_pool* / conv5_3_sum): scratch = ppm_scratch_floats(N, H, C) floats
int64_t ppm_scratch_floats(int N, int H, int C);
hipError_t launch_ppm(const float *x, int N, int H, int W, int C, float *scratch, float *y, hipStream_t s);

// conv6_interp (4x bilinear, legacy mapping) fused with the acquisition score (active_learning.py:234-263):
// lq = 1/4-resolution logits [N,H,W,K]; outputs at [N,4H,4W]; partial: [N * upscore_blocks(H,W)] doubles
int upscore_blocks(int H, int W);
hipError_t launch_upscore(const float *lq, int N, int H, int W, int K, int measure, float threshold, double *partial,
                          uint8_t *label, uint8_t *mask, float *conf, hipStream_t s);

}  // namespace ssal
