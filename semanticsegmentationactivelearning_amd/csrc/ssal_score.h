// ssal_score.h -- per-pixel acquisition score + block reductions shared by the score kernels
// (k_final_score / k_score_logits in ssal_kernels.hip, k_upscore in ssal_icnet_kernels.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace ssal {

// ------------------------------------------------------------------------------------------------
// Per-pixel acquisition score (active_learning.py:239-260) on K logits held in registers.
//   softmax: p_k = exp(x_k - m) / S,  S = sum_k exp(x_k - m)
//   entropy   : conf = 1 - H/log(K),  H = -sum p log p = log S - sum_k e_k (x_k - m) / S
//               (the reference adds FLT_MIN inside the log; it changes H by < 1e-36)
//   margin    : conf = p_(1) - p_(2) = (1 - e_(2)) / S
//   confidence: conf = p_(1) = 1 / S
// label = first maximum of the logits (tf.math.argmax, :234-236).
// Non-finite logits (policy, tests/test_gpu_parity.py::test_score_nonfinite_policy):
//   * exp underflow (x_k - m < -104) and x_k = -inf give p_k = 0 exactly and contribute 0 to H -- the value the
//     reference's  -p * log(p + FLT_MIN)  takes at p = 0;
//   * a NaN logit, or a +inf maximum (inf - inf in the softmax), makes the pixel's confidence NaN, as the
//     reference's tf.nn.softmax does; the per-image mean is then NaN and np.argpartition ranks it last.
// ------------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ float pixel_score(const float (&l)[K], int measure, float inv_logK,
                                             int &label)
{
    float m = l[0];
    int am = 0;
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (l[k] > m) { m = l[k]; am = k; }
    label = am;
    float S = 0.0f, T = 0.0f, e2 = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float d = l[k] - m;
        const float e = __expf(d);  // NaN (NaN logit, or inf - inf) propagates into S and from there into every measure
        S += e;
        // -inf logits (p = 0 exactly, like the reference's softmax): keep 0 * d finite so T stays a number
        T = fmaf(e, fmaxf(d, -3.0e38f), T);
        if (k != am && e > e2) e2 = e;
    }
    if (measure == 0) {
        const float Hn = __logf(S) - T / S;
        return 1.0f - Hn * inv_logK;
    } else if (measure == 1) {
        return (1.0f - e2) / S;
    }
    return 1.0f / S;
}

// Score-only form (no label): the same arithmetic per measure, but only the quantities that measure needs -- the
// class-argmax index and the running second-largest exponential of the generic form above cost more VALU work than
// the softmax itself.  `measure` is wave-uniform, so the three bodies are three branches of one kernel.
//   entropy   : m, S, T                                    (bit-identical to pixel_score: same ops, same order)
//   margin    : (m1, m2) = the two largest logits (m2 == m1 on a tie), S;  e_(2) = exp(m2 - m1) is the very value the
//               generic form finds as the largest exponential among the other classes
//   confidence: m, S
template <int K>
__device__ __forceinline__ float pixel_score_only(const float (&l)[K], int measure, float inv_logK)
{
    if (measure == 1) {
        float m1 = l[0], m2 = -__builtin_inff();
#pragma unroll
        for (int k = 1; k < K; ++k) {
            m2 = fmaxf(m2, fminf(m1, l[k]));
            m1 = fmaxf(m1, l[k]);
        }
        float S = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) S += __expf(l[k] - m1);
        return (1.0f - __expf(m2 - m1)) / S;
    }
    float m = l[0];
#pragma unroll
    for (int k = 1; k < K; ++k) m = fmaxf(m, l[k]);
    float S = 0.0f, T = 0.0f;
    if (measure == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = l[k] - m;
            const float e = __expf(d);
            S += e;
            T = fmaf(e, fmaxf(d, -3.0e38f), T);
        }
        const float Hn = __logf(S) - T / S;
        return 1.0f - Hn * inv_logK;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) S += __expf(l[k] - m);
    return 1.0f / S;
}

// The score-only form on class PAIRS (v_pk_add_f32 / v_pk_mul_f32 process two fp32 operations per lane and issue slot): the
// subtraction of the maximum and the exp2 pre-scale (__expf(d) IS v_exp_f32(d * log2(e)): one multiply, one v_exp) run
// packed; every operation, and the order of the sums, is pixel_score_only's -- the same bits.  l2[p] = (l[2p], l[2p + 1]);
// the upper half of the last pair of an odd K is never read.
typedef float score_f32x2 __attribute__((ext_vector_type(2)));
template <int K>
__device__ __forceinline__ float pixel_score_only_pk(const score_f32x2 (&l2)[(K + 1) / 2], int measure, float inv_logK)
{
    constexpr int KP = (K + 1) / 2;
    auto el = [&](int k) { return (k & 1) ? l2[k >> 1].y : l2[k >> 1].x; };
    const score_f32x2 log2e = {0x1.715476p+0f, 0x1.715476p+0f};
    float m1 = el(0), m2 = -__builtin_inff();
    if (measure == 1) {
#pragma unroll
        for (int k = 1; k < K; ++k) {
            m2 = fmaxf(m2, fminf(m1, el(k)));
            m1 = fmaxf(m1, el(k));
        }
    } else {
#pragma unroll
        for (int k = 1; k < K; ++k) m1 = fmaxf(m1, el(k));
    }
    const score_f32x2 mm = {m1, m1};
    score_f32x2 d2[KP], p2[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        d2[p] = l2[p] - mm;
        p2[p] = d2[p] * log2e;
    }
    auto dk = [&](int k) { return (k & 1) ? d2[k >> 1].y : d2[k >> 1].x; };
    auto ek = [&](int k) { return __builtin_amdgcn_exp2f((k & 1) ? p2[k >> 1].y : p2[k >> 1].x); };
    float S = 0.0f, T = 0.0f;
    if (measure == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float e = ek(k);
            S += e;
            T = fmaf(e, fmaxf(dk(k), -3.0e38f), T);
        }
        const float Hn = __logf(S) - T / S;
        return 1.0f - Hn * inv_logK;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) S += ek(k);
    if (measure == 1) return (1.0f - __expf(m2 - m1)) / S;
    return 1.0f / S;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide fp64 sum, result valid in thread 0 (256 threads = 4 waves)
__device__ __forceinline__ double block_sum_256(double v, double *lds4)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) lds4[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
    return r;
}

}  // namespace ssal
