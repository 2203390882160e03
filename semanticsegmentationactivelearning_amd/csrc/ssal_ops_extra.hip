// ssal_ops_extra.hip -- stand-alone operators added after the first round: spatial dropout.
// gfx950 (MI355X) only.
#include "ssal_internal.h"
#include "ssal_prof.h"

namespace ssal {

static inline int cdiv_l(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ uint64_t splitmix64_x(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// xops.spatial_dropout (models/util/extra_ops.py:137-151): tf.nn.dropout(x, rate, noise_shape=[N,1,1,C]).
// TF-1.13 dropout: keep_prob = 1 - rate; binary = floor(keep_prob + uniform[0,1)); y = (x / keep_prob) * binary,
// with ONE uniform draw per (image, channel) plane.  The draw here is a counter-based hash of (seed, n*C + c)
// (restated for the tests in oracle/dropout_oracle.py) -- TensorFlow's own random stream cannot be
// reproduced, the distribution and the arithmetic are.
// One float4 (4 consecutive channels of one pixel) per thread per step: coalesced NHWC, HBM-bound
// (4 B in + 4 B out per element).
__global__ __launch_bounds__(256) void k_spatial_dropout(const float4 *__restrict__ x, long total4, long plane4,
                                                         int C, float keep_prob, uint64_t seed,
                                                         float4 *__restrict__ y)
{
    const int C4 = C >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long n = i / plane4;
        const int c0 = (int)(i % C4) * 4;
        const float4 v = x[i];
        float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t h = splitmix64_x(seed ^ ((uint64_t)(n * C + c0 + k) * 0xD1342543DE82EF95ull));
            const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
            const float keep = floorf(keep_prob + u);
            r[k] = (r[k] / keep_prob) * keep;
        }
        y[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// generic channel count (C % 4 != 0): one element per thread per step
__global__ __launch_bounds__(256) void k_spatial_dropout_1(const float *__restrict__ x, long total, long plane,
                                                           int C, float keep_prob, uint64_t seed,
                                                           float *__restrict__ y)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / plane;
        const int c = (int)(i % C);
        const uint64_t h = splitmix64_x(seed ^ ((uint64_t)(n * C + c) * 0xD1342543DE82EF95ull));
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
        const float keep = floorf(keep_prob + u);
        y[i] = (x[i] / keep_prob) * keep;
    }
}

hipError_t launch_spatial_dropout(const float *x, int N, int64_t pixels_per_image, int C, float rate, uint64_t seed,
                                  float *y, hipStream_t s)
{
    const long plane = pixels_per_image * C, total = plane * N;
    const float keep_prob = 1.0f - rate;
    ProfScope prof("k_spatial_dropout", 0.0, 8.0 * (double)total, s);
    if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        int grid = cdiv_l(total / 4, 256);
        if (grid > 262144) grid = 262144;
        hipLaunchKernelGGL(k_spatial_dropout, dim3(grid), dim3(256), 0, s, (const float4 *)x, total / 4, plane / 4, C,
                           keep_prob, seed, (float4 *)y);
    } else {
        int grid = cdiv_l(total, 256);
        if (grid > 262144) grid = 262144;
        hipLaunchKernelGGL(k_spatial_dropout_1, dim3(grid), dim3(256), 0, s, x, total, plane, C, keep_prob, seed, y);
    }
    return hipGetLastError();
}

}  // namespace ssal
