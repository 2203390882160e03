// ssal_probe.hip -- MEASUREMENT LIBRARY ONLY (libssal_hip_measure.so / libssal_hip_trace.so, built by
// tools/phase_trace.py with -DSSAL_MEASURE; build.py leaves this file out of the product libssal_hip.so).
// Memory-pattern probes (tools/mem_probe.py) and bare fp32 MFMA loops (tools/mfma_peak.py): y = x for an NHWC
// tensor with 64 channels, using the access shapes the fused kernels use, to find out what the memory
// system sustains for each shape.  No product path calls these.
#include "ssal_internal.h"
#include "ssal_mfma.h"
#include "ssal_prof.h"

namespace ssal {

// mode 0: linear, one float4 per thread
__global__ __launch_bounds__(256) void k_probe_linear(const float4 *x, float4 *y, long n4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) y[i] = x[i];
}

// modes 1..: one workgroup per 8 x 32 pixel tile (C = 64: 256 B per pixel), 4 waves, each wave 4 groups
// of 16 consecutive pixels (the M-tiles of k_bottleneck16).
//   FRAG = true : lane (i16, g) moves bytes [64 m + 16 g, +16) of pixel i16, m = 0..3  (MFMA operand shape)
//   FRAG = false: lane l moves bytes [1024 m + 16 l, +16) of the 4 KB group              (fully coalesced)
//   HALO        : additionally reads the one-pixel ring of the tile (as phase A does), result discarded
//   SPIN        : shader-clock cycles of dependent ALU work between the loads and the stores
// modes 10..13: linear addressing again, but U float4 per thread like the tile kernels: workgroup b owns the
// contiguous chunk of 256 * U float4; BATCH: all U loads, then all U stores; else load/store pairs in a loop
template <int U, bool BATCH>
__global__ __launch_bounds__(256) void k_probe_linear_u(const float4 *x, float4 *y, long n4)
{
    const long base = (long)blockIdx.x * 256 * U + threadIdx.x;
    if (BATCH) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = base + 256 * k < n4 ? x[base + 256 * k] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < U; ++k) if (base + 256 * k < n4) y[base + 256 * k] = v[k];
    } else {
#pragma unroll 1
        for (int k = 0; k < U; ++k) if (base + 256 * k < n4) y[base + 256 * k] = x[base + 256 * k];
    }
}

// mode 14: 16 float4 per thread as in mode 10, but slab addressing: chunk k of workgroup b is float4
// (k * gridDim + b) * 256 + t, so that at any moment all workgroups touch one compact 1/16 of the tensor
__global__ __launch_bounds__(256) void k_probe_linear_slab(const float4 *x, float4 *y, long n4)
{
    constexpr int U = 16;
    const long stride = (long)gridDim.x * 256;
    const long base = (long)blockIdx.x * 256 + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = base + stride * k < n4 ? x[base + stride * k] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < U; ++k) if (base + stride * k < n4) y[base + stride * k] = v[k];
}

// mode 9: the same 8 x 32 tile, but each wave moves its four 16-pixel groups one after the other
// (4 loads, 4 stores, next group) instead of 16 loads followed by 16 stores
__global__ __launch_bounds__(256) void k_probe_tile_seq(const float *x, float *y, int H, int W)
{
    constexpr int C = 64, TH = 8, TW = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int tiles_x = W / TW, tiles_y = H / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const float *ximg = x + (long)b * H * W * C;
    float *yimg = y + (long)b * H * W * C;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const int t0 = (wave + 4 * k) * 16;
        const int r = t0 / TW, c = t0 % TW;
        const long off = ((long)(ty * TH + r) * W + tx * TW + c + i16) * C + 4 * g;
        float4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = *reinterpret_cast<const float4 *>(ximg + off + 16 * m);
#pragma unroll
        for (int m = 0; m < 4; ++m) *reinterpret_cast<float4 *>(yimg + off + 16 * m) = v[m];
    }
}

// STAG > 0 (modes 19-21): every second first-round workgroup of a CU starts STAG cycles late, so that half of the resident
// workgroups store while the other half load (are the chip-wide read and write bursts of lock-stepped workgroups what
// keeps a tile-organised kernel below a linear copy?)
template <bool FRAG, bool HALO, int TH = 8, int TW = 32, int STAG = 0>
__global__ __launch_bounds__(256) void k_probe_tile(const float *x, float *y, int H, int W, int spin)
{
    constexpr int C = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (STAG > 0 && blockIdx.x < 1024 && ((blockIdx.x >> 8) & 1)) {
        const long t0 = __builtin_amdgcn_s_memtime();
        while ((long)__builtin_amdgcn_s_memtime() - t0 < STAG) __builtin_amdgcn_s_sleep(8);
    }
    const int i16 = lane & 15, g = lane >> 4;
    const int tiles_x = W / TW, tiles_y = H / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const float *ximg = x + (long)b * H * W * C;
    float *yimg = y + (long)b * H * W * C;
    float4 v[4][4];
    long off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t0 = (wave + 4 * k) * 16;  // first pixel of the group inside the tile
        const int r = t0 / TW, c = t0 % TW;
        const long base = ((long)(ty * TH + r) * W + tx * TW + c) * C;
        off[k] = FRAG ? base + (long)i16 * C + 4 * g : base + 4 * lane;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            v[k][m] = *reinterpret_cast<const float4 *>(ximg + off[k] + (FRAG ? 16 * m : 256 * m));
    }
    float4 hv[2][4];
    if (HALO) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = (wave + 4 * k) * 16 + i16;  // ring pixel 0..83
            int hr, hc;
            if (u < 34) { hr = -1; hc = u - 1; }
            else if (u < 68) { hr = TH; hc = u - 35; }
            else { const int q = u - 68; hr = q >> 1; hc = (q & 1) ? TW : -1; }
            const int py = ty * TH + hr, px = tx * TW + hc;
            const bool ok = u < 84 && py >= 0 && py < H && px >= 0 && px < W;
            const float *p = ximg + (ok ? ((long)py * W + px) * C : 0) + 4 * g;
#pragma unroll
            for (int m = 0; m < 4; ++m) hv[k][m] = *reinterpret_cast<const float4 *>(p + 16 * m);
        }
    }
    if (spin > 0) {
        float acc = v[0][0].x;
        const long t0 = __builtin_amdgcn_s_memtime();
        while ((long)__builtin_amdgcn_s_memtime() - t0 < spin) acc = fmaf(acc, 1.0000001f, 1e-9f);
        if (acc == 12345.678f) v[0][0].x = acc;
    }
    if (HALO) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m) s += hv[k][m].x + hv[k][m].w;
        if (s == 12345.678f) v[0][0].y = s;  // keeps the ring loads alive
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            *reinterpret_cast<float4 *>(yimg + off[k] + (FRAG ? 16 * m : 256 * m)) = v[k][m];
}

// modes 15 / 16: PERSISTENT workgroups (grid = 768 / 1024) walking 8 x 32 (mode 15) or 8 x 16 (mode 16) tiles with the
// fragment-shaped accesses of mode 1: the loads of tile i+1 are issued BEFORE the spin + stores of tile i (two static
// register sets), i.e. what a software-pipelined tile kernel would put on the memory system
template <int TW>
__global__ __launch_bounds__(256) void k_probe_tile_persistent(const float *x, float *y, int N, int H, int W, int spin)
{
    constexpr int C = 64, TH = 8, G = (TH * TW) / 64;  // 16-pixel groups per wave: 4 or 2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int ntiles = N * tiles_x * tiles_y;
    auto offs = [&](int t, long (&off)[G]) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int t0 = (wave + 4 * k) * 16;
            const int r = t0 / TW, c = t0 % TW;
            off[k] = ((long)n * H * W + (long)(ty * TH + r) * W + tx * TW + c + i16) * C + 4 * g;
        }
    };
    auto load = [&](const long (&off)[G], float4 (&v)[G][4]) {
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m) v[k][m] = *reinterpret_cast<const float4 *>(x + off[k] + 16 * m);
    };
    auto work_store = [&](const long (&off)[G], float4 (&v)[G][4]) {
        if (spin > 0) {
            float acc = v[0][0].x;
            const long t0 = __builtin_amdgcn_s_memtime();
            while ((long)__builtin_amdgcn_s_memtime() - t0 < spin) acc = fmaf(acc, 1.0000001f, 1e-9f);
            if (acc == 12345.678f) v[0][0].x = acc;
        }
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m) *reinterpret_cast<float4 *>(y + off[k] + 16 * m) = v[k][m];
    };
    float4 va[G][4], vb[G][4];
    long oa[G], ob[G];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    offs(t, oa);
    load(oa, va);
    for (;;) {
        const int t1 = t + gridDim.x;
        if (t1 < ntiles) { offs(t1, ob); load(ob, vb); }
        work_store(oa, va);
        if (t1 >= ntiles) break;
        const int t2 = t1 + gridDim.x;
        if (t2 < ntiles) { offs(t2, oa); load(oa, va); }
        work_store(ob, vb);
        if (t2 >= ntiles) break;
        t = t2;
    }
}

// modes 22 / 23: the 8 x 32 tile walked ROW BY ROW by its workgroup: the loads of row r+1 (2 float4 per thread) are
// issued, then row r is stored -- what a row-streaming form of the fused kernels would put on the memory system
// (mode 23: two rows per step, 4 float4 per thread)
template <int RS>
__global__ __launch_bounds__(256) void k_probe_tile_rows(const float *x, float *y, int H, int W, int spin)
{
    constexpr int C = 64, TH = 8, TW = 32, Q = RS * TW * C / 4 / 256;  // float4 per thread and step
    const int tiles_x = W / TW, tiles_y = H / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const long base = ((long)b * H * W + (long)(ty * TH) * W + tx * TW) * C;
    auto off = [&](int step, int q) {
        const int e = (int)threadIdx.x + 256 * q;            // float4 index inside the step's RS rows
        const int row = e / (TW * C / 4), c4 = e % (TW * C / 4);
        return base + ((long)(step * RS + row) * W) * C + 4 * c4;
    };
    float4 cur[Q], nxt[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) cur[q] = *reinterpret_cast<const float4 *>(x + off(0, q));
#pragma unroll 1
    for (int step = 0; step < TH / RS; ++step) {
        if (step + 1 < TH / RS) {
#pragma unroll
            for (int q = 0; q < Q; ++q) nxt[q] = *reinterpret_cast<const float4 *>(x + off(step + 1, q));
        }
        if (spin > 0) {
            float acc = cur[0].x;
            const long t0 = __builtin_amdgcn_s_memtime();
            while ((long)__builtin_amdgcn_s_memtime() - t0 < spin / (TH / RS)) acc = fmaf(acc, 1.0000001f, 1e-9f);
            if (acc == 12345.678f) cur[0].x = acc;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) *reinterpret_cast<float4 *>(y + off(step, q)) = cur[q];
#pragma unroll
        for (int q = 0; q < Q; ++q) cur[q] = nxt[q];
    }
}

hipError_t launch_copy_probe(int mode, const float *x, float *y, int N, int H, int W, int spin, hipStream_t s)
{
    if (H % 16 || W % 256) return hipErrorInvalidValue;
    const long n = (long)N * H * W * 64;
    static const char *names[] = {"probe linear", "probe tile frag", "probe tile coalesced", "probe tile frag+halo",
                                  "probe tile coalesced+halo"};
    if (mode < 0 || mode > 23) return hipErrorInvalidValue;
    ProfScope prof(mode <= 4 ? names[mode] : "probe tile shape", 0.0, 8.0 * n, s);
    const unsigned tiles = (unsigned)((long)N * (H / 8) * (W / 32));
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_probe_linear, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s,
                               (const float4 *)x, (float4 *)y, n / 4); break;
    case 1: hipLaunchKernelGGL((k_probe_tile<true, false>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 2: hipLaunchKernelGGL((k_probe_tile<false, false>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 3: hipLaunchKernelGGL((k_probe_tile<true, true>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 4: hipLaunchKernelGGL((k_probe_tile<false, true>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    // fragment-shaped accesses, other tile shapes with the same 256 pixels per workgroup
    case 5: hipLaunchKernelGGL((k_probe_tile<true, false, 4, 64>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 6: hipLaunchKernelGGL((k_probe_tile<true, false, 2, 128>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 7: hipLaunchKernelGGL((k_probe_tile<true, false, 1, 256>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 8: hipLaunchKernelGGL((k_probe_tile<true, false, 16, 16>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 9: hipLaunchKernelGGL(k_probe_tile_seq, dim3(tiles), dim3(256), 0, s, x, y, H, W); break;
    case 10: hipLaunchKernelGGL((k_probe_linear_u<16, true>), dim3((unsigned)((n / 4 + 4095) / 4096)), dim3(256), 0, s,
                                (const float4 *)x, (float4 *)y, n / 4); break;
    case 11: hipLaunchKernelGGL((k_probe_linear_u<16, false>), dim3((unsigned)((n / 4 + 4095) / 4096)), dim3(256), 0, s,
                                (const float4 *)x, (float4 *)y, n / 4); break;
    case 12: hipLaunchKernelGGL((k_probe_linear_u<4, true>), dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, s,
                                (const float4 *)x, (float4 *)y, n / 4); break;
    case 14: hipLaunchKernelGGL(k_probe_linear_slab, dim3((unsigned)((n / 4 + 4095) / 4096)), dim3(256), 0, s,
                                (const float4 *)x, (float4 *)y, n / 4); break;
    case 15: hipLaunchKernelGGL((k_probe_tile_persistent<32>), dim3(768), dim3(256), 0, s, x, y, N, H, W, spin); break;
    case 16: hipLaunchKernelGGL((k_probe_tile_persistent<16>), dim3(1024), dim3(256), 0, s, x, y, N, H, W, spin); break;
    case 17: hipLaunchKernelGGL((k_probe_tile_persistent<32>), dim3(512), dim3(256), 0, s, x, y, N, H, W, spin); break;
    case 18: hipLaunchKernelGGL((k_probe_tile_persistent<16>), dim3(2048), dim3(256), 0, s, x, y, N, H, W, spin); break;
    case 19: hipLaunchKernelGGL((k_probe_tile<true, false, 8, 32, 12000>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 20: hipLaunchKernelGGL((k_probe_tile<true, false, 8, 32, 25000>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 21: hipLaunchKernelGGL((k_probe_tile<true, false, 8, 32, 50000>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 22: hipLaunchKernelGGL((k_probe_tile_rows<1>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 23: hipLaunchKernelGGL((k_probe_tile_rows<2>), dim3(tiles), dim3(256), 0, s, x, y, H, W, spin); break;
    case 13: hipLaunchKernelGGL((k_probe_linear_u<2, true>), dim3((unsigned)((n / 4 + 511) / 512)), dim3(256), 0, s,
                                (const float4 *)x, (float4 *)y, n / 4); break;
    }
    return hipGetLastError();
}

// ---- measurement aid: what the fp32 matrix pipe of THIS device sustains (bare dependent-free MFMA
// loop, operands in registers, 4 accumulators per wave, 1 or 2 waves per SIMD) -------------------------
// SHAPE 32 / 16: four independent accumulators; SHAPE 132: ONE dependent 32x32x2 chain per wave;
// SHAPE 232: one dependent chain with a v_permlane32_swap feeding every MFMA pair (the conv loop shape)
template <int SHAPE>
__global__ __launch_bounds__(256) void k_mfma_peak(float *out, int iters)
{
    float a = 1.0f + 1e-3f * threadIdx.x, b = 0.999f;
    if (SHAPE == 32) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = mfma32(a, b, c0); c1 = mfma32(a, b, c1); c2 = mfma32(a, b, c2); c3 = mfma32(a, b, c3);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (SHAPE == 132) {
        f32x16 c0 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = mfma32(a, b, c0); c0 = mfma32(b, a, c0); c0 = mfma32(a, b, c0); c0 = mfma32(b, a, c0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c0[5];
    } else if (SHAPE == 332 || SHAPE == 432) {
        // four independent accumulators + 8 (332) / 16 (432) independent v_fma_f32 per four MFMAs: does vector work of
        // the same wave / of co-resident waves execute in the shadow of the matrix pipe, or does it add to it?
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a + k;
        for (int i = 0; i < iters; ++i) {
            c0 = mfma32(a, b, c0);
#pragma unroll
            for (int k = 0; k < (SHAPE == 432 ? 4 : 2); ++k) v[k] = fmaf(v[k], 1.0000001f, b);
            c1 = mfma32(a, b, c1);
#pragma unroll
            for (int k = 0; k < (SHAPE == 432 ? 4 : 2); ++k) v[2 + k] = fmaf(v[2 + k], 1.0000001f, b);
            c2 = mfma32(a, b, c2);
#pragma unroll
            for (int k = 0; k < (SHAPE == 432 ? 4 : 2); ++k) v[4 + (k & 3)] = fmaf(v[4 + (k & 3)], 1.0000001f, b);
            c3 = mfma32(a, b, c3);
#pragma unroll
            for (int k = 0; k < (SHAPE == 432 ? 4 : 2); ++k) v[(6 + k) & 7] = fmaf(v[(6 + k) & 7], 1.0000001f, b);
        }
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + t;
    } else if (SHAPE == 516 || SHAPE == 616) {
        // "beyond the fp32 wall" probe (measurement library only, nothing of it ships): one K = 16
        // step of a 32x32 fp32 GEMM tile as SIX v_mfma_f32_32x32x16_bf16 on operands split into three bf16 terms each
        // (x = x1 + x2 + x3 exactly by truncation; products x1w1, x1w2, x2w1, x1w3, x2w2, x3w1 accumulated in fp32 --
        // what is dropped is O(2^-24) relative).  516: the activation operand is split in registers every step (the form a
        // kernel fed with fp32 tensors needs: 4 and / sub + 1.5 v_perm per value); 616: both operands arrive pre-split
        // (weights, or a tensor its producer stored as three bf16 planes).  Reported as fp32-EQUIVALENT flops: 2*32*32*16
        // per step.
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        f32x16 c0 = {0};
        unsigned xr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xr[k] = __float_as_uint(a + 0.37f * k);
        uint4 w1 = make_uint4(0x3f803f80u, 0x3f813f7fu, 0x3f823f7eu, 0x3f833f7du), w2 = w1, w3 = w1;
        w2.x ^= threadIdx.x; w3.y ^= threadIdx.x;
        uint4 p1 = w1, p2 = w2, p3 = w3;
        for (int i = 0; i < iters; ++i) {
            if (SHAPE == 516) {
                unsigned t1[8], t2[8], t3[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned x = xr[k] + (unsigned)i;            // a new value every step: nothing to hoist
                    const unsigned h1 = x & 0xffff0000u;
                    const float r1 = __uint_as_float(x) - __uint_as_float(h1);
                    const unsigned h2 = __float_as_uint(r1) & 0xffff0000u;
                    const float r2 = r1 - __uint_as_float(h2);
                    t1[k] = h1; t2[k] = h2; t3[k] = __float_as_uint(r2);
                }
                auto pack = [](unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); };
                p1 = make_uint4(pack(t1[0], t1[1]), pack(t1[2], t1[3]), pack(t1[4], t1[5]), pack(t1[6], t1[7]));
                p2 = make_uint4(pack(t2[0], t2[1]), pack(t2[2], t2[3]), pack(t2[4], t2[5]), pack(t2[6], t2[7]));
                p3 = make_uint4(pack(t3[0], t3[1]), pack(t3[2], t3[3]), pack(t3[4], t3[5]), pack(t3[6], t3[7]));
            } else {
                p1.x += 1u;  // keep the operands loop-variant
            }
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, p1), a2 = __builtin_bit_cast(bf16x8, p2), a3 = __builtin_bit_cast(bf16x8, p3);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, w1), b2 = __builtin_bit_cast(bf16x8, w2), b3 = __builtin_bit_cast(bf16x8, w3);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, c0, 0, 0, 0);  // small terms first
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c0, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c0[5];
    } else if (SHAPE == 232) {
        f32x16 c0 = {0};
        float p = a, q = b;
        for (int i = 0; i < iters; ++i) {
            swap32(p, q); c0 = mfma32(a, p, c0); c0 = mfma32(b, q, c0);
            swap32(p, q); c0 = mfma32(a, p, c0); c0 = mfma32(b, q, c0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c0[5];
    } else {
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = mfma16(a, b, c0); c1 = mfma16(a, b, c1); c2 = mfma16(a, b, c2); c3 = mfma16(a, b, c3);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

hipError_t launch_mfma_peak(int shape, int blocks, int iters, float *out, hipStream_t s)
{
    const bool is32 = shape != 16;
    const bool split = shape == 516 || shape == 616;  // one K = 16 step of a 32x32 tile per iteration: 2 * 32 * 32 * 16 flops
    const double flop = (double)blocks * 4 /*waves*/ * iters * (split ? 32768.0 : 4.0 * (is32 ? 4096.0 : 2048.0));
    const char *nm = shape == 32 ? "k_mfma_peak<32x32x2 4acc>" : shape == 132 ? "k_mfma_peak<32x32x2 1chain>"
                   : shape == 232 ? "k_mfma_peak<32x32x2 1chain+swap>"
                   : shape == 332 ? "k_mfma_peak<32x32x2 4acc + 8 v_fma / 4 mfma>"
                   : shape == 432 ? "k_mfma_peak<32x32x2 4acc + 16 v_fma / 4 mfma>"
                   : shape == 516 ? "k_mfma_peak<bf16x3: 6 x 32x32x16_bf16 per K=16 step, operand split in registers> (fp32-equivalent)"
                   : shape == 616 ? "k_mfma_peak<bf16x3: 6 x 32x32x16_bf16 per K=16 step, operands pre-split> (fp32-equivalent)"
                   : "k_mfma_peak<16x16x4 4acc>";
    ProfScope prof(nm, flop, 0.0, s);
    if (shape == 32) hipLaunchKernelGGL(k_mfma_peak<32>, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (shape == 132) hipLaunchKernelGGL(k_mfma_peak<132>, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (shape == 232) hipLaunchKernelGGL(k_mfma_peak<232>, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (shape == 332) hipLaunchKernelGGL(k_mfma_peak<332>, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (shape == 432) hipLaunchKernelGGL(k_mfma_peak<432>, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (shape == 516) hipLaunchKernelGGL(k_mfma_peak<516>, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (shape == 616) hipLaunchKernelGGL(k_mfma_peak<616>, dim3(blocks), dim3(256), 0, s, out, iters);
    else hipLaunchKernelGGL(k_mfma_peak<16>, dim3(blocks), dim3(256), 0, s, out, iters);
    return hipGetLastError();
}

}  // namespace ssal
