// ssal_bf16x3.h -- kernel packing of the opt-in SSAL_ARITH_BF16X3 mode (ssal_bottleneck_bf16x3.hip), shared by the commit
// step (host: ssal_api.hip) and the kernels.  A layer's three kernels are pre-split into bf16 triples (w = w1 + w2 + w3
// exactly, by truncation) and laid out per K = 16 chunk in MFMA operand order: 16-byte UNITS, [chunk][term][lane], where
// operand lane (j = lane & 31, h = lane >> 5) of a v_mfma_f32_32x32x16_bf16 holds k = 8 h + i, i = 0..7, of row / column j.
//   chunks 0 .. 7                      projection   Wp[ci = 16 c + 8 h + i][co = j]
//   chunks 8 .. 8 + 2 taps - 1         convolution  Wc[tap][ci = 16 c2 + 8 h + i][co = j], chunk = 8 + 2 tap + c2
//                                      (asymmetric block: taps 0..4 = the (5,1) kernel, 5..9 = the (1,5) kernel)
//   then 8 chunks                      expansion    We[ci = 16 c + 8 h + i][co = 32 nt + j], chunk = 2 nt + c
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace ssal {
namespace bf16x3 {

constexpr int CHUNK_UNITS = 192;  // 3 terms x 64 lanes
constexpr int WP_OFF = 0;
constexpr int WC_OFF = 8 * CHUNK_UNITS;
constexpr __host__ __device__ int we_off(int taps) { return WC_OFF + 2 * taps * CHUNK_UNITS; }
constexpr __host__ __device__ int units(int taps) { return we_off(taps) + 8 * CHUNK_UNITS; }

// host twin of the device split (ssal_bottleneck_bf16x3.hip: split1): three truncations, each remainder exact in fp32
inline void split_host(float x, uint32_t t[3])
{
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t h1 = u & 0xffff0000u;
    float f1;
    memcpy(&f1, &h1, 4);
    volatile float r1 = x - f1;  // volatile: one rounding per operation, whatever the host compiler would like to contract
    float r1f = r1;
    memcpy(&u, &r1f, 4);
    const uint32_t h2 = u & 0xffff0000u;
    float f2;
    memcpy(&f2, &h2, 4);
    volatile float r2 = r1f - f2;
    float r2f = r2;
    memcpy(&u, &r2f, 4);
    t[0] = h1 >> 16;
    t[1] = h2 >> 16;
    t[2] = u >> 16;
}

// appends one chunk: val(h, i, j) = the fp32 operand element of lane (j, h), k-slot i
template <typename Fn> void pack_chunk(std::vector<float> &out, Fn val)
{
    const size_t base = out.size();
    out.resize(base + (size_t)CHUNK_UNITS * 4);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out.data() + base);
    for (int lane = 0; lane < 64; ++lane) {
        const int j = lane & 31, h = lane >> 5;
        uint32_t t[8][3];
        for (int i = 0; i < 8; ++i) split_host(val(h, i, j), t[i]);
        for (int term = 0; term < 3; ++term)
            for (int m = 0; m < 4; ++m)  // dword m = elements 2m (low half) and 2m + 1 (high half)
                dst[((size_t)term * 64 + lane) * 4 + m] = t[2 * m][term] | (t[2 * m + 1][term] << 16);
    }
}

// the whole layer: wp [128][32], wc [taps][32][32] (asymmetric: the (5,1) kernel followed by the (1,5) kernel), we [32][128];
// returned as floats (bit patterns) so that it travels in the handle's one weight arena
inline std::vector<float> pack_layer(const float *wp, const float *wc, const float *wc2, int taps, const float *we)
{
    std::vector<float> out;
    out.reserve((size_t)units(taps) * 4);
    for (int c = 0; c < 8; ++c) pack_chunk(out, [&](int h, int i, int j) { return wp[(16 * c + 8 * h + i) * 32 + j]; });
    for (int tap = 0; tap < taps; ++tap) {
        const float *w = (wc2 && tap >= 5) ? wc2 + (size_t)(tap - 5) * 32 * 32 : wc + (size_t)tap * 32 * 32;
        for (int c2 = 0; c2 < 2; ++c2) pack_chunk(out, [&](int h, int i, int j) { return w[(16 * c2 + 8 * h + i) * 32 + j]; });
    }
    for (int nt = 0; nt < 4; ++nt)
        for (int c = 0; c < 2; ++c) pack_chunk(out, [&](int h, int i, int j) { return we[(16 * c + 8 * h + i) * 128 + 32 * nt + j]; });
    return out;
}

// Downsample bottleneck 64 -> 128 (Bottleneck2_0): wp [2][2][64][32] (K = 4 taps x 64 channels = 16 chunks), wc [9][32][32],
// we [32][128] (the expansion runs as D[co][pixel] there: the chunk is the A operand, rows = output channels 32 nt + j)
constexpr int DN_WP_OFF = 0;
constexpr int DN_WC_OFF = 16 * CHUNK_UNITS;
constexpr int DN_WE_OFF = DN_WC_OFF + 18 * CHUNK_UNITS;
constexpr int DN_UNITS = DN_WE_OFF + 8 * CHUNK_UNITS;
inline std::vector<float> pack_down_layer(const float *wp, const float *wc, const float *we)
{
    std::vector<float> out;
    out.reserve((size_t)DN_UNITS * 4);
    // chunk s = 4 cc + t: tap t = dy * 2 + dx of channel group cc; k-slot (h, i) = channel 16 cc + 4 h + (i & 3) + 8 (i >> 2) -- a
    // lane half then holds the channels c with (c >> 2) & 1 == h, the ones a D[co][pixel] accumulator gives it: the pooled
    // residual of the block meets the expansion's output in the same lane (ssal_bottleneck_bf16x3.hip: project_down)
    for (int s = 0; s < 16; ++s)
        pack_chunk(out, [&](int h, int i, int j) {
            return wp[((s & 3) * 64 + 16 * (s >> 2) + 4 * h + (i & 3) + 8 * (i >> 2)) * 32 + j];
        });
    for (int tap = 0; tap < 9; ++tap)
        for (int c2 = 0; c2 < 2; ++c2)
            pack_chunk(out, [&](int h, int i, int j) { return wc[((size_t)tap * 32 + 16 * c2 + 8 * h + i) * 32 + j]; });
    for (int nt = 0; nt < 4; ++nt)
        for (int c = 0; c < 2; ++c) pack_chunk(out, [&](int h, int i, int j) { return we[(16 * c + 8 * h + i) * 128 + 32 * nt + j]; });
    return out;
}

// Upsample bottleneck 128 -> 64 (Bottleneck4_0): wp [128][32], wr (residual 1x1) [128][64], ws = the stacked transposed-conv
// kernel [6 slots][32 ci][32 rows = two output-parity classes x 16 channels] (ssal_api.hip: stack_convT), we [16][64].  Every
// chunk is an A operand (rows = output channels / stacked rows), the activations are the B operand (columns = pixels).
constexpr int UP_WP_OFF = 0;                                   // 8 chunks
constexpr int UP_WR_OFF = 8 * CHUNK_UNITS;                     // chunk 2 c + nt, c = 0..7
constexpr int UP_WS_OFF = UP_WR_OFF + 16 * CHUNK_UNITS;        // chunk 2 slot + c2
constexpr int UP_WE_OFF = UP_WS_OFF + 12 * CHUNK_UNITS;        // chunk nt (K = 16: one chunk)
constexpr int UP_UNITS = UP_WE_OFF + 2 * CHUNK_UNITS;
inline std::vector<float> pack_up_layer(const float *wp, const float *wr, const float *ws, const float *we)
{
    std::vector<float> out;
    out.reserve((size_t)UP_UNITS * 4);
    for (int c = 0; c < 8; ++c) pack_chunk(out, [&](int h, int i, int j) { return wp[(16 * c + 8 * h + i) * 32 + j]; });
    for (int c = 0; c < 8; ++c)
        for (int nt = 0; nt < 2; ++nt) pack_chunk(out, [&](int h, int i, int j) { return wr[(16 * c + 8 * h + i) * 64 + 32 * nt + j]; });
    for (int sl = 0; sl < 6; ++sl)
        for (int c2 = 0; c2 < 2; ++c2) pack_chunk(out, [&](int h, int i, int j) { return ws[((size_t)sl * 32 + 16 * c2 + 8 * h + i) * 32 + j]; });
    for (int nt = 0; nt < 2; ++nt) pack_chunk(out, [&](int h, int i, int j) { return we[(8 * h + i) * 64 + 32 * nt + j]; });
    return out;
}

}  // namespace bf16x3
}  // namespace ssal
