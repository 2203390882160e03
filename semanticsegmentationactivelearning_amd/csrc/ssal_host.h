// ssal_host.h -- host-side helpers shared by the C-ABI translation units (ssal_api.hip: ENet handle and stand-alone
// operators; ssal_icnet_api.hip: ICNet handle).  Host C++ only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#define SSAL_API extern "C" __attribute__((visibility("default")))

namespace ssal {

// sets the thread-local message behind ssal_last_error() and returns `code`
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return ssal::fail(SSAL_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

// one named parameter tensor staged on the host until commit
struct HostTensor {
    std::string name;
    std::vector<int64_t> dims;
    std::vector<float> data;
    bool set = false;
    int64_t numel() const
    {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
};

// all weights of a handle go into ONE device arena (256-B aligned slices)
struct ArenaBuilder {
    std::vector<float> host;
    size_t push(const float *p, size_t n)
    {
        size_t off = (host.size() + 63) / 64 * 64;  // 256-B alignment
        host.resize(off + n);
        memcpy(host.data() + off, p, n * sizeof(float));
        return off;
    }
    size_t push(const std::vector<float> &v) { return push(v.data(), v.size()); }
};

// QUAD layout of the regular 128-channel bottleneck's kernels (csrc/ssal_bottleneck_args.h: quad::): for every group of four MFMA
// steps the four fragments of a lane sit in one float4 -- wp [128][32]: (q, lane (j, h), i) = wp[2 (4q + i) + h][j], q < 16;
// wc [taps][32][32]: (tap, q, lane, i) = wc[tap][2 (4q + i) + h][j], q < 4; we [32][128]: (nt, q, lane, i) = we[2 (4q + i) + h][32 nt + j]
inline std::vector<float> bnk_quad_layout(const float *wp, const float *wc, const float *wc2, int taps, const float *we)
{   // wc2 != NULL (asymmetric block, taps = 10): taps 0..4 = wc (the (5,1) kernel), 5..9 = wc2 (the (1,5) kernel)
    std::vector<float> out((size_t)128 * 32 + (size_t)taps * 32 * 32 + 32 * 128);
    size_t o = 0;
    for (int q = 0; q < 16; ++q)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i) out[o++] = wp[(2 * (4 * q + i) + (lane >> 5)) * 32 + (lane & 31)];
    for (int tap = 0; tap < taps; ++tap) {
        const float *w = (wc2 && tap >= 5) ? wc2 + (size_t)(tap - 5) * 32 * 32 : wc + (size_t)tap * 32 * 32;
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) out[o++] = w[(2 * (4 * q + i) + (lane >> 5)) * 32 + (lane & 31)];
    }
    for (int nt = 0; nt < 4; ++nt)
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) out[o++] = we[(2 * (4 * q + i) + (lane >> 5)) * 128 + 32 * nt + (lane & 31)];
    return out;
}

// bump allocator over a caller-owned workspace (base == NULL: size query)
struct Bump {
    char *base;
    int64_t cap, off = 0;
    bool ok = true;
    Bump(void *b, int64_t c) : base((char *)b), cap(c) {}
    template <typename Tp> Tp *take(int64_t count)
    {
        int64_t o = (off + 255) / 256 * 256;
        int64_t bytes = count * (int64_t)sizeof(Tp);
        off = o + bytes;
        if (base && off > cap) ok = false;
        return base ? (Tp *)(base + o) : nullptr;
    }
};

}  // namespace ssal
