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
