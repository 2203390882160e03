// ssal_api.hip -- the C ABI (include/ssal_enet.h): handle management, weight staging, layer
// sequencing.  Host C++ only; no torch types.  gfx950 (MI355X) only.
#include "../../include/ssal_enet.h"
#include "ssal_internal.h"
#include "ssal_host.h"
#include "ssal_prof.h"
#include "ssal_bottleneck_args.h"
#include "ssal_bf16x3.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace ssal;


// ------------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

namespace ssal {
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace ssal

SSAL_API const char *ssal_version(void) { 
#ifdef SSAL_MEASURE
    return "ssal-hip 0.2 (gfx950, MEASUREMENT build: not for results)";
#else
    return "ssal-hip 0.2 (gfx950)";
#endif
 }
SSAL_API const char *ssal_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// ENet topology (models/enet/enet.py:35-247)
// ------------------------------------------------------------------------------------------------
enum Kind { K_INITIAL, K_REGULAR, K_DOWN, K_UP, K_FINAL };

struct LayerSpec {
    const char *name;
    Kind kind;
    int cout;   // output channels (0 for Final: = classes)
    int dil;    // dilation of the 3x3 conv
    bool asym;  // (5,1) then (1,5) instead of 3x3
};

static const LayerSpec kSpecs[] = {
    {"Initial", K_INITIAL, 16, 1, false},
    {"Bottleneck1_0", K_DOWN, 64, 1, false},
    {"Bottleneck1_1", K_REGULAR, 64, 1, false},
    {"Bottleneck1_2", K_REGULAR, 64, 1, false},
    {"Bottleneck1_3", K_REGULAR, 64, 1, false},
    {"Bottleneck1_4", K_REGULAR, 64, 1, false},
    {"Bottleneck2_0", K_DOWN, 128, 1, false},
    {"Bottleneck2_1", K_REGULAR, 128, 1, false},
    {"Bottleneck2_2", K_REGULAR, 128, 2, false},
    {"Bottleneck2_3", K_REGULAR, 128, 1, true},
    {"Bottleneck2_4", K_REGULAR, 128, 4, false},
    {"Bottleneck2_5", K_REGULAR, 128, 1, false},
    {"Bottleneck2_6", K_REGULAR, 128, 8, false},
    {"Bottleneck2_7", K_REGULAR, 128, 1, true},
    {"Bottleneck2_8", K_REGULAR, 128, 16, false},
    {"Bottleneck3_1", K_REGULAR, 128, 1, false},
    {"Bottleneck3_2", K_REGULAR, 128, 2, false},
    {"Bottleneck3_3", K_REGULAR, 128, 1, true},
    {"Bottleneck3_4", K_REGULAR, 128, 4, false},
    {"Bottleneck3_5", K_REGULAR, 128, 1, false},
    {"Bottleneck3_6", K_REGULAR, 128, 8, false},
    {"Bottleneck3_7", K_REGULAR, 128, 1, true},
    {"Bottleneck3_8", K_REGULAR, 128, 16, false},
    {"Bottleneck4_0", K_UP, 64, 1, false},
    {"Bottleneck4_1", K_REGULAR, 64, 1, false},
    {"Bottleneck4_2", K_REGULAR, 64, 1, false},
    {"Bottleneck5_0", K_UP, 16, 1, false},
    {"Bottleneck5_1", K_REGULAR, 16, 1, false},
    {"Final", K_FINAL, 0, 1, false},
};
static const int kNumLayers = (int)(sizeof(kSpecs) / sizeof(kSpecs[0]));

// device-side view of one layer after commit
struct DevLayer {
    Kind kind;
    int cin, cout, dil;
    bool asym;
    int f;  // bottleneck width (proj output channels)
    int cf; // conv output channels (== f except Upsample: f/2)
    const float *w = nullptr;  // Initial conv kernel / Final re-laid-out kernel
    const float *scale = nullptr, *shift = nullptr, *alpha = nullptr;  // Initial
    const float *proj_w = nullptr, *proj_scale = nullptr, *proj_shift = nullptr, *proj_alpha = nullptr;
    const float *conv_w = nullptr, *conv_w1 = nullptr;  // conv_w1: second (1,5) kernel when asym
    const float *conv_scale = nullptr, *conv_shift = nullptr, *conv_alpha = nullptr;
    const float *exp_w = nullptr, *exp_scale = nullptr, *exp_shift = nullptr;
    const float *res_w = nullptr, *res_alpha = nullptr;
    const float *convT_stacked = nullptr;  // Upsample: [6][f][2*cf] parity-stacked transposed-conv kernel
    const float *wq = nullptr;   // regular 128-channel blocks: wp | wc | we in quad layout (bnk_quad_layout)
    const float *bf3 = nullptr;  // SSAL_ARITH_BF16X3: the block's kernels pre-split into bf16 triples (ssal_bf16x3.h); NULL where the mode has no kernel
};

struct ssal_enet {
    int c_in = 3, classes = 19;
    std::vector<HostTensor> tensors;
    std::map<std::string, int> index;
    std::vector<DevLayer> layers;
    float *arena = nullptr;
    size_t arena_floats = 0;
    bool committed = false;
    int device = -1;  // the device the arena lives on (set by commit); calls on another device are refused
};

static void add_tensor(ssal_enet *h, const std::string &name, std::vector<int64_t> dims)
{
    HostTensor t;
    t.name = name;
    t.dims = std::move(dims);
    h->index[name] = (int)h->tensors.size();
    h->tensors.push_back(std::move(t));
}

static void add_bn(ssal_enet *h, const std::string &p, int c)
{
    add_tensor(h, p + "mean", {c});
    add_tensor(h, p + "variance", {c});
    add_tensor(h, p + "gamma", {c});
    add_tensor(h, p + "beta", {c});
}

// parameter inventory in the reference's add_weight order
// (enet_modules.py:139-187, 366-523, 730-865, 1070-1214, 1349-1356)
static void declare_tensors(ssal_enet *h)
{
    int c = h->c_in;
    for (int li = 0; li < kNumLayers; ++li) {
        const LayerSpec &s = kSpecs[li];
        const std::string n = std::string(s.name) + ".";
        switch (s.kind) {
        case K_INITIAL:
            add_tensor(h, n + "kernel", {3, 3, c, 16 - c});
            add_bn(h, n, 16);
            add_tensor(h, n + "alpha", {16});
            c = 16;
            break;
        case K_REGULAR: {
            const int f = c / 4;
            add_tensor(h, n + "proj_kernel", {1, 1, c, f});
            add_tensor(h, n + "proj_alpha", {f});
            add_bn(h, n + "proj_", f);
            if (s.asym) {
                add_tensor(h, n + "conv_kernel.0", {5, 1, f, f});
                add_tensor(h, n + "conv_kernel.1", {1, 5, f, f});
            } else {
                add_tensor(h, n + "conv_kernel", {3, 3, f, f});
            }
            add_tensor(h, n + "conv_alpha", {f});
            add_bn(h, n + "conv_", f);
            add_tensor(h, n + "exp_kernel", {1, 1, f, s.cout});
            add_bn(h, n + "exp_", s.cout);
            add_tensor(h, n + "residual_alpha", {s.cout});
            break;
        }
        case K_DOWN: {
            const int f = 2 * (c / 4);  // enet_modules.py:705
            add_tensor(h, n + "proj_kernel", {2, 2, c, f});
            add_tensor(h, n + "proj_alpha", {f});
            add_bn(h, n + "proj_", f);
            add_tensor(h, n + "conv_kernel", {3, 3, f, f});
            add_tensor(h, n + "conv_alpha", {f});
            add_bn(h, n + "conv_", f);
            add_tensor(h, n + "exp_kernel", {1, 1, f, s.cout});
            add_bn(h, n + "exp_", s.cout);
            add_tensor(h, n + "residual_alpha", {s.cout});
            c = s.cout;
            break;
        }
        case K_UP: {
            const int pf = c / 4, cf = pf / 2;  // enet_modules.py:1042-1043
            add_tensor(h, n + "proj_kernel", {1, 1, c, pf});
            add_tensor(h, n + "proj_alpha", {pf});
            add_bn(h, n + "proj_", pf);
            add_tensor(h, n + "conv_kernel", {3, 3, cf, pf});  // HW-O-I (:1046)
            add_tensor(h, n + "conv_alpha", {cf});
            add_bn(h, n + "conv_", cf);
            add_tensor(h, n + "exp_kernel", {1, 1, cf, s.cout});
            add_bn(h, n + "exp_", s.cout);
            add_tensor(h, n + "res_kernel", {1, 1, c, s.cout});
            add_tensor(h, n + "residual_alpha", {s.cout});
            c = s.cout;
            break;
        }
        case K_FINAL:
            add_tensor(h, n + "kernel", {3, 3, h->classes, c});  // HW-O-I (:1341)
            break;
        }
    }
}

SSAL_API int ssal_enet_create(int c_in, int classes, ssal_enet **out)
{
    if (!out) return fail(SSAL_EINVAL, "out is NULL");
    if (!(c_in == 1 || c_in == 3 || c_in == 4))
        return fail(SSAL_EINVAL, "c_in must be 1, 3 or 4 (got %d)", c_in);
    if (classes < 2 || classes > 32)
        return fail(SSAL_EINVAL, "classes must be in [2,32] (got %d)", classes);
    ssal_enet *h = new ssal_enet();
    h->c_in = c_in;
    h->classes = classes;
    declare_tensors(h);
    *out = h;
    return SSAL_OK;
}

SSAL_API int ssal_enet_destroy(ssal_enet *net)
{
    if (!net) return SSAL_OK;
    if (net->arena) (void)hipFree(net->arena);
    delete net;
    return SSAL_OK;
}

SSAL_API int ssal_enet_num_tensors(const ssal_enet *net) { return net ? (int)net->tensors.size() : 0; }

SSAL_API int ssal_enet_tensor_info(const ssal_enet *net, int i, const char **name, int *ndim,
                                   int64_t dims[4])
{
    if (!net || i < 0 || i >= (int)net->tensors.size()) return fail(SSAL_EINVAL, "bad tensor index %d", i);
    const HostTensor &t = net->tensors[i];
    if (name) *name = t.name.c_str();
    if (ndim) *ndim = (int)t.dims.size();
    if (dims)
        for (size_t d = 0; d < 4; ++d) dims[d] = d < t.dims.size() ? t.dims[d] : 1;
    return SSAL_OK;
}

SSAL_API int ssal_enet_set_tensor(ssal_enet *net, const char *name, const float *host, int64_t numel)
{
    if (!net || !name || !host) return fail(SSAL_EINVAL, "NULL argument");
    auto it = net->index.find(name);
    if (it == net->index.end()) return fail(SSAL_EINVAL, "unknown tensor '%s'", name);
    HostTensor &t = net->tensors[it->second];
    if (numel != t.numel())
        return fail(SSAL_EINVAL, "tensor '%s': expected %lld elements, got %lld", name,
                    (long long)t.numel(), (long long)numel);
    t.data.assign(host, host + numel);
    t.set = true;
    net->committed = false;
    return SSAL_OK;
}

// ---- commit: fold BN, re-layout transposed kernels, upload one arena ---------------------------
namespace {
const std::vector<float> &T(const ssal_enet *h, const std::string &name)
{
    return h->tensors[h->index.at(name)].data;
}

// tf.nn.fused_batch_norm(is_training=False), eps=1e-3 (extra_ops.py:181-184) folded:
//   s = gamma / sqrt(var + eps), t = fma(-mean, s, beta)
void fold_bn(const ssal_enet *h, const std::string &prefix, int c, std::vector<float> &s,
             std::vector<float> &t)
{
    const auto &mean = T(h, prefix + "mean"), &var = T(h, prefix + "variance");
    const auto &gamma = T(h, prefix + "gamma"), &beta = T(h, prefix + "beta");
    s.resize(c);
    t.resize(c);
    for (int i = 0; i < c; ++i) {
        const float sg = gamma[i] / sqrtf(var[i] + 1e-3f);
        s[i] = sg;
        t[i] = fmaf(-mean[i], sg, beta[i]);
    }
}

// Parity-stacked transposed-conv kernel for the MFMA upsample kernel: ws[slot][ci][row], rows
// [0,O) = first class, [O,2O) = second class of the slot (zeros where the class has no tap):
//   slot 0: P(i,j)     [W00 | W01]     slot 1: P(i,j-1)   [W02 | 0]
//   slot 2: P(i-1,j)   [W20 | W21]     slot 3: P(i-1,j-1) [W22 | 0]
//   slot 4: P(i,j)     [W10 | W11]     slot 5: P(i,j-1)   [W12 | 0]
// with Wab[ci][co] = kernel[a][b][co][ci] (TF HW-O-I layout).
std::vector<float> stack_convT(const std::vector<float> &w, int O, int I)
{
    static const int taps[6][2] = {{0, 1}, {2, -1}, {6, 7}, {8, -1}, {3, 4}, {5, -1}};  // kh*3+kw
    const int R = 2 * O;
    std::vector<float> r((size_t)6 * I * R, 0.0f);
    for (int sl = 0; sl < 6; ++sl)
        for (int half = 0; half < 2; ++half) {
            const int t = taps[sl][half];
            if (t < 0) continue;
            for (int ci = 0; ci < I; ++ci)
                for (int co = 0; co < O; ++co)
                    r[((size_t)sl * I + ci) * R + half * O + co] = w[((size_t)t * O + co) * I + ci];
        }
    return r;
}

// [3,3,O,I] (TF conv2d_transpose kernel) -> [3][3][I][O]
std::vector<float> hwoi_to_hwio(const std::vector<float> &w, int O, int I)
{
    std::vector<float> r(w.size());
    for (int t = 0; t < 9; ++t)
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) r[((size_t)t * I + i) * O + o] = w[((size_t)t * O + o) * I + i];
    return r;
}

// [rows][K] -> [rows][K2], K2 = K rounded up to even, zero padded: k_final_score reads class PAIRS
std::vector<float> pad_cols_even(const std::vector<float> &w, int K)
{
    const int K2 = (K + 1) / 2 * 2;
    const size_t rows = w.size() / K;
    std::vector<float> r(rows * K2, 0.0f);
    for (size_t i = 0; i < rows; ++i)
        for (int k = 0; k < K; ++k) r[i * K2 + k] = w[i * K + k];
    return r;
}
}  // namespace

SSAL_API int ssal_enet_commit(ssal_enet *net, void *stream)
{
    if (!net) return fail(SSAL_EINVAL, "net is NULL");
    for (const auto &t : net->tensors)
        if (!t.set) return fail(SSAL_ESTATE, "tensor '%s' has not been set", t.name.c_str());

    ArenaBuilder ab;
    struct Off { size_t v[20]; };
    std::vector<Off> offs(kNumLayers);
    std::vector<DevLayer> layers(kNumLayers);
    int c = net->c_in;
    std::vector<float> s, t;
    for (int li = 0; li < kNumLayers; ++li) {
        const LayerSpec &sp = kSpecs[li];
        const std::string n = std::string(sp.name) + ".";
        DevLayer &L = layers[li];
        Off &o = offs[li];
        L.kind = sp.kind;
        L.cin = c;
        L.dil = sp.dil;
        L.asym = sp.asym;
        L.cout = sp.kind == K_FINAL ? net->classes : sp.cout;
        switch (sp.kind) {
        case K_INITIAL:
            o.v[0] = ab.push(T(net, n + "kernel"));
            fold_bn(net, n, 16, s, t);
            o.v[1] = ab.push(s);
            o.v[2] = ab.push(t);
            o.v[3] = ab.push(T(net, n + "alpha"));
            L.f = L.cf = 0;
            break;
        case K_REGULAR:
        case K_DOWN:
        case K_UP: {
            const int f = sp.kind == K_DOWN ? 2 * (c / 4) : c / 4;
            const int cf = sp.kind == K_UP ? f / 2 : f;
            L.f = f;
            L.cf = cf;
            o.v[0] = ab.push(T(net, n + "proj_kernel"));
            fold_bn(net, n + "proj_", f, s, t);
            o.v[1] = ab.push(s);
            o.v[2] = ab.push(t);
            o.v[3] = ab.push(T(net, n + "proj_alpha"));
            if (sp.asym) {
                o.v[4] = ab.push(T(net, n + "conv_kernel.0"));
                o.v[5] = ab.push(T(net, n + "conv_kernel.1"));
            } else if (sp.kind == K_UP) {
                o.v[4] = ab.push(hwoi_to_hwio(T(net, n + "conv_kernel"), cf, f));
                o.v[5] = 0;
            } else {
                o.v[4] = ab.push(T(net, n + "conv_kernel"));
                o.v[5] = 0;
            }
            fold_bn(net, n + "conv_", cf, s, t);
            o.v[6] = ab.push(s);
            o.v[7] = ab.push(t);
            o.v[8] = ab.push(T(net, n + "conv_alpha"));
            o.v[9] = ab.push(T(net, n + "exp_kernel"));
            fold_bn(net, n + "exp_", L.cout, s, t);
            o.v[10] = ab.push(s);
            o.v[11] = ab.push(t);
            o.v[12] = sp.kind == K_UP ? ab.push(T(net, n + "res_kernel")) : 0;
            o.v[13] = ab.push(T(net, n + "residual_alpha"));
            o.v[14] = sp.kind == K_UP ? ab.push(stack_convT(T(net, n + "conv_kernel"), cf, f)) : 0;
            o.v[17] = 0;
            if (sp.kind == K_REGULAR && c == 128 && f == 32)
                o.v[17] = ab.push(bnk_quad_layout(T(net, n + "proj_kernel").data(),
                                                  T(net, n + (sp.asym ? "conv_kernel.0" : "conv_kernel")).data(),
                                                  sp.asym ? T(net, n + "conv_kernel.1").data() : nullptr, sp.asym ? 10 : 9,
                                                  T(net, n + "exp_kernel").data()));
            o.v[15] = 0;
            if (sp.kind == K_REGULAR && bottleneck_bf16x3_supported(c, f))  // the opt-in arithmetic mode: 104 / 112 KB per layer
                o.v[15] = ab.push(bf16x3::pack_layer(T(net, n + "proj_kernel").data(),
                                                     T(net, n + (sp.asym ? "conv_kernel.0" : "conv_kernel")).data(),
                                                     sp.asym ? T(net, n + "conv_kernel.1").data() : nullptr, sp.asym ? 10 : 9,
                                                     T(net, n + "exp_kernel").data()));
            if (sp.kind == K_DOWN && downsample_bf16x3_supported(c, L.cout))  // Bottleneck2_0 of the opt-in mode (129 KB)
                o.v[15] = ab.push(bf16x3::pack_down_layer(T(net, n + "proj_kernel").data(), T(net, n + "conv_kernel").data(),
                                                          T(net, n + "exp_kernel").data()));
            if (sp.kind == K_UP && upsample_bf16x3_supported(c, L.cout))  // Bottleneck4_0 of the opt-in mode (117 KB)
                o.v[15] = ab.push(bf16x3::pack_up_layer(T(net, n + "proj_kernel").data(), T(net, n + "res_kernel").data(),
                                                        stack_convT(T(net, n + "conv_kernel"), cf, f).data(),
                                                        T(net, n + "exp_kernel").data()));
            break;
        }
        case K_FINAL:
            o.v[0] = ab.push(pad_cols_even(hwoi_to_hwio(T(net, n + "kernel"), net->classes, c), net->classes));
            L.f = L.cf = 0;
            break;
        }
        c = L.cout;
    }

    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (net->arena && (net->arena_floats < ab.host.size() || net->device != dev)) {  // a handle lives on ONE device
        HIP_TRY(hipFree(net->arena));
        net->arena = nullptr;
    }
    if (!net->arena) {
        HIP_TRY(hipMalloc((void **)&net->arena, ab.host.size() * sizeof(float)));
        net->arena_floats = ab.host.size();
        net->device = dev;
    }
    HIP_TRY(hipMemcpyAsync(net->arena, ab.host.data(), ab.host.size() * sizeof(float),
                           hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // the staging vector dies at return

    const float *A = net->arena;
    for (int li = 0; li < kNumLayers; ++li) {
        DevLayer &L = layers[li];
        const Off &o = offs[li];
        switch (L.kind) {
        case K_INITIAL:
            L.w = A + o.v[0]; L.scale = A + o.v[1]; L.shift = A + o.v[2]; L.alpha = A + o.v[3];
            break;
        case K_FINAL:
            L.w = A + o.v[0];
            break;
        default:
            L.proj_w = A + o.v[0]; L.proj_scale = A + o.v[1]; L.proj_shift = A + o.v[2];
            L.proj_alpha = A + o.v[3];
            L.conv_w = A + o.v[4]; L.conv_w1 = L.asym ? A + o.v[5] : nullptr;
            L.conv_scale = A + o.v[6]; L.conv_shift = A + o.v[7]; L.conv_alpha = A + o.v[8];
            L.exp_w = A + o.v[9]; L.exp_scale = A + o.v[10]; L.exp_shift = A + o.v[11];
            L.res_w = L.kind == K_UP ? A + o.v[12] : nullptr;
            L.res_alpha = A + o.v[13];
            L.convT_stacked = L.kind == K_UP ? A + o.v[14] : nullptr;
            L.bf3 = o.v[15] ? A + o.v[15] : nullptr;
            L.wq = o.v[17] ? A + o.v[17] : nullptr;
            break;
        }
    }
    net->layers = std::move(layers);
    net->committed = true;
    return SSAL_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------
namespace {
struct LayerTemps {
    float *t0, *t1, *t2, *t3;
};

// temp tensor sizes (floats) for one layer at INPUT dims (n,h,w); general = scatter-based unpool
void layer_temp_floats(const DevLayer &L, int64_t n, int64_t h, int64_t w, bool general, int64_t sz[4])
{
    sz[0] = sz[1] = sz[2] = sz[3] = 0;
    switch (L.kind) {
    case K_REGULAR:
        sz[0] = sz[1] = n * h * w * L.f;
        if (L.asym) sz[2] = n * h * w * L.f;
        break;
    case K_DOWN:
        sz[0] = sz[1] = n * (h / 2) * (w / 2) * L.f;
        break;
    case K_UP:
        sz[0] = n * h * w * L.f;
        sz[1] = n * 4 * h * w * L.cf;
        sz[2] = n * h * w * L.cout;
        if (general) sz[3] = n * 4 * h * w * L.cout;
        break;
    default:
        break;
    }
}

ConvArgs conv_args(const float *x, int N, int H, int W, int Cin, const float *w, int KH, int KW,
                   int Cout, int stride, int dil, float *y)
{
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.KH = KH; a.KW = KW; a.stride = stride; a.dil = dil;
    // TF "SAME": out = ceil(in/stride); pad_total = max((out-1)*stride + (k-1)*dil + 1 - in, 0); before = total/2
    a.Ho = (H + stride - 1) / stride;
    a.Wo = (W + stride - 1) / stride;
    int th = (a.Ho - 1) * stride + (KH - 1) * dil + 1 - H; if (th < 0) th = 0;
    int tw = (a.Wo - 1) * stride + (KW - 1) * dil + 1 - W; if (tw < 0) tw = 0;
    a.pad_t = th / 2;
    a.pad_l = tw / 2;
    a.res_mode = RES_NONE;
    return a;
}

#define HIP_RET(expr)                          \
    do {                                       \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return e_;       \
    } while (0)

// kernel-family switch (A/B measurements and cross-checks): 1 = MFMA-fused bottlenecks where the
// shape is supported (default), 0 = generic kernels everywhere.  Both are bit-identical.
bool g_use_mfma = true;

// Bottleneck.call (enet_modules.py:526-599)
// size guards of the fused launchers (32-bit byte offsets inside one image)
bool regular_fused(const DevLayer &L, int h, int w)
{
    return g_use_mfma && (long)h * w * L.cin <= (1L << 29) && bottleneck_mfma_supported(L.cin, L.f, L.asym);
}
bool down_fused(const DevLayer &L, int h, int w)
{
    return g_use_mfma && (long)h * w * L.cout < (1L << 31) && downsample_mfma_supported(L.cin, L.cout);
}
bool up_fused(const DevLayer &L, int h, int w)
{
    return g_use_mfma && (long)h * w * 4 * L.cout < (1L << 31) && upsample_mfma_supported(L.cin, L.cout);
}

hipError_t run_regular(const DevLayer &L, const float *x, int n, int h, int w, float *y,
                       const LayerTemps &T, hipStream_t s, int arith = SSAL_ARITH_F32)
{
    const int C = L.cin, f = L.f;
    if (arith == SSAL_ARITH_BF16X3 && L.bf3 && regular_fused(L, h, w)) {  // opt-in: split-operand bf16 MFMAs, NOT bit-identical
        BnkArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x; a.y = y;
        a.wp = L.proj_w; a.ps = L.proj_scale; a.pt = L.proj_shift; a.pa = L.proj_alpha;
        a.wc = L.conv_w; a.wc2 = L.asym ? L.conv_w1 : nullptr; a.cs = L.conv_scale; a.ct = L.conv_shift; a.ca = L.conv_alpha;
        a.we = L.exp_w; a.es = L.exp_scale; a.et = L.exp_shift; a.ra = L.res_alpha;
        a.N = n; a.H = h; a.W = w; a.dil = L.dil;
        return launch_bottleneck_bf16x3(a, L.bf3, s);
    }
    if (regular_fused(L, h, w))
        return launch_bottleneck_mfma(x, y, n, h, w, C, L.dil, L.proj_w, L.proj_scale, L.proj_shift,
                                      L.proj_alpha, L.conv_w, L.asym ? L.conv_w1 : nullptr,
                                      L.conv_scale, L.conv_shift, L.conv_alpha, L.exp_w, L.exp_scale,
                                      L.exp_shift, L.res_alpha, s, L.wq);
    ConvArgs p = conv_args(x, n, h, w, C, L.proj_w, 1, 1, f, 1, 1, T.t0);
    p.scale = L.proj_scale; p.shift = L.proj_shift; p.alpha = L.proj_alpha;
    HIP_RET(launch_conv(p, s));
    const float *mid;
    if (L.asym) {
        ConvArgs c0 = conv_args(T.t0, n, h, w, f, L.conv_w, 5, 1, f, 1, 1, T.t2);
        HIP_RET(launch_conv(c0, s));  // no BN / activation between the two 1-D convs (:553-563)
        ConvArgs c1 = conv_args(T.t2, n, h, w, f, L.conv_w1, 1, 5, f, 1, 1, T.t1);
        c1.scale = L.conv_scale; c1.shift = L.conv_shift; c1.alpha = L.conv_alpha;
        HIP_RET(launch_conv(c1, s));
        mid = T.t1;
    } else {
        ConvArgs c0 = conv_args(T.t0, n, h, w, f, L.conv_w, 3, 3, f, 1, L.dil, T.t1);
        c0.scale = L.conv_scale; c0.shift = L.conv_shift; c0.alpha = L.conv_alpha;
        HIP_RET(launch_conv(c0, s));
        mid = T.t1;
    }
    ConvArgs e = conv_args(mid, n, h, w, f, L.exp_w, 1, 1, L.cout, 1, 1, y);
    e.scale = L.exp_scale; e.shift = L.exp_shift; e.alpha = nullptr;
    e.res_mode = RES_ADD; e.res = x; e.res_C = C; e.res_alpha = L.res_alpha;
    return launch_conv(e, s);
}

// BottleneckDownsample.call (enet_modules.py:868-938); code: [n,h/2,w/2,cin] window codes
hipError_t run_down(const DevLayer &L, const float *x, int n, int h, int w, float *y, uint8_t *code,
                    const LayerTemps &T, hipStream_t s, int arith = SSAL_ARITH_F32)
{
    const int C = L.cin, f = L.f;
    if (arith == SSAL_ARITH_BF16X3 && L.bf3 && down_fused(L, h, w)) {  // opt-in mode; the pooling residual / codes stay exact
        DownArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x; a.y = y; a.code = code;
        a.wp = L.proj_w; a.ps = L.proj_scale; a.pt = L.proj_shift; a.pa = L.proj_alpha;
        a.wc = L.conv_w; a.cs = L.conv_scale; a.ct = L.conv_shift; a.ca = L.conv_alpha;
        a.we = L.exp_w; a.es = L.exp_scale; a.et = L.exp_shift; a.ra = L.res_alpha;
        a.N = n; a.H = h; a.W = w;
        return launch_downsample_bf16x3(a, L.bf3, s);
    }
    if (down_fused(L, h, w))
        return launch_downsample_mfma(x, y, code, n, h, w, C, L.proj_w, L.proj_scale, L.proj_shift,
                                      L.proj_alpha, L.conv_w, L.conv_scale, L.conv_shift, L.conv_alpha,
                                      L.exp_w, L.exp_scale, L.exp_shift, L.res_alpha, s);
    ConvArgs p = conv_args(x, n, h, w, C, L.proj_w, 2, 2, f, 2, 1, T.t0);
    p.scale = L.proj_scale; p.shift = L.proj_shift; p.alpha = L.proj_alpha;
    HIP_RET(launch_conv(p, s));
    ConvArgs c0 = conv_args(T.t0, n, h / 2, w / 2, f, L.conv_w, 3, 3, f, 1, L.dil, T.t1);
    c0.scale = L.conv_scale; c0.shift = L.conv_shift; c0.alpha = L.conv_alpha;
    HIP_RET(launch_conv(c0, s));
    ConvArgs e = conv_args(T.t1, n, h / 2, w / 2, f, L.exp_w, 1, 1, L.cout, 1, 1, y);
    e.scale = L.exp_scale; e.shift = L.exp_shift;
    e.res_mode = RES_POOL; e.res = x; e.res_C = C; e.code_out = code; e.res_alpha = L.res_alpha;
    return launch_conv(e, s);
}

// BottleneckUpsample.call (enet_modules.py:1217-1292).  Either window codes (fast path) or the
// reference's int64 argmax tensor (general scatter path) selects the unpooling.
hipError_t run_up(const DevLayer &L, const float *x, int n, int h, int w, float *y,
                  const uint8_t *code, const int64_t *argmax, const LayerTemps &T, hipStream_t s, int arith = SSAL_ARITH_F32)
{
    const int C = L.cin, pf = L.f, cf = L.cf;
    if (arith == SSAL_ARITH_BF16X3 && L.bf3 && code && up_fused(L, h, w)) {  // opt-in mode; the unpool gather is the exact kernel's
        UpArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x; a.y = y; a.code = code;
        a.wp = L.proj_w; a.ps = L.proj_scale; a.pt = L.proj_shift; a.pa = L.proj_alpha;
        a.ws = L.convT_stacked; a.cs = L.conv_scale; a.ct = L.conv_shift; a.ca = L.conv_alpha;
        a.we = L.exp_w; a.es = L.exp_scale; a.et = L.exp_shift; a.wr = L.res_w; a.ra = L.res_alpha;
        a.N = n; a.H = h; a.W = w; a.dil = 1;
        return launch_upsample_bf16x3(a, L.bf3, s);
    }
    if (code && up_fused(L, h, w))
        return launch_upsample_mfma(x, y, code, n, h, w, C, L.proj_w, L.proj_scale, L.proj_shift,
                                    L.proj_alpha, L.convT_stacked, L.conv_scale, L.conv_shift,
                                    L.conv_alpha, L.exp_w, L.exp_scale, L.exp_shift, L.res_w,
                                    L.res_alpha, s);
    ConvArgs p = conv_args(x, n, h, w, C, L.proj_w, 1, 1, pf, 1, 1, T.t0);
    p.scale = L.proj_scale; p.shift = L.proj_shift; p.alpha = L.proj_alpha;
    HIP_RET(launch_conv(p, s));
    HIP_RET(launch_convT(T.t0, n, h, w, pf, L.conv_w, cf, L.conv_scale, L.conv_shift, L.conv_alpha,
                         T.t1, s));
    ConvArgs r = conv_args(x, n, h, w, C, L.res_w, 1, 1, L.cout, 1, 1, T.t2);  // no BN (:1285-1287)
    HIP_RET(launch_conv(r, s));
    ConvArgs e = conv_args(T.t1, n, 2 * h, 2 * w, cf, L.exp_w, 1, 1, L.cout, 1, 1, y);
    e.scale = L.exp_scale; e.shift = L.exp_shift; e.res_alpha = L.res_alpha; e.res_C = L.cout;
    if (code) {
        e.res_mode = RES_UNPOOL; e.res = T.t2; e.code_in = code;
    } else {
        HIP_RET(launch_unpool_scatter(T.t2, argmax, n, h, w, L.cout, 0, T.t3, s));
        e.res_mode = RES_ADD; e.res = T.t3;
    }
    return launch_conv(e, s);
}

struct NetWorkspace {
    float *a0, *a1;        // [n,h/2,w/2,16]
    float *s1a, *s1b;      // [n,h/4,w/4,64]
    float *s2a, *s2b;      // [n,h/8,w/8,128]
    LayerTemps T;
    int64_t t_img[3];      // floats of T.t0 / t1 / t2 per image (every temporary is sized in proportion to n)
    uint8_t *code1, *code2;
    double *partial;
    int64_t bytes;
    bool ok;
};

NetWorkspace carve(const ssal_enet *net, void *ws, int64_t ws_bytes, int64_t n, int64_t h, int64_t w)
{
    Bump b(ws, ws_bytes);
    NetWorkspace W;
    W.a0 = b.take<float>(n * (h / 2) * (w / 2) * 16);
    W.a1 = b.take<float>(n * (h / 2) * (w / 2) * 16);
    W.s1a = b.take<float>(n * (h / 4) * (w / 4) * 64);
    W.s1b = b.take<float>(n * (h / 4) * (w / 4) * 64);
    W.s2a = b.take<float>(n * (h / 8) * (w / 8) * 128);
    W.s2b = b.take<float>(n * (h / 8) * (w / 8) * 128);
    int64_t mx[4] = {0, 0, 0, 0};
    int64_t lh = h, lw = w;
    for (int li = 0; li < kNumLayers; ++li) {
        const DevLayer &L = net->layers[li];
        int64_t sz[4];
        layer_temp_floats(L, n, lh, lw, false, sz);
        for (int k = 0; k < 4; ++k) mx[k] = sz[k] > mx[k] ? sz[k] : mx[k];
        if (L.kind == K_INITIAL || L.kind == K_DOWN) { lh /= 2; lw /= 2; }
        else if (L.kind == K_UP) { lh *= 2; lw *= 2; }
    }
    W.T.t0 = b.take<float>(mx[0]);
    W.T.t1 = b.take<float>(mx[1]);
    W.T.t2 = b.take<float>(mx[2]);
    W.T.t3 = nullptr;
    for (int k = 0; k < 3; ++k) W.t_img[k] = mx[k] / n;
    W.code1 = b.take<uint8_t>(n * (h / 4) * (w / 4) * 16);
    W.code2 = b.take<uint8_t>(n * (h / 8) * (w / 8) * 64);
    W.partial = b.take<double>(n * (int64_t)final_score_blocks((int)(h / 2), (int)(w / 2)));
    W.bytes = b.off;
    W.ok = b.ok;
    return W;
}

// every entry point that launches against the handle's weight arena: the arena lives on ONE device
int check_device(const ssal_enet *net)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != net->device)
        return fail(SSAL_ESTATE, "the handle was committed on device %d but the current device is %d (one handle per device)",
                    net->device, dev);
    return SSAL_OK;
}

int check_dims(const ssal_enet *net, int n, int h, int w)
{
    if (!net) return fail(SSAL_EINVAL, "net is NULL");
    if (!net->committed) return fail(SSAL_ESTATE, "ssal_enet_commit() has not been called");
    if (int rc = check_device(net)) return rc;
    if (n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d", n, h, w);
    if (h % 8 || w % 8)
        return fail(SSAL_EINVAL, "ENet needs H and W divisible by 8 (got %dx%d)", h, w);
    if ((int64_t)n * h * w > ((int64_t)1 << 34))
        return fail(SSAL_EINVAL, "batch too large (n=%d h=%d w=%d): split it", n, h, w);
    return SSAL_OK;
}

// layer li (0 = Initial .. 27 = Bottleneck5_1) on the fixed buffer plan of the workspace: Initial -> a0; 1_0: a0 -> s1a;
// 1_1..1_4 ping-pong s1a / s1b; 2_0: s1a -> s2a; 2_1..3_8 ping-pong s2a / s2b (ends in s2a); 4_0: s2a -> s1a; 4_1, 4_2
// ping-pong; 5_0: s1a -> a0; 5_1: a0 -> a1.  V holds the buffers of the images this call works on.
hipError_t run_layer_idx(const ssal_enet *net, int li, const void *x, bool x_is_u8, const NetWorkspace &V, int n, int h,
                         int w, hipStream_t s, int arith = SSAL_ARITH_F32)
{
    const DevLayer &L = net->layers[li];
    // Initial + Bottleneck1_0 in one launch: Initial's output (a0; no endpoint) is never written
    // (the image window is read in quads of four elements there: a frame pointer that is not aligned to a quad -- 16 bytes for
    // float32 frames, 4 for uint8 frames -- takes the two-launch form)
    const bool fuse01 = (ssal::knobs().fuse_ends & 1) && g_use_mfma && initial_down16_supported(net->c_in) &&
                        (long)h * w * 16 < (1L << 31) && ((uintptr_t)x & (x_is_u8 ? 3 : 15)) == 0;
    if (li == 0) return fuse01 ? hipSuccess : launch_initial(x, x_is_u8, n, h, w, net->c_in, L.w, L.scale, L.shift, L.alpha, V.a0, s);
    if (li == 1 && fuse01) {
        const DevLayer &I = net->layers[0];
        return launch_initial_down16(x, x_is_u8, n, h, w, net->c_in, I.w, I.scale, I.shift, I.alpha, V.s1a, V.code1,
                                     L.proj_w, L.proj_scale, L.proj_shift, L.proj_alpha, L.conv_w, L.conv_scale, L.conv_shift,
                                     L.conv_alpha, L.exp_w, L.exp_scale, L.exp_shift, L.res_alpha, s);
    }
    if (li == 1) return run_down(L, V.a0, n, h / 2, w / 2, V.s1a, V.code1, V.T, s);
    if (li <= 5) return run_regular(L, (li - 2) % 2 == 0 ? V.s1a : V.s1b, n, h / 4, w / 4, (li - 2) % 2 == 0 ? V.s1b : V.s1a, V.T, s);
    if (li == 6) return run_down(L, V.s1a, n, h / 4, w / 4, V.s2a, V.code2, V.T, s, arith);
    if (li <= 22) return run_regular(L, (li - 7) % 2 == 0 ? V.s2a : V.s2b, n, h / 8, w / 8, (li - 7) % 2 == 0 ? V.s2b : V.s2a, V.T, s, arith);
    if (li == 23) return run_up(L, V.s2a, n, h / 8, w / 8, V.s1a, V.code2, nullptr, V.T, s, arith);
    if (li <= 25) return run_regular(L, li == 24 ? V.s1a : V.s1b, n, h / 4, w / 4, li == 24 ? V.s1b : V.s1a, V.T, s);
    if (li == 26) return run_up(L, V.s1a, n, h / 4, w / 4, V.a0, V.code1, nullptr, V.T, s);
    return run_regular(L, V.a0, n, h / 2, w / 2, V.a1, V.T, s);
}

// Final (transposed conv) fused with the score (k_final_score): outputs of the images [i0, i0 + n)
struct FinalOut {
    float *logits;
    int measure;
    float threshold;
    uint8_t *label, *mask;
    float *conf;
    int arith = SSAL_ARITH_F32;  // arithmetic mode of the whole call (rides here: every run_net caller builds one)
};

// the ranking pass (no logits / label / mask / confidence output) evaluates Bottleneck5_1 inside the Final + score kernel:
// its output (an endpoint of the FORWARD path only) is neither written nor read
bool fuse_5_1(const ssal_enet *net, const FinalOut &f)
{
    const DevLayer &L = net->layers[27];
    return (ssal::knobs().fuse_ends & 2) && g_use_mfma && !f.logits && !f.label && !f.mask && !f.conf && L.cin == 16 &&
           L.f == 4 && L.dil == 1 && !L.asym;
}

hipError_t run_final(const ssal_enet *net, const NetWorkspace &V, const FinalOut &f, long i0, int n, int h, int w,
                     hipStream_t s)
{
    const long px = (long)h * w;
    if (fuse_5_1(net, f)) {
        const DevLayer &L = net->layers[27];
        return launch_bnk4_final_score(V.a0, n, h / 2, w / 2, L.proj_w, L.proj_scale, L.proj_shift, L.proj_alpha, L.conv_w,
                                       L.conv_scale, L.conv_shift, L.conv_alpha, L.exp_w, L.exp_scale, L.exp_shift,
                                       L.res_alpha, net->layers[kNumLayers - 1].w, net->classes, f.measure, V.partial, s);
    }
    return launch_final_score(V.a1, n, h / 2, w / 2, net->layers[kNumLayers - 1].w, net->classes,
                              f.logits ? f.logits + i0 * px * net->classes : nullptr, f.measure, f.threshold, V.partial,
                              f.label ? f.label + i0 * px : nullptr, f.mask ? f.mask + i0 * px : nullptr,
                              f.conf ? f.conf + i0 * px : nullptr, s);
}

// runs Initial .. Bottleneck5_1 and Final + score (per-block float64 partials in W.partial)
hipError_t run_net(const ssal_enet *net, const void *x, bool x_is_u8, int n, int h, int w, NetWorkspace &W,
                   const FinalOut &fin, hipStream_t s)
{
    // image-group schedule: images are independent, so a span of layers runs as G chains of ~n / G images on G
    // library-owned side streams (fork / join with events; the caller's stream order is kept).  Launches of different
    // chains overlap on the chip: the tail of one launch (the last, partly filled round of workgroups) and the load-only
    // head of the next are covered by the other chain's workgroups.  Measured on one box, batch 8, 40-60 steps, identical
    // bits in every row (profiles/r03_ab_image_group_streams.txt): one stream 2062 images/s; stages 2 + 3 in two chains
    // 2087-2125; Bottleneck1_0..5_1 2126-2149; Initial..5_1 2154-2160; Initial..Final + score 2167-2174 (+5.2 %, the
    // default); four chains 1897-1915 (launches too small); an extra one-launch offset between two chains: no change.
    const ssal::Knobs &kn = ssal::knobs();
    int G = kn.img_groups, ga = 7, gb = 23;  // span 0: Bottleneck2_1 .. Bottleneck3_8
    if (kn.img_span == 1) ga = 6;
    else if (kn.img_span == 2) { ga = 1; gb = 28; }
    else if (kn.img_span == 3) { ga = 0; gb = 28; }
    else if (kn.img_span == 4) { ga = 0; gb = 29; }  // layer 28 = Final + score
    if (G < 2 || G > 8 || n < G || !g_use_mfma || ssal::prof_enabled()) G = 1;
    NetWorkspace V[8];
    int first[9];  // group g = images [first[g], first[g + 1]): as even as n allows (a pool's last batch may be odd)
    for (int g = 0; g <= G; ++g) first[g] = (int)((long)g * n / G);
    const size_t xelt = x_is_u8 ? 1 : 4;
    for (int g = 0; g < G && G > 1; ++g) {
        const long i0 = first[g];
        V[g] = W;
        V[g].a0 += i0 * (h / 2) * (w / 2) * 16;  V[g].a1 += i0 * (h / 2) * (w / 2) * 16;
        V[g].s1a += i0 * (h / 4) * (w / 4) * 64; V[g].s1b += i0 * (h / 4) * (w / 4) * 64;
        V[g].s2a += i0 * (h / 8) * (w / 8) * 128; V[g].s2b += i0 * (h / 8) * (w / 8) * 128;
        // the temporaries of the generic (multi-launch) layer forms: a layer whose size guard sends it there must not
        // share them with the other chains
        V[g].T.t0 += i0 * W.t_img[0]; V[g].T.t1 += i0 * W.t_img[1]; V[g].T.t2 += i0 * W.t_img[2];
        V[g].code1 += i0 * (h / 4) * (w / 4) * 16; V[g].code2 += i0 * (h / 8) * (w / 8) * 64;
        V[g].partial += i0 * final_score_blocks(h / 2, w / 2);
    }
    // fork / join events are private to this call (ssal::ChainSet): two host threads may score on one handle, each on
    // its own stream and workspace
    ssal::ChainSet cs;
    auto issue = [&](int li, int g) -> hipError_t {  // layer li of chain g (g < 0: the whole batch on the caller's stream)
        const NetWorkspace &Vg = g < 0 ? W : V[g];
        const int i0 = g < 0 ? 0 : first[g], ng = g < 0 ? n : first[g + 1] - first[g];
        hipStream_t sg = g < 0 ? s : cs.side[g];
        if (li == 28) return run_final(net, Vg, fin, i0, ng, h, w, sg);
        if (li == 27 && fuse_5_1(net, fin)) return hipSuccess;
        return run_layer_idx(net, li, (const char *)x + (size_t)i0 * h * w * net->c_in * xelt, x_is_u8, Vg, ng, h, w, sg, fin.arith);
    };
    if (G == 1) {
        for (int li = 0; li < 29; ++li) HIP_RET(issue(li, -1));
        return hipSuccess;
    }
    for (int li = 0; li < ga; ++li) HIP_RET(issue(li, -1));
    HIP_RET(cs.begin(G, s));
    // layer-major issue order; with img_lag = L > 0 chain g runs L layers behind chain g - 1 (it starts when its
    // predecessor has finished its first L layers): measured, see the table above.  An error inside the span still joins
    // the chains (~ChainSet) before the call returns.
    const int lag = kn.img_lag > 0 ? kn.img_lag : 0, nl = gb - ga;
    for (int t = 0; t < nl + lag * (G - 1); ++t)
        for (int g = 0; g < G; ++g) {
            const int k = t - lag * g;
            if (k < 0 || k >= nl) continue;
            HIP_RET(issue(ga + k, g));
            if (lag > 0 && k == lag - 1 && g + 1 < G) HIP_RET(cs.link(g, g + 1));
        }
    HIP_RET(cs.end());
    for (int li = gb; li < 29; ++li) HIP_RET(issue(li, -1));
    return hipSuccess;
}
}  // namespace

SSAL_API int64_t ssal_enet_workspace_bytes(const ssal_enet *net, int n, int h, int w)
{
    if (!net || !net->committed || n <= 0 || h <= 0 || w <= 0) return -1;
    NetWorkspace W = carve(net, nullptr, 0, n, h, w);
    return W.bytes + 256;
}

static int check_arith(int arithmetic)
{
    if (arithmetic != SSAL_ARITH_F32 && arithmetic != SSAL_ARITH_BF16X3)
        return fail(SSAL_EINVAL, "arithmetic must be SSAL_ARITH_F32 (0) or SSAL_ARITH_BF16X3 (1), got %d", arithmetic);
    return SSAL_OK;
}

static int forward_any(ssal_enet *net, const void *x_dev, bool x_is_u8, int n, int h, int w, float *logits_dev,
                       void *ws_dev, int64_t ws_bytes, void *stream, int arith = SSAL_ARITH_F32)
{
    int rc = check_dims(net, n, h, w);
    if (rc) return rc;
    if ((rc = check_arith(arith))) return rc;
    if (!x_dev || !logits_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    NetWorkspace W = carve(net, ws_dev, ws_bytes, n, h, w);
    if (!W.ok) return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes, got %lld",
                           (long long)W.bytes, (long long)ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    // logits only: the score outputs of the fused kernel go to the scratch partial buffer
    FinalOut fin = {logits_dev, SSAL_MEASURE_CONFIDENCE, 0.0f, nullptr, nullptr, nullptr};
    fin.arith = arith;
    HIP_TRY(run_net(net, x_dev, x_is_u8, n, h, w, W, fin, s));
    return SSAL_OK;
}

SSAL_API int ssal_enet_forward_nhwc(ssal_enet *net, const float *x_dev, int n, int h, int w,
                                    float *logits_dev, void *ws_dev, int64_t ws_bytes, void *stream)
{
    return forward_any(net, x_dev, false, n, h, w, logits_dev, ws_dev, ws_bytes, stream);
}

SSAL_API int ssal_enet_forward_nhwc_u8(ssal_enet *net, const uint8_t *x_dev, int n, int h, int w,
                                       float *logits_dev, void *ws_dev, int64_t ws_bytes, void *stream)
{
    return forward_any(net, x_dev, true, n, h, w, logits_dev, ws_dev, ws_bytes, stream);
}

static int score_any(ssal_enet *net, const void *x_dev, bool x_is_u8, int n, int h, int w, int measure,
                     float threshold, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev, float *conf_dev,
                     void *ws_dev, int64_t ws_bytes, void *stream, int arith = SSAL_ARITH_F32)
{
    int rc = check_dims(net, n, h, w);
    if (rc) return rc;
    if ((rc = check_arith(arith))) return rc;
    if (measure < 0 || measure > 2)
        return fail(SSAL_ENOTIMPL, "Uncertainty function not implemented (measure=%d)", measure);
    if (!x_dev || !scores_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    NetWorkspace W = carve(net, ws_dev, ws_bytes, n, h, w);
    if (!W.ok) return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes, got %lld",
                           (long long)W.bytes, (long long)ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    FinalOut fin = {nullptr, measure, threshold, label_dev, mask_dev, conf_dev};
    fin.arith = arith;
    HIP_TRY(run_net(net, x_dev, x_is_u8, n, h, w, W, fin, s));
    HIP_TRY(launch_reduce_mean(W.partial, n, final_score_blocks(h / 2, w / 2), (double)h * (double)w,
                               scores_dev, s));
    return SSAL_OK;
}

SSAL_API int ssal_enet_score_nhwc(ssal_enet *net, const float *x_dev, int n, int h, int w,
                                  int measure, float threshold, double *scores_dev,
                                  uint8_t *label_dev, uint8_t *mask_dev, float *conf_dev,
                                  void *ws_dev, int64_t ws_bytes, void *stream)
{
    return score_any(net, x_dev, false, n, h, w, measure, threshold, scores_dev, label_dev, mask_dev, conf_dev,
                     ws_dev, ws_bytes, stream);
}

SSAL_API int ssal_enet_score_nhwc_u8(ssal_enet *net, const uint8_t *x_dev, int n, int h, int w,
                                     int measure, float threshold, double *scores_dev,
                                     uint8_t *label_dev, uint8_t *mask_dev, float *conf_dev,
                                     void *ws_dev, int64_t ws_bytes, void *stream)
{
    return score_any(net, x_dev, true, n, h, w, measure, threshold, scores_dev, label_dev, mask_dev, conf_dev,
                     ws_dev, ws_bytes, stream);
}

// ---- the same entry points with an explicit arithmetic mode (include/ssal_enet.h: SSAL_ARITH_*) ----
SSAL_API int ssal_enet_forward_nhwc_arith(ssal_enet *net, const void *x_dev, int x_is_u8, int n, int h, int w, int arithmetic,
                                          float *logits_dev, void *ws_dev, int64_t ws_bytes, void *stream)
{
    return forward_any(net, x_dev, x_is_u8 != 0, n, h, w, logits_dev, ws_dev, ws_bytes, stream, arithmetic);
}

SSAL_API int ssal_enet_score_nhwc_arith(ssal_enet *net, const void *x_dev, int x_is_u8, int n, int h, int w, int measure,
                                        float threshold, int arithmetic, double *scores_dev, uint8_t *label_dev,
                                        uint8_t *mask_dev, float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream)
{
    return score_any(net, x_dev, x_is_u8 != 0, n, h, w, measure, threshold, scores_dev, label_dev, mask_dev, conf_dev, ws_dev,
                     ws_bytes, stream, arithmetic);
}

// byte offsets (into the workspace passed to forward/score) of the tensors behind
// ENet.endpoint_outputs (enet.py:311-318): [0] bottleneck5_1 [n,h/2,w/2,16],
// [1] bottleneck4_2 [n,h/4,w/4,64], [2] bottleneck3_8 [n,h/8,w/8,128]; valid until the next call.
SSAL_API int ssal_enet_endpoint_offsets(const ssal_enet *net, int n, int h, int w, int64_t offs[3])
{
    if (!net || !offs) return fail(SSAL_EINVAL, "NULL argument");
    if (!net->committed) return fail(SSAL_ESTATE, "ssal_enet_commit() has not been called");
    NetWorkspace W = carve(net, (void *)256, ((int64_t)1 << 62), n, h, w);
    const char *base = (const char *)256;
    offs[0] = (const char *)W.a1 - base;
    offs[1] = (const char *)W.s1a - base;  // 4_0 -> s1a, 4_1 -> s1b, 4_2 -> s1a
    offs[2] = (const char *)W.s2a - base;  // 2_0 -> s2a, then 16 ping-pong steps end in s2a
    return SSAL_OK;
}

// The pooling indices of the last forward/score call that used this workspace, in the reference's int64 form
// (argmax1 of Bottleneck1_0: [n,h/4,w/4,16]; argmax2 of Bottleneck2_0: [n,h/8,w/8,64]; enet.py:331,338).
SSAL_API int ssal_enet_export_argmax(const ssal_enet *net, const void *ws_dev, int64_t ws_bytes, int n, int h, int w,
                                     int which, int64_t *argmax_out_dev, void *stream)
{
    int rc = check_dims(net, n, h, w);
    if (rc) return rc;
    if (!ws_dev || !argmax_out_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (which != 1 && which != 2) return fail(SSAL_EINVAL, "which must be 1 (argmax1) or 2 (argmax2), got %d", which);
    NetWorkspace W = carve(net, const_cast<void *>(ws_dev), ws_bytes, n, h, w);
    if (!W.ok) return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes, got %lld", (long long)W.bytes,
                           (long long)ws_bytes);
    if (which == 1)
        HIP_TRY(launch_codes_to_argmax(W.code1, n, h / 4, w / 4, 16, argmax_out_dev, (hipStream_t)stream));
    else
        HIP_TRY(launch_codes_to_argmax(W.code2, n, h / 8, w / 8, 64, argmax_out_dev, (hipStream_t)stream));
    return SSAL_OK;
}

// ---- single layer -------------------------------------------------------------------------------
static int find_layer(const ssal_enet *net, const char *layer)
{
    for (int li = 0; li < kNumLayers; ++li)
        if (!strcmp(kSpecs[li].name, layer)) return li;
    (void)net;
    return -1;
}

SSAL_API int64_t ssal_enet_layer_workspace_bytes(const ssal_enet *net, const char *layer, int n,
                                                 int h, int w)
{
    if (!net || !net->committed || !layer) return -1;
    int li = find_layer(net, layer);
    if (li < 0) return -1;
    const DevLayer &L = net->layers[li];
    int64_t sz[4];
    layer_temp_floats(L, n, h, w, true, sz);
    int64_t bytes = 1024;
    for (int k = 0; k < 4; ++k) bytes += sz[k] * 4 + 256;
    if (L.kind == K_DOWN) bytes += (int64_t)n * (h / 2) * (w / 2) * L.cin + 256;
    if (L.kind == K_UP) bytes += (int64_t)n * h * w * L.cout + 1024;
    if (L.kind == K_FINAL) bytes += (int64_t)n * final_score_blocks(h, w) * 8 + 256;
    return bytes;
}

static int run_layer_any(ssal_enet *net, const char *layer, const float *x_dev, int n, int h, int w, float *y_dev,
                         int64_t *argmax_out_dev, const int64_t *argmax_in_dev, void *ws_dev, int64_t ws_bytes, void *stream,
                         int arith);

SSAL_API int ssal_enet_run_layer(ssal_enet *net, const char *layer, const float *x_dev, int n, int h,
                                 int w, float *y_dev, int64_t *argmax_out_dev,
                                 const int64_t *argmax_in_dev, void *ws_dev, int64_t ws_bytes,
                                 void *stream)
{
    return run_layer_any(net, layer, x_dev, n, h, w, y_dev, argmax_out_dev, argmax_in_dev, ws_dev, ws_bytes, stream, SSAL_ARITH_F32);
}

SSAL_API int ssal_enet_run_layer_arith(ssal_enet *net, const char *layer, const float *x_dev, int n, int h, int w, int arithmetic,
                                       float *y_dev, int64_t *argmax_out_dev, const int64_t *argmax_in_dev, void *ws_dev,
                                       int64_t ws_bytes, void *stream)
{
    if (int rc = check_arith(arithmetic)) return rc;
    return run_layer_any(net, layer, x_dev, n, h, w, y_dev, argmax_out_dev, argmax_in_dev, ws_dev, ws_bytes, stream, arithmetic);
}

static int run_layer_any(ssal_enet *net, const char *layer, const float *x_dev, int n, int h, int w, float *y_dev,
                         int64_t *argmax_out_dev, const int64_t *argmax_in_dev, void *ws_dev, int64_t ws_bytes, void *stream,
                         int arith)
{
    if (!net || !layer) return fail(SSAL_EINVAL, "NULL argument");
    if (!net->committed) return fail(SSAL_ESTATE, "ssal_enet_commit() has not been called");
    if (int rc = check_device(net)) return rc;
    if (!x_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d", n, h, w);
    int li = find_layer(net, layer);
    if (li < 0) return fail(SSAL_EINVAL, "unknown layer '%s'", layer);
    const DevLayer &L = net->layers[li];
    if ((L.kind == K_INITIAL || L.kind == K_DOWN) && (h % 2 || w % 2))
        return fail(SSAL_EINVAL, "layer '%s' needs even H and W (got %dx%d)", layer, h, w);
    const int64_t need = ssal_enet_layer_workspace_bytes(net, layer, n, h, w);
    if (need > 1024 && (!ws_dev || ws_bytes < need))
        return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes, got %lld", (long long)need,
                    (long long)ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws_dev, ws_bytes);
    int64_t sz[4];
    layer_temp_floats(L, n, h, w, true, sz);
    LayerTemps T;
    T.t0 = b.take<float>(sz[0]);
    T.t1 = b.take<float>(sz[1]);
    T.t2 = b.take<float>(sz[2]);
    T.t3 = b.take<float>(sz[3]);
    switch (L.kind) {
    case K_INITIAL:
        HIP_TRY(launch_initial(x_dev, false, n, h, w, net->c_in, L.w, L.scale, L.shift, L.alpha, y_dev, s));
        break;
    case K_REGULAR:
        HIP_TRY(run_regular(L, x_dev, n, h, w, y_dev, T, s, arith));
        break;
    case K_DOWN: {
        uint8_t *code = b.take<uint8_t>((int64_t)n * (h / 2) * (w / 2) * L.cin);
        HIP_TRY(run_down(L, x_dev, n, h, w, y_dev, code, T, s, arith));
        if (argmax_out_dev)
            HIP_TRY(launch_codes_to_argmax(code, n, h / 2, w / 2, L.cin, argmax_out_dev, s));
        break;
    }
    case K_UP: {
        if (!argmax_in_dev) return fail(SSAL_EINVAL, "layer '%s' needs argmax_in_dev", layer);
        // indices produced by a 2x2/s2 pooling always lie in their own window: then the 1-byte window
        // code form (gather unpool, fused kernels) is exact; arbitrary indices take the scatter form
        uint8_t *code = b.take<uint8_t>((int64_t)n * h * w * L.cout);
        int *bad = b.take<int>(1);
        HIP_TRY(launch_argmax_to_codes(argmax_in_dev, n, h, w, L.cout, code, bad, s));
        int bad_host = 0;
        HIP_TRY(hipMemcpyAsync(&bad_host, bad, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (bad_host == 0)
            HIP_TRY(run_up(L, x_dev, n, h, w, y_dev, code, nullptr, T, s, arith));
        else
            HIP_TRY(run_up(L, x_dev, n, h, w, y_dev, nullptr, argmax_in_dev, T, s));
        break;
    }
    case K_FINAL: {
        double *partial = b.take<double>((int64_t)n * final_score_blocks(h, w));
        HIP_TRY(launch_final_score(x_dev, n, h, w, L.w, net->classes, y_dev,
                                   SSAL_MEASURE_CONFIDENCE, 0.0f, partial, nullptr, nullptr,
                                   nullptr, s));
        break;
    }
    }
    return SSAL_OK;
}

// ------------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------------
SSAL_API int64_t ssal_score_workspace_bytes(int n, int h, int w)
{
    if (n <= 0 || h <= 0 || w <= 0) return -1;
    return (int64_t)n * score_blocks(h, w) * 8 + 256;
}

SSAL_API int ssal_score_logits_nhwc(const float *logits_dev, int n, int h, int w, int classes,
                                    int measure, float threshold, double *scores_dev,
                                    uint8_t *label_dev, uint8_t *mask_dev, float *conf_dev,
                                    void *ws_dev, int64_t ws_bytes, void *stream)
{
    if (measure < 0 || measure > 2)
        return fail(SSAL_ENOTIMPL, "Uncertainty function not implemented (measure=%d)", measure);
    if (n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d", n, h, w);
    if (classes < 2 || classes > 32) return fail(SSAL_EINVAL, "classes must be in [2,32] (got %d)", classes);
    if (!logits_dev || !scores_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (ws_bytes < ssal_score_workspace_bytes(n, h, w))
        return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes, got %lld",
                    (long long)ssal_score_workspace_bytes(n, h, w), (long long)ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws_dev, ws_bytes);
    double *partial = b.take<double>((int64_t)n * score_blocks(h, w));
    HIP_TRY(launch_score_logits(logits_dev, n, h, w, classes, measure, threshold, partial, label_dev,
                                mask_dev, conf_dev, s));
    HIP_TRY(launch_reduce_mean(partial, n, score_blocks(h, w), (double)h * (double)w, scores_dev, s));
    return SSAL_OK;
}

SSAL_API int64_t ssal_xent_workspace_bytes(int h, int w)
{
    if (h <= 0 || w <= 0) return -1;
    return (int64_t)xent_blocks(h, w) * 16 + 256;
}

SSAL_API int ssal_masked_softmax_cross_entropy(const float *logits_dev, const uint8_t *labels_dev,
                                               const float *mask_dev, int n, int h, int w, int classes,
                                               float weight, float label_smoothing, double *loss_dev,
                                               void *ws_dev, int64_t ws_bytes, void *stream)
{
    if (!logits_dev || !labels_dev || !mask_dev || !loss_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d", n, h, w);
    if (classes < 2 || classes > 32) return fail(SSAL_EINVAL, "classes must be in [2,32] (got %d)", classes);
    if (ws_bytes < ssal_xent_workspace_bytes(h, w))
        return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes", (long long)ssal_xent_workspace_bytes(h, w));
    Bump b(ws_dev, ws_bytes);
    double *partial = b.take<double>(2 * (int64_t)xent_blocks(h, w));
    HIP_TRY(launch_masked_xent(logits_dev, labels_dev, mask_dev, n, h, w, classes, weight,
                               label_smoothing, partial, loss_dev, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_max_pool_with_argmax_2x2(const float *x_dev, int n, int h, int w, int c,
                                           float *y_dev, int64_t *argmax_dev, int include_batch,
                                           void *stream)
{
    if (!x_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || h % 2 || w % 2)
        return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d c=%d (H, W must be even)", n, h, w, c);
    HIP_TRY(launch_maxpool_argmax(x_dev, n, h, w, c, y_dev, argmax_dev, include_batch, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_unpool_2d(const float *x_dev, const int64_t *idx_dev, int n, int h, int w, int c,
                            int idx_has_batch, float *y_dev, void *stream)
{
    if (!x_dev || !idx_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return fail(SSAL_EINVAL, "bad dims");
    HIP_TRY(launch_unpool_scatter(x_dev, idx_dev, n, h, w, c, idx_has_batch, y_dev, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_prelu(const float *x_dev, int64_t pixels, int c, const float *alpha_dev,
                        float *y_dev, void *stream)
{
    if (!x_dev || !alpha_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (pixels <= 0 || c <= 0) return fail(SSAL_EINVAL, "bad dims");
    HIP_TRY(launch_prelu(x_dev, pixels, c, alpha_dev, y_dev, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_spatial_dropout(const float *x_dev, int n, int64_t pixels_per_image, int c, float rate,
                                  uint64_t seed, float *y_dev, void *stream)
{
    if (!x_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || pixels_per_image <= 0 || c <= 0) return fail(SSAL_EINVAL, "bad dims");
    if (!(rate >= 0.0f && rate < 1.0f)) return fail(SSAL_EINVAL, "rate must be in [0, 1) (got %g)", (double)rate);
    HIP_TRY(launch_spatial_dropout(x_dev, n, pixels_per_image, c, rate, seed, y_dev, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_batch_norm_inference(const float *x_dev, int64_t pixels, int c,
                                       const float *mean_dev, const float *var_dev,
                                       const float *gamma_dev, const float *beta_dev, float *y_dev,
                                       void *stream)
{
    if (!x_dev || !mean_dev || !var_dev || !gamma_dev || !beta_dev || !y_dev)
        return fail(SSAL_EINVAL, "NULL device pointer");
    if (pixels <= 0 || c <= 0) return fail(SSAL_EINVAL, "bad dims");
    hipStream_t s = (hipStream_t)stream;
    float *fold = nullptr;
    HIP_TRY(hipMallocAsync((void **)&fold, sizeof(float) * 2 * c, s));
    hipError_t e = launch_bn_fold(mean_dev, var_dev, gamma_dev, beta_dev, c, fold, fold + c, s);
    if (e == hipSuccess) e = launch_affine(x_dev, pixels, c, fold, fold + c, y_dev, s);
    hipError_t e2 = hipFreeAsync(fold, s);
    if (e != hipSuccess) return fail(SSAL_EHIP, "batch_norm launch failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(SSAL_EHIP, "hipFreeAsync failed: %s", hipGetErrorString(e2));
    return SSAL_OK;
}

SSAL_API int ssal_conv2d_same(const float *x_dev, int n, int h, int w, int cin,
                              const float *kernel_dev, int kh, int kw, int cout, int stride,
                              int dilation, float *y_dev, void *stream)
{
    if (!x_dev || !kernel_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || dilation <= 0)
        return fail(SSAL_EINVAL, "bad dims");
    if (cin % 4 || cout <= 0 || cout > 256 || 256 % cout)
        return fail(SSAL_EINVAL, "conv2d needs cin %% 4 == 0 and cout a power of two <= 256 (cin=%d cout=%d)", cin, cout);
    ConvArgs a = conv_args(x_dev, n, h, w, cin, kernel_dev, kh, kw, cout, stride, dilation, y_dev);
    HIP_TRY(launch_conv(a, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_conv2d_transpose_3x3_s2(const float *x_dev, int n, int h, int w, int cin,
                                          const float *kernel_dev, int cout, float *y_dev,
                                          void *stream)
{
    if (!x_dev || !kernel_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad dims");
    if (cin % 4 || cout <= 0 || cout > 256 || 256 % cout)
        return fail(SSAL_EINVAL, "conv2d_transpose needs cin %% 4 == 0 and cout a power of two <= 256 (cin=%d cout=%d)", cin, cout);
    // the kernel arrives in TF layout [3,3,cout,cin]; re-lay it out on the device copy
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> hw((size_t)9 * cout * cin);
    HIP_TRY(hipMemcpyAsync(hw.data(), kernel_dev, hw.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<float> wt = hwoi_to_hwio(hw, cout, cin);
    float *dw = nullptr;
    HIP_TRY(hipMalloc((void **)&dw, wt.size() * sizeof(float)));
    hipError_t e = hipMemcpyAsync(dw, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_convT(x_dev, n, h, w, cin, dw, cout, nullptr, nullptr, nullptr, y_dev, s);
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(SSAL_EHIP, "conv2d_transpose failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(SSAL_EHIP, "stream sync failed: %s", hipGetErrorString(e2));
    return SSAL_OK;
}

SSAL_API int ssal_resize_bilinear(const float *x_dev, int n, int h, int w, int c, int oh, int ow,
                                  float *y_dev, void *stream)
{
    if (!x_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || oh <= 0 || ow <= 0) return fail(SSAL_EINVAL, "bad dims");
    HIP_TRY(launch_resize_bilinear(x_dev, n, h, w, c, oh, ow, y_dev, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_synth_frames_nhwc(uint64_t seed, int64_t first_frame, int count, int h, int w,
                                    int c, float *out_dev, void *stream)
{
    if (!out_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (count <= 0 || h <= 0 || w <= 0 || c <= 0 || h % 8 || w % 8)
        return fail(SSAL_EINVAL, "bad dims count=%d h=%d w=%d c=%d (H, W must be divisible by 8)", count, h, w, c);
    HIP_TRY(launch_synth_frames(seed, first_frame, count, h, w, c, out_dev, false, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_synth_frames_nhwc_u8(uint64_t seed, int64_t first_frame, int count, int h, int w,
                                       int c, uint8_t *out_dev, void *stream)
{
    if (!out_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (count <= 0 || h <= 0 || w <= 0 || c <= 0 || h % 8 || w % 8)
        return fail(SSAL_EINVAL, "bad dims count=%d h=%d w=%d c=%d (H, W must be divisible by 8)", count, h, w, c);
    HIP_TRY(launch_synth_frames(seed, first_frame, count, h, w, c, out_dev, true, (hipStream_t)stream));
    return SSAL_OK;
}

// ------------------------------------------------------------------------------------------------
// per-kernel timing (HIP events on the launch stream); see ssal_prof.h
// ------------------------------------------------------------------------------------------------
namespace ssal {
namespace {
struct ProfRec {
    const char *name;
    hipEvent_t a, b;
    double flops, bytes;
};
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t get_event()
{
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool prof_enabled() { return g_prof_on; }
bool mfma_family() { return ::g_use_mfma; }

void prof_begin(const char *kernel, double flops, double bytes, hipStream_t s)
{
    ProfRec r{kernel, get_event(), get_event(), flops, bytes};
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}

void prof_end(hipStream_t s)
{
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, s);
}
}  // namespace ssal

namespace ssal {
namespace {
struct DevPool {
    hipStream_t side[8] = {};
    std::vector<hipEvent_t> events;
};
std::mutex g_pool_mu;
std::map<int, DevPool> g_pools;  // keyed by device ordinal: a stream / event belongs to the device it was created on
}  // namespace

hipError_t side_stream(int g, hipStream_t *out)
{
    if (g < 0 || g >= 8) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    DevPool &p = g_pools[dev];
    if (!p.side[g]) {
        e = hipStreamCreateWithFlags(&p.side[g], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = p.side[g];
    return hipSuccess;
}

static hipError_t take_event(int dev, hipEvent_t *out)
{
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        DevPool &p = g_pools[dev];
        if (!p.events.empty()) {
            *out = p.events.back();
            p.events.pop_back();
            return hipSuccess;
        }
    }
    return hipEventCreateWithFlags(out, hipEventDisableTiming);
}

static void give_event(int dev, hipEvent_t ev)
{
    if (!ev) return;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pools[dev].events.push_back(ev);
}

hipError_t ChainSet::begin(int groups, hipStream_t s)
{
    if (groups < 1 || groups > 8 || open) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    G = groups;
    caller = s;
    for (int g = 0; g < G && e == hipSuccess; ++g) e = side_stream(g, &side[g]);
    if (e == hipSuccess) e = take_event(dev, &fork_ev);
    for (int g = 0; g < G && e == hipSuccess; ++g) e = take_event(dev, &join_ev[g]);
    if (e == hipSuccess) e = hipEventRecord(fork_ev, s);
    for (int g = 0; g < G && e == hipSuccess; ++g) e = hipStreamWaitEvent(side[g], fork_ev, 0);
    if (e != hipSuccess) {  // nothing has been launched on a side stream yet: hand the events back, no join needed
        give_event(dev, fork_ev);
        for (int g = 0; g < G; ++g) give_event(dev, join_ev[g]);
        fork_ev = nullptr;
        for (int g = 0; g < 8; ++g) join_ev[g] = nullptr;
        return e;
    }
    open = true;
    return hipSuccess;
}

hipError_t ChainSet::link(int g, int g2)
{
    if (!open || g < 0 || g >= G || g2 < 0 || g2 >= G) return hipErrorInvalidValue;
    hipError_t e = hipEventRecord(join_ev[g], side[g]);
    return e != hipSuccess ? e : hipStreamWaitEvent(side[g2], join_ev[g], 0);
}

hipError_t ChainSet::end()
{
    if (!open) return hipSuccess;
    open = false;
    int dev = 0;
    hipError_t first = hipGetDevice(&dev);
    for (int g = 0; g < G; ++g) {  // every chain is joined even if one of the calls fails
        hipError_t e = hipEventRecord(join_ev[g], side[g]);
        if (e == hipSuccess) e = hipStreamWaitEvent(caller, join_ev[g], 0);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(side[g]);  // last resort: the chain must not outlive the call
            if (first == hipSuccess) first = e;
        }
    }
    give_event(dev, fork_ev);
    for (int g = 0; g < G; ++g) give_event(dev, join_ev[g]);
    return first;
}
}  // namespace ssal

SSAL_API int ssal_set_kernel_family(int use_mfma)
{
    g_use_mfma = use_mfma != 0;
    return SSAL_OK;
}

SSAL_API int ssal_debug_probe(float *out_dev_256, void *stream)
{
    if (!out_dev_256) return fail(SSAL_EINVAL, "NULL device pointer");
    HIP_TRY(launch_probe_swap(out_dev_256, (hipStream_t)stream));
    return SSAL_OK;
}

#ifdef SSAL_MEASURE  // measurement library only (csrc/ssal_measure_api.h)
SSAL_API int ssal_debug_mfma_peak(int shape, int blocks, int iters, float *out_dev, void *stream)
{
    if (!out_dev || blocks <= 0 || iters <= 0 || (shape != 32 && shape != 16 && shape != 132 && shape != 232 && shape != 332 && shape != 432 && shape != 516 && shape != 616))
        return fail(SSAL_EINVAL, "bad arguments");
    HIP_TRY(launch_mfma_peak(shape, blocks, iters, out_dev, (hipStream_t)stream));
    return SSAL_OK;
}
#endif

SSAL_API int ssal_debug_set_knob(const char *name, int value)
{
    if (!name) return fail(SSAL_EINVAL, "NULL knob name");
    ssal::Knobs &k = ssal::knobs();
    const std::string n(name);
    if (n == "bnk_tw") k.bnk_tw = value;
    else if (n == "bnk_xcd") k.bnk_xcd = value;
    else if (n == "bnk_o4") k.bnk_o4 = value;
    else if (n == "asym_tw16") k.asym_tw16 = value != 0;
    else if (n == "bnk_qepi") k.bnk_qepi = value;
    else if (n == "img_groups") k.img_groups = value;
    else if (n == "img_span") k.img_span = value;
    else if (n == "fuse_ends") k.fuse_ends = value;
    else if (n == "img_lag") k.img_lag = value;
    else if (n == "ig_div") k.ig_div = value > 0 ? value : 0;
    else if (n == "ic_front") k.ic_front = value & 3;
    else if (n == "ic_dual") k.ic_dual = value != 0;
    else if (n == "ig_sb") k.ig_sb = value;
    else if (n == "ic_groups") k.ic_groups = value;
#ifdef SSAL_MEASURE
    else if (n == "ablate") k.ablate = value;
    else if (n == "bnk_split") k.bnk_split = value;
#endif
    else return fail(SSAL_EINVAL, "unknown knob '%s'", name);
    return SSAL_OK;
}

// {"kernel_family": 1, "bnk_tw": 0, "bnk_xcd": 1, "measure_build": 0, "defaults": 1}; bench.py prints it and
// refuses to time a library whose knobs are not at their defaults
SSAL_API int ssal_debug_get_knobs(char *json_out, int64_t cap)
{
    if (!json_out || cap < 384) return fail(SSAL_EINVAL, "json_out too small");
    const ssal::Knobs &k = ssal::knobs();
    int measure = 0, ablate = 0;
#ifdef SSAL_MEASURE
    measure = 1;
    ablate = k.ablate + 100 * k.bnk_split;  // any non-zero value makes `defaults` 0: bench.py refuses to time it as a result
#endif
    const int dflt = g_use_mfma && k.bnk_tw == 0 && k.bnk_o4 == 2 && k.bnk_xcd == 1 && k.asym_tw16 == ssal::ASYM_TW16_DEFAULT && k.bnk_qepi == ssal::BNK_QEPI_DEFAULT && k.img_groups == 2 && k.img_span == 4 && k.fuse_ends == 3 && k.img_lag == 0 && k.ig_div == 0 && k.ic_front == ssal::IC_FRONT_DEFAULT && k.ic_dual == ssal::IC_DUAL_DEFAULT && k.ig_sb == ssal::IG_SB_DEFAULT && k.ic_groups == ssal::IC_GROUPS_DEFAULT && ablate == 0 && !ssal::prof_enabled()
                     && ssal::g_trace_buf == nullptr;
    snprintf(json_out, (size_t)cap, "{\"kernel_family\": %d, \"bnk_tw\": %d, \"bnk_o4\": %d, \"bnk_xcd\": %d, \"asym_tw16\": %d, \"bnk_qepi\": %d, \"img_groups\": %d, \"img_span\": %d, \"fuse_ends\": %d, "
             "\"img_lag\": %d, \"ig_div\": %d, \"ic_front\": %d, \"ic_dual\": %d, \"ig_sb\": %d, \"ic_groups\": %d, \"ablate\": %d, \"measure_build\": %d, \"profiling\": %d, \"defaults\": %d}", g_use_mfma ? 1 : 0,
             k.bnk_tw, k.bnk_o4, k.bnk_xcd, k.asym_tw16, k.bnk_qepi, k.img_groups, k.img_span, k.fuse_ends, k.img_lag, k.ig_div, k.ic_front, k.ic_dual, k.ig_sb, k.ic_groups, ablate, measure, ssal::prof_enabled() ? 1 : 0, dflt);
    return SSAL_OK;
}

#ifdef SSAL_MEASURE  // measurement library only (csrc/ssal_measure_api.h)
SSAL_API int ssal_debug_copy_probe(int mode, const float *x_dev, float *y_dev, int n, int h, int w, int spin,
                                   void *stream)
{
    if (!x_dev || !y_dev || n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad arguments");
    HIP_TRY(launch_copy_probe(mode, x_dev, y_dev, n, h, w, spin, (hipStream_t)stream));
    return SSAL_OK;
}
#endif

SSAL_API int ssal_debug_set_trace(void *buf_dev, int64_t bytes)
{
    ssal::g_trace_buf = (unsigned long long *)buf_dev;
    ssal::g_trace_bytes = buf_dev ? (long)bytes : 0;
#ifdef SSAL_PHASE_TRACE
    return SSAL_OK;
#else
    return buf_dev ? fail(SSAL_ENOTIMPL, "this build has no phase trace (compile with -DSSAL_PHASE_TRACE)") : SSAL_OK;
#endif
}

SSAL_API int ssal_profile_enable(int on)
{
    ssal::g_prof_on = on != 0;
    return SSAL_OK;
}

// Waits for every recorded launch, aggregates per kernel name and writes a JSON object
// {"kernel": {"launches": n, "ms": total, "flops": total, "bytes": total}, ...}; clears the records.
SSAL_API int ssal_profile_collect(char *json_out, int64_t cap)
{
    if (!json_out || cap < 64) return fail(SSAL_EINVAL, "json_out too small");
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto &r : ssal::g_recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        Agg &g = agg[r.name];
        g.n += 1; g.ms += ms; g.flops += r.flops; g.bytes += r.bytes;
        ssal::g_event_pool.push_back(r.a);
        ssal::g_event_pool.push_back(r.b);
    }
    ssal::g_recs.clear();
    std::string js = "{";
    bool first = true;
    for (auto &kv : agg) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
        js += buf;
        first = false;
    }
    js += "}";
    if ((int64_t)js.size() + 1 > cap) return fail(SSAL_ENOMEM, "json_out too small (%zu needed)", js.size() + 1);
    memcpy(json_out, js.c_str(), js.size() + 1);
    return SSAL_OK;
}
