// ssal_icnet_api.hip -- the C ABI of the ICNet row (include/ssal_icnet.h): handle, weight staging, layer
// sequencing of ICNET_SPEC.md.  Host C++ only; no torch types.  gfx950 (MI355X) only.
#include "../../include/ssal_enet.h"
#include "../../include/ssal_icnet.h"
#include "ssal_host.h"
#include "ssal_icnet.h"
#include "ssal_internal.h"
#include "ssal_prof.h"

#include <math.h>

#include <map>
#include <set>
#include <string>
#include <vector>

using namespace ssal;

namespace {

// ---- topology (ICNET_SPEC.md sections 1-4) -----------------------------------------------------------
struct ConvSpec {
    std::string name;
    int k, cin, cout, stride, dil;
    bool bn;  // false: conv6_cls (bias instead)
};

struct BneckSpec {
    const char *name;
    int cin, mid, cout, stride, dil;
    bool proj;
};

const BneckSpec kBnecks[] = {
    {"conv2_1", 64, 32, 128, 1, 1, true},     {"conv2_2", 128, 32, 128, 1, 1, false},
    {"conv2_3", 128, 32, 128, 1, 1, false},   {"conv3_1", 128, 64, 256, 2, 1, true},
    {"conv3_2", 256, 64, 256, 1, 1, false},   {"conv3_3", 256, 64, 256, 1, 1, false},
    {"conv3_4", 256, 64, 256, 1, 1, false},   {"conv4_1", 256, 128, 512, 1, 2, true},
    {"conv4_2", 512, 128, 512, 1, 2, false},  {"conv4_3", 512, 128, 512, 1, 2, false},
    {"conv4_4", 512, 128, 512, 1, 2, false},  {"conv4_5", 512, 128, 512, 1, 2, false},
    {"conv4_6", 512, 128, 512, 1, 2, false},  {"conv5_1", 512, 256, 1024, 1, 4, true},
    {"conv5_2", 1024, 256, 1024, 1, 4, false}, {"conv5_3", 1024, 256, 1024, 1, 4, false},
};
const int kNumBnecks = (int)(sizeof(kBnecks) / sizeof(kBnecks[0]));
const int kStemBnecks = 4;  // conv2_1 .. conv3_1 run on the shared 1/2-image stem

std::vector<ConvSpec> conv_specs(int c_in, int classes)
{
    std::vector<ConvSpec> L;
    auto conv = [&](const std::string &n, int k, int cin, int cout, int s = 1, int d = 1, bool bn = true) {
        L.push_back({n, k, cin, cout, s, d, bn});
    };
    conv("conv1_1_3x3_s2", 3, c_in, 32, 2);
    conv("conv1_2_3x3", 3, 32, 32);
    conv("conv1_3_3x3", 3, 32, 64);
    for (int i = 0; i < kNumBnecks; ++i) {
        const BneckSpec &b = kBnecks[i];
        const std::string n = b.name;
        conv(n + "_1x1_reduce", 1, b.cin, b.mid, b.stride);
        conv(n + "_3x3", 3, b.mid, b.mid, 1, b.dil);
        conv(n + "_1x1_increase", 1, b.mid, b.cout);
        if (b.proj) conv(n + "_1x1_proj", 1, b.cin, b.cout, b.stride);
    }
    conv("conv5_4_k1", 1, 1024, 256);
    conv("conv_sub4", 3, 256, 128, 1, 2);
    conv("conv3_1_sub2_proj", 1, 256, 128);
    conv("conv_sub2", 3, 128, 128, 1, 2);
    conv("conv1_sub1", 3, c_in, 32, 2);
    conv("conv2_sub1", 3, 32, 32, 2);
    conv("conv3_sub1", 3, 32, 64, 2);
    conv("conv3_sub1_proj", 1, 64, 128);
    conv("conv6_cls", 1, 128, classes, 1, 1, false);
    return L;
}

struct ConvDev {
    ConvSpec spec;
    const float *w = nullptr;      // igemm layout (cin % 32 == 0) or raw HWIO (first convs)
    const float *scale = nullptr;  // [CoutP]
    const float *shift = nullptr;  // [CoutP]
    const float *w_hwio = nullptr; // plain HWIO copy: only the layers of the blocks in fused_bneck() (else NULL)
    const float *wq = nullptr;     // on the *_1x1_reduce layer of a fused_bneck() block: its three kernels in quad layout (bnk_quad_layout)
};

// identity-shortcut bottlenecks with ENet's stage-2 shape (128 -> 32 -> 3x3 -> 128 on the 1/8-resolution map): the score
// path runs them as ONE launch of ENet's fused bottleneck kernel (k_bottleneck_mfma, PReLU slopes = 0 == ReLU); the
// three 1x1-layer launches of the per-layer path read / write 302 MB each at that size and are HBM-bound
bool fused_bneck(const BneckSpec &b) { return !b.proj && b.cin == 128 && b.mid == 32 && b.cout == 128 && b.stride == 1; }

// one materialised activation tensor: spatial divisor relative to the input and channel count
struct ActSpec {
    std::string name;
    int div, c;
};

}  // namespace

struct ssal_icnet {
    int c_in = 3, classes = 19;
    std::vector<ConvSpec> specs;
    std::vector<HostTensor> tensors;
    std::map<std::string, int> index;
    std::map<std::string, ConvDev> convs;
    std::vector<ActSpec> acts;
    float *arena = nullptr;
    size_t arena_floats = 0;
    const float *zeros128 = nullptr;  // PReLU slopes of the fused bottleneck launches (slope 0 == ReLU)
    bool committed = false;
    int device = -1;  // the device the arena lives on (set by commit); calls on another device are refused
};

namespace {

void add_tensor(ssal_icnet *h, const std::string &name, std::vector<int64_t> dims)
{
    HostTensor t;
    t.name = name;
    t.dims = std::move(dims);
    h->index[name] = (int)h->tensors.size();
    h->tensors.push_back(std::move(t));
}

void declare(ssal_icnet *h)
{
    h->specs = conv_specs(h->c_in, h->classes);
    for (const ConvSpec &s : h->specs) {
        add_tensor(h, s.name + ".kernel", {s.k, s.k, s.cin, s.cout});
        if (s.bn) {
            add_tensor(h, s.name + ".mean", {s.cout});
            add_tensor(h, s.name + ".variance", {s.cout});
            add_tensor(h, s.name + ".gamma", {s.cout});
            add_tensor(h, s.name + ".beta", {s.cout});
        } else {
            add_tensor(h, s.name + ".bias", {s.cout});
        }
    }
    // materialised activations, in execution order (each keeps its own buffer: nothing is aliased, so every
    // ICNET_SPEC layer output of the last call can be inspected through ssal_icnet_endpoint_info)
    auto act = [&](const std::string &n, int div, int c) { h->acts.push_back({n, div, c}); };
    act("conv1_1_3x3_s2", 4, 32);
    act("conv1_2_3x3", 4, 32);
    act("conv1_3_3x3", 4, 64);
    act("pool1_3x3_s2", 8, 64);
    int div = 8;
    for (int i = 0; i < kNumBnecks; ++i) {
        const BneckSpec &b = kBnecks[i];
        if (i == kStemBnecks) {
            act("conv3_1_sub4", 32, 256);
            div = 32;
        }
        const int odiv = div * b.stride;
        const std::string n = b.name;
        act(n + "_1x1_reduce", odiv, b.mid);
        act(n + "_3x3", odiv, b.mid);
        if (b.proj) act(n + "_1x1_proj", odiv, b.cout);
        act(n, odiv, b.cout);
        div = odiv;
    }
    act("conv5_3_sum", 32, 1024);
    act("conv5_4_k1", 32, 256);
    act("conv1_sub1", 2, 32);
    act("conv2_sub1", 4, 32);
    act("conv3_sub1", 8, 64);
    act("conv3_1_sub2_proj", 16, 128);
    act("sub24_sum", 16, 128);
    act("conv3_sub1_proj", 8, 128);
    act("sub12_sum", 8, 128);
    act("conv6_cls", 4, h->classes);
}

const std::vector<float> &T(const ssal_icnet *h, const std::string &name) { return h->tensors[h->index.at(name)].data; }

struct IcWorkspace {
    std::map<std::string, float *> act;
    float *pooled = nullptr;
    double *partial = nullptr;
    int64_t bytes = 0;
    bool ok = true;
};

IcWorkspace carve(const ssal_icnet *net, void *ws, int64_t ws_bytes, int64_t n, int64_t h, int64_t w)
{
    Bump b(ws, ws_bytes);
    IcWorkspace W;
    for (const ActSpec &a : net->acts) {
        W.act[a.name] = b.take<float>(n * (h / a.div) * (w / a.div) * a.c);
    }
    W.pooled = b.take<float>(ppm_scratch_floats((int)n, (int)(h / 32), 1024));
    W.partial = b.take<double>(n * (int64_t)upscore_blocks((int)(h / 4), (int)(w / 4)));
    W.bytes = b.off;
    W.ok = b.ok;
    return W;
}

int check_dims(const ssal_icnet *net, int n, int h, int w)
{
    if (!net) return fail(SSAL_EINVAL, "net is NULL");
    if (!net->committed) return fail(SSAL_ESTATE, "ssal_icnet_commit() has not been called");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != net->device)
        return fail(SSAL_ESTATE, "the handle was committed on device %d but the current device is %d (one handle per device)",
                    net->device, dev);
    if (n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d", n, h, w);
    if (h % 32 || w % 32) return fail(SSAL_EINVAL, "ICNet needs H and W divisible by 32 (got %dx%d)", h, w);
    // the convolution kernels address each tensor with 32-bit byte offsets; the largest one (conv1_sub1's output) holds
    // n*h*w/4 pixels x 32 floats = 32 bytes per input pixel
    if ((int64_t)n * h * w >= ((int64_t)1 << 27)) return fail(SSAL_EINVAL, "batch too large (n=%d h=%d w=%d): split it", n, h, w);
    return SSAL_OK;
}

#define HIP_RET(expr)                    \
    do {                                 \
        hipError_t e_ = (expr);          \
        if (e_ != hipSuccess) return e_; \
    } while (0)

// conv -> BN -> [+res] -> [relu] on the matrix cores
hipError_t run_conv(const ssal_icnet *net, const std::string &name, const float *x, int n, int h, int w,
                    const float *res, bool relu, bool up2, float *y, hipStream_t s)
{
    const ConvDev &c = net->convs.at(name);
    return launch_igemm(x, n, h, w, c.spec.cin, c.w, c.spec.k, c.spec.k, c.spec.cout, c.spec.stride, c.spec.dil,
                        c.scale, c.shift, res, relu, up2, y, s);
}

// runs everything up to the 1/4-resolution class logits (ICNET_SPEC conv6_cls)
// one chain of the image-group schedule: a contiguous range of the batch with its views of the workspace and its stream
struct Grp {
    IcWorkspace W;
    const void *x;
    int n;
    hipStream_t s;
    float *A(const std::string &nm) const { return W.act.at(nm); }
};

// fused: the score path (nobody reads the intermediate layer outputs) may run a whole block as one launch; the forward
// path keeps one launch per ICNET_SPEC layer, so that every layer output of the last forward call is an endpoint.
// Every layer is issued for all groups before the next layer (layer-major order: the chains advance together on their
// streams instead of one chain's 70 launches queueing up in front of the other's).
// written != NULL: DRY run -- nothing is launched, the names of the activation buffers this schedule would write are
// collected instead (ssal_icnet_endpoint_valid_after_score: the one place that knows which layer outputs a fused launch
// swallows is the schedule itself)
hipError_t run_trunk(const ssal_icnet *net, std::vector<Grp> &grp, bool x_is_u8, int h, int w, bool fused,
                     std::set<std::string> *written = nullptr)
{
    const bool dry = written != nullptr;
#define OUT(name) do { if (dry) written->insert(name); } while (0)
#define EACH(expr)                                     \
    for (Grp & q : grp) {                              \
        if (dry) break;                                \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return e_;               \
    }
    // ---- medium-resolution branch / shared stem (section 1) ----
    // fused score path: the first two convolutions of a branch in one launch (k_front2), the first one's output never
    // reaches HBM (and is no endpoint of such a call)
    const int front = fused && ssal::mfma_family() ? ssal::knobs().ic_front : 0;
    if ((front & 2) && front2_supported(h, w, net->c_in, 2, 1)) {
        const ConvDev &c = net->convs.at("conv1_1_3x3_s2"), &c2 = net->convs.at("conv1_2_3x3");
        OUT("conv1_2_3x3");
        EACH(launch_front2(q.x, x_is_u8, q.n, h, w, net->c_in, 2, c.w, c.scale, c.shift, c2.w, c2.scale, c2.shift, 1,
                           q.A("conv1_2_3x3"), q.s));
    } else {
        const ConvDev &c = net->convs.at("conv1_1_3x3_s2");
        OUT("conv1_1_3x3_s2");
        OUT("conv1_2_3x3");
        EACH(launch_conv_first(q.x, x_is_u8, q.n, h, w, net->c_in, 2, c.w, c.scale, c.shift, q.A("conv1_1_3x3_s2"), q.s));
        EACH(run_conv(net, "conv1_2_3x3", q.A("conv1_1_3x3_s2"), q.n, h / 4, w / 4, nullptr, true, false, q.A("conv1_2_3x3"), q.s));
    }
    OUT("conv1_3_3x3");
    OUT("pool1_3x3_s2");
    EACH(run_conv(net, "conv1_3_3x3", q.A("conv1_2_3x3"), q.n, h / 4, w / 4, nullptr, true, false, q.A("conv1_3_3x3"), q.s));
    EACH(launch_maxpool3x3_s2(q.A("conv1_3_3x3"), q.n, h / 4, w / 4, 64, q.A("pool1_3x3_s2"), q.s));
    std::string cur = "pool1_3x3_s2";
    int ch = h / 8, cw = w / 8;
    for (int i = 0; i < kNumBnecks; ++i) {
        const BneckSpec &b = kBnecks[i];
        const std::string nm = b.name;
        if (i == kStemBnecks) {
            // section 2: conv3_1_sub4 = resize_bilinear(conv3_1, 1/2)
            OUT("conv3_1_sub4");
            EACH(launch_resize_bilinear(q.A(cur), q.n, ch, cw, 256, ch / 2, cw / 2, q.A("conv3_1_sub4"), q.s));
            cur = "conv3_1_sub4";
            ch /= 2;
            cw /= 2;
        }
        const int oh = ch / b.stride, ow = cw / b.stride;
        if (fused && ssal::mfma_family() && fused_bneck(b) && b.dil >= 1) {
            // reduce -> 3x3 -> increase + identity shortcut -> ReLU == ENet's regular bottleneck with zero slopes: same
            // (kh, kw, ci)-ascending fmaf chains, same folded batch-norm, same `fmaf(e, s, t) + x` merge; a ReLU written as
            // PReLU(slope 0) yields -0.0 where max(v, 0) yields +0.0, which no later layer can tell apart (every consumer
            // adds it into a +0-initialised chain)
            const ConvDev &r = net->convs.at(nm + "_1x1_reduce"), &c3 = net->convs.at(nm + "_3x3"),
                          &inc = net->convs.at(nm + "_1x1_increase");
            OUT(nm);
            EACH(launch_bottleneck_mfma(q.A(cur), q.A(nm), q.n, ch, cw, b.cin, b.dil, r.w_hwio, r.scale, r.shift,
                                        net->zeros128, c3.w_hwio, nullptr, c3.scale, c3.shift, net->zeros128, inc.w_hwio,
                                        inc.scale, inc.shift, net->zeros128, q.s, r.wq));
            cur = nm;
            continue;
        }
        std::string shortcut = cur;
        // fused score path (knob ic_dual): the projection shortcut is evaluated inside the increase launch (k_igemm<.., DUAL>)
        // and never reaches HBM
        const bool dual = b.proj && fused && ssal::mfma_family() && ssal::knobs().ic_dual;
        if (b.proj && !dual) {
            OUT(nm + "_1x1_proj");
            EACH(run_conv(net, nm + "_1x1_proj", q.A(cur), q.n, ch, cw, nullptr, false, false, q.A(nm + "_1x1_proj"), q.s));
            shortcut = nm + "_1x1_proj";
        }
        OUT(nm + "_1x1_reduce");
        OUT(nm + "_3x3");
        OUT(nm);
        EACH(run_conv(net, nm + "_1x1_reduce", q.A(cur), q.n, ch, cw, nullptr, true, false, q.A(nm + "_1x1_reduce"), q.s));
        EACH(run_conv(net, nm + "_3x3", q.A(nm + "_1x1_reduce"), q.n, oh, ow, nullptr, true, false, q.A(nm + "_3x3"), q.s));
        if (dual) {
            const ConvDev &ci_ = net->convs.at(nm + "_1x1_increase"), &cp = net->convs.at(nm + "_1x1_proj");
            EACH(launch_igemm_dual(q.A(nm + "_3x3"), q.n, oh, ow, ci_.spec.cin, ci_.w, ci_.spec.cout, ci_.scale, ci_.shift,
                                   q.A(cur), cp.spec.cin, b.stride, cp.w, cp.scale, cp.shift, true, q.A(nm), q.s));
        } else {
            EACH(run_conv(net, nm + "_1x1_increase", q.A(nm + "_3x3"), q.n, oh, ow, q.A(shortcut), true, false, q.A(nm), q.s));
        }
        cur = nm;
        ch = oh;
        cw = ow;
    }
    // pyramid pooling + conv5_4_k1
    OUT("conv5_3_sum");
    OUT("conv5_4_k1");
    EACH(launch_ppm(q.A(cur), q.n, ch, cw, 1024, q.W.pooled, q.A("conv5_3_sum"), q.s));
    EACH(run_conv(net, "conv5_4_k1", q.A("conv5_3_sum"), q.n, ch, cw, nullptr, true, false, q.A("conv5_4_k1"), q.s));
    // ---- high-resolution branch (section 3) ----
    if ((front & 1) && front2_supported(h, w, net->c_in, 1, 2)) {
        const ConvDev &c = net->convs.at("conv1_sub1"), &c2 = net->convs.at("conv2_sub1");
        OUT("conv2_sub1");
        EACH(launch_front2(q.x, x_is_u8, q.n, h, w, net->c_in, 1, c.w, c.scale, c.shift, c2.w, c2.scale, c2.shift, 2,
                           q.A("conv2_sub1"), q.s));
    } else {
        const ConvDev &c = net->convs.at("conv1_sub1");
        OUT("conv1_sub1");
        OUT("conv2_sub1");
        EACH(launch_conv_first(q.x, x_is_u8, q.n, h, w, net->c_in, 1, c.w, c.scale, c.shift, q.A("conv1_sub1"), q.s));
        EACH(run_conv(net, "conv2_sub1", q.A("conv1_sub1"), q.n, h / 2, w / 2, nullptr, true, false, q.A("conv2_sub1"), q.s));
    }
    for (const char *nm_ : {"conv3_sub1", "conv3_1_sub2_proj", "sub24_sum", "conv3_sub1_proj", "sub12_sum", "conv6_cls"}) OUT(nm_);
    EACH(run_conv(net, "conv3_sub1", q.A("conv2_sub1"), q.n, h / 4, w / 4, nullptr, true, false, q.A("conv3_sub1"), q.s));
    // ---- cascade feature fusion (section 4): the 2x interpolations are evaluated inside the dilated convs ----
    EACH(run_conv(net, "conv3_1_sub2_proj", q.A("conv3_1"), q.n, h / 16, w / 16, nullptr, false, false,
                  q.A("conv3_1_sub2_proj"), q.s));
    EACH(run_conv(net, "conv_sub4", q.A("conv5_4_k1"), q.n, h / 32, w / 32, q.A("conv3_1_sub2_proj"), true, true,
                  q.A("sub24_sum"), q.s));
    EACH(run_conv(net, "conv3_sub1_proj", q.A("conv3_sub1"), q.n, h / 8, w / 8, nullptr, false, false,
                  q.A("conv3_sub1_proj"), q.s));
    EACH(run_conv(net, "conv_sub2", q.A("sub24_sum"), q.n, h / 16, w / 16, q.A("conv3_sub1_proj"), true, true,
                  q.A("sub12_sum"), q.s));
    // sub12_sum_interp (2x) + conv6_cls (1x1, bias)
    EACH(run_conv(net, "conv6_cls", q.A("sub12_sum"), q.n, h / 8, w / 8, nullptr, false, true, q.A("conv6_cls"), q.s));
#undef EACH
#undef OUT
    return hipSuccess;
}

int forward_any(ssal_icnet *net, const void *x_dev, bool u8, int n, int h, int w, float *logits_dev, void *ws_dev,
                int64_t ws_bytes, void *stream)
{
    int rc = check_dims(net, n, h, w);
    if (rc) return rc;
    if (!x_dev || !logits_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    IcWorkspace W = carve(net, ws_dev, ws_bytes, n, h, w);
    if (!W.ok) return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes, got %lld", (long long)W.bytes, (long long)ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    std::vector<Grp> one(1);
    one[0] = {W, x_dev, n, s};
    HIP_TRY(run_trunk(net, one, u8, h, w, false));
    HIP_TRY(launch_resize_bilinear(W.act.at("conv6_cls"), n, h / 4, w / 4, net->classes, h, w, logits_dev, s));
    return SSAL_OK;
}

int score_any(ssal_icnet *net, const void *x_dev, bool u8, int n, int h, int w, int measure, float threshold,
              double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev, float *conf_dev, void *ws_dev,
              int64_t ws_bytes, void *stream)
{
    int rc = check_dims(net, n, h, w);
    if (rc) return rc;
    if (measure < 0 || measure > 2) return fail(SSAL_ENOTIMPL, "Uncertainty function not implemented (measure=%d)", measure);
    if (!x_dev || !scores_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (((uintptr_t)label_dev | (uintptr_t)mask_dev) & 3 || ((uintptr_t)conf_dev & 15))
        return fail(SSAL_EINVAL, "label_dev / mask_dev must be 4-byte aligned and conf_dev 16-byte aligned");
    IcWorkspace W = carve(net, ws_dev, ws_bytes, n, h, w);
    if (!W.ok) return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes, got %lld", (long long)W.bytes, (long long)ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    // image-group schedule (same knob and same reasoning as ENet's run_net, ssal_api.hip): the batch runs as G chains of
    // ~n / G images on library-owned side streams, forked from / joined into the caller's stream with events
    int G = ssal::knobs().ic_groups;  // ICNet's own chain count (default 1; ENet: img_groups)
    if (G < 2 || G > 8 || n < G || !ssal::mfma_family() || ssal::prof_enabled()) G = 1;
    const int64_t px = (int64_t)h * w, ppm_img = ppm_scratch_floats(1, h / 32, 1024);
    const int blocks = upscore_blocks(h / 4, w / 4);
    std::vector<Grp> grp(G);
    std::vector<int64_t> first(G + 1);
    for (int g = 0; g <= G; ++g) first[g] = (int64_t)g * n / G;
    // fork / join events are private to this call (ssal::ChainSet, ssal_api.hip); a failure half way still joins the chains
    ssal::ChainSet cs;
    if (G > 1) HIP_TRY(cs.begin(G, s));
    for (int g = 0; g < G; ++g) {
        const int64_t i0 = first[g];
        Grp &q = grp[g];
        q.W = W;
        for (const ActSpec &a : net->acts) q.W.act[a.name] = W.act.at(a.name) + i0 * (h / a.div) * (w / a.div) * a.c;
        q.W.pooled = W.pooled + i0 * ppm_img;
        q.W.partial = W.partial + i0 * blocks;
        q.x = (const char *)x_dev + (size_t)i0 * px * net->c_in * (u8 ? 1 : 4);
        q.n = (int)(first[g + 1] - i0);
        q.s = s;
        if (G > 1) q.s = cs.side[g];
    }
    set_launch_concurrency(G);
    const hipError_t trunk_rc = run_trunk(net, grp, u8, h, w, true);
    set_launch_concurrency(1);
    HIP_TRY(trunk_rc);
    for (int g = 0; g < G; ++g) {
        const Grp &q = grp[g];
        const int64_t i0 = first[g];
        HIP_TRY(launch_upscore(q.A("conv6_cls"), q.n, h / 4, w / 4, net->classes, measure, threshold, q.W.partial,
                               label_dev ? label_dev + i0 * px : nullptr, mask_dev ? mask_dev + i0 * px : nullptr,
                               conf_dev ? conf_dev + i0 * px : nullptr, q.s));
    }
    HIP_TRY(cs.end());
    HIP_TRY(launch_reduce_mean(W.partial, n, upscore_blocks(h / 4, w / 4), (double)h * (double)w, scores_dev, s));
    return SSAL_OK;
}

// folded batch-norm (extra_ops.py:181-184, eps 1e-3) padded to CoutP entries
void fold_bn_padded(const float *mean, const float *var, const float *gamma, const float *beta, const float *bias,
                    int c, std::vector<float> &s, std::vector<float> &t)
{
    const int cp = (c + 31) / 32 * 32;
    s.assign(cp, 0.0f);
    t.assign(cp, 0.0f);
    for (int i = 0; i < c; ++i) {
        if (mean) {
            const float sg = gamma[i] / sqrtf(var[i] + 1e-3f);
            s[i] = sg;
            t[i] = fmaf(-mean[i], sg, beta[i]);
        } else {
            s[i] = 1.0f;  // fmaf(acc, 1, bias) == acc + bias exactly
            t[i] = bias ? bias[i] : 0.0f;
        }
    }
}

}  // namespace

SSAL_API int ssal_icnet_create(int c_in, int classes, ssal_icnet **out)
{
    if (!out) return fail(SSAL_EINVAL, "out is NULL");
    if (!(c_in == 1 || c_in == 3 || c_in == 4)) return fail(SSAL_EINVAL, "c_in must be 1, 3 or 4 (got %d)", c_in);
    if (classes < 2 || classes > 32) return fail(SSAL_EINVAL, "classes must be in [2,32] (got %d)", classes);
    ssal_icnet *h = new ssal_icnet();
    h->c_in = c_in;
    h->classes = classes;
    declare(h);
    *out = h;
    return SSAL_OK;
}

SSAL_API int ssal_icnet_destroy(ssal_icnet *net)
{
    if (!net) return SSAL_OK;
    if (net->arena) (void)hipFree(net->arena);
    delete net;
    return SSAL_OK;
}

SSAL_API int ssal_icnet_num_tensors(const ssal_icnet *net) { return net ? (int)net->tensors.size() : 0; }

SSAL_API int ssal_icnet_tensor_info(const ssal_icnet *net, int i, const char **name, int *ndim, int64_t dims[4])
{
    if (!net || i < 0 || i >= (int)net->tensors.size()) return fail(SSAL_EINVAL, "bad tensor index %d", i);
    const HostTensor &t = net->tensors[i];
    if (name) *name = t.name.c_str();
    if (ndim) *ndim = (int)t.dims.size();
    if (dims)
        for (size_t d = 0; d < 4; ++d) dims[d] = d < t.dims.size() ? t.dims[d] : 1;
    return SSAL_OK;
}

SSAL_API int ssal_icnet_set_tensor(ssal_icnet *net, const char *name, const float *host, int64_t numel)
{
    if (!net || !name || !host) return fail(SSAL_EINVAL, "NULL argument");
    auto it = net->index.find(name);
    if (it == net->index.end()) return fail(SSAL_EINVAL, "unknown tensor '%s'", name);
    HostTensor &t = net->tensors[it->second];
    if (numel != t.numel())
        return fail(SSAL_EINVAL, "tensor '%s': expected %lld elements, got %lld", name, (long long)t.numel(), (long long)numel);
    t.data.assign(host, host + numel);
    t.set = true;
    net->committed = false;
    return SSAL_OK;
}

SSAL_API int ssal_icnet_commit(ssal_icnet *net, void *stream)
{
    if (!net) return fail(SSAL_EINVAL, "net is NULL");
    for (const auto &t : net->tensors)
        if (!t.set) return fail(SSAL_ESTATE, "tensor '%s' has not been set", t.name.c_str());
    ArenaBuilder ab;
    struct Off { size_t w, s, t, hwio, wq; };
    std::map<std::string, Off> offs;
    std::vector<float> s, t, wt;
    std::map<std::string, bool> wants_hwio;
    for (int i = 0; i < kNumBnecks; ++i)
        if (fused_bneck(kBnecks[i]))
            for (const char *suf : {"_1x1_reduce", "_3x3", "_1x1_increase"}) wants_hwio[std::string(kBnecks[i].name) + suf] = true;
    const size_t zeros_off = ab.push(std::vector<float>(128, 0.0f));
    for (const ConvSpec &sp : net->specs) {
        Off o;
        o.hwio = (size_t)-1;
        o.wq = (size_t)-1;
        for (int i = 0; i < kNumBnecks; ++i)
            if (fused_bneck(kBnecks[i]) && sp.name == std::string(kBnecks[i].name) + "_1x1_reduce") {
                const std::string b = kBnecks[i].name;
                o.wq = ab.push(bnk_quad_layout(T(net, b + "_1x1_reduce.kernel").data(), T(net, b + "_3x3.kernel").data(), nullptr, 9,
                                               T(net, b + "_1x1_increase.kernel").data()));
            }
        if (wants_hwio.count(sp.name)) o.hwio = ab.push(T(net, sp.name + ".kernel"));
        const std::vector<float> &k = T(net, sp.name + ".kernel");
        if (sp.cin % 32 == 0) {
            wt.resize(igemm_relayout_floats(sp.k, sp.k, sp.cin, sp.cout));
            igemm_relayout(k.data(), sp.k, sp.k, sp.cin, sp.cout, wt.data());
            o.w = ab.push(wt);
        } else {
            o.w = ab.push(k);
        }
        if (sp.bn)
            fold_bn_padded(T(net, sp.name + ".mean").data(), T(net, sp.name + ".variance").data(),
                           T(net, sp.name + ".gamma").data(), T(net, sp.name + ".beta").data(), nullptr, sp.cout, s, t);
        else
            fold_bn_padded(nullptr, nullptr, nullptr, nullptr, T(net, sp.name + ".bias").data(), sp.cout, s, t);
        o.s = ab.push(s);
        o.t = ab.push(t);
        offs[sp.name] = o;
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
    HIP_TRY(hipMemcpyAsync(net->arena, ab.host.data(), ab.host.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // the staging vector dies at return
    net->convs.clear();
    for (const ConvSpec &sp : net->specs) {
        ConvDev d;
        d.spec = sp;
        const Off &o = offs.at(sp.name);
        d.w = net->arena + o.w;
        d.scale = net->arena + o.s;
        d.shift = net->arena + o.t;
        d.w_hwio = o.hwio == (size_t)-1 ? nullptr : net->arena + o.hwio;
        d.wq = o.wq == (size_t)-1 ? nullptr : net->arena + o.wq;
        net->convs[sp.name] = d;
    }
    net->zeros128 = net->arena + zeros_off;
    net->committed = true;
    return SSAL_OK;
}

SSAL_API int64_t ssal_icnet_workspace_bytes(const ssal_icnet *net, int n, int h, int w)
{
    if (!net || !net->committed || n <= 0 || h <= 0 || w <= 0) return -1;
    IcWorkspace W = carve(net, nullptr, 0, n, h, w);
    return W.bytes + 256;
}

SSAL_API int ssal_icnet_forward_nhwc(ssal_icnet *net, const float *x_dev, int n, int h, int w, float *logits_dev,
                                     void *ws_dev, int64_t ws_bytes, void *stream)
{
    return forward_any(net, x_dev, false, n, h, w, logits_dev, ws_dev, ws_bytes, stream);
}

SSAL_API int ssal_icnet_forward_nhwc_u8(ssal_icnet *net, const uint8_t *x_dev, int n, int h, int w, float *logits_dev,
                                        void *ws_dev, int64_t ws_bytes, void *stream)
{
    return forward_any(net, x_dev, true, n, h, w, logits_dev, ws_dev, ws_bytes, stream);
}

SSAL_API int ssal_icnet_score_nhwc(ssal_icnet *net, const float *x_dev, int n, int h, int w, int measure,
                                   float threshold, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev,
                                   float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream)
{
    return score_any(net, x_dev, false, n, h, w, measure, threshold, scores_dev, label_dev, mask_dev, conf_dev, ws_dev,
                     ws_bytes, stream);
}

SSAL_API int ssal_icnet_score_nhwc_u8(ssal_icnet *net, const uint8_t *x_dev, int n, int h, int w, int measure,
                                      float threshold, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev,
                                      float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream)
{
    return score_any(net, x_dev, true, n, h, w, measure, threshold, scores_dev, label_dev, mask_dev, conf_dev, ws_dev,
                     ws_bytes, stream);
}

SSAL_API int ssal_icnet_num_endpoints(const ssal_icnet *net) { return net ? (int)net->acts.size() : 0; }

SSAL_API int ssal_icnet_endpoint_name(const ssal_icnet *net, int i, const char **name)
{
    if (!net || !name || i < 0 || i >= (int)net->acts.size()) return fail(SSAL_EINVAL, "bad endpoint index %d", i);
    *name = net->acts[i].name.c_str();
    return SSAL_OK;
}

SSAL_API int ssal_icnet_endpoint_info(const ssal_icnet *net, const char *name, int n, int h, int w, int64_t *offset,
                                      int64_t dims[4])
{
    if (!net || !name || !offset || !dims) return fail(SSAL_EINVAL, "NULL argument");
    if (n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d", n, h, w);
    Bump b((void *)256, (int64_t)1 << 62);  // same carving as the forward pass, on a fake base
    for (const ActSpec &a : net->acts) {
        const int64_t cnt = (int64_t)n * (h / a.div) * (w / a.div) * a.c;
        const char *p = (const char *)b.take<float>(cnt);
        if (a.name == name) {
            *offset = p - (const char *)256;
            dims[0] = n; dims[1] = h / a.div; dims[2] = w / a.div; dims[3] = a.c;
            return SSAL_OK;
        }
    }
    return fail(SSAL_EINVAL, "'%s' is not a materialised ICNet tensor", name);
}

// 1: a SCORE call at h x w (with the knobs as they are now) writes the named endpoint; 0: a fused launch swallows it (its
// buffer keeps whatever an earlier call left); -1 (+ last error): unknown name / bad arguments
SSAL_API int ssal_icnet_endpoint_valid_after_score(const ssal_icnet *net, const char *name, int h, int w)
{
    if (!net || !name) { (void)fail(SSAL_EINVAL, "NULL argument"); return -1; }
    if (!net->committed) { (void)fail(SSAL_ESTATE, "ssal_icnet_commit() has not been called"); return -1; }
    if (h <= 0 || w <= 0 || h % 32 || w % 32) { (void)fail(SSAL_EINVAL, "bad dims h=%d w=%d", h, w); return -1; }
    bool known = false;
    for (const ActSpec &a : net->acts) known = known || a.name == name;
    if (!known) { (void)fail(SSAL_EINVAL, "'%s' is not a materialised ICNet tensor", name); return -1; }
    std::set<std::string> written;
    std::vector<Grp> none;
    (void)run_trunk(net, none, false, h, w, true, &written);
    return written.count(name) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------------
SSAL_API int64_t ssal_conv_bn_workspace_bytes(int kh, int kw, int cin, int cout)
{
    if (kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return -1;
    const int64_t wfl = cin % 32 == 0 ? (int64_t)igemm_relayout_floats(kh, kw, cin, cout) : (int64_t)kh * kw * cin * cout;
    return (wfl + 2 * ((cout + 31) / 32 * 32)) * 4 + 1024;
}

SSAL_API int ssal_conv_bn_act(const float *x_dev, int n, int h, int w, int cin, const float *kernel_host, int kh,
                              int kw, int cout, int stride, int dilation, const float *mean_host,
                              const float *var_host, const float *gamma_host, const float *beta_host,
                              const float *bias_host, const float *res_dev, int relu, int upsample2x, float *y_dev,
                              void *ws_dev, int64_t ws_bytes, void *stream)
{
    if (!x_dev || !kernel_host || !y_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL pointer");
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || stride < 1 || dilation < 1)
        return fail(SSAL_EINVAL, "bad dims");
    const bool bn = mean_host || var_host || gamma_host || beta_host;
    if (bn && !(mean_host && var_host && gamma_host && beta_host))
        return fail(SSAL_EINVAL, "batch-norm needs all of mean / variance / gamma / beta");
    if (ws_bytes < ssal_conv_bn_workspace_bytes(kh, kw, cin, cout))
        return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes", (long long)ssal_conv_bn_workspace_bytes(kh, kw, cin, cout));
    const bool first = cin % 32 != 0;
    if (first && !((cin == 1 || cin == 3 || cin == 4) && kh == 3 && kw == 3 && stride == 2 && dilation == 1 &&
                   cout == 32 && relu && !res_dev && !upsample2x))
        return fail(SSAL_EINVAL, "unsupported convolution: cin %% 32 == 0, or the 3x3 / stride-2 / 32-channel first layer "
                                 "on 1, 3 or 4 channels");
    if (!first && !igemm_supported(cin, cout, kh, kw)) return fail(SSAL_EINVAL, "unsupported kernel size %dx%d", kh, kw);
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> sc, sh, wt;
    fold_bn_padded(mean_host, var_host, gamma_host, beta_host, bias_host, cout, sc, sh);
    if (first) wt.assign(kernel_host, kernel_host + (size_t)kh * kw * cin * cout);
    else {
        wt.resize(igemm_relayout_floats(kh, kw, cin, cout));
        igemm_relayout(kernel_host, kh, kw, cin, cout, wt.data());
    }
    Bump b(ws_dev, ws_bytes);
    float *wd = b.take<float>((int64_t)wt.size());
    float *sd = b.take<float>((int64_t)sc.size());
    float *td = b.take<float>((int64_t)sh.size());
    HIP_TRY(hipMemcpyAsync(wd, wt.data(), wt.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(sd, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(td, sh.data(), sh.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // the staging vectors die at return
    if (first) HIP_TRY(launch_conv_first(x_dev, false, n, h, w, cin, 1, wd, sd, td, y_dev, s));
    else
        HIP_TRY(launch_igemm(x_dev, n, h, w, cin, wd, kh, kw, cout, stride, dilation, sd, td, res_dev, relu != 0,
                             upsample2x != 0, y_dev, s));
    return SSAL_OK;
}

SSAL_API int ssal_max_pool_3x3_s2(const float *x_dev, int n, int h, int w, int c, float *y_dev, void *stream)
{
    if (!x_dev || !y_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 4) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d c=%d (c %% 4 == 0)", n, h, w, c);
    HIP_TRY(launch_maxpool3x3_s2(x_dev, n, h, w, c, y_dev, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int ssal_pyramid_pooling(const float *x_dev, int n, int h, int w, int c, float *y_dev, void *ws_dev,
                                  int64_t ws_bytes, void *stream)
{
    if (!x_dev || !y_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 4) return fail(SSAL_EINVAL, "bad dims");
    if (ws_bytes < ppm_scratch_floats(n, h, c) * 4)
        return fail(SSAL_ENOMEM, "workspace too small: need %lld bytes", (long long)ppm_scratch_floats(n, h, c) * 4);
    HIP_TRY(launch_ppm(x_dev, n, h, w, c, (float *)ws_dev, y_dev, (hipStream_t)stream));
    return SSAL_OK;
}

SSAL_API int64_t ssal_upscore_workspace_bytes(int n, int h, int w)
{
    if (n <= 0 || h <= 0 || w <= 0) return -1;
    return (int64_t)n * upscore_blocks(h, w) * 8 + 256;
}

SSAL_API int ssal_upscore_logits_nhwc(const float *lq_dev, int n, int h, int w, int classes, int measure,
                                      float threshold, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev,
                                      float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream)
{
    if (measure < 0 || measure > 2) return fail(SSAL_ENOTIMPL, "Uncertainty function not implemented (measure=%d)", measure);
    if (n <= 0 || h <= 0 || w <= 0) return fail(SSAL_EINVAL, "bad dims n=%d h=%d w=%d", n, h, w);
    if (classes < 2 || classes > 32) return fail(SSAL_EINVAL, "classes must be in [2,32] (got %d)", classes);
    if (!lq_dev || !scores_dev || !ws_dev) return fail(SSAL_EINVAL, "NULL device pointer");
    if (((uintptr_t)label_dev | (uintptr_t)mask_dev) & 3 || ((uintptr_t)conf_dev & 15))
        return fail(SSAL_EINVAL, "label_dev / mask_dev must be 4-byte aligned and conf_dev 16-byte aligned");
    if (ws_bytes < ssal_upscore_workspace_bytes(n, h, w)) return fail(SSAL_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws_dev, ws_bytes);
    double *partial = b.take<double>((int64_t)n * upscore_blocks(h, w));
    HIP_TRY(launch_upscore(lq_dev, n, h, w, classes, measure, threshold, partial, label_dev, mask_dev, conf_dev, s));
    HIP_TRY(launch_reduce_mean(partial, n, upscore_blocks(h, w), 16.0 * (double)h * (double)w, scores_dev, s));
    return SSAL_OK;
}
