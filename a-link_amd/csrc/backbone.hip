// backbone.hip — host side of the IR-ResNet ("ArcFace") feature extractor behind the C ABI.
//
// Replaces, for the hot path only:
//   face_model.get_model            reference code/face_model.py:28-41  (load_checkpoint, slice at
//                                   fc1_output, bind, set_params)
//   FaceModel.get_feature           reference code/face_model.py:86-93  (forward + L2 normalise)
// The network arithmetic itself is insightface's LResNet-E-IR symbol (fresnet.py, version_unit 3,
// version_input 1, version_output 'E'), which the reference loads from a downloaded checkpoint
// (code/arcface_prepreq.sh:18-19); SURVEY.md §8 row a5 is the spec followed here:
//   data -> (x-127.5)*0.0078125 -> conv0 3x3 -> bn0 -> prelu
//   4 stages of units:  bn1 -> conv1 3x3 s1 -> bn2 -> prelu -> conv2 3x3 s{2,1} -> bn3  (+ shortcut:
//                       identity, or conv1sc 1x1 s2 -> bn 'sc' on the first unit of a stage)
//   bn1 -> dropout(identity) -> flatten(C,H,W) -> pre_fc1 -> fc1 (BN, fix_gamma)
//
// Everything BN is folded at finalize() (inference only, like the reference's is_train=False):
//   * post-conv BN scale -> weight rows, shift -> bias;
//   * the pre-activation bn1 of a unit: scale -> weight input channels; its shift cannot be folded
//     into a single bias because the conv zero-pads AFTER the BN, so border pixels miss some taps.
//     We keep it exact with 9 bias classes (top/middle/bottom x left/middle/right): class bias =
//     full bias minus the per-tap shift contributions of the taps that fall outside the image.
//   * final bn1 + fc1 BN fold into the FC weights/bias (no padding there).
#include "alink_common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace alink {

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return ALINK_EHIP;
}

uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
uint16_t f32_to_f16_rne(float f) {
    _Float16 h = (_Float16)f;   // host clang: IEEE RNE conversion
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static inline uint16_t cvt(int dtype, float f) {
    return dtype == ALINK_DT_BF16 ? f32_to_bf16_rne(f) : f32_to_f16_rne(f);
}
// ALINK_DT_F16X2: x -> f16 pair, hi = RN16(x), lo = RN16(x - hi): |x - hi - lo| <= 2^-22 |x| (while lo stays normal)
static inline void split16(double x, uint16_t* hi, uint16_t* lo) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (double)h);
    memcpy(hi, &h, 2);
    memcpy(lo, &l, 2);
}
// exponent e with maxabs * 2^e in [1024, 2048): 32x below the f16 overflow threshold, lo = 2^-11 hi still >= 2^-1
static inline int scale_exp(double maxabs) {
    if (!(maxabs > 0.0) || !std::isfinite(maxabs)) return 0;
    return 10 - std::ilogb(maxabs);
}

hipError_t conv_set_attributes();
static unsigned long long g_kernels_ready = 0;       // one bit per device (function attributes and the probe are per device)
int init_kernels() {
    const int dev = current_device();
    if (dev >= 0 && dev < 64 && (g_kernels_ready >> dev & 1ull)) return ALINK_OK;
    hipError_t e = conv_set_attributes();
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute", __FILE__, __LINE__);
    e = direct_set_attributes();
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(direct)", __FILE__, __LINE__);
    e = linear_check_contract();
    if (e != hipSuccess) return hip_fail(e, "LDS out-of-range read probe", __FILE__, __LINE__);
    if (dev >= 0 && dev < 64) g_kernels_ready |= 1ull << dev;
    return ALINK_OK;
}

// ---------------------------------------------------------------------------------------------
struct BN { std::vector<double> a, b; };   // y = a*x + b

struct ConvLayer {
    int Cin, Cout, ksz, stride, pad, Hin, Win, Hout, Wout;
    bool border_cls, has_alpha;
    int variant = 0;                       // conv3x3_direct variant (0 = conv_igemm)
    int in_buf, out_buf, resid_buf;        // workspace buffer ids, -1 = none
    void*  d_w = nullptr;                   // T [Cout][K] permuted
    float* d_bias = nullptr;                // [ncls][Cout]
    float* d_alpha = nullptr;               // [Cout]
    std::string name;
    // input-gradient pass (alink_backbone_enable_grad): transposed + flipped folded weights for the
    // convolution that maps d(output) to d(input), and where this layer sits in its residual unit
    void* d_wb = nullptr;                   // T [Cin][k*k*Cout], rows permuted for `bvariant`
    int   bvariant = 0;                     // direct variant of the backward convolution (0 = conv_igemm)
    int   role = 0, unit = -1;              // 1 conv1, 2 shortcut, 3 conv2; index of the residual unit
    // ALINK_DT_F16X2: stored value = true value x 2^e.  e_w is fixed at finalize (folded weights), e_out by
    // alink_backbone_calibrate (the largest output the calibration images produce lands in [1024, 2048))
    int   e_w = 0, e_out = 0;
    // fused projection shortcut (16-bit modes, forward only): the unit's 1x1 stride-2 conv1sc + its BN ride as extra
    // K-steps of this (conv2) launch, read from workspace buffer in2_buf with Cin2 channels
    int   Cin2 = 0, in2_buf = -1;
    int   stage = 0;                        // 0..3
};

}  // namespace alink

using namespace alink;

// A/B hook: projection shortcuts fused into the conv2 launch (read at alink_backbone_create; default on)
static bool g_fuse_shortcut = true;
extern "C" void alink_debug_set_fuse_shortcut(int on) { g_fuse_shortcut = on != 0; }

struct alink_backbone {
    int device = -1;                                           // device of every allocation / stream of this handle
    alink_ir_cfg cfg;
    std::vector<std::pair<std::string, size_t>> expected;     // name, count (load order)
    std::map<std::string, std::vector<float>> raw;
    bool finalized = false;

    // device side
    void*  d_stem_w = nullptr;
    float* d_stem_bias = nullptr;
    float* d_stem_alpha = nullptr;
    std::vector<ConvLayer> convs;
    void*  d_fc_w = nullptr;
    float* d_fc_bias = nullptr;
    void*  d_zero = nullptr;
    int fc_K = 0, fc_splitk = 1, fc_kps = 0;
    int Hf = 0, Wf = 0;                                        // final feature map size
    // input-gradient support
    bool grad = false;
    bool split_small = false;   // alink_backbone_set_small_batch_split
    int  siblings = 1;                // shards of the alink_embed call in progress (they run side by side on internal streams)
    bool front_slopes_le_1 = false;   // no PReLU slope of the stem or of stage1_unit1 conv1 exceeds 1 (front_c64.hip: PReLU as a max)
    bool fuse_shortcut = true;  // alink_debug_set_fuse_shortcut (A/B): projection shortcuts inside the conv2 launch
    F32Net* f32 = nullptr;      // cfg.dtype == ALINK_DT_F32: the float32 precision mode (backbone_f32.hip) runs every call
    // ALINK_DT_F16X2 (split precision)
    bool calibrated = false;
    int  products = 3;                      // split precision: matrix-core products per multiplication (alink_backbone_set_products)
    int  stem_e_w = 0, stem_e_out = 0, fc_e_w = 0;
    unsigned* d_absmax = nullptr;           // calibration scratch: bits of the largest |value| of a tensor
    int* h_flag = nullptr;                  // pinned, device-visible: set by fc_finish when an embedding is not finite
    int* d_flag = nullptr;                  // the same word as the device addresses it
    void*  d_fc_wb = nullptr;                                  // T [C*Hf*Wf][emb]: FC transposed (rows permuted)
    float* d_stem_wf = nullptr;                                // f32 [64][27] folded stem weights
    float* d_zero_bias = nullptr;                              // zeros, >= 9 * max width floats
    int n_units = 0;
    std::vector<void*> allocs;
    // Optional sub-batch streams (alink_backbone_set_streams): one call is split into image shards on
    // internal streams.  Measured: pays only when shards from SEVERAL calls overlap without a join
    // (host code pipelines 256-image chunks over streams instead — backbone.py); within one call the
    // join at the end costs more than the de-synchronised HBM bursts gain.
    static constexpr int MAXSUB = 8;
    int nsub = 1;   // default off: a per-call join costs more than the de-synchronisation gains
    hipStream_t sub[MAXSUB] = {};
    hipEvent_t ev_start = nullptr, ev_done[MAXSUB] = {};
    hipEvent_t ev_front[MAXSUB] = {};       // shard i has issued its HBM-bound front (stem + stage 1)

    ~alink_backbone() {
        if (f32) f32net_destroy(f32);
        for (void* p : allocs) (void)hipFree(p);
        for (int i = 0; i < MAXSUB; ++i) {
            if (sub[i]) (void)hipStreamDestroy(sub[i]);
            if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
            if (ev_front[i]) (void)hipEventDestroy(ev_front[i]);
        }
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (h_flag) (void)hipHostFree(h_flag);
    }
};

namespace {

void expect(alink_backbone* bb, const std::string& n, size_t c) { bb->expected.emplace_back(n, c); }
void expect_bn(alink_backbone* bb, const std::string& n, size_t c) {
    expect(bb, n + "_gamma", c);
    expect(bb, n + "_beta", c);
    expect(bb, n + "_moving_mean", c);
    expect(bb, n + "_moving_var", c);
}

int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

BN get_bn(const alink_backbone* bb, const std::string& n, bool fix_gamma) {
    const auto& g = bb->raw.at(n + "_gamma");
    const auto& be = bb->raw.at(n + "_beta");
    const auto& mu = bb->raw.at(n + "_moving_mean");
    const auto& var = bb->raw.at(n + "_moving_var");
    BN r;
    r.a.resize(g.size());
    r.b.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        const double gamma = fix_gamma ? 1.0 : (double)g[i];
        r.a[i] = gamma / std::sqrt((double)var[i] + (double)bb->cfg.bn_eps);
        r.b[i] = (double)be[i] - (double)mu[i] * r.a[i];
    }
    return r;
}

template <typename V>
int upload(alink_backbone* bb, const std::vector<V>& h, void** d) {
    ALINK_HIP(hipMalloc(d, h.size() * sizeof(V)));
    bb->allocs.push_back(*d);
    ALINK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
    return ALINK_OK;
}

// Fold + upload one convolution.  w is MXNet (O, I, kh, kw).
int build_conv(alink_backbone* bb, ConvLayer& L, const std::vector<float>& w, const BN* pre,
               const BN& post, const std::vector<float>* prelu, const std::vector<float>* w_sc = nullptr,
               const BN* post_sc = nullptr) {
    const int O = L.Cout, I = L.Cin, k = L.ksz, K = k * k * I, dt = bb->cfg.dtype;
    const int I2 = w_sc ? L.Cin2 : 0, KR = K + I2;             // KR: weight row pitch of the 16-bit forms
    const bool x2 = dt == ALINK_DT_F16X2;
    // split precision: the linear-tile kernel where it applies, the implicit-GEMM kernel everywhere else
    L.variant = x2 ? linear_variant_x2(L.ksz, L.stride, L.pad, L.Hin, L.Win, L.Cin, L.Cout)
                   : direct_variant(L.ksz, L.stride, L.pad, L.Hin, L.Win, L.Cin, L.Cout);
    const int cpl = L.variant ? direct_variant_cpl(L.variant) : 16;
    std::vector<uint16_t> wq(x2 ? (size_t)O * K * 2 : (size_t)O * KR);
    std::vector<double> tapb((size_t)k * k * O, 0.0);          // [tap][co] shift contribution
    if (x2) {
        double mx = 0.0;
        for (int co = 0; co < O; ++co)
            for (int ci = 0; ci < I; ++ci)
                for (int t = 0; t < k * k; ++t)
                    mx = std::max(mx, std::fabs(post.a[co] * (double)w[((size_t)co * I + ci) * k * k + t] * (pre ? pre->a[ci] : 1.0)));
        L.e_w = scale_exp(mx);
    }
    const double wscale = std::ldexp(1.0, L.e_w);
    for (int co = 0; co < O; ++co) {
        const int row = permuted_row(co, cpl);
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                double tb = 0.0;
                for (int ci = 0; ci < I; ++ci) {
                    const double wv = (double)w[(((size_t)co * I + ci) * k + ky) * k + kx];
                    const double ai = pre ? pre->a[ci] : 1.0;
                    // K order: conv_igemm walks tap-major [tap][ci]; conv3x3_direct walks 64-channel
                    // chunks outermost [ci/64][tap][ci%64]
                    if (x2) {
                        // rows of 2K: linear kernel [chunk][hi | lo][tap][64], implicit GEMM [tap][chunk][hi 64 | lo 64]
                        const int tap = ky * k + kx, cc = ci >> 6;
                        const size_t khi = L.variant ? (((size_t)cc * 2) * 9 + tap) * 64 + (ci & 63)
                                                     : (((size_t)tap * (I >> 6) + cc) * 2) * 64 + (ci & 63);
                        const size_t klo = khi + (L.variant ? 9 * 64 : 64);
                        split16(post.a[co] * wv * ai * wscale, &wq[(size_t)row * 2 * K + khi], &wq[(size_t)row * 2 * K + klo]);
                        if (pre) tb += wv * pre->b[ci];
                        continue;
                    }
                    const size_t kidx = L.variant ? ((size_t)(ci >> 6) * 9 + (ky * 3 + kx)) * 64 + (ci & 63)
                                                  : (size_t)(ky * k + kx) * I + ci;
                    wq[(size_t)row * KR + kidx] = cvt(dt, (float)(post.a[co] * wv * ai));
                    if (pre) tb += wv * pre->b[ci];
                }
                tapb[(size_t)(ky * k + kx) * O + co] = post.a[co] * tb;
            }
        // fused shortcut: conv1sc (O, I2, 1, 1) scaled by its own BN, behind the taps of the same row
        for (int ci = 0; ci < I2; ++ci)
            wq[(size_t)row * KR + K + ci] = cvt(dt, (float)(post_sc->a[co] * (double)(*w_sc)[(size_t)co * I2 + ci]));
    }
    const int ncls = L.border_cls ? 9 : 1;
    std::vector<float> bias((size_t)ncls * O);
    for (int co = 0; co < O; ++co) {
        double full = post.b[co] + (I2 ? post_sc->b[co] : 0.0);
        for (int t = 0; t < k * k; ++t) full += tapb[(size_t)t * O + co];
        if (!L.border_cls) { bias[co] = (float)full; continue; }
        for (int rc = 0; rc < 3; ++rc)
            for (int cc = 0; cc < 3; ++cc) {
                double b = full;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool miss = (rc == 0 && ky == 0) || (rc == 2 && ky == 2) ||
                                          (cc == 0 && kx == 0) || (cc == 2 && kx == 2);
                        if (miss) b -= tapb[(size_t)(ky * 3 + kx) * O + co];
                    }
                bias[(size_t)(rc * 3 + cc) * O + co] = (float)b;
            }
    }
    int rc;
    if ((rc = upload(bb, wq, &L.d_w))) return rc;
    if ((rc = upload(bb, bias, (void**)&L.d_bias))) return rc;
    if (prelu) {
        if ((rc = upload(bb, *prelu, (void**)&L.d_alpha))) return rc;
    }
    if (bb->grad) {
        // d(input) = conv(d(output), Wb): Wb[ci][tap'][co] = Wfolded[co][k*k-1-tap'][ci].  The backward
        // convolution always runs at stride 1 over the layer's INPUT grid (stride-2 layers see a
        // zero-inserted d(output)), with the layer's own padding.
        const int Hb = (L.ksz == 3) ? L.Hin : L.Hout, Wb = (L.ksz == 3) ? L.Win : L.Wout;
        L.bvariant = direct_variant_tiles(L.ksz, 1, L.pad, Hb, Wb, O, I);
        const int bcpl = L.bvariant ? direct_variant_cpl(L.bvariant) : 16;
        const int KB = k * k * O;
        std::vector<uint16_t> wb((size_t)I * KB);
        for (int ci = 0; ci < I; ++ci) {
            const size_t row = (size_t)permuted_row(ci, bcpl) * KB;
            for (int tap = 0; tap < k * k; ++tap) {
                const int ft = k * k - 1 - tap, ky = ft / k, kx = ft % k;
                for (int co = 0; co < O; ++co) {
                    const double wv = (double)w[(((size_t)co * I + ci) * k + ky) * k + kx];
                    const double ai = pre ? pre->a[ci] : 1.0;
                    const size_t kidx = L.bvariant ? ((size_t)(co >> 6) * 9 + tap) * 64 + (co & 63) : (size_t)tap * O + co;
                    wb[row + kidx] = cvt(dt, (float)(post.a[co] * wv * ai));
                }
            }
        }
        if ((rc = upload(bb, wb, &L.d_wb))) return rc;
    }
    return ALINK_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" {

const char* alink_last_error(void) { return g_err; }
int alink_version(void) { return 1; }

int alink_init(int device) {
    // prepares `device` (function attributes, hardware-contract probe) and leaves the caller's current device
    // as it was: handles are created on whatever device is current at their create call
    int count = 0;
    ALINK_HIP(hipGetDeviceCount(&count));
    ALINK_REQUIRE(device >= 0 && device < count, ALINK_EINVAL, "device %d outside the %d visible device(s)", device, count);
    DeviceGuard dg(device);
    ALINK_REQUIRE(current_device() == device, ALINK_EHIP, "cannot make device %d current", device);
    return init_kernels();
}

alink_backbone_t* alink_backbone_create(const alink_ir_cfg* cfg) {
    if (!cfg) { set_error("cfg is NULL"); return nullptr; }
    for (int i = 0; i < 5; ++i)
        if (cfg->widths[i] <= 0 || cfg->widths[i] % 64) {
            set_error("widths[%d]=%d must be a positive multiple of 64", i, cfg->widths[i]);
            return nullptr;
        }
    if (cfg->widths[0] != 64) { set_error("stem width must be 64 (got %d)", cfg->widths[0]); return nullptr; }
    for (int i = 0; i < 4; ++i)
        if (cfg->units[i] < 1) { set_error("units[%d] must be >= 1", i); return nullptr; }
    if (cfg->emb <= 0 || cfg->emb % 64) { set_error("emb must be a multiple of 64"); return nullptr; }
    if (cfg->height < 16 || cfg->width < 16 || cfg->height % 16 || cfg->width % 16) {
        set_error("input size %dx%d must be a multiple of 16 (four stride-2 stages)", cfg->height, cfg->width);
        return nullptr;
    }
    if (cfg->dtype != ALINK_DT_BF16 && cfg->dtype != ALINK_DT_F16 && cfg->dtype != ALINK_DT_F32 && cfg->dtype != ALINK_DT_F16X2) {
        set_error("bad dtype");
        return nullptr;
    }
    alink_backbone* bb = new alink_backbone();
    bb->device = current_device();
    bb->cfg = *cfg;
    bb->fuse_shortcut = g_fuse_shortcut;
    if (!(bb->cfg.bn_eps > 0.f)) bb->cfg.bn_eps = 2e-5f;

    // expected tensors, MXNet names
    const int* w = cfg->widths;
    expect(bb, "conv0_weight", (size_t)w[0] * 3 * 9);
    expect_bn(bb, "bn0", w[0]);
    expect(bb, "relu0_gamma", w[0]);
    int H = cfg->height, W = cfg->width;
    for (int s = 0; s < 4; ++s) {
        const int cin_stage = w[s], c = w[s + 1];
        for (int u = 0; u < cfg->units[s]; ++u) {
            char pfx[64];
            snprintf(pfx, sizeof(pfx), "stage%d_unit%d", s + 1, u + 1);
            const std::string P(pfx);
            const int cin = (u == 0) ? cin_stage : c;
            expect_bn(bb, P + "_bn1", cin);
            expect(bb, P + "_conv1_weight", (size_t)c * cin * 9);
            expect_bn(bb, P + "_bn2", c);
            expect(bb, P + "_relu1_gamma", c);
            expect(bb, P + "_conv2_weight", (size_t)c * c * 9);
            expect_bn(bb, P + "_bn3", c);
            if (u == 0) {
                expect(bb, P + "_conv1sc_weight", (size_t)c * cin);
                expect_bn(bb, P + "_sc", c);
            }
        }
        H = conv_out(H, 3, 2, 1);
        W = conv_out(W, 3, 2, 1);
    }
    bb->Hf = H;
    bb->Wf = W;
    expect_bn(bb, "bn1", w[4]);
    expect(bb, "pre_fc1_weight", (size_t)cfg->emb * w[4] * H * W);
    expect(bb, "pre_fc1_bias", cfg->emb);
    expect_bn(bb, "fc1", cfg->emb);
    return bb;
}

void alink_backbone_destroy(alink_backbone_t* bb) {
    if (!bb) return;
    DeviceGuard dg(bb->device);
    delete bb;
}

int alink_backbone_num_tensors(const alink_backbone_t* bb) { return bb ? (int)bb->expected.size() : 0; }

int alink_backbone_tensor_info(const alink_backbone_t* bb, int i, const char** name, size_t* count) {
    ALINK_REQUIRE(bb && i >= 0 && i < (int)bb->expected.size(), ALINK_EINVAL, "tensor index out of range");
    if (name) *name = bb->expected[i].first.c_str();
    if (count) *count = bb->expected[i].second;
    return ALINK_OK;
}

int alink_backbone_load(alink_backbone_t* bb, const char* name, const float* host, size_t count) {
    ALINK_REQUIRE(bb && name && host, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(!bb->finalized, ALINK_ESTATE, "backbone already finalized");
    for (const auto& e : bb->expected)
        if (e.first == name) {
            ALINK_REQUIRE(e.second == count, ALINK_EINVAL, "tensor %s: expected %zu elements, got %zu", name,
                          e.second, count);
            bb->raw[name].assign(host, host + count);
            return ALINK_OK;
        }
    set_error("tensor %s is not part of the configured network", name);
    return ALINK_ENOTFOUND;
}

int alink_backbone_finalize(alink_backbone_t* bb) {
    ALINK_REQUIRE(bb, ALINK_EINVAL, "NULL backbone");
    ALINK_REQUIRE(!bb->finalized, ALINK_ESTATE, "backbone already finalized");
    DeviceGuard dg(bb->device);
    { const int rc0 = init_kernels(); if (rc0) return rc0; }
    for (const auto& e : bb->expected)
        ALINK_REQUIRE(bb->raw.count(e.first), ALINK_ESTATE, "tensor %s was never loaded", e.first.c_str());
    if (bb->grad)   // the backward pass tells the PReLU side from the sign of the stored activation
        for (const auto& e : bb->expected)
            if (e.first.find("relu") != std::string::npos)
                for (float a : bb->raw.at(e.first))
                    ALINK_REQUIRE(a >= 0.f, ALINK_EINVAL, "%s has a negative PReLU slope: input gradients are not "
                                  "supported for this checkpoint", e.first.c_str());
    int rc = init_kernels();
    if (rc) return rc;
    const alink_ir_cfg& cfg = bb->cfg;
    const int dt = cfg.dtype;
    const int* w = cfg.widths;
    if (dt == ALINK_DT_F32) {      // float32 precision mode: its own (unfused-bn1, f32 GEMM) network, nothing of the bf16 path
        ALINK_REQUIRE(!bb->grad && !bb->split_small, ALINK_ESTATE, "the float32 mode has no gradient pass and no small-batch split");
        bb->f32 = f32net_build(bb->raw, cfg);
        ALINK_REQUIRE(bb->f32, ALINK_ENOMEM, "could not build the float32 network (device memory)");
        bb->raw.clear();
        bb->finalized = true;
        return ALINK_OK;
    }

    const bool x2 = dt == ALINK_DT_F16X2;
    if (x2) {
        ALINK_REQUIRE(!bb->grad, ALINK_ESTATE, "the split-precision mode has no gradient pass");
        ALINK_HIP(hipMalloc((void**)&bb->d_absmax, 256));
        bb->allocs.push_back(bb->d_absmax);
    }
    ALINK_HIP(hipHostMalloc((void**)&bb->h_flag, 64, hipHostMallocMapped));
    *bb->h_flag = 0;
    ALINK_HIP(hipHostGetDevicePointer((void**)&bb->d_flag, bb->h_flag, 0));

    // zero page
    ALINK_HIP(hipMalloc(&bb->d_zero, 4096));
    bb->allocs.push_back(bb->d_zero);
    ALINK_HIP(hipMemset(bb->d_zero, 0, 4096));

    // ---- stem: W'[co][k=ky*9+kx*3+c] = a0[co] * W[co][c][ky][kx]; input normalisation happens in
    // the kernel's loader so the zero frame is a true zero and no border classes are needed.
    {
        const auto& cw = bb->raw.at("conv0_weight");
        const BN bn0 = get_bn(bb, "bn0", false);
        // 16-bit modes: K = 64 in two MFMA steps, [ky 0: kx*3+c (9) + 7 zeros | ky 1: the same] [ky 2: the same | 16 zeros] — a
        // pixel's window of one row is one aligned 32-byte record for the fused front kernel (front_c64.hip), and stem_kernel
        // walks the same K so that the two agree bit for bit; split precision: K = 27 padded to 32, [hi | lo]
        std::vector<uint16_t> wq((size_t)64 * 64, x2 ? (uint16_t)0 : cvt(dt, 0.f));
        if (x2) {
            double mx = 0.0;
            for (int co = 0; co < 64; ++co)
                for (int i = 0; i < 27; ++i) mx = std::max(mx, std::fabs(bn0.a[co] * (double)cw[(size_t)co * 27 + i]));
            bb->stem_e_w = scale_exp(mx);
        }
        for (int co = 0; co < 64; ++co) {
            const int row = perm64_row_of_channel(co);
            for (int c = 0; c < 3; ++c)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const double v = bn0.a[co] * (double)cw[(((size_t)co * 3 + c) * 3 + ky) * 3 + kx];
                        const int kk = ky * 9 + kx * 3 + c;
                        if (x2) split16(std::ldexp(v, bb->stem_e_w), &wq[(size_t)row * 64 + kk], &wq[(size_t)row * 64 + 32 + kk]);
                        else    wq[(size_t)row * 64 + ky * 16 + kx * 3 + c] = cvt(dt, (float)v);
                    }
        }
        std::vector<float> bias(64);
        for (int co = 0; co < 64; ++co) bias[co] = (float)bn0.b[co];
        if ((rc = upload(bb, wq, &bb->d_stem_w))) return rc;
        if ((rc = upload(bb, bias, (void**)&bb->d_stem_bias))) return rc;
        if ((rc = upload(bb, bb->raw.at("relu0_gamma"), (void**)&bb->d_stem_alpha))) return rc;
        bb->front_slopes_le_1 = true;
        for (const char* nm : {"relu0_gamma", "stage1_unit1_relu1_gamma"})
            if (bb->raw.count(nm))
                for (float a : bb->raw.at(nm)) bb->front_slopes_le_1 = bb->front_slopes_le_1 && a <= 1.f;
        if (bb->grad) {
            std::vector<float> wf((size_t)64 * 27);
            for (int co = 0; co < 64; ++co)
                for (int c = 0; c < 3; ++c)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx)
                            wf[(size_t)co * 27 + ky * 9 + kx * 3 + c] =
                                (float)(bn0.a[co] * (double)cw[(((size_t)co * 3 + c) * 3 + ky) * 3 + kx]);
            if ((rc = upload(bb, wf, (void**)&bb->d_stem_wf))) return rc;
            std::vector<float> zb(std::max((size_t)9 * 2048, (size_t)cfg.widths[4] * (cfg.height / 16) * (cfg.width / 16)), 0.f);
            if ((rc = upload(bb, zb, (void**)&bb->d_zero_bias))) return rc;
        }
    }

    // ---- residual stages.  Workspace buffers: 0,1 = "big" (stem resolution), 2,3,4 = "small".
    // x lives in `xb`; conv1 -> T, shortcut -> S, conv2(+resid) -> Y.
    int H = cfg.height, W = cfg.width;
    int xb = 0;                                  // stem writes buffer 0
    for (int s = 0; s < 4; ++s) {
        const int c = w[s + 1];
        for (int u = 0; u < cfg.units[s]; ++u) {
            char pfx[64];
            snprintf(pfx, sizeof(pfx), "stage%d_unit%d", s + 1, u + 1);
            const std::string P(pfx);
            const int cin = (u == 0) ? w[s] : c;
            const int stride = (u == 0) ? 2 : 1;
            const int Ho = conv_out(H, 3, stride, 1), Wo = conv_out(W, 3, stride, 1);
            // pick three buffers different from xb
            int free_ids[4], nf = 0;
            for (int b = 1; b <= 4 && nf < 4; ++b)
                if (b != xb) free_ids[nf++] = b;
            // the first unit of stage 1 produces a stem-resolution conv1 output: it must use a big
            // buffer (0 or 1); xb is 0 there so T = 1.  Everything later fits the small buffers.
            const int tb = free_ids[0], sb = free_ids[1], yb = free_ids[2];

            const BN bn1 = get_bn(bb, P + "_bn1", false), bn2 = get_bn(bb, P + "_bn2", false),
                     bn3 = get_bn(bb, P + "_bn3", false);
            ConvLayer c1{};
            c1.name = P + "_conv1";
            c1.Cin = cin; c1.Cout = c; c1.ksz = 3; c1.stride = 1; c1.pad = 1;
            c1.Hin = H; c1.Win = W; c1.Hout = H; c1.Wout = W;
            c1.border_cls = true; c1.has_alpha = true;
            c1.in_buf = xb; c1.out_buf = tb; c1.resid_buf = -1; c1.role = 1; c1.unit = bb->n_units; c1.stage = s;
            if ((rc = build_conv(bb, c1, bb->raw.at(P + "_conv1_weight"), &bn1, bn2, &bb->raw.at(P + "_relu1_gamma"))))
                return rc;
            bb->convs.push_back(c1);

            int resid = xb;
            // the projection shortcut of a stage's first unit: fused into the conv2 launch (extra K-steps reading x) in the
            // 16-bit inference modes; a launch of its own where the backward pass needs it as a layer, and in split precision
            // (whose two products would need one common scale, fixed before calibration knows the tensors' exponents)
            const bool fuse_sc = u == 0 && !bb->grad && !x2 && bb->fuse_shortcut;
            if (u == 0 && !fuse_sc) {
                const BN bsc = get_bn(bb, P + "_sc", false);
                ConvLayer sc{};
                sc.name = P + "_conv1sc";
                sc.Cin = cin; sc.Cout = c; sc.ksz = 1; sc.stride = stride; sc.pad = 0;
                sc.Hin = H; sc.Win = W; sc.Hout = Ho; sc.Wout = Wo;
                sc.border_cls = false; sc.has_alpha = false;
                sc.in_buf = xb; sc.out_buf = sb; sc.resid_buf = -1; sc.role = 2; sc.unit = bb->n_units; sc.stage = s;
                if ((rc = build_conv(bb, sc, bb->raw.at(P + "_conv1sc_weight"), nullptr, bsc, nullptr))) return rc;
                bb->convs.push_back(sc);
                resid = sb;
            }
            ConvLayer c2{};
            c2.name = P + "_conv2";
            c2.Cin = c; c2.Cout = c; c2.ksz = 3; c2.stride = stride; c2.pad = 1;
            c2.Hin = H; c2.Win = W; c2.Hout = Ho; c2.Wout = Wo;
            c2.border_cls = false; c2.has_alpha = false;
            c2.in_buf = tb; c2.out_buf = yb; c2.resid_buf = resid; c2.role = 3; c2.unit = bb->n_units; c2.stage = s;
            if (fuse_sc) {
                const BN bsc = get_bn(bb, P + "_sc", false);
                c2.resid_buf = -1; c2.Cin2 = cin; c2.in2_buf = xb;
                if ((rc = build_conv(bb, c2, bb->raw.at(P + "_conv2_weight"), nullptr, bn3, nullptr, &bb->raw.at(P + "_conv1sc_weight"), &bsc)))
                    return rc;
            } else
            if ((rc = build_conv(bb, c2, bb->raw.at(P + "_conv2_weight"), nullptr, bn3, nullptr))) return rc;
            bb->convs.push_back(c2);
            ++bb->n_units;
            xb = yb;
            H = Ho;
            W = Wo;
        }
    }

    // ---- FC: flatten is (C,H,W) in the reference symbol; our activations are (H,W,C).
    {
        const int C = w[4], E = cfg.emb, HW = H * W, K = C * HW;
        const auto& fw = bb->raw.at("pre_fc1_weight");
        const auto& fb = bb->raw.at("pre_fc1_bias");
        const BN bnl = get_bn(bb, "bn1", false), bfc = get_bn(bb, "fc1", true);
        std::vector<uint16_t> wq((size_t)E * K * (x2 ? 2 : 1));
        std::vector<float> bias(E);
        if (x2) {
            double mx = 0.0;
            for (int o = 0; o < E; ++o)
                for (int ch = 0; ch < C; ++ch)
                    for (int pos = 0; pos < HW; ++pos)
                        mx = std::max(mx, std::fabs(bfc.a[o] * (double)fw[(size_t)o * K + (size_t)ch * HW + pos] * bnl.a[ch]));
            bb->fc_e_w = scale_exp(mx);
        }
        for (int o = 0; o < E; ++o) {
            const int row = (o & ~63) + perm64_row_of_channel(o & 63);
            double b = (double)fb[o];
            for (int ch = 0; ch < C; ++ch)
                for (int pos = 0; pos < HW; ++pos) {
                    const double wv = (double)fw[(size_t)o * K + (size_t)ch * HW + pos];
                    const size_t kk = (size_t)pos * C + ch;
                    if (x2) {
                        const size_t khi = (kk >> 6) * 128 + (kk & 63);
                        split16(std::ldexp(bfc.a[o] * wv * bnl.a[ch], bb->fc_e_w), &wq[(size_t)row * 2 * K + khi], &wq[(size_t)row * 2 * K + khi + 64]);
                    } else {
                        wq[(size_t)row * K + kk] = cvt(dt, (float)(bfc.a[o] * wv * bnl.a[ch]));
                    }
                    b += wv * bnl.b[ch];
                }
            bias[o] = (float)(bfc.a[o] * b + bfc.b[o]);
        }
        if ((rc = upload(bb, wq, &bb->d_fc_w))) return rc;
        if ((rc = upload(bb, bias, (void**)&bb->d_fc_bias))) return rc;
        if (bb->grad) {
            // d(x4)[pos][ch] = sum_o d(z)[o] * Wfolded[o][pos][ch]: a 1x1 "convolution" emb -> C*H*W
            std::vector<uint16_t> wb((size_t)K * E);
            for (int kk = 0; kk < K; ++kk) {
                const int pos = kk / C, ch = kk - pos * C;
                const size_t row = (size_t)((kk & ~63) + perm64_row_of_channel(kk & 63)) * E;
                for (int o = 0; o < E; ++o) {
                    const double wv = (double)fw[(size_t)o * K + (size_t)ch * HW + pos];
                    wb[row + o] = cvt(dt, (float)(bfc.a[o] * wv * bnl.a[ch]));
                }
            }
            if ((rc = upload(bb, wb, &bb->d_fc_wb))) return rc;
        }
        bb->fc_K = K;
        const int nk = K / 64;
        int S = nk / 14;                         // ~14 K-steps per split
        if (S < 1) S = 1;
        if (S > 64) S = 64;
        bb->fc_kps = (nk + S - 1) / S;                  // real K-steps per split (split precision walks 3 per real step)
        bb->fc_splitk = (nk + bb->fc_kps - 1) / bb->fc_kps;
        bb->Hf = H;
        bb->Wf = W;
    }
    for (int i = 0; i < alink_backbone::MAXSUB; ++i) {
        ALINK_HIP(hipStreamCreateWithFlags(&bb->sub[i], hipStreamNonBlocking));
        ALINK_HIP(hipEventCreateWithFlags(&bb->ev_done[i], hipEventDisableTiming));
        ALINK_HIP(hipEventCreateWithFlags(&bb->ev_front[i], hipEventDisableTiming));
    }
    ALINK_HIP(hipEventCreateWithFlags(&bb->ev_start, hipEventDisableTiming));
    bb->raw.clear();
    bb->finalized = true;
    return ALINK_OK;
}

int alink_backbone_set_streams(alink_backbone_t* bb, int n) {
    ALINK_REQUIRE(bb && n >= 1 && n <= alink_backbone::MAXSUB, ALINK_EINVAL, "streams must be 1..%d",
                  alink_backbone::MAXSUB);
    bb->nsub = n;
    return ALINK_OK;
}

// Small batches (N <= SPLIT_MAX_N) run their few-workgroup convolutions split over K (plan_split) into f32
// partial slabs of at most SPLIT_SLAB_BYTES (beyond that the slab traffic costs more than the shorter chain saves:
// measured, 64 images went from 3.8 to 4.3 ms with 51 MB slabs).
constexpr int SPLIT_MAX_N = 32;
constexpr size_t SPLIT_SLAB_BYTES = (size_t)8 << 20;

// workspace layout: [big0][big1][small2][small3][small4][fc slabs][conv split-K slabs (small N only)]
static void ws_layout(const alink_backbone* bb, int N, size_t off[7], size_t* total) {
    const alink_ir_cfg& c = bb->cfg;
    const size_t esz = c.dtype == ALINK_DT_F16X2 ? 4 : 2;          // split precision: an f16 pair per value
    // big buffers: the stem output (widths[0]) and the first unit's conv1 output (widths[1]) at input resolution
    const size_t big = (size_t)N * c.height * c.width * std::max(c.widths[0], c.widths[1]) * esz;
    // small buffers hold shortcut and unit outputs (conv1 outputs always go to big buffer 1)
    size_t small = 0;
    int H = c.height, W = c.width;
    for (int s = 0; s < 4; ++s) {
        const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
        small = std::max(small, (size_t)N * Ho * Wo * c.widths[s + 1] * esz);
        H = Ho;
        W = Wo;
    }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    off[0] = o; o += al(big);
    off[1] = o; o += al(big);
    for (int i = 2; i < 5; ++i) { off[i] = o; o += al(small); }
    off[5] = o; o += al((size_t)bb->fc_splitk * N * c.emb * 4);
    off[6] = o; o += (bb->split_small && N <= SPLIT_MAX_N) ? SPLIT_SLAB_BYTES : 0;
    *total = o;
}

// gradient-mode workspace, after the forward layout: [t cache per unit][norms][d(fc out)][5 gradient buffers]
struct GradLayout {
    std::vector<size_t> toff;
    size_t norms, dfc, g[5], total;
};
static void grad_layout(const alink_backbone* bb, int N, GradLayout* L) {
    size_t off[7], o;
    ws_layout(bb, N, off, &o);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L->toff.assign(bb->n_units, 0);
    for (const ConvLayer& c : bb->convs)
        if (c.role == 1) { L->toff[c.unit] = o; o += al((size_t)N * c.Hout * c.Wout * c.Cout * 2); }
    L->norms = o; o += al((size_t)N * 4);
    L->dfc = o; o += al((size_t)N * bb->cfg.emb * 2);
    const size_t big = al((size_t)N * bb->cfg.height * bb->cfg.width * bb->cfg.widths[0] * 2);
    for (int i = 0; i < 5; ++i) { L->g[i] = o; o += big; }
    L->total = o;
}

// shard sizes of an N-image forward: at least 64 images per shard, at most bb->nsub shards
static int split_plan(const alink_backbone* bb, int N, int counts[alink_backbone::MAXSUB]) {
    int S = bb->nsub;
    while (S > 1 && N / S < 64) --S;
    for (int i = 0; i < S; ++i) counts[i] = N / S + (i < N % S ? 1 : 0);
    return S;
}

size_t alink_backbone_workspace_bytes(const alink_backbone_t* bb, int n_images) {
    if (!bb || !bb->finalized || n_images <= 0) return 0;
    if (bb->f32) return f32net_workspace_bytes(bb->f32, n_images);
    size_t off[7], total, single;
    ws_layout(bb, n_images, off, &single);
    // any stream count up to MAXSUB may be selected later: take the worst case
    size_t worst = single;
    for (int S = 2; S <= alink_backbone::MAXSUB; ++S) {
        if (n_images / S < 64) break;
        size_t sum = 0;
        for (int i = 0; i < S; ++i) {
            ws_layout(bb, n_images / S + (i < n_images % S ? 1 : 0), off, &total);
            sum += total;
        }
        worst = std::max(worst, sum);
    }
    return worst;
}

// Split factor of one convolution at batch N (1 = fused launch).  A 3x3 layer at batch 1..16 has a handful of
// workgroups, each walking all of K (stage 3: 2 workgroups x 36 K-steps on a 256-CU chip); splitting K over
// up to 8 workgroup rows that leave f32 slabs, summed in order by conv_split_finish_kernel, cuts the
// dependent chain per layer by the same factor.  Only when the fused grid covers at most half the CUs.
static int plan_split(const alink_backbone* bb, const ConvLayer& L, int N) {
    if (!bb->split_small || N > SPLIT_MAX_N || L.Cin2) return 1;     // a fused shortcut is not split over K
    const long long M = (long long)N * L.Hout * L.Wout;
    const int ncc = L.Cin / 64;
    long long nwg;
    int units, kpu;                              // what a split divides: input chunks of 9 K-steps (linear) or K-steps (igemm)
    if (L.variant >= 11 && L.variant <= 15) {
        const int bn = (L.variant == 13 || L.variant == 15) ? 64 : 128;     // (the 64-channel form chosen for small batches doubles nwg: still <= 512)
        nwg = ((M + 223) / 224) * (L.Cout / bn);
        units = ncc; kpu = 9;
    } else if (L.variant == 0) {
        const bool wide = (L.Cout % 128) == 0;
        nwg = wide ? ((M + 127) / 128) * (L.Cout / 128) : ((M + 255) / 256) * (L.Cout / 64);
        units = L.ksz * L.ksz * ncc; kpu = 1;
    } else {
        return 1;
    }
    if (nwg > 64) return 1;
    int best = 1;
    for (int S = 2; S <= 8; ++S)
        if (units % S == 0 && units / S * kpu >= 4 && nwg * S <= 512 && (size_t)S * M * L.Cout * 4 <= SPLIT_SLAB_BYTES) best = S;
    return best;
}

int g_sibling_aware = 1;
extern "C" void alink_debug_set_sibling_aware(int on) { g_sibling_aware = on != 0; }
int g_fine_max = 384;     // measured (r100, one launch at a time): the 64-channel form wins while the 128-channel grid fills < 3/4 of the 512 slots
extern "C" void alink_debug_set_fine_max(int n) { g_fine_max = n; }
int g_ablate = 0;
int g_stop_after = 0;       // diagnostic: alink_embed returns after this many convolution launches (0 = the whole chain)
extern "C" void alink_debug_set_stop_after(int n) { g_stop_after = n; }
int g_stagger = 0;
int g_shard_stagger = 0;
extern "C" void alink_debug_set_shard_stagger(int on) { g_shard_stagger = on; }

extern "C" void alink_debug_set_stagger(int n) { g_stagger = n < 0 ? 0 : n; }
void* g_stamps = nullptr;
// alink_embed_profile launches every kernel of the chain this many times back to back between its two
// events (all launches are idempotent: no kernel writes a buffer it reads) and reports the mean, so
// that the event pair's own cost and the idle gap it opens are not booked as kernel time.
int g_prof_reps = 4;
extern "C" void alink_debug_set_profile_reps(int n) { g_prof_reps = n < 1 ? 1 : n; }
extern "C" void alink_debug_set_ablate(int a) { g_ablate = a; }
extern "C" void alink_debug_set_stamps(void* p) { g_stamps = p; }

// calib (ALINK_DT_F16X2 only): 0 = a normal forward; 1 = choose every tensor's scale exponent from this batch; 2 = the
// same, never above the exponents already held (re-calibration after a batch left the range).  Synchronous when != 0.
static int embed_impl(alink_backbone_t* bb, const void* dev_in, int layout, int N, float* dev_out,
                      void* ws, size_t ws_bytes, hipStream_t stream, float* ms, double* flops, int* kind,
                      int* n_launches, const GradLayout* cache = nullptr, int calib = 0, hipEvent_t front_done = nullptr) {
    ALINK_REQUIRE(bb && dev_in && dev_out && ws, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(bb->finalized, ALINK_ESTATE, "alink_embed before alink_backbone_finalize");
    ALINK_REQUIRE(N > 0, ALINK_EINVAL, "n_images must be positive");
    ALINK_REQUIRE(layout >= 0 && layout <= 2, ALINK_EINVAL, "unknown pixel layout %d", layout);
    const alink_ir_cfg& cfg = bb->cfg;
    ALINK_REQUIRE((long long)N * cfg.height * cfg.width * 64 * (cfg.dtype == ALINK_DT_F16X2 ? 2 : 1) < (1ll << 31), ALINK_EINVAL,
                  "batch of %d images exceeds the 2^31-element activation limit; split the batch", N);
    size_t off[7], total;
    ws_layout(bb, N, off, &total);
    if (cache) total = cache->total;
    ALINK_REQUIRE(ws_bytes >= total, ALINK_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, total);
    ALINK_REQUIRE(((uintptr_t)ws & 255) == 0, ALINK_EINVAL, "workspace must be 256-byte aligned");
    char* base = (char*)ws;
    auto buf = [&](int id) -> void* { return base + off[id]; };

    const bool prof = ms != nullptr;
    const int reps = prof ? g_prof_reps : 1;
    const int cap = prof ? *n_launches : 0;
    int nl = 0;
    std::vector<hipEvent_t> ev;
    auto mark = [&]() -> int {
        if (!prof) return ALINK_OK;
        hipEvent_t e;
        ALINK_HIP(hipEventCreate(&e));
        ALINK_HIP(hipEventRecord(e, stream));
        ev.push_back(e);
        return ALINK_OK;
    };
    auto note = [&](double f, int k) {
        if (prof && nl < cap) { flops[nl] = f; kind[nl] = k; }
        ++nl;
    };
    int rc;
    if ((rc = mark())) return rc;

    const bool x2 = cfg.dtype == ALINK_DT_F16X2;
    ALINK_REQUIRE(!calib || x2, ALINK_ESTATE, "only the split-precision mode is calibrated");
    ALINK_REQUIRE(!x2 || calib || bb->calibrated, ALINK_ESTATE, "split-precision backbone: alink_backbone_calibrate has not run");
    int bexp[5] = {0, 0, 0, 0, 0};        // split precision: scale exponent of the tensor each workspace buffer holds
    // calibration: run `launch(e)` with the output exponent e until the largest |output| lies in [1024, 2048) (f16 pairs:
    // 32x below overflow, lo halves normal); a power-of-two scale changes no bit of the result, only where it sits
    auto settle = [&](int* e_io, const void* out, size_t n_elems, auto&& launch) -> int {
        int e = *e_io;
        for (int attempt = 0; attempt < 24; ++attempt) {
            const int rcl = launch(e);
            if (rcl) return rcl;
            if (!calib) break;
            unsigned bits = 0;
            ALINK_HIP(hipMemsetAsync(bb->d_absmax, 0, 4, stream));
            ALINK_HIP(launch_absmax_f16(out, n_elems, bb->d_absmax, stream));
            ALINK_HIP(hipMemcpyAsync(&bits, bb->d_absmax, 4, hipMemcpyDeviceToHost, stream));
            ALINK_HIP(hipStreamSynchronize(stream));
            float m;
            memcpy(&m, &bits, 4);
            if (bits >= 0x7f800000u) { e -= 8; continue; }                 // left the range: lower the scale and redo
            if (m == 0.f) break;
            int want = e + (10 - std::ilogb(m));
            if (calib == 2 && bb->calibrated) want = std::min(want, *e_io);
            if (want == e) break;
            e = want;
        }
        *e_io = e;
        return ALINK_OK;
    };

    StemParams sp{};
    sp.in = dev_in; sp.wgt = bb->d_stem_w; sp.bias = bb->d_stem_bias; sp.alpha = bb->d_stem_alpha;
    sp.out = buf(0); sp.N = N; sp.H = cfg.height; sp.W = cfg.width; sp.C0 = 64; sp.layout = layout;
    sp.sub[0] = sp.sub[1] = sp.sub[2] = 127.5f; sp.mul = 0.0078125f; sp.flip = 0;
    // The front in one launch (front_c64.hip): stem and the first unit's conv1, the stem's activation kept in LDS; the
    // quarter of it that the unit's projection shortcut samples lands in buffer 0 as a compact tensor.  16-bit inference
    // with the shortcut fused into conv2 only: the gradient pass, split precision and a stand-alone shortcut layer read
    // the whole stem activation.
    const bool front = !x2 && !cache && bb->convs.size() >= 2 && bb->convs[0].variant == 21 && bb->convs[0].in_buf == 0 &&
                       bb->convs[1].Cin2 == 64 && bb->convs[1].in2_buf == 0 &&
                       front_c64_applies(cfg.dtype, cfg.height, cfg.width, 64, bb->convs[0].Cout) &&
                       plan_split(bb, bb->convs[0], N) == 1;
    if (front) {
        // launched with conv1 below
    } else if (x2) {
        rc = settle(&bb->stem_e_out, buf(0), (size_t)N * cfg.height * cfg.width * 128, [&](int e) -> int {
            sp.acc_scale = std::ldexp(1.f, e - 8 - bb->stem_e_w);          // the loader stores normalised pixels x 2^8
            sp.bias_scale = std::ldexp(1.f, e);
            for (int r = 0; r < reps; ++r) ALINK_HIP(launch_stem(cfg.dtype, sp, stream));
            return ALINK_OK;
        });
        if (rc) return rc;
        bexp[0] = bb->stem_e_out;
    } else {
        for (int r = 0; r < reps; ++r) ALINK_HIP(launch_stem(cfg.dtype, sp, stream));
    }
    if (!front) {
        note(2.0 * N * cfg.height * cfg.width * 64.0 * 27.0, 0);
        if ((rc = mark())) return rc;
    }

    int last_out = 0, n_done = 0;
    bool front_marked = false;
    for (ConvLayer& L : bb->convs) {
        if (g_stop_after && n_done++ >= g_stop_after) return ALINK_OK;
        if (front_done && !front_marked && L.stage >= 1) {       // everything before this launch is the HBM-bound front
            ALINK_HIP(hipEventRecord(front_done, stream));
            front_marked = true;
        }
        ConvParams p{};
        p.in = buf(L.in_buf); p.wgt = L.d_w; p.bias = L.d_bias; p.alpha = L.d_alpha;
        p.resid = L.resid_buf >= 0 ? buf(L.resid_buf) : nullptr;
        p.out = buf(L.out_buf); p.zero = bb->d_zero;
        if (cache) {   // keep every unit's conv1 activation for the PReLU derivative of the backward pass
            if (L.role == 1) p.out = base + cache->toff[L.unit];
            if (L.role == 3) p.in = base + cache->toff[L.unit];
        }
        p.N = N; p.H = L.Hin; p.W = L.Win; p.Cin = L.Cin; p.Cout = L.Cout; p.Ho = L.Hout; p.Wo = L.Wout;
        p.stride = L.stride; p.ksz = L.ksz; p.pad = L.pad; p.M = N * L.Hout * L.Wout;
        p.border_cls = L.border_cls ? 1 : 0; p.splitk = 1;
        p.ksteps_per_split = L.ksz * L.ksz * (L.Cin / 64) + L.Cin2 / 64;
        if (L.Cin2) { p.in2 = buf(L.in2_buf); p.Cin2 = L.Cin2; p.in2_compact = (front && &L == &bb->convs[1]) ? 1 : 0; }
        p.ablate = g_ablate;
        p.stagger = g_stagger;
        p.nprod = (x2 && !calib) ? bb->products : 0;
        // few images (128-channel grid under 3/4 of the chip, g_fine_max): those workgroups cover only part of the chip and
        // each walks all of K alone on its CU; the 64-channel form doubles their number and halves a K-step
        // (bit-identical results: same weights, same summation order per output)
        if (L.variant == 11 || L.variant == 12 || L.variant == 14) {
            const long long nwg128 = (((long long)p.M + 223) / 224) * (L.Cout / 128);
            // (a shard of an in-call split runs beside its siblings: it is their workgroups together that fill the chip —
            // measured for the 14- and 28-wide layers: IR-50, one 256-image batch +1.5 %, IR-100 at 292 images +1.7 %; the
            // 7-wide layers, 72 K-steps per workgroup, do better blind: −1 % otherwise at two 292-image shards)
            const int sib = (g_sibling_aware && L.variant != 14) ? bb->siblings : 1;
            p.fine = nwg128 * sib <= g_fine_max ? 1 : 0;
        }
        int S = plan_split(bb, L, N);
        // the opt-in K split (order-changing) gives way where the bit-identical latency form applies: that one is faster
        if (S > 1) {
            ConvParams probe = p;
            probe.splitk = 1;
            if (L.variant >= 11 ? conv3x3_lat_applies(cfg.dtype, probe) : (L.variant == 0 && conv_gemm_lat_applies(cfg.dtype, probe))) S = 1;
        }
        if (x2) {
            p.ksteps_per_split *= 3;
            rc = settle(&L.e_out, p.out, (size_t)p.M * L.Cout * 2, [&](int e) -> int {
                p.acc_scale = std::ldexp(1.f, e - bexp[L.in_buf] - L.e_w);
                p.bias_scale = std::ldexp(1.f, e);
                p.res_scale = L.resid_buf >= 0 ? std::ldexp(1.f, e - bexp[L.resid_buf]) : 1.f;
                for (int r = 0; r < reps; ++r) {
                    ConvParams q = p;
                    if (S > 1) {           // latency mode: K split into f32 slabs of raw accumulators, scales applied by the finish kernel
                        q.out = buf(6); q.splitk = S; q.ksteps_per_split = p.ksteps_per_split / S;
                    }
                    if (L.variant) ALINK_HIP(launch_conv3x3_direct(L.variant, cfg.dtype, q, stream));
                    else           ALINK_HIP(launch_conv_igemm(cfg.dtype, q, stream));
                    if (S > 1) ALINK_HIP(launch_conv_split_finish(cfg.dtype, p, (const float*)buf(6), S, stream));
                }
                return ALINK_OK;
            });
            if (rc) return rc;
            bexp[L.out_buf] = L.e_out;
        } else if (front && &L == &bb->convs[0]) {
            p.stamps = g_stamps;
            for (int r = 0; r < reps; ++r) ALINK_HIP(launch_front_c64(cfg.dtype, p, sp, buf(0), bb->front_slopes_le_1, stream));
            note(conv_flops(p) + 2.0 * N * cfg.height * cfg.width * 64.0 * 27.0, 1);
            if ((rc = mark())) return rc;
            last_out = L.out_buf;
            continue;
        } else
        for (int r = 0; r < reps; ++r) {
            ConvParams q = p;
            if (S > 1) {
                q.out = buf(6); q.splitk = S; q.ksteps_per_split = p.ksteps_per_split / S;
            }
            if (L.variant) ALINK_HIP(launch_conv3x3_direct(L.variant, cfg.dtype, q, stream));
            else           ALINK_HIP(launch_conv_igemm(cfg.dtype, q, stream));
            if (S > 1) ALINK_HIP(launch_conv_split_finish(cfg.dtype, p, (const float*)buf(6), S, stream));
        }
        note(conv_flops(p), 1);
        if ((rc = mark())) return rc;
        last_out = L.out_buf;
    }

    // FC as a 1x1 "convolution" over N pixels with Cin = C*Hf*Wf, split-K into f32 slabs
    {
        ConvParams p{};
        p.in = buf(last_out); p.wgt = bb->d_fc_w; p.bias = nullptr; p.alpha = nullptr; p.resid = nullptr;
        p.out = buf(5); p.zero = bb->d_zero;
        p.N = N; p.H = 1; p.W = 1; p.Cin = bb->fc_K; p.Cout = cfg.emb; p.Ho = 1; p.Wo = 1;
        p.stride = 1; p.ksz = 1; p.pad = 0; p.M = N; p.border_cls = 0;
        p.splitk = bb->fc_splitk; p.ksteps_per_split = bb->fc_kps * (x2 ? 3 : 1);
        p.nprod = (x2 && !calib) ? bb->products : 0;
        ALINK_REQUIRE(p.splitk > 1, ALINK_EINVAL, "FC split-K must be > 1 (K=%d)", bb->fc_K);
        for (int r = 0; r < reps; ++r) ALINK_HIP(launch_conv_igemm(cfg.dtype, p, stream));
        note(conv_flops(p), 2);
        if ((rc = mark())) return rc;
        FcFinishParams f{};
        f.slabs = (const float*)buf(5); f.bias = bb->d_fc_bias; f.out = dev_out;
        f.norms = cache ? (float*)(base + cache->norms) : nullptr;
        f.S = bb->fc_splitk; f.M = N; f.E = cfg.emb;
        f.scale = x2 ? std::ldexp(1.f, -(bexp[last_out] + bb->fc_e_w)) : 1.f;
        f.nonfinite = bb->d_flag;       // 16-bit storage: an activation beyond the range ends as a non-finite embedding
        for (int r = 0; r < reps; ++r) ALINK_HIP(launch_fc_finish(f, stream));
        note(0.0, 3);
        if ((rc = mark())) return rc;
    }

    if (prof) {
        ALINK_HIP(hipStreamSynchronize(stream));
        for (int i = 0; i + 1 < (int)ev.size() && i < cap; ++i) {
            float t = 0.f;
            ALINK_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
            ms[i] = t / (float)reps;
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        *n_launches = nl;
    }
    if (calib) {
        ALINK_HIP(hipStreamSynchronize(stream));
        ALINK_REQUIRE(*bb->h_flag == 0, ALINK_EINVAL, "calibration batch produced non-finite embeddings");
        bb->calibrated = true;
    }
    return ALINK_OK;
}

int alink_embed(alink_backbone_t* bb, const void* dev_in, int layout, int n_images, float* dev_out,
                void* dev_workspace, size_t workspace_bytes, void* stream) {
    ALINK_REQUIRE(bb && bb->finalized, ALINK_ESTATE, "alink_embed before alink_backbone_finalize");
    DeviceGuard dg(bb->device);
    ALINK_REQUIRE(n_images > 0, ALINK_EINVAL, "n_images must be positive");
    ALINK_REQUIRE(layout >= 0 && layout <= 2, ALINK_EINVAL, "unknown pixel layout %d", layout);
    if (bb->f32) return f32net_embed(bb->f32, dev_in, layout, n_images, dev_out, dev_workspace, workspace_bytes, (hipStream_t)stream);
    int counts[alink_backbone::MAXSUB];
    const int S = split_plan(bb, n_images, counts);
    hipStream_t st = (hipStream_t)stream;
    if (S == 1)
        return embed_impl(bb, dev_in, layout, n_images, dev_out, dev_workspace, workspace_bytes, st, nullptr,
                          nullptr, nullptr, nullptr);
    ALINK_REQUIRE(dev_workspace && ((uintptr_t)dev_workspace & 255) == 0, ALINK_EINVAL,
                  "workspace must be 256-byte aligned");
    const size_t px_bytes = (size_t)bb->cfg.height * bb->cfg.width * 3 * (layout == ALINK_LAYOUT_NHWC_U8 ? 1 : 4);
    ALINK_HIP(hipEventRecord(bb->ev_start, st));
    size_t woff = 0;
    int n0 = 0;
    bb->siblings = S;
    struct Reset { alink_backbone* b; ~Reset() { b->siblings = 1; } } reset{bb};
    for (int i = 0; i < S; ++i) {
        size_t off[7], need;
        ws_layout(bb, counts[i], off, &need);
        ALINK_REQUIRE(woff + need <= workspace_bytes, ALINK_ENOMEM, "workspace too small: %zu < %zu", workspace_bytes,
                      woff + need);
        ALINK_HIP(hipStreamWaitEvent(bb->sub[i], bb->ev_start, 0));
        // staggered shards (g_shard_stagger): shard i starts when shard i - 1 has got through its HBM-bound front (stem +
        // stage 1), so that one shard's front runs beside the other's MFMA-bound stages instead of beside its front
        if (g_shard_stagger && i > 0) ALINK_HIP(hipStreamWaitEvent(bb->sub[i], bb->ev_front[i - 1], 0));
        const int rc = embed_impl(bb, (const char*)dev_in + (size_t)n0 * px_bytes, layout, counts[i],
                                  dev_out + (size_t)n0 * bb->cfg.emb, (char*)dev_workspace + woff, need, bb->sub[i],
                                  nullptr, nullptr, nullptr, nullptr, nullptr, 0, g_shard_stagger ? bb->ev_front[i] : nullptr);
        if (rc) return rc;
        ALINK_HIP(hipEventRecord(bb->ev_done[i], bb->sub[i]));
        ALINK_HIP(hipStreamWaitEvent(st, bb->ev_done[i], 0));
        woff += need;
        n0 += counts[i];
    }
    return ALINK_OK;
}

int alink_embed_profile(alink_backbone_t* bb, const void* dev_in, int layout, int n_images, float* dev_out,
                        void* dev_workspace, size_t workspace_bytes, void* stream, float* ms, double* flops,
                        int* kind, int* n_launches) {
    ALINK_REQUIRE(ms && flops && kind && n_launches && *n_launches > 0, ALINK_EINVAL, "NULL profile buffers");
    ALINK_REQUIRE(bb, ALINK_EINVAL, "NULL backbone");
    ALINK_REQUIRE(!bb->f32, ALINK_ESTATE, "per-launch profiling is for the bf16 / f16 kernels, not the float32 mode");
    DeviceGuard dg(bb->device);
    return embed_impl(bb, dev_in, layout, n_images, dev_out, dev_workspace, workspace_bytes,
                      (hipStream_t)stream, ms, flops, kind, n_launches);
}

int alink_backbone_calibrate(alink_backbone_t* bb, const void* dev_in, int layout, int n_images, void* dev_workspace,
                             size_t workspace_bytes, int merge, void* stream) {
    ALINK_REQUIRE(bb && bb->finalized, ALINK_ESTATE, "alink_backbone_calibrate before alink_backbone_finalize");
    ALINK_REQUIRE(bb->cfg.dtype == ALINK_DT_F16X2, ALINK_ESTATE, "only the split-precision mode (ALINK_DT_F16X2) is calibrated");
    ALINK_REQUIRE(dev_in && dev_workspace && n_images > 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(bb->device);
    hipStream_t st = (hipStream_t)stream;
    ALINK_HIP(hipStreamSynchronize(st));
    // a report still pending from an earlier forward (lazy range checks read the flag later) must survive the calibration run
    const int pending = *(volatile int*)bb->h_flag;
    struct Keep { int* f; int v; ~Keep() { if (v) *(volatile int*)f = 1; } } keep{bb->h_flag, pending};
    *bb->h_flag = 0;
    float* scratch = nullptr;           // the embeddings of the calibration batch are not wanted
    ALINK_HIP(hipMalloc((void**)&scratch, (size_t)n_images * bb->cfg.emb * sizeof(float)));
    const int rc = embed_impl(bb, dev_in, layout, n_images, scratch, dev_workspace, workspace_bytes, st, nullptr, nullptr,
                              nullptr, nullptr, nullptr, merge ? 2 : 1);
    (void)hipFree(scratch);
    return rc;
}

int alink_backbone_num_scales(const alink_backbone_t* bb) {
    if (!bb || !bb->finalized || bb->cfg.dtype != ALINK_DT_F16X2) return 0;
    return 1 + (int)bb->convs.size();
}

int alink_backbone_get_scales(const alink_backbone_t* bb, int* exponents, int n) {
    ALINK_REQUIRE(bb && bb->finalized, ALINK_ESTATE, "alink_backbone_get_scales before alink_backbone_finalize");
    ALINK_REQUIRE(bb->cfg.dtype == ALINK_DT_F16X2, ALINK_ESTATE, "only the split-precision mode (ALINK_DT_F16X2) has scales");
    ALINK_REQUIRE(bb->calibrated, ALINK_ESTATE, "alink_backbone_get_scales before alink_backbone_calibrate / set_scales");
    ALINK_REQUIRE(exponents && n == 1 + (int)bb->convs.size(), ALINK_EINVAL, "expected room for %d exponents, got %d",
                  1 + (int)bb->convs.size(), n);
    exponents[0] = bb->stem_e_out;
    for (size_t i = 0; i < bb->convs.size(); ++i) exponents[1 + i] = bb->convs[i].e_out;
    return ALINK_OK;
}

int alink_backbone_set_scales(alink_backbone_t* bb, const int* exponents, int n) {
    ALINK_REQUIRE(bb && bb->finalized, ALINK_ESTATE, "alink_backbone_set_scales before alink_backbone_finalize");
    ALINK_REQUIRE(bb->cfg.dtype == ALINK_DT_F16X2, ALINK_ESTATE, "only the split-precision mode (ALINK_DT_F16X2) has scales");
    ALINK_REQUIRE(exponents && n == 1 + (int)bb->convs.size(), ALINK_EINVAL, "expected %d exponents, got %d",
                  1 + (int)bb->convs.size(), n);
    for (int i = 0; i < n; ++i)
        ALINK_REQUIRE(exponents[i] >= -126 && exponents[i] <= 126, ALINK_EINVAL, "exponent %d of tensor %d is not a float32 power of two", exponents[i], i);
    bb->stem_e_out = exponents[0];
    for (size_t i = 0; i < bb->convs.size(); ++i) bb->convs[i].e_out = exponents[1 + i];
    bb->calibrated = true;
    return ALINK_OK;
}

int alink_backbone_device(const alink_backbone_t* bb) { return bb ? bb->device : -1; }

int alink_backbone_range_flag(alink_backbone_t* bb, int reset) {
    ALINK_REQUIRE(bb && bb->finalized, ALINK_ESTATE, "alink_backbone_range_flag before alink_backbone_finalize");
    if (!bb->h_flag) return 0;          // float32 mode: nothing to leave
    const int v = *(volatile int*)bb->h_flag != 0 ? 1 : 0;
    if (reset) *(volatile int*)bb->h_flag = 0;
    return v;
}

// ---- input gradient (FGSM / PGD extension) ------------------------------------------------------------
int alink_backbone_set_small_batch_split(alink_backbone_t* bb, int on) {
    ALINK_REQUIRE(bb, ALINK_EINVAL, "NULL backbone");
    bb->split_small = on != 0;
    return ALINK_OK;
}

int alink_backbone_set_products(alink_backbone_t* bb, int n) {
    ALINK_REQUIRE(bb && bb->finalized, ALINK_ESTATE, "alink_backbone_set_products before alink_backbone_finalize");
    ALINK_REQUIRE(bb->cfg.dtype == ALINK_DT_F16X2, ALINK_ESTATE, "only the split-precision mode (ALINK_DT_F16X2) has a product count");
    ALINK_REQUIRE(n == 1 || n == 3, ALINK_EINVAL, "products must be 3 (exact) or 1 (screening), got %d", n);
    bb->products = n;
    return ALINK_OK;
}

int alink_backbone_enable_grad(alink_backbone_t* bb) {
    ALINK_REQUIRE(bb, ALINK_EINVAL, "NULL backbone");
    ALINK_REQUIRE(!bb->finalized, ALINK_ESTATE, "alink_backbone_enable_grad must precede alink_backbone_finalize");
    bb->grad = true;
    return ALINK_OK;
}

size_t alink_backbone_grad_workspace_bytes(const alink_backbone_t* bb, int n_images) {
    if (!bb || !bb->finalized || !bb->grad || n_images <= 0) return 0;
    GradLayout L;
    grad_layout(bb, n_images, &L);
    return L.total;
}

int alink_embed_cached(alink_backbone_t* bb, const void* dev_in, int layout, int n_images, float* dev_out,
                       void* dev_workspace, size_t workspace_bytes, void* stream) {
    ALINK_REQUIRE(bb && bb->finalized && bb->grad, ALINK_ESTATE, "needs alink_backbone_enable_grad + finalize");
    DeviceGuard dg(bb->device);
    GradLayout L;
    grad_layout(bb, n_images > 0 ? n_images : 1, &L);
    return embed_impl(bb, dev_in, layout, n_images, dev_out, dev_workspace, workspace_bytes, (hipStream_t)stream,
                      nullptr, nullptr, nullptr, nullptr, &L);
}

int alink_embed_input_grad(alink_backbone_t* bb, const float* dev_demb, const float* dev_emb, int layout, int N,
                           float* dev_dpix, void* ws, size_t ws_bytes, void* stream) {
    ALINK_REQUIRE(bb && bb->finalized && bb->grad, ALINK_ESTATE, "needs alink_backbone_enable_grad + finalize");
    DeviceGuard dg(bb->device);
    ALINK_REQUIRE(dev_demb && dev_emb && dev_dpix && ws && N > 0, ALINK_EINVAL, "bad argument");
    ALINK_REQUIRE(layout == ALINK_LAYOUT_NHWC_F32 || layout == ALINK_LAYOUT_NCHW_F32, ALINK_EINVAL,
                  "gradient layout must be float32 NHWC or NCHW");
    const alink_ir_cfg& cfg = bb->cfg;
    const int dt = cfg.dtype;
    hipStream_t st = (hipStream_t)stream;
    GradLayout GL;
    grad_layout(bb, N, &GL);
    ALINK_REQUIRE(ws_bytes >= GL.total, ALINK_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, GL.total);
    char* base = (char*)ws;
    auto G = [&](int i) -> void* { return base + GL.g[i]; };
    // 1) through the L2 normalisation, 2) through the folded FC
    ALINK_HIP(launch_l2norm_bwd(dt, dev_demb, dev_emb, (const float*)(base + GL.norms), base + GL.dfc, N, cfg.emb, st));
    int cur = 0;
    {
        ConvParams p{};
        p.in = base + GL.dfc; p.wgt = bb->d_fc_wb; p.bias = bb->d_zero_bias; p.out = G(cur); p.zero = bb->d_zero;
        p.N = N; p.H = 1; p.W = 1; p.Cin = cfg.emb; p.Cout = bb->fc_K; p.Ho = 1; p.Wo = 1; p.stride = 1; p.ksz = 1;
        p.pad = 0; p.M = N; p.splitk = 1; p.ksteps_per_split = cfg.emb / 64;
        ALINK_HIP(launch_conv_igemm(dt, p, st));
    }
    // 3) residual units in reverse
    std::vector<const ConvLayer*> c1(bb->n_units, nullptr), sc(bb->n_units, nullptr), c2(bb->n_units, nullptr);
    for (const ConvLayer& L : bb->convs) (L.role == 1 ? c1 : (L.role == 2 ? sc : c2))[L.unit] = &L;
    for (int u = bb->n_units - 1; u >= 0; --u) {
        const ConvLayer &A = *c1[u], &B = *c2[u];
        const int H = A.Hin, W = A.Win, c = A.Cout, cin = A.Cin, Ho = B.Hout, Wo = B.Wout;
        int ids[4], k = 0;
        for (int i = 0; i < 5; ++i) if (i != cur) ids[k++] = i;
        const void* dy = G(cur);
        const void* dy_full = dy;                       // d(output) on the unit's input grid
        if (B.stride == 2) {
            ALINK_HIP(launch_zero_insert(dt, dy, G(ids[0]), N, H, W, Ho, Wo, c, st));
            dy_full = G(ids[0]);
        }
        {   // through conv2 (+ bn3) and the PReLU: d(z1) = conv(dy, W2b) * PReLU'(t)
            ConvParams p{};
            p.in = dy_full; p.wgt = B.d_wb; p.bias = bb->d_zero_bias; p.alpha = A.d_alpha;
            p.dact = base + GL.toff[u]; p.out = G(ids[1]); p.zero = bb->d_zero;
            p.N = N; p.H = H; p.W = W; p.Cin = c; p.Cout = c; p.Ho = H; p.Wo = W; p.stride = 1; p.ksz = 3; p.pad = 1;
            p.M = N * H * W; p.splitk = 1; p.ksteps_per_split = 9 * (c / 64);
            if (B.bvariant) ALINK_HIP(launch_conv3x3_direct(B.bvariant, dt, p, st));
            else            ALINK_HIP(launch_conv_igemm(dt, p, st));
        }
        const void* r = dy;                             // gradient arriving through the shortcut
        if (sc[u]) {
            const ConvLayer& S = *sc[u];
            ConvParams p{};
            p.in = dy; p.wgt = S.d_wb; p.bias = bb->d_zero_bias; p.out = G(ids[0]); p.zero = bb->d_zero;
            p.N = N; p.H = Ho; p.W = Wo; p.Cin = c; p.Cout = cin; p.Ho = Ho; p.Wo = Wo; p.stride = 1; p.ksz = 1; p.pad = 0;
            p.M = N * Ho * Wo; p.splitk = 1; p.ksteps_per_split = c / 64;
            ALINK_HIP(launch_conv_igemm(dt, p, st));
            ALINK_HIP(launch_zero_insert(dt, G(ids[0]), G(ids[2]), N, H, W, Ho, Wo, cin, st));
            r = G(ids[2]);
        }
        {   // through conv1 (+ bn1, bn2 folded) and add the shortcut gradient
            ConvParams p{};
            p.in = G(ids[1]); p.wgt = A.d_wb; p.bias = bb->d_zero_bias; p.resid = r; p.out = G(ids[3]); p.zero = bb->d_zero;
            p.N = N; p.H = H; p.W = W; p.Cin = c; p.Cout = cin; p.Ho = H; p.Wo = W; p.stride = 1; p.ksz = 3; p.pad = 1;
            p.M = N * H * W; p.splitk = 1; p.ksteps_per_split = 9 * (c / 64);
            if (A.bvariant) ALINK_HIP(launch_conv3x3_direct(A.bvariant, dt, p, st));
            else            ALINK_HIP(launch_conv_igemm(dt, p, st));
        }
        cur = ids[3];
    }
    // 4) through the stem PReLU, the stem convolution and the input normalisation
    size_t off[7], fwd_total;
    ws_layout(bb, N, off, &fwd_total);
    ALINK_HIP(launch_stem_bwd(dt, G(cur), base + off[0], bb->d_stem_wf, bb->d_stem_alpha, dev_dpix, N, cfg.height,
                              cfg.width, 0.0078125f, layout == ALINK_LAYOUT_NCHW_F32, st));
    return ALINK_OK;
}

// ---- diagnostic single-convolution entry (unit tests) -------------------------------------------
int alink_conv_nhwc(int dtype, const void* dev_in, const void* dev_w, const float* dev_bias,
                    const float* dev_alpha, const void* dev_resid, void* dev_out, int N, int H, int W, int Cin,
                    int Cout, int ksz, int stride, int pad, int border_cls, int fine, void* stream) {
    ALINK_REQUIRE(dev_in && dev_w && dev_bias && dev_out, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(device_of_pointer(dev_in));
    ALINK_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, ALINK_EINVAL, "Cin/Cout must be multiples of 64");
    ALINK_REQUIRE(ksz == 1 || ksz == 3, ALINK_EINVAL, "ksz must be 1 or 3");
    ALINK_REQUIRE(!border_cls || (ksz == 3 && stride == 1 && pad == 1), ALINK_EINVAL,
                  "border classes need a 3x3 stride-1 pad-1 convolution");
    int rc = init_kernels();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int K = ksz * ksz * Cin;
    // permute weight rows into a private copy
    std::vector<uint16_t> h((size_t)Cout * K), hp((size_t)Cout * K);
    ALINK_HIP(hipStreamSynchronize(st));
    ALINK_HIP(hipMemcpy(h.data(), dev_w, h.size() * 2, hipMemcpyDeviceToHost));
    const int variant = direct_variant(ksz, stride, pad, H, W, Cin, Cout);
    const int cpl = variant ? direct_variant_cpl(variant) : 16;
    for (int co = 0; co < Cout; ++co) {
        const size_t r = (size_t)permuted_row(co, cpl) * K;
        if (!variant) { memcpy(&hp[r], &h[(size_t)co * K], (size_t)K * 2); continue; }
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < Cin; ++ci)
                hp[r + ((size_t)(ci >> 6) * 9 + tap) * 64 + (ci & 63)] = h[(size_t)co * K + (size_t)tap * Cin + ci];
    }
    void *d_wp = nullptr, *d_zero = nullptr;
    ALINK_HIP(hipMalloc(&d_wp, hp.size() * 2));
    ALINK_HIP(hipMalloc(&d_zero, 4096));
    ALINK_HIP(hipMemcpy(d_wp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
    ALINK_HIP(hipMemset(d_zero, 0, 4096));
    ConvParams p{};
    p.in = dev_in; p.wgt = d_wp; p.bias = dev_bias; p.alpha = dev_alpha; p.resid = dev_resid; p.out = dev_out;
    p.zero = d_zero; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.Ho = conv_out(H, ksz, stride, pad); p.Wo = conv_out(W, ksz, stride, pad);
    p.stride = stride; p.ksz = ksz; p.pad = pad; p.M = N * p.Ho * p.Wo; p.border_cls = border_cls;
    p.splitk = 1; p.ksteps_per_split = ksz * ksz * (Cin / 64);
    p.ablate = g_ablate; p.stamps = g_stamps;
    // fine: 1 / 0 force the 64- / 128-channel form of the linear-tile kernel, < 0 chooses as alink_embed does
    if (variant == 11 || variant == 12 || variant == 14) {
        const long long nwg128 = (((long long)p.M + 223) / 224) * (Cout / 128);
        p.fine = fine < 0 ? (nwg128 <= g_fine_max ? 1 : 0) : (fine ? 1 : 0);
    }
    hipError_t e = variant ? launch_conv3x3_direct(variant, dtype, p, st) : launch_conv_igemm(dtype, p, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_wp);
    (void)hipFree(d_zero);
    if (e != hipSuccess) return hip_fail(e, "launch_conv_igemm", __FILE__, __LINE__);
    if (e2 != hipSuccess) return hip_fail(e2, "hipStreamSynchronize", __FILE__, __LINE__);
    return ALINK_OK;
}


// Split-precision twin of alink_conv_nhwc (unit tests): float32 tensors in natural layouts on the device, converted
// to / from the f16-pair layouts on the HOST (synchronous, test use only).  Stored value = true value x 2^e.
int alink_conv_nhwc_x2(const float* dev_in, const float* dev_w, const float* dev_bias, const float* dev_alpha,
                       const float* dev_resid, float* dev_out, int N, int H, int W, int Cin, int Cout, int ksz, int stride,
                       int pad, int border_cls, int fine, int e_in, int e_w, int e_out, int e_res, void* stream) {
    ALINK_REQUIRE(dev_in && dev_w && dev_bias && dev_out, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(device_of_pointer(dev_in));
    ALINK_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, ALINK_EINVAL, "Cin/Cout must be multiples of 64");
    ALINK_REQUIRE(ksz == 1 || ksz == 3, ALINK_EINVAL, "ksz must be 1 or 3");
    ALINK_REQUIRE(!border_cls || (ksz == 3 && stride == 1 && pad == 1), ALINK_EINVAL,
                  "border classes need a 3x3 stride-1 pad-1 convolution");
    int rc = init_kernels();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ALINK_HIP(hipStreamSynchronize(st));
    const int K = ksz * ksz * Cin, Ho = conv_out(H, ksz, stride, pad), Wo = conv_out(W, ksz, stride, pad);
    const size_t Min = (size_t)N * H * W, M = (size_t)N * Ho * Wo;
    const int variant = linear_variant_x2(ksz, stride, pad, H, W, Cin, Cout);
    const int cpl = variant ? direct_variant_cpl(variant) : 16;
    auto pack_act = [&](const float* dev, size_t rows, int C, int e, std::vector<uint16_t>& q) -> int {
        std::vector<float> h(rows * C);
        ALINK_HIP(hipMemcpy(h.data(), dev, h.size() * 4, hipMemcpyDeviceToHost));
        q.assign(rows * C * 2, 0);
        for (size_t m = 0; m < rows; ++m)
            for (int c = 0; c < C; ++c) {
                const size_t at = m * 2 * C + (size_t)(c >> 6) * 128 + (c & 63);
                split16(std::ldexp((double)h[m * C + c], e), &q[at], &q[at + 64]);
            }
        return ALINK_OK;
    };
    std::vector<uint16_t> qin, qres, qw((size_t)Cout * K * 2);
    if ((rc = pack_act(dev_in, Min, Cin, e_in, qin))) return rc;
    if (dev_resid && (rc = pack_act(dev_resid, M, Cout, e_res, qres))) return rc;
    {
        std::vector<float> h((size_t)Cout * K);
        ALINK_HIP(hipMemcpy(h.data(), dev_w, h.size() * 4, hipMemcpyDeviceToHost));
        for (int co = 0; co < Cout; ++co) {
            const size_t r = (size_t)permuted_row(co, cpl) * 2 * K;
            for (int tap = 0; tap < ksz * ksz; ++tap)
                for (int ci = 0; ci < Cin; ++ci) {
                    const int cc = ci >> 6;
                    const size_t khi = variant ? (((size_t)cc * 2) * 9 + tap) * 64 + (ci & 63)
                                               : (((size_t)tap * (Cin >> 6) + cc) * 2) * 64 + (ci & 63);
                    split16(std::ldexp((double)h[(size_t)co * K + (size_t)tap * Cin + ci], e_w), &qw[r + khi],
                            &qw[r + khi + (variant ? 9 * 64 : 64)]);
                }
        }
    }
    void *d_in = nullptr, *d_res = nullptr, *d_w = nullptr, *d_out = nullptr, *d_zero = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_in); (void)hipFree(d_res); (void)hipFree(d_w); (void)hipFree(d_out); (void)hipFree(d_zero); };
#define X2_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return hip_fail(e__, #call, __FILE__, __LINE__); } } while (0)
    X2_TRY(hipMalloc(&d_in, qin.size() * 2));
    X2_TRY(hipMalloc(&d_w, qw.size() * 2));
    X2_TRY(hipMalloc(&d_out, M * Cout * 4));
    X2_TRY(hipMalloc(&d_zero, 4096));
    X2_TRY(hipMemcpy(d_in, qin.data(), qin.size() * 2, hipMemcpyHostToDevice));
    X2_TRY(hipMemcpy(d_w, qw.data(), qw.size() * 2, hipMemcpyHostToDevice));
    X2_TRY(hipMemset(d_zero, 0, 4096));
    if (dev_resid) {
        X2_TRY(hipMalloc(&d_res, qres.size() * 2));
        X2_TRY(hipMemcpy(d_res, qres.data(), qres.size() * 2, hipMemcpyHostToDevice));
    }
    ConvParams p{};
    p.in = d_in; p.wgt = d_w; p.bias = dev_bias; p.alpha = dev_alpha; p.resid = d_res; p.out = d_out;
    p.zero = d_zero; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo;
    p.stride = stride; p.ksz = ksz; p.pad = pad; p.M = (int)M; p.border_cls = border_cls;
    p.splitk = 1; p.ksteps_per_split = 3 * ksz * ksz * (Cin / 64);
    p.acc_scale = std::ldexp(1.f, e_out - e_in - e_w);
    p.bias_scale = std::ldexp(1.f, e_out);
    p.res_scale = std::ldexp(1.f, e_out - e_res);
    if (variant == 11 || variant == 12 || variant == 14) {
        const long long nwg128 = (((long long)p.M + 223) / 224) * (Cout / 128);
        p.fine = fine < 0 ? (nwg128 <= g_fine_max ? 1 : 0) : (fine ? 1 : 0);
    }
    X2_TRY(variant ? launch_conv3x3_direct(variant, ALINK_DT_F16X2, p, st) : launch_conv_igemm(ALINK_DT_F16X2, p, st));
    X2_TRY(hipStreamSynchronize(st));
    std::vector<uint16_t> qo(M * Cout * 2);
    X2_TRY(hipMemcpy(qo.data(), d_out, qo.size() * 2, hipMemcpyDeviceToHost));
    std::vector<float> ho(M * Cout);
    for (size_t m = 0; m < M; ++m)
        for (int c = 0; c < Cout; ++c) {
            const size_t at = m * 2 * Cout + (size_t)(c >> 6) * 128 + (c & 63);
            _Float16 hh, ll;
            memcpy(&hh, &qo[at], 2);
            memcpy(&ll, &qo[at + 64], 2);
            ho[m * Cout + c] = (float)std::ldexp((double)hh + (double)ll, -e_out);
        }
    X2_TRY(hipMemcpy(dev_out, ho.data(), ho.size() * 4, hipMemcpyHostToDevice));
#undef X2_TRY
    cleanup();
    return ALINK_OK;
}

}  // extern "C"
