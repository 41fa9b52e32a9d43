// backbone_f32.hip — the IR-ResNet feature extractor in FLOAT32, the arithmetic the reference itself runs
// (MXNet float32 forward: reference code/face_model.py:86-93), as a precision mode of the same handle
// (alink_ir_cfg.dtype = ALINK_DT_F32).
//
// Why it exists: BASELINE.json's north_star asks for "identical top-k active-learning selection indices".  A selection
// is a set of threshold decisions; with bf16 activations (1 - cos ~3e-4) a third of config 3's 1,024 most uncertain
// pairs differ from the f32 arithmetic, with f16 (4e-6) 5 % do.  Only the reference's own precision
// reproduces its decisions; this mode is that, at about a twelfth of the bf16 throughput (IR-100: 3.8 k embeddings/s =
// 92 TFLOP/s, 58 % of the f32 MFMA peak) — for audits, for settling
// pairs that sit on a cut, and as an on-device cross-check of the reduced-precision kernels at full depth.
//
// Every convolution and the FC layer is the exact-f32 MFMA GEMM of sgemm.hip (v_mfma_f32_32x32x2_f32: bit for bit an
// ordered fmaf chain) with its implicit-im2col gather, here with a kernel size (3 | 1) and a stride (1 | 2).  BatchNorm
// after a convolution is folded into its weights and bias (exact in real arithmetic, done in f64 on the host); a
// unit's pre-activation bn1 is NOT folded — it runs as an elementwise pass before conv1, so zero padding follows the
// normalisation exactly as in the symbol and no border classes are needed; PReLU and the residual add are the GEMM
// epilogue.  Layer order and tensor names: backbone.hip (insightface LResNet-E-IR).
#include "alink_common.h"
#include "sgemm.h"

#include <cmath>
#include <map>
#include <string>
#include <vector>

namespace alink {

namespace {

struct Affine { std::vector<double> a, b; };      // y = a x + b per channel

Affine bn_affine(const std::map<std::string, std::vector<float>>& raw, const std::string& n, bool fix_gamma, double eps) {
    const auto &g = raw.at(n + "_gamma"), &be = raw.at(n + "_beta"), &mu = raw.at(n + "_moving_mean"),
               &var = raw.at(n + "_moving_var");
    Affine r;
    r.a.resize(g.size());
    r.b.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        const double gamma = fix_gamma ? 1.0 : (double)g[i];
        r.a[i] = gamma / std::sqrt((double)var[i] + eps);
        r.b[i] = (double)be[i] - (double)mu[i] * r.a[i];
    }
    return r;
}

__global__ void affine_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ a,
                              const float* __restrict__ b, long long n4, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const f32x4 v = *(const f32x4*)(x + i * 4), av = *(const f32x4*)(a + c), bv = *(const f32x4*)(b + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaf(av[j], v[j], bv[j]);
    *(f32x4*)(y + i * 4) = o;
}

// pixels of any accepted layout -> NHWC float32
__global__ void to_nhwc_f32_kernel(const void* __restrict__ in, float* __restrict__ out, int layout, long long n,
                                   int H, int W) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // output index (img, y, x, c)
    if (i >= n) return;
    if (layout == ALINK_LAYOUT_NHWC_U8) { out[i] = (float)((const uint8_t*)in)[i]; return; }
    const int c = (int)(i % 3);
    const long long p = i / 3;
    const long long hw = (long long)H * W, img = p / hw, yx = p - img * hw;
    out[i] = ((const float*)in)[(img * 3 + c) * hw + yx];               // NCHW
}

}  // namespace

struct F32Layer {
    int kind;                 // 0: convolution (GEMM), 1: per-channel affine (bn1)
    int Cin, Cout, ks, stride, pad, Hin, Win, Hout, Wout;
    int in, out, resid;       // workspace buffer ids (resid -1: none)
    bool prescale;            // stem: (x - 127.5) * 0.0078125 on load
    float *d_w = nullptr, *d_bias = nullptr, *d_alpha = nullptr;     // conv: B [ks*ks*Cin][Cout], bias, PReLU slopes
    float *d_a = nullptr, *d_b = nullptr;                            // affine
};

struct F32Net {
    std::vector<F32Layer> layers;
    float *d_fc_w = nullptr, *d_fc_bias = nullptr;      // [K][emb] (k = pos * C + ch), folded bias
    int fcK = 0, emb = 0, H = 0, W = 0, last_buf = 0;
    int wide0 = 64;                                     // max(widths[0], widths[1]): channels of the widest full-resolution tensor
    std::vector<void*> allocs;
    std::string err;
    ~F32Net() {
        for (void* p : allocs) (void)hipFree(p);
    }
};

namespace {

bool up(F32Net* n, const std::vector<float>& h, float** d) {
    if (hipMalloc((void**)d, h.size() * sizeof(float)) != hipSuccess) return false;
    n->allocs.push_back(*d);
    return hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
}

// B[(ky*ks + kx)*Cin + ci][co] = post.a[co] * w[co][ci][ky][kx] (w: MXNet O,I,kh,kw), bias = post.b
bool conv_layer(F32Net* n, F32Layer& L, const std::vector<float>& w, const Affine& post, const std::vector<float>* prelu) {
    const int O = L.Cout, I = L.Cin, k = L.ks;
    std::vector<float> B((size_t)k * k * I * O), bias(O);
    for (int co = 0; co < O; ++co) {
        bias[co] = (float)post.b[co];
        for (int ci = 0; ci < I; ++ci)
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx)
                    B[((size_t)(ky * k + kx) * I + ci) * O + co] =
                        (float)(post.a[co] * (double)w[(((size_t)co * I + ci) * k + ky) * k + kx]);
    }
    if (!up(n, B, &L.d_w) || !up(n, bias, &L.d_bias)) return false;
    if (prelu && !up(n, *prelu, &L.d_alpha)) return false;
    return true;
}

int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

}  // namespace

F32Net* f32net_build(const std::map<std::string, std::vector<float>>& raw, const alink_ir_cfg& cfg) {
    F32Net* n = new F32Net();
    const double eps = cfg.bn_eps > 0.f ? (double)cfg.bn_eps : 2e-5;
    const int* w = cfg.widths;
    bool ok = true;
    int H = cfg.height, W = cfg.width;
    n->H = H;
    n->W = W;
    n->emb = cfg.emb;
    n->wide0 = cfg.widths[0] > cfg.widths[1] ? cfg.widths[0] : cfg.widths[1];
    {   // stem: buffer 0 <- conv0(pixels) + bn0 + PReLU; the pixel buffer is id 4
        F32Layer L{};
        L.kind = 0; L.Cin = 3; L.Cout = w[0]; L.ks = 3; L.stride = 1; L.pad = 1; L.Hin = H; L.Win = W; L.Hout = H; L.Wout = W;
        L.in = 4; L.out = 0; L.resid = -1; L.prescale = true;
        ok = ok && conv_layer(n, L, raw.at("conv0_weight"), bn_affine(raw, "bn0", false, eps), &raw.at("relu0_gamma"));
        n->layers.push_back(L);
    }
    int xb = 0;
    for (int s = 0; s < 4 && ok; ++s) {
        const int c = w[s + 1];
        for (int u = 0; u < cfg.units[s] && ok; ++u) {
            char pfx[64];
            snprintf(pfx, sizeof(pfx), "stage%d_unit%d", s + 1, u + 1);
            const std::string P(pfx);
            const int cin = u == 0 ? w[s] : c, stride = u == 0 ? 2 : 1;
            const int Ho = conv_out(H, 3, stride, 1), Wo = conv_out(W, 3, stride, 1);
            int ids[3], k = 0;
            for (int b = 0; b < 4; ++b) if (b != xb) ids[k++] = b;
            const int yb = ids[0], tb = ids[1], sb = ids[2];
            {   // y = bn1(x): elementwise, so that conv1's zero padding follows the normalisation as in the symbol
                F32Layer L{};
                L.kind = 1; L.Cin = L.Cout = cin; L.Hin = L.Hout = H; L.Win = L.Wout = W; L.in = xb; L.out = yb; L.resid = -1;
                const Affine a = bn_affine(raw, P + "_bn1", false, eps);
                std::vector<float> fa(a.a.begin(), a.a.end()), fb(a.b.begin(), a.b.end());
                ok = ok && up(n, fa, &L.d_a) && up(n, fb, &L.d_b);
                n->layers.push_back(L);
            }
            {   // t = PReLU(bn2(conv1(y)))
                F32Layer L{};
                L.kind = 0; L.Cin = cin; L.Cout = c; L.ks = 3; L.stride = 1; L.pad = 1; L.Hin = H; L.Win = W; L.Hout = H; L.Wout = W;
                L.in = yb; L.out = tb; L.resid = -1;
                ok = ok && conv_layer(n, L, raw.at(P + "_conv1_weight"), bn_affine(raw, P + "_bn2", false, eps),
                                      &raw.at(P + "_relu1_gamma"));
                n->layers.push_back(L);
            }
            int rb = xb;
            if (u == 0) {   // shortcut: bn(conv1x1 stride 2 (x))
                F32Layer L{};
                L.kind = 0; L.Cin = cin; L.Cout = c; L.ks = 1; L.stride = stride; L.pad = 0; L.Hin = H; L.Win = W; L.Hout = Ho; L.Wout = Wo;
                L.in = xb; L.out = sb; L.resid = -1;
                ok = ok && conv_layer(n, L, raw.at(P + "_conv1sc_weight"), bn_affine(raw, P + "_sc", false, eps), nullptr);
                n->layers.push_back(L);
                rb = sb;
            }
            {   // out = bn3(conv2(t)) + shortcut, into the buffer y no longer needs
                F32Layer L{};
                L.kind = 0; L.Cin = c; L.Cout = c; L.ks = 3; L.stride = stride; L.pad = 1; L.Hin = H; L.Win = W; L.Hout = Ho; L.Wout = Wo;
                L.in = tb; L.out = yb; L.resid = rb;
                ok = ok && conv_layer(n, L, raw.at(P + "_conv2_weight"), bn_affine(raw, P + "_bn3", false, eps), nullptr);
                n->layers.push_back(L);
            }
            xb = yb;
            H = Ho;
            W = Wo;
        }
    }
    if (ok) {   // bn1 folded into the FC (no padding there), fc1 BatchNorm (fix_gamma) folded too; flatten is (C,H,W) in the symbol
        const int C = w[4], E = cfg.emb, HW = H * W, K = C * HW;
        const auto &fw = raw.at("pre_fc1_weight"), &fb = raw.at("pre_fc1_bias");
        const Affine bnl = bn_affine(raw, "bn1", false, eps), bfc = bn_affine(raw, "fc1", true, eps);
        std::vector<float> B((size_t)K * E), bias(E);
        for (int o = 0; o < E; ++o) {
            double b = (double)fb[o];
            for (int ch = 0; ch < C; ++ch)
                for (int pos = 0; pos < HW; ++pos) {
                    const double wv = (double)fw[(size_t)o * K + (size_t)ch * HW + pos];
                    B[((size_t)pos * C + ch) * E + o] = (float)(bfc.a[o] * wv * bnl.a[ch]);
                    b += wv * bnl.b[ch];
                }
            bias[o] = (float)(bfc.a[o] * b + bfc.b[o]);
        }
        ok = up(n, B, &n->d_fc_w) && up(n, bias, &n->d_fc_bias);
        n->fcK = K;
        n->last_buf = xb;
    }
    if (!ok) {
        delete n;
        return nullptr;
    }
    return n;
}

void f32net_destroy(F32Net* n) { delete n; }

namespace {
constexpr size_t GEMM_WS_BYTES = (size_t)32 << 20;
size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct WsLayout { size_t buf[5], fc, gemm, total; };
WsLayout ws_layout(const F32Net* n, int N, const alink_ir_cfg&) {
    WsLayout L;
    size_t o = 0;
    // the widest full-resolution tensor: the stem output (widths[0]) or the first unit's conv1 output (widths[1])
    const size_t big = al256((size_t)N * n->H * n->W * (size_t)n->wide0 * 4);
    for (int i = 0; i < 4; ++i) { L.buf[i] = o; o += big; }
    L.buf[4] = o; o += al256((size_t)N * n->H * n->W * 3 * 4);
    L.fc = o; o += al256((size_t)N * n->emb * 4);
    L.gemm = o; o += GEMM_WS_BYTES;
    L.total = o;
    return L;
}
}  // namespace

size_t f32net_workspace_bytes(const F32Net* n, int N) {
    alink_ir_cfg dummy{};
    return ws_layout(n, N, dummy).total;
}

int f32net_embed(const F32Net* n, const void* dev_in, int layout, int N, float* dev_out, void* ws, size_t ws_bytes,
                 hipStream_t st) {
    ALINK_REQUIRE(n && dev_in && dev_out && ws && N > 0, ALINK_EINVAL, "bad argument");
    ALINK_REQUIRE((long long)N * n->H * n->W * 64 < (1ll << 31) && ((long long)N * n->H * n->W + 63) / 64 <= 65535, ALINK_EINVAL,
                  "batch of %d images too large for one float32 launch chain (grid rows); split it", N);
    alink_ir_cfg dummy{};
    const WsLayout WL = ws_layout(n, N, dummy);
    ALINK_REQUIRE(ws_bytes >= WL.total, ALINK_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, WL.total);
    ALINK_REQUIRE(((uintptr_t)ws & 255) == 0, ALINK_EINVAL, "workspace must be 256-byte aligned");
    char* base = (char*)ws;
    auto buf = [&](int id) -> float* { return (float*)(base + WL.buf[id]); };
    float* gws = (float*)(base + WL.gemm);
    const float* pixels = (const float*)dev_in;
    if (layout != ALINK_LAYOUT_NHWC_F32) {
        const long long cnt = (long long)N * n->H * n->W * 3;
        hipLaunchKernelGGL(to_nhwc_f32_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, dev_in, buf(4), layout, cnt,
                           n->H, n->W);
        pixels = buf(4);
    }
    for (const F32Layer& L : n->layers) {
        const float* in = L.in == 4 ? pixels : buf(L.in);
        if (L.kind == 1) {
            const long long n4 = (long long)N * L.Hin * L.Win * L.Cin / 4;
            hipLaunchKernelGGL(affine_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, in, buf(L.out), L.d_a, L.d_b, n4, L.Cin);
            continue;
        }
        GemmP g{};
        g.A = in; g.B = L.d_w; g.C = buf(L.out);
        g.M = N * L.Hout * L.Wout; g.N = L.Cout; g.K = L.ks * L.ks * L.Cin;
        g.lda = 0; g.ldb = L.Cout; g.ldc = L.Cout;
        g.amode = A_CONV; g.bmode = B_ROW;
        g.H = L.Hin; g.W = L.Win; g.Ci = L.Cin; g.Ho = L.Hout; g.Wo = L.Wout; g.pad = L.pad; g.ks = L.ks; g.cstride = L.stride;
        if (L.prescale) { g.prescale = 2; g.pre_sub = 127.5f; g.pre_mul = 0.0078125f; }
        g.bias = L.d_bias; g.alpha = L.d_alpha; g.resid = L.resid >= 0 ? buf(L.resid) : nullptr;
        gemm32_plan_split(g, 1);
        ALINK_HIP(launch_gemm32(g, gws, st));
    }
    {   // FC over the (H, W, C)-flattened final map, K split into f32 slabs reduced in slab order, then bias + L2 normalise
        GemmP g{};
        g.A = buf(n->last_buf); g.B = n->d_fc_w; g.C = (float*)(base + WL.fc);
        g.M = N; g.N = n->emb; g.K = n->fcK; g.lda = n->fcK; g.ldb = n->emb; g.ldc = n->emb;
        g.amode = A_ROW; g.bmode = B_ROW;
        // a FIXED split (16 slabs whatever the batch): an image's embedding must not depend on the batch it arrives in,
        // and the plan that fills the chip best would choose the split by the number of rows
        g.splitk = 16;
        g.kper = ((g.K + g.splitk - 1) / g.splitk + 15) / 16 * 16;
        g.splitk = (g.K + g.kper - 1) / g.kper;
        ALINK_REQUIRE(gemm32_workspace_floats(g) * 4 <= GEMM_WS_BYTES, ALINK_ENOMEM, "FC split workspace too small");
        ALINK_HIP(launch_gemm32(g, gws, st));
        FcFinishParams f{};
        f.slabs = (const float*)(base + WL.fc); f.bias = n->d_fc_bias; f.out = dev_out; f.norms = nullptr;
        f.S = 1; f.M = N; f.E = n->emb; f.scale = 1.f;
        ALINK_HIP(launch_fc_finish(f, st));
    }
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

}  // namespace alink
