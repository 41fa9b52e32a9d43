// resnet50.hip — the VGGFace2 ResNet-50 feature extractor behind siamese.RESNET50
// (reference code/siamese.py:203-216: keras_vggface VGGFace(model='resnet50', include_top=False),
// output of 'avg_pool' flattened -> 2048-d; preprocess = utils.preprocess_input(version=2)).
//
// keras-vggface 0.5 (reference requirements.txt:19) is not vendored; the graph restated here is its
// RESNET50 (keras_vggface/models.py):
//   conv1/7x7_s2 (64, 7x7, stride 2, 'same', no bias) -> BN -> ReLU -> MaxPool 3x3 stride 2 ('valid')
//   stages conv2..conv5 of bottleneck units [3, 4, 6, 3], widths (64,64,256) ... (512,512,2048):
//     1x1_reduce (stride s) -> BN -> ReLU -> 3x3 ('same') -> BN -> ReLU -> 1x1_increase -> BN,
//     shortcut = input, or 1x1_proj (stride s) -> BN on the first unit; add; ReLU
//     (s = 1 for conv2_1, 2 for conv3_1 / conv4_1 / conv5_1; the stride sits on the FIRST 1x1)
//   AveragePooling2D((7,7)) -> Flatten.      BN epsilon 1e-3 (Keras default), all convs bias-free.
//   preprocess_input(version=2): RGB -> BGR, subtract (91.4953, 103.8827, 131.0912) per BGR channel.
//
// All 1x1 and 3x3 convolutions run on the same MFMA kernels as the IR backbone (conv_igemm.hip,
// conv3x3_direct.hip) with BN folded into weights/bias, ReLU as a zero-slope PReLU epilogue and the
// unit's final ReLU as `post_relu`.  New here: the 7x7/2 stem (MFMA, K = 7 rows x 24 = 168 padded to
// 192, input band mean-subtracted into LDS with a true-zero frame for TensorFlow's asymmetric 'same'
// padding 2/3), the 3x3/2 max-pool and the 7x7 average pool.
#include "alink_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace alink {
namespace {

template <typename T> struct Vec8;
template <> struct Vec8<__bf16>   { typedef bf16x8 type; };
template <> struct Vec8<_Float16> { typedef f16x8 type; };
template <typename T>
__device__ __forceinline__ f32x4 mfma16(typename Vec8<T>::type a, typename Vec8<T>::type b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<__bf16>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<_Float16>(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

struct Stem7Params {
    const float* in;      // [N][H][W][3] f32
    const void*  wgt;     // [64'][192] T: k = ky*24 + kx*3 + c (c in the network's channel order), zero padded
    const float* bias;    // [64]
    void*        out;     // [N][Ho][Wo][64] T
    int N, H, W, Ho, Wo, pad_t, pad_l;
    float mean[3];        // subtracted from network channel c
    int flip;             // 1: input pixels are RGB and the network expects BGR (raw images);
                          // 0: input already in network order (preprocessed by the caller)
};

constexpr int S7_ROWS = 4;                      // output rows per workgroup
constexpr int S7_IN_ROWS = (S7_ROWS - 1) * 2 + 7;

template <typename T>
__global__ __launch_bounds__(256) void stem7_kernel(const Stem7Params p) {
    typedef typename Vec8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = (T*)smem_raw;
    const int H = p.H, W = p.W, Wo = p.Wo;
    const int PW = 2 * (((Wo + 15) >> 4) << 4) + 6;  // staged pixels per row: covers every tile column + 7 taps
    const int RP = (PW * 3 + 7) & ~7;               // LDS row pitch in elements (multiple of 8)
    const int n = blockIdx.y, oy0 = blockIdx.x * S7_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- stage the input band: LDS pixel j of row r <-> image (2*oy0 - pad_t + r, j - pad_l) -------
    const int total = S7_IN_ROWS * PW * 3;
    for (int i = tid; i < total; i += 256) {
        const int r = i / (PW * 3), e = i - r * (PW * 3);
        const int j = e / 3, ci = e - j * 3;                // ci = channel as stored in the input
        const int iy = 2 * oy0 - p.pad_t + r, ix = j - p.pad_l;
        const int c = p.flip ? 2 - ci : ci;                 // network channel
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = p.in[(((size_t)n * H + iy) * W + ix) * 3 + ci] - p.mean[c];
        tile[r * RP + j * 3 + c] = (T)v;
    }
    for (int i = tid; i < S7_IN_ROWS; i += 256)             // pitch padding: finite (zero-weight k's read it)
        for (int e = PW * 3; e < RP; ++e) tile[i * RP + e] = (T)0.f;
    if (tid < 32) tile[S7_IN_ROWS * RP + tid] = (T)0.f;     // slack after the last row

    const int q = lane >> 4, lr = lane & 15;
    // A fragments: K-step s, channel tile t: row 16t + lr, k = 32 s + 8 q ..
    vec8 wf[6][4];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[s][t] = *(const vec8*)((const T*)p.wgt + (16 * t + lr) * 192 + 32 * s + 8 * q);
    // B gather: k0 = 32 s + 8 q -> ky = k0 / 24 (7 -> no such row: weights are zero, read row 0), e0 = k0 % 24
    int koff[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k0 = 32 * s + 8 * q;
        const int ky = k0 / 24, e0 = k0 - 24 * ky;
        koff[s] = (ky < 7 ? ky : 0) * RP + e0;
    }
    const int cbase = 16 * q;
    float bi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bi[i] = p.bias[cbase + i];
    __syncthreads();

    const int tpr = (Wo + 15) >> 4;
    for (int tl = wave; tl < S7_ROWS * tpr; tl += 4) {
        const int ry = tl / tpr, xt = tl - ry * tpr;
        const int oy = oy0 + ry, ox = xt * 16 + lr;
        const T* base = tile + (2 * ry) * RP + 6 * ox;      // 2*ox pixels x 3 channels; even -> 4-byte aligned
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const unsigned int* src = (const unsigned int*)(base + koff[s]);
            union { unsigned int u[4]; vec8 v; } pf;
            pf.u[0] = src[0]; pf.u[1] = src[1]; pf.u[2] = src[2]; pf.u[3] = src[3];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16<T>(wf[s][t], pf.v, acc[t]);
        }
        if (oy < p.Ho && ox < Wo) {
            vec8 o0, o1;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * t + j;
                    const float v = relu_keep_nan(acc[t][j] + bi[i]);
                    if (i < 8) o0[i] = (T)v; else o1[i - 8] = (T)v;
                }
            T* o = (T*)p.out + (((size_t)n * p.Ho + oy) * Wo + ox) * 64 + cbase;
            *(vec8*)o = o0;
            *(vec8*)(o + 8) = o1;
        }
    }
}

// MaxPooling2D((3,3), strides 2, 'valid') on NHWC T; one thread = 8 channels of one output pixel
template <typename T>
__global__ void maxpool3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C,
                                  int Ho, int Wo) {
    typedef typename Vec8<T>::type vec8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8n = C >> 3;
    if (i >= (long long)N * Ho * Wo * c8n) return;
    const int c8 = (int)(i % c8n);
    long long t = i / c8n;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) {
            const vec8 v = *(const vec8*)(in + (((size_t)n * H + 2 * oy + dy) * W + 2 * ox + dx) * C + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = max_keep_nan(m[j], (float)v[j]);
        }
    vec8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)m[j];
    *(vec8*)(out + (size_t)i * 8) = o;
}

// AveragePooling2D over the whole HW x C map -> f32 [N][C]; one thread = 8 channels of one image
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int HW, int C, int* nonfinite) {
    typedef typename Vec8<T>::type vec8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c8n = C >> 3;
    if (i >= N * c8n) return;
    const int n = i / c8n, c8 = i - n * c8n;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int pos = 0; pos < HW; ++pos) {
        const vec8 v = *(const vec8*)(in + ((size_t)n * HW + pos) * C + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
    const float inv = 1.f / (float)HW;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        out[(size_t)n * C + c8 * 8 + j] = s[j] * inv;
        bad = bad || !(fabsf(s[j]) <= 3.4e38f);
    }
    if (bad && nonfinite) *nonfinite = 1;      // float16 storage left its range somewhere upstream
}

// ---- split precision (ALINK_DT_F16X2): every value an f16 pair hi + lo under a power-of-two scale per tensor; tensors
// [pixel][2 C] with each 64-channel chunk stored [hi 64 | lo 64] (conv_igemm.hip / conv3x3_linear.hip, SP forms) --------
// stem: (pixel - mean) x 2^6 split into two LDS images, weights [64'][hi 192 | lo 192], three MFMAs per tile and K-step
__global__ __launch_bounds__(256) void stem7_x2_kernel(const Stem7Params p, float acc_scale, float bias_scale) {
    typedef _Float16 T;
    typedef f16x8 vec8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int H = p.H, W = p.W, Wo = p.Wo;
    const int PW = 2 * (((Wo + 15) >> 4) << 4) + 6;
    const int RP = (PW * 3 + 7) & ~7;
    T* tile = (T*)smem_raw;
    T* tile_lo = tile + S7_IN_ROWS * RP + 32;
    const int n = blockIdx.y, oy0 = blockIdx.x * S7_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = S7_IN_ROWS * PW * 3;
    for (int i = tid; i < total; i += 256) {
        const int r = i / (PW * 3), e = i - r * (PW * 3);
        const int j = e / 3, ci = e - j * 3;
        const int iy = 2 * oy0 - p.pad_t + r, ix = j - p.pad_l;
        const int c = p.flip ? 2 - ci : ci;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = (p.in[(((size_t)n * H + iy) * W + ix) * 3 + ci] - p.mean[c]) * 64.f;
        const T hi = (T)v;
        tile[r * RP + j * 3 + c] = hi;
        tile_lo[r * RP + j * 3 + c] = (T)(v - (float)hi);
    }
    for (int i = tid; i < S7_IN_ROWS; i += 256)
        for (int e = PW * 3; e < RP; ++e) { tile[i * RP + e] = (T)0.f; tile_lo[i * RP + e] = (T)0.f; }
    if (tid < 32) { tile[S7_IN_ROWS * RP + tid] = (T)0.f; tile_lo[S7_IN_ROWS * RP + tid] = (T)0.f; }

    const int q = lane >> 4, lr = lane & 15;
    int koff[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k0 = 32 * s + 8 * q;
        const int ky = k0 / 24, e0 = k0 - 24 * ky;
        koff[s] = (ky < 7 ? ky : 0) * RP + e0;
    }
    const int cbase = 16 * q;
    float bi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bi[i] = p.bias[cbase + i] * bias_scale;
    __syncthreads();

    const T* wq = (const T*)p.wgt;
    const int tpr = (Wo + 15) >> 4;
    for (int tl = wave; tl < S7_ROWS * tpr; tl += 4) {
        const int ry = tl / tpr, xt = tl - ry * tpr;
        const int oy = oy0 + ry, ox = xt * 16 + lr;
        const int b0 = (2 * ry) * RP + 6 * ox;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            union { unsigned int u[4]; vec8 v; } ph, pl;
            const unsigned int* sh = (const unsigned int*)(tile + b0 + koff[s]);
            const unsigned int* sl = (const unsigned int*)(tile_lo + b0 + koff[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ph.u[j] = sh[j]; pl.u[j] = sl[j]; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const vec8 wh = *(const vec8*)(wq + (16 * t + lr) * 384 + 32 * s + 8 * q);
                const vec8 wl = *(const vec8*)(wq + (16 * t + lr) * 384 + 192 + 32 * s + 8 * q);
                acc[t] = mfma16<T>(wh, pl.v, acc[t]);
                acc[t] = mfma16<T>(wl, ph.v, acc[t]);
                acc[t] = mfma16<T>(wh, ph.v, acc[t]);
            }
        }
        if (oy < p.Ho && ox < Wo) {
            vec8 o0, o1, l0, l1;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * t + j;
                    const float v = relu_keep_nan(fmaf(acc[t][j], acc_scale, bi[i]));
                    const T hi = (T)v;
                    const T lo = (T)(v - (float)hi);
                    if (i < 8) { o0[i] = hi; l0[i] = lo; } else { o1[i - 8] = hi; l1[i - 8] = lo; }
                }
            T* o = (T*)p.out + (((size_t)n * p.Ho + oy) * Wo + ox) * 128 + cbase;
            *(vec8*)o = o0;
            *(vec8*)(o + 8) = o1;
            *(vec8*)(o + 64) = l0;
            *(vec8*)(o + 72) = l1;
        }
    }
}

// max pooling on f16 pairs: the maximum of hi + lo (formed in f32: at most one rounding), stored as a pair again; the scale is the input's
__global__ void maxpool3s2_x2_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, int N, int H, int W, int C,
                                     int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8n = C >> 3;
    if (i >= (long long)N * Ho * Wo * c8n) return;
    const int c8 = (int)(i % c8n);
    long long t = i / c8n;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int coff = ((c8 * 8) >> 6) * 128 + ((c8 * 8) & 63);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) {
            const _Float16* px = in + (((size_t)n * H + 2 * oy + dy) * W + 2 * ox + dx) * (2 * C) + coff;
            const f16x8 h = *(const f16x8*)px, l = *(const f16x8*)(px + 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = max_keep_nan(m[j], (float)h[j] + (float)l[j]);
        }
    f16x8 oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) { oh[j] = (_Float16)m[j]; ol[j] = (_Float16)(m[j] - (float)oh[j]); }
    _Float16* o = out + (((size_t)n * Ho + oy) * Wo + ox) * (2 * C) + coff;
    *(f16x8*)o = oh;
    *(f16x8*)(o + 64) = ol;
}

// average pooling of f16 pairs -> f32 [N][C] in true units (x scale); raises *nonfinite when a mean is not finite
__global__ void avgpool_x2_kernel(const _Float16* __restrict__ in, float* __restrict__ out, int N, int HW, int C, float scale,
                                  int* nonfinite) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c8n = C >> 3;
    if (i >= N * c8n) return;
    const int n = i / c8n, c8 = i - n * c8n;
    const int coff = ((c8 * 8) >> 6) * 128 + ((c8 * 8) & 63);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int pos = 0; pos < HW; ++pos) {
        const _Float16* px = in + ((size_t)n * HW + pos) * (2 * C) + coff;
        const f16x8 h = *(const f16x8*)px, l = *(const f16x8*)(px + 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)h[j] + (float)l[j];
    }
    const float inv = scale / (float)HW;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = s[j] * inv;
        bad = bad || !(fabsf(v) <= 3.4e38f);
        out[(size_t)n * C + c8 * 8 + j] = v;
    }
    if (bad && nonfinite) *nonfinite = 1;
}

static inline void split16(double x, uint16_t* hi, uint16_t* lo) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (double)h);
    memcpy(hi, &h, 2);
    memcpy(lo, &l, 2);
}
static inline int scale_exp(double maxabs) {       // maxabs * 2^e in [1024, 2048)
    if (!(maxabs > 0.0) || !std::isfinite(maxabs)) return 0;
    return 10 - std::ilogb(maxabs);
}

uint16_t cvt(int dtype, float f) { return dtype == ALINK_DT_BF16 ? f32_to_bf16_rne(f) : f32_to_f16_rne(f); }

struct Op {
    int kind;             // 0 stem7, 1 maxpool, 2 conv, 3 avgpool
    ConvParams cp;        // kind 2: everything but the per-call pointers/N/M
    int variant = 0;
    int in_buf = -1, out_buf = -1, resid_buf = -1;
    int e_w = 0, e_out = 0;   // split precision: stored value = true value x 2^e (weights: fixed at finalize; output: calibrated)
    std::string name;
};

}  // namespace
}  // namespace alink

using namespace alink;

struct alink_resnet50 {
    int device = -1;
    int H, W, dtype;
    float eps;
    std::vector<std::pair<std::string, size_t>> expected;
    std::map<std::string, std::vector<float>> raw;
    bool finalized = false;
    std::vector<Op> ops;
    void* d_stem_w = nullptr;
    float* d_stem_bias = nullptr;
    void* d_zero = nullptr;
    float* d_zero_alpha = nullptr;       // 2048 zeros: PReLU slope 0 == ReLU
    int Ho1, Wo1, Hp, Wp, Hf, Wf;
    size_t buf_elems_per_image = 0;
    // split precision
    bool calibrated = false;
    int stem_e_w = 0;
    unsigned* d_absmax = nullptr;
    int *h_flag = nullptr, *d_flag = nullptr;      // pinned word the average-pool kernel raises on a non-finite feature
    std::vector<void*> allocs;
    ~alink_resnet50() {
        for (void* p : allocs) (void)hipFree(p);
        if (h_flag) (void)hipHostFree(h_flag);
    }
};

namespace {

const int kUnits[4] = {3, 4, 6, 3};
const int kMid[4] = {64, 128, 256, 512};

int same_out(int x, int s) { return (x + s - 1) / s; }

void expect_conv(alink_resnet50* r, const std::string& n, int kh, int cin, int cout) {
    r->expected.emplace_back(n + "/kernel", (size_t)kh * kh * cin * cout);
    for (const char* s : {"/bn/gamma", "/bn/beta", "/bn/moving_mean", "/bn/moving_variance"})
        r->expected.emplace_back(n + s, (size_t)cout);
}

template <typename V>
int upload(alink_resnet50* r, const std::vector<V>& h, void** d) {
    ALINK_HIP(hipMalloc(d, h.size() * sizeof(V)));
    r->allocs.push_back(*d);
    ALINK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
    return ALINK_OK;
}

void bn_fold(const alink_resnet50* r, const std::string& n, std::vector<double>& a, std::vector<double>& b) {
    const auto& g = r->raw.at(n + "/bn/gamma");
    const auto& be = r->raw.at(n + "/bn/beta");
    const auto& mu = r->raw.at(n + "/bn/moving_mean");
    const auto& var = r->raw.at(n + "/bn/moving_variance");
    a.resize(g.size());
    b.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        a[i] = (double)g[i] / std::sqrt((double)var[i] + (double)r->eps);
        b[i] = (double)be[i] - (double)mu[i] * a[i];
    }
}

// Keras kernel (kh, kw, in, out) + following BN -> permuted T rows + f32 bias; appends a conv op
int add_conv(alink_resnet50* r, const std::string& name, int k, int stride, int cin, int cout, int Hin, int Win,
             int in_buf, int out_buf, int resid_buf, bool relu, bool post_relu) {
    const auto& w = r->raw.at(name + "/kernel");
    std::vector<double> a, b;
    bn_fold(r, name, a, b);
    const int pad = k == 3 ? 1 : 0;
    const int Ho = (Hin + 2 * pad - k) / stride + 1, Wo = (Win + 2 * pad - k) / stride + 1;
    Op op;
    op.kind = 2;
    op.name = name;
    const bool x2 = r->dtype == ALINK_DT_F16X2;
    op.variant = x2 ? linear_variant_x2(k, stride, pad, Hin, Win, cin, cout) : direct_variant_tiles(k, stride, pad, Hin, Win, cin, cout);
    const int cpl = op.variant ? direct_variant_cpl(op.variant) : 16;
    const int K = k * k * cin;
    std::vector<uint16_t> wq((size_t)cout * K * (x2 ? 2 : 1));
    if (x2) {
        double mx = 0.0;
        for (int co = 0; co < cout; ++co)
            for (size_t i = 0; i < (size_t)K; ++i) mx = std::max(mx, std::fabs(a[co] * (double)w[i * cout + co]));
        op.e_w = scale_exp(mx);
    }
    for (int co = 0; co < cout; ++co) {
        const size_t row = (size_t)permuted_row(co, cpl) * K * (x2 ? 2 : 1);
        for (int tap = 0; tap < k * k; ++tap)
            for (int ci = 0; ci < cin; ++ci) {
                const double v = a[co] * (double)w[((size_t)tap * cin + ci) * cout + co];
                if (x2) {   // linear kernel [chunk][hi | lo][tap][64], implicit GEMM [tap][chunk][hi 64 | lo 64]
                    const int cc = ci >> 6;
                    const size_t khi = op.variant ? (((size_t)cc * 2) * 9 + tap) * 64 + (ci & 63)
                                                  : (((size_t)tap * (cin >> 6) + cc) * 2) * 64 + (ci & 63);
                    split16(std::ldexp(v, op.e_w), &wq[row + khi], &wq[row + khi + (op.variant ? 9 * 64 : 64)]);
                    continue;
                }
                const size_t kidx = op.variant ? ((size_t)(ci >> 6) * 9 + tap) * 64 + (ci & 63) : (size_t)tap * cin + ci;
                wq[row + kidx] = cvt(r->dtype, (float)v);
            }
    }
    std::vector<float> bias(cout);
    for (int co = 0; co < cout; ++co) bias[co] = (float)b[co];
    void* d_w = nullptr;
    float* d_b = nullptr;
    int rc;
    if ((rc = upload(r, wq, &d_w))) return rc;
    if ((rc = upload(r, bias, (void**)&d_b))) return rc;
    ConvParams& p = op.cp;
    memset(&p, 0, sizeof(p));
    p.wgt = d_w; p.bias = d_b; p.alpha = relu ? r->d_zero_alpha : nullptr; p.zero = r->d_zero;
    p.H = Hin; p.W = Win; p.Cin = cin; p.Cout = cout; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.ksz = k; p.pad = pad;
    p.border_cls = 0; p.splitk = 1; p.ksteps_per_split = k * k * (cin / 64) * (x2 ? 3 : 1); p.post_relu = post_relu ? 1 : 0;
    op.in_buf = in_buf; op.out_buf = out_buf; op.resid_buf = resid_buf;
    r->ops.push_back(op);
    r->buf_elems_per_image = std::max(r->buf_elems_per_image, (size_t)Ho * Wo * cout);
    return ALINK_OK;
}

std::string unit_name(int stage, int block, const char* what) {
    char b[64];
    snprintf(b, sizeof(b), "conv%d_%d_%s", stage, block, what);
    return b;
}

}  // namespace

extern "C" {

alink_resnet50_t* alink_resnet50_create(int height, int width, int dtype, float bn_eps) {
    if (dtype != ALINK_DT_BF16 && dtype != ALINK_DT_F16 && dtype != ALINK_DT_F16X2) { set_error("bad dtype"); return nullptr; }
    if (height < 32 || width < 32) { set_error("input %dx%d too small", height, width); return nullptr; }
    alink_resnet50* r = new alink_resnet50();
    r->device = current_device();
    r->H = height; r->W = width; r->dtype = dtype; r->eps = bn_eps > 0.f ? bn_eps : 1e-3f;
    r->Ho1 = same_out(height, 2); r->Wo1 = same_out(width, 2);
    r->Hp = (r->Ho1 - 3) / 2 + 1; r->Wp = (r->Wo1 - 3) / 2 + 1;
    int H = r->Hp, W = r->Wp;
    for (int s = 1; s < 4; ++s) { H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1; }
    r->Hf = H; r->Wf = W;
    if (H != 7 || W != 7) {
        // AveragePooling2D((7,7)) then Flatten gives 2048 values only for a 7x7 map (224x224 input)
        set_error("input %dx%d leaves a %dx%d map before avg_pool; the 2048-d feature needs 7x7 (224x224 input)",
                  height, width, H, W);
        delete r;
        return nullptr;
    }
    expect_conv(r, "conv1/7x7_s2", 7, 3, 64);
    int cin = 64;
    for (int s = 0; s < 4; ++s) {
        const int mid = kMid[s], out = 4 * mid;
        for (int u = 1; u <= kUnits[s]; ++u) {
            expect_conv(r, unit_name(s + 2, u, "1x1_reduce"), 1, cin, mid);
            expect_conv(r, unit_name(s + 2, u, "3x3"), 3, mid, mid);
            expect_conv(r, unit_name(s + 2, u, "1x1_increase"), 1, mid, out);
            if (u == 1) expect_conv(r, unit_name(s + 2, u, "1x1_proj"), 1, cin, out);
            cin = out;
        }
    }
    return r;
}

void alink_resnet50_destroy(alink_resnet50_t* r) {
    if (!r) return;
    DeviceGuard dg(r->device);
    delete r;
}
int alink_resnet50_num_tensors(const alink_resnet50_t* r) { return r ? (int)r->expected.size() : 0; }
int alink_resnet50_tensor_info(const alink_resnet50_t* r, int i, const char** name, size_t* count) {
    ALINK_REQUIRE(r && i >= 0 && i < (int)r->expected.size(), ALINK_EINVAL, "tensor index out of range");
    if (name) *name = r->expected[i].first.c_str();
    if (count) *count = r->expected[i].second;
    return ALINK_OK;
}

int alink_resnet50_load(alink_resnet50_t* r, const char* name, const float* host, size_t count) {
    ALINK_REQUIRE(r && name && host, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(!r->finalized, ALINK_ESTATE, "network already finalized");
    for (const auto& e : r->expected)
        if (e.first == name) {
            ALINK_REQUIRE(e.second == count, ALINK_EINVAL, "tensor %s: expected %zu elements, got %zu", name, e.second, count);
            r->raw[name].assign(host, host + count);
            return ALINK_OK;
        }
    set_error("tensor %s is not part of the VGGFace2 ResNet-50", name);
    return ALINK_ENOTFOUND;
}

int alink_resnet50_finalize(alink_resnet50_t* r) {
    ALINK_REQUIRE(r && !r->finalized, ALINK_ESTATE, "bad state");
    DeviceGuard dg(r->device);
    for (const auto& e : r->expected)
        ALINK_REQUIRE(r->raw.count(e.first), ALINK_ESTATE, "tensor %s was never loaded", e.first.c_str());
    int rc = init_kernels();
    if (rc) return rc;
    ALINK_HIP(hipFuncSetAttribute((const void*)stem7_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    ALINK_HIP(hipFuncSetAttribute((const void*)stem7_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const bool x2 = r->dtype == ALINK_DT_F16X2;
    ALINK_HIP(hipFuncSetAttribute((const void*)stem7_x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    ALINK_HIP(hipHostMalloc((void**)&r->h_flag, 64, hipHostMallocMapped));
    *r->h_flag = 0;
    ALINK_HIP(hipHostGetDevicePointer((void**)&r->d_flag, r->h_flag, 0));
    if (x2) {
        ALINK_HIP(hipMalloc((void**)&r->d_absmax, 256));
        r->allocs.push_back(r->d_absmax);
    }
    ALINK_HIP(hipMalloc(&r->d_zero, 4096));
    r->allocs.push_back(r->d_zero);
    ALINK_HIP(hipMemset(r->d_zero, 0, 4096));
    ALINK_HIP(hipMalloc((void**)&r->d_zero_alpha, 2048 * 4));
    r->allocs.push_back(r->d_zero_alpha);
    ALINK_HIP(hipMemset(r->d_zero_alpha, 0, 2048 * 4));
    // ---- stem weights [64'][192], k = ky*24 + kx*3 + c
    {
        const auto& w = r->raw.at("conv1/7x7_s2/kernel");     // (7, 7, 3, 64)
        std::vector<double> a, b;
        bn_fold(r, "conv1/7x7_s2", a, b);
        std::vector<uint16_t> wq((size_t)64 * 192 * (x2 ? 2 : 1), x2 ? (uint16_t)0 : cvt(r->dtype, 0.f));
        if (x2) {
            double mx = 0.0;
            for (int co = 0; co < 64; ++co)
                for (int i = 0; i < 147; ++i) mx = std::max(mx, std::fabs(a[co] * (double)w[(size_t)i * 64 + co]));
            r->stem_e_w = scale_exp(mx);
        }
        for (int co = 0; co < 64; ++co)
            for (int ky = 0; ky < 7; ++ky)
                for (int kx = 0; kx < 7; ++kx)
                    for (int c = 0; c < 3; ++c) {
                        const double v = a[co] * (double)w[(((size_t)ky * 7 + kx) * 3 + c) * 64 + co];
                        const size_t kk = ky * 24 + kx * 3 + c;
                        if (x2) split16(std::ldexp(v, r->stem_e_w), &wq[(size_t)perm64_row_of_channel(co) * 384 + kk],
                                        &wq[(size_t)perm64_row_of_channel(co) * 384 + 192 + kk]);
                        else    wq[(size_t)perm64_row_of_channel(co) * 192 + kk] = cvt(r->dtype, (float)v);
                    }
        std::vector<float> bias(64);
        for (int co = 0; co < 64; ++co) bias[co] = (float)b[co];
        if ((rc = upload(r, wq, &r->d_stem_w))) return rc;
        if ((rc = upload(r, bias, (void**)&r->d_stem_bias))) return rc;
    }
    r->buf_elems_per_image = (size_t)r->Ho1 * r->Wo1 * 64;
    // ---- op list.  Buffers 0..3; stem -> 0, pool -> 1 (= x)
    Op st; st.kind = 0; st.out_buf = 0; st.name = "conv1/7x7_s2"; r->ops.push_back(st);
    Op mp; mp.kind = 1; mp.in_buf = 0; mp.out_buf = 1; mp.name = "max_pool"; r->ops.push_back(mp);
    int x = 1, H = r->Hp, W = r->Wp, cin = 64;
    for (int s = 0; s < 4; ++s) {
        const int mid = kMid[s], out = 4 * mid;
        for (int u = 1; u <= kUnits[s]; ++u) {
            const int stride = (u == 1 && s > 0) ? 2 : 1;
            int fr[3], nf = 0;
            for (int b = 0; b < 4; ++b) if (b != x) fr[nf++] = b;
            const int t1 = fr[0], t2 = fr[1], sb = fr[2];
            const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
            if ((rc = add_conv(r, unit_name(s + 2, u, "1x1_reduce"), 1, stride, cin, mid, H, W, x, t1, -1, true, false))) return rc;
            if ((rc = add_conv(r, unit_name(s + 2, u, "3x3"), 3, 1, mid, mid, Ho, Wo, t1, t2, -1, true, false))) return rc;
            int resid = x;
            if (u == 1) {
                if ((rc = add_conv(r, unit_name(s + 2, u, "1x1_proj"), 1, stride, cin, out, H, W, x, sb, -1, false, false))) return rc;
                resid = sb;
            }
            if ((rc = add_conv(r, unit_name(s + 2, u, "1x1_increase"), 1, 1, mid, out, Ho, Wo, t2, t1, resid, false, true))) return rc;
            x = t1; H = Ho; W = Wo; cin = out;
        }
    }
    Op ap; ap.kind = 3; ap.in_buf = x; ap.name = "avg_pool"; r->ops.push_back(ap);
    r->raw.clear();
    r->finalized = true;
    return ALINK_OK;
}

size_t alink_resnet50_workspace_bytes(const alink_resnet50_t* r, int n_images) {
    if (!r || !r->finalized || n_images <= 0) return 0;
    const size_t one = ((size_t)n_images * r->buf_elems_per_image * (r->dtype == ALINK_DT_F16X2 ? 4 : 2) + 255) & ~(size_t)255;
    return 4 * one;
}

// calib (split precision only): 0 = a forward; 1 = choose every tensor's scale exponent from this batch; 2 = the same, never
// above the exponents already held.  Synchronous when != 0.  (The scheme is csrc/backbone.hip's: embed_impl, settle.)
static int r50_run(alink_resnet50_t* r, const float* dev_in, int n, int preprocessed, float* dev_out, void* ws,
                   size_t ws_bytes, hipStream_t st, float* ms, double* flops, int* n_ops, int calib = 0) {
    ALINK_REQUIRE(r && r->finalized, ALINK_ESTATE, "alink_resnet50_embed before finalize");
    ALINK_REQUIRE(dev_in && dev_out && ws && n > 0, ALINK_EINVAL, "bad argument");
    ALINK_REQUIRE(((uintptr_t)ws & 255) == 0, ALINK_EINVAL, "workspace must be 256-byte aligned");
    ALINK_REQUIRE(ws_bytes >= alink_resnet50_workspace_bytes(r, n), ALINK_ENOMEM, "workspace too small");
    const bool x2 = r->dtype == ALINK_DT_F16X2;
    ALINK_REQUIRE((long long)n * r->buf_elems_per_image * (x2 ? 2 : 1) < (1ll << 31), ALINK_EINVAL, "batch of %d too large; split it", n);
    ALINK_REQUIRE(!calib || x2, ALINK_ESTATE, "only the split-precision mode is calibrated");
    ALINK_REQUIRE(!x2 || calib || r->calibrated, ALINK_ESTATE, "split-precision network: alink_resnet50_calibrate has not run");
    const size_t one = ((size_t)n * r->buf_elems_per_image * (x2 ? 4 : 2) + 255) & ~(size_t)255;
    int bexp[4] = {0, 0, 0, 0};
    auto settle = [&](int* e_io, const void* out, size_t n_elems, auto&& launch) -> int {
        int e = *e_io;
        for (int attempt = 0; attempt < 24; ++attempt) {
            const int rcl = launch(e);
            if (rcl) return rcl;
            if (!calib) break;
            unsigned bits = 0;
            ALINK_HIP(hipMemsetAsync(r->d_absmax, 0, 4, st));
            ALINK_HIP(launch_absmax_f16(out, n_elems, r->d_absmax, st));
            ALINK_HIP(hipMemcpyAsync(&bits, r->d_absmax, 4, hipMemcpyDeviceToHost, st));
            ALINK_HIP(hipStreamSynchronize(st));
            float m;
            memcpy(&m, &bits, 4);
            if (bits >= 0x7f800000u) { e -= 8; continue; }
            if (m == 0.f) break;
            int want = e + (10 - std::ilogb(m));
            if (calib == 2 && r->calibrated) want = std::min(want, *e_io);
            if (want == e) break;
            e = want;
        }
        *e_io = e;
        return ALINK_OK;
    };
    auto buf = [&](int id) -> void* { return (char*)ws + one * id; };
    const bool prof = ms != nullptr;
    std::vector<hipEvent_t> ev;
    auto mark = [&]() -> int {
        if (!prof) return ALINK_OK;
        hipEvent_t e;
        ALINK_HIP(hipEventCreate(&e));
        ALINK_HIP(hipEventRecord(e, st));
        ev.push_back(e);
        return ALINK_OK;
    };
    int rc, k = 0;
    if ((rc = mark())) return rc;
    for (Op& op : r->ops) {
        double fl = 0.0;
        if (op.kind == 0) {
            Stem7Params p{};
            p.in = dev_in; p.wgt = r->d_stem_w; p.bias = r->d_stem_bias; p.out = buf(op.out_buf);
            p.N = n; p.H = r->H; p.W = r->W; p.Ho = r->Ho1; p.Wo = r->Wo1;
            // TensorFlow 'same': total = (Ho-1)*2 + 7 - H, before = total / 2
            const int th = std::max((r->Ho1 - 1) * 2 + 7 - r->H, 0), tw = std::max((r->Wo1 - 1) * 2 + 7 - r->W, 0);
            p.pad_t = th / 2; p.pad_l = tw / 2;
            if (preprocessed) { p.mean[0] = p.mean[1] = p.mean[2] = 0.f; p.flip = 0; }
            else { p.mean[0] = 91.4953f; p.mean[1] = 103.8827f; p.mean[2] = 131.0912f; p.flip = 1; }
            const int PW = 2 * (((p.Wo + 15) >> 4) << 4) + 6, RP = (PW * 3 + 7) & ~7;
            const size_t lds = ((size_t)S7_IN_ROWS * RP + 32) * 2 * (x2 ? 2 : 1);
            ALINK_REQUIRE(lds <= 64 * 1024, ALINK_EINVAL, "image too wide for the stem band (%zu B of LDS)", lds);
            dim3 grid((p.Ho + S7_ROWS - 1) / S7_ROWS, n);
            if (x2) {
                rc = settle(&op.e_out, p.out, (size_t)n * p.Ho * p.Wo * 128, [&](int e) -> int {
                    // the loader stores (pixel - mean) x 2^6: |.| < 2^14
                    hipLaunchKernelGGL(stem7_x2_kernel, grid, dim3(256), lds, st, p, std::ldexp(1.f, e - 6 - r->stem_e_w), std::ldexp(1.f, e));
                    ALINK_HIP(hipGetLastError());
                    return ALINK_OK;
                });
                if (rc) return rc;
                bexp[op.out_buf] = op.e_out;
            } else
            if (r->dtype == ALINK_DT_BF16) hipLaunchKernelGGL(stem7_kernel<__bf16>, grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL(stem7_kernel<_Float16>, grid, dim3(256), lds, st, p);
            fl = 2.0 * n * p.Ho * p.Wo * 64.0 * 147.0;
        } else if (op.kind == 1) {
            const long long tot = (long long)n * r->Hp * r->Wp * 8;
            const dim3 grid((unsigned)((tot + 255) / 256));
            if (x2) {
                hipLaunchKernelGGL(maxpool3s2_x2_kernel, grid, dim3(256), 0, st, (const _Float16*)buf(op.in_buf),
                                   (_Float16*)buf(op.out_buf), n, r->Ho1, r->Wo1, 64, r->Hp, r->Wp);
                bexp[op.out_buf] = bexp[op.in_buf];
            } else
            if (r->dtype == ALINK_DT_BF16)
                hipLaunchKernelGGL(maxpool3s2_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)buf(op.in_buf),
                                   (__bf16*)buf(op.out_buf), n, r->Ho1, r->Wo1, 64, r->Hp, r->Wp);
            else
                hipLaunchKernelGGL(maxpool3s2_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)buf(op.in_buf),
                                   (_Float16*)buf(op.out_buf), n, r->Ho1, r->Wo1, 64, r->Hp, r->Wp);
        } else if (op.kind == 2) {
            ConvParams p = op.cp;
            p.in = buf(op.in_buf); p.out = buf(op.out_buf); p.resid = op.resid_buf >= 0 ? buf(op.resid_buf) : nullptr;
            p.N = n; p.M = n * p.Ho * p.Wo;
            if (x2) {
                rc = settle(&op.e_out, p.out, (size_t)p.M * p.Cout * 2, [&](int e) -> int {
                    p.acc_scale = std::ldexp(1.f, e - bexp[op.in_buf] - op.e_w);
                    p.bias_scale = std::ldexp(1.f, e);
                    p.res_scale = op.resid_buf >= 0 ? std::ldexp(1.f, e - bexp[op.resid_buf]) : 1.f;
                    if (op.variant) ALINK_HIP(launch_conv3x3_direct(op.variant, r->dtype, p, st));
                    else            ALINK_HIP(launch_conv_igemm(r->dtype, p, st));
                    return ALINK_OK;
                });
                if (rc) return rc;
                bexp[op.out_buf] = op.e_out;
            } else
            if (op.variant) ALINK_HIP(launch_conv3x3_direct(op.variant, r->dtype, p, st));
            else            ALINK_HIP(launch_conv_igemm(r->dtype, p, st));
            fl = conv_flops(p);
        } else {
            const int tot = n * 2048 / 8;
            if (x2)
                hipLaunchKernelGGL(avgpool_x2_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, (const _Float16*)buf(op.in_buf),
                                   dev_out, n, r->Hf * r->Wf, 2048, std::ldexp(1.f, -bexp[op.in_buf]), r->d_flag);
            else
            if (r->dtype == ALINK_DT_BF16)
                hipLaunchKernelGGL(avgpool_kernel<__bf16>, dim3((tot + 255) / 256), dim3(256), 0, st,
                                   (const __bf16*)buf(op.in_buf), dev_out, n, r->Hf * r->Wf, 2048, r->d_flag);
            else
                hipLaunchKernelGGL(avgpool_kernel<_Float16>, dim3((tot + 255) / 256), dim3(256), 0, st,
                                   (const _Float16*)buf(op.in_buf), dev_out, n, r->Hf * r->Wf, 2048, r->d_flag);
        }
        ALINK_HIP(hipGetLastError());
        if (prof && k < *n_ops) flops[k] = fl;
        ++k;
        if ((rc = mark())) return rc;
    }
    if (prof) {
        ALINK_HIP(hipStreamSynchronize(st));
        for (int i = 0; i + 1 < (int)ev.size() && i < *n_ops; ++i) {
            float t = 0.f;
            ALINK_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
            ms[i] = t;
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        *n_ops = k;
    }
    if (calib) {
        ALINK_HIP(hipStreamSynchronize(st));
        ALINK_REQUIRE(*r->h_flag == 0, ALINK_EINVAL, "calibration batch produced non-finite features");
        r->calibrated = true;
    }
    return ALINK_OK;
}

int alink_resnet50_calibrate(alink_resnet50_t* r, const float* dev_in, int n_images, int preprocessed, void* dev_workspace,
                             size_t workspace_bytes, int merge, void* stream) {
    ALINK_REQUIRE(r && r->finalized, ALINK_ESTATE, "alink_resnet50_calibrate before finalize");
    ALINK_REQUIRE(r->dtype == ALINK_DT_F16X2, ALINK_ESTATE, "only the split-precision mode (ALINK_DT_F16X2) is calibrated");
    ALINK_REQUIRE(dev_in && dev_workspace && n_images > 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(r->device);
    hipStream_t st = (hipStream_t)stream;
    ALINK_HIP(hipStreamSynchronize(st));
    const int pending = *(volatile int*)r->h_flag;          // a report not read yet survives the calibration run
    struct Keep { int* f; int v; ~Keep() { if (v) *(volatile int*)f = 1; } } keep{r->h_flag, pending};
    *r->h_flag = 0;
    float* scratch = nullptr;
    ALINK_HIP(hipMalloc((void**)&scratch, (size_t)n_images * 2048 * sizeof(float)));
    const int rc = r50_run(r, dev_in, n_images, preprocessed, scratch, dev_workspace, workspace_bytes, st, nullptr, nullptr, nullptr,
                           merge ? 2 : 1);
    (void)hipFree(scratch);
    return rc;
}

int alink_resnet50_num_scales(const alink_resnet50_t* r) {
    if (!r || !r->finalized || r->dtype != ALINK_DT_F16X2) return 0;
    return (int)r->ops.size();
}

int alink_resnet50_get_scales(const alink_resnet50_t* r, int* exponents, int n) {
    ALINK_REQUIRE(r && r->finalized, ALINK_ESTATE, "alink_resnet50_get_scales before finalize");
    ALINK_REQUIRE(r->dtype == ALINK_DT_F16X2, ALINK_ESTATE, "only the split-precision mode (ALINK_DT_F16X2) has scales");
    ALINK_REQUIRE(r->calibrated, ALINK_ESTATE, "alink_resnet50_get_scales before alink_resnet50_calibrate / set_scales");
    ALINK_REQUIRE(exponents && n == (int)r->ops.size(), ALINK_EINVAL, "expected room for %d exponents, got %d", (int)r->ops.size(), n);
    for (size_t i = 0; i < r->ops.size(); ++i) exponents[i] = r->ops[i].e_out;
    return ALINK_OK;
}

int alink_resnet50_set_scales(alink_resnet50_t* r, const int* exponents, int n) {
    ALINK_REQUIRE(r && r->finalized, ALINK_ESTATE, "alink_resnet50_set_scales before finalize");
    ALINK_REQUIRE(r->dtype == ALINK_DT_F16X2, ALINK_ESTATE, "only the split-precision mode (ALINK_DT_F16X2) has scales");
    ALINK_REQUIRE(exponents && n == (int)r->ops.size(), ALINK_EINVAL, "expected %d exponents, got %d", (int)r->ops.size(), n);
    for (int i = 0; i < n; ++i)
        ALINK_REQUIRE(exponents[i] >= -126 && exponents[i] <= 126, ALINK_EINVAL, "exponent %d of op %d is not a float32 power of two", exponents[i], i);
    for (size_t i = 0; i < r->ops.size(); ++i) r->ops[i].e_out = exponents[i];
    r->calibrated = true;
    return ALINK_OK;
}

int alink_resnet50_range_flag(alink_resnet50_t* r, int reset) {
    ALINK_REQUIRE(r && r->finalized, ALINK_ESTATE, "alink_resnet50_range_flag before finalize");
    const int v = *(volatile int*)r->h_flag != 0 ? 1 : 0;
    if (reset) *(volatile int*)r->h_flag = 0;
    return v;
}

int alink_resnet50_embed(alink_resnet50_t* r, const float* dev_in, int n_images, int preprocessed, float* dev_out,
                         void* dev_workspace, size_t workspace_bytes, void* stream) {
    ALINK_REQUIRE(r, ALINK_EINVAL, "NULL network");
    DeviceGuard dg(r->device);
    return r50_run(r, dev_in, n_images, preprocessed, dev_out, dev_workspace, workspace_bytes, (hipStream_t)stream,
                   nullptr, nullptr, nullptr);
}

int alink_resnet50_profile(alink_resnet50_t* r, const float* dev_in, int n_images, float* dev_out, void* dev_workspace,
                           size_t workspace_bytes, void* stream, float* ms, double* flops, int* n_ops) {
    ALINK_REQUIRE(ms && flops && n_ops && *n_ops > 0, ALINK_EINVAL, "NULL profile buffers");
    ALINK_REQUIRE(r, ALINK_EINVAL, "NULL network");
    DeviceGuard dg(r->device);
    return r50_run(r, dev_in, n_images, 0, dev_out, dev_workspace, workspace_bytes, (hipStream_t)stream, ms, flops, n_ops);
}

const char* alink_resnet50_op_name(const alink_resnet50_t* r, int i) {
    return (r && i >= 0 && i < (int)r->ops.size()) ? r->ops[i].name.c_str() : "";
}

}  // extern "C"
