// vgg16.hip — the VGGFace VGG-16 feature extractor behind siamese.FaceVGG16
// (reference code/siamese.py:187-200: keras_vggface VGGFace(model='vgg16', include_top=False), output
// of 'pool5' flattened -> 25088-d; preprocess = utils.preprocess_input(version=1)).
//
// keras-vggface 0.5 (reference requirements.txt:19) is not vendored; the graph restated here is its
// VGG16 (keras_vggface/models.py): thirteen 3x3 'same' convolutions WITH bias + ReLU in five blocks of
// widths 64, 128, 256, 512, 512 (2, 2, 3, 3, 3 layers), each block closed by MaxPooling2D((2,2), 2);
// Flatten of the 7 x 7 x 512 pool5 map is (h, w, c)-major.  preprocess_input(version=1): RGB -> BGR,
// subtract (93.5940, 104.7624, 129.1863) per BGR channel.
//
// conv1_1 (3 -> 64) is the IR backbone's K = 27 MFMA stem kernel with this network's normalisation in
// its loader; every other convolution runs on conv3x3_direct / conv_igemm (bias as the folded-BN
// bias slot, ReLU as a zero-slope PReLU epilogue).  New here: the 2x2 max-pool, whose last instance
// writes the float32 features.
#include "alink_common.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace alink {
namespace {

template <typename T> struct Vec8;
template <> struct Vec8<__bf16>   { typedef bf16x8 type; };
template <> struct Vec8<_Float16> { typedef f16x8 type; };

// MaxPooling2D((2,2), strides 2) on NHWC T (floor on odd sizes); OUT = T, or float for the last pool
template <typename T, typename OUT>
__global__ void maxpool2_kernel(const T* __restrict__ in, OUT* __restrict__ out, int N, int H, int W, int C) {
    typedef typename Vec8<T>::type vec8;
    const int Ho = H / 2, Wo = W / 2, c8n = C >> 3;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * Ho * Wo * c8n) return;
    const int c8 = (int)(i % c8n);
    long long t = i / c8n;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const T* base = in + (((size_t)n * H + 2 * oy) * W + 2 * ox) * C + c8 * 8;
    const vec8 a = *(const vec8*)base, b = *(const vec8*)(base + C);
    const vec8 c = *(const vec8*)(base + (size_t)W * C), d = *(const vec8*)(base + (size_t)W * C + C);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float m = max_keep_nan(max_keep_nan((float)a[j], (float)b[j]), max_keep_nan((float)c[j], (float)d[j]));
        out[(size_t)i * 8 + j] = (OUT)m;
    }
}

const int kBlocks[5] = {2, 2, 3, 3, 3};
const int kWidth[5] = {64, 128, 256, 512, 512};

struct VOp {
    int kind;          // 0 stem (conv1_1), 1 conv, 2 pool
    ConvParams cp;
    int variant = 0, in_buf = 0, out_buf = 0, H = 0, W = 0, C = 0;
    std::string name;
};

uint16_t cvt16(int dtype, float f) { return dtype == ALINK_DT_BF16 ? f32_to_bf16_rne(f) : f32_to_f16_rne(f); }

}  // namespace
}  // namespace alink

using namespace alink;

struct alink_vgg16 {
    int device = -1;
    int H, W, dtype;
    std::vector<std::pair<std::string, size_t>> expected;
    std::map<std::string, std::vector<float>> raw;
    bool finalized = false;
    std::vector<VOp> ops;
    void* d_stem_w = nullptr;
    float *d_stem_bias = nullptr, *d_zero_alpha = nullptr;
    void* d_zero = nullptr;
    int Hf = 0, Wf = 0;
    size_t buf_elems_per_image = 0;
    std::vector<void*> allocs;
    ~alink_vgg16() { for (void* p : allocs) (void)hipFree(p); }
};

namespace {

template <typename V>
int upload(alink_vgg16* r, const std::vector<V>& h, void** d) {
    ALINK_HIP(hipMalloc(d, h.size() * sizeof(V)));
    r->allocs.push_back(*d);
    ALINK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
    return ALINK_OK;
}

std::string lname(int b, int l) {
    char s[32];
    snprintf(s, sizeof(s), "conv%d_%d", b + 1, l + 1);
    return s;
}

}  // namespace

extern "C" {

alink_vgg16_t* alink_vgg16_create(int height, int width, int dtype) {
    if (dtype != ALINK_DT_BF16 && dtype != ALINK_DT_F16) { set_error("bad dtype"); return nullptr; }
    if (height < 32 || width < 32 || height > 512 || width > 512) { set_error("input %dx%d unsupported", height, width); return nullptr; }
    alink_vgg16* r = new alink_vgg16();
    r->device = current_device();
    r->H = height; r->W = width; r->dtype = dtype;
    int cin = 3;
    for (int b = 0; b < 5; ++b)
        for (int l = 0; l < kBlocks[b]; ++l) {
            r->expected.emplace_back(lname(b, l) + "/kernel", (size_t)9 * cin * kWidth[b]);
            r->expected.emplace_back(lname(b, l) + "/bias", (size_t)kWidth[b]);
            cin = kWidth[b];
        }
    r->Hf = height >> 5; r->Wf = width >> 5;
    return r;
}

void alink_vgg16_destroy(alink_vgg16_t* r) {
    if (!r) return;
    DeviceGuard dg(r->device);
    delete r;
}
int alink_vgg16_num_tensors(const alink_vgg16_t* r) { return r ? (int)r->expected.size() : 0; }
int alink_vgg16_tensor_info(const alink_vgg16_t* r, int i, const char** name, size_t* count) {
    ALINK_REQUIRE(r && i >= 0 && i < (int)r->expected.size(), ALINK_EINVAL, "tensor index out of range");
    if (name) *name = r->expected[i].first.c_str();
    if (count) *count = r->expected[i].second;
    return ALINK_OK;
}
int alink_vgg16_feature_size(const alink_vgg16_t* r) { return r ? r->Hf * r->Wf * 512 : 0; }

int alink_vgg16_load(alink_vgg16_t* r, const char* name, const float* host, size_t count) {
    ALINK_REQUIRE(r && name && host, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(!r->finalized, ALINK_ESTATE, "network already finalized");
    for (const auto& e : r->expected)
        if (e.first == name) {
            ALINK_REQUIRE(e.second == count, ALINK_EINVAL, "tensor %s: expected %zu elements, got %zu", name, e.second, count);
            r->raw[name].assign(host, host + count);
            return ALINK_OK;
        }
    set_error("tensor %s is not part of the VGGFace VGG-16", name);
    return ALINK_ENOTFOUND;
}

int alink_vgg16_finalize(alink_vgg16_t* r) {
    ALINK_REQUIRE(r && !r->finalized, ALINK_ESTATE, "bad state");
    DeviceGuard dg(r->device);
    for (const auto& e : r->expected)
        ALINK_REQUIRE(r->raw.count(e.first), ALINK_ESTATE, "tensor %s was never loaded", e.first.c_str());
    int rc = init_kernels();
    if (rc) return rc;
    ALINK_HIP(hipMalloc(&r->d_zero, 4096));
    r->allocs.push_back(r->d_zero);
    ALINK_HIP(hipMemset(r->d_zero, 0, 4096));
    ALINK_HIP(hipMalloc((void**)&r->d_zero_alpha, 512 * 4));
    r->allocs.push_back(r->d_zero_alpha);
    ALINK_HIP(hipMemset(r->d_zero_alpha, 0, 512 * 4));
    {   // conv1_1: [64'][64] T, k = ky*16 + kx*3 + c (StemParams::wgt) from the Keras kernel (3,3,3,64)
        const auto& w = r->raw.at("conv1_1/kernel");
        std::vector<uint16_t> wq((size_t)64 * 64, cvt16(r->dtype, 0.f));
        for (int co = 0; co < 64; ++co)
            for (int k = 0; k < 27; ++k)
                wq[(size_t)perm64_row_of_channel(co) * 64 + (k / 9) * 16 + k % 9] = cvt16(r->dtype, w[(size_t)k * 64 + co]);
        if ((rc = upload(r, wq, &r->d_stem_w))) return rc;
        if ((rc = upload(r, r->raw.at("conv1_1/bias"), (void**)&r->d_stem_bias))) return rc;
    }
    int H = r->H, W = r->W, cin = 64, cur = 0;
    r->buf_elems_per_image = (size_t)H * W * 64;
    VOp st; st.kind = 0; st.out_buf = 0; st.H = H; st.W = W; st.name = "conv1_1"; r->ops.push_back(st);
    for (int b = 0; b < 5; ++b) {
        for (int l = (b == 0 ? 1 : 0); l < kBlocks[b]; ++l) {
            const int cout = kWidth[b], K = 9 * cin;
            const auto& w = r->raw.at(lname(b, l) + "/kernel");       // (3, 3, cin, cout)
            VOp op; op.kind = 1; op.name = lname(b, l);
            op.variant = direct_variant_tiles(3, 1, 1, H, W, cin, cout);
            const int cpl = op.variant ? direct_variant_cpl(op.variant) : 16;
            std::vector<uint16_t> wq((size_t)cout * K);
            for (int co = 0; co < cout; ++co) {
                const size_t row = (size_t)permuted_row(co, cpl) * K;
                for (int tap = 0; tap < 9; ++tap)
                    for (int ci = 0; ci < cin; ++ci) {
                        const size_t kidx = op.variant ? ((size_t)(ci >> 6) * 9 + tap) * 64 + (ci & 63) : (size_t)tap * cin + ci;
                        wq[row + kidx] = cvt16(r->dtype, w[((size_t)tap * cin + ci) * cout + co]);
                    }
            }
            void* d_w = nullptr;
            float* d_b = nullptr;
            if ((rc = upload(r, wq, &d_w))) return rc;
            if ((rc = upload(r, r->raw.at(lname(b, l) + "/bias"), (void**)&d_b))) return rc;
            ConvParams& p = op.cp;
            memset(&p, 0, sizeof(p));
            p.wgt = d_w; p.bias = d_b; p.alpha = r->d_zero_alpha; p.zero = r->d_zero;
            p.H = H; p.W = W; p.Cin = cin; p.Cout = cout; p.Ho = H; p.Wo = W; p.stride = 1; p.ksz = 3; p.pad = 1;
            p.splitk = 1; p.ksteps_per_split = 9 * (cin / 64);
            op.in_buf = cur; op.out_buf = cur ^ 1; cur ^= 1;
            r->ops.push_back(op);
            r->buf_elems_per_image = std::max(r->buf_elems_per_image, (size_t)H * W * cout);
            cin = cout;
        }
        VOp mp; mp.kind = 2; mp.in_buf = cur; mp.out_buf = cur ^ 1; mp.H = H; mp.W = W; mp.C = cin; mp.name = "pool";
        cur ^= 1;
        r->ops.push_back(mp);
        H /= 2; W /= 2;
    }
    r->raw.clear();
    r->finalized = true;
    return ALINK_OK;
}

size_t alink_vgg16_workspace_bytes(const alink_vgg16_t* r, int n_images) {
    if (!r || !r->finalized || n_images <= 0) return 0;
    const size_t one = ((size_t)n_images * r->buf_elems_per_image * 2 + 255) & ~(size_t)255;
    return 2 * one;
}

int alink_vgg16_embed(alink_vgg16_t* r, const float* dev_in, int n, int preprocessed, float* dev_out, void* ws,
                      size_t ws_bytes, void* stream) {
    ALINK_REQUIRE(r && r->finalized, ALINK_ESTATE, "alink_vgg16_embed before finalize");
    DeviceGuard dg(r->device);
    ALINK_REQUIRE(dev_in && dev_out && ws && n > 0, ALINK_EINVAL, "bad argument");
    ALINK_REQUIRE(((uintptr_t)ws & 255) == 0, ALINK_EINVAL, "workspace must be 256-byte aligned");
    ALINK_REQUIRE(ws_bytes >= alink_vgg16_workspace_bytes(r, n), ALINK_ENOMEM, "workspace too small");
    ALINK_REQUIRE((long long)n * r->buf_elems_per_image < (1ll << 31), ALINK_EINVAL, "batch of %d too large; split it", n);
    hipStream_t st = (hipStream_t)stream;
    const size_t one = ((size_t)n * r->buf_elems_per_image * 2 + 255) & ~(size_t)255;
    auto buf = [&](int id) -> void* { return (char*)ws + one * id; };
    const size_t nops = r->ops.size();
    for (size_t i = 0; i < nops; ++i) {
        const VOp& op = r->ops[i];
        if (op.kind == 0) {
            StemParams sp{};
            sp.in = dev_in; sp.wgt = r->d_stem_w; sp.bias = r->d_stem_bias; sp.alpha = r->d_zero_alpha; sp.out = buf(0);
            sp.N = n; sp.H = r->H; sp.W = r->W; sp.C0 = 64; sp.layout = ALINK_LAYOUT_NHWC_F32; sp.mul = 1.f;
            if (preprocessed) { sp.sub[0] = sp.sub[1] = sp.sub[2] = 0.f; sp.flip = 0; }
            else { sp.sub[0] = 93.5940f; sp.sub[1] = 104.7624f; sp.sub[2] = 129.1863f; sp.flip = 1; }
            ALINK_HIP(launch_stem(r->dtype, sp, st));
        } else if (op.kind == 1) {
            ConvParams p = op.cp;
            p.in = buf(op.in_buf); p.out = buf(op.out_buf); p.N = n; p.M = n * p.Ho * p.Wo;
            if (op.variant) ALINK_HIP(launch_conv3x3_direct(op.variant, r->dtype, p, st));
            else            ALINK_HIP(launch_conv_igemm(r->dtype, p, st));
        } else {
            const bool last = i + 1 == nops;
            const long long tot = (long long)n * (op.H / 2) * (op.W / 2) * (op.C / 8);
            const dim3 grid((unsigned)((tot + 255) / 256));
            if (r->dtype == ALINK_DT_BF16) {
                if (last) hipLaunchKernelGGL((maxpool2_kernel<__bf16, float>), grid, dim3(256), 0, st, (const __bf16*)buf(op.in_buf), dev_out, n, op.H, op.W, op.C);
                else      hipLaunchKernelGGL((maxpool2_kernel<__bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)buf(op.in_buf), (__bf16*)buf(op.out_buf), n, op.H, op.W, op.C);
            } else {
                if (last) hipLaunchKernelGGL((maxpool2_kernel<_Float16, float>), grid, dim3(256), 0, st, (const _Float16*)buf(op.in_buf), dev_out, n, op.H, op.W, op.C);
                else      hipLaunchKernelGGL((maxpool2_kernel<_Float16, _Float16>), grid, dim3(256), 0, st, (const _Float16*)buf(op.in_buf), (_Float16*)buf(op.out_buf), n, op.H, op.W, op.C);
            }
        }
        ALINK_HIP(hipGetLastError());
    }
    return ALINK_OK;
}

}  // extern "C"
