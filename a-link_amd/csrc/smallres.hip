// smallres.hip — SmallRes: the low-resolution siamese CNN that the Multi-PIE driver trains end to end.
//
// Replaces the Keras graph at reference code/siamese.py:134-170 and the calls made on it
// (predict code/siamese.py:183-184; train_on_batch / test_on_batch via customTrainModel
// code/siamese.py:81-112, called from code/ALINK_MTP.py:121; finetune code/siamese.py:52-58).
// Keras semantics restated: Conv2D default padding 'valid' unless 'same'; MaxPooling2D 2x2 stride 2
// (floor); Dropout(rate) in training multiplies kept units by 1/(1-rate); Flatten of NHWC is
// (h, w, c)-major; Dense kernel (in,out).
//
// Every convolution and the wide Dense layer — forward, input-gradient and weight-gradient — is an
// exact-f32 GEMM on the f32-input matrix cores (sgemm.hip: implicit im2col gathers, flipped-kernel
// gather for the input gradient, pixel-reduction with a ones row for weight + bias gradients,
// deterministic split-K); pooling / dropout / ReLU masks are small elementwise kernels.
// The pair head (|l-r| -> 128 -> 32 -> 2) is an alink_head handle (head.hip); this file adds the
// tower forward/backward and ties the two Adadelta states together.
#include "alink_common.h"
#include "sgemm.h"
#include "philox.h"

#include <cstring>
#include <vector>

using namespace alink;

extern "C" {
int alink_head_train_step_input_grads(alink_head_t* h, const float* dev_L, const float* dev_R, const float* dev_y,
                                      const float* dev_sw, int n, float grad_scale, int relu_inputs, float* dev_dL,
                                      float* dev_dR, float* dev_colsum, float* dev_metrics, void* stream);
float* alink_head_params_dev(alink_head_t* h);
int alink_keep_masks(uint8_t* dev_out, int64_t count, float keep, uint64_t seed, void* stream);
int alink_head_apply_update_with(alink_head_t* h, float* dev_params2, const float* dev_grads2, float* dev_acc2, float* dev_dacc2,
                                 size_t n2, void* stream);
}

namespace {

constexpr int MAXN = 256;           // pairs per call (host code chunks larger requests)

inline dim3 g1(long long n) { return dim3((unsigned)((n + 255) / 256), 1, 1); }

// 2x2/2 max pool (+ optional dropout: keep-mask u8, scale 1/(1-rate)); records the argmax (0..3)
__global__ void pool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, uint8_t* __restrict__ arg,
                                const uint8_t* __restrict__ mask, float scale, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * Ho * Wo * C) return;
    const int c = (int)(i % C);
    long long r = i / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float best = -INFINITY;
    int bi = 0;
    for (int k = 0; k < 4; ++k) {
        const float v = in[(((size_t)n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * C + c];
        if (v > best) { best = v; bi = k; }
    }
    if (arg) arg[i] = (uint8_t)bi;
    if (mask) best = mask[i] ? best * scale : 0.f;
    out[i] = best;
}

// din (pre-pool, post-relu tensor `act`) = relu'(act) * [argmax] * dropout * dpooled
__global__ void pool_bwd_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ arg,
                                const uint8_t* __restrict__ mask, float scale, const float* __restrict__ act,
                                float* __restrict__ dact, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W * C) return;
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    float g = 0.f;
    const int oy = y >> 1, ox = x >> 1;
    if (oy < Ho && ox < Wo) {
        const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + c;
        if (arg[o] == ((y & 1) * 2 + (x & 1))) {
            g = dpool[o];
            if (mask) g = mask[o] ? g * scale : 0.f;
        }
    }
    dact[i] = act[i] > 0.f ? g : 0.f;          // relu of the conv that produced `act`
}

__global__ void adadelta2_kernel(float* __restrict__ prm, const float* __restrict__ g, float* __restrict__ a,
                                 float* __restrict__ d, size_t n, float lr, float rho, float eps) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float na = rho * a[i] + (1.f - rho) * gi * gi;
    const float u = gi * sqrtf(d[i] + eps) / sqrtf(na + eps);
    prm[i] = prm[i] - lr * u;
    d[i] = rho * d[i] + (1.f - rho) * u * u;
    a[i] = na;
}

// ---- the first layer (3 -> 32 channels, 3x3 'same') on the vector units ---------------------------------------------------
// K = 27 is not a GEMM's shape: as one the layer was a single half-empty MFMA stage fed by 32 four-byte gathers per thread
// (12.6 us at 2 x 16 images), its weight gradient a 28-row tile reduced over 32,768 pixels by scalar gathers (26 + 6 us, the
// LAST link of the backward chain).  Here a thread owns a pixel: 27 inputs, 32 accumulators, the weights broadcast from LDS,
// k ascending like the GEMM's reduction (the matrix instruction is an ordered fmaf chain too: sgemm.hip).
struct Conv1P {
    const float *L, *R;       // images 0 .. split-1 from L, the rest from R (R == nullptr: all from L); [n][H][W][3]
    const float *w, *b;       // (3,3,3,32) = [27][32], [32]
    float* out;               // [nb][H][W][32]
    int nb, split, H, W, prescale;
};
// this pixel's 27 inputs (3 x 3 taps x 3 channels, zero outside the image): every load is issued from a clamped, always valid
// address and its value kept or dropped by an AND with a mask the compiler cannot see through — `ok ? load : 0` becomes a branch
// around the load, and 27 of those are 27 trips to memory one after the other (the lesson of sgemm.hip's loaders)
__device__ __forceinline__ void conv1_patch(const Conv1P& p, int pix, float (&in)[27]) {
    const int x = pix % p.W, r = pix / p.W, y = r % p.H, n = r / p.H;
    const float* img = (p.R && n >= p.split) ? p.R + (size_t)(n - p.split) * p.H * p.W * 3 : p.L + (size_t)n * p.H * p.W * 3;
    const float sub = p.prescale ? 128.f : 0.f, mul = p.prescale ? 0.0078125f : 1.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int off = ok ? (iy * p.W + ix) * 3 : 0;
        int mk = ok ? -1 : 0;
        asm volatile("" : "+v"(mk));
#pragma unroll
        for (int c = 0; c < 3; ++c) in[t * 3 + c] = __int_as_float(__float_as_int((img[off + c] - sub) * mul) & mk);
    }
}

// thread = (pixel, 8 output channels): 131,072 threads at 2 x 16 images of 32 x 32 (a thread per pixel left half the chip idle)
// Workgroups beyond the layer's own (`conv_blocks`) draw the step's Dropout keep-masks (mk: alink_keep_masks' bytes, philox.h) —
// the masks are first read two launches later, and a launch of their own was 2.7 us + a launch gap at the head of every step.
struct MaskDraw { unsigned char* out; long long count; float keep; unsigned long long seed; int conv_blocks; };
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const Conv1P p, const MaskDraw mk) {
    if (mk.out && (int)blockIdx.x >= mk.conv_blocks) {
        keep_mask_block(mk.out, mk.count, mk.keep, mk.seed, 0ull, (unsigned long long)(blockIdx.x - mk.conv_blocks) * 256 + threadIdx.x);
        return;
    }
    __shared__ __attribute__((aligned(16))) float ws[27 * 32 + 32];
    for (int i = threadIdx.x; i < 27 * 32 + 32; i += 256) ws[i] = i < 27 * 32 ? p.w[i] : p.b[i - 27 * 32];
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x, pix = g >> 2, c0 = (g & 3) * 8;
    if (pix >= p.nb * p.H * p.W) return;
    float in[27];
    conv1_patch(p, pix, in);
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const f32x4 wa = *(const f32x4*)(ws + k * 32 + c0), wb = *(const f32x4*)(ws + k * 32 + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = fmaf(in[k], wa[j], acc[j]); acc[4 + j] = fmaf(in[k], wb[j], acc[4 + j]); }
    }
    float* o = p.out + (size_t)pix * 32 + c0;
    f32x4 v0, v1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(acc[j] + ws[27 * 32 + c0 + j], 0.f); v1[j] = fmaxf(acc[4 + j] + ws[27 * 32 + c0 + 4 + j], 0.f); }
    *(f32x4*)o = v0;
    *(f32x4*)(o + 4) = v1;
}

// partial weight + bias gradients of the first layer over 128 pixels per workgroup: part[block][28][32] (row 27 = bias);
// summed over the blocks, in a fixed order, by sgemm's slab reduction
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const Conv1P p, const float* __restrict__ dz, float* __restrict__ part) {
    __shared__ float xs[128 * 29];                     // [pixel][27 inputs + the ones column], pitch 29 (odd: conflict-free rows)
    __shared__ __attribute__((aligned(16))) float ds[128 * 32];
    const int tid = threadIdx.x, p0 = blockIdx.x * 128, P = p.nb * p.H * p.W;
    if (tid < 128) {                                   // half the threads fetch a pixel's patch each, the other half the dz rows
        const int pix = p0 + tid;
        float in[27];
        conv1_patch(p, pix < P ? pix : 0, in);
#pragma unroll
        for (int k = 0; k < 27; ++k) xs[tid * 29 + k] = pix < P ? in[k] : 0.f;
        xs[tid * 29 + 27] = pix < P ? 1.f : 0.f;       // the ones column: the bias gradient
    } else {
        for (int i = tid - 128; i < 128 * 8; i += 128) {
            const int lp = i >> 3, c4 = (i & 7) * 4;
            const bool ok = p0 + lp < P;
            f32x4 v = *(const f32x4*)(dz + (size_t)(ok ? p0 + lp : 0) * 32 + c4);
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(ds + lp * 32 + c4) = v;
        }
    }
    __syncthreads();
    const int co = tid & 31, kb = tid >> 5;                                 // outputs (kb + 8 j, co), j < 4 (k < 28)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int lp = 0; lp < 128; ++lp) {
        const float d = ds[lp * 32 + co];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(xs[lp * 29 + ((kb + 8 * j) < 28 ? kb + 8 * j : 27)], (kb + 8 * j) < 28 ? d : 0.f, acc[j]);
    }
    float* o = part + (size_t)blockIdx.x * (28 * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (kb + 8 * j < 28) o[(kb + 8 * j) * 32 + co] = acc[j];
}

}  // namespace

struct alink_smallres {
    int device = -1;
    int H, W, feat;
    float lr, rho, eps;
    // tower geometry
    int H1, W1;      // after conv2 (valid): H-2
    int P1h, P1w;    // after pool1
    int H3, W3;      // after conv4 (valid): P1-2
    int P2h, P2w;    // after pool2
    int flat;
    // tower parameter offsets in the flat buffer (Keras order)
    size_t oW[4], oB[4], oDW, oDB, ntower;
    alink_head_t* head = nullptr;      // |l-r| -> 128 -> 32 -> 2
    float *d_p = nullptr, *d_g = nullptr, *d_a = nullptr, *d_d = nullptr;   // tower params / grads / Adadelta
    // activations for up to 2*MAXN tower passes
    float *a1 = nullptr, *a2 = nullptr, *p1 = nullptr, *a3 = nullptr, *a4 = nullptr, *p2 = nullptr, *f = nullptr;
    // activation gradients, one buffer per tensor (round 6: a ping-pong pair until then — the weight gradients now run on
    // a side stream beside the input-gradient chain, so a dz must stay put until its weight gradient has read it)
    float *gf = nullptr, *gp2 = nullptr, *ga4 = nullptr, *ga3 = nullptr, *gp1 = nullptr, *ga2 = nullptr, *ga1 = nullptr;
    hipStream_t side = nullptr;        // weight gradients (independent of the dz chain once their dz exists)
    hipEvent_t ev_dz[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, ev_side = nullptr;
    float* ws2 = nullptr;              // split-K slabs of the side stream's GEMMs
    // the whole train step as a captured graph, per distinct (operand pointers, n, flags, lr) — OPTIONAL and off: a step is ~40
    // short launches on two streams, the host needs ~170 us to enqueue them and the device runs dry in the backward; but a replayed
    // node costs more than a launch on this ROCm (0.457 ms replayed against 0.375: tools/experiments/smallres_graph_ab.py).
    // `seen` = 1: the key ran once as plain launches (function attributes set, LDS sized), the next call captures.
    struct StepGraph {
        const void *L, *R, *y, *sw, *masks, *metrics;
        int n, prescale, apply, overlap, seen;
        float grad_scale, lr;
        hipGraphExec_t exec;
    };
    std::vector<StepGraph> graphs;
    bool use_graph = false;
    uint8_t *arg1 = nullptr, *arg2 = nullptr;
    // alink_smallres_train_on_batch_host: the step's operands staged through pinned memory owned by the handle
    float *h_stage = nullptr, *d_stage = nullptr, *h_metrics = nullptr;
    uint8_t* d_masks_own = nullptr;
    int stage_rows = 0;                // pairs the staging is sized for
    float* d_all_grads = nullptr;      // [tower grads | head grads] contiguous copy for all-reduce
    float* ws = nullptr;               // split-K slabs of sgemm
    size_t ws_floats = 0;
    std::vector<void*> allocs;
    ~alink_smallres() {
        for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        for (hipEvent_t e : ev_dz) if (e) (void)hipEventDestroy(e);
        if (ev_side) (void)hipEventDestroy(ev_side);
        for (void* p : allocs) (void)hipFree(p);
        if (h_stage) (void)hipHostFree(h_stage);
        if (h_metrics) (void)hipHostFree(h_metrics);
        if (d_stage) (void)hipFree(d_stage);
        if (d_masks_own) (void)hipFree(d_masks_own);
        if (head) alink_head_destroy(head);
    }
};

namespace {

template <typename V>
int sr_alloc(alink_smallres* m, V** p, size_t count) {
    ALINK_HIP(hipMalloc((void**)p, count * sizeof(V)));
    m->allocs.push_back(*p);
    ALINK_HIP(hipMemset(*p, 0, count * sizeof(V)));
    return ALINK_OK;
}

const int CI[4] = {3, 32, 32, 64}, CO[4] = {32, 32, 64, 64}, PAD[4] = {1, 0, 1, 0};

int run_gemm(alink_smallres* m, GemmP& g, int max_split, hipStream_t st, float* ws = nullptr) {
    gemm32_plan_split(g, max_split);
    ALINK_REQUIRE(gemm32_workspace_floats(g) <= m->ws_floats, ALINK_ENOMEM, "sgemm workspace too small (%zu floats)",
                  gemm32_workspace_floats(g));
    ALINK_HIP(launch_gemm32(g, ws ? ws : m->ws, st));
    return ALINK_OK;
}

// A layer of at most 128 tiles with a long reduction (a small batch's deeper layers: conv4 at 2 x 16 images is 85 tiles x 9 stages on
// 256 CUs, its input gradient 57 tiles x 9) is split over K: the chip fills, and the slab sum — which applies the epilogue — is one
// short launch.  Larger grids are not (the slab sum of a 28,800 x 32 output costs more than the split saves: measured).
inline bool small_grid(const GemmP& g) {
    const int bm = g.N <= 32 ? 128 : 64, bn = g.N <= 32 ? 32 : 64;
    return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) <= 128 && g.K >= 512;
}

// out = relu(conv3x3(in, w[layer]) + b[layer]); in is [nb][H][W][Ci]
int conv_fwd(alink_smallres* m, const float* in, float* out, int layer, int nb, int H, int W, int prescale,
             hipStream_t st, const float* in2 = nullptr, int split = 0) {
    const int Ci = CI[layer], Co = CO[layer], pad = PAD[layer];
    GemmP g{};
    g.A = in; g.A2 = in2; g.a_split = split; g.B = m->d_p + m->oW[layer]; g.C = out;
    g.Ho = H + 2 * pad - 2; g.Wo = W + 2 * pad - 2;
    g.M = nb * g.Ho * g.Wo; g.N = Co; g.K = 9 * Ci; g.ldb = Co; g.ldc = Co;
    g.amode = A_CONV; g.bmode = B_ROW; g.H = H; g.W = W; g.Ci = Ci; g.pad = pad; g.prescale = prescale;
    g.bias = m->d_p + m->oB[layer]; g.relu = 1;
    return run_gemm(m, g, small_grid(g) ? 4 : 1, st);
}

// tower forward on the nb = 2n images of a siamese batch — images 0 .. n-1 from `L`, n .. 2n-1 from `R` (R == nullptr: nb
// images from L) — in ONE pass (until round 6: one pass per side; the two sides share every weight, and a layer's launch on
// 2n images costs what it costs on n: none of them fills the chip); masks == nullptr -> inference (no dropout)
int tower_fwd(alink_smallres* m, const float* L, const float* R, int n, int prescale, const uint8_t* mask1, const uint8_t* mask2,
              hipStream_t st, const MaskDraw* draw = nullptr) {
    const float* P = m->d_p;
    const float keep_scale = 1.f / (1.f - 0.25f);
    const int nb = R ? 2 * n : n;
    int rc;
    {   // the first layer on the vector units (K = 27 is no GEMM: see conv1_fwd_kernel)
        Conv1P c{L, R, P + m->oW[0], P + m->oB[0], m->a1, nb, n, m->H, m->W, prescale};
        MaskDraw mk{nullptr, 0, 0.f, 0ull, 0};
        if (draw) mk = *draw;
        mk.conv_blocks = (int)g1((long long)nb * m->H * m->W * 4).x;
        const int mask_blocks = mk.out ? (int)((mk.count + 1023) / 1024) : 0;      // 4 bytes per thread
        hipLaunchKernelGGL(conv1_fwd_kernel, dim3(mk.conv_blocks + mask_blocks), dim3(256), 0, st, c, mk);
    }
    if ((rc = conv_fwd(m, m->a1, m->a2, 1, nb, m->H, m->W, 0, st))) return rc;
    hipLaunchKernelGGL(pool_fwd_kernel, g1((long long)nb * m->P1h * m->P1w * 32), dim3(256), 0, st, m->a2, m->p1,
                       m->arg1, mask1, keep_scale, nb, m->H1, m->W1, 32);
    if ((rc = conv_fwd(m, m->p1, m->a3, 2, nb, m->P1h, m->P1w, 0, st))) return rc;
    if ((rc = conv_fwd(m, m->a3, m->a4, 3, nb, m->P1h, m->P1w, 0, st))) return rc;
    hipLaunchKernelGGL(pool_fwd_kernel, g1((long long)nb * m->P2h * m->P2w * 64), dim3(256), 0, st, m->a4, m->p2,
                       m->arg2, mask2, keep_scale, nb, m->H3, m->W3, 64);
    // f = relu(p2 . W + b): [nb][flat] x [flat][feat]
    GemmP g{};
    g.A = m->p2; g.B = P + m->oDW; g.C = m->f; g.M = nb; g.N = m->feat; g.K = m->flat;
    g.lda = m->flat; g.ldb = m->feat; g.ldc = m->feat; g.amode = A_ROW; g.bmode = B_ROW;
    g.bias = P + m->oDB; g.relu = 1;
    if ((rc = run_gemm(m, g, 16, st))) return rc;
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

// dW[layer], db[layer] (contiguous in the gradient buffer) from the layer input `in` [nb][H][W][Ci]
// and dz [nb][Ho][Wo][Co]; `accumulate` adds to what is there (second siamese branch of conv1)
int wgrad(alink_smallres* m, const float* in, const float* dz, int layer, int nb, int H, int W, int prescale,
          int accumulate, hipStream_t st, float* ws = nullptr, const float* in2 = nullptr, int split = 0) {
    const int Ci = CI[layer], Co = CO[layer], pad = PAD[layer];
    GemmP g{};
    g.A = in; g.A2 = in2; g.a_split = split; g.B = dz; g.C = m->d_g + m->oW[layer];
    g.Ho = H + 2 * pad - 2; g.Wo = W + 2 * pad - 2;
    g.M = 9 * Ci + 1; g.N = Co; g.K = nb * g.Ho * g.Wo; g.ldb = Co; g.ldc = Co;      // row 9 Ci = bias gradient
    g.amode = A_CONVT; g.bmode = B_ROW; g.H = H; g.W = W; g.Ci = Ci; g.pad = pad; g.prescale = prescale;
    g.accumulate = accumulate;
    return run_gemm(m, g, 128, st, ws);
}

// din [nb][H][W][Ci] = conv-transpose of dz [nb][Ho][Wo][Co] with w[layer], masked by relu'(act) if act
int dgrad(alink_smallres* m, const float* dz, float* din, const float* act, int layer, int nb, int H, int W,
          hipStream_t st) {
    const int Ci = CI[layer], Co = CO[layer], pad = PAD[layer];
    GemmP g{};
    g.A = dz; g.B = m->d_p + m->oW[layer]; g.C = din;
    g.H = H + 2 * pad - 2; g.W = W + 2 * pad - 2;          // the gathered tensor is dz
    g.Ci = Co; g.Ho = H; g.Wo = W; g.pad = 2 - pad;
    g.M = nb * H * W; g.N = Ci; g.K = 9 * Co; g.ldc = Ci;
    g.amode = A_CONV; g.bmode = B_FLIP; g.act = act;
    return run_gemm(m, g, small_grid(g) ? 4 : 1, st);
}

}  // namespace

extern "C" {

alink_smallres_t* alink_smallres_create(int img_h, int img_w, int feat, float lr, float rho, float eps) {
    if (img_h < 12 || img_w < 12 || img_h > 128 || img_w > 128) { set_error("image size %dx%d unsupported", img_h, img_w); return nullptr; }
    if (feat <= 0 || feat % 8) { set_error("feat must be a positive multiple of 8"); return nullptr; }
    alink_smallres* m = new alink_smallres();
    m->device = current_device();
    m->H = img_h; m->W = img_w; m->feat = feat; m->lr = lr; m->rho = rho; m->eps = eps;
    m->H1 = img_h - 2; m->W1 = img_w - 2;
    m->P1h = m->H1 / 2; m->P1w = m->W1 / 2;
    m->H3 = m->P1h - 2; m->W3 = m->P1w - 2;
    m->P2h = m->H3 / 2; m->P2w = m->W3 / 2;
    m->flat = m->P2h * m->P2w * 64;
    size_t o = 0;
    for (int l = 0; l < 4; ++l) { m->oW[l] = o; o += (size_t)9 * CI[l] * CO[l]; m->oB[l] = o; o += CO[l]; }
    m->oDW = o; o += (size_t)m->flat * feat; m->oDB = o; o += feat; m->ntower = o;
    m->head = alink_head_create(feat, 128, 32, lr, rho, eps);
    if (!m->head) { delete m; return nullptr; }
    const size_t nb = 2 * MAXN;
    int rc = 0;
    rc |= sr_alloc(m, &m->d_p, m->ntower);
    rc |= sr_alloc(m, &m->d_a, m->ntower); rc |= sr_alloc(m, &m->d_d, m->ntower);
    rc |= sr_alloc(m, &m->a1, nb * img_h * img_w * 32);
    rc |= sr_alloc(m, &m->a2, nb * m->H1 * m->W1 * 32);
    rc |= sr_alloc(m, &m->p1, nb * m->P1h * m->P1w * 32);
    rc |= sr_alloc(m, &m->a3, nb * m->P1h * m->P1w * 64);
    rc |= sr_alloc(m, &m->a4, nb * m->H3 * m->W3 * 64);
    rc |= sr_alloc(m, &m->p2, nb * m->flat);
    rc |= sr_alloc(m, &m->f, nb * feat);
    rc |= sr_alloc(m, &m->gf, nb * feat);
    rc |= sr_alloc(m, &m->gp2, nb * m->flat);
    rc |= sr_alloc(m, &m->ga4, nb * m->H3 * m->W3 * 64);
    rc |= sr_alloc(m, &m->ga3, nb * m->P1h * m->P1w * 64);
    rc |= sr_alloc(m, &m->gp1, nb * m->P1h * m->P1w * 32);
    rc |= sr_alloc(m, &m->ga2, nb * m->H1 * m->W1 * 32);
    rc |= sr_alloc(m, &m->ga1, nb * img_h * img_w * 32);
    rc |= sr_alloc(m, &m->arg1, nb * m->P1h * m->P1w * 32);
    rc |= sr_alloc(m, &m->arg2, nb * m->P2h * m->P2w * 64);
    // the tower's gradients are written where the data-parallel caller reads them: [tower | head] in one buffer
    rc |= sr_alloc(m, &m->d_all_grads, m->ntower + alink_head_num_params(m->head) + 4);      // + 4 spare floats (a data-parallel step's metrics travel with the gradients)
    m->d_g = m->d_all_grads;
    m->ws_floats = (size_t)16 << 20;
    rc |= sr_alloc(m, &m->ws, m->ws_floats);
    rc |= sr_alloc(m, &m->ws2, m->ws_floats);
    if (rc) { delete m; return nullptr; }
    bool ok = hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) == hipSuccess;
    for (hipEvent_t& e : m->ev_dz) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&m->ev_side, hipEventDisableTiming) == hipSuccess;
    if (!ok) { set_error("SmallRes: side stream / events could not be created"); delete m; return nullptr; }
    return m;
}

void alink_smallres_destroy(alink_smallres_t* m) {
    if (!m) return;
    DeviceGuard dg(m->device);
    delete m;
}
size_t alink_smallres_num_params(const alink_smallres_t* m) { return m ? m->ntower + alink_head_num_params(m->head) : 0; }

int alink_smallres_set_params(alink_smallres_t* m, const float* host, size_t count) {
    ALINK_REQUIRE(m && host, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(m->device);
    ALINK_REQUIRE(count == alink_smallres_num_params(m), ALINK_EINVAL, "expected %zu parameters, got %zu",
                  alink_smallres_num_params(m), count);
    ALINK_HIP(hipMemcpy(m->d_p, host, m->ntower * sizeof(float), hipMemcpyHostToDevice));
    return alink_head_set_params(m->head, host + m->ntower, count - m->ntower);
}
int alink_smallres_get_params(const alink_smallres_t* m, float* host, size_t count) {
    ALINK_REQUIRE(m && host, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(count == alink_smallres_num_params(m), ALINK_EINVAL, "bad parameter count");
    DeviceGuard dg(m->device);
    ALINK_HIP(hipDeviceSynchronize());
    ALINK_HIP(hipMemcpy(host, m->d_p, m->ntower * sizeof(float), hipMemcpyDeviceToHost));
    return alink_head_get_params(m->head, host + m->ntower, count - m->ntower);
}
int alink_smallres_set_lr(alink_smallres_t* m, float lr) {
    ALINK_REQUIRE(m && lr >= 0.f, ALINK_EINVAL, "bad lr");
    m->lr = lr;
    return alink_head_set_lr(m->head, lr);
}
float* alink_smallres_grads_dev(alink_smallres_t* m) { return m ? m->d_all_grads : nullptr; }

int alink_smallres_mask_sizes(const alink_smallres_t* m, int* a, int* b) {
    ALINK_REQUIRE(m && a && b, ALINK_EINVAL, "NULL argument");
    *a = m->P1h * m->P1w * 32;
    *b = m->P2h * m->P2w * 64;
    return ALINK_OK;
}

int alink_smallres_forward(alink_smallres_t* m, const float* dev_L, const float* dev_R, int n, int prescale,
                           float* dev_probs, void* stream) {
    ALINK_REQUIRE(m && dev_L && dev_R && dev_probs, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= MAXN, ALINK_EINVAL, "n=%d outside 1..%d", n, MAXN);
    DeviceGuard dg(m->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = tower_fwd(m, dev_L, dev_R, n, prescale, nullptr, nullptr, st);       // features of L in f[0:n], of R in f[n:2n]
    if (rc) return rc;
    return alink_head_forward(m->head, m->f, m->f + (size_t)n * m->feat, nullptr, nullptr, n, dev_probs, stream);
}

int alink_smallres_eval(alink_smallres_t* m, const float* dev_L, const float* dev_R, const float* dev_y, int n,
                        int prescale, float* dev_metrics, void* stream) {
    ALINK_REQUIRE(m && dev_L && dev_R && dev_y && dev_metrics, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= MAXN, ALINK_EINVAL, "n=%d outside 1..%d", n, MAXN);
    DeviceGuard dg(m->device);
    int rc = tower_fwd(m, dev_L, dev_R, n, prescale, nullptr, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    return alink_head_eval(m->head, m->f, m->f + (size_t)n * m->feat, dev_y, n, dev_metrics, stream);
}

static bool g_smallres_overlap = true;
void alink_debug_set_smallres_overlap(int on) { g_smallres_overlap = on != 0; }      // include/alink_hip_debug.h
static bool g_smallres_one_update = true;
void alink_debug_set_smallres_one_update(int on) { g_smallres_one_update = on != 0; }

// draw_seed != nullptr: dev_masks is an OUTPUT first — the step's first launch fills it with alink_keep_masks(…, 0.75, *draw_seed)
static int train_step_launches(alink_smallres_t* m, const float* dev_L, const float* dev_R, const float* dev_y,
                               const float* dev_sw, int n, int prescale, const uint8_t* dev_masks, float grad_scale,
                               int apply, float* dev_metrics, void* stream, const uint64_t* draw_seed = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    const size_t e1 = (size_t)m->P1h * m->P1w * 32, e2 = (size_t)m->P2h * m->P2w * 64;
    const uint8_t* m1 = dev_masks;
    const uint8_t* m2 = dev_masks ? dev_masks + 2 * (size_t)n * e1 : nullptr;
    MaskDraw mk{const_cast<uint8_t*>(dev_masks), (long long)(2 * (size_t)n * (e1 + e2)), 1.f - 0.25f, draw_seed ? *draw_seed : 0ull, 0};
    int rc = tower_fwd(m, dev_L, dev_R, n, prescale, m1, m2, st, draw_seed && dev_masks ? &mk : nullptr);
    if (rc) return rc;
    float* fL = m->f;
    float* fR = m->f + (size_t)n * m->feat;
    // the head's gradients, and dz of the tower's Dense(feat, relu) straight from the head: d loss / d feature times relu'(feature),
    // with its column sums — that layer's bias gradient (a launch of its own on the side stream until round 6)
    if ((rc = alink_head_train_step_input_grads(m->head, fL, fR, dev_y, dev_sw, n, grad_scale, 1, m->gf, m->gf + (size_t)n * m->feat,
                                                m->d_g + m->oDB, dev_metrics, stream))) return rc;
    // The shared tower saw 2n images ([L ; R] contiguous in every activation buffer): ONE backward pass over 2n images
    // gives both branches' weight gradients.  Two chains from here: the input gradients (dz of a layer from the dz of
    // the next) on the caller's stream, and the weight gradients — each needs only its layer's input and dz — on a side
    // stream beside it (round 6; ws2 = that stream's own split-K slabs).
    const bool two = g_smallres_overlap;
    hipStream_t sw = two ? m->side : st;
    float* wsw = two ? m->ws2 : m->ws;
    auto fork = [&](int i) -> int {        // the side stream may start what needs the dz just produced
        if (!two) return ALINK_OK;
        ALINK_HIP(hipEventRecord(m->ev_dz[i], st));
        ALINK_HIP(hipStreamWaitEvent(m->side, m->ev_dz[i], 0));
        return ALINK_OK;
    };
    const float* P = m->d_p;
    float* G = m->d_g;
    const int nb = 2 * n;
    const float keep_scale = 1.f / (1.f - 0.25f);
    if ((rc = fork(0))) return rc;
    {   // gW[flat][feat] = p2^T . dz
        GemmP g{};
        g.A = m->p2; g.B = m->gf; g.C = G + m->oDW; g.M = m->flat; g.N = m->feat; g.K = nb;
        g.lda = m->flat; g.ldb = m->feat; g.ldc = m->feat; g.amode = A_COL; g.bmode = B_ROW;
        if ((rc = run_gemm(m, g, 1, sw, wsw))) return rc;
    }
    {   // d(p2)[nb][flat] = dz . W^T
        GemmP g{};
        g.A = m->gf; g.B = P + m->oDW; g.C = m->gp2; g.M = nb; g.N = m->flat; g.K = m->feat;
        g.lda = m->feat; g.ldb = m->feat; g.ldc = m->flat; g.amode = A_ROW; g.bmode = B_COLT;
        if ((rc = run_gemm(m, g, 16, st))) return rc;
    }
    const bool early = two && apply;
    hipLaunchKernelGGL(pool_bwd_kernel, g1((long long)nb * m->H3 * m->W3 * 64), dim3(256), 0, st, m->gp2, m->arg2, m2,
                       keep_scale, m->a4, m->ga4, nb, m->H3, m->W3, 64);
    if ((rc = fork(1))) return rc;
    if (early) {
        // The wide Dense layer is 99 % of the parameters (flat x feat = 4.7 M of 4.8 M at 32 x 32 / 2048) and its update 20 us
        // of HBM traffic: it runs on the side stream beside the dz chain instead of after it — behind this fork, which says both
        // that its gradient exists (the side stream's own order) and that the input gradient above has read the old W (an
        // event of its own for that cost the dz chain ~6 us: a record in a stream delays the stream's next kernel by that much).
        const size_t nd = m->ntower - m->oDW;
        hipLaunchKernelGGL(adadelta2_kernel, g1((long long)nd), dim3(256), 0, m->side, m->d_p + m->oDW, m->d_g + m->oDW,
                           m->d_a + m->oDW, m->d_d + m->oDW, nd, m->lr, m->rho, m->eps);
    }
    if ((rc = wgrad(m, m->a3, m->ga4, 3, nb, m->P1h, m->P1w, 0, 0, sw, wsw))) return rc;
    if ((rc = dgrad(m, m->ga4, m->ga3, m->a3, 3, nb, m->P1h, m->P1w, st))) return rc;
    if ((rc = fork(2))) return rc;
    if ((rc = wgrad(m, m->p1, m->ga3, 2, nb, m->P1h, m->P1w, 0, 0, sw, wsw))) return rc;
    if ((rc = dgrad(m, m->ga3, m->gp1, nullptr, 2, nb, m->P1h, m->P1w, st))) return rc;
    hipLaunchKernelGGL(pool_bwd_kernel, g1((long long)nb * m->H1 * m->W1 * 32), dim3(256), 0, st, m->gp1, m->arg1, m1,
                       keep_scale, m->a2, m->ga2, nb, m->H1, m->W1, 32);
    if ((rc = fork(3))) return rc;
    if ((rc = wgrad(m, m->a1, m->ga2, 1, nb, m->H, m->W, 0, 0, sw, wsw))) return rc;
    if ((rc = dgrad(m, m->ga2, m->ga1, m->a1, 1, nb, m->H, m->W, st))) return rc;
    // conv1's weight + bias gradient (the last link of the dz chain: on the caller's stream): partial sums over 128 pixels per
    // workgroup on the vector units, then the slab sum (which runs while the join below waits for the side stream: folding it
    // into the update that follows the join put its 256 loads per output on the critical path — measured slower by 2.5 us)
    {
        const long long Px = (long long)nb * m->H * m->W;
        const int blocks = (int)((Px + 127) / 128);
        ALINK_REQUIRE((size_t)blocks * 28 * 32 <= m->ws_floats, ALINK_ENOMEM, "workspace too small for conv1's gradient");
        Conv1P c{dev_L, dev_R, nullptr, nullptr, nullptr, nb, n, m->H, m->W, prescale};
        hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(blocks), dim3(256), 0, st, c, m->ga1, m->ws);
        ALINK_HIP(launch_slab_sum(m->ws, G + m->oW[0], 28 * 32, blocks, st));
    }
    ALINK_HIP(hipGetLastError());
    if (two) {                             // join: everything after this sees every gradient
        ALINK_HIP(hipEventRecord(m->ev_side, m->side));
        ALINK_HIP(hipStreamWaitEvent(st, m->ev_side, 0));
    }
    if (apply) {
        const size_t nu = early ? m->oDW : m->ntower;        // (the Dense layer's share is already under way)
        if (!g_smallres_one_update) {
            hipLaunchKernelGGL(adadelta2_kernel, g1((long long)nu), dim3(256), 0, st, m->d_p, m->d_g, m->d_a, m->d_d, nu, m->lr, m->rho, m->eps);
            ALINK_HIP(hipGetLastError());
            return alink_head_apply_update(m->head, stream);
        }
        // ONE launch ends the step: the tower's update from the gradients where they are and the head's from its own buffer
        return alink_head_apply_update_with(m->head, m->d_p, m->d_g, m->d_a, m->d_d, nu, stream);
    }
    // gradients only (a data-parallel caller all-reduces the contiguous buffer, then alink_smallres_apply_update)
    ALINK_HIP(hipMemcpyAsync(m->d_all_grads + m->ntower, alink_head_grads_dev(m->head),
                             alink_head_num_params(m->head) * sizeof(float), hipMemcpyDeviceToDevice, st));
    return ALINK_OK;
}

int alink_smallres_set_graph(alink_smallres_t* m, int on) {
    ALINK_REQUIRE(m, ALINK_EINVAL, "NULL model");
    m->use_graph = on != 0;
    return ALINK_OK;
}

int alink_smallres_train_step(alink_smallres_t* m, const float* dev_L, const float* dev_R, const float* dev_y,
                              const float* dev_sw, int n, int prescale, const uint8_t* dev_masks, float grad_scale,
                              int apply, float* dev_metrics, void* stream) {
    ALINK_REQUIRE(m && dev_L && dev_R && dev_y && dev_metrics, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= MAXN, ALINK_EINVAL, "n=%d outside 1..%d", n, MAXN);
    DeviceGuard dg(m->device);
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    // the legacy default stream cannot be captured; a stream the caller is already capturing simply receives the launches
    const bool can_graph = m->use_graph && st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    if (!can_graph) return train_step_launches(m, dev_L, dev_R, dev_y, dev_sw, n, prescale, dev_masks, grad_scale, apply, dev_metrics, stream);
    const int overlap = g_smallres_overlap ? 1 : 0;
    alink_smallres::StepGraph* hit = nullptr;
    for (auto& g : m->graphs)
        if (g.L == dev_L && g.R == dev_R && g.y == dev_y && g.sw == dev_sw && g.masks == dev_masks && g.metrics == dev_metrics && g.n == n &&
            g.prescale == prescale && g.apply == apply && g.overlap == overlap && g.grad_scale == grad_scale && g.lr == m->lr) { hit = &g; break; }
    if (hit && hit->exec) {
        ALINK_HIP(hipGraphLaunch(hit->exec, st));
        if (apply) (void)alink_head_params_dev(m->head);      // the head's derived weight copies are stale (what the plain path notes on the host)
        return ALINK_OK;
    }
    if (!hit) {                                               // first sight of these operands: plain launches (and every one-off set-up they trigger)
        if (m->graphs.size() >= 8) {
            if (m->graphs.front().exec) (void)hipGraphExecDestroy(m->graphs.front().exec);
            m->graphs.erase(m->graphs.begin());
        }
        m->graphs.push_back({dev_L, dev_R, dev_y, dev_sw, dev_masks, dev_metrics, n, prescale, apply, overlap, 1, grad_scale, m->lr, nullptr});
        return train_step_launches(m, dev_L, dev_R, dev_y, dev_sw, n, prescale, dev_masks, grad_scale, apply, dev_metrics, stream);
    }
    hipGraph_t graph = nullptr;
    ALINK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = train_step_launches(m, dev_L, dev_R, dev_y, dev_sw, n, prescale, dev_masks, grad_scale, apply, dev_metrics, stream);
    const hipError_t ee = hipStreamEndCapture(st, &graph);
    hipGraphExec_t exec = nullptr;
    hipError_t ei = hipErrorUnknown;
    if (!rc && ee == hipSuccess && graph) ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc || ee != hipSuccess || ei != hipSuccess) {
        (void)hipGetLastError();
        if (rc) return rc;
        m->use_graph = false;                                 // capture is not available here: plain launches from now on
        return train_step_launches(m, dev_L, dev_R, dev_y, dev_sw, n, prescale, dev_masks, grad_scale, apply, dev_metrics, stream);
    }
    hit->exec = exec;
    ALINK_HIP(hipGraphLaunch(exec, st));
    if (apply) (void)alink_head_params_dev(m->head);
    return ALINK_OK;
}

int alink_smallres_train_step_drawn(alink_smallres_t* m, const float* dev_L, const float* dev_R, const float* dev_y,
                                    const float* dev_sw, int n, int prescale, uint8_t* dev_masks, uint64_t mask_seed,
                                    float grad_scale, int apply, float* dev_metrics, void* stream) {
    ALINK_REQUIRE(m && dev_L && dev_R && dev_y && dev_metrics && dev_masks, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= MAXN, ALINK_EINVAL, "n=%d outside 1..%d", n, MAXN);
    if (m->use_graph) {                    // (a captured step's arguments are fixed, a step's seed is not: the masks in a launch of their own)
        int a = 0, b = 0;
        alink_smallres_mask_sizes(m, &a, &b);
        const int rc = alink_keep_masks(dev_masks, (int64_t)2 * n * ((int64_t)a + b), 0.75f, mask_seed, stream);
        if (rc) return rc;
        return alink_smallres_train_step(m, dev_L, dev_R, dev_y, dev_sw, n, prescale, dev_masks, grad_scale, apply, dev_metrics, stream);
    }
    DeviceGuard dg(m->device);
    return train_step_launches(m, dev_L, dev_R, dev_y, dev_sw, n, prescale, dev_masks, grad_scale, apply, dev_metrics, stream, &mask_seed);
}

int alink_smallres_train_on_batch_host(alink_smallres_t* m, const float* host_L, const float* host_R, const float* host_y,
                                       const float* host_sw, int n, int prescale, int dropout, uint64_t mask_seed,
                                       float* host_metrics, void* stream) {
    ALINK_REQUIRE(m && host_L && host_R && host_y && host_metrics, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= MAXN, ALINK_EINVAL, "n=%d outside 1..%d", n, MAXN);
    DeviceGuard dg(m->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t img = (size_t)m->H * m->W * 3;
    const size_t e1 = (size_t)m->P1h * m->P1w * 32, e2 = (size_t)m->P2h * m->P2w * 64;
    if (n > m->stage_rows) {               // (grown to the largest batch seen; a step is synchronous, so nothing is in flight here)
        ALINK_HIP(hipStreamSynchronize(st));
        if (m->h_stage) (void)hipHostFree(m->h_stage);
        if (m->d_stage) (void)hipFree(m->d_stage);
        if (m->d_masks_own) (void)hipFree(m->d_masks_own);
        m->h_stage = m->d_stage = nullptr; m->d_masks_own = nullptr; m->stage_rows = 0;
        const int rows = n < 16 ? 16 : n;
        const size_t floats = (size_t)rows * (2 * img + 4);
        ALINK_HIP(hipHostMalloc((void**)&m->h_stage, floats * sizeof(float), hipHostMallocDefault));
        ALINK_HIP(hipMalloc((void**)&m->d_stage, floats * sizeof(float)));
        ALINK_HIP(hipMalloc((void**)&m->d_masks_own, 2 * (size_t)rows * (e1 + e2)));
        if (!m->h_metrics) ALINK_HIP(hipHostMalloc((void**)&m->h_metrics, 4 * sizeof(float), hipHostMallocDefault));
        m->stage_rows = rows;
    }
    // [L | R | y | sw] in one pinned block, one upload
    const size_t oL = 0, oR = (size_t)n * img, oy = 2 * (size_t)n * img, osw = oy + 2 * (size_t)n, total = osw + (host_sw ? (size_t)n : 0);
    memcpy(m->h_stage + oL, host_L, (size_t)n * img * sizeof(float));
    memcpy(m->h_stage + oR, host_R, (size_t)n * img * sizeof(float));
    memcpy(m->h_stage + oy, host_y, 2 * (size_t)n * sizeof(float));
    if (host_sw) memcpy(m->h_stage + osw, host_sw, (size_t)n * sizeof(float));
    ALINK_HIP(hipMemcpyAsync(m->d_stage, m->h_stage, total * sizeof(float), hipMemcpyHostToDevice, st));
    const float* dsw = host_sw ? m->d_stage + osw : nullptr;
    int rc;
    if (dropout && m->use_graph) {
        rc = alink_smallres_train_step_drawn(m, m->d_stage + oL, m->d_stage + oR, m->d_stage + oy, dsw, n, prescale, m->d_masks_own, mask_seed,
                                             0.f, 1, m->h_metrics, stream);
    } else {
        rc = train_step_launches(m, m->d_stage + oL, m->d_stage + oR, m->d_stage + oy, dsw, n, prescale, dropout ? m->d_masks_own : nullptr,
                                 0.f, 1, m->h_metrics, stream, dropout ? &mask_seed : nullptr);
    }
    if (rc) return rc;
    ALINK_HIP(hipStreamSynchronize(st));
    host_metrics[0] = m->h_metrics[0];
    host_metrics[1] = m->h_metrics[1];
    return ALINK_OK;
}

int alink_smallres_apply_update(alink_smallres_t* m, void* stream) {
    ALINK_REQUIRE(m, ALINK_EINVAL, "NULL model");
    DeviceGuard dg(m->device);
    hipStream_t st = (hipStream_t)stream;
    // gradients are taken from the contiguous buffer (the caller may have all-reduced it)
    hipLaunchKernelGGL(adadelta2_kernel, g1((long long)m->ntower), dim3(256), 0, st, m->d_p, m->d_all_grads, m->d_a,
                       m->d_d, m->ntower, m->lr, m->rho, m->eps);
    ALINK_HIP(hipGetLastError());
    ALINK_HIP(hipMemcpyAsync(alink_head_grads_dev(m->head), m->d_all_grads + m->ntower,
                             alink_head_num_params(m->head) * sizeof(float), hipMemcpyDeviceToDevice, st));
    return alink_head_apply_update(m->head, stream);
}

}  // extern "C"
