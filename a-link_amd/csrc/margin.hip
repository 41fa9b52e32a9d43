// margin.hip — two training losses named by BASELINE.json's north_star that the REFERENCE DOES NOT CONTAIN
// (the ArcFace checkpoint is cut at fc1_output, reference code/face_model.py:35-36,53 — the margin head
// is never built; the pair scorer is |l - r| -> Dense -> softmax with binary cross-entropy, reference
// code/siamese.py:27-35 — no contrastive loss).  Both are therefore EXTENSIONS, labelled as such in the header, in
// a-link_amd/extensions.py and in DESIGN.md, and checked against torch autograd (tests/test_gpu_extensions.py), not
// against the reference.
//
//   X1  additive angular margin softmax (ArcFace, Deng et al. 2019; insightface's fc7 + margin):
//         c_ij = <e_i/|e_i|, w_j/|w_j|>,  target t = y_i:  c'_it = cos(theta_it + m) if c_it > cos(pi - m)
//                                                               else c_it - m sin(pi - m)        (easy_margin: c_it > 0 ? cos(theta + m) : c_it)
//         loss = mean_i CE(softmax(s c'_i), y_i);  gradients w.r.t. the raw embeddings and the raw class centres.
//       Three exact-f32 MFMA GEMMs (sgemm.hip: cos = E^ W^T, dE^ = dC W^, dW^ = dC^T E^) + row kernels.
//   X2  pairwise-L2 contrastive loss (Hadsell, Chopra, LeCun 2006, in the form of Keras' mnist_siamese example):
//         d_p = sqrt(max(|l_p - r_p|^2, 1e-7)),  loss = mean_p [ y_p d_p^2 + (1 - y_p) max(margin - d_p, 0)^2 ]
//       HBM-bound: one wave per pair, 2 D floats read, 2 D floats of gradient written.
#include "alink_common.h"
#include "sgemm.h"

namespace alink {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out[r][:] = in[r][:] / |in[r]|, norm[r] = |in[r]| (1 for a zero row).  One wave per row.
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     float* __restrict__ norm, int rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* x = in + (size_t)r * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s = fmaf(x[k], x[k], s);
    s = wave_sum(s);
    const float n = s > 0.f ? sqrtf(s) : 1.f;
    if (lane == 0) norm[r] = n;
    const float inv = 1.f / n;
    for (int k = lane; k < D; k += 64) out[(size_t)r * D + k] = x[k] * inv;
}

// d(raw)[r] = (d(hat)[r] - hat[r] <hat[r], d(hat)[r]>) / norm[r]
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ dhat, const float* __restrict__ hat,
                                                         const float* __restrict__ norm, float* __restrict__ draw,
                                                         int rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* g = dhat + (size_t)r * D;
    const float* h = hat + (size_t)r * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s = fmaf(g[k], h[k], s);
    s = wave_sum(s);
    const float inv = 1.f / norm[r];
    for (int k = lane; k < D; k += 64) draw[(size_t)r * D + k] = (g[k] - h[k] * s) * inv;
}

struct MarginP {
    float* cosm;            // [N][C] in: cosines; out: d(loss)/d(cosine)
    const int32_t* labels;  // [N]
    float* row_loss;        // [N]
    int N, C;
    float s, cos_m, sin_m, thresh, mm, inv_n;
    int easy;
};

// One workgroup per row: margin on the target column, softmax cross-entropy, gradient back through s and the margin.
__global__ __launch_bounds__(256) void margin_softmax_kernel(const MarginP p) {
    __shared__ float red[4];
    __shared__ float bc[2];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* row = p.cosm + (size_t)i * p.C;
    const int t = p.labels[i];
    const float ct = row[t];
    const float sin_t = sqrtf(fmaxf(1.f - ct * ct, 1e-12f));
    const bool cond = p.easy ? ct > 0.f : ct > p.thresh;
    const float ct_new = cond ? ct * p.cos_m - sin_t * p.sin_m : (p.easy ? ct : ct - p.mm);
    const float dnew = cond ? p.cos_m + ct / sin_t * p.sin_m : 1.f;
    // max
    float mx = -INFINITY;
    for (int j = tid; j < p.C; j += 256) mx = fmaxf(mx, p.s * (j == t ? ct_new : row[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[w] = mx;
    __syncthreads();
    if (tid == 0) bc[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    mx = bc[0];
    float se = 0.f;
    for (int j = tid; j < p.C; j += 256) se += expf(p.s * (j == t ? ct_new : row[j]) - mx);
    se = wave_sum(se);
    __syncthreads();
    if (lane == 0) red[w] = se;
    __syncthreads();
    if (tid == 0) bc[1] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    se = bc[1];
    if (tid == 0) p.row_loss[i] = (mx + logf(se)) - p.s * ct_new;
    const float gs = p.s * p.inv_n;
    for (int j = tid; j < p.C; j += 256) {
        const float z = p.s * (j == t ? ct_new : row[j]);
        const float pr = expf(z - mx) / se;
        row[j] = j == t ? (pr - 1.f) * gs * dnew : pr * gs;
    }
}

// out[0] = scale * sum(v[0..n)) in index order (one workgroup: deterministic)
__global__ __launch_bounds__(256) void ordered_sum_kernel(const float* __restrict__ v, long long n, float scale,
                                                         float* __restrict__ out) {
    __shared__ float part[256];
    const long long per = (n + 255) / 256, lo = per * threadIdx.x, hi = lo + per < n ? lo + per : n;
    float s = 0.f;
    for (long long i = lo; i < hi; ++i) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i) t += part[i];
        out[0] = t * scale;
    }
}

// One wave per pair.
__global__ __launch_bounds__(256) void contrastive_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                         const float* __restrict__ y, long long P, int D, float margin,
                                                         float gscale, float* __restrict__ pair_loss,
                                                         float* __restrict__ dL, float* __restrict__ dR) {
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    const float* l = L + (size_t)p * D;
    const float* r = R + (size_t)p * D;
    float s = 0.f;
    for (int k = lane * 4; k < D; k += 256) {
        const f32x4 a = *(const f32x4*)(l + k), b = *(const f32x4*)(r + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(a[j] - b[j], a[j] - b[j], s);
    }
    s = wave_sum(s);
    const bool clamped = !(s > 1e-7f);                      // K.maximum(sum_sq, K.epsilon()): no gradient when clamped
    const float d = sqrtf(clamped ? 1e-7f : s);
    const float yy = y[p], gap = fmaxf(margin - d, 0.f);
    if (lane == 0) pair_loss[p] = yy * d * d + (1.f - yy) * gap * gap;
    if (!dL) return;
    // d(loss)/d(d) = 2 y d - 2 (1 - y) max(margin - d, 0);  d(d)/d(l) = (l - r) / d
    const float coef = clamped ? 0.f : gscale * (2.f * yy * d - 2.f * (1.f - yy) * gap) / d;
    for (int k = lane * 4; k < D; k += 256) {
        const f32x4 a = *(const f32x4*)(l + k), b = *(const f32x4*)(r + k);
        f32x4 g;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = coef * (a[j] - b[j]);
        *(f32x4*)(dL + (size_t)p * D + k) = g;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = -g[j];
        *(f32x4*)(dR + (size_t)p * D + k) = g;
    }
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct MarginLayout {
    size_t ehat, what, enorm, wnorm, cosm, rowloss, dhat, gemm, total;
};
MarginLayout margin_layout(int N, int D, int C) {
    MarginLayout L;
    size_t o = 0;
    L.ehat = o; o += al256((size_t)N * D * 4);
    L.what = o; o += al256((size_t)C * D * 4);
    L.enorm = o; o += al256((size_t)N * 4);
    L.wnorm = o; o += al256((size_t)C * 4);
    L.cosm = o; o += al256((size_t)N * C * 4);
    L.rowloss = o; o += al256((size_t)N * 4);
    L.dhat = o; o += al256((size_t)(N > C ? N : C) * D * 4);
    L.gemm = o; o += (size_t)16 << 20;                        // split-K slabs of the three GEMMs
    L.total = o;
    return L;
}

int gemm(GemmP& g, float* ws, size_t ws_floats, int max_split, hipStream_t st) {
    gemm32_plan_split(g, max_split);
    ALINK_REQUIRE(gemm32_workspace_floats(g) <= ws_floats, ALINK_ENOMEM, "sgemm workspace too small (%zu floats)",
                  gemm32_workspace_floats(g));
    ALINK_HIP(launch_gemm32(g, ws, st));
    return ALINK_OK;
}

}  // namespace
}  // namespace alink

using namespace alink;

extern "C" {

size_t alink_arcface_margin_workspace_bytes(int N, int D, int C) {
    if (N <= 0 || D <= 0 || C <= 0) return 0;
    return margin_layout(N, D, C).total;
}

int alink_arcface_margin_loss(const float* dev_emb, const float* dev_W, const int32_t* dev_labels, int N, int D, int C,
                              float s, float m, int easy_margin, float* dev_loss, float* dev_demb, float* dev_dW,
                              void* dev_workspace, size_t workspace_bytes, void* stream) {
    ALINK_REQUIRE(dev_emb && dev_W && dev_labels && dev_loss && dev_workspace, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(N > 0 && C > 1 && D > 0 && D % 4 == 0, ALINK_EINVAL, "N=%d C=%d D=%d: need N > 0, C > 1, D a multiple of 4", N, C, D);
    ALINK_REQUIRE((long long)N * C < (1ll << 31) && (long long)C * D < (1ll << 31), ALINK_EINVAL, "N*C or C*D exceeds 2^31");
    ALINK_REQUIRE(s > 0.f && m >= 0.f && m < 3.14159265f, ALINK_EINVAL, "scale must be > 0 and 0 <= m < pi");
    ALINK_REQUIRE(((uintptr_t)dev_workspace & 255) == 0, ALINK_EINVAL, "workspace must be 256-byte aligned");
    const MarginLayout L = margin_layout(N, D, C);
    ALINK_REQUIRE(workspace_bytes >= L.total, ALINK_ENOMEM, "workspace too small: %zu < %zu", workspace_bytes, L.total);
    DeviceGuard dg(device_of_pointer(dev_emb));
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)dev_workspace;
    float *ehat = (float*)(base + L.ehat), *what = (float*)(base + L.what), *enorm = (float*)(base + L.enorm),
          *wnorm = (float*)(base + L.wnorm), *cosm = (float*)(base + L.cosm), *rowloss = (float*)(base + L.rowloss),
          *dhat = (float*)(base + L.dhat), *gws = (float*)(base + L.gemm);
    const size_t gws_floats = ((size_t)16 << 20) / 4;
    hipLaunchKernelGGL(rownorm_kernel, dim3((N + 3) / 4), dim3(256), 0, st, dev_emb, ehat, enorm, N, D);
    hipLaunchKernelGGL(rownorm_kernel, dim3((C + 3) / 4), dim3(256), 0, st, dev_W, what, wnorm, C, D);
    int rc;
    {   // cos[N][C] = E^ . W^^T
        GemmP g{};
        g.A = ehat; g.B = what; g.C = cosm; g.M = N; g.N = C; g.K = D; g.lda = D; g.ldb = D; g.ldc = C;
        g.amode = A_ROW; g.bmode = B_COLT;
        if ((rc = gemm(g, gws, gws_floats, 8, st))) return rc;
    }
    MarginP p{};
    p.cosm = cosm; p.labels = dev_labels; p.row_loss = rowloss; p.N = N; p.C = C; p.s = s;
    p.cos_m = cosf(m); p.sin_m = sinf(m); p.thresh = cosf(3.14159265358979f - m); p.mm = sinf(3.14159265358979f - m) * m;
    p.inv_n = 1.f / (float)N; p.easy = easy_margin ? 1 : 0;
    hipLaunchKernelGGL(margin_softmax_kernel, dim3(N), dim3(256), 0, st, p);
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, st, rowloss, (long long)N, 1.f / (float)N, dev_loss);
    if (dev_demb) {   // dE^ = dC . W^, then through the row normalisation
        GemmP g{};
        g.A = cosm; g.B = what; g.C = dhat; g.M = N; g.N = D; g.K = C; g.lda = C; g.ldb = D; g.ldc = D;
        g.amode = A_ROW; g.bmode = B_ROW;
        if ((rc = gemm(g, gws, gws_floats, 16, st))) return rc;
        hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, st, dhat, ehat, enorm, dev_demb, N, D);
    }
    if (dev_dW) {     // dW^ = dC^T . E^
        GemmP g{};
        g.A = cosm; g.B = ehat; g.C = dhat; g.M = C; g.N = D; g.K = N; g.lda = C; g.ldb = D; g.ldc = D;
        g.amode = A_COL; g.bmode = B_ROW;
        if ((rc = gemm(g, gws, gws_floats, 16, st))) return rc;
        hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((C + 3) / 4), dim3(256), 0, st, dhat, what, wnorm, dev_dW, C, D);
    }
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_contrastive_loss(const float* dev_L, const float* dev_R, const float* dev_y, int64_t P, int D, float margin,
                           float* dev_loss, float* dev_pair_loss, float* dev_dL, float* dev_dR, void* stream) {
    ALINK_REQUIRE(dev_L && dev_R && dev_y && dev_loss && dev_pair_loss, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE((dev_dL == nullptr) == (dev_dR == nullptr), ALINK_EINVAL, "pass both gradient buffers or neither");
    ALINK_REQUIRE(P > 0 && P < (1ll << 31) && D > 0 && D % 4 == 0, ALINK_EINVAL, "P=%lld D=%d: need 0 < P < 2^31, D a multiple of 4",
                  (long long)P, D);
    ALINK_REQUIRE(margin >= 0.f, ALINK_EINVAL, "margin must be >= 0");
    DeviceGuard dg(device_of_pointer(dev_L));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(contrastive_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, dev_L, dev_R, dev_y, (long long)P,
                       D, margin, 1.f / (float)P, dev_pair_loss, dev_dL, dev_dR);
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, st, dev_pair_loss, (long long)P, 1.f / (float)P, dev_loss);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

}  // extern "C"
