// backward.hip — the small kernels of the backbone's input-gradient pass (the convolutions of that
// pass are conv3x3_direct / conv_igemm with transposed, flipped weights: backbone.hip).
//
// The reference has no gradient attack (code/attack.py is a black-box few-pixel attack);
// BASELINE.json's north_star and SURVEY.md §8f N1 name FGSM/PGD as a labelled EXTENSION of the A2-LINK
// noise stage.  It needs d(loss)/d(pixels) through the frozen IR-ResNet (inference-mode BN, so every
// layer is linear except the PReLUs and the final L2 normalisation):
//   * l2norm_bwd_kernel     e = z/|z|:  dz = (g - (g.e) e) / |z|      (sklearn semantics: |z| = 0 -> 1)
//   * zero_insert_kernel    stride-2 layers: dy placed on the even positions of a zero map, after which
//                           the transposed convolution is an ordinary stride-1 one
//   * stem_bwd_kernel       d(pixels) from d(stem output): PReLU' of the stored stem activation, the
//                           transposed 3x3x3x64 convolution (K = 576 -> 3 outputs: VALU) and the 1/128
//                           of the input normalisation
#include "alink_common.h"

namespace alink {
namespace {

template <typename T> struct Vec8;
template <> struct Vec8<__bf16>   { typedef bf16x8 type; };
template <> struct Vec8<_Float16> { typedef f16x8 type; };

// one wave per row
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ e,
                                                         const float* __restrict__ norms, T* __restrict__ dz, int M,
                                                         int E) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float dot = 0.f;
    for (int c = lane; c < E; c += 64) dot += g[(size_t)m * E + c] * e[(size_t)m * E + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    const float inv = 1.f / norms[m];
    for (int c = lane; c < E; c += 64)
        dz[(size_t)m * E + c] = (T)((g[(size_t)m * E + c] - dot * e[(size_t)m * E + c]) * inv);
}

// out [N][2Ho'][2Wo'] (H x W) <- in [N][Ho][Wo]; out[2oy][2ox] = in[oy][ox], zero elsewhere; 8 channels per thread
template <typename T>
__global__ void zero_insert_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int Ho, int Wo,
                                   int C) {
    typedef typename Vec8<T>::type vec8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8n = C >> 3;
    if (i >= (long long)N * H * W * c8n) return;
    const int c8 = (int)(i % c8n);
    long long t = i / c8n;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    vec8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
        v = *(const vec8*)(in + (((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c8 * 8);
    *(vec8*)(out + (size_t)i * 8) = v;
}

// dpix[n][iy][ix][c] = mul * sum_{ky,kx,co} dz0[n][iy-ky+1][ix-kx+1][co] * w[co][ky*9 + kx*3 + c],
// dz0 = dy0 * PReLU'(y0).  A workgroup owns a 14 x 14 tile of pixels (112 = 8 x 14): each of its 256
// threads first turns ONE position of the 16 x 16 halo region into its 27 tap contributions
// t[tap][c] = sum_co dz0[co] w[co][tap][c] (dy0 / y0 are read once per position, not once per tap; the
// folded weights sit in LDS as [64][28] and are fetched by 16-B broadcast reads; the FMAs are written on
// float2 so that they issue as v_pk_fma_f32), parks them in LDS, then sums the nine neighbours of its pixel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const T* __restrict__ dy0, const T* __restrict__ y0,
                                                       const float* __restrict__ w, const float* __restrict__ alpha,
                                                       float* __restrict__ dpix, int N, int H, int W, float mul,
                                                       int nchw) {
    typedef typename Vec8<T>::type vec8;
    constexpr int TS = 14, HS = TS + 2;                    // 256 halo positions, 27 floats each (odd pitch: conflict-free)
    __shared__ float ts[HS * HS * 27];
    __shared__ __attribute__((aligned(16))) float sw[64 * 28];
    __shared__ float sa[64];
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 28; i += 256) sw[i] = (i % 28) < 27 ? w[(i / 28) * 27 + i % 28] : 0.f;
    if (tid < 64) sa[tid] = alpha[tid];
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int py = tid / HS, px = tid % HS;
    const int oy = ty * TS - 1 + py, ox = tx * TS - 1 + px;
    const bool inside = (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
    vec8 g[8], a[8];
    if (inside) {
        const size_t base = (((size_t)n * H + oy) * W + ox) * 64;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            g[c8] = *(const vec8*)(dy0 + base + c8 * 8);
            a[c8] = *(const vec8*)(y0 + base + c8 * 8);
        }
    }
    __syncthreads();
    f32x2 t[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) t[j] = f32x2{0.f, 0.f};
    if (inside) {
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) {
                const int co = c8 * 8 + j8;
                const float d = (float)g[c8][j8] * ((float)a[c8][j8] > 0.f ? 1.f : sa[co]);
                const f32x2 d2 = {d, d};
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const f32x4 w4 = *(const f32x4*)(sw + co * 28 + q * 4);
                    t[2 * q] += d2 * f32x2{w4[0], w4[1]};
                    t[2 * q + 1] += d2 * f32x2{w4[2], w4[3]};
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 27; ++j) ts[tid * 27 + j] = t[j >> 1][j & 1];
    __syncthreads();
    const int ly = tid / TS, lx = tid % TS;
    const int iy = ty * TS + ly, ix = tx * TS + lx;
    if (tid >= TS * TS || iy >= H || ix >= W) return;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // output position (iy - ky + 1, ix - kx + 1) = halo position (ly + 2 - ky, lx + 2 - kx)
            const float* tp = ts + ((ly + 2 - ky) * HS + (lx + 2 - kx)) * 27 + ky * 9 + kx * 3;
            acc[0] += tp[0];
            acc[1] += tp[1];
            acc[2] += tp[2];
        }
    const size_t i = ((size_t)n * H + iy) * W + ix;
    for (int c = 0; c < 3; ++c) {
        const size_t o = nchw ? (((size_t)n * 3 + c) * H + iy) * W + ix : i * 3 + c;
        dpix[o] = acc[c] * mul;
    }
}

}  // namespace

hipError_t launch_l2norm_bwd(int dtype, const float* g, const float* e, const float* norms, void* dz, int M, int E,
                             hipStream_t st) {
    dim3 grid((M + 3) / 4), block(256);
    if (dtype == ALINK_DT_BF16) hipLaunchKernelGGL(l2norm_bwd_kernel<__bf16>, grid, block, 0, st, g, e, norms, (__bf16*)dz, M, E);
    else hipLaunchKernelGGL(l2norm_bwd_kernel<_Float16>, grid, block, 0, st, g, e, norms, (_Float16*)dz, M, E);
    return hipGetLastError();
}

hipError_t launch_zero_insert(int dtype, const void* in, void* out, int N, int H, int W, int Ho, int Wo, int C,
                              hipStream_t st) {
    const long long tot = (long long)N * H * W * (C / 8);
    dim3 grid((unsigned)((tot + 255) / 256)), block(256);
    if (dtype == ALINK_DT_BF16)
        hipLaunchKernelGGL(zero_insert_kernel<__bf16>, grid, block, 0, st, (const __bf16*)in, (__bf16*)out, N, H, W, Ho, Wo, C);
    else
        hipLaunchKernelGGL(zero_insert_kernel<_Float16>, grid, block, 0, st, (const _Float16*)in, (_Float16*)out, N, H, W, Ho, Wo, C);
    return hipGetLastError();
}

hipError_t launch_stem_bwd(int dtype, const void* dy0, const void* y0, const float* w, const float* alpha, float* dpix,
                           int N, int H, int W, float mul, int nchw, hipStream_t st) {
    const long long tiles = (long long)N * ((H + 13) / 14) * ((W + 13) / 14);
    if (tiles <= 0 || tiles >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 grid((unsigned)tiles), block(256);
    if (dtype == ALINK_DT_BF16)
        hipLaunchKernelGGL(stem_bwd_kernel<__bf16>, grid, block, 0, st, (const __bf16*)dy0, (const __bf16*)y0, w, alpha, dpix, N, H, W, mul, nchw);
    else
        hipLaunchKernelGGL(stem_bwd_kernel<_Float16>, grid, block, 0, st, (const _Float16*)dy0, (const _Float16*)y0, w, alpha, dpix, N, H, W, mul, nchw);
    return hipGetLastError();
}

}  // namespace alink
