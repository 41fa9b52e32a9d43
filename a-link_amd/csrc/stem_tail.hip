// stem_tail.hip — the two ends of the IR-ResNet backbone that are not plain implicit GEMMs.
//
//  stem   : (x - 127.5) * 0.0078125 -> conv3x3(3->C0, pad 1, no bias) -> BN -> PReLU
//           (insightface fresnet `conv0/bn0/relu0`; executed inside model.forward at reference
//           code/face_model.py:90).  Pixels arrive as the reference holds them: float32 RGB
//           0..255, HWC (code/readDFW.py:82) or CHW after FaceModel.get_input (code/face_model.py:83).
//  finish : fixed-order reduction of the FC split-K slabs + folded bias, then the L2 row normalise
//           that the reference does on the host with sklearn (code/face_model.py:92).
#include "alink_common.h"

#include <algorithm>
#include <type_traits>

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

constexpr int STEM_ROWS = 8;   // output rows per workgroup

// One workgroup = one band of STEM_ROWS output rows of one image.  The (STEM_ROWS+2) input rows are
// normalised, converted to T and parked in LDS with a zero frame (x = -1, x = W, rows outside the
// image), so the 3x3x3 patch of a pixel is three runs of 9 consecutive LDS elements.  K = 27 is
// walked as two MFMA K-steps of 32 (rows 0, 1 with 7 zeros behind each run of 9, then row 2) per 16 pixels x 16
// channels — the K walk of the fused front kernel (front_c64.hip).  Only C0 = 64 is supported.
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void stem_kernel(const StemParams p) {
    typedef typename Vec8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = (T*)smem_raw;

    const int H = p.H, W = p.W;
    const int RP = ((W + 2) * 3 + 7) & ~7;          // LDS row pitch in elements
    const int n = blockIdx.y, y0 = blockIdx.x * STEM_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- stage + normalise the input band ------------------------------------------------------
    const int row_elems = (W + 2) * 3;
    const int total = (STEM_ROWS + 2) * row_elems;
    for (int i = tid; i < total; i += 256) {
        const int r = i / row_elems, e = i - r * row_elems;
        int ixp, c;
        if (LAYOUT == ALINK_LAYOUT_NCHW_F32) { c = e / (W + 2); ixp = e - c * (W + 2); }
        else                                 { ixp = e / 3;     c = e - ixp * 3; }
        const int iy = y0 - 1 + r, ix = ixp - 1;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            float px;
            if (LAYOUT == ALINK_LAYOUT_NHWC_F32)
                px = ((const float*)p.in)[(((size_t)n * H + iy) * W + ix) * 3 + c];
            else if (LAYOUT == ALINK_LAYOUT_NCHW_F32)
                px = ((const float*)p.in)[(((size_t)n * 3 + c) * H + iy) * W + ix];
            else
                px = (float)((const uint8_t*)p.in)[(((size_t)n * H + iy) * W + ix) * 3 + c];
            v = (px - p.sub[p.flip ? 2 - c : c]) * p.mul;
        }
        tile[r * RP + ixp * 3 + (p.flip ? 2 - c : c)] = (T)v;
    }

    // ---- per-lane constants ----------------------------------------------------------------------
    const int q = lane >> 4, lr = lane & 15;
    // A fragments (weights): tile t, row lr, K = 64 in two steps of 32, k = 8q..8q+7 of each (rows already perm64-permuted
    // on host).  K order: [ky 0: kx*3+c (9 values) + 7 zeros | ky 1: the same] then [ky 2: the same | 16 zeros] — the walk of
    // front_c64_kernel, whose image ring holds a pixel's 9-value window as one aligned record: the two kernels agree bit for bit
    vec8 wf[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int st = 0; st < 2; ++st) wf[t][st] = *(const vec8*)((const T*)p.wgt + (16 * t + lr) * 64 + st * 32 + 8 * q);
    // B fragment gather offsets: step st, k = 8q + j -> ky = 2 st + (q >> 1), e = 8 (q & 1) + j -> ky*RP + e ; e >= 9 or ky > 2 -> zero
    int koff[2][8];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ky = 2 * st + (q >> 1), e = 8 * (q & 1) + j;
            koff[st][j] = (ky < 3 && e < 9) ? ky * RP + e : -1;
        }
    const int cbase = 16 * q;                       // lane's 16 consecutive channels
    float bi[16], al[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { bi[i] = p.bias[cbase + i]; al[i] = p.alpha[cbase + i]; }

    __syncthreads();

    const int tpr = (W + 15) >> 4;                  // 16-pixel tiles per row
    const int ntiles = STEM_ROWS * tpr;
    for (int tl = wave; tl < ntiles; tl += 4) {
        const int ry = tl / tpr, xt = tl - ry * tpr;
        const int y = y0 + ry;
        const int x = xt * 16 + lr;
        const int xc = x < W ? x : W - 1;           // clamp: keep LDS reads in bounds, skip the store
        const T* base = tile + ry * RP + xc * 3;
        vec8 pf[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[st][j] = koff[st][j] >= 0 ? base[koff[st][j]] : (T)0.f;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)       // the folded-BN bias is the accumulator's start value (front_c64_kernel does the same)
            acc[t] = mfma16<T>(wf[t][0], pf[0], f32x4{bi[4 * t], bi[4 * t + 1], bi[4 * t + 2], bi[4 * t + 3]});
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16<T>(wf[t][1], pf[1], acc[t]);
        if (y < H && x < W) {
            vec8 o0, o1;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * t + j;
                    float v = acc[t][j];
                    v = v > 0.f ? v : v * al[i];
                    if (i < 8) o0[i] = (T)v; else o1[i - 8] = (T)v;
                }
            T* o = (T*)p.out + (((size_t)n * H + y) * W + x) * 64 + cbase;
            *(vec8*)o = o0;
            *(vec8*)(o + 8) = o1;
        }
    }
}

// ALINK_DT_F16X2 form of stem_kernel: normalised pixels x 2^8 as f16 pairs (hi tile, lo tile) in LDS, weights
// [64'][hi 32 | lo 32], three MFMAs per 16 x 16 tile (lo x hi, hi x lo, hi x hi), output [pixel][hi 64 | lo 64].
template <int LAYOUT>
__global__ __launch_bounds__(256) void stem_x2_kernel(const StemParams p) {
    typedef _Float16 T;
    typedef f16x8 vec8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int H = p.H, W = p.W;
    const int RP = ((W + 2) * 3 + 7) & ~7;
    T* tile = (T*)smem_raw;
    T* tile_lo = tile + (STEM_ROWS + 2) * RP;
    const int n = blockIdx.y, y0 = blockIdx.x * STEM_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const int row_elems = (W + 2) * 3;
    const int total = (STEM_ROWS + 2) * row_elems;
    for (int i = tid; i < total; i += 256) {
        const int r = i / row_elems, e = i - r * row_elems;
        int ixp, c;
        if (LAYOUT == ALINK_LAYOUT_NCHW_F32) { c = e / (W + 2); ixp = e - c * (W + 2); }
        else                                 { ixp = e / 3;     c = e - ixp * 3; }
        const int iy = y0 - 1 + r, ix = ixp - 1;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            float px;
            if (LAYOUT == ALINK_LAYOUT_NHWC_F32)
                px = ((const float*)p.in)[(((size_t)n * H + iy) * W + ix) * 3 + c];
            else if (LAYOUT == ALINK_LAYOUT_NCHW_F32)
                px = ((const float*)p.in)[(((size_t)n * 3 + c) * H + iy) * W + ix];
            else
                px = (float)((const uint8_t*)p.in)[(((size_t)n * H + iy) * W + ix) * 3 + c];
            v = (px - p.sub[p.flip ? 2 - c : c]) * p.mul * 256.f;
        }
        const T hi = (T)v;
        const int at = r * RP + ixp * 3 + (p.flip ? 2 - c : c);
        tile[at] = hi;
        tile_lo[at] = (T)(v - (float)hi);
    }

    const int q = lane >> 4, lr = lane & 15;
    vec8 wh[4], wl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        wh[t] = *(const vec8*)((const T*)p.wgt + (16 * t + lr) * 64 + 8 * q);
        wl[t] = *(const vec8*)((const T*)p.wgt + (16 * t + lr) * 64 + 32 + 8 * q);
    }
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        const int ky = (k * 57) >> 9;
        koff[j] = (k < 27) ? ky * RP + (k - 9 * ky) : -1;
    }
    const int cbase = 16 * q;
    float bi[16], al[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { bi[i] = p.bias[cbase + i] * p.bias_scale; al[i] = p.alpha[cbase + i]; }

    __syncthreads();

    const int tpr = (W + 15) >> 4;
    const int ntiles = STEM_ROWS * tpr;
    for (int tl = wave; tl < ntiles; tl += 4) {
        const int ry = tl / tpr, xt = tl - ry * tpr;
        const int y = y0 + ry;
        const int x = xt * 16 + lr;
        const int xc = x < W ? x : W - 1;
        const int b0 = ry * RP + xc * 3;
        vec8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ph[j] = koff[j] >= 0 ? tile[b0 + koff[j]] : (T)0.f;
            pl[j] = koff[j] >= 0 ? tile_lo[b0 + koff[j]] : (T)0.f;
        }
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = mfma16<T>(wh[t], pl, f32x4{0.f, 0.f, 0.f, 0.f});
            acc[t] = mfma16<T>(wl[t], ph, acc[t]);
            acc[t] = mfma16<T>(wh[t], ph, acc[t]);
        }
        if (y < H && x < W) {
            vec8 o0, o1, l0, l1;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * t + j;
                    float v = fmaf(acc[t][j], p.acc_scale, bi[i]);
                    v = v > 0.f ? v : v * al[i];
                    const T hi = (T)v;
                    const T lo = (T)(v - (float)hi);
                    if (i < 8) { o0[i] = hi; l0[i] = lo; } else { o1[i - 8] = hi; l1[i - 8] = lo; }
                }
            T* o = (T*)p.out + (((size_t)n * H + y) * W + x) * 128 + cbase;
            *(vec8*)o = o0;
            *(vec8*)(o + 8) = o1;
            *(vec8*)(o + 64) = l0;
            *(vec8*)(o + 72) = l1;
        }
    }
}

// One wave per embedding row: sum the split-K slabs in slab order (bit-reproducible), add the folded
// bias, L2-normalise with sklearn.preprocessing.normalize semantics (zero norm -> divide by 1).
__global__ __launch_bounds__(256) void fc_finish_kernel(const FcFinishParams p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int E = p.E;
    float ss = 0.f;
    // E is a multiple of 64: lane owns columns lane*4 + 256*i .. +3
    for (int c0 = lane * 4; c0 < E; c0 += 256) {
        f32x4 s = *(const f32x4*)(p.bias + c0);
        // bias first then slabs would change rounding vs "sum then bias"; keep sum-then-bias
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < p.S; ++z) a += *(const f32x4*)(p.slabs + ((size_t)z * p.M + m) * E + c0);
        a = a * p.scale + s;        // scale: an exact power of two (1 outside the split-precision mode)
        *(f32x4*)(p.out + (size_t)m * E + c0) = a;
        ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    float nrm = sqrtf(ss);
    if (p.nonfinite && lane == 0 && !(nrm <= 3.4e38f)) *p.nonfinite = 1;     // inf or NaN somewhere in the row
    if (nrm == 0.f) nrm = 1.f;
    if (p.norms && lane == 0) p.norms[m] = nrm;
    for (int c0 = lane * 4; c0 < E; c0 += 256) {
        f32x4 a = *(f32x4*)(p.out + (size_t)m * E + c0);
        a[0] /= nrm; a[1] /= nrm; a[2] /= nrm; a[3] /= nrm;
        *(f32x4*)(p.out + (size_t)m * E + c0) = a;
    }
}

// Split-K convolutions (small batches): out[m][c] = epilogue(sum_z slab[z][m][c]) with the epilogue of the conv
// kernels — folded-BN bias by border class, PReLU | PReLU' of the stored activation, residual, ReLU.
// One thread per pixel and 8 consecutive channels; slabs are added in slab order (bit-reproducible).
template <typename T>
__global__ __launch_bounds__(256) void conv_split_finish_kernel(const ConvParams p, const float* __restrict__ slabs, int S) {
    typedef typename std::conditional<std::is_same<T, __bf16>::value, bf16x8, f16x8>::type vec8;
    const int c8n = p.Cout >> 3;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.M * c8n) return;
    const int m = (int)(i / c8n), c0 = (int)(i % c8n) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int z = 0; z < S; ++z) {
        const float* sp = slabs + ((size_t)z * p.M + m) * p.Cout + c0;
        const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
    }
    int cls = 0;
    if (p.border_cls) {
        const int rem = m % (p.Ho * p.Wo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        cls = ((oy == 0) ? 0 : ((oy == p.Ho - 1) ? 2 : 1)) * 3 + ((ox == 0) ? 0 : ((ox == p.Wo - 1) ? 2 : 1));
    }
    const size_t off = (size_t)m * p.Cout + c0;
    const T* extra = (const T*)(p.dact ? p.dact : p.resid);
    vec8 e8;
    if (extra) e8 = *(const vec8*)(extra + off);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = v[j] + p.bias[(size_t)cls * p.Cout + c0 + j];
        if (p.dact) {
            x *= (float)e8[j] > 0.f ? 1.f : p.alpha[c0 + j];
        } else {
            if (p.alpha) x = x > 0.f ? x : x * p.alpha[c0 + j];
            if (p.resid) x += (float)e8[j];
        }
        if (p.post_relu) x = relu_keep_nan(x);
        v[j] = x;
    }
    vec8 o8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = (T)v[j];
    *(vec8*)((T*)p.out + off) = o8;
}

// split precision (ALINK_DT_F16X2): the slabs hold raw accumulators in units 2^(e_in + e_w); out / resid are f16 pairs
// [pixel][2 Cout] with every 64-channel chunk stored [hi 64 | lo 64]
__global__ __launch_bounds__(256) void conv_split_finish_x2_kernel(const ConvParams p, const float* __restrict__ slabs, int S) {
    typedef _Float16 T;
    const int c8n = p.Cout >> 3;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.M * c8n) return;
    const int m = (int)(i / c8n), c0 = (int)(i % c8n) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int z = 0; z < S; ++z) {
        const float* sp = slabs + ((size_t)z * p.M + m) * p.Cout + c0;
        const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
    }
    int cls = 0;
    if (p.border_cls) {
        const int rem = m % (p.Ho * p.Wo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        cls = ((oy == 0) ? 0 : ((oy == p.Ho - 1) ? 2 : 1)) * 3 + ((ox == 0) ? 0 : ((ox == p.Wo - 1) ? 2 : 1));
    }
    const size_t off = (size_t)m * (2 * p.Cout) + (size_t)(c0 >> 6) * 128 + (c0 & 63);
    f16x8 rh, rl;
    if (p.resid) {
        rh = *(const f16x8*)((const T*)p.resid + off);
        rl = *(const f16x8*)((const T*)p.resid + off + 64);
    }
    f16x8 oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = fmaf(v[j], p.acc_scale, p.bias[(size_t)cls * p.Cout + c0 + j] * p.bias_scale);
        if (p.alpha) x = x > 0.f ? x : x * p.alpha[c0 + j];
        if (p.resid) x = fmaf((float)rh[j] + (float)rl[j], p.res_scale, x);
        if (p.post_relu) x = relu_keep_nan(x);
        oh[j] = (T)x;
        ol[j] = (T)(x - (float)oh[j]);
    }
    *(f16x8*)((T*)p.out + off) = oh;
    *(f16x8*)((T*)p.out + off + 64) = ol;
}

template <typename T>
hipError_t launch_stem_t(const StemParams& p, hipStream_t stream) {
    const int RP = ((p.W + 2) * 3 + 7) & ~7;
    const size_t lds = (size_t)(STEM_ROWS + 2) * RP * sizeof(T);
    dim3 grid((p.H + STEM_ROWS - 1) / STEM_ROWS, p.N, 1), block(256, 1, 1);
    switch (p.layout) {
        case ALINK_LAYOUT_NHWC_F32:
            hipLaunchKernelGGL((stem_kernel<T, ALINK_LAYOUT_NHWC_F32>), grid, block, lds, stream, p); break;
        case ALINK_LAYOUT_NCHW_F32:
            hipLaunchKernelGGL((stem_kernel<T, ALINK_LAYOUT_NCHW_F32>), grid, block, lds, stream, p); break;
        case ALINK_LAYOUT_NHWC_U8:
            hipLaunchKernelGGL((stem_kernel<T, ALINK_LAYOUT_NHWC_U8>), grid, block, lds, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_stem(int dtype, const StemParams& p, hipStream_t stream) {
    if (p.C0 != 64 || p.N <= 0 || p.W < 2 || p.H < 1) return hipErrorInvalidValue;
    if ((size_t)(STEM_ROWS + 2) * (((p.W + 2) * 3 + 7) & ~7) * 2 > 60000) return hipErrorInvalidValue;
    if (dtype == ALINK_DT_BF16) return launch_stem_t<__bf16>(p, stream);
    if (dtype == ALINK_DT_F16) return launch_stem_t<_Float16>(p, stream);
    if (dtype == ALINK_DT_F16X2) {
        const int RP = ((p.W + 2) * 3 + 7) & ~7;
        const size_t lds = (size_t)2 * (STEM_ROWS + 2) * RP * 2;
        dim3 grid((p.H + STEM_ROWS - 1) / STEM_ROWS, p.N, 1), block(256, 1, 1);
        switch (p.layout) {
            case ALINK_LAYOUT_NHWC_F32: hipLaunchKernelGGL(stem_x2_kernel<ALINK_LAYOUT_NHWC_F32>, grid, block, lds, stream, p); break;
            case ALINK_LAYOUT_NCHW_F32: hipLaunchKernelGGL(stem_x2_kernel<ALINK_LAYOUT_NCHW_F32>, grid, block, lds, stream, p); break;
            case ALINK_LAYOUT_NHWC_U8:  hipLaunchKernelGGL(stem_x2_kernel<ALINK_LAYOUT_NHWC_U8>, grid, block, lds, stream, p); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv_split_finish(int dtype, const ConvParams& p, const float* slabs, int S, hipStream_t stream) {
    if (p.Cout % 8 || p.M <= 0 || S < 2 || !slabs) return hipErrorInvalidValue;
    const long long tot = (long long)p.M * (p.Cout / 8);
    dim3 grid((unsigned)((tot + 255) / 256), 1, 1), block(256, 1, 1);
    if (dtype == ALINK_DT_BF16) hipLaunchKernelGGL(conv_split_finish_kernel<__bf16>, grid, block, 0, stream, p, slabs, S);
    else if (dtype == ALINK_DT_F16) hipLaunchKernelGGL(conv_split_finish_kernel<_Float16>, grid, block, 0, stream, p, slabs, S);
    else if (dtype == ALINK_DT_F16X2 && !p.dact) hipLaunchKernelGGL(conv_split_finish_x2_kernel, grid, block, 0, stream, p, slabs, S);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// calibration of the split-precision mode: bits of max |x| over an f16 tensor (non-negative floats order like their
// bit patterns; an inf or NaN anywhere comes out as >= 0x7f800000).  *out must be zero before the launch.
namespace {
__global__ __launch_bounds__(256) void absmax_f16_kernel(const f16x8* __restrict__ x, size_t n8, unsigned* out) {
    float m = 0.f;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f16x8 v = x[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = fabsf((float)v[j]);
            bad = bad || !(a <= 65504.f);
            m = fmaxf(m, a);
        }
    }
    unsigned bits = bad ? 0x7fc00000u : __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(out, bits);
}
}  // namespace
hipError_t launch_absmax_f16(const void* x, size_t n, unsigned* out, hipStream_t stream) {
    if (n % 8 || !x || !out) return hipErrorInvalidValue;
    const size_t n8 = n / 8;
    const unsigned grid = (unsigned)std::min<size_t>((n8 + 255) / 256, 2048);
    hipLaunchKernelGGL(absmax_f16_kernel, dim3(grid), dim3(256), 0, stream, (const f16x8*)x, n8, out);
    return hipGetLastError();
}

hipError_t launch_fc_finish(const FcFinishParams& p, hipStream_t stream) {
    if (p.E % 64 || p.M <= 0 || p.S < 1) return hipErrorInvalidValue;
    dim3 grid((p.M + 3) / 4, 1, 1), block(256, 1, 1);
    hipLaunchKernelGGL(fc_finish_kernel, grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace alink
