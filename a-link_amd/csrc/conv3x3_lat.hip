// conv3x3_lat.hip — the LATENCY form of the 3x3 / stride 1 / pad 1 convolution: one wave per output block, no LDS, no barrier.
//
// FaceModel.get_feature (reference code/face_model.py:86-93) is a batch-1 call.  At batch 1 a stage-3 layer is 196 pixels: the
// tile kernels (conv3x3_linear.hip) put that on 4-8 workgroups, each walking all of K alone — 36 K-steps in 16 bits, 108 in split
// precision — at ~0.5 us per step, because a lone workgroup's K-step is its own in-order chain (operand reads whose latency
// nothing covers, its MFMAs, the DMA issues, the barrier: profiles/experiments/r04_deep_weight_ring_small_batches.txt).  22 us /
// 57 us per launch, 2.0 ms / 5.1 ms per image — and splitting K over workgroups (the opt-in latency mode) changes the order of
// every sum, which the exact mode cannot do: its contract is "the same bits whatever batch an image arrives in".
//
// Here each WAVE owns a small output block (16 pixels x 16 or 32 channels: 1 or 2 MFMA tiles) for the whole of K and nothing is shared:
//   * operands come STRAIGHT from global memory / L2 into the MFMA operand registers — the A fragment is 16 weight rows x 8
//     consecutive k of the packed weight tensor the tile kernels read ([cout'][chunk][tap][64], split precision
//     [chunk][hi 9 x 64 | lo 9 x 64]), the B fragment 16 pixels x 8 channels of the NHWC activation at the tap's shift (a lane
//     whose tap falls outside the image reads the zero page instead);
//   * EIGHTEEN K-sub-steps of loads are in flight (a ring of register sets: 36-54 loads of 1 KiB), the compiler's counted waits
//     order them: a sub-step costs max(its MFMAs, an L2 round trip / 18), not a round trip;
//   * no LDS, no barrier, no workgroup: 208 independent waves for a 14 x 14 x 256 layer, 392 at 28 wide, spread over the chip.
// Every output is the SAME sum in the SAME order as in conv3x3_linear (chunk, [product phase,] tap, k half; the same 8-channel
// slices on the same MFMA k positions; a border tap adds the same zeros) and the epilogue is the same sequence of float
// operations, so the result is bit-identical to the tile kernels' — asserted by every batch-1-equals-in-batch test of the suite.
// Traffic grows with the batch (each wave re-reads its weight rows and pixels through L2: ~90 MB per image and stage-3 layer),
// so the form is used for launches of at most g_lat_max_pixels output pixels (four 14 x 14 maps); above, the tile kernels.
// Measured (IR-100, one image): 2.07 -> 1.01 ms in bf16, 6.05 -> 2.91 ms in split precision, embeddings unchanged bit for bit.
#include "alink_common.h"

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

// TA x TB MFMA tiles per wave (TA: 16-channel tiles of one 32-channel block of the packed weights, TB: 16-pixel tiles); DEPTH
// K-sub-steps of operand loads in flight (a ring of register sets; 18 sub-steps per chunk and phase, so 6, 9 and 18 divide every
// walk).  A wave's step costs max(its TA x TB MFMAs, the L2 round trip / DEPTH): small blocks and a deep ring for a lone image.
// SP: split precision (T = _Float16): NP product phases per chunk — 3 (hi x W_hi, hi x W_lo, lo x W_hi) or 1 (the screening form)
template <typename T, bool SP, int NP, int TA, int TB, int DEPTH>
__global__ __launch_bounds__(64) void conv3x3_lat_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    const int lane = threadIdx.x;
    const int q = lane >> 4, lr = lane & 15;
    const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
    const int CinP = SP ? 2 * Cin : Cin;
    const int K = 9 * CinP;                                   // weight row pitch
    const int ncc = Cin >> 6;
    const int ncb = Cout / (16 * TA);                         // channel blocks of this launch
    const int cb = (int)blockIdx.x % ncb, bp = (int)blockIdx.x / ncb;
    const long long totpix = (long long)p.N * H * W;

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- this lane's pixels (tile u, column lr), their image coordinates and the taps that stay inside the image ----
    long long pix[TB];
    bool okp[TB];
    unsigned inside[TB];
    int py[TB], px[TB], xoff[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
        pix[u] = (long long)bp * (16 * TB) + 16 * u + lr;
        okp[u] = pix[u] < totpix;
        const long long pc = okp[u] ? pix[u] : 0;
        const int rem = (int)(pc % ((long long)H * W));
        py[u] = rem / W;
        px[u] = rem - py[u] * W;
        unsigned m = 0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                if (okp[u] && (unsigned)(py[u] + ky - 1) < (unsigned)H && (unsigned)(px[u] + kx - 1) < (unsigned)W) m |= 1u << (ky * 3 + kx);
        inside[u] = m;
        pix[u] = pc;
        xoff[u] = (int)(pc * CinP) + q * 8;                   // element offsets fit 32 bits: the launcher checks
    }
    // packed weight rows: 32-channel block b holds rows 32 b + 16 t + 4 q' + j  <->  channel 32 b + 8 q' + 4 t + j
    const int row0 = TA == 2 ? cb * 32 : (cb >> 1) * 32 + (cb & 1) * 16;
    const int woff = (row0 + lr) * K + q * 8;

    // ---- the walk: S = ((cc * NPH + ph) * 9 + tap) * 2 + ks, the tile kernels' order --------------------------------------------
    constexpr int NPH = SP ? NP : 1;
    const int total = ncc * NPH * 18;
    struct Frag { vec8 a[TA], b[TB]; };
    auto fetch = [&](int S) -> Frag {
        const int ks = S & 1, t9 = S >> 1;
        const int tap = t9 % 9, cp = t9 / 9;
        const int ph = SP ? cp % NPH : 0, cc = SP ? cp / NPH : cp;
        const int ky = tap / 3, kx = tap - 3 * ky;
        // weights: 16-bit [chunk][tap][64]; split precision [chunk][hi: 9 x 64 | lo: 9 x 64], W_lo in phase 1
        const int wk = SP ? (cc * 18 + (ph == 1 ? 9 : 0) + tap) * 64 + ks * 32 : (cc * 9 + tap) * 64 + ks * 32;
        // pixels: the chunk's 64 channels (split precision: its hi half, or its lo half in phase 2) at the tap's shift
        const int xk = ((ky - 1) * W + (kx - 1)) * CinP + (SP ? (2 * cc + (ph == 2 ? 1 : 0)) * 64 : cc * 64) + ks * 32;
        Frag f;
#pragma unroll
        for (int t = 0; t < TA; ++t) f.a[t] = *(const vec8*)(gw + (woff + 16 * t * K + wk));
#pragma unroll
        for (int u = 0; u < TB; ++u) f.b[u] = *(const vec8*)(((inside[u] >> tap) & 1u) ? gin + (xoff[u] + xk) : gz + q * 8);
        return f;
    };

    f32x4 acc[TA][TB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    Frag ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = fetch(d);       // total >= 36 > DEPTH
    for (int S0 = 0; S0 < total; S0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const Frag f = ring[d];
            if (S0 + d + DEPTH < total) ring[d] = fetch(S0 + d + DEPTH);
#pragma unroll
            for (int t = 0; t < TA; ++t)
#pragma unroll
                for (int u = 0; u < TB; ++u) acc[t][u] = mfma16<T>(f.a[t], f.b[u], acc[t][u]);
        }
    }

    // ---- epilogue: lane (q, lr) holds, for pixel u, channels c0 .. c0 + 4 TA - 1 (consecutive).  The float operations and their
    // order are conv3x3_linear's (bias by border class, PReLU, residual, ReLU, conversion). ----------------------------------------
    constexpr int CPL = 4 * TA;
    typedef T vecC __attribute__((ext_vector_type(CPL)));
    const int c0 = TA == 2 ? cb * 32 + 8 * q : (cb >> 1) * 32 + 8 * q + 4 * (cb & 1);
#pragma unroll
    for (int u = 0; u < TB; ++u) {
        if (!okp[u]) continue;
        const int rc = py[u] == 0 ? 0 : (py[u] == H - 1 ? 2 : 1);
        const int ccl = px[u] == 0 ? 0 : (px[u] == W - 1 ? 2 : 1);
        const int cls = p.border_cls ? rc * 3 + ccl : 0;
        float v[CPL];
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = p.bias[cls * Cout + c0 + 4 * t + j];
                v[4 * t + j] = SP ? fmaf(acc[t][u][j], p.acc_scale, b * p.bias_scale) : acc[t][u][j] + b;
            }
        if (p.alpha) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.alpha[c0 + i];
        }
        if constexpr (SP) {
            const size_t off = (size_t)pix[u] * (2 * Cout) + (size_t)(c0 >> 6) * 128 + (c0 & 63);
            if (p.resid) {
                const T* r = (const T*)p.resid;
                const vecC rh = *(const vecC*)(r + off), rl = *(const vecC*)(r + off + 64);
#pragma unroll
                for (int i = 0; i < CPL; ++i) v[i] = fmaf((float)rh[i], p.res_scale, v[i]);
#pragma unroll
                for (int i = 0; i < CPL; ++i) v[i] = fmaf((float)rl[i], p.res_scale, v[i]);
            }
            vecC o8, l8;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                float x = v[i];
                if (p.post_relu) x = relu_keep_nan(x);
                o8[i] = (T)x;
                l8[i] = (T)(x - (float)o8[i]);
            }
            *(vecC*)((T*)p.out + off) = o8;
            *(vecC*)((T*)p.out + off + 64) = l8;
        } else {
            const size_t off = (size_t)pix[u] * Cout + c0;
            if (p.resid) {
                const vecC r8 = *(const vecC*)((const T*)p.resid + off);
#pragma unroll
                for (int i = 0; i < CPL; ++i) v[i] += (float)r8[i];
            }
            vecC o8;
#pragma unroll
            for (int i = 0; i < CPL; ++i) o8[i] = (T)(p.post_relu ? relu_keep_nan(v[i]) : v[i]);
            *(vecC*)((T*)p.out + off) = o8;
        }
    }
}

// launches of at most this many output pixels take the latency form (0 = never).  784 = four 14 x 14 maps, one 28 x 28 map.
int g_lat_max_pixels = 784;
int g_lat_form = -1;              // A/B (alink_debug_set_latency_tiles): -1 = by size (below); 0 = 1x1 tiles, ring 18; 1 = 2x1, 18; 2 = 2x2, 9; 3 = 2x2, 6

}  // namespace

extern "C" void alink_debug_set_latency_form(int max_pixels) { g_lat_max_pixels = max_pixels; }
extern "C" void alink_debug_set_latency_tiles(int form) { g_lat_form = form; }

bool conv3x3_lat_applies(int dtype, const ConvParams& p) {
    if (g_lat_max_pixels <= 0 || p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.splitk != 1 || p.dact || p.stamps || p.in2) return false;
    if (p.Cin % 64 || p.Cout % 32 || p.Cin < 128) return false;                 // the 64-channel layers are HBM-shaped, not K walks
    if ((long long)p.N * p.H * p.W > g_lat_max_pixels) return false;
    if ((long long)p.N * p.H * p.W * p.Cin * (dtype == ALINK_DT_F16X2 ? 2 : 1) >= (1ll << 31)) return false;
    if ((long long)p.Cout * 9 * p.Cin * (dtype == ALINK_DT_F16X2 ? 2 : 1) >= (1ll << 31)) return false;
    return dtype == ALINK_DT_BF16 || dtype == ALINK_DT_F16 || dtype == ALINK_DT_F16X2;
}

namespace {
template <int TA, int TB, int D>
hipError_t launch_form(int dtype, const ConvParams& p, hipStream_t st) {
    const long long totpix = (long long)p.N * p.H * p.W;
    const unsigned grid = (unsigned)(((totpix + 16 * TB - 1) / (16 * TB)) * (p.Cout / (16 * TA)));
    if (dtype == ALINK_DT_BF16)       hipLaunchKernelGGL((conv3x3_lat_kernel<__bf16, false, 1, TA, TB, D>), dim3(grid), dim3(64), 0, st, p);
    else if (dtype == ALINK_DT_F16)   hipLaunchKernelGGL((conv3x3_lat_kernel<_Float16, false, 1, TA, TB, D>), dim3(grid), dim3(64), 0, st, p);
    else if (p.nprod == 1)            hipLaunchKernelGGL((conv3x3_lat_kernel<_Float16, true, 1, TA, TB, D>), dim3(grid), dim3(64), 0, st, p);
    else                              hipLaunchKernelGGL((conv3x3_lat_kernel<_Float16, true, 3, TA, TB, D>), dim3(grid), dim3(64), 0, st, p);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_conv3x3_lat(int dtype, const ConvParams& p, hipStream_t st) {
    // By size: 16 x 16 blocks (one MFMA per sub-step, 36 loads in flight) while that makes at most 448 waves — a lone 14 x 14 x
    // 256 map is 208 of them, each bounded by its L2 round trips / 18 —, 32-channel blocks above (half the pixel traffic per MFMA).
    // Measured, IR-100, forward of n images in bf16 / split precision (tile kernels 2.07 / 6.05 ms whatever n <= 4):
    //   n = 1: 1.01 / 2.91 ms with 16 x 16 blocks, 1.13 / 3.20 with 32-channel blocks;  n = 4: 1.78 / 4.57 against 1.64 / 4.20.
    int form = g_lat_form;
    if (form < 0) form = ((long long)p.N * p.H * p.W + 15) / 16 * (p.Cout / 16) <= 448 ? 0 : 1;
    switch (form) {
        case 1: return launch_form<2, 1, 18>(dtype, p, st);
        case 2: return launch_form<2, 2, 9>(dtype, p, st);
        case 3: return launch_form<2, 2, 6>(dtype, p, st);
        default: return launch_form<1, 1, 18>(dtype, p, st);
    }
}

}  // namespace alink
