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
//   * EIGHTEEN K-sub-steps of loads are in flight (a ring of register sets: 36-72 loads of 1 KiB), the compiler's counted waits
//     order them, and a sub-step is nothing but its loads and MFMAs (addresses: a VGPR set up once + an SGPR per 18 sub-steps + the
//     instruction's immediate; zero padding by the buffer descriptor's range check);
//   * no LDS, no barrier, no workgroup: 208 independent waves for a 14 x 14 x 256 layer, 392 at 28 wide, spread over the chip.
// Every output is the SAME sum in the SAME order as in conv3x3_linear (chunk, [product phase,] tap, k half; the same 8-channel
// slices on the same MFMA k positions; a border tap adds the same zeros) and the epilogue is the same sequence of float
// operations, so the result is bit-identical to the tile kernels' — asserted by every batch-1-equals-in-batch test of the suite.
// Traffic grows with the batch (each wave re-reads its weight rows and pixels through L2: ~90 MB per image and stage-3 layer),
// so the form is used for launches of at most g_lat_max_pixels output pixels (eight 14 x 14 maps); above, the tile kernels.
// Measured (IR-100, one image): 2.07 -> 1.04 ms in bf16, 6.05 -> 2.28 ms in split precision, embeddings unchanged bit for bit.
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

template <int A>
struct IC { static constexpr int a = A; };

// TA x TB MFMA tiles per wave (TA: 16-channel tiles of one 32-channel block of the packed weights, TB: 16-pixel tiles).
// SP: split precision (T = _Float16): NP product phases per chunk — 3 (hi x W_hi, hi x W_lo, lo x W_hi) or 1 (the screening form).
//
// The walk is the tile kernels': for every chunk, [phase,] tap, k half.  One ITERATION = the 18 sub-steps (tap, k half) of a
// (chunk, phase); its operand loads are issued one whole iteration ahead into a ring of 18 register sets, so 18 sub-steps (36-54
// loads of 1 KiB) are always in flight.  A lone wave per SIMD issues in order, so what a sub-step COSTS is its instructions: the
// first form of this kernel computed every address in vector registers (~20 VALU per sub-step = ~80 cycles against one 16-cycle
// MFMA).  Here a sub-step is TA + TB buffer loads and TA x TB MFMAs and nothing else:
//   * weights: one buffer descriptor over the packed tensor, the lane's row offset in a VGPR for the whole kernel, the iteration's
//     (chunk, phase) offset in an SGPR, the sub-step's (tap, k half) in the instruction's immediate;
//   * pixels: one descriptor PER TAP whose base is the activation shifted by the tap ((ky - 1) W + kx - 1 pixels), the lane's pixel
//     offset in a VGPR per tap — or an offset beyond the descriptor's range where the tap leaves the image: the load then returns
//     zeros without touching memory (the zero padding) —, the iteration's chunk / half offset in an SGPR, the k half in the immediate.
template <typename T, bool SP, int NP, int TA, int TB, int AHEAD = 1>
__global__ __launch_bounds__(64) void conv3x3_lat_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr unsigned RSRC3 = 0x00020000u;                   // raw buffer, 32-bit data format (gfx9 family)
    constexpr unsigned OOB = 0x80000000u;                      // beyond every descriptor's num_records (< 2^31), and no immediate wraps it
    const int lane = threadIdx.x;
    const int q = lane >> 4, lr = lane & 15;
    const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
    const int CinP = SP ? 2 * Cin : Cin;
    const int K = 9 * CinP;                                   // weight row pitch
    const int ncc = Cin >> 6;
    const int ncb = Cout / (16 * TA);                         // channel blocks of this launch
    const int cb = (int)blockIdx.x % ncb, bp = (int)blockIdx.x / ncb;
    const long long totpix = (long long)p.N * H * W;
    const unsigned xbytes = (unsigned)(totpix * CinP * 2), wbytes = (unsigned)((long long)Cout * K * 2);   // < 2^31: the launcher checks

    // ---- this lane's pixels (tile u, column lr), their image coordinates, and per tap the byte offset of its 8-channel slice
    // or OOB where the tap leaves the image ----
    long long pix[TB];
    bool okp[TB];
    int py[TB], px[TB];
    unsigned vox[TB][9];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
        pix[u] = (long long)bp * (16 * TB) + 16 * u + lr;
        okp[u] = pix[u] < totpix;
        const long long pc = okp[u] ? pix[u] : 0;
        const int rem = (int)(pc % ((long long)H * W));
        py[u] = rem / W;
        px[u] = rem - py[u] * W;
        pix[u] = pc;
        const unsigned off = (unsigned)((pc * CinP + q * 8) * 2);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const bool in = okp[u] && (unsigned)(py[u] + ky - 1) < (unsigned)H && (unsigned)(px[u] + kx - 1) < (unsigned)W;
            vox[u][tap] = in ? off : OOB;
        }
    }
    // packed weight rows: 32-channel block b holds rows 32 b + 16 t + 4 q' + j  <->  channel 32 b + 8 q' + 4 t + j
    const int row0 = TA == 2 ? cb * 32 : (cb >> 1) * 32 + (cb & 1) * 16;
    unsigned vow[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) vow[t] = (unsigned)(((row0 + 16 * t + lr) * K + q * 8) * 2);

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, (int)wbytes, RSRC3);
    __amdgpu_buffer_rsrc_t rx[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const long long shift = ((long long)(tap / 3 - 1) * W + (tap % 3 - 1)) * CinP * 2;
        rx[tap] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + shift), 0, (int)xbytes, RSRC3);
    }

    constexpr int NPH = SP ? NP : 1;
    const int iters = ncc * NPH;
    // scalar byte offsets of iteration `it`: weights [chunk][tap][64] (split precision [chunk][hi 9 x 64 | lo 9 x 64], W_lo in
    // phase 1); pixels: the chunk's 64 channels (split precision: its hi half, its lo half in phase 2)
    auto s_w = [&](int it) { const int cc = it / NPH, ph = it - cc * NPH; return (SP ? (cc * 18 + (ph == 1 ? 9 : 0)) : cc * 9) * 128; };
    auto s_x = [&](int it) { const int cc = it / NPH, ph = it - cc * NPH; return (SP ? 2 * cc + (ph == 2 ? 1 : 0) : cc) * 128; };

    struct Frag { vec8 a[TA], b[TB]; };
    auto fetch = [&](int sub, int sw, int sx) -> Frag {       // sub = tap * 2 + ks: a compile-time constant after unrolling
        const int tap = sub >> 1, ks = sub & 1;
        Frag f;
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, vow[t] + (unsigned)(tap * 128 + ks * 64), sw, 0);
            __builtin_memcpy(&f.a[t], &v, 16);
        }
#pragma unroll
        for (int u = 0; u < TB; ++u) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx[tap], vox[u][tap] + (unsigned)(ks * 64), sx, 0);
            __builtin_memcpy(&f.b[u], &v, 16);
        }
        return f;
    };

    f32x4 acc[TA][TB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    // AHEAD whole iterations of loads in flight: 18 x AHEAD sub-steps (AHEAD = 2 where the registers allow: a layer's weights come
    // from DRAM once per forward and 18 round trips in flight do not cover a 16-cycle MFMA step)
    Frag ring[18 * AHEAD];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
        if (a < iters) {
            const int sw = s_w(a), sx = s_x(a);
#pragma unroll
            for (int d = 0; d < 18; ++d) ring[a * 18 + d] = fetch(d, sw, sx);
        }
    }
    for (int it0 = 0; it0 < iters; it0 += AHEAD) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const int it = it0 + a;
            if (it < iters) {
                const bool more = it + AHEAD < iters;
                const int sw = more ? s_w(it + AHEAD) : 0, sx = more ? s_x(it + AHEAD) : 0;
#pragma unroll
                for (int d = 0; d < 18; ++d) {
                    const Frag f = ring[a * 18 + d];
                    if (more) ring[a * 18 + d] = fetch(d, sw, sx);
#pragma unroll
                    for (int t = 0; t < TA; ++t)
#pragma unroll
                        for (int u = 0; u < TB; ++u) acc[t][u] = mfma16<T>(f.a[t], f.b[u], acc[t][u]);
                }
            }
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

// ---- the same form for the implicit-GEMM layers of a lone image: the stride-2 3x3 convolution of a stage's first unit (16-bit
// modes: with the unit's 1x1 projection shortcut as extra K-steps from a second input) and the stand-alone 1x1 stride-2 shortcut
// of split precision.  conv_igemm's walk — tap-major, chunk inside the tap, per real step the products X_lo W_hi, X_hi W_hi,
// X_hi W_lo (split precision), k halves inside; the shortcut's chunks after the last tap — and its epilogue, so the results are
// bit-identical to conv_igemm_kernel's.  CPT (Cin / 64) is a template parameter: the whole walk unrolls, every offset is an SGPR
// constant or an immediate.  Weights [cout'][tap][chunk][64] (split precision [tap][chunk][hi 64 | lo 64]), rows permuted per
// 64-channel block: row 64 b + 16 t + 4 q' + j  <->  channel 64 b + 16 q' + 4 t + j.
template <typename T, bool SP, int NP, int CPT, int KSZ, int CPT2>
__global__ __launch_bounds__(64) void conv_gemm_lat_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr unsigned RSRC3 = 0x00020000u, OOB = 0x80000000u;
    constexpr int NTAP = KSZ * KSZ, NPH = SP ? NP : 1;
    constexpr int XE = SP ? 2 : 1;                             // elements per value (hi | lo)
    const int lane = threadIdx.x;
    const int q = lane >> 4, lr = lane & 15;
    const int H = p.H, W = p.W, Cout = p.Cout;
    constexpr int Cin = CPT * 64, CinP = XE * Cin, Cin2 = CPT2 * 64;
    constexpr int K = XE * NTAP * Cin + Cin2;                  // weight row pitch
    const int ncb = Cout >> 4;
    const int cb = (int)blockIdx.x % ncb, bp = (int)blockIdx.x / ncb;
    const int HoWo = p.Ho * p.Wo;

    // this lane's output pixel and the byte offset of its window's first input pixel, biased by PAD rows + PAD pixels so that it is
    // never negative (the descriptor's base is lowered by the same amount); OOB where a tap leaves the image
    const int m = bp * 16 + lr;
    const bool okp = m < p.M;
    const int mc = okp ? m : 0;
    const int n = mc / HoWo, rem = mc - n * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    const long long bias_px = (long long)p.pad * W + p.pad;
    const unsigned xoff = (unsigned)((((long long)(n * H + iy0) * W + ix0 + bias_px) * CinP + q * 8) * 2);
    unsigned vox[NTAP];
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
        const int ky = tap / KSZ, kx = tap % KSZ;
        vox[tap] = (okp && (unsigned)(iy0 + ky) < (unsigned)H && (unsigned)(ix0 + kx) < (unsigned)W) ? xoff : OOB;
    }
    const unsigned vox2 = okp ? (unsigned)((((long long)(p.in2_compact ? mc : (n * H + oy * p.stride) * W + ox * p.stride)) * Cin2 + q * 8) * 2) : OOB;
    const int t16 = cb & 3, b64 = cb >> 2;
    const unsigned vow = (unsigned)(((b64 * 64 + 16 * t16 + lr) * K + q * 8) * 2);

    const unsigned xbytes = (unsigned)((long long)p.N * H * W * CinP * 2), wbytes = (unsigned)((long long)Cout * K * 2);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, (int)wbytes, RSRC3);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in - bias_px * CinP * 2), 0,
                                                                        (int)(xbytes + (unsigned)(bias_px * CinP * 2)), RSRC3);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(CPT2 ? p.in2 : p.in), 0,
                                                                         CPT2 ? (int)((long long)(p.in2_compact ? p.M : p.N * H * W) * Cin2 * 2) : 0, RSRC3);
    const int tap_bytes = 0;
    (void)tap_bytes;

    // the walk, flattened and fully static: S = ((tap * CPT + cc) * NPH + ph) * 2 + ks, then the shortcut's (cc2, ks)
    constexpr int MAIN = NTAP * CPT * NPH * 2, TOTAL = MAIN + CPT2 * 2;
    constexpr int DEPTH = TOTAL < 24 ? TOTAL : 24;
    struct Frag { vec8 a, b; };
    auto fetch = [&](auto sc) -> Frag {
        constexpr int S = decltype(sc)::a;
        Frag f;
        u32x4 va, vb;
        if constexpr (S < MAIN) {
            constexpr int ks = S & 1, r = S >> 1, ph = r % NPH, step = r / NPH, cc = step % CPT, tap = step / CPT;
            constexpr int ky = tap / KSZ, kx = tap % KSZ;
            // split precision, conv_igemm's product order: 0 = X_lo W_hi, 1 = X_hi W_hi, 2 = X_hi W_lo; the screening form: X_hi W_hi
            constexpr int xpart = SP ? ((NP == 3 && ph == 0) ? 1 : 0) : 0, wblk = SP ? ((NP == 3 && ph == 2) ? 1 : 0) : 0;
            constexpr int wk = SP ? ((tap * CPT + cc) * 2 + wblk) * 128 + ks * 64 : (tap * CPT + cc) * 128 + ks * 64;
            constexpr int xk = (SP ? (2 * cc + xpart) : cc) * 128 + ks * 64;
            const int xs = (ky * W + kx) * CinP * 2 + xk;                        // uniform: an SGPR
            va = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, wk, 0);
            vb = __builtin_amdgcn_raw_buffer_load_b128(rx, vox[tap], xs, 0);
        } else {
            constexpr int S2 = S - MAIN, ks = S2 & 1, cc = S2 >> 1;
            va = __builtin_amdgcn_raw_buffer_load_b128(rw, vow, (XE * NTAP * Cin + cc * 64) * 2 + ks * 64, 0);
            vb = __builtin_amdgcn_raw_buffer_load_b128(rx2, vox2, cc * 128 + ks * 64, 0);
        }
        __builtin_memcpy(&f.a, &va, 16);
        __builtin_memcpy(&f.b, &vb, 16);
        return f;
    };

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag ring[DEPTH];
    auto prologue = [&](auto self, auto dc) {
        constexpr int d = decltype(dc)::a;
        if constexpr (d < DEPTH) {
            ring[d] = fetch(IC<d>{});
            self(self, IC<d + 1>{});
        }
    };
    prologue(prologue, IC<0>{});
    auto body = [&](auto self, auto sc) {
        constexpr int S = decltype(sc)::a;
        if constexpr (S < TOTAL) {
            const Frag f = ring[S % DEPTH];
            if constexpr (S + DEPTH < TOTAL) ring[S % DEPTH] = fetch(IC<S + DEPTH>{});
            acc = mfma16<T>(f.a, f.b, acc);
            self(self, IC<S + 1>{});
        }
    };
    body(body, IC<0>{});

    // ---- conv_igemm's epilogue for the lane's 4 channels of pixel m ------------------------------------------------------------
    if (!okp) return;
    const int c0 = b64 * 64 + 16 * q + 4 * t16;
    int cls = 0;
    if (p.border_cls) {
        const int rc = (oy == 0) ? 0 : ((oy == p.Ho - 1) ? 2 : 1);
        const int cc = (ox == 0) ? 0 : ((ox == p.Wo - 1) ? 2 : 1);
        cls = rc * 3 + cc;
    }
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const size_t off = SP ? (size_t)mc * (2 * Cout) + (size_t)(c0 >> 6) * 128 + (c0 & 63) : (size_t)mc * Cout + c0;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float b = p.bias[cls * Cout + c0 + j];
        v[j] = SP ? fmaf(acc[j], p.acc_scale, b * p.bias_scale) : acc[j] + b;
    }
    if (p.alpha) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * p.alpha[c0 + j];
    }
    if (p.resid) {
        const T* r = (const T*)p.resid;
        const vec4 rh = *(const vec4*)(r + off);
        if constexpr (SP) {
            const vec4 rl = *(const vec4*)(r + off + 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf((float)rh[j] + (float)rl[j], p.res_scale, v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)rh[j];
        }
    }
    if (p.post_relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = relu_keep_nan(v[j]);
    }
    vec4 o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = (T)v[j];
    *(vec4*)((T*)p.out + off) = o4;
    if constexpr (SP) {
        vec4 l4;
#pragma unroll
        for (int j = 0; j < 4; ++j) l4[j] = (T)(v[j] - (float)o4[j]);
        *(vec4*)((T*)p.out + off + 64) = l4;
    }
}

// launches of at most this many output pixels — and at most 448 waves in the 32 x 32 form — take the latency form (0 = never).
// 1600 = eight 14 x 14 maps, two 28 x 28 maps (sweep: profiles/r04g_small_batch_latency.txt; 3200 loses at 16 images).
int g_lat_max_pixels = 1600;
int g_lat_form = -1;              // A/B (alink_debug_set_latency_tiles): -1 = by size (below); 0 = 1 x 1 MFMA tiles per wave; 1 = 2 x 1; 2 = 2 x 2

}  // namespace

extern "C" void alink_debug_set_latency_form(int max_pixels) { g_lat_max_pixels = max_pixels; }
extern "C" void alink_debug_set_latency_tiles(int form) { g_lat_form = form; }

bool conv3x3_lat_applies(int dtype, const ConvParams& p) {
    if (g_lat_max_pixels <= 0 || p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.splitk != 1 || p.dact || p.stamps || p.in2) return false;
    if (p.Cin % 64 || p.Cout % 32 || p.Cin < 128) return false;                 // the 64-channel layers are HBM-shaped, not K walks
    if ((long long)p.N * p.H * p.W > g_lat_max_pixels) return false;
    if (((long long)p.N * p.H * p.W + 31) / 32 * (p.Cout / 32) > 448) return false;     // more waves than that: the tile kernels' reuse wins
    if ((long long)p.N * p.H * p.W * p.Cin * (dtype == ALINK_DT_F16X2 ? 4 : 2) >= (1ll << 31)) return false;       // byte offsets in 31 bits
    if ((long long)p.Cout * 9 * p.Cin * (dtype == ALINK_DT_F16X2 ? 4 : 2) >= (1ll << 31)) return false;
    return dtype == ALINK_DT_BF16 || dtype == ALINK_DT_F16 || dtype == ALINK_DT_F16X2;
}

namespace {
template <int TA, int TB, int AHEAD = 1>
hipError_t launch_form(int dtype, const ConvParams& p, hipStream_t st) {
    const long long totpix = (long long)p.N * p.H * p.W;
    const unsigned grid = (unsigned)(((totpix + 16 * TB - 1) / (16 * TB)) * (p.Cout / (16 * TA)));
    if (dtype == ALINK_DT_BF16)       hipLaunchKernelGGL((conv3x3_lat_kernel<__bf16, false, 1, TA, TB, AHEAD>), dim3(grid), dim3(64), 0, st, p);
    else if (dtype == ALINK_DT_F16)   hipLaunchKernelGGL((conv3x3_lat_kernel<_Float16, false, 1, TA, TB, AHEAD>), dim3(grid), dim3(64), 0, st, p);
    else if (p.nprod == 1)            hipLaunchKernelGGL((conv3x3_lat_kernel<_Float16, true, 1, TA, TB, AHEAD>), dim3(grid), dim3(64), 0, st, p);
    else                              hipLaunchKernelGGL((conv3x3_lat_kernel<_Float16, true, 3, TA, TB, AHEAD>), dim3(grid), dim3(64), 0, st, p);
    return hipGetLastError();
}
}  // namespace

bool conv_gemm_lat_applies(int dtype, const ConvParams& p) {
    if (g_lat_max_pixels <= 0 || p.splitk != 1 || p.dact || p.stamps) return false;
    const bool c3 = p.ksz == 3 && p.stride == 2 && p.pad == 1, c1 = p.ksz == 1 && p.pad == 0 && !p.in2;
    if (!c3 && !c1) return false;
    const int cpt = p.Cin / 64, cpt2 = p.in2 ? p.Cin2 / 64 : 0;
    if (p.Cin % 64 || p.Cout % 64 || (cpt != 1 && cpt != 2 && cpt != 4 && cpt != 8)) return false;
    if (p.in2 && (dtype == ALINK_DT_F16X2 || cpt2 != cpt / 2 || cpt < 2)) return false;        // the fused shortcut of a stage's first unit
    if (p.M > 784 || (long long)((p.M + 15) / 16) * (p.Cout / 16) > 512) return false;           // a lone image's stride-2 layers
    const long long two = dtype == ALINK_DT_F16X2 ? 4 : 2;
    if ((long long)p.N * p.H * p.W * p.Cin * two + ((long long)p.pad * p.W + p.pad) * p.Cin * two >= (1ll << 31)) return false;
    if ((long long)p.Cout * (p.ksz * p.ksz * p.Cin * (two / 2) + (p.in2 ? p.Cin2 : 0)) * 2 >= (1ll << 31)) return false;
    return dtype == ALINK_DT_BF16 || dtype == ALINK_DT_F16 || dtype == ALINK_DT_F16X2;
}

namespace {
template <int CPT, int KSZ, int CPT2>
hipError_t launch_gemm_form(int dtype, const ConvParams& p, hipStream_t st) {
    const unsigned grid = (unsigned)(((p.M + 15) / 16) * (p.Cout / 16));
    if (dtype == ALINK_DT_BF16)       hipLaunchKernelGGL((conv_gemm_lat_kernel<__bf16, false, 1, CPT, KSZ, CPT2>), dim3(grid), dim3(64), 0, st, p);
    else if (dtype == ALINK_DT_F16)   hipLaunchKernelGGL((conv_gemm_lat_kernel<_Float16, false, 1, CPT, KSZ, CPT2>), dim3(grid), dim3(64), 0, st, p);
    else if constexpr (CPT2 == 0) {
        if (p.nprod == 1)             hipLaunchKernelGGL((conv_gemm_lat_kernel<_Float16, true, 1, CPT, KSZ, 0>), dim3(grid), dim3(64), 0, st, p);
        else                          hipLaunchKernelGGL((conv_gemm_lat_kernel<_Float16, true, 3, CPT, KSZ, 0>), dim3(grid), dim3(64), 0, st, p);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
}  // namespace

hipError_t launch_conv_gemm_lat(int dtype, const ConvParams& p, hipStream_t st) {
    const int cpt = p.Cin / 64;
    if (p.ksz == 3) {
        if (p.in2) {
            switch (cpt) {
                case 2: return launch_gemm_form<2, 3, 1>(dtype, p, st);
                case 4: return launch_gemm_form<4, 3, 2>(dtype, p, st);
                case 8: return launch_gemm_form<8, 3, 4>(dtype, p, st);
            }
            return hipErrorInvalidValue;
        }
        switch (cpt) {
            case 1: return launch_gemm_form<1, 3, 0>(dtype, p, st);
            case 2: return launch_gemm_form<2, 3, 0>(dtype, p, st);
            case 4: return launch_gemm_form<4, 3, 0>(dtype, p, st);
            case 8: return launch_gemm_form<8, 3, 0>(dtype, p, st);
        }
        return hipErrorInvalidValue;
    }
    switch (cpt) {
        case 1: return launch_gemm_form<1, 1, 0>(dtype, p, st);
        case 2: return launch_gemm_form<2, 1, 0>(dtype, p, st);
        case 4: return launch_gemm_form<4, 1, 0>(dtype, p, st);
        case 8: return launch_gemm_form<8, 1, 0>(dtype, p, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv3x3_lat(int dtype, const ConvParams& p, hipStream_t st) {
    // By size: 16 x 16 blocks (one MFMA per sub-step) while that makes at most 448 waves — a lone 14 x 14 x 256 map is 208 —, 32 x 32
    // blocks above (half the loads per MFMA).  Measured, IR-100, forward of n images in bf16 / split precision (tile kernels: 2.07-2.10 /
    // 6.02-6.09 ms for n <= 4): n = 1: 1.04 / 2.28 ms with 16 x 16 blocks, 1.51 / 3.39 with 32 x 32; n = 2: 1.38 / 3.15 against 1.59 /
    // 3.74; n = 4: 1.77 / 4.48 against 1.60 / 3.82.
    int form = g_lat_form;
    if (form < 0) form = ((long long)p.N * p.H * p.W + 15) / 16 * (p.Cout / 16) <= 448 ? 0 : 2;
    switch (form) {
        case 1: return launch_form<2, 1>(dtype, p, st);
        case 2: return launch_form<2, 2>(dtype, p, st);
        case 4: return launch_form<1, 1, 1>(dtype, p, st);
        default: return launch_form<1, 1, 2>(dtype, p, st);
    }
}

}  // namespace alink
