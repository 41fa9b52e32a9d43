// conv_igemm.hip — NHWC implicit-GEMM convolution on the gfx950 matrix cores.
//
// This is the kernel that carries 99.8 % of the backbone FLOPs (the device
// work behind `self.model.forward(db, is_train=False)` at reference code/face_model.py:90).
//
//   out[m][co] = epilogue( sum_{ky,kx,ci} in[n][oy*s+ky-pad][ox*s+kx-pad][ci] * w[co][ky][kx][ci] )
//   m = (n*Ho + oy)*Wo + ox,   GEMM:  M = N*Ho*Wo pixels,  Ncols = Cout,  K = ksz*ksz*Cin
//
// Design (CDNA4):
//   * 256-thread workgroup = 4 waves; every wave owns a 64-pixel x 64-channel output block as 4x4
//     tiles of v_mfma_f32_16x16x32_{bf16,f16} (64 accumulator VGPRs).  WP x WC waves give a
//     (64*WP)-pixel x (64*WC)-channel workgroup tile: 256x64 for 64-channel layers, 128x128 otherwise.
//   * The weights are the MFMA *A* operand (rows = channels), the im2col pixels the *B* operand
//     (cols = pixels), so an accumulator register holds 4 consecutive MFMA rows of ONE pixel.  The
//     weight rows of each 64-block are stored permuted (perm64: tile t, row 4q+j  <->  channel
//     16q+4t+j), which makes a lane's 16 accumulators 16 CONSECUTIVE channels: the epilogue writes
//     32 contiguous bytes per lane and a full 128-B line per pixel with no LDS transpose.
//   * K is walked tap-major in 64-channel steps.  One step stages a [pixels][64] and a [channels][64]
//     tile (128-B rows) into LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip).  The
//     DMA destination is lane-linear, so the bank-conflict swizzle (16-B chunk c of row r lives at
//     chunk c ^ ((r>>1)&7)) is applied to the per-lane SOURCE address and again on the ds_read_b128
//     side.  Padded taps and tail rows source a zero page instead of branching.
//   * Two LDS buffers, one barrier per K-step: DMA for step k+1 is issued before the MFMAs of step
//     k and retired by the vmcnt(0) that __syncthreads() carries.
//   * blockIdx is remapped so that the workgroups sharing an XCD (blockIdx % 8) walk neighbouring
//     pixel tiles: halo rows and weights are then shared in that XCD's L2.
//   * Epilogue fuses: folded-BN bias (9 position classes when a pre-activation BN shift was folded
//     through zero padding), PReLU, residual add, conversion to T.  Split-K mode (the 25088->512
//     FC) writes f32 partial slabs instead (reduced in fixed order by fc_finish: deterministic).
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

// 16 bytes per lane, global -> LDS, no VGPR destination.  `lds_wave_base` must be wave-uniform:
// lane l lands at lds_wave_base + 16*l.
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Bijective XCD-aware remap (cdna_hip_programming.md §5 "XCD swizzle must be bijective"): blocks
// with equal blockIdx % 8 share an XCD; give each such group one contiguous range of logical ids.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// SP (T = _Float16 only): the split-precision mode ALINK_DT_F16X2.  Every value is an f16 pair hi + lo (22 significant
// bits); activations are [pixel][2 Cin] with each 64-channel chunk stored as [hi 64 | lo 64], weights
// [Cout][tap][chunk][hi 64 | lo 64].  A real K-step becomes three: hi x W_hi, hi x W_lo, lo x W_hi into the same f32
// accumulators (lo x W_lo, 2^-22 of the sum, is dropped), so the kernel body is the same with a longer K walk.  Tensors
// carry power-of-two scales (ConvParams::acc_scale ...) that keep hi inside the f16 range and lo out of the subnormals.
template <typename T, int WP, int WC, bool DMA, bool SP>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    constexpr int BM = WP * 64, BN = WC * 64;
    constexpr int PJ = BM / 32, WJ = BN / 32;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WC, wc = wave % WC;

    const int ntn = p.Cout / BN;
    const int nwg = gridDim.x;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tile_n = lid % ntn, tile_m = lid / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;

    const int Cin = p.Cin, W = p.W, ksz = p.ksz;
    const int CinP = SP ? 2 * Cin : Cin;       // pixel pitch of the input tensor in elements
    const int Cin2 = SP ? 0 : p.Cin2;          // fused projection shortcut: extra K-steps from a second input (16-bit forms)
    const int K = (SP ? 2 : 1) * ksz * ksz * Cin + Cin2;   // weight row pitch
    const int cpt = Cin >> 6;                  // 64-channel steps per tap
    const int cpt2 = Cin2 >> 6;
    const int ntap = ksz * ksz;
    const int nk = ntap * cpt * (SP ? 3 : 1) + cpt2;
    const int kt0 = split * p.ksteps_per_split;
    const int kt1 = min(kt0 + p.ksteps_per_split, nk);

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gin2 = (const T*)p.in2;
    const T* __restrict__ gw  = (const T*)p.wgt;
    const T* __restrict__ gz  = (const T*)p.zero;

    // ---- per-thread staging state: thread stages LDS chunk (tid&7) of rows (tid>>3) + 32*j ------
    // swizzle: row r keeps logical chunk c at position c ^ ((r>>1)&7); (r>>1)&7 == (tid>>4)&7 for
    // every j because rows advance by 32.
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    int      poff[PJ], poff2[PJ];
    unsigned pmask[PJ];
    {
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int m = m0 + (tid >> 3) + 32 * j;
            unsigned mask = 0;
            int off = 0;
            poff2[j] = -1;
            if (m < p.M) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                off = ((n * p.H + iy0) * W + ix0) * CinP + chunk * 8;
                poff2[j] = (p.in2_compact ? m : (n * p.H + oy * p.stride) * W + ox * p.stride) * Cin2 + chunk * 8;
                for (int ky = 0; ky < ksz; ++ky)
                    for (int kx = 0; kx < ksz; ++kx)
                        if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)W)
                            mask |= 1u << (ky * ksz + kx);
            }
            poff[j] = off;
            pmask[j] = mask;
        }
    }
    int woff[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) woff[j] = (n0 + (tid >> 3) + 32 * j) * K + chunk * 8;

    // K-step cursor (uniform): tap index, channel step inside the tap, tap coordinates
    const int r0 = SP ? kt0 / 3 : kt0;
    int s_ph = SP ? kt0 - 3 * r0 : 0;          // SP: 0 = hi x W_hi, 1 = hi x W_lo, 2 = lo x W_hi
    int s_tap = r0 / cpt, s_cc = r0 - s_tap * cpt;
    int s_ky = s_tap / ksz, s_kx = s_tap - s_ky * ksz;

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * TILE_BYTES;
        const int tap_off = (s_ky * W + s_kx) * CinP + (SP ? 2 * s_cc + (s_ph == 2 ? 1 : 0) : s_cc) * 64;
        const int wk_off = SP ? ((s_tap * cpt + s_cc) * 2 + (s_ph == 1 ? 1 : 0)) * 64 : kt * 64;
        const bool shortcut = !SP && s_tap == ntap;        // (uniform) the K-steps of the fused projection shortcut
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const bool ok = shortcut ? poff2[j] >= 0 : (pmask[j] >> s_tap) & 1u;
            const T* src = ok ? (shortcut ? gin2 + (poff2[j] + s_cc * 64) : gin + (poff[j] + tap_off)) : gz + (tid & 7) * 8;
            if constexpr (DMA) {
                dma16(src, base + (wave * 8 + 32 * j) * 128);
            } else {
                *(uint4*)(base + ((tid >> 3) + 32 * j) * 128 + (tid & 7) * 16) = *(const uint4*)src;
            }
        }
        char* wbase = base + BM * 128;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const T* src = gw + (woff[j] + wk_off);
            if constexpr (DMA) {
                dma16(src, wbase + (wave * 8 + 32 * j) * 128);
            } else {
                *(uint4*)(wbase + ((tid >> 3) + 32 * j) * 128 + (tid & 7) * 16) = *(const uint4*)src;
            }
        }
        // advance the cursor to the next K-step
        if (SP && ++s_ph < 3) return;
        s_ph = 0;
        if (++s_cc == (shortcut ? cpt2 : cpt)) {
            s_cc = 0;
            ++s_tap;
            if (++s_kx == ksz) { s_kx = 0; ++s_ky; }
        }
    };

    // ---- fragment addressing: lane (q = lane>>4, lr = lane&15) reads row (tile*16 + lr), logical
    // chunk 4*ksub + q, at position chunk ^ ((row>>1)&7) == chunk ^ (lr>>1)
    const int q = lane >> 4, lr = lane & 15;
    const int sw = lr >> 1;
    const int foff0 = lr * 128 + (((0 | q) ^ sw) << 4);
    const int foff1 = lr * 128 + (((4 | q) ^ sw) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const char* pb, const char* wb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? foff1 : foff0;
            vec8 wf[4], pf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *(const vec8*)(wb + t * 2048 + fo);
#pragma unroll
            for (int u = 0; u < 4; ++u) pf[u] = *(const vec8*)(pb + u * 2048 + fo);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = mfma16<T>(wf[t], pf[u], acc[t][u]);
        }
    };
    if constexpr (SP) {
        // Split precision: per REAL K-step (tap, 64-channel chunk) four tiles are staged — X_lo, W_hi, X_hi, W_lo — and
        // three products taken: X_lo W_hi, X_hi W_hi, X_hi W_lo.  The four LDS tiles have fixed roles (A0 = X_lo, A1 = X_hi,
        // W0 = W_hi, W1 = W_lo); each phase's prefetch goes into a tile the phase does not read:
        //   phase a (A0, W0): fetch X_hi -> A1      phase b (A1, W0): fetch W_lo -> W1      phase c (A1, W1): fetch the next
        //   step's X_lo -> A0 and W_hi -> W0.   (The first form of this kernel staged a pixel tile AND a weight tile for each
        //   of the three products: six tile loads per real step instead of four.)
        char* const A0 = smem, * const W0 = smem + BM * 128, * const A1 = smem + TILE_BYTES, * const W1 = smem + TILE_BYTES + BM * 128;
        auto stage_a = [&](char* dst, int part) {          // the cursor's pixel tile: part 0 = hi half of the chunk, 1 = lo half
            const int tap_off = (s_ky * W + s_kx) * CinP + (2 * s_cc + part) * 64;
#pragma unroll
            for (int j = 0; j < PJ; ++j) {
                const bool ok = (pmask[j] >> s_tap) & 1u;
                dma16(ok ? gin + (poff[j] + tap_off) : gz + (tid & 7) * 8, dst + (wave * 8 + 32 * j) * 128);
            }
        };
        auto stage_w = [&](char* dst, int blk) {           // the cursor's weight tile: blk 0 = hi, 1 = lo
            const int wk_off = ((s_tap * cpt + s_cc) * 2 + blk) * 64;
#pragma unroll
            for (int j = 0; j < WJ; ++j) dma16(gw + (woff[j] + wk_off), dst + (wave * 8 + 32 * j) * 128);
        };
        auto advance = [&]() {
            if (++s_cc == cpt) {
                s_cc = 0;
                ++s_tap;
                if (++s_kx == ksz) { s_kx = 0; ++s_ky; }
            }
        };
        const int r1 = kt1 / 3;                            // real steps r0 .. r1 (a K split cuts between real steps)
        if (p.nprod == 1) {
            // the screening form (ConvParams::nprod): X_hi W_hi alone — (A1, W0) and (A0, W1) as a plain double buffer
            if (r0 < r1) {
                stage_a(A1, 0);
                stage_w(W0, 0);
                __syncthreads();
                for (int r = r0; r < r1; ++r) {
                    const bool odd = (r - r0) & 1;
                    advance();
                    if (r + 1 < r1) {
                        stage_a(odd ? A1 : A0, 0);
                        stage_w(odd ? W0 : W1, 0);
                    }
                    mma((odd ? A0 : A1) + (wp * 64) * 128, (odd ? W1 : W0) + (wc * 64) * 128);
                    __syncthreads();
                }
            }
        } else
        if (r0 < r1) {
            stage_a(A0, 1);
            stage_w(W0, 0);
            __syncthreads();
            const char* const pa0 = A0 + (wp * 64) * 128, * const pa1 = A1 + (wp * 64) * 128;
            const char* const pw0 = W0 + (wc * 64) * 128, * const pw1 = W1 + (wc * 64) * 128;
            for (int r = r0; r < r1; ++r) {
                stage_a(A1, 0);                            // X_hi of this step
                mma(pa0, pw0);                             // X_lo W_hi
                __syncthreads();
                stage_w(W1, 1);                            // W_lo of this step
                mma(pa1, pw0);                             // X_hi W_hi
                __syncthreads();
                advance();
                if (r + 1 < r1) {                          // X_lo and W_hi of the next step
                    stage_a(A0, 1);
                    stage_w(W0, 0);
                }
                mma(pa1, pw1);                             // X_hi W_lo
                __syncthreads();
            }
        }
    } else
    if (kt0 < kt1) {
        stage(0, kt0);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
            mma(smem + cur * TILE_BYTES + (wp * 64) * 128, smem + cur * TILE_BYTES + (BM + wc * 64) * 128);
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds, for pixel (u*16 + lr), channels cbase + 4t + j  (16 consecutive) -
    const int cbase = n0 + wc * 64 + 16 * q;
    if (p.splitk > 1) {
        float* slab = (float*)p.out + (size_t)split * p.M * p.Cout;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + wp * 64 + 16 * u + lr;
            if (m < p.M) {
                float* o = slab + (size_t)m * p.Cout + cbase;
#pragma unroll
                for (int t = 0; t < 4; ++t) *(f32x4*)(o + 4 * t) = acc[t][u];
            }
        }
        return;
    }

    // All residual loads first (independent: one exposed latency instead of one per pixel tile), then
    // bias (by border class) / PReLU / add / store.
    const int HoWo = p.Ho * p.Wo;
    size_t off[4];
    bool ok[4];
    int cls[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int m = m0 + wp * 64 + 16 * u + lr;
        ok[u] = m < p.M;
        const int mc = ok[u] ? m : p.M - 1;
        // SP: pixel pitch 2 Cout, the lane's 16 channels are the hi run of chunk cbase / 64 (lo run: + 64 elements)
        off[u] = SP ? (size_t)mc * (2 * p.Cout) + (size_t)(cbase >> 6) * 128 + (cbase & 63) : (size_t)mc * p.Cout + cbase;
        cls[u] = 0;
        if (p.border_cls) {
            const int rem = mc % HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int rc = (oy == 0) ? 0 : ((oy == p.Ho - 1) ? 2 : 1);
            const int cc = (ox == 0) ? 0 : ((ox == p.Wo - 1) ? 2 : 1);
            cls[u] = rc * 3 + cc;
        }
    }
    vec8 res[4][SP ? 4 : 2];   // the residual, or (backward mode) the stored forward activation; SP: hi, hi, lo, lo
    const T* extra = (const T*)(p.dact ? p.dact : p.resid);
    if (extra) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            res[u][0] = *(const vec8*)(extra + off[u]);
            res[u][1] = *(const vec8*)(extra + off[u] + 8);
            if (SP) {
                res[u][2] = *(const vec8*)(extra + off[u] + 64);
                res[u][3] = *(const vec8*)(extra + off[u] + 72);
            }
        }
    }
    f32x4 bia[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) bia[u][t] = *(const f32x4*)(p.bias + cls[u] * p.Cout + cbase + 4 * t);
    float al[16];
    if (p.alpha) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a4 = *(const f32x4*)(p.alpha + cbase + 4 * t);
#pragma unroll
            for (int j = 0; j < 4; ++j) al[4 * t + j] = a4[j];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[4 * t + j] = SP ? fmaf(acc[t][u][j], p.acc_scale, bia[u][t][j] * p.bias_scale) : acc[t][u][j] + bia[u][t][j];
        if (!SP && p.dact) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] *= (float)res[u][0][i] > 0.f ? 1.f : al[i];
                v[8 + i] *= (float)res[u][1][i] > 0.f ? 1.f : al[8 + i];
            }
        } else {
            if (p.alpha) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * al[i];
            }
            if (p.resid) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (SP) {   // hi + lo in f32: at most one rounding, 2^-24 relative
                        v[i] = fmaf((float)res[u][0][i] + (float)res[u][2][i], p.res_scale, v[i]);
                        v[8 + i] = fmaf((float)res[u][1][i] + (float)res[u][3][i], p.res_scale, v[8 + i]);
                    } else {
                        v[i] += (float)res[u][0][i];
                        v[8 + i] += (float)res[u][1][i];
                    }
                }
            }
        }
        if (p.post_relu) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = relu_keep_nan(v[i]);
        }
        if (ok[u]) {
            vec8 o0, o1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o0[i] = (T)v[i];
                o1[i] = (T)v[8 + i];
            }
            *(vec8*)((T*)p.out + off[u]) = o0;
            *(vec8*)((T*)p.out + off[u] + 8) = o1;
            if (SP) {
                vec8 l0, l1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    l0[i] = (T)(v[i] - (float)o0[i]);
                    l1[i] = (T)(v[8 + i] - (float)o1[i]);
                }
                *(vec8*)((T*)p.out + off[u] + 64) = l0;
                *(vec8*)((T*)p.out + off[u] + 72) = l1;
            }
        }
    }
}

template <typename T, int WP, int WC, bool DMA, bool SP = false>
hipError_t launch_one(const ConvParams& p, hipStream_t stream) {
    constexpr int BM = WP * 64, BN = WC * 64;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
    const int ntm = (p.M + BM - 1) / BM, ntn = p.Cout / BN;
    dim3 grid(ntm * ntn, p.splitk, 1), block(256, 1, 1);
    hipLaunchKernelGGL((conv_igemm_kernel<T, WP, WC, DMA, SP>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <typename T, int WP, int WC, bool DMA, bool SP = false>
hipError_t set_attr_one() {
    constexpr size_t lds = 2 * (size_t)(WP * 64 + WC * 64) * 128;
    return hipFuncSetAttribute((const void*)conv_igemm_kernel<T, WP, WC, DMA, SP>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

bool g_use_dma = true;

}  // namespace

// test hook: 0 = register staging (global_load -> ds_write), 1 = LDS-DMA (default)
extern "C" void alink_debug_set_dma(int on) { g_use_dma = on != 0; }

hipError_t conv_set_attributes() {
    hipError_t e;
#define A(T, WP, WC)                                                    \
    if ((e = set_attr_one<T, WP, WC, true>()) != hipSuccess) return e;  \
    if ((e = set_attr_one<T, WP, WC, false>()) != hipSuccess) return e;
    A(__bf16, 4, 1) A(__bf16, 2, 2) A(_Float16, 4, 1) A(_Float16, 2, 2)
#undef A
    if ((e = set_attr_one<_Float16, 4, 1, true, true>()) != hipSuccess) return e;
    if ((e = set_attr_one<_Float16, 2, 2, true, true>()) != hipSuccess) return e;
    return hipSuccess;
}

double conv_flops(const ConvParams& p) {
    return 2.0 * (double)p.M * (double)p.Cout * (double)(p.ksz * p.ksz * p.Cin + (p.in2 ? p.Cin2 : 0));
}

hipError_t launch_conv_igemm(int dtype, const ConvParams& p, hipStream_t stream) {
    // host-side shape contract of the kernel (checked by callers too; never launch out of contract)
    if (p.Cin % 64 || p.Cout % 64 || p.M <= 0 || p.splitk < 1) return hipErrorInvalidValue;
    // fused shortcut: whole 64-channel steps, one fused launch (no K split), 16-bit storage only
    if (p.in2 ? (p.Cin2 <= 0 || p.Cin2 % 64 || p.splitk != 1 || dtype == ALINK_DT_F16X2) : p.Cin2 != 0) return hipErrorInvalidValue;
    if ((long long)p.N * p.H * p.W * p.Cin2 >= (1ll << 31)) return hipErrorInvalidValue;
    if (p.ksz * p.ksz > 32) return hipErrorInvalidValue;  // tap mask is 32 bits
    const int two = dtype == ALINK_DT_F16X2 ? 2 : 1;
    if ((long long)p.N * p.H * p.W * p.Cin * two >= (1ll << 31)) return hipErrorInvalidValue;
    if ((long long)p.Cout * p.ksz * p.ksz * p.Cin * two >= (1ll << 31)) return hipErrorInvalidValue;
    // a lone image: the latency form (conv3x3_lat.hip: one wave per 16 x 16 output block, this kernel's walk and epilogue: bit-identical)
    if (conv_gemm_lat_applies(dtype, p)) return launch_conv_gemm_lat(dtype, p, stream);
    const bool wide = (p.Cout % 128) == 0;
    if (dtype == ALINK_DT_F16X2) {
        // split precision: LDS-DMA staging only; no backward mode; a K split must cut between real K-steps
        if (p.dact || (p.splitk > 1 && p.ksteps_per_split % 3)) return hipErrorInvalidValue;
        return wide ? launch_one<_Float16, 2, 2, true, true>(p, stream) : launch_one<_Float16, 4, 1, true, true>(p, stream);
    }
#define L(T)                                                                          \
    (wide ? (g_use_dma ? launch_one<T, 2, 2, true>(p, stream)                         \
                       : launch_one<T, 2, 2, false>(p, stream))                       \
          : (g_use_dma ? launch_one<T, 4, 1, true>(p, stream)                         \
                       : launch_one<T, 4, 1, false>(p, stream)))
    if (dtype == ALINK_DT_BF16) return L(__bf16);
    if (dtype == ALINK_DT_F16) return L(_Float16);
#undef L
    return hipErrorInvalidValue;
}

}  // namespace alink
