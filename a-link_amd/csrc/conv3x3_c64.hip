// conv3x3_c64.hip — 3x3 / stride 1 / pad 1 NHWC convolution for the 64 -> 64 channel layers at the FRONT of the network
// (stage 1: the 112-wide conv1 of the first unit and the 56-wide units), as a ROLLING-ROW kernel with the WEIGHTS IN
// REGISTERS.
//
// These layers have K = 576 only: nine K-steps.  In the tile kernels (conv3x3_direct / conv3x3_linear) a workgroup
// therefore spends more than half of its life in its prologue and epilogue, re-streams the whole 73 KB weight tensor
// from L2 for every 57 KB of output, and meets a barrier every 56 MFMAs — they run at 2.5 TB/s of HBM traffic and 30 % of
// the MFMA rate, bound by neither.  Here:
//
//   * the folded weights of a wave's 32 output channels (32 x 576 x 2 B = 36 KB = 144 VGPRs per lane) are loaded ONCE per
//     workgroup into registers and are the MFMA A operand for the whole kernel: no weight DMA, no weight reads from LDS;
//   * a workgroup is persistent: it walks bands of BR output rows of one image; inside a band the input rows live in a
//     ring of 3P + 2 row slots in LDS (P = 224 / W output rows = 14 MFMA pixel tiles per pass): the P + 2 rows a pass
//     reads and the P + P rows of the next TWO passes, fetched by LDS-DMA while the current pass computes (one pass of
//     lead leaves 28 KB in flight per CU: by Little's law 2.4 TB/s chip-wide at the ~3 us a loaded HBM takes).  Every input
//     row is fetched once per band (1 + 2 / BR of the tensor in all) and there is ONE barrier per pass (252 MFMAs per
//     wave) instead of one per K-step;
//   * left / right padding are two zero pixels kept at the ends of every row slot, top / bottom padding rows are fetched
//     from the zero page; a 16-pixel MFMA tile is 16 consecutive pixels of one row, lanes permuted by delta() and 16-B
//     pieces XOR-swizzled by (position >> 1) & 7 as in the other kernels, so every ds_read_b128 is conflict-free;
//   * epilogue as everywhere: folded-BN bias by border class, PReLU or residual, 8 consecutive channels per lane.
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

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ int delta(int lr) { return lr < 4 ? 2 * lr : (lr < 12 ? 2 * (lr - 4) + 1 : 2 * (lr - 8)); }

// this wave's LDS-DMAs have landed and its own LDS reads have returned, then the workgroup barrier
// (conv3x3_linear.hip, wait_dma_then_barrier: why both)
__device__ __forceinline__ void wait_all_then_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the same, leaving the wave's N youngest vector-memory operations in flight: at the top of a pass those are the output
// stores of the pass before (issued after the prefetch DMAs this wait is for), whose latency nobody needs to see
template <int N>
__device__ __forceinline__ void wait_all_but_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int NT = 256;

template <int W>
struct C64 {
    static constexpr int P = 224 / W;                 // output rows per pass (14 pixel tiles)
    static constexpr int RING = 3 * P + 2;            // row slots: the P + 2 a pass reads and the P + P of the next two
    static constexpr int BR = W == 112 ? 16 : 8;      // output rows per band (divides H = W)
    static constexpr int PITCHPX = W + 2;             // a slot: zero pixel, W pixels, zero pixel
    static constexpr int PITCH = PITCHPX * 128;
    static constexpr int XBYTES = RING * PITCH;
    static constexpr size_t lds_bytes() { return (size_t)XBYTES + 10 * 64 * 4; }
    static_assert(224 % W == 0 && W % 16 == 0, "whole 16-pixel tiles per row");
    static_assert(W % BR == 0 && BR % P == 0 && BR / P >= 4 && (BR / P) % 2 == 0, "bands of an even number (>= 4) of whole passes");
};

template <typename T, int W, int EPI>
__global__ __launch_bounds__(NT, 1) void conv3x3_c64_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    typedef C64<W> G;
    constexpr int P = G::P, RING = G::RING, BR = G::BR, PITCH = G::PITCH, TPR = W / 16;   // TPR: tiles per row
    constexpr int TPWV = 7;                           // tiles per wave: 14 per pass over two pixel halves
    static_assert(TPR == TPWV * 2 / P, "a wave's 7 tiles lie in one row (W = 112) — the only geometry instantiated");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, ch = wave & 1;          // pixel half (= output row of the pass for W = 112), channel half
    const int q = lane >> 4, lr = lane & 15;
    const int H = p.H;                                 // == W (launch condition)

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- once per workgroup: zero borders of every slot, epilogue tables, weights into registers ---------------------
    for (int i = tid; i < RING * 16; i += NT) {
        const int slot = i >> 4, side = (i >> 3) & 1, piece = i & 7;
        *(uint4*)(smem + slot * PITCH + (side ? (W + 1) * 128 : 0) + piece * 16) = uint4{0u, 0u, 0u, 0u};
    }
    const int ncls = p.border_cls ? 9 : 1;
    float* const ebias = (float*)(smem + G::XBYTES);
    float* const ealpha = ebias + 9 * 64;
    for (int i = tid; i < ncls * 64; i += NT) ebias[i] = p.bias[i];
    if (p.alpha)
        for (int i = tid; i < 64; i += NT) ealpha[i] = p.alpha[i];

    vec8 wr[2][9][2];                                  // [channel tile][tap][K half]: rows perm32-permuted, K = [tap][64]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wr[ct][tap][ks] = *(const vec8*)(gw + (size_t)(ch * 32 + 16 * ct + lr) * 576 + tap * 64 + ks * 32 + 8 * q);

    // per-lane operand offsets inside a slot for the three horizontal taps and the two K halves: pixel x = 16 u + d,
    // position x + kx (slot position 0 is the left zero pixel), 16-B piece (4 ks + q) ^ ((position >> 1) & 7); the tile
    // (16 u) leaves the swizzle term alone, so 2048 u goes into the ds_read immediate
    const int d = delta(lr);
    int loff[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            loff[kx][ks] = (d + kx) * 128 + ((((ks << 2) | q) ^ (((d + kx) >> 1) & 7)) << 4);

    // ---- row staging: wave-DMA unit = 8 pixels of one row (1 KB); a row is W / 8 units; the units of a request are
    // dealt round-robin to the four waves -------------------------------------------------------------------------------
    auto stage_rows = [&](long long img_row0, int r0, int count, int slot0) {
        const int units = count * (W / 8);
        for (int uidx = wave; uidx < units; uidx += 4) {
            const int ri = uidx / (W / 8), seg = uidx - ri * (W / 8);
            const int r = r0 + ri;
            int slot = slot0 + ri;
            if (slot >= RING) slot -= RING;
            const int px = seg * 8 + (lane >> 3);
            const int piece = (lane & 7) ^ (((px + 1) >> 1) & 7);
            const bool ok = (unsigned)r < (unsigned)H && (EPI != 0 || p.ablate != 2);   // (diagnostic timing mode 2: every row from the zero page)
            const T* src = ok ? gin + ((size_t)((img_row0 + r) * W + px) * 64 + piece * 8) : gz + (lane & 7) * 8;
            dma16(src, smem + slot * PITCH + (seg * 8 + 1) * 128);
        }
    };

    const int bands_per_img = H / BR;
    const long long nbands = (long long)p.N * bands_per_img;
    const int nwg = gridDim.x;
    const int lid = xcd_remap(blockIdx.x, nwg);        // consecutive bands (shared halo rows) on one XCD's L2

    float al[8];
    if (p.alpha) {
#pragma unroll
        for (int i = 0; i < 8; ++i) al[i] = p.alpha[ch * 32 + 8 * q + i];
    }

    // one pass of MFMAs: output rows (first row of the pass) + ph, taps from the ring slots starting at s0
    auto compute = [&](f32x4 (&acc)[2][TPWV], int s0) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int u = 0; u < TPWV; ++u) acc[ct][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == 0 && p.ablate == 1) return;     // (diagnostic timing mode 1, generic form only: no operand reads, no MFMAs)
        int sb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int slot = s0 + ph + ky;                                 // row (y + ph) + ky - 1
            if (slot >= RING) slot -= RING;
            sb[ky] = slot * PITCH;
        }
        vec8 pf[2][TPWV];
#pragma unroll
        for (int u = 0; u < TPWV; ++u) pf[0][u] = *(const vec8*)(smem + (sb[0] + loff[0][0]) + 2048 * u);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) {
                const int tn = (st + 1) >> 1, ksn = (st + 1) & 1;
#pragma unroll
                for (int u = 0; u < TPWV; ++u)
                    pf[(st + 1) & 1][u] = *(const vec8*)(smem + (sb[tn / 3] + loff[tn % 3][ksn]) + 2048 * u);
            }
            const int tap = st >> 1, ks = st & 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int u = 0; u < TPWV; ++u) acc[ct][u] = mfma16<T>(wr[ct][tap][ks], pf[st & 1][u], acc[ct][u]);
        }
    };
    // epilogue of one pass: output row y, pixel x = 16 u + d, channels 32 ch + 8 q .. + 7.  EPI fixes the form at compile
    // time (1: bias by border class + PReLU, 2: bias + residual, 0: by run-time flags) so that it is straight-line code the
    // compiler can weave into the NEXT pass's MFMAs (it is issued one pass late, from the previous pass's accumulators:
    // with one wave per SIMD nothing else would cover its ~2 k cycles of VALU, LDS reads and stores)
    auto epilogue = [&](const f32x4 (&acc)[2][TPWV], int y, long long img_row0) {
        const bool has_alpha = EPI == 1 || (EPI == 0 && p.alpha), has_resid = EPI == 2 || (EPI == 0 && p.resid);
        const bool classes = EPI == 1 || (EPI == 0 && p.border_cls);
        const int rc = classes ? (y == 0 ? 0 : (y == H - 1 ? 2 : 1)) : 0;
        const size_t rowoff = (size_t)((img_row0 + y) * W) * 64 + ch * 32 + 8 * q;
        vec8 res[TPWV];
        if (has_resid) {
#pragma unroll
            for (int u = 0; u < TPWV; ++u) res[u] = *(const vec8*)((const T*)p.resid + rowoff + (size_t)(16 * u + d) * 64);
        }
#pragma unroll
        for (int u = 0; u < TPWV; ++u) {
            const int x = 16 * u + d;
            const int cls = classes ? rc * 3 + (x == 0 ? 0 : (x == W - 1 ? 2 : 1)) : 0;
            const f32x4 b0 = *(const f32x4*)(ebias + cls * 64 + ch * 32 + 8 * q);
            const f32x4 b1 = *(const f32x4*)(ebias + cls * 64 + ch * 32 + 8 * q + 4);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc[0][u][j] + b0[j]; v[4 + j] = acc[1][u][j] + b1[j]; }
            if (has_alpha) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * al[i];
            }
            if (has_resid) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += (float)res[u][i];
            }
            vec8 o8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o8[i] = (T)v[i];
            if (EPI != 0 || p.ablate != 3 || o8[0] == (T)12345.f)   // (diagnostic timing mode 3, generic form only: no output stores)
                *(vec8*)((T*)p.out + rowoff + (size_t)x * 64) = o8;
        }
    };
    // (A scheduling request that spreads the epilogue evenly over the block — __builtin_amdgcn_sched_group_barrier, two
    // MFMAs / one LDS read / three VALU instructions repeated — was tried and is slower: 0.373 against 0.307 ms; it breaks up
    // the operand prefetch the compiler's own order keeps.  Left to itself the compiler weaves the epilogue into the last
    // fifth of the MFMAs.)
    constexpr int VMI = TPWV * ((EPI == 1) ? 2 : 3);   // vector-memory operations a wave issues per steady-state iteration
                                                       // (7 row DMAs + 7 stores [+ 7 residual loads]; EPI 0 counts the loads too)

    for (long long band = lid; band < nbands; band += nwg) {
        const int n = (int)(band / bands_per_img);
        const int y0 = (int)(band - (long long)n * bands_per_img) * BR;
        const long long img_row0 = (long long)n * H;
        // everyone is past the previous band's reads (and the table / border stores of the prologue): the ring is free
        wait_all_then_barrier();
        stage_rows(img_row0, y0 - 1, P + 2, 0);                     // rows y0-1 .. y0+P -> slots 0 .. P+1   (pass 0)
        stage_rows(img_row0, y0 + P + 1, P, P + 2);                 // rows of pass 1
        int s0 = 0;                                                  // slot of row (first output row of the pass) - 1
        f32x4 accA[2][TPWV], accB[2][TPWV];
        // ---- pass 0: its rows are everything but the wave's 7 youngest DMAs (those are pass 1's) -----------------------
        wait_all_but_then_barrier<TPWV>();
        stage_rows(img_row0, y0 + 2 * P + 1, P, 2 * P + 2);         // rows of pass 2
        compute(accA, 0);
        s0 = P;
        // ---- passes 1 .. NPASS-1, two per trip (accumulator sets alternate); each trip also finishes the pass before ----
        // Waits: a wave's vector-memory operations retire in issue order (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)"), so
        // "all but the N youngest" with N = what ONE iteration issues leaves this iteration's prefetch and stores in flight,
        // in whatever order the compiler emitted them, and guarantees everything older — the rows of the pass about to
        // be computed among it.  (Pass 1: iteration 0 issued 7 DMAs only.)
#pragma unroll 1
        for (int k = 1; k < BR / P; k += 2) {
            if (k == 1) wait_all_but_then_barrier<TPWV>();
            else        wait_all_but_then_barrier<VMI>();
            if (k + 2 < BR / P) {
                int sn = s0 + 2 * P + 2;
                if (sn >= RING) sn -= RING;
                stage_rows(img_row0, y0 + (k + 2) * P + 1, P, sn);
            }
            compute(accB, s0);
            epilogue(accA, y0 + (k - 1) * P + ph, img_row0);
            s0 += P;
            if (s0 >= RING) s0 -= RING;
            if (k + 1 < BR / P) {
                wait_all_but_then_barrier<VMI>();
                if (k + 3 < BR / P) {
                    int sn = s0 + 2 * P + 2;
                    if (sn >= RING) sn -= RING;
                    stage_rows(img_row0, y0 + (k + 3) * P + 1, P, sn);
                }
                compute(accA, s0);
                epilogue(accB, y0 + k * P + ph, img_row0);
                s0 += P;
                if (s0 >= RING) s0 -= RING;
            } else {
                epilogue(accB, y0 + k * P + ph, img_row0);           // (odd number of passes: not instantiated today)
                goto band_done;
            }
        }
        epilogue(accA, y0 + (BR / P - 1) * P + ph, img_row0);       // the last pass (NPASS even: it went to accA)
    band_done:;
    }
}

bool g_use_c64 = true;

}  // namespace

extern "C" void alink_debug_set_c64(int on) { g_use_c64 = on != 0; }

// 21: the rolling-row kernel for 112 x 112 x 64 -> 64 (0 = not applicable)
int c64_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout) {
    if (!g_use_c64 || ksz != 3 || stride != 1 || pad != 1 || Cin != 64 || Cout != 64 || H != W) return 0;
    return W == 112 ? 21 : 0;
}

template <typename T, int EPI>
static hipError_t c64_attr() {
    return hipFuncSetAttribute((const void*)conv3x3_c64_kernel<T, 112, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)C64<112>::lds_bytes());
}
hipError_t c64_set_attributes() {
    hipError_t e;
    if ((e = c64_attr<__bf16, 0>()) != hipSuccess || (e = c64_attr<__bf16, 1>()) != hipSuccess || (e = c64_attr<__bf16, 2>()) != hipSuccess) return e;
    if ((e = c64_attr<_Float16, 0>()) != hipSuccess || (e = c64_attr<_Float16, 1>()) != hipSuccess || (e = c64_attr<_Float16, 2>()) != hipSuccess) return e;
    return hipSuccess;
}
template <typename T>
static void c64_launch(const ConvParams& p, unsigned grid, hipStream_t st) {
    const size_t lds = C64<112>::lds_bytes();
    if (p.alpha && !p.resid && p.border_cls && !p.ablate)       hipLaunchKernelGGL((conv3x3_c64_kernel<T, 112, 1>), dim3(grid), dim3(NT), lds, st, p);
    else if (!p.alpha && p.resid && !p.border_cls && !p.ablate) hipLaunchKernelGGL((conv3x3_c64_kernel<T, 112, 2>), dim3(grid), dim3(NT), lds, st, p);
    else                                                         hipLaunchKernelGGL((conv3x3_c64_kernel<T, 112, 0>), dim3(grid), dim3(NT), lds, st, p);
}

hipError_t launch_conv3x3_c64(int variant, int dtype, const ConvParams& p, hipStream_t st) {
    if (variant != 21 || p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.Cin != 64 || p.Cout != 64 || p.H != 112 || p.W != 112)
        return hipErrorInvalidValue;
    if (p.splitk != 1 || p.dact || p.post_relu || p.in2 || p.N <= 0) return hipErrorInvalidValue;   // forward forms only
    if ((long long)p.N * p.H * p.W * 64 >= (1ll << 31)) return hipErrorInvalidValue;
    if (dtype != ALINK_DT_BF16 && dtype != ALINK_DT_F16) return hipErrorInvalidValue;
    const long long nbands = (long long)p.N * (p.H / C64<112>::BR);
    const unsigned grid = (unsigned)(nbands < 256 ? nbands : 256);                                   // one persistent workgroup per CU
    if (dtype == ALINK_DT_BF16) c64_launch<__bf16>(p, grid, st);
    else                        c64_launch<_Float16>(p, grid, st);
    return hipGetLastError();
}

}  // namespace alink
