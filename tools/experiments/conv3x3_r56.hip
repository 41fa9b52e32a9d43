// conv3x3_r56.hip — 3x3 / stride 1 / pad 1 NHWC convolution for the 64-channel-input layers at 56 x 56 (the stage-1 units,
// 64 -> 64, and the first convolution of stage 2, 64 -> 128): the ROLLING-ROW kernel with the WEIGHTS IN REGISTERS of
// conv3x3_c64.hip on a LINEAR ring.
//
// conv3x3_c64.hip keeps every row in a slot with a zero pixel at either end; that needs whole 16-pixel MFMA tiles per row,
// and 56 is three and a half.  Here the ring is linear — position = ring row x 56 + x, no padding pixels — so a tile is
// 16 consecutive positions across row ends, and what padding would do is done by ADDRESS as in conv3x3_linear.hip: a lane
// whose tap crosses the left / right image border adds a bit far above the allocation and reads zeros; rows above / below
// the image come from the zero page.  A pass reads one row before and one after its own; so that this window never wraps
// inside the ring, ring row R - 1 is mirrored in front of row 0 and row 0 behind row R - 1 (those rows are DMA'd twice).
//
//   * 64 -> 64:  pass = 4 rows = 224 pixels = 14 tiles; waves = 2 pixel halves x 2 channel halves, 7 x 2 tiles each.
//   * 64 -> 128: pass = 2 rows = 112 pixels = 7 tiles; the four waves are the four channel quarters and share every operand.
//   Either way a wave holds 32 output channels x 576 = 144 registers of weights, loaded once; workgroups are persistent, one
//   per CU, and own a CONTIGUOUS range of passes (runs inside an image roll through the ring); rows arrive by LDS-DMA two
//   passes ahead; one barrier per pass; the epilogue of a pass is issued one pass late, beside the next pass's MFMAs.
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

__device__ __forceinline__ int delta(int lr) { return lr < 4 ? 2 * lr : (lr < 12 ? 2 * (lr - 4) + 1 : 2 * (lr - 8)); }

// this wave's LDS-DMAs have landed and its own LDS reads have returned (conv3x3_linear.hip, wait_dma_then_barrier: why
// both), leaving the wave's N youngest vector-memory operations in flight
template <int N>
__device__ __forceinline__ void wait_all_but_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int NT = 256;
constexpr int W = 56, H = 56, CIN = 64;
constexpr int PXB = 128;                        // bytes per position
constexpr int LEAD = 16 * PXB;                  // unused positions in front of the ring: a lane's base address never goes negative
constexpr int TPWV = 7;                         // tiles per wave and pass

template <int COUT>
struct G56 {
    static constexpr int NPH = COUT == 64 ? 2 : 1;         // pixel halves among the four waves
    static constexpr int P = 2 * NPH;                       // rows per pass
    static constexpr int PPI = H / P;                       // passes per image
    static constexpr int R = 4 * P;                         // ring rows: the P + 2 a pass reads, the P + P of the next two (rounded to 4 groups)
    static constexpr int RINGPX = (R + 2) * W;              // [copy of row R-1][rows 0 .. R-1][copy of row 0]
    static constexpr int XBYTES = LEAD + RINGPX * PXB;
    static constexpr int UPP = P * W / 8;                   // DMA units (8 positions = 1 KB) per group of P rows; a row is 7
    static constexpr int DPG = UPP / 4;                     // what every wave issues per group at the least
    static constexpr size_t lds_bytes() { return (size_t)XBYTES + 10 * COUT * 4; }
};

// EPI: 1 = bias by border class + PReLU (a unit's conv1), 2 = bias + residual (conv2), 0 = by run-time flags
template <typename T, int COUT, int EPI>
__global__ __launch_bounds__(NT, 1) void conv3x3_r56_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    typedef G56<COUT> G;
    constexpr int NPH = G::NPH, P = G::P, PPI = G::PPI, R = G::R, DPG = G::DPG;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = NPH == 2 ? wave >> 1 : 0;                         // pixel half of the pass (64 -> 64)
    const int cq = NPH == 2 ? wave & 1 : wave;                       // the wave's 32 output channels
    const int q = lane >> 4, lr = lane & 15;

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- once per workgroup: epilogue tables, weights into registers ----------------------------------------------------
    const bool has_alpha = EPI == 1 || (EPI == 0 && p.alpha), has_resid = EPI == 2 || (EPI == 0 && p.resid);
    const bool classes = EPI == 1 || (EPI == 0 && p.border_cls);
    const int ncls = classes ? 9 : 1;
    float* const ebias = (float*)(smem + G::XBYTES);
    float* const ealpha = ebias + 9 * COUT;
    for (int i = tid; i < ncls * COUT; i += NT) ebias[i] = p.bias[i];
    if (has_alpha)
        for (int i = tid; i < COUT; i += NT) ealpha[i] = p.alpha[i];

    vec8 wr[2][9][2];                                  // [channel tile][tap][K half]: rows perm32-permuted, K = [tap][64]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wr[ct][tap][ks] = *(const vec8*)(gw + (size_t)(cq * 32 + 16 * ct + lr) * 576 + tap * 64 + ks * 32 + 8 * q);

    // A pass's output pixel o sits at ring position (s0 + 1) W + o, s0 (a multiple of P) the ring row of the pass's first row:
    // = 8 mod 16 + o (56 = 8, P x 56 = 0 mod 16), so the swizzle term of a tap's operand depends on neither pass nor tile.
    const int d = delta(lr);
    int toff[9];                                       // K half 0; K half 1 is the same address ^ 64
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int sh = (tap / 3 - 1) * W + tap % 3 - 1;
        const int sw = ((8 + 7 * 16 + d + sh) >> 1) & 7;
        toff[tap] = (ph * 7 * 16 + d + sh) * PXB + ((q ^ sw) << 4);
    }
    // per tile of the wave: is the lane's pixel in the first / last column (bits u, 8 + u), which of the pass's rows is it in (2 bits at 16 + 2 u)
    unsigned bits = 0;
#pragma unroll
    for (int u = 0; u < TPWV; ++u) {
        const int o = 16 * (7 * ph + u) + d, ro = o / W, x = o - ro * W;
        bits |= (x == 0 ? 1u : 0u) << u | (x == W - 1 ? 1u : 0u) << (8 + u) | (unsigned)ro << (16 + 2 * u);
    }

    // ---- staging: image rows r0 .. r0 + nrows - 1 into ring rows g0 .. (g0 = -1: the slot in front of ring row 0); a unit = 8
    // consecutive positions, dealt round-robin to the four waves --------------------------------------------------------------
    auto stage = [&](long long img_row0, int r0, int nrows, int g0) __attribute__((always_inline)) {
        for (int j = wave; j < nrows * (W / 8); j += 4) {
            const int r = r0 + j / (W / 8);
            const int pos = (g0 + 1) * W + 8 * j + (lane >> 3);
            const int piece = (lane & 7) ^ ((pos >> 1) & 7);
            const bool ok = (unsigned)r < (unsigned)H;
            const T* src = ok ? gin + ((size_t)(img_row0 + r0) * W + 8 * j + (lane >> 3)) * CIN + piece * 8 : gz + (lane & 7) * 8;
            dma16(src, smem + LEAD + ((g0 + 1) * W + 8 * j) * PXB);
        }
    };
    // group m of a run = the P rows BEHIND pass m's first row (the last of them is the row below pass m), into ring rows
    // P (m % 4) + 1 ..; the fourth group runs into the slot behind ring row R - 1, so its last row is also staged at ring
    // row 0, and ring row R - 1 (the one before) in the slot in front of ring row 0
    auto stage_group = [&](long long img_row0, int y0, int m) __attribute__((always_inline)) {
        const int g0 = (m & 3) * P + 1;
        stage(img_row0, y0 + P * m + 1, P, g0);
        if (g0 == R - P + 1) {
            stage(img_row0, y0 + P * m + P, 1, 0);
            stage(img_row0, y0 + P * m + P - 1, 1, -1);
        }
    };

    // one pass of MFMAs from the window at ring row s0; K order [tap][K half]
    auto compute = [&](f32x4 (&acc)[2][TPWV], int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int u = 0; u < TPWV; ++u) acc[ct][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int sbase = LEAD + (s0 + 1) * W * PXB;
        auto frag = [&](int st, int u) __attribute__((always_inline)) -> vec8 {
            const int tap = st >> 1, ks = st & 1, kx = tap % 3;
            int a = (toff[tap] + sbase) ^ (ks << 6);
            if (kx == 0) a += (int)((bits >> u) & 1u) << 18;         // beyond the allocation: the read returns zero
            if (kx == 2) a += (int)((bits >> (8 + u)) & 1u) << 18;
            return *(const vec8*)(smem + a + u * 16 * PXB);
        };
        vec8 pf[2][TPWV];
#pragma unroll
        for (int u = 0; u < TPWV; ++u) pf[0][u] = frag(0, u);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) {
#pragma unroll
                for (int u = 0; u < TPWV; ++u) pf[(st + 1) & 1][u] = frag(st + 1, u);
            }
            const int tap = st >> 1, ks = st & 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int u = 0; u < TPWV; ++u) acc[ct][u] = mfma16<T>(wr[ct][tap][ks], pf[st & 1][u], acc[ct][u]);
        }
    };
    // epilogue of the pass whose first row is image row y: the wave's pixel o = 16 (7 ph + u) + d, channels 32 cq + 8 q .. + 7
    auto epilogue = [&](const f32x4 (&acc)[2][TPWV], int y, long long img_row0) __attribute__((always_inline)) {
        const size_t pix0 = (size_t)(img_row0 + y) * W + 16 * 7 * ph + d;
        const size_t choff = (size_t)cq * 32 + 8 * q;
        vec8 res[TPWV];
        if (has_resid) {
#pragma unroll
            for (int u = 0; u < TPWV; ++u) res[u] = *(const vec8*)((const T*)p.resid + (pix0 + 16 * u) * COUT + choff);
        }
        f32x4 al0, al1, b0, b1;
        if (has_alpha) { al0 = *(const f32x4*)(ealpha + choff); al1 = *(const f32x4*)(ealpha + choff + 4); }
        if (!classes) { b0 = *(const f32x4*)(ebias + choff); b1 = *(const f32x4*)(ebias + choff + 4); }
#pragma unroll
        for (int u = 0; u < TPWV; ++u) {
            if (classes) {
                const int yy = y + (int)((bits >> (16 + 2 * u)) & 3u);
                const int rc = yy == 0 ? 0 : (yy == H - 1 ? 2 : 1);
                const int cx = (bits >> u) & 1u ? 0 : ((bits >> (8 + u)) & 1u ? 2 : 1);
                const int cls = rc * 3 + cx;
                b0 = *(const f32x4*)(ebias + cls * COUT + choff);
                b1 = *(const f32x4*)(ebias + cls * COUT + choff + 4);
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc[0][u][j] + b0[j]; v[4 + j] = acc[1][u][j] + b1[j]; }
            if (has_alpha) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * (k < 4 ? al0[k & 3] : al1[k & 3]);
            }
            if (has_resid) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += (float)res[u][k];
            }
            vec8 o8;
#pragma unroll
            for (int k = 0; k < 8; ++k) o8[k] = (T)v[k];
            *(vec8*)((T*)p.out + (pix0 + 16 * u) * COUT + choff) = o8;
        }
    };
    // Waits.  A wave's vector-memory operations retire in issue order (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)"); pass k needs
    // group k, issued two iterations earlier, so "all but what the iteration before issued AT THE LEAST" is enough and never
    // too little: DPG row DMAs if it staged (the mirror DMAs of one group in four and the odd unit of a wave come on top),
    // then — behind them in program order — 7 stores (+ 7 residual loads) if it ran an epilogue.
    constexpr int EOPS = TPWV * (EPI == 2 ? 2 : 1);
    auto top_of_pass = [&](bool staged, bool epi) __attribute__((always_inline)) {
        if (staged) { if (epi) wait_all_but_then_barrier<DPG + EOPS>(); else wait_all_but_then_barrier<DPG>(); }
        else        { if (epi) wait_all_but_then_barrier<EOPS>();       else wait_all_but_then_barrier<0>(); }
    };

    const long long npass_all = (long long)p.N * PPI;
    const long long p0 = npass_all * blockIdx.x / gridDim.x, p1 = npass_all * (blockIdx.x + 1) / gridDim.x;
    for (long long pp = p0; pp < p1;) {
        // a run: this workgroup's passes inside one image
        const int n = (int)(pp / PPI), j0 = (int)(pp - (long long)n * PPI);
        int np = PPI - j0;
        if (np > p1 - pp) np = (int)(p1 - pp);
        pp += np;
        const int y0 = j0 * P;
        const long long img_row0 = (long long)n * H;
        wait_all_but_then_barrier<0>();                              // everyone is past the reads of the run before: the ring is free
        stage(img_row0, y0 - 1, 2, -1);                              // the row above the run and its first row
        stage_group(img_row0, y0, 0);
        if (np > 1) stage_group(img_row0, y0, 1);
        f32x4 accA[2][TPWV], accB[2][TPWV];
        // pass 0: everything but group 1
        top_of_pass(np > 1, false);
        if (2 < np) stage_group(img_row0, y0, 2);
        compute(accA, 0);
#pragma unroll 1
        for (int k = 1; k < np; k += 2) {
            top_of_pass(k + 1 < np, k >= 2);
            if (k + 2 < np) stage_group(img_row0, y0, k + 2);
            compute(accB, (k & 3) * P);
            epilogue(accA, y0 + P * (k - 1), img_row0);
            if (k + 1 < np) {
                top_of_pass(k + 2 < np, true);
                if (k + 3 < np) stage_group(img_row0, y0, k + 3);
                compute(accA, ((k + 1) & 3) * P);
                epilogue(accB, y0 + P * k, img_row0);
            } else {
                epilogue(accB, y0 + P * k, img_row0);               // an even number of passes: the last went to accB
                goto run_done;
            }
        }
        epilogue(accA, y0 + P * (np - 1), img_row0);                // an odd number of passes
    run_done:;
    }
}

bool g_use_r56 = true;

}  // namespace

extern "C" void alink_debug_set_r56(int on) { g_use_r56 = on != 0; }

// 23: 56 x 56 x 64 -> 64, 24: 56 x 56 x 64 -> 128 (0 = not applicable)
int r56_variant(int ksz, int stride, int pad, int H_, int W_, int Cin, int Cout) {
    if (!g_use_r56 || ksz != 3 || stride != 1 || pad != 1 || Cin != CIN || H_ != H || W_ != W) return 0;
    return Cout == 64 ? 23 : (Cout == 128 ? 24 : 0);
}

template <typename T, int COUT>
static hipError_t r56_attr() {
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)conv3x3_r56_kernel<T, COUT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G56<COUT>::lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)conv3x3_r56_kernel<T, COUT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G56<COUT>::lds_bytes())) != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)conv3x3_r56_kernel<T, COUT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G56<COUT>::lds_bytes());
}
hipError_t r56_set_attributes() {
    hipError_t e;
    if ((e = r56_attr<__bf16, 64>()) != hipSuccess || (e = r56_attr<__bf16, 128>()) != hipSuccess) return e;
    if ((e = r56_attr<_Float16, 64>()) != hipSuccess || (e = r56_attr<_Float16, 128>()) != hipSuccess) return e;
    return hipSuccess;
}

template <typename T, int COUT>
static void r56_launch(const ConvParams& p, hipStream_t st) {
    const long long npass = (long long)p.N * G56<COUT>::PPI;
    const unsigned grid = (unsigned)(npass < 256 ? npass : 256);        // one persistent workgroup per CU
    const size_t lds = G56<COUT>::lds_bytes();
    if (p.alpha && !p.resid && p.border_cls)       hipLaunchKernelGGL((conv3x3_r56_kernel<T, COUT, 1>), dim3(grid), dim3(NT), lds, st, p);
    else if (!p.alpha && p.resid && !p.border_cls) hipLaunchKernelGGL((conv3x3_r56_kernel<T, COUT, 2>), dim3(grid), dim3(NT), lds, st, p);
    else                                            hipLaunchKernelGGL((conv3x3_r56_kernel<T, COUT, 0>), dim3(grid), dim3(NT), lds, st, p);
}

hipError_t launch_conv3x3_r56(int variant, int dtype, const ConvParams& p, hipStream_t st) {
    if ((variant != 23 && variant != 24) || p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.Cin != CIN || p.H != H || p.W != W) return hipErrorInvalidValue;
    if (p.Cout != (variant == 23 ? 64 : 128)) return hipErrorInvalidValue;
    if (p.splitk != 1 || p.dact || p.post_relu || p.in2 || p.N <= 0) return hipErrorInvalidValue;   // forward forms only
    if ((long long)p.N * H * W * p.Cout >= (1ll << 31)) return hipErrorInvalidValue;
    if (dtype != ALINK_DT_BF16 && dtype != ALINK_DT_F16) return hipErrorInvalidValue;
    if (dtype == ALINK_DT_BF16) { if (variant == 23) r56_launch<__bf16, 64>(p, st); else r56_launch<__bf16, 128>(p, st); }
    else                        { if (variant == 23) r56_launch<_Float16, 64>(p, st); else r56_launch<_Float16, 128>(p, st); }
    return hipGetLastError();
}

}  // namespace alink
