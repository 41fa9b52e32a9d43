// conv3x3_rs.hip — 3x3 / stride 1 / pad 1 NHWC convolution for the 128 -> 128 channel layers at 28 x 28 (the stage-2
// units), REGISTER-STATIONARY: the whole folded weight tensor lives in the registers of the CU for the length of the launch.
//
// The tile kernels (conv3x3_linear / conv3x3_direct) stream the weights of every 224-pixel workgroup from L2 through LDS
// — 295 KB per 57 KB of output at this shape, 73 bytes of LDS-DMA per MFMA — and a third of a workgroup's life is prologue
// and epilogue around an 18-step K walk.  Here, as in conv3x3_c64.hip:
//
//   * 128 x 1152 weights = 295 KB = the wave's 32 output channels x 1152 x 2 B = 288 registers per lane, loaded once; the
//     four waves of the one workgroup per CU are the four channel quarters and share every pixel operand;
//   * a workgroup is persistent and owns a CONTIGUOUS range of passes (4 image rows = 112 pixels = seven 16-pixel MFMA
//     tiles that run across row ends); inside an image it rolls: the input rows sit in a ring of 16 row slots in LDS,
//     fetched by LDS-DMA two passes ahead, every row once;
//   * the ring is LINEAR — position = ring row x 28 + x, no padding pixels — so a tile's operand for any tap is 16
//     consecutive positions (delta() lane permutation, XOR swizzle: conflict-free ds_read_b128).  What padding would do is
//     done by address, as in conv3x3_linear.hip: a lane whose tap crosses the left / right image border adds a bit far
//     above the allocation and reads zeros; rows above / below the image are fetched from the zero page.  A pass reads one
//     row before and one after its four; so that this window never wraps, ring row 15 is mirrored in front of row 0 and
//     row 0 behind row 15 (two more slots, the same rows DMA'd twice);
//   * one barrier per pass = per 504 MFMAs of a wave.
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

template <int A>
struct IC { static constexpr int a = A; };

constexpr int NT = 256;
constexpr int W = 28, H = 28, C = 128;
constexpr int P = 4;                            // rows per pass: 112 pixels = 7 tiles
constexpr int PPI = H / P;                      // passes per image
constexpr int R = 16;                           // ring rows: the 6 a pass reads (4 + one either side), the 4 + 4 of the next two
constexpr int PXB = 2 * C;                      // bytes per position: two 64-channel halves of 128 B
constexpr int RINGPX = (R + 2) * W;             // [copy of row R-1][rows 0 .. R-1][copy of row 0]
constexpr int LEAD = 16 * PXB;                  // unused positions in front of the ring: a lane's base address never goes negative
constexpr int XBYTES = LEAD + RINGPX * PXB;
constexpr int TBYTES = 10 * C * 4;              // bias classes + PReLU slopes
constexpr int FAR = 1 << 18;                    // beyond the allocation: a DS read there returns zero
constexpr int TPWV = 7;
constexpr int UPP = P * W / 4;                  // DMA units (4 positions = 1 KB) per pass: 28, seven per wave; a row is 7
constexpr size_t lds_bytes() { return (size_t)XBYTES + TBYTES; }
static_assert(W % 4 == 0 && (P * W) % 16 == 0 && H % P == 0, "whole DMA units per row, whole tiles per pass, whole passes per image");
static_assert(((R / P) & (R / P - 1)) == 0, "the ring holds a power-of-two number of passes");

// EPI: 1 = bias by border class + PReLU (a unit's conv1), 2 = bias + residual (conv2)
template <typename T, int EPI>
__global__ __launch_bounds__(NT, 1) void conv3x3_rs128_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int cq = __builtin_amdgcn_readfirstlane(tid >> 6);       // the wave's channel quarter
    const int q = lane >> 4, lr = lane & 15;

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- once per workgroup: epilogue tables, weights into registers ----------------------------------------------------
    const int ncls = EPI == 1 ? 9 : 1;
    float* const ebias = (float*)(smem + XBYTES);
    float* const ealpha = ebias + 9 * C;
    for (int i = tid; i < ncls * C; i += NT) ebias[i] = p.bias[i];
    if (EPI == 1)
        for (int i = tid; i < C; i += NT) ealpha[i] = p.alpha[i];

    vec8 wr[2][2][9][2];                               // [channel tile][input half][tap][K half]; rows perm32, K = [half][tap][64]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wr[ct][cc][tap][ks] = *(const vec8*)(gw + (size_t)(cq * 32 + 16 * ct + lr) * (9 * C) + (cc * 9 + tap) * 64 + ks * 32 + 8 * q);

    // A pass's output pixel o = 16 u + d sits at ring position (s0 + 1) W + o, s0 (a multiple of 4) the ring row of the pass's
    // first row: = 12 mod 16 + o, so the swizzle term of a tap's operand does not depend on the pass or the tile.
    const int d = delta(lr);
    int toff[9];                                       // K half 0; K half 1 is the same address ^ 64
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int sh = (tap / 3 - 1) * W + tap % 3 - 1;
        const int sw = ((12 + 7 * 16 + d + sh) >> 1) & 7;
        toff[tap] = (d + sh) * PXB + ((q ^ sw) << 4);
    }
    // per tile: is the lane's pixel in the first / last column (bits u, 8 + u), which of the pass's rows is it in (2 bits at 16 + 2 u)
    unsigned bits = 0;
#pragma unroll
    for (int u = 0; u < TPWV; ++u) {
        const int o = 16 * u + d, ro = o / W, x = o - ro * W;
        bits |= (x == 0 ? 1u : 0u) << u | (x == W - 1 ? 1u : 0u) << (8 + u) | (unsigned)ro << (16 + 2 * u);
    }

    // ---- staging: image rows r0 .. r0 + nrows - 1 into ring rows g0 .. (g0 = -1: the slot in front of ring row 0); a unit = 4
    // consecutive positions, both halves ------------------------------------------------------------------------------------
    auto stage = [&](long long img_row0, int r0, int nrows, int g0) __attribute__((always_inline)) {
        const int sub = lane >> 4, slot16 = lane & 15;                 // position in the unit, 16-B slot in the position
        for (int j = cq; j < nrows * (W / 4); j += 4) {
            const int ri = j / (W / 4), r = r0 + ri;
            const int pos = (g0 + 1) * W + 4 * j + sub;
            const int src16 = (slot16 & 8) | ((slot16 & 7) ^ ((pos >> 1) & 7));
            const bool ok = (unsigned)r < (unsigned)H;
            const T* src = ok ? gin + ((size_t)(img_row0 + r0) * W + 4 * j + sub) * C + src16 * 8 : gz + (lane & 7) * 8;
            dma16(src, smem + LEAD + ((g0 + 1) * W + 4 * j) * PXB);
        }
    };
    // one image row r again at ring row gm (a mirror slot, or the true slot of a row staged into the mirror slot)
    auto mirror = [&](long long img_row0, int r, int gm) __attribute__((always_inline)) {
        const int sub = lane >> 4, slot16 = lane & 15;
        for (int j = cq; j < W / 4; j += 4) {
            const int pos = (gm + 1) * W + 4 * j + sub;
            const int src16 = (slot16 & 8) | ((slot16 & 7) ^ ((pos >> 1) & 7));
            const bool ok = (unsigned)r < (unsigned)H;
            const T* src = ok ? gin + ((size_t)(img_row0 + r) * W + 4 * j + sub) * C + src16 * 8 : gz + (lane & 7) * 8;
            dma16(src, smem + LEAD + ((gm + 1) * W + 4 * j) * PXB);
        }
    };
    // group m of a run = the four rows BEHIND pass m's first row (rows y0 + 4 m + 1 .. + 4: the last is the row below pass m),
    // into ring rows 4 (m % 4) + 1 ..; the fourth group runs into the slot behind ring row 15, so its last row is also
    // staged at ring row 0, and ring row 15 (its third) in the slot in front of ring row 0
    auto stage_group = [&](long long img_row0, int y0, int m) __attribute__((always_inline)) {
        const int g0 = (m & (R / P - 1)) * P + 1;
        stage(img_row0, y0 + P * m + 1, P, g0);
        if (g0 == R - P + 1) {
            mirror(img_row0, y0 + P * m + P, 0);
            mirror(img_row0, y0 + P * m + P - 1, -1);
        }
    };

    // A pass in two halves of tiles (4 + 3), each with its own accumulators and its own walk of K: 288 of the 512 registers
    // are weights, and seven tiles' accumulators + two operand sets beside them leave the compiler no room to keep the operand
    // reads ahead of the MFMAs (it then reads, waits and multiplies tile by tile: 25 k cycles per pass instead of 8 k).  The
    // second half's MFMAs also cover the first half's epilogue.
    // the operand of (tile u, step): K order [tap][input half][K half]
    auto frag = [&](int sbase, int st, int u) __attribute__((always_inline)) -> vec8 {
        const int tap = st >> 2, cc = (st >> 1) & 1, ks = st & 1, kx = tap % 3;
        int a = (toff[tap] + sbase) ^ (ks << 6);
        if (kx == 0) a += (int)((bits >> u) & 1u) << 18;
        if (kx == 2) a += (int)((bits >> (8 + u)) & 1u) << 18;
        // (recomputed at every use — two or three VALU instructions beside eight MFMA slots; without this the compiler
        // keeps the ~100 distinct addresses of a pass in registers the kernel does not have)
        asm volatile("" : "+v"(a));
        return *(const vec8*)(smem + a + u * 16 * PXB + cc * 128);
    };
    auto compute = [&](auto U0, auto NU, f32x4 (&acc)[2][4], int s0) __attribute__((always_inline)) {
        constexpr int u0 = decltype(U0)::a, nu = decltype(NU)::a;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int u = 0; u < nu; ++u) acc[ct][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int sbase = LEAD + (s0 + 1) * W * PXB;
        vec8 pf[2][nu];
#pragma unroll
        for (int u = 0; u < nu; ++u) pf[0][u] = frag(sbase, 0, u0 + u);
        __builtin_amdgcn_sched_group_barrier(0x100, nu, 0);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            if (st + 1 < 36) {
#pragma unroll
                for (int u = 0; u < nu; ++u) pf[(st + 1) & 1][u] = frag(sbase, st + 1, u0 + u);
            }
            const int tap = st >> 2, cc = (st >> 1) & 1, ks = st & 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int u = 0; u < nu; ++u) acc[ct][u] = mfma16<T>(wr[ct][cc][tap][ks], pf[st & 1][u], acc[ct][u]);
            // the order asked of the scheduler: the next step's operand reads, THEN this step's MFMAs (left alone it sinks every
            // read to just before its two MFMAs — one operand register set, a full LDS latency per tile)
            if (st + 1 < 36) __builtin_amdgcn_sched_group_barrier(0x100, nu, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * nu, 0);
        }
    };
    // epilogue of tiles u0 .. of a pass whose first row is image row y: pixel o = 16 u + d, channels 32 cq + 8 q .. + 7
    auto epilogue = [&](auto U0, auto NU, const f32x4 (&acc)[2][4], int y, long long img_row0) __attribute__((always_inline)) {
        constexpr int u0 = decltype(U0)::a, nu = decltype(NU)::a;
        const size_t pix0 = (size_t)(img_row0 + y) * W + d;
        const size_t choff = (size_t)cq * 32 + 8 * q;
        vec8 res[nu];
        if (EPI == 2) {
#pragma unroll
            for (int u = 0; u < nu; ++u) res[u] = *(const vec8*)((const T*)p.resid + (pix0 + 16 * (u0 + u)) * C + choff);
        }
        f32x4 al0, al1, b0, b1;
        if (EPI == 1) { al0 = *(const f32x4*)(ealpha + choff); al1 = *(const f32x4*)(ealpha + choff + 4); }
        if (EPI == 2) { b0 = *(const f32x4*)(ebias + choff); b1 = *(const f32x4*)(ebias + choff + 4); }
#pragma unroll
        for (int i = 0; i < nu; ++i) {
            const int u = u0 + i;
            if (EPI == 1) {
                const int yy = y + (int)((bits >> (16 + 2 * u)) & 3u);
                const int rc = yy == 0 ? 0 : (yy == H - 1 ? 2 : 1);
                const int cx = (bits >> u) & 1u ? 0 : ((bits >> (8 + u)) & 1u ? 2 : 1);
                const int cls = rc * 3 + cx;
                b0 = *(const f32x4*)(ebias + cls * C + choff);
                b1 = *(const f32x4*)(ebias + cls * C + choff + 4);
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc[0][i][j] + b0[j]; v[4 + j] = acc[1][i][j] + b1[j]; }
            if (EPI == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * (k < 4 ? al0[k & 3] : al1[k & 3]);
            }
            if (EPI == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += (float)res[i][k];
            }
            vec8 o8;
#pragma unroll
            for (int k = 0; k < 8; ++k) o8[k] = (T)v[k];
            *(vec8*)((T*)p.out + (pix0 + 16 * u) * C + choff) = o8;
        }
    };
    // what a wave issues per steady-state pass at the least: 7 row DMAs + 7 stores (+ 7 residual loads); the mirror DMAs of
    // two passes in four come on top — waiting for "all but this many" then also waits for a few of the newest, never too few
    constexpr int VMI = TPWV * (EPI == 2 ? 3 : 2);

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
        // everyone is past the reads of the run before: the ring is free
        wait_all_but_then_barrier<0>();
        stage(img_row0, y0 - 1, 2, -1);                              // the row above the run and its first row
        stage_group(img_row0, y0, 0);
        if (np > 1) stage_group(img_row0, y0, 1);
#pragma unroll 1
        for (int k = 0; k < np; ++k) {
            // group k has landed: everything but what the iteration before issued — DMAs first, then its stores and residual
            // loads — (the first pass: but group 1's seven DMAs per wave)
            if (k == 0) {
                if (np > 1) wait_all_but_then_barrier<TPWV>();
                else        wait_all_but_then_barrier<0>();
            } else {
                wait_all_but_then_barrier<VMI>();
            }
            if (k + 2 < np) stage_group(img_row0, y0, k + 2);
            const int s0 = (k & (R / P - 1)) * P;
            f32x4 accA[2][4], accB[2][4];
            compute(IC<0>{}, IC<4>{}, accA, s0);
            epilogue(IC<0>{}, IC<4>{}, accA, y0 + P * k, img_row0);
            compute(IC<4>{}, IC<3>{}, accB, s0);
            epilogue(IC<4>{}, IC<3>{}, accB, y0 + P * k, img_row0);
        }
    }
}

bool g_use_rs = true;

}  // namespace

extern "C" void alink_debug_set_rs(int on) { g_use_rs = on != 0; }

// 22: the register-stationary kernel for 28 x 28 x 128 -> 128 (0 = not applicable)
int rs_variant(int ksz, int stride, int pad, int H_, int W_, int Cin, int Cout) {
    if (!g_use_rs || ksz != 3 || stride != 1 || pad != 1 || Cin != C || Cout != C || H_ != H || W_ != W) return 0;
    return 22;
}

template <typename T, int EPI>
static hipError_t rs_attr() {
    return hipFuncSetAttribute((const void*)conv3x3_rs128_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes());
}
hipError_t rs_set_attributes() {
    hipError_t e;
    if ((e = rs_attr<__bf16, 1>()) != hipSuccess || (e = rs_attr<__bf16, 2>()) != hipSuccess) return e;
    if ((e = rs_attr<_Float16, 1>()) != hipSuccess || (e = rs_attr<_Float16, 2>()) != hipSuccess) return e;
    return hipSuccess;
}

// the two forms the forward pass uses; anything else (run-time epilogue flags, split sums, the gradient pass) stays on the tile kernels
bool rs_takes(const ConvParams& p) {
    if (p.splitk != 1 || p.dact || p.post_relu || p.in2 || p.ablate || p.stamps) return false;
    const bool e1 = p.alpha && !p.resid && p.border_cls, e2 = !p.alpha && p.resid && !p.border_cls;
    return e1 || e2;
}

hipError_t launch_conv3x3_rs(int variant, int dtype, const ConvParams& p, hipStream_t st) {
    if (variant != 22 || p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.Cin != C || p.Cout != C || p.H != H || p.W != W) return hipErrorInvalidValue;
    if (!rs_takes(p) || p.N <= 0) return hipErrorInvalidValue;
    if ((long long)p.N * H * W * C >= (1ll << 31)) return hipErrorInvalidValue;
    if (dtype != ALINK_DT_BF16 && dtype != ALINK_DT_F16) return hipErrorInvalidValue;
    const long long npass = (long long)p.N * PPI;
    const unsigned grid = (unsigned)(npass < 256 ? npass : 256);        // one persistent workgroup per CU
    const bool e1 = p.alpha != nullptr;
    if (dtype == ALINK_DT_BF16) {
        if (e1) hipLaunchKernelGGL((conv3x3_rs128_kernel<__bf16, 1>), dim3(grid), dim3(NT), lds_bytes(), st, p);
        else    hipLaunchKernelGGL((conv3x3_rs128_kernel<__bf16, 2>), dim3(grid), dim3(NT), lds_bytes(), st, p);
    } else {
        if (e1) hipLaunchKernelGGL((conv3x3_rs128_kernel<_Float16, 1>), dim3(grid), dim3(NT), lds_bytes(), st, p);
        else    hipLaunchKernelGGL((conv3x3_rs128_kernel<_Float16, 2>), dim3(grid), dim3(NT), lds_bytes(), st, p);
    }
    return hipGetLastError();
}

}  // namespace alink
