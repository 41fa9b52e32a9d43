// conv3x3_s2c64.hip — the DIRECT stride-2 3x3 convolution of the first residual unit (112 x 112 x 64 -> 56 x 56 x 64) with the
// unit's 1 x 1 stride-2 projection shortcut as two more K-steps: the rolling-row kernel of conv3x3_c64.hip at stride 2.
//
// On the implicit-GEMM kernel this launch stages a 32 KB pixel tile per tap — every input pixel travels from L2 into LDS
// 2.25 times, 40 LDS-DMA pieces per 64-channel K-step against 32 MFMAs a wave — and reads its 469 MB input at 3.3 TB/s,
// bound by neither HBM nor MFMA.  Here:
//
//   * weights in registers: the wave's 32 output channels x (9 x 64 + 64 shortcut) = 160 registers per lane, loaded once;
//   * persistent workgroups, one per CU, each with a CONTIGUOUS range of output rows; a pass = one output row of 56 pixels
//     = 4 MFMA tiles (the last one half empty), the four waves = 2 tile pairs x 2 channel halves;
//   * the input rows live in a ring of 8 slots in LDS, every row fetched ONCE by LDS-DMA two passes ahead (two new rows per
//     pass), DE-INTERLEAVED on the way in: a slot holds the row's odd pixels (with a zero pixel in front: the left padding)
//     and its even pixels as two planes, so that the 16 output pixels of a tile read 16 CONSECUTIVE positions of one plane
//     for every tap (kx = 0: odd plane at x - 1, kx = 1: even plane at x, kx = 2: odd plane at x) — the same swizzle and
//     lane permutation as everywhere, conflict-free ds_read_b128, addresses by immediates only; the row above the image
//     comes from the zero page (there is no row below: 2 y + 1 <= 111);
//   * the shortcut's operand (the stem's activation at (2 y, 2 x): a compact tensor when the fused front kernel wrote it) is
//     read straight from global memory into the MFMA operand registers, one pass ahead;
//   * one barrier per pass.
// Reference: insightface fresnet stage1_unit1 conv2 (stride 2) + conv1sc / sc, inside model.forward at
// /root/reference/code/face_model.py:90.
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

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int delta(int lr) { return lr < 4 ? 2 * lr : (lr < 12 ? 2 * (lr - 4) + 1 : 2 * (lr - 8)); }

template <int N>
__device__ __forceinline__ void wait_all_but_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int NT = 256;
constexpr int WI = 112, HI = 112, WO = 56, HO = 56, C = 64;
constexpr int RING = 8;                          // input row slots: the 3 a pass reads, the 2 + 2 of the next two passes
constexpr int ODDB = (WO + 8 + 1) * 128;         // odd plane: zero pixel, 56 pixels, 8 positions the empty half tile reads
constexpr int EVENB = (WO + 8) * 128;            // even plane
constexpr int PITCH = ODDB + EVENB;
constexpr int XBYTES = RING * PITCH;
constexpr int KR = 9 * C + C;                    // weight row pitch with the shortcut's columns
constexpr size_t lds_bytes() { return (size_t)XBYTES + C * 4; }

// SC: the projection shortcut as two more K-steps (p.in2); else an ordinary stride-2 layer: optional PReLU and residual
template <typename T, bool SC>
__global__ __launch_bounds__(NT, 1) void conv3x3_s2c64_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int th = wave >> 1, ch = wave & 1;           // tile pair (tiles 2 th, 2 th + 1), channel half
    const int q = lane >> 4, lr = lane & 15;

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- once per workgroup: the zero pixel in front of every odd plane, bias table, weights --------------------------------
    for (int i = tid; i < RING * 8; i += NT) *(uint4*)(smem + (i >> 3) * PITCH + (i & 7) * 16) = uint4{0u, 0u, 0u, 0u};
    float* const ebias = (float*)(smem + XBYTES);
    for (int i = tid; i < C; i += NT) ebias[i] = p.bias[i];
    const int wpitch = SC ? KR : 9 * C;
    vec8 wr[2][9][2];                                  // [channel tile][tap][K half]: rows perm32-permuted
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wr[ct][tap][ks] = *(const vec8*)(gw + (size_t)(ch * 32 + 16 * ct + lr) * wpitch + tap * 64 + ks * 32 + 8 * q);
    vec8 ws[2][2];                                     // the shortcut's columns
    if (SC) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) ws[ct][ks] = *(const vec8*)(gw + (size_t)(ch * 32 + 16 * ct + lr) * KR + 9 * C + ks * 32 + 8 * q);
    }

    // output pixel x = 16 u + d reads position x (kx = 0: odd plane, pixel 2 x - 1; kx = 1: even plane, pixel 2 x) or x + 1
    // (kx = 2: odd plane, pixel 2 x + 1) of its plane; (16 u) leaves the swizzle alone
    const int d = delta(lr);
    int loff[2][2];                                    // [position d / d + 1][K half]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) loff[e][ks] = (d + e) * 128 + ((((ks << 2) | q) ^ (((d + e) >> 1) & 7)) << 4);

    // ---- staging: one input row = 14 units of 8 plane positions (7 odd, 7 even), dealt round-robin to the waves --------------
    auto stage_rows = [&](long long in_row0, int r0, int count) __attribute__((always_inline)) {
        for (int j = wave; j < count * 14; j += 4) {
            const int ri = j / 14, jj = j - ri * 14, odd = jj < 7 ? 1 : 0, seg = odd ? jj : jj - 7;
            const int r = r0 + ri, slot = (r + 1) & (RING - 1);
            const int pj = seg * 8 + (lane >> 3);                    // index in the plane: pixel 2 pj + odd
            const int pos = pj + odd;                                // its position (the odd plane starts with the zero pixel)
            const int piece = (lane & 7) ^ ((pos >> 1) & 7);
            const bool ok = r >= 0;
            const T* src = ok ? gin + ((size_t)(in_row0 + r) * WI + 2 * pj + odd) * C + piece * 8 : gz + (lane & 7) * 8;
            dma16(src, smem + slot * PITCH + (odd ? 0 : ODDB) + (seg * 8 + odd) * 128);
        }
    };
    // the shortcut's operand for output row y: pixel 16 u + d (clamped into the row for the empty half tile), K half ks
    const T* __restrict__ gx = (const T*)p.in2;
    auto load_sc = [&](long long n, int y, vec8 (&xs)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int x = 16 * (2 * th + t) + d;
            x = x < WO ? x : WO - 1;
            const size_t pix = p.in2_compact ? ((size_t)n * HO + y) * WO + x : ((size_t)n * HI + 2 * y) * WI + 2 * x;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xs[t][ks] = *(const vec8*)(gx + pix * C + ks * 32 + 8 * q);
        }
    };
    // one pass: output row y of image rows in_row0 .., taps from the slots of input rows 2 y - 1, 2 y, 2 y + 1
    auto compute = [&](f32x4 (&acc)[2][2], int y, const vec8 (&xs)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        int sb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) sb[ky] = ((2 * y + ky) & (RING - 1)) * PITCH + 2 * th * 2048;      // slot of row 2 y + ky - 1
        auto frag = [&](int st, int t) __attribute__((always_inline)) -> vec8 {
            const int tap = st >> 1, ks = st & 1, ky = tap / 3, kx = tap % 3;
            return *(const vec8*)(smem + sb[ky] + loff[kx == 2 ? 1 : 0][ks] + (kx == 1 ? ODDB : 0) + 2048 * t);
        };
        vec8 pf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) pf[0][t] = frag(0, t);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) {
#pragma unroll
                for (int t = 0; t < 2; ++t) pf[(st + 1) & 1][t] = frag(st + 1, t);
            }
            const int tap = st >> 1, ks = st & 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[ct][t] = mfma16<T>(wr[ct][tap][ks], pf[st & 1][t], acc[ct][t]);
        }
        if (SC) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[ct][t] = mfma16<T>(ws[ct][ks], xs[t][ks], acc[ct][t]);
        }
    };
    // epilogue: bias (+ PReLU, + residual in the plain form); the lanes of the empty half tile store nothing (an ordinary
    // predicated store: buffer stores with out-of-range lanes were found unsafe under multi-stream load, front_c64.hip)
    auto epilogue = [&](const f32x4 (&acc)[2][2], long long n, int y) __attribute__((always_inline)) {
        const f32x4 b0 = *(const f32x4*)(ebias + ch * 32 + 8 * q), b1 = *(const f32x4*)(ebias + ch * 32 + 8 * q + 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int x = 16 * (2 * th + t) + d;
            const bool live = x < WO;
            const size_t el = (((size_t)n * HO + y) * WO + (live ? x : 0)) * C + ch * 32 + 8 * q;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc[0][t][j] + b0[j]; v[4 + j] = acc[1][t][j] + b1[j]; }
            if (!SC && p.alpha) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.alpha[ch * 32 + 8 * q + i];
            }
            if (!SC && p.resid) {
                const vec8 r8 = *(const vec8*)((const T*)p.resid + el);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += (float)r8[i];
            }
            vec8 o8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o8[i] = (T)v[i];
            if (live) *(vec8*)((T*)p.out + el) = o8;
        }
    };
    // Waits (a wave's vector-memory operations retire in issue order): pass k needs the rows requested two iterations earlier,
    // so "all but what the iteration before issued" — 7 row DMAs if it staged, 4 shortcut loads if it loaded, 2 stores (+ 2
    // residual loads at the most: not counted, so a wave then waits for two more than it must).
    auto top_of_pass = [&](bool staged, bool loaded, bool stored) __attribute__((always_inline)) {
        const int n = (staged ? 7 : 0) + (loaded ? 4 : 0) + (stored ? 2 : 0);
        switch (n) {
            case 13: wait_all_but_then_barrier<13>(); break;
            case 11: wait_all_but_then_barrier<11>(); break;
            case 9:  wait_all_but_then_barrier<9>(); break;
            case 7:  wait_all_but_then_barrier<7>(); break;
            case 6:  wait_all_but_then_barrier<6>(); break;
            case 4:  wait_all_but_then_barrier<4>(); break;
            case 2:  wait_all_but_then_barrier<2>(); break;
            default: wait_all_but_then_barrier<0>(); break;
        }
    };

    const long long nrows = (long long)p.N * HO;
    const long long r0 = nrows * blockIdx.x / gridDim.x, r1 = nrows * (blockIdx.x + 1) / gridDim.x;
    for (long long rr = r0; rr < r1;) {
        // a run: this workgroup's output rows inside one image
        const long long n = rr / HO;
        const int y0 = (int)(rr - n * HO);
        int np = HO - y0;
        if (np > r1 - rr) np = (int)(r1 - rr);
        rr += np;
        const long long in_row0 = n * HI;
        wait_all_but_then_barrier<0>();                              // everyone is past the reads of the run before
        stage_rows(in_row0, 2 * y0 - 1, 3);                          // pass 0's rows
        if (np > 1) stage_rows(in_row0, 2 * y0 + 2, 2);              // pass 1's
        vec8 xsA[2][2], xsB[2][2];
        if (SC) load_sc(n, y0, xsA);
        f32x4 acc[2][2];
#pragma unroll 1
        for (int k = 0; k < np; k += 2) {
            // pass k (shortcut operand in xsA), then pass k + 1 (xsB)
            if (k == 0) { if (np > 1) wait_all_but_then_barrier<7 + (SC ? 4 : 0)>(); else wait_all_but_then_barrier<SC ? 4 : 0>(); }
            else        top_of_pass(k + 1 < np, SC && k < np, true);
            if (k + 2 < np) stage_rows(in_row0, 2 * (y0 + k + 2), 2);
            if (SC && k + 1 < np) load_sc(n, y0 + k + 1, xsB);
            compute(acc, y0 + k, xsA);
            epilogue(acc, n, y0 + k);
            if (k + 1 < np) {
                top_of_pass(k + 2 < np, SC && k + 1 < np, true);
                if (k + 3 < np) stage_rows(in_row0, 2 * (y0 + k + 3), 2);
                if (SC && k + 2 < np) load_sc(n, y0 + k + 2, xsA);
                compute(acc, y0 + k + 1, xsB);
                epilogue(acc, n, y0 + k + 1);
            }
        }
    }
}

bool g_use_s2c64 = true;

}  // namespace

extern "C" void alink_debug_set_s2direct(int on) { g_use_s2c64 = on != 0; }

// 25: the direct stride-2 kernel for 112 x 112 x 64 -> 56 x 56 x 64 (0 = not applicable)
int s2c64_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout) {
    if (!g_use_s2c64 || ksz != 3 || stride != 2 || pad != 1 || Cin != C || Cout != C || H != HI || W != WI) return 0;
    return 25;
}

hipError_t s2c64_set_attributes() {
    hipError_t e;
#define A(T, SC) if ((e = hipFuncSetAttribute((const void*)conv3x3_s2c64_kernel<T, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes())) != hipSuccess) return e;
    A(__bf16, true) A(__bf16, false) A(_Float16, true) A(_Float16, false)
#undef A
    return hipSuccess;
}

hipError_t launch_conv3x3_s2c64(int variant, int dtype, const ConvParams& p, hipStream_t st) {
    if (variant != 25 || p.ksz != 3 || p.stride != 2 || p.pad != 1 || p.Cin != C || p.Cout != C || p.H != HI || p.W != WI) return hipErrorInvalidValue;
    if (p.splitk != 1 || p.dact || p.post_relu || p.border_cls || p.N <= 0) return hipErrorInvalidValue;
    if (p.in2 && (p.Cin2 != C || p.alpha || p.resid)) return hipErrorInvalidValue;
    if ((long long)p.N * HI * WI * C >= (1ll << 31)) return hipErrorInvalidValue;
    if (dtype != ALINK_DT_BF16 && dtype != ALINK_DT_F16) return hipErrorInvalidValue;
    const long long nrows = (long long)p.N * HO;
    const unsigned grid = (unsigned)(nrows < 256 ? nrows : 256);        // one persistent workgroup per CU
    if (dtype == ALINK_DT_BF16) {
        if (p.in2) hipLaunchKernelGGL((conv3x3_s2c64_kernel<__bf16, true>), dim3(grid), dim3(NT), lds_bytes(), st, p);
        else       hipLaunchKernelGGL((conv3x3_s2c64_kernel<__bf16, false>), dim3(grid), dim3(NT), lds_bytes(), st, p);
    } else {
        if (p.in2) hipLaunchKernelGGL((conv3x3_s2c64_kernel<_Float16, true>), dim3(grid), dim3(NT), lds_bytes(), st, p);
        else       hipLaunchKernelGGL((conv3x3_s2c64_kernel<_Float16, false>), dim3(grid), dim3(NT), lds_bytes(), st, p);
    }
    return hipGetLastError();
}

}  // namespace alink
