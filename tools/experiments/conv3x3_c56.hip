// conv3x3_c56.hip — 3x3 / stride 1 / pad 1 NHWC convolution for the 64 -> 64 channel layers at 56 x 56 (the stage-1 units):
// the ROLLING-ROW kernel with the WEIGHTS IN REGISTERS of conv3x3_c64.hip for rows that are three and a half MFMA tiles wide.
//
// conv3x3_c64.hip keeps every input row in an LDS slot with a zero pixel at either end and reads 16 consecutive positions per
// 16-pixel MFMA tile; a 56-pixel row does not divide into such tiles.  Here a tile is 8 pixels of row y and the 8 pixels
// below them in row y + 1: seven tiles per pair of rows, no empty column, and every operand address is still
// (per-lane constant) + (per-pass row slot) + immediate:
//
//   * lanes 0 - 7 of a 16-lane row hold the upper pixels, lanes 8 - 15 the lower ones, each half in a fixed permutation
//     (found by search over the bank model of MI355X_MICROARCH.md: with slot pitches that are multiples of 256 B every
//     ds_read_b128 lane group touches 16 distinct bank quads for every tap and tile — the two-row analogue of delta());
//   * the XOR swizzle of a position is (position >> 1) & 7 as everywhere; a tile starts at position 8 u + 1, so for odd u it
//     is the even tile's swizzle ^ 4 — the address of the OTHER K half: no second offset table;
//   * a lane's row slot is the pass's slot + its half: a per-lane select of two uniform slot bases, three per pass.
//
// Everything else as in conv3x3_c64.hip / front_c64.hip: one persistent workgroup per CU with a CONTIGUOUS range of 4-row
// passes (runs inside an image roll through a ring of 16 row slots, row r in slot (r + 1) & 15), the wave's 32 output channels
// x 576 weights in 144 registers, rows by LDS-DMA two passes ahead (every row fetched once), one barrier per pass, a pass's
// epilogue one pass late beside the next pass's MFMAs, waits by what the iteration before issued at the least.
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

// pixel (0..7) of an 8-pixel half tile handled by MFMA column lr: upper half (lr < 8) and lower half (lr >= 8) in permutations
// that keep every ds_read_b128 lane group on 16 distinct bank quads (tools/experiments/two_row_tile_banks.py)
__device__ __forceinline__ int xi_of(int lr) {
    return lr < 8 ? (int)((0x17350264u >> (4 * lr)) & 7u) : (int)((0x40261753u >> (4 * (lr - 8))) & 7u);
}

template <int N>
__device__ __forceinline__ void wait_all_but_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int NT = 256;
constexpr int W = 56, H = 56, C = 64;
constexpr int P = 4;                            // rows per pass: two pairs of rows = 14 tiles, 7 per pixel half of the waves
constexpr int PPI = H / P;                      // passes per image
constexpr int RING = 16;                        // row slots: the 6 a pass reads, the 4 + 4 of the next two
constexpr int PITCH = (W + 2) * 128;            // a slot: zero pixel, 56 pixels, zero pixel; a multiple of 256 B (the bank model above)
constexpr int XBYTES = RING * PITCH;
constexpr int TPWV = 7;
constexpr int DPG = P * (W / 8) / 4;            // row DMAs a wave issues per group of P rows: 28 units of 8 pixels over 4 waves
constexpr size_t lds_bytes() { return (size_t)XBYTES + 10 * C * 4; }
static_assert(PITCH % 256 == 0 && P * (W / 8) % 4 == 0, "slot pitch keeps the bank phase; whole DMA units per wave");

// EPI: 1 = bias by border class + PReLU (a unit's conv1), 2 = bias + residual (conv2), 0 = by run-time flags
template <typename T, int EPI>
__global__ __launch_bounds__(NT, 1) void conv3x3_c56_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, ch = wave & 1;           // pair of rows of the pass, channel half
    const int q = lane >> 4, lr = lane & 15;
    const int h = lr >> 3, xi = xi_of(lr);             // the lane's row of the pair, its pixel in the 8-pixel half tile

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- once per workgroup: zero pixels at both ends of every slot, epilogue tables, weights into registers -------------------
    for (int i = tid; i < RING * 16; i += NT) {
        const int slot = i >> 4, side = (i >> 3) & 1, piece = i & 7;
        *(uint4*)(smem + slot * PITCH + (side ? (W + 1) * 128 : 0) + piece * 16) = uint4{0u, 0u, 0u, 0u};
    }
    const bool has_alpha = EPI == 1 || (EPI == 0 && p.alpha), has_resid = EPI == 2 || (EPI == 0 && p.resid);
    const bool classes = EPI == 1 || (EPI == 0 && p.border_cls);
    const int ncls = classes ? 9 : 1;
    float* const ebias = (float*)(smem + XBYTES);
    float* const ealpha = ebias + 9 * C;
    for (int i = tid; i < ncls * C; i += NT) ebias[i] = p.bias[i];
    if (has_alpha)
        for (int i = tid; i < C; i += NT) ealpha[i] = p.alpha[i];

    vec8 wr[2][9][2];                                  // [channel tile][tap][K half]: rows perm32-permuted, K = [tap][64]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wr[ct][tap][ks] = *(const vec8*)(gw + (size_t)(ch * 32 + 16 * ct + lr) * 576 + tap * 64 + ks * 32 + 8 * q);

    // per-lane operand offsets inside a slot for an EVEN tile: pixel x = 8 u + xi at position x + 1 + (kx - 1) = 8 u + xi + kx;
    // (8 u) adds 4 u to the swizzle term: nothing for even u, ^ 4 for odd u = the other K half's offset
    int loff[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            loff[kx][ks] = (xi + kx) * 128 + ((((ks << 2) | q) ^ (((xi + kx) >> 1) & 7)) << 4);

    // ---- staging: image rows r0 .. r0 + count - 1 into their slots; a unit = 8 pixels of one row (1 KB), dealt round-robin -----
    auto stage_rows = [&](long long img_row0, int r0, int count) __attribute__((always_inline)) {
        for (int j = wave; j < count * (W / 8); j += 4) {
            const int ri = j / (W / 8), seg = j - ri * (W / 8);
            const int r = r0 + ri, slot = (r + 1) & (RING - 1);
            const int px = seg * 8 + (lane >> 3);
            const int piece = (lane & 7) ^ (((px + 1) >> 1) & 7);
            const bool ok = (unsigned)r < (unsigned)H;
            const T* src = ok ? gin + ((size_t)(img_row0 + r) * W + px) * C + piece * 8 : gz + (lane & 7) * 8;
            dma16(src, smem + slot * PITCH + (seg * 8 + 1) * 128);
        }
    };

    // one pass of MFMAs: the wave's pair of output rows y + 2 ph, y + 2 ph + 1 (y the pass's first row); K order [tap][K half]
    auto compute = [&](f32x4 (&acc)[2][TPWV], int y) __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int u = 0; u < TPWV; ++u) acc[ct][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        int sb[3];                                     // the lane's slot base for ky = 0, 1, 2: row (y + 2 ph + h) + ky - 1
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sa = ((y + 2 * ph + ky) & (RING - 1)) * PITCH, sbb = ((y + 2 * ph + 1 + ky) & (RING - 1)) * PITCH;
            sb[ky] = h ? sbb : sa;
        }
        auto frag = [&](int st, int u) __attribute__((always_inline)) -> vec8 {
            const int tap = st >> 1, ks = st & 1, ky = tap / 3, kx = tap % 3;
            return *(const vec8*)(smem + (sb[ky] + loff[kx][(u & 1) ? ks ^ 1 : ks]) + 1024 * u);
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
    // epilogue of the pass whose first row is image row y: the lane's pixels (y + 2 ph + h, 8 u + xi), channels 32 ch + 8 q .. + 7
    auto epilogue = [&](const f32x4 (&acc)[2][TPWV], int y, long long img_row0) __attribute__((always_inline)) {
        const int yy = y + 2 * ph + h;
        const size_t pix0 = (size_t)(img_row0 + yy) * W + xi;
        const size_t choff = (size_t)ch * 32 + 8 * q;
        vec8 res[TPWV];
        if (has_resid) {
#pragma unroll
            for (int u = 0; u < TPWV; ++u) res[u] = *(const vec8*)((const T*)p.resid + (pix0 + 8 * u) * C + choff);
        }
        f32x4 al0, al1, b0, b1;
        if (has_alpha) { al0 = *(const f32x4*)(ealpha + choff); al1 = *(const f32x4*)(ealpha + choff + 4); }
        if (!classes) { b0 = *(const f32x4*)(ebias + choff); b1 = *(const f32x4*)(ebias + choff + 4); }
        const int rc = yy == 0 ? 0 : (yy == H - 1 ? 2 : 1);
#pragma unroll
        for (int u = 0; u < TPWV; ++u) {
            if (classes) {
                const int x = 8 * u + xi;
                const int cls = rc * 3 + (x == 0 ? 0 : (x == W - 1 ? 2 : 1));
                b0 = *(const f32x4*)(ebias + cls * C + choff);
                b1 = *(const f32x4*)(ebias + cls * C + choff + 4);
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
            *(vec8*)((T*)p.out + (pix0 + 8 * u) * C + choff) = o8;
        }
    };
    // Waits.  A wave's vector-memory operations retire in issue order (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)"); pass k needs
    // the rows requested two iterations earlier, so "all but what the iteration before issued AT THE LEAST" is enough and never
    // too little: DPG row DMAs if it staged, then — behind them in program order — 7 stores (+ 7 residual loads) if it ran an
    // epilogue.
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
        stage_rows(img_row0, y0 - 1, P + 2);                         // pass 0's rows
        if (np > 1) stage_rows(img_row0, y0 + P + 1, P);             // pass 1's
        f32x4 accA[2][TPWV], accB[2][TPWV];
        top_of_pass(np > 1, false);                                  // pass 0: everything but pass 1's rows
        if (2 < np) stage_rows(img_row0, y0 + 2 * P + 1, P);
        compute(accA, y0);
#pragma unroll 1
        for (int k = 1; k < np; k += 2) {
            top_of_pass(k + 1 < np, k >= 2);
            if (k + 2 < np) stage_rows(img_row0, y0 + (k + 2) * P + 1, P);
            compute(accB, y0 + P * k);
            epilogue(accA, y0 + P * (k - 1), img_row0);
            if (k + 1 < np) {
                top_of_pass(k + 2 < np, true);
                if (k + 3 < np) stage_rows(img_row0, y0 + (k + 3) * P + 1, P);
                compute(accA, y0 + P * (k + 1));
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

bool g_use_c56 = true;

}  // namespace

extern "C" void alink_debug_set_c56(int on) { g_use_c56 = on != 0; }

// 23: the rolling-row kernel for 56 x 56 x 64 -> 64 (0 = not applicable)
int c56_variant(int ksz, int stride, int pad, int H_, int W_, int Cin, int Cout) {
    if (!g_use_c56 || ksz != 3 || stride != 1 || pad != 1 || Cin != C || Cout != C || H_ != H || W_ != W) return 0;
    return 23;
}

hipError_t c56_set_attributes() {
    hipError_t e;
#define A(T, E) if ((e = hipFuncSetAttribute((const void*)conv3x3_c56_kernel<T, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes())) != hipSuccess) return e;
    A(__bf16, 0) A(__bf16, 1) A(__bf16, 2) A(_Float16, 0) A(_Float16, 1) A(_Float16, 2)
#undef A
    return hipSuccess;
}

template <typename T>
static void c56_launch(const ConvParams& p, hipStream_t st) {
    const long long npass = (long long)p.N * PPI;
    const unsigned grid = (unsigned)(npass < 256 ? npass : 256);        // one persistent workgroup per CU
    if (p.alpha && !p.resid && p.border_cls)       hipLaunchKernelGGL((conv3x3_c56_kernel<T, 1>), dim3(grid), dim3(NT), lds_bytes(), st, p);
    else if (!p.alpha && p.resid && !p.border_cls) hipLaunchKernelGGL((conv3x3_c56_kernel<T, 2>), dim3(grid), dim3(NT), lds_bytes(), st, p);
    else                                            hipLaunchKernelGGL((conv3x3_c56_kernel<T, 0>), dim3(grid), dim3(NT), lds_bytes(), st, p);
}

hipError_t launch_conv3x3_c56(int variant, int dtype, const ConvParams& p, hipStream_t st) {
    if (variant != 23 || p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.Cin != C || p.Cout != C || p.H != H || p.W != W) return hipErrorInvalidValue;
    if (p.splitk != 1 || p.dact || p.post_relu || p.in2 || p.N <= 0) return hipErrorInvalidValue;   // forward forms only
    if ((long long)p.N * H * W * C >= (1ll << 31)) return hipErrorInvalidValue;
    if (dtype == ALINK_DT_BF16) c56_launch<__bf16>(p, st);
    else if (dtype == ALINK_DT_F16) c56_launch<_Float16>(p, st);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace alink
