// front_c64.hip — the FRONT of the IR-ResNet in one launch: stem (normalise, conv3x3 3 -> 64, BN, PReLU) and the first
// unit's conv1 (BN, conv3x3 64 -> 64 at 112 x 112, BN, PReLU), the stem's activation never leaving the CU.
//
// The stem writes 25.7 KB per image row of 112 pixels and conv1 reads it back: 2 x 469 MB per 292 images, two thirds of
// everything the two launches move, for a 27-MAC-per-output layer.  conv3x3_c64.hip already keeps the input rows of conv1 in
// a ring of row slots in LDS and fetches every row once; here the ring is FILLED BY COMPUTATION instead of by LDS-DMA:
//
//   * the raw pixel rows (336 values each) are loaded one pass ahead into registers, normalised and parked as T in a small
//     image ring in LDS — per pixel a 32-byte record holding its 3-pixel x 3-channel window of the row (every value is
//     parked three times), so that the stem's MFMA operand is two aligned 16-byte reads per lane and tile, no gather;
//   * each pass, every wave computes 7 tiles x 8 channels of one of the two stem rows the NEXT pass needs — the
//     arithmetic of stem_kernel (same K walk, same roundings: the result is bit-identical to the two-launch path) — and
//     writes them into the ring slot in the swizzled layout conv1's operand reads expect;
//   * the only other reader of the stem's activation, the projection shortcut of the unit (1 x 1, stride 2), gets the
//     quarter it samples as a compact [N][56][56][64] tensor `xs`, stored from the same registers;
//   * a workgroup owns a CONTIGUOUS range of 4-row bands and rolls through consecutive bands of one image without
//     restarting, so the 3-barrier prologue (load, park, two pairs of stem rows) is paid once per image it touches.
//
// What is left of the launch's HBM traffic is the pixels (11 MB as u8) and conv1's output.
// Reference: insightface fresnet conv0/bn0/relu0 + stage1_unit1 bn1/conv1/bn2/relu1, executed inside model.forward at
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

__device__ __forceinline__ int delta(int lr) { return lr < 4 ? 2 * lr : (lr < 12 ? 2 * (lr - 4) + 1 : 2 * (lr - 8)); }

// every LDS operation of this wave has completed (the stem rows and pixel rows it wrote, the operand reads of the pass
// before), then the workgroup barrier.  No vector-memory wait: nothing is DMA'd, loads land in registers (the compiler
// waits where they are used) and the output stores of the pass before may stay in flight.
__device__ __forceinline__ void lds_done_then_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int A>
struct IC { static constexpr int a = A; };

constexpr int NT = 256;
constexpr int W = 112, H = 112;
constexpr int P = 2;                            // output rows per pass: 14 pixel tiles
constexpr int RING = 8;                         // row slots of the stem activation (6 are live at any time)
constexpr int BR = 4;                           // rows per band = the unit the rows are dealt to workgroups in (two passes: the pass
                                                // loop runs in pairs); runs roll through consecutive bands, so a small unit costs nothing
                                                // and balances: 128 images = 14 bands per workgroup (3.5 with 16-row bands: 4 against 3)
constexpr int BPI = H / BR;                     // bands per image
constexpr int PITCH = (W + 2) * 128;            // a slot: zero pixel, W pixels, zero pixel
constexpr int XBYTES = RING * PITCH;
constexpr int TBYTES = 11 * 64 * 4;             // conv1's bias classes + PReLU slopes, the stem's PReLU slopes
constexpr int IMPX = 32;                        // image ring: a pixel's record = its 3 x 3-channel window of the row (9 values) + 7 zeros
constexpr int IMROWB = W * IMPX;
constexpr int IMROWS = 8;                       // pixel rows in the ring (row iy in slot (iy + 8) & 7; six are live at any time)
constexpr int TPWV = 7;                         // pixel tiles per wave and pass
constexpr size_t lds_bytes() { return (size_t)XBYTES + TBYTES + IMROWS * IMROWB + 16; }

// STAMP: diagnostic build (alink_debug_set_stamps; no product call runs it) — wave 0 sums the cycles it spends waiting at the
// pass barrier, in the stem rows and in conv1 + epilogue, 8 x u64 per workgroup
// AMAX: no PReLU slope of the two layers exceeds 1 (the launcher's caller checks the folded tensors), so that
// PReLU(v) = max(v, slope * v) — the same two values, one instruction fewer than compare + select.
template <typename T, int LAYOUT, bool AMAX, bool STAMP>
__global__ __launch_bounds__(NT, 1) void front_c64_kernel(const ConvParams p, const StemParams s, T* __restrict__ xs) {
    typedef typename Vec8<T>::type vec8;
    auto prelu = [](float v, float slope) __attribute__((always_inline)) {
        const float w = v * slope;
        return AMAX ? __builtin_fmaxf(v, w) : (v > 0.f ? v : w);
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const imring = smem + XBYTES + TBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, ch = wave & 1;          // conv1: output row of the pass, channel half; stem: row of the pair, half of a lane's 16 channels
    const int q = lane >> 4, lr = lane & 15;

    // ---- once per workgroup ------------------------------------------------------------------------------------------
    for (int i = tid; i < RING * 16; i += NT) {       // zero pixels at both ends of every slot
        const int slot = i >> 4, side = (i >> 3) & 1, piece = i & 7;
        *(uint4*)(smem + slot * PITCH + (side ? (W + 1) * 128 : 0) + piece * 16) = uint4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < IMROWS * IMROWB / 16; i += NT) // the image ring: the window elements beyond the left / right border and the pads are never written again
        *(uint4*)(imring + i * 16) = uint4{0u, 0u, 0u, 0u};
    float* const ebias = (float*)(smem + XBYTES);
    float* const ealpha = ebias + 9 * 64;
    for (int i = tid; i < 9 * 64; i += NT) ebias[i] = p.bias[i];
    float* const salpha = ealpha + 64;
    for (int i = tid; i < 64; i += NT) { ealpha[i] = p.alpha[i]; salpha[i] = s.alpha[i]; }

    const T* __restrict__ gw = (const T*)p.wgt;
    vec8 wr[2][9][2];                                  // conv1 weights: [channel tile][tap][K half]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wr[ct][tap][ks] = *(const vec8*)(gw + (size_t)(ch * 32 + 16 * ct + lr) * 576 + tap * 64 + ks * 32 + 8 * q);
    const int d = delta(lr);
    int loff[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            loff[kx][ks] = (d + kx) * 128 + ((((ks << 2) | q) ^ (((d + kx) >> 1) & 7)) << 4);

    // stem: weight rows 16 t + lr of tiles t = 2 ch, 2 ch + 1 (rows perm64-permuted: a lane's results are channels
    // 16 q + 8 ch .. + 7); K = 64 in two steps of 32: [ky 0: 9 window values + 7 zeros | ky 1: the same] and [ky 2 | zeros]
    // (stem_kernel walks the same K in the same two steps: the two paths agree bit for bit)
    vec8 swf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int st = 0; st < 2; ++st) swf[t][st] = *(const vec8*)((const T*)s.wgt + (16 * (2 * ch + t) + lr) * 64 + st * 32 + 8 * q);
    float sbi[8];                                      // (the slopes of both layers are read from their LDS tables where they are used:
#pragma unroll                                         //  16 registers this kernel does not have)
    for (int i = 0; i < 8; ++i) sbi[i] = s.bias[16 * q + 8 * ch + i];
    const int rec_off = lr * IMPX + (q & 1) * 16;     // the lane's half of the record of pixel x = lr
    const int sw_off = (lr + 1) * 128 + (((2 * q + ch) ^ (((lr + 1) >> 1) & 7)) << 4);   // ring slot offset of the lane's 16 B at x = lr

    // ---- pixel rows: a pair of rows is 672 values, up to three per thread; value jj of a thread is element e of row r of the
    // pair.  Fixed per thread: its offset in an image row, where it is parked, what is subtracted.  Nothing here branches
    // (a branch would cut the pass into blocks the scheduler cannot weave): loads of rows outside the image are clamped
    // into it and zeroed when parked, the spare threads of the third round read element 0 and park in a sink.
    // A value (ix, c) is parked three times: as window element kx * 3 + c of the records of pixels ix + 1 - kx.
    int goff[3], soff[3], prow[3];
    float ssub[3];
    {
        float sb0 = s.sub[0], sb1 = s.sub[1], sb2 = s.sub[2];
        asm volatile("" : "+s"(sb0), "+s"(sb1), "+s"(sb2));          // (three scalars, not an indexed copy of the array in scratch)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int i = tid + NT * jj;
            const bool valid = i < 672;
            const int r = i >= 336 ? 1 : 0, e = valid ? i - 336 * r : 0;
            int c, ix;
            if (LAYOUT == ALINK_LAYOUT_NCHW_F32) { c = e / W; ix = e - c * W; }
            else                                 { ix = e / 3; c = e - ix * 3; }
            const int cn = s.flip ? 2 - c : c;
            ssub[jj] = cn == 0 ? sb0 : (cn == 1 ? sb1 : sb2);
            goff[jj] = LAYOUT == ALINK_LAYOUT_NCHW_F32 ? c * H * W + ix : e;
            soff[jj] = valid ? (ix + 1) * IMPX + cn * 2 : -1;       // the kx = 0 copy; kx = 1: - IMPX + 6; kx = 2: - 2 IMPX + 12
            prow[jj] = r;
        }
    }
    constexpr int ROWP = LAYOUT == ALINK_LAYOUT_NCHW_F32 ? W : W * 3;          // elements from one image row to the next
    char* const sink = imring + IMROWS * IMROWB;                                 // 16 B nobody reads
    auto load_pair = [&](int n, int iy0, float (&raw)[3]) __attribute__((always_inline)) {
        const size_t base = (size_t)n * 3 * H * W;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            int iy = iy0 + prow[jj];
            iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
            const long long idx = (long long)base + iy * ROWP + goff[jj];
            if (LAYOUT == ALINK_LAYOUT_NHWC_U8) raw[jj] = (float)((const uint8_t*)s.in)[idx];
            else                                raw[jj] = ((const float*)s.in)[idx];
        }
    };
    // normalise and park the pair (rows iy0, iy0 + 1; iy0 even) in its ring rows
    auto store_pair = [&](int iy0, const float (&raw)[3]) __attribute__((always_inline)) {
        char* const rb0 = imring + ((iy0 + 8) & 7) * IMROWB;
        char* const rb1 = imring + ((iy0 + 9) & 7) * IMROWB;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const bool ok = (unsigned)(iy0 + prow[jj]) < (unsigned)H;
            const T v = (T)(ok ? (raw[jj] - ssub[jj]) * s.mul : 0.f);
            char* const dst = (prow[jj] ? rb1 : rb0) + soff[jj];
            *(T*)(soff[jj] >= 0 && soff[jj] < W * IMPX ? dst : sink) = v;
            *(T*)(soff[jj] >= 0 ? dst - IMPX + 6 : sink) = v;
            *(T*)(soff[jj] >= 2 * IMPX ? dst - 2 * IMPX + 12 : sink) = v;
        }
    };
    // the wave's row (R0 + ph) of a pair of stem rows, its 8 channels of every pixel, from the three pixel rows around it in
    // the image ring, into ring slot (slot0 + ph); rows outside the image are zero rows (conv1's padding).  Rows in
    // [own_lo, own_hi) also go to `xs` where the shortcut samples them (even row, even pixel).
    auto stem_rows = [&](int R0, int slot0, int n, int own_lo, int own_hi) __attribute__((always_inline)) {
        const int R = R0 + ph;
        int slot = slot0 + ph;
        if (slot >= RING) slot -= RING;
        const bool inimg = (unsigned)R < (unsigned)H;
        // first K step: lanes q = 0, 1 hold the halves of the record in pixel row R - 1, q = 2, 3 those in row R; second
        // step: q = 0, 1 row R + 1 (q = 2, 3 meet zero weights: they read the same, finite, values)
        const char* const rbm = imring + ((R + 7) & 7) * IMROWB;
        const char* const rbz = imring + ((R + 8) & 7) * IMROWB;
        const char* const rbp = imring + ((R + 9) & 7) * IMROWB;
        const char* const wa = (q < 2 ? rbm : rbz) + rec_off;
        const char* const wb = rbp + rec_off;
        char* const ob = smem + slot * PITCH + sw_off;
        // The quarter-resolution copy is stored by an ordinary predicated store.  (A buffer store whose unwanted lanes carry an
        // out-of-range offset — no branch in the block — was the first form, and WRONG on this part: with launches overlapping
        // on four streams a buffer_store_dwordx4 now and then sent what its data registers held ~15 instructions LATER (the
        // compiler reuses them at once: nothing on gfx9 counts a store's data reads), one pixel quad in ~10^5 images, never on
        // one stream; global stores in the same place: 0 in 300 repetitions.  profiles/experiments/r03_fused_front.txt.)
        const bool to_xs = inimg && !(R & 1) && R >= own_lo && R < own_hi && !(lr & 1);
        T* const xrow = xs + ((size_t)(n * (H / 2) + (R >> 1)) * (W / 2) + (lr >> 1)) * 64 + 16 * q + 8 * ch;
        // in two groups of tiles (4 + 3): the operand reads of a group, its MFMAs, its epilogues — the LDS and MFMA latencies are
        // waited out once per group, not per tile
        const int keep = inimg ? -1 : 0;
        const f32x4 sal0 = *(const f32x4*)(salpha + 16 * q + 8 * ch), sal1 = *(const f32x4*)(salpha + 16 * q + 8 * ch + 4);
        auto group = [&](auto X0, auto NX) __attribute__((always_inline)) {
            constexpr int x0 = decltype(X0)::a, nx = decltype(NX)::a;
            vec8 pa[nx], pb[nx];
#pragma unroll
            for (int i = 0; i < nx; ++i) {
                pa[i] = *(const vec8*)(wa + 16 * IMPX * (x0 + i));
                pb[i] = *(const vec8*)(wb + 16 * IMPX * (x0 + i));
            }
            f32x4 a0[nx], a1[nx];
#pragma unroll
            for (int i = 0; i < nx; ++i) {
                a0[i] = mfma16<T>(swf[0][0], pa[i], f32x4{sbi[0], sbi[1], sbi[2], sbi[3]});      // the bias is the start value
                a1[i] = mfma16<T>(swf[1][0], pa[i], f32x4{sbi[4], sbi[5], sbi[6], sbi[7]});
            }
#pragma unroll
            for (int i = 0; i < nx; ++i) {
                a0[i] = mfma16<T>(swf[0][1], pb[i], a0[i]);
                a1[i] = mfma16<T>(swf[1][1], pb[i], a1[i]);
            }
#pragma unroll
            for (int i = 0; i < nx; ++i) {
                vec8 o8;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v0 = a0[i][j], v1 = a1[i][j];
                    v0 = prelu(v0, sal0[j]);
                    v1 = prelu(v1, sal1[j]);
                    o8[j] = (T)v0;
                    o8[4 + j] = (T)v1;
                }
                i32x4 ob4 = __builtin_bit_cast(i32x4, o8);
#pragma unroll
                for (int j = 0; j < 4; ++j) ob4[j] &= keep;
                *(i32x4*)(ob + 2048 * (x0 + i)) = ob4;
                if (to_xs) *(i32x4*)(xrow + (size_t)(8 * (x0 + i)) * 64) = ob4;
            }
        };
        group(IC<0>{}, IC<4>{});
        group(IC<4>{}, IC<3>{});
    };

    // one pass of conv1's MFMAs (conv3x3_c64.hip): output row (first row of the pass) + ph, taps from the slots at s0
    auto compute = [&](f32x4 (&acc)[2][TPWV], int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int u = 0; u < TPWV; ++u) acc[ct][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        int sb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int slot = s0 + ph + ky;
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
    // conv1's epilogue: folded-BN bias by border class + PReLU, 8 consecutive channels per lane
    auto epilogue = [&](const f32x4 (&acc)[2][TPWV], int y, long long img_row0) __attribute__((always_inline)) {
        const int rc = y == 0 ? 0 : (y == H - 1 ? 2 : 1);
        const size_t rowoff = (size_t)((img_row0 + y) * W) * 64 + ch * 32 + 8 * q;
        const f32x4 al0 = *(const f32x4*)(ealpha + ch * 32 + 8 * q), al1 = *(const f32x4*)(ealpha + ch * 32 + 8 * q + 4);
#pragma unroll
        for (int u = 0; u < TPWV; ++u) {
            const int x = 16 * u + d;
            const int cls = rc * 3 + (x == 0 ? 0 : (x == W - 1 ? 2 : 1));
            const f32x4 b0 = *(const f32x4*)(ebias + cls * 64 + ch * 32 + 8 * q);
            const f32x4 b1 = *(const f32x4*)(ebias + cls * 64 + ch * 32 + 8 * q + 4);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc[0][u][j] + b0[j]; v[4 + j] = acc[1][u][j] + b1[j]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = prelu(v[i], i < 4 ? al0[i & 3] : al1[i & 3]);
            vec8 o8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o8[i] = (T)v[i];
            *(vec8*)((T*)p.out + rowoff + (size_t)x * 64) = o8;
        }
    };

    unsigned long long t_bar = 0, t_stem = 0, t_main = 0, t_pro = 0, t_a = 0, t_b = 0, t_c = 0, n_pass = 0;
    const unsigned long long t_start = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    const long long nbands = (long long)p.N * BPI;
    const long long b0 = nbands * blockIdx.x / gridDim.x, b1 = nbands * (blockIdx.x + 1) / gridDim.x;
    for (long long band = b0; band < b1;) {
        // a run: this workgroup's bands inside one image
        const int n = (int)(band / BPI), j0 = (int)(band - (long long)n * BPI);
        int rb = BPI - j0;
        if (rb > b1 - band) rb = (int)(b1 - band);
        band += rb;
        const int y0 = j0 * BR, npass = rb * (BR / P), yend = y0 + rb * BR;
        const long long img_row0 = (long long)n * H;

        // ---- prologue.  Pixel pairs I_j = rows y0 + 2 j + 2, + 3; stem pairs N_m = rows
        // y0 + 2 m + 1, + 2 need I_(m-1) and I_m; pass k reads N_(k-1), N_k and the row before.
        float raw[3];
        if (STAMP) t_a = __builtin_amdgcn_s_memtime();
        {
            float rawA[3], rawB[3], rawC[3];
            load_pair(n, y0 - 2, rawA);
            load_pair(n, y0, rawB);
            load_pair(n, y0 + 2, rawC);
            load_pair(n, y0 + 4, raw);
            lds_done_then_barrier();                                 // the run before is past its last reads
            store_pair(y0 - 2, rawA);                                // I_-2
            store_pair(y0, rawB);                                    // I_-1
            lds_done_then_barrier();
            store_pair(y0 + 2, rawC);                                // I_0
            stem_rows(y0 - 1, 0, n, y0, yend);                       // N_-1 from I_-2, I_-1 -> slots 0, 1
            lds_done_then_barrier();
            store_pair(y0 + 4, raw);                                 // I_1
            load_pair(n, y0 + 6, raw);                               // I_2, parked by pass 0
            stem_rows(y0 + 1, 2, n, y0, yend);                       // N_0 from I_-1, I_0 -> slots 2, 3
        }
        // ---- passes.  Pass k: park I_(k+2), request I_(k+3), stem rows N_(k+1), conv1 of rows y0 + 2 k + ph, and the
        // epilogue of the pass before (from the other accumulator set: straight-line code the compiler weaves in)
        int s0 = 0;                                                  // ring slot of row y0 + 2 k - 1
        f32x4 accA[2][TPWV], accB[2][TPWV];
        if (STAMP) t_pro += __builtin_amdgcn_s_memtime() - t_a;
        auto head = [&](int k) __attribute__((always_inline)) {
            if (STAMP) t_a = __builtin_amdgcn_s_memtime();
            lds_done_then_barrier();
            if (STAMP) { t_b = __builtin_amdgcn_s_memtime(); t_bar += t_b - t_a; }
            store_pair(y0 + 2 * k + 6, raw);
            load_pair(n, y0 + 2 * k + 8, raw);
            int sn = s0 + 4;
            if (sn >= RING) sn -= RING;
            stem_rows(y0 + 2 * k + 3, sn, n, y0, yend);
            if (STAMP) { t_c = __builtin_amdgcn_s_memtime(); t_stem += t_c - t_b; ++n_pass; }
            __builtin_amdgcn_sched_barrier(0);       // the stem rows stay ahead of conv1's block (the compiler weaves the epilogue of the pass before into that one)
        };
        auto advance = [&]() __attribute__((always_inline)) {
            if (STAMP) t_main += __builtin_amdgcn_s_memtime() - t_c;
            s0 += P;
            if (s0 >= RING) s0 -= RING;
        };
        head(0);
        compute(accA, s0);
        advance();
#pragma unroll 1
        for (int k = 1; k < npass; k += 2) {
            head(k);
            compute(accB, s0);
            epilogue(accA, y0 + (k - 1) * P + ph, img_row0);
            advance();
            if (k + 1 < npass) {
                head(k + 1);
                compute(accA, s0);
                epilogue(accB, y0 + k * P + ph, img_row0);
                    advance();
            } else {
                epilogue(accB, y0 + k * P + ph, img_row0);           // the run's last pass (npass is even)
            }
        }
    }
    if (STAMP && p.stamps && tid == 0) {
        unsigned long long* st = (unsigned long long*)p.stamps + (size_t)blockIdx.x * 8;
        st[0] = __builtin_amdgcn_s_memtime() - t_start; st[1] = t_bar; st[2] = t_stem; st[3] = t_main; st[4] = t_pro; st[5] = n_pass;
    }
}

bool g_fuse_stem = true;

}  // namespace

extern "C" void alink_debug_set_fuse_stem(int on) { g_fuse_stem = on != 0; }

// the front of the IR-ResNet (112 x 112 pixels, 64-channel stem, 64 -> 64 conv1 with PReLU and border classes) in one launch?
bool front_c64_applies(int dtype, int H_, int W_, int C0, int Cout) {
    return g_fuse_stem && (dtype == ALINK_DT_BF16 || dtype == ALINK_DT_F16) && H_ == H && W_ == W && C0 == 64 && Cout == 64;
}

template <typename T, int LAYOUT>
static hipError_t front_attr() {
    hipError_t e;
    if constexpr (LAYOUT == ALINK_LAYOUT_NHWC_U8) {
        e = hipFuncSetAttribute((const void*)front_c64_kernel<T, LAYOUT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes());
        if (e != hipSuccess) return e;
    }
    e = hipFuncSetAttribute((const void*)front_c64_kernel<T, LAYOUT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)front_c64_kernel<T, LAYOUT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes());
}
hipError_t front_c64_set_attributes() {
    hipError_t e;
    if ((e = front_attr<__bf16, ALINK_LAYOUT_NHWC_F32>()) != hipSuccess || (e = front_attr<__bf16, ALINK_LAYOUT_NCHW_F32>()) != hipSuccess ||
        (e = front_attr<__bf16, ALINK_LAYOUT_NHWC_U8>()) != hipSuccess)
        return e;
    if ((e = front_attr<_Float16, ALINK_LAYOUT_NHWC_F32>()) != hipSuccess || (e = front_attr<_Float16, ALINK_LAYOUT_NCHW_F32>()) != hipSuccess ||
        (e = front_attr<_Float16, ALINK_LAYOUT_NHWC_U8>()) != hipSuccess)
        return e;
    return hipSuccess;
}

template <typename T, int LAYOUT>
static void front_launch_l(const ConvParams& p, const StemParams& s, void* xs, bool amax, unsigned grid, hipStream_t st) {
    if (LAYOUT == ALINK_LAYOUT_NHWC_U8 && p.stamps && amax)
        hipLaunchKernelGGL((front_c64_kernel<T, ALINK_LAYOUT_NHWC_U8, true, true>), dim3(grid), dim3(NT), lds_bytes(), st, p, s, (T*)xs);
    else if (amax) hipLaunchKernelGGL((front_c64_kernel<T, LAYOUT, true, false>), dim3(grid), dim3(NT), lds_bytes(), st, p, s, (T*)xs);
    else           hipLaunchKernelGGL((front_c64_kernel<T, LAYOUT, false, false>), dim3(grid), dim3(NT), lds_bytes(), st, p, s, (T*)xs);
}
template <typename T>
static hipError_t front_launch(const ConvParams& p, const StemParams& s, void* xs, bool amax, unsigned grid, hipStream_t st) {
    switch (s.layout) {
        case ALINK_LAYOUT_NHWC_F32: front_launch_l<T, ALINK_LAYOUT_NHWC_F32>(p, s, xs, amax, grid, st); break;
        case ALINK_LAYOUT_NCHW_F32: front_launch_l<T, ALINK_LAYOUT_NCHW_F32>(p, s, xs, amax, grid, st); break;
        case ALINK_LAYOUT_NHWC_U8:  front_launch_l<T, ALINK_LAYOUT_NHWC_U8>(p, s, xs, amax, grid, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// p: conv1 as the backbone would launch it on the rolling-row kernel (p.in unused); s: the stem as launch_stem would get it
// (s.out unused); xs: [N][56][56][64] T, the stem's activation at even rows and pixels; slopes_le_1: no PReLU slope of
// either layer is above 1
hipError_t launch_front_c64(int dtype, const ConvParams& p, const StemParams& s, void* xs, bool slopes_le_1, hipStream_t st) {
    if (p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.Cin != 64 || p.Cout != 64 || p.H != H || p.W != W) return hipErrorInvalidValue;
    if (s.H != H || s.W != W || s.C0 != 64 || s.N != p.N || p.N <= 0 || !xs) return hipErrorInvalidValue;
    if (p.splitk != 1 || p.dact || p.post_relu || p.in2 || p.resid || !p.alpha || !p.border_cls) return hipErrorInvalidValue;
    if ((long long)p.N * H * W * 64 >= (1ll << 31)) return hipErrorInvalidValue;
    const long long nbands = (long long)p.N * BPI;
    const unsigned grid = (unsigned)(nbands < 256 ? nbands : 256);       // one persistent workgroup per CU
    if (dtype == ALINK_DT_BF16) return front_launch<__bf16>(p, s, xs, slopes_le_1, grid, st);
    if (dtype == ALINK_DT_F16)  return front_launch<_Float16>(p, s, xs, slopes_le_1, grid, st);
    return hipErrorInvalidValue;
}

}  // namespace alink
